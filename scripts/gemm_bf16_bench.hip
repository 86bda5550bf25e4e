// Micro-benchmark of the bf16-staged grouped GEMM (gemm_bf16.hip) on the imagenet224 shapes of the per-op
// path (cfg 4: N=196, D=384, M=1536, L=6, B=64: 12 544 rows per block).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics scripts/gemm_bf16_bench.hip -o scripts/bin/gemm_bf16_bench
#include "../cora_amd/csrc/gemm_bf16.hip"
#include <cstdio>
#include <vector>

struct Shape { const char* name; int O, K; int bf16; int mode; int a_gelu; int res; double bytes_per_row; };

__global__ void fill_kernel(float* x, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    x[i] = ((h & 0xffff) / 65536.0f - 0.5f) * 0.25f;
  }
}

int main(int argc, char** argv) {
  const int R = 12544, G = 6, D = 384, M = 1536;
  std::vector<Shape> shapes = {
      {"qkv  split O=1152 K=384 store  ", 3 * D, D, 2, OUT_STORE, 0, 0, 4.0 * (D + 3 * D)},
      {"wo   split O=384  K=384 res1   ", D, D, 2, OUT_STORE, 0, 1, 4.0 * 3 * D},
      {"w1   split O=1536 K=384 store  ", M, D, 2, OUT_STORE, 0, 0, 4.0 * (D + M)},
      {"w2   split O=384  K=1536 gelu+err", D, M, 2, OUT_ERR, 1, 2, 4.0 * (M + 4 * D)},
      {"w2T  plain O=1536 K=384 gelugrad", M, D, 1, OUT_GELUGRAD, 0, 0, 4.0 * (D + 2 * M)},
      {"w1T  plain O=384  K=1536 store ", D, M, 1, OUT_STORE, 0, 0, 4.0 * (M + D)},
      {"woT  plain O=384  K=384 store  ", D, D, 1, OUT_STORE, 0, 0, 4.0 * 2 * D},
      {"qkvT plain O=384  K=1152 store ", D, 3 * D, 1, OUT_STORE, 0, 0, 4.0 * 4 * D},
  };
  size_t big = (size_t)R * G * M;
  float *A, *B, *C, *X1, *X2; double* red;
  hipMalloc(&A, big * 4); hipMalloc(&B, (size_t)G * M * M * 4); hipMalloc(&C, big * 4);
  hipMalloc(&X1, big * 4); hipMalloc(&X2, big * 4); hipMalloc(&red, 1024 * 8);
  fill_kernel<<<2048, 256>>>(A, big, 1); fill_kernel<<<2048, 256>>>(B, (size_t)G * M * M, 2);
  fill_kernel<<<2048, 256>>>(X1, big, 3); fill_kernel<<<2048, 256>>>(X2, big, 4);
  hipMemset(red, 0, 1024 * 8);
  unsigned short *Bh, *Bl;      // pre-converted weights (timing only: contents do not matter here)
  hipMalloc(&Bh, (size_t)G * M * M * 2); hipMalloc(&Bl, (size_t)G * M * M * 2);
  hipMemset(Bh, 0x3c, (size_t)G * M * M * 2); hipMemset(Bl, 0x38, (size_t)G * M * M * 2);
  const bool pre = argc > 1;    // any argument: B operand pre-converted
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double tot = 0;
  for (auto& s : shapes) {
    GemmP p = {};
    p.A = A; p.Bw = B; p.C = C; p.R = R; p.O = s.O; p.K = s.K;
    p.lda = s.K; p.ldb = s.K; p.ldc = s.O;
    p.sA = (long long)R * s.K; p.sB = (long long)s.O * s.K; p.sC = (long long)R * s.O;
    p.groups = G; p.splitk = 1; p.alpha = 1.f; p.alpha2 = 1.f; p.mode = s.mode; p.a_gelu = s.a_gelu;
    p.red_stride = 1; p.bf16 = s.bf16;
    if (pre) { p.Bh = Bh; p.Bl = Bl; }
    if (pre && argc > 2 && s.bf16 == 1 && s.mode == OUT_GELUGRAD) p.c16 = 1;   // ... and the gelu' epilogue writes bf16 (da)
    if (pre && argc > 2 && s.bf16 == 1 && s.mode == OUT_STORE) p.a16 = 1;   // second argument: plain GEMMs read A as bf16 (ctx.hip: da16 / dqkv16)
    if (s.mode == OUT_GELUGRAD) p.aux = X1;
    if (s.mode == OUT_ERR) { p.tgt = X1; p.red = red; }
    if (s.res >= 1) p.res1 = X2;
    if (s.res >= 2) p.res2 = X1;
    for (int i = 0; i < 3; ++i) launch_gemm_bf16_nt(p, st);
    hipStreamSynchronize(st);
    const int it = 20;
    hipEventRecord(e0, st);
    for (int i = 0; i < it; ++i) launch_gemm_bf16_nt(p, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double us = ms * 1e3 / it, fl = 2.0 * R * G * (double)s.O * s.K;
    tot += us;
    printf("%-34s %8.1f us  %6.1f TFLOP/s   HBM floor %6.1f us (%.2f TB/s achieved)\n", s.name, us, fl / us * 1e-6,
           s.bytes_per_row * R * G / 5.5e6, s.bytes_per_row * R * G / us * 1e-6);
  }
  printf("total %.1f us\n", tot);
  return 0;
}
