// Micro-benchmark of the grouped GEMM on the 12block / imagenet224 shapes of the per-op path.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics scripts/gemm_bench.hip -o scripts/bin/gemm_bench
#include "../cora_amd/csrc/gemm.hip"
#include <cstdio>
#include <vector>

struct Shape { const char* name; int R, O, K, groups; bool bko; int mode; int a_gelu; int res; };

int main(int argc, char** argv) {
  const int R = 12800, G = 12;
  std::vector<Shape> shapes = {
      {"qkv   NT O=384 K=128 store", R, 384, 128, G, false, OUT_STORE, 0, 0},
      {"w1    NT O=512 K=128 store", R, 512, 128, G, false, OUT_STORE, 0, 0},
      {"wo    NT O=128 K=128 res1 ", R, 128, 128, G, false, OUT_STORE, 0, 1},
      {"w2    NT O=128 K=512 gelu+err", R, 128, 512, G, false, OUT_ERR, 1, 2},
      {"w2T   NN O=512 K=128 gelugrad", R, 512, 128, G, true, OUT_GELUGRAD, 0, 0},
      {"w2T   NN O=512 K=128 store", R, 512, 128, G, true, OUT_STORE, 0, 0},
      {"w1T   NN O=128 K=512 store", R, 128, 512, G, true, OUT_STORE, 0, 0},
      {"qkvT  NN O=128 K=384 store", R, 128, 384, G, true, OUT_STORE, 0, 0},
  };
  size_t big = (size_t)R * G * 512;
  float *A, *B, *C, *X1, *X2; double* red;
  hipMalloc(&A, big * 4); hipMalloc(&B, (size_t)G * 512 * 512 * 4); hipMalloc(&C, big * 4);
  hipMalloc(&X1, big * 4); hipMalloc(&X2, big * 4); hipMalloc(&red, 1024 * 8);
  hipMemset(A, 0, big * 4); hipMemset(B, 0, (size_t)G * 512 * 512 * 4); hipMemset(X1, 0, big * 4); hipMemset(X2, 0, big * 4);
  hipMemset(red, 0, 1024 * 8);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto& s : shapes) {
    GemmP p = {};
    p.A = A; p.Bw = B; p.C = C; p.R = s.R; p.O = s.O; p.K = s.K;
    p.lda = s.K; p.ldb = s.bko ? s.O : s.K; p.ldc = s.O;
    p.sA = (long long)s.R * s.K; p.sB = (long long)s.O * s.K; p.sC = (long long)s.R * s.O;
    p.groups = s.groups; p.splitk = 1; p.alpha = 1.f; p.alpha2 = 1.f; p.mode = s.mode; p.a_gelu = s.a_gelu;
    p.red_stride = 1;
    if (s.mode == OUT_GELUGRAD) p.aux = X1;
    if (s.mode == OUT_ERR) { p.tgt = X1; p.red = red; }
    if (s.res >= 1) p.res1 = X2;
    if (s.res >= 2) p.res2 = X1;
    for (int i = 0; i < 3; ++i) launch_gemm(p, false, s.bko, st);
    hipStreamSynchronize(st);
    const int it = 20;
    hipEventRecord(e0, st);
    for (int i = 0; i < it; ++i) launch_gemm(p, false, s.bko, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double us = ms * 1e3 / it, fl = 2.0 * s.R * s.groups * s.O * s.K;
    printf("%-32s %8.1f us  %6.1f TFLOP/s (%4.1f%% of 157.3)\n", s.name, us, fl / us * 1e-6, fl / us * 1e-6 / 157.3 * 100);
  }
  return 0;
}
