// Microbenchmark: cycles per v_mfma_f32_16x16x4_f32 from ONE wave per SIMD (256-thread block per CU),
// 4 interleaved accumulators, operands in registers; variants add LDS reads / VALU between the MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int VAR>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[64 * 132];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int j = tid; j < 64 * 132; j += 256) lds[j] = (float)(j % 7) * 0.25f;
  __syncthreads();
  f32x4 acc[4];
  for (int a = 0; a < 4; ++a) acc[a] = (f32x4){0, 0, 0, 0};
  float b[4] = {1.0f + lane, 0.5f, 0.25f, 2.0f};
  float4 av[4];
  const float* tl = lds + (lane & 15) * 132 + 4 * (lane >> 4);
  for (int ot = 0; ot < 4; ++ot) av[ot] = *reinterpret_cast<const float4*>(tl + 16 * ot * 132);
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  if (VAR == 2 || VAR == 4) {
    // ping-pong operand sets, no copies: NR ds_read_b128 per 16 MFMAs
    float4 bv[4];
    for (int ot = 0; ot < 4; ++ot) bv[ot] = av[ot];
    constexpr int NR = (VAR == 2) ? 4 : 2;
    for (int it = 0; it < iters; it += 2) {
#define GROUP(CUR, NXT, IT)                                                                      \
      _Pragma("unroll") for (int ot = 0; ot < NR; ++ot) NXT[ot] = *reinterpret_cast<const float4*>(tl + 16 * ot * 132 + 16 * (((IT) + 1) & 7)); \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      _Pragma("unroll") for (int ot = 0; ot < 4; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(CUR[ot].x, b[0], acc[ot], 0, 0, 0); \
      _Pragma("unroll") for (int ot = 0; ot < 4; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(CUR[ot].y, b[1], acc[ot], 0, 0, 0); \
      _Pragma("unroll") for (int ot = 0; ot < 4; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(CUR[ot].z, b[2], acc[ot], 0, 0, 0); \
      _Pragma("unroll") for (int ot = 0; ot < 4; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(CUR[ot].w, b[3], acc[ot], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);
      GROUP(av, bv, it)
      GROUP(bv, av, it + 1)
    }
  } else
  for (int it = 0; it < iters; ++it) {
    float4 nv[4];
    if (VAR == 3) { for (int ot = 0; ot < 4; ++ot) { nv[ot] = av[ot]; asm volatile("" : "+v"(nv[ot].x), "+v"(nv[ot].y), "+v"(nv[ot].z), "+v"(nv[ot].w)); } }
    if (VAR == 1) {
      for (int ot = 0; ot < 4; ++ot) nv[ot] = *reinterpret_cast<const float4*>(tl + 16 * ot * 132 + 16 * ((it + 1) & 7));
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ot].x, b[0], acc[ot], 0, 0, 0);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ot].y, b[1], acc[ot], 0, 0, 0);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ot].z, b[2], acc[ot], 0, 0, 0);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ot].w, b[3], acc[ot], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (VAR == 1 || VAR == 3) for (int ot = 0; ot < 4; ++ot) av[ot] = nv[ot];
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  float s = 0;
  for (int a = 0; a < 4; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  const int nb = 256, iters = 4000;
  hipMalloc(&out, nb * 256 * 4); hipMalloc(&cyc, nb * 8);
  std::vector<unsigned long long> h(nb);
  for (int var = 0; var < 5; ++var) {
    for (int blocks : {256}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (var == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        else if (var == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        else if (var == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        else if (var == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
      double c = 0; for (int i = 0; i < blocks; ++i) c += (double)h[i]; c /= blocks;
      printf("variant %d blocks %3d: %.1f cycles per MFMA (s_memtime), kernel %.3f ms -> %.2f ns per MFMA -> implied clock %.2f GHz, %.1f TFLOP/s chip-equivalent\n",
             var, blocks, c / (16.0 * iters), ms, ms * 1e6 / (16.0 * iters), (c / (16.0 * iters)) / (ms * 1e6 / (16.0 * iters)),
             blocks * 4.0 * 16.0 * iters * 2048 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
