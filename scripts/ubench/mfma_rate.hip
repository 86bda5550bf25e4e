// Throughput of the MFMA forms the attention / fused kernels use or could use (gfx950): a grid that fills
// every SIMD several times over (2048 workgroups of 4 waves), four independent accumulators per wave,
// aggregate TFLOP/s from HIP events (workgroup placement decides per-wave rates, so only the aggregate is used).
// Build: hipcc -O3 --offload-arch=gfx950 scripts/ubench/mfma_rate.hip -o scripts/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, unsigned long long* cyc, int iters) {
  // four accumulators with different operands, so the compiler cannot fold them into one chain
  f32x4 acc[4];
  f32x16 big[4];
  for (int i = 0; i < 4; ++i) {
    acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
  }
  float a[4], b = 0.5f;
  s16x4 a4[4], b4 = {3, 2, 1, (short)threadIdx.x};
  bf16x8 a8[4], b8;
  for (int i = 0; i < 4; ++i) {
    a[i] = 1.0f + threadIdx.x * 1e-3f + i;
    a4[i] = (s16x4){(short)(threadIdx.x + i), 1, 2, (short)i};
    for (int j = 0; j < 8; ++j) a8[i][j] = (__bf16)(float)(i + j + threadIdx.x % 3);
  }
  for (int j = 0; j < 8; ++j) b8[j] = (__bf16)(float)(j + 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b, acc[i], 0, 0, 0);
      if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4[i], b4, acc[i], 0, 0, 0);
      if (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[i], b8, acc[i], 0, 0, 0);
      if (KIND == 3) big[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b, big[i], 0, 0, 0);
      if (KIND == 4) big[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8, big[i], 0, 0, 0);
    }
  }
  (void)cyc;
  float s = 0.f;
  for (int i = 0; i < 4; ++i) {
    s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int j = 0; j < 16; ++j) s += big[i][j];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, double flop_per_inst) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 2048 * 256 * sizeof(float)); hipMalloc(&cyc, 8);
  const int iters = 1500;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((rate_kernel<KIND>), dim3(2048), dim3(256), 0, 0, out, cyc, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((rate_kernel<KIND>), dim3(2048), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double inst = 16.0 * iters;                      // per wave
  const double tflops = flop_per_inst * inst * 2048 * 4 / (ms * 1e-3) / 1e12;
  // per SIMD: 2048 * 4 waves / 1024 SIMDs = 8 waves' instructions share one pipe
  // 8192 waves, 8 per SIMD, 16 MFMAs per loop trip (with 4 per trip the loop branch costs the 16x16x4 form 17 %)
  printf("%-28s %.3f ms  %.1f TFLOP/s aggregate  %.2f ns per instruction per SIMD\n", name, ms, tflops, ms * 1e6 / (inst * 8));
  (void)c;
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("v_mfma_f32_16x16x4_f32", 2.0 * 16 * 16 * 4);
  run<1>("v_mfma_f32_16x16x16_bf16", 2.0 * 16 * 16 * 16);
  run<2>("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32);
  run<3>("v_mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2);
  run<4>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16);
  return 0;
}
