// How exact is a three-term bf16 split (x = hi + mid + lo, six products) on the bf16 MFMA forms?  One wave
// computes a 16x16 (or 32x32) tile of C = A * B^T, K = 64, A and B random fp32 in [-1, 1) plus a large common
// offset in C (a running accumulator ~2^16 times the small products), and compares with fp64:
//   f32      v_mfma_f32_16x16x4_f32 (the reference form of the fused kernels)
//   x3/16    six v_mfma_f32_16x16x32_bf16 per k-step into ONE accumulator (small terms first)
//   x3/16s   the same with one accumulator per magnitude class, summed in fp32 at the end
//   x3/32    six v_mfma_f32_32x32x16_bf16 per k-step into one accumulator (what gemm_bf16.hip's SPLIT = 2 does)
// Prints max and rms error in units of 2^-24 * max|C|.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/ubench/mfma_x3_precision.hip -o scripts/bin/mfma_x3_precision
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
constexpr int K = 64;

__device__ inline void split(const float* x, bf8& h, bf8& m, bf8& l) {
  for (int j = 0; j < 8; ++j) {
    h[j] = (__bf16)x[j];
    const float r = x[j] - (float)h[j];
    m[j] = (__bf16)r;
    l[j] = (__bf16)(r - (float)m[j]);
  }
}

// A [M][K], B [N][K] row-major; C [M][N]; c0 = initial accumulator value
__global__ void k_f32(const float* A, const float* B, float* C, float c0) {
  const int lane = threadIdx.x, c = lane & 15, grp = lane >> 4;
  f32x4 acc = {c0, c0, c0, c0};
  for (int s = 0; s < K / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c * K + 4 * s + grp], B[c * K + 4 * s + grp], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[(4 * grp + r) * 16 + c] = acc[r];
}
template <int SEP>
__global__ void k_x3_16(const float* A, const float* B, float* C, float c0) {
  const int lane = threadIdx.x, c = lane & 15, grp = lane >> 4;
  f32x4 acc = {c0, c0, c0, c0}, z = {0.f, 0.f, 0.f, 0.f}, a1 = z, a2 = z;
  for (int s = 0; s < K / 32; ++s) {
    bf8 ah, am, al, bh, bm, bl;
    split(A + c * K + 32 * s + 8 * grp, ah, am, al);
    split(B + c * K + 32 * s + 8 * grp, bh, bm, bl);
    f32x4& lo = SEP ? a2 : acc;
    f32x4& mi = SEP ? a1 : acc;
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, lo, 0, 0, 0);
    mi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, mi, 0, 0, 0);
    mi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, mi, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
  }
  if (SEP) acc += a1 + a2;
  for (int r = 0; r < 4; ++r) C[(4 * grp + r) * 16 + c] = acc[r];
}
__global__ void k_x3_32(const float* A, const float* B, float* C, float c0) {   // 32 x 32 tile
  const int lane = threadIdx.x, r = lane & 31, hi = lane >> 5;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = c0;
  for (int s = 0; s < K / 16; ++s) {
    bf8 ah, am, al, bh, bm, bl;
    split(A + r * K + 16 * s + 8 * hi, ah, am, al);
    split(B + r * K + 16 * s + 8 * hi, bh, bm, bl);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * hi) * 32 + r] = acc[i];   // C/D map of 32x32
}

static void report(const char* name, const std::vector<float>& C, const std::vector<float>& A, const std::vector<float>& B,
                   int M, int N, float c0) {
  double mx = 0, err = 0, sq = 0;
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      double ref = c0;
      for (int k = 0; k < K; ++k) ref += (double)A[i * K + k] * (double)B[j * K + k];
      mx = std::fmax(mx, std::fabs(ref));
      const double e = std::fabs((double)C[i * N + j] - ref);
      err = std::fmax(err, e); sq += e * e;
    }
  const double ulp = std::ldexp(mx, -24);
  printf("%-8s c0 = %-8g max err %8.2f  rms %7.2f   (units of 2^-24 max|C|)\n", name, c0, err / ulp, std::sqrt(sq / (M * N)) / ulp);
}

int main() {
  std::vector<float> A(32 * K), B(32 * K), C(32 * 32);
  srand(7);
  for (auto& v : A) v = 2.f * rand() / RAND_MAX - 1.f;
  for (auto& v : B) v = 2.f * rand() / RAND_MAX - 1.f;
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  for (float c0 : {0.f, 64.f, 4096.f}) {
    hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, dA, dB, dC, c0);
    hipMemcpy(C.data(), dC, 16 * 16 * 4, hipMemcpyDeviceToHost); report("f32", C, A, B, 16, 16, c0);
    hipLaunchKernelGGL(k_x3_16<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, c0);
    hipMemcpy(C.data(), dC, 16 * 16 * 4, hipMemcpyDeviceToHost); report("x3/16", C, A, B, 16, 16, c0);
    hipLaunchKernelGGL(k_x3_16<1>, dim3(1), dim3(64), 0, 0, dA, dB, dC, c0);
    hipMemcpy(C.data(), dC, 16 * 16 * 4, hipMemcpyDeviceToHost); report("x3/16s", C, A, B, 16, 16, c0);
    hipLaunchKernelGGL(k_x3_32, dim3(1), dim3(64), 0, 0, dA, dB, dC, c0);
    hipMemcpy(C.data(), dC, 32 * 32 * 4, hipMemcpyDeviceToHost); report("x3/32", C, A, B, 32, 32, c0);
  }
  return 0;
}
