// Context number for scripts/gemm_bf16_bench.hip (NOT part of the product): the vendor library's bf16 GEMM
// (rocblas_gemm_ex: bf16 operands ALREADY in memory, fp32 output, fp32 accumulation, no epilogue, no operand split) on the
// eight GEMM shapes of one imagenet224 inference step (cfg 4: 12 544 rows per block, 6 blocks as a strided batch).
// Our kernel reads fp32 activations (twice the A bytes), rounds / splits them on the way to LDS and carries the step's
// epilogues (residuals, gelu', error + energy), so the two are not like for like: this is the library's rate on the shapes.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/rocblas_bf16_ref.hip -lrocblas -o scripts/bin/rocblas_bf16_ref
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>

struct Shape { const char* name; int O, K; };

int main() {
  const int R = 12544, G = 6, D = 384, M = 1536;
  std::vector<Shape> shapes = {{"qkv  O=1152 K=384 ", 3 * D, D}, {"wo   O=384  K=384 ", D, D},     {"w1   O=1536 K=384 ", M, D},
                               {"w2   O=384  K=1536", D, M},     {"w2T  O=1536 K=384 ", M, D},     {"w1T  O=384  K=1536", D, M},
                               {"woT  O=384  K=384 ", D, D},     {"qkvT O=384  K=1152", D, 3 * D}};
  const size_t big = (size_t)R * G * M;
  void *A, *B, *C;
  hipMalloc(&A, big * 2); hipMalloc(&B, (size_t)G * M * M * 2); hipMalloc(&C, big * 4);
  hipMemset(A, 0x3c, big * 2); hipMemset(B, 0x3c, (size_t)G * M * M * 2);
  rocblas_handle h; rocblas_create_handle(&h);
  hipStream_t st; hipStreamCreate(&st); rocblas_set_stream(h, st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const float alpha = 1.f, beta = 0.f;
  double tot = 0;
  for (auto& s : shapes) {
    // row-major C[R][O] = A[R][K] * W[O][K]^T  ==  column-major C^T[O][R] = W^T(op T on [K][O] col-major) * A^T
    auto run = [&]() {
      return rocblas_gemm_strided_batched_ex(h, rocblas_operation_transpose, rocblas_operation_none, s.O, R, s.K, &alpha,
                                             B, rocblas_datatype_bf16_r, s.K, (rocblas_stride)s.O * s.K,
                                             A, rocblas_datatype_bf16_r, s.K, (rocblas_stride)R * s.K, &beta,
                                             C, rocblas_datatype_f32_r, s.O, (rocblas_stride)R * s.O,
                                             C, rocblas_datatype_f32_r, s.O, (rocblas_stride)R * s.O, G,
                                             rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
    };
    rocblas_status rs = rocblas_status_success;
    for (int i = 0; i < 3; ++i) rs = run();
    hipStreamSynchronize(st);
    if (rs != rocblas_status_success) { printf("%s rocblas status %d\n", s.name, (int)rs); continue; }
    const int it = 20;
    hipEventRecord(e0, st);
    for (int i = 0; i < it; ++i) run();
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / it, fl = 2.0 * R * G * (double)s.O * s.K;
    const double bytes = (double)R * G * (2.0 * s.K + 4.0 * s.O);
    tot += us;
    printf("%-20s %8.1f us  %7.1f TFLOP/s   (bf16 A in, fp32 C out: %.2f TB/s)\n", s.name, us, fl / us * 1e-6, bytes / us * 1e-6);
  }
  printf("total %.1f us\n", tot);
  return 0;
}
