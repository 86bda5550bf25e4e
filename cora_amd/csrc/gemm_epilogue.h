// gemm_epilogue.h -- the fused PC epilogues of the GEMM kernels (gemm.hip: fp32 / in-register bf16;
// gemm_bf16.hip: bf16-staged), one 32x32 MFMA accumulator tile at a time.
#pragma once
#include "common.h"

struct GemmEp {
  float* C;
  const float *bias, *res1, *res2, *tgt, *hreg, *aux;
};

__device__ __forceinline__ GemmEp gemm_ep_ptrs(const GemmP& p, int g) {
  const long long offC = (long long)g * p.sC;
  GemmEp q;
  q.C = p.C + offC;
  q.bias = p.bias ? p.bias + (long long)g * p.sBias : nullptr;
  q.res1 = p.res1 ? p.res1 + offC : nullptr;
  q.res2 = p.res2 ? p.res2 + offC : nullptr;
  q.tgt = p.tgt ? p.tgt + offC : nullptr;
  q.hreg = p.hreg ? p.hreg + offC : nullptr;
  q.aux = p.aux ? p.aux + offC : nullptr;
  return q;
}

// acc: the f32x16 result of a 32x32 MFMA chain; lane (r = lane & 31, hi = lane >> 5) holds column `col`
// of rows rbase + (i & 3) + 8 (i >> 2), i < 16, with rbase = first tile row + 4 hi.
// Every operand the epilogue reads is fetched for all 16 values BEFORE the first store: the compiler
// must assume C aliases them, and a load -> wait -> store chain per element costs one exposed HBM
// latency per element.  Offsets are 32-bit (the launcher checks R * ldc < 2^31) to keep the batch's
// address registers from pushing the kernel below its occupancy.
// STREAM: C and the read-once epilogue operands bypass the cache hierarchy's retention (nontemporal): a
// GEMM whose operands are re-read from L2 by neighbouring workgroups should not have its output stream
// evict them.
template <bool STREAM = false>
__device__ __forceinline__ void gemm_ep_tile(const GemmP& p, const GemmEp& q, const f32x16& acc, const int rbase, const int col, double& red) {
  const bool colok = col < p.O;
  const float bv = (q.bias && colok) ? q.bias[col] : 0.f;
  const int obase = rbase * p.ldc + col;
  float* C = q.C;
#define EP_ROW(i) (rbase + ((i) & 3) + 8 * ((i) >> 2))
#define EP_OK(i) (colok && EP_ROW(i) < p.R)
#define EP_IDX(i) (obase + (((i) & 3) + 8 * ((i) >> 2)) * p.ldc)
#define EP_FETCH(dst, src)                                                                   \
  _Pragma("unroll") for (int i = 0; i < 16; ++i) dst[i] = EP_OK(i) ? EP_LD(&(src)[EP_IDX(i)]) : 0.f
#define EP_LD(ptr) (STREAM ? __builtin_nontemporal_load(ptr) : *(ptr))
#define EP_ST(ptr, val) do { if (STREAM) __builtin_nontemporal_store((val), (ptr)); else *(ptr) = (val); } while (0)
  float v[16], t0[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = p.alpha * acc[i] + bv;
  if (p.pos) {
#pragma unroll
    for (int i = 0; i < 16; ++i) t0[i] = EP_OK(i) ? p.pos[(EP_ROW(i) % p.posN) * p.O + col] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] += t0[i];
  }
  if (q.res1) {
    EP_FETCH(t0, q.res1);
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] += t0[i];
  }
  if (q.res2) {
    EP_FETCH(t0, q.res2);
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] += t0[i];
  }
  switch (p.mode) {
    case OUT_STORE:
#pragma unroll
      for (int i = 0; i < 16; ++i) if (EP_OK(i)) EP_ST(&C[EP_IDX(i)], v[i]);
      break;
    case OUT_ERR: {
      EP_FETCH(t0, q.tgt);
#pragma unroll
      for (int i = 0; i < 16; ++i) if (EP_OK(i)) {
        const float h = t0[i], e = h - v[i];
        EP_ST(&C[EP_IDX(i)], e);
        red += (double)(0.5f * e * e + p.l1 * fabsf(h) + p.l2 * h * h);
      }
    } break;
    case OUT_GELUGRAD: {
      EP_FETCH(t0, q.aux);
#ifdef EXP_NOGELUGRAD   /* timing experiment: gelu'(a) as if left by the MLP1 epilogue */
#pragma unroll
      for (int i = 0; i < 16; ++i) if (EP_OK(i)) EP_ST(&C[EP_IDX(i)], v[i] * t0[i]);
#else
      if (p.c16) {
        // da kept as bf16 for the next GEMM (GemmP::c16): the value that GEMM would round it to.  Neighbouring lanes hold
        // neighbouring columns: they pair up, the even lane stores rows 0..7 of both columns, the odd lane rows 8..15 --
        // 8 four-byte stores per lane instead of 16 two-byte ones (W2^T GEMM of the cfg 4 shape: 333 -> 290 us).
        // O and ldc are even (launcher), so a pair is inside or outside as a whole.
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        __bf16* C16 = reinterpret_cast<__bf16*>(p.C) + (q.C - p.C);
        const bool odd = (col & 1) != 0;
        const int cole = col & ~1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float lo = v[i] * cora_gelu_grad(t0[i]), hi = v[i + 8] * cora_gelu_grad(t0[i + 8]);
          const float got = __shfl_xor(odd ? lo : hi, 1, 64);
          const int ii = odd ? i + 8 : i;
          bf2 pr;
          pr[0] = (__bf16)(odd ? got : lo);      // even column
          pr[1] = (__bf16)(odd ? hi : got);      // odd column
          if (cole < p.O && EP_ROW(ii) < p.R) {
            bf2* dst = reinterpret_cast<bf2*>(&C16[EP_ROW(ii) * p.ldc + cole]);
            if (STREAM) __builtin_nontemporal_store(pr, dst); else *dst = pr;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) if (EP_OK(i)) EP_ST(&C[EP_IDX(i)], v[i] * cora_gelu_grad(t0[i]));
      }
#endif
    } break;
    case OUT_GRAD: {
      float hr[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { t0[i] = 0.f; hr[i] = 0.f; }
      if (q.tgt) { EP_FETCH(t0, q.tgt); }
      if (q.hreg) { EP_FETCH(hr, q.hreg); }
#pragma unroll
      for (int i = 0; i < 16; ++i) if (EP_OK(i)) {
        const float h = hr[i];
        const float sg = (h > 0.f) ? 1.f : ((h < 0.f) ? -1.f : 0.f);
        const float gv = p.alpha2 * (t0[i] + p.l1 * sg + 2.f * p.l2 * h - v[i]);
        EP_ST(&C[EP_IDX(i)], gv);
        red += (double)gv * (double)gv;
      }
    } break;
    case OUT_ATOMIC:
#pragma unroll
      for (int i = 0; i < 16; ++i) if (EP_OK(i)) atomicAdd(&C[EP_IDX(i)], v[i]);
      break;
  }
#undef EP_ROW
#undef EP_OK
#undef EP_IDX
#undef EP_FETCH
#undef EP_LD
#undef EP_ST
}

// energy / sum-of-squares of the whole workgroup -> p.red (OUT_GRAD: 8 slots per vode, folded on the host)
__device__ __forceinline__ void gemm_ep_reduce(const GemmP& p, int g, int bx, double red, double* red_sh) {
  if (p.red) {
    double tot = block_sum_256(red, red_sh);
    if (threadIdx.x == 0) {
      if (p.mode == OUT_GRAD) atomicAdd(&p.red[((long long)g * p.red_stride) * 8 + (bx & 7)], tot);
      else atomicAdd(&p.red[(long long)g * p.red_stride], tot);
    }
  }
}
