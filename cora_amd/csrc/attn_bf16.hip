// attn_bf16.hip -- self-attention BACKWARD on bf16 MFMA (v_mfma_f32_16x16x16_bf16) for the opt-in bf16 mode.
// Same math, same workgroup decomposition and the same lane maps as attn_bwd_q / attn_bwd_kv of attn.hip
// (eqx.nn.MultiheadAttention backward as used by /root/reference/pcx/nn/_layer.py:359-402); HBM tensors,
// row constants (lse2, delta), the softmax recompute and every accumulator stay fp32.  Operands are rounded
// to bf16 once, on the way from the staging registers into LDS (own-row fragments: once per workgroup), as
// the bf16 GEMMs of this mode do for the other backward-data products.  The forward keeps the fp32 kernel:
// its output feeds e = h - u, which cancels most of u.
//
// Operand maps (lane l, c = l & 15, grp = l >> 4; one MFMA contracts 16 values of k, 4 per lane):
//   score  tile  D[i][j] = sum_dd X[i][dd] Y[j][dd] : A = X[i = c][16 s + 4 grp + t]  (row-major LDS image, b64)
//                                                     B = Y[j = c][16 s + 4 grp + t]  (own fragment, registers)
//                lane holds D[i = 4 grp + reg][j = c]
//   accum  tile  D[cc][j] += sum_i Z[i][cc] W[i][j] : A = Z[i = 4 grp + t][cc = c]    (TRANSPOSED LDS image, b64)
//                                                     B = W[i = 4 grp + t][j = c]     = the score tile's own
//                                                         four registers, rounded to bf16 in place
// so a chunk of the streamed side lives in LDS twice where both products read it: [row][dd] (stride DH + 8
// bf16: 16 rows x 4 dwords tile the 64 banks of a ds_read_b64 half) and [dd][row] (stride 68: the staging's
// b32 stores of row pairs are conflict-free, the b64 reads two-way on one bank pair).
#include "common.h"
#include "attn_map.h"

namespace {

#define CHB 64  // streamed rows per chunk

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_bf(const bf16x4 a, const bf16x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ bf16x4 cvt4(const float4 v) {
  bf16x4 h;
  h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
  return h;
}
__device__ __forceinline__ bf16x4 cvt4(const f32x4 v) {
  bf16x4 h;
  h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
  return h;
}

template <int DH> struct Img {
  static constexpr int LR = DH + 8;   // bf16 per row of the row-major image
  static constexpr int LT = 68;       // bf16 per row of the transposed image (64 streamed rows + 4)
  static constexpr int NR = CHB * LR, NT = DH * LT;
};

// Stage rows [r0, r0 + 64) x [0, DH) of two fp32 matrices as bf16.  A unit is 2 rows x 4 columns (two
// float4 loads); 64 consecutive units = 16 row pairs x 4 column quads (64 B of a row per 4 lanes), which
// keeps the b32 stores into the transposed image on 32 different banks.  R / T may be null (image not needed).
template <int DH, int NTH>
__device__ __forceinline__ void stage2(__bf16* R0, __bf16* T0, const float* src0, int ld0,
                                       __bf16* R1, __bf16* T1, const float* src1, int ld1, int r0, int nrows) {
  constexpr int LR = Img<DH>::LR, LT = Img<DH>::LT;
  constexpr int NU = 8 * DH;           // units per matrix (a multiple of 64: the matrix index is wave-uniform)
#pragma unroll
  for (int u0 = 0; u0 < 2 * NU; u0 += NTH) {
    const int u = u0 + (int)threadIdx.x;
    if (2 * NU % NTH != 0 && u >= 2 * NU) break;
    const int m = u >= NU;
    const int v = m ? u - NU : u;
    const int wl = v >> 6, i = v & 63;
    const int c4 = 4 * (wl >> 1) + (i & 3);
    const int row = 2 * (16 * (wl & 1) + (i >> 2));
    const float* src = m ? src1 : src0;
    const int ld = m ? ld1 : ld0;
    __bf16* R = m ? R1 : R0;
    __bf16* T = m ? T1 : T0;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (r0 + row < nrows) a = *reinterpret_cast<const float4*>(src + (long long)(r0 + row) * ld + 4 * c4);
    if (r0 + row + 1 < nrows) b = *reinterpret_cast<const float4*>(src + (long long)(r0 + row + 1) * ld + 4 * c4);
    const bf16x4 ha = cvt4(a), hb = cvt4(b);
    if (R) {
      *reinterpret_cast<bf16x4*>(&R[row * LR + 4 * c4]) = ha;
      *reinterpret_cast<bf16x4*>(&R[(row + 1) * LR + 4 * c4]) = hb;
    }
    if (T) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x2 pr;
        pr[0] = ha[j]; pr[1] = hb[j];
        *reinterpret_cast<bf16x2*>(&T[(4 * c4 + j) * LT + row]) = pr;
      }
    }
  }
}

// own-row fragment: f[s] = X[row][16 s + 4 grp .. + 3] as bf16
template <int DH>
__device__ __forceinline__ void load_frag_bf(bf16x4* f, const float* src, int ld, int row, int nrows, int grp) {
#pragma unroll
  for (int s = 0; s < DH / 16; ++s) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows) v = *reinterpret_cast<const float4*>(src + (long long)row * ld + 16 * s + 4 * grp);
    f[s] = cvt4(v);
  }
}

// S = streamed rows 16 tt .. 16 tt + 15 (row-major image) x own fragment
template <int DH>
__device__ __forceinline__ f32x4 score_bf(const __bf16* R, const bf16x4* frag, int tt, int c, int grp) {
  constexpr int LR = Img<DH>::LR;
  f32x4 S = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < DH / 16; ++s)
    S = mfma_bf(*reinterpret_cast<const bf16x4*>(&R[(16 * tt + c) * LR + 16 * s + 4 * grp]), frag[s], S);
  return S;
}
// acc[ct] += sum over streamed rows 16 tt .. + 15 of Z[row][16 ct + c] * w[row][own]  (Z from the transposed image)
template <int DH>
__device__ __forceinline__ void accum_bf(f32x4* acc, const __bf16* T, const bf16x4 w, int tt, int c, int grp) {
  constexpr int LT = Img<DH>::LT;
#pragma unroll
  for (int ct = 0; ct < DH / 16; ++ct)
    acc[ct] = mfma_bf(*reinterpret_cast<const bf16x4*>(&T[(16 * ct + c) * LT + 16 * tt + 4 * grp]), w, acc[ct]);
}

// ---------------------------------------------------------------------------------------------
// own = queries (+ dO): streams K (both images) and V (row-major); dQ, and delta = rowsum(dO * O) for bwd-kv
template <int DH> struct SmemQ {
  __bf16 Kr[Img<DH>::NR], Vr[Img<DH>::NR], Kt[Img<DH>::NT];
};
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, DH <= 64 ? 4 : 1) void attn_bwd_q_bf16_kernel(
    const float* __restrict__ qkv, const float* __restrict__ O, const float* __restrict__ lse2, const float* __restrict__ dO,
    float* __restrict__ delta, float* __restrict__ dqkv, int nimg, int N, int H, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SmemQ<DH>& sm = *reinterpret_cast<SmemQ<DH>*>(smem_raw);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, grp = lane >> 4;
  int rb, h, img;
  if (!attn_coords((N + 16 * NW - 1) / (16 * NW), H, nimg, rb, h, img)) return;
  const int ld = 3 * D;
  const float* base = qkv + (long long)img * N * ld;
  const float* Q = base + h * DH;
  const float* K = base + D + h * DH;
  const float* V = base + 2 * D + h * DH;
  const float* dOp = dO + (long long)img * N * D + h * DH;
  const float* Op = O + (long long)img * N * D + h * DH;
  const int qrow = rb * 16 * NW + 16 * w + c;
  const float scale = rsqrtf((float)DH);
  const float sc = scale * CORA_LOG2E;

  bf16x4 qf[DH / 16], dof[DH / 16];
  load_frag_bf<DH>(qf, Q, ld, qrow, N, grp);
  float dl = 0.f;
#pragma unroll
  for (int s = 0; s < DH / 16; ++s) {   // delta in fp32, from the fp32 dO and O
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (qrow < N) {
      a = *reinterpret_cast<const float4*>(dOp + (long long)qrow * D + 16 * s + 4 * grp);
      b = *reinterpret_cast<const float4*>(Op + (long long)qrow * D + 16 * s + 4 * grp);
    }
    dl += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    dof[s] = cvt4(a);
  }
  dl = grp_sum(dl);
  const float l2 = (qrow < N) ? lse2[((long long)img * N + qrow) * H + h] : 0.f;
  if (qrow < N && grp == 0) delta[((long long)img * N + qrow) * H + h] = dl;

  f32x4 acc[DH / 16];
#pragma unroll
  for (int ct = 0; ct < DH / 16; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool wave_active = rb * 16 * NW + 16 * w < N;
  for (int k0 = 0; k0 < N; k0 += CHB) {
    __syncthreads();
    stage2<DH, 64 * NW>(sm.Kr, sm.Kt, K, ld, sm.Vr, nullptr, V, ld, k0, N);
    __syncthreads();
    if (!wave_active) continue;
    // keys past the sequence are staged as zeros: their K column of the accumulation is zero, no mask needed
    const int nvt = min(4, (N - k0 + 15) >> 4);
#pragma unroll 1
    for (int tt = 0; tt < nvt; ++tt) {
      const f32x4 S = score_bf<DH>(sm.Kr, qf, tt, c, grp);     // S[key][query]
      const f32x4 dP = score_bf<DH>(sm.Vr, dof, tt, c, grp);   // dP[key][query] = V[key] . dO[query]
      f32x4 dS;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) dS[reg] = __builtin_amdgcn_exp2f(S[reg] * sc - l2) * (dP[reg] - dl);
      accum_bf<DH>(acc, sm.Kt, cvt4(dS), tt, c, grp);          // dQ^T += K^T dS
    }
  }
  if (qrow < N) {
    float* drow = dqkv + ((long long)img * N + qrow) * ld + h * DH;
#pragma unroll
    for (int ct = 0; ct < DH / 16; ++ct)
      *reinterpret_cast<float4*>(drow + 16 * ct + 4 * grp) =
          make_float4(acc[ct][0] * scale, acc[ct][1] * scale, acc[ct][2] * scale, acc[ct][3] * scale);
  }
}

// ---------------------------------------------------------------------------------------------
// own = keys (+ V): streams Q and dO (both images each) with their row constants; dK, dV
template <int DH> struct SmemKV {
  __bf16 Qr[Img<DH>::NR], dOr[Img<DH>::NR], Qt[Img<DH>::NT], dOt[Img<DH>::NT];
  float rowc[2 * CHB];   // lse2, delta of the streamed query rows
};
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, 1) void attn_bwd_kv_bf16_kernel(
    const float* __restrict__ qkv, const float* __restrict__ lse2, const float* __restrict__ dO, const float* __restrict__ delta,
    float* __restrict__ dqkv, int nimg, int N, int H, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SmemKV<DH>& sm = *reinterpret_cast<SmemKV<DH>*>(smem_raw);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, grp = lane >> 4;
  int rb, h, img;
  if (!attn_coords((N + 16 * NW - 1) / (16 * NW), H, nimg, rb, h, img)) return;
  const int ld = 3 * D;
  const float* base = qkv + (long long)img * N * ld;
  const float* Q = base + h * DH;
  const float* K = base + D + h * DH;
  const float* V = base + 2 * D + h * DH;
  const float* dOp = dO + (long long)img * N * D + h * DH;
  const int krow = rb * 16 * NW + 16 * w + c;
  const float scale = rsqrtf((float)DH);
  const float sc = scale * CORA_LOG2E;

  bf16x4 kf[DH / 16], vf[DH / 16];
  load_frag_bf<DH>(kf, K, ld, krow, N, grp);
  load_frag_bf<DH>(vf, V, ld, krow, N, grp);
  f32x4 dk[DH / 16], dv[DH / 16];
#pragma unroll
  for (int ct = 0; ct < DH / 16; ++ct) {
    dk[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dv[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const bool wave_active = rb * 16 * NW + 16 * w < N;
  for (int q0 = 0; q0 < N; q0 += CHB) {
    __syncthreads();
    stage2<DH, 64 * NW>(sm.Qr, sm.Qt, Q, ld, sm.dOr, sm.dOt, dOp, D, q0, N);
    if (threadIdx.x < CHB) {
      const int q = q0 + threadIdx.x;
      const long long ri = ((long long)img * N + q) * H + h;
      sm.rowc[threadIdx.x] = (q < N) ? lse2[ri] : 0.f;
      sm.rowc[CHB + threadIdx.x] = (q < N) ? delta[ri] : 0.f;
    }
    __syncthreads();
    if (!wave_active) continue;
    // query rows past the sequence are staged as zeros (Q, dO, row constants): P = 1 meets dO = 0 and
    // dS = 1 * (0 - 0), so they add nothing and need no mask
    const int nvt = min(4, (N - q0 + 15) >> 4);
#pragma unroll 1
    for (int tt = 0; tt < nvt; ++tt) {
      const f32x4 S = score_bf<DH>(sm.Qr, kf, tt, c, grp);     // S[q][key]
      const f32x4 dP = score_bf<DH>(sm.dOr, vf, tt, c, grp);   // dP[q][key] = dO[q] . V[key]
      const f32x4 l2 = *reinterpret_cast<const f32x4*>(&sm.rowc[16 * tt + 4 * grp]);
      const f32x4 dl = *reinterpret_cast<const f32x4*>(&sm.rowc[CHB + 16 * tt + 4 * grp]);
      f32x4 P, dS;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        P[reg] = __builtin_amdgcn_exp2f(S[reg] * sc - l2[reg]);
        dS[reg] = P[reg] * (dP[reg] - dl[reg]);
      }
      accum_bf<DH>(dv, sm.dOt, cvt4(P), tt, c, grp);    // dV^T += dO^T P
      accum_bf<DH>(dk, sm.Qt, cvt4(dS), tt, c, grp);    // dK^T += Q^T dS
    }
  }
  if (krow < N) {
    float* drow = dqkv + ((long long)img * N + krow) * ld + h * DH;
#pragma unroll
    for (int ct = 0; ct < DH / 16; ++ct) {
      *reinterpret_cast<float4*>(drow + D + 16 * ct + 4 * grp) =
          make_float4(dk[ct][0] * scale, dk[ct][1] * scale, dk[ct][2] * scale, dk[ct][3] * scale);
      *reinterpret_cast<float4*>(drow + 2 * D + 16 * ct + 4 * grp) = make_float4(dv[ct][0], dv[ct][1], dv[ct][2], dv[ct][3]);
    }
  }
}

template <int DH, int NW>
int run_bwd_bf16(const float* qkv, const float* O, const float* lse2, const float* dO, float* delta,
                 float* dqkv, int nimg, int N, int H, int D, hipStream_t st) {
  const size_t smq = sizeof(SmemQ<DH>), smkv = sizeof(SmemKV<DH>);
  static bool attr_set[64] = {};
  if (smkv > 48 * 1024 && first_use_on_device(attr_set)) {
    if (hipFuncSetAttribute((const void*)attn_bwd_q_bf16_kernel<DH, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smq) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void*)attn_bwd_kv_bf16_kernel<DH, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smkv) != hipSuccess) return -1;
  }
  const int nrb = (N + 16 * NW - 1) / (16 * NW);
  const dim3 grid(attn_grid(nrb, H, nimg));
  hipLaunchKernelGGL((attn_bwd_q_bf16_kernel<DH, NW>), grid, dim3(64 * NW), smq, st, qkv, O, lse2, dO, delta, dqkv, nimg, N, H, D);
  hipLaunchKernelGGL((attn_bwd_kv_bf16_kernel<DH, NW>), grid, dim3(64 * NW), smkv, st, qkv, lse2, dO, delta, dqkv, nimg, N, H, D);
  return 0;
}

template <int DH>
int run_bwd_bf16_dh(const float* qkv, const float* O, const float* lse2, const float* dO, float* delta,
                    float* dqkv, int nimg, int N, int H, int D, hipStream_t st) {
  return attn_waves(nimg, N, H, true) == 8 ? run_bwd_bf16<DH, 8>(qkv, O, lse2, dO, delta, dqkv, nimg, N, H, D, st)
                                           : run_bwd_bf16<DH, 4>(qkv, O, lse2, dO, delta, dqkv, nimg, N, H, D, st);
}

}  // namespace

// head dims that are a multiple of 16, sequences longer than one chunk (shorter ones keep the fused fp32
// backward of attn.hip, which is HBM-bound); anything else: false, the caller runs the fp32 kernels
bool attn_bwd_bf16_supported(int N, int H, int D) {
  const int dh = D / H;
  if (getenv("CORA_BF16_FP32_ATTN")) return false;
  return N > CHB && (dh == 16 || dh == 32 || dh == 48 || dh == 64 || dh == 96 || dh == 128);
}

int launch_attn_bwd_bf16(const float* qkv, const float* O, const float* lse2, const float* dO, float* delta,
                         float* dqkv, int nimg, int N, int H, int D, hipStream_t st) {
  if (!attn_bwd_bf16_supported(N, H, D)) return -1;
#define CALL(X) run_bwd_bf16_dh<X>(qkv, O, lse2, dO, delta, dqkv, nimg, N, H, D, st)
  switch (D / H) {
    case 16: return CALL(16);
    case 32: return CALL(32);
    case 48: return CALL(48);
    case 64: return CALL(64);
    case 96: return CALL(96);
    case 128: return CALL(128);
    default: return -1;
  }
#undef CALL
}
