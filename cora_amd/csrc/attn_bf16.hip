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
// four consecutive values as bf16 (8 bytes): the backward-data scratch kept as bf16 for the GEMM that reads it next
__device__ __forceinline__ void store_bf16x4(__bf16* dst, const float4 v) {
  typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
  bf4 b; b[0] = (__bf16)v.x; b[1] = (__bf16)v.y; b[2] = (__bf16)v.z; b[3] = (__bf16)v.w;
  *reinterpret_cast<bf4*>(dst) = b;
}


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

// lo = bf16(x - hi): hi + lo carries x to about 2^-17
__device__ __forceinline__ bf16x4 cvt4_lo(const float4 v, const bf16x4 h) {
  bf16x4 l;
  l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]);
  l[2] = (__bf16)(v.z - (float)h[2]); l[3] = (__bf16)(v.w - (float)h[3]);
  return l;
}
__device__ __forceinline__ bf16x4 cvt4_lo(const f32x4 v, const bf16x4 h) {
  bf16x4 l;
  l[0] = (__bf16)(v[0] - (float)h[0]); l[1] = (__bf16)(v[1] - (float)h[1]);
  l[2] = (__bf16)(v[2] - (float)h[2]); l[3] = (__bf16)(v[3] - (float)h[3]);
  return l;
}

template <int DH> struct Img {
  static constexpr int LR = DH + 8;   // bf16 per row of the row-major image
  static constexpr int LT = 68;       // bf16 per row of the transposed image (64 streamed rows + 4)
  static constexpr int NR = CHB * LR, NT = DH * LT;
};

// Staging of rows [r0, r0 + 64) x [0, DH) of two fp32 matrices as bf16, in two halves so that the global
// loads of the NEXT chunk are in flight while the workgroup computes on the current one (few workgroups
// fit a CU, so there is little else to hide that latency): load() fills registers, store() converts and
// writes LDS.  A unit is 2 rows x 4 columns (two float4 loads); 64 consecutive units = 16 row pairs x 4
// column quads (64 B of a row per 4 lanes), which keeps the b32 stores into the transposed image on 32
// different banks.  R / T may be null (image not needed).
template <int DH, int NTH>
struct Stage2 {
  static constexpr int LR = Img<DH>::LR, LT = Img<DH>::LT;
  static constexpr int NU = 8 * DH;    // units per matrix (a multiple of 64: the matrix index is wave-uniform)
  static constexpr int NI = (2 * NU + NTH - 1) / NTH;
  float4 a[NI], b[NI];

  __device__ __forceinline__ void load(const float* src0, int ld0, const float* src1, int ld1, int r0, int nrows) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int u = it * NTH + (int)threadIdx.x;
      a[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      b[it] = a[it];
      if (2 * NU % NTH != 0 && u >= 2 * NU) continue;
      const int m = u >= NU;
      const int v = m ? u - NU : u;
      const int wl = v >> 6, i = v & 63;
      const int c4 = 4 * (wl >> 1) + (i & 3);
      const int row = r0 + 2 * (16 * (wl & 1) + (i >> 2));
      const float* src = m ? src1 : src0;
      const int ld = m ? ld1 : ld0;
      if (row < nrows) a[it] = *reinterpret_cast<const float4*>(src + (long long)row * ld + 4 * c4);
      if (row + 1 < nrows) b[it] = *reinterpret_cast<const float4*>(src + (long long)(row + 1) * ld + 4 * c4);
    }
  }
  // two-term split x = hi + lo (forward): matrix 0 row-major (hi, lo), matrix 1 transposed (hi, lo)
  __device__ __forceinline__ void store_split(__bf16* R0h, __bf16* R0l, __bf16* T1h, __bf16* T1l) const {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int u = it * NTH + (int)threadIdx.x;
      if (2 * NU % NTH != 0 && u >= 2 * NU) continue;
      const int m = u >= NU;
      const int v = m ? u - NU : u;
      const int wl = v >> 6, i = v & 63;
      const int c4 = 4 * (wl >> 1) + (i & 3);
      const int row = 2 * (16 * (wl & 1) + (i >> 2));
      const bf16x4 ha = cvt4(a[it]), hb = cvt4(b[it]);
      const bf16x4 la = cvt4_lo(a[it], ha), lb = cvt4_lo(b[it], hb);
      if (!m) {
        *reinterpret_cast<bf16x4*>(&R0h[row * LR + 4 * c4]) = ha;
        *reinterpret_cast<bf16x4*>(&R0h[(row + 1) * LR + 4 * c4]) = hb;
        *reinterpret_cast<bf16x4*>(&R0l[row * LR + 4 * c4]) = la;
        *reinterpret_cast<bf16x4*>(&R0l[(row + 1) * LR + 4 * c4]) = lb;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bf16x2 pr;
          pr[0] = ha[j]; pr[1] = hb[j];
          *reinterpret_cast<bf16x2*>(&T1h[(4 * c4 + j) * LT + row]) = pr;
          pr[0] = la[j]; pr[1] = lb[j];
          *reinterpret_cast<bf16x2*>(&T1l[(4 * c4 + j) * LT + row]) = pr;
        }
      }
    }
  }
  __device__ __forceinline__ void store(__bf16* R0, __bf16* T0, __bf16* R1, __bf16* T1) const {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int u = it * NTH + (int)threadIdx.x;
      if (2 * NU % NTH != 0 && u >= 2 * NU) continue;
      const int m = u >= NU;
      const int v = m ? u - NU : u;
      const int wl = v >> 6, i = v & 63;
      const int c4 = 4 * (wl >> 1) + (i & 3);
      const int row = 2 * (16 * (wl & 1) + (i >> 2));
      __bf16* R = m ? R1 : R0;
      __bf16* T = m ? T1 : T0;
      const bf16x4 ha = cvt4(a[it]), hb = cvt4(b[it]);
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
};

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

// fragments of streamed rows 16 tt .. 16 tt + 15 from the row-major image (score A operands)
template <int DH>
__device__ __forceinline__ void row_frags(bf16x4* a, const __bf16* R, int tt, int c, int grp) {
  constexpr int LR = Img<DH>::LR;
#pragma unroll
  for (int s = 0; s < DH / 16; ++s) a[s] = *reinterpret_cast<const bf16x4*>(&R[(16 * tt + c) * LR + 16 * s + 4 * grp]);
}
// fragments of the same rows from the transposed image (accumulation A operands, one per 16 output columns)
template <int DH>
__device__ __forceinline__ void col_frags(bf16x4* a, const __bf16* T, int tt, int c, int grp) {
  constexpr int LT = Img<DH>::LT;
#pragma unroll
  for (int ct = 0; ct < DH / 16; ++ct) a[ct] = *reinterpret_cast<const bf16x4*>(&T[(16 * ct + c) * LT + 16 * tt + 4 * grp]);
}
template <int DH>
__device__ __forceinline__ f32x4 score_bf(const bf16x4* a, const bf16x4* frag) {
  f32x4 S = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < DH / 16; ++s) S = mfma_bf(a[s], frag[s], S);
  return S;
}
template <int DH>
__device__ __forceinline__ void accum_bf(f32x4* acc, const bf16x4* a, const bf16x4 w) {
#pragma unroll
  for (int ct = 0; ct < DH / 16; ++ct) acc[ct] = mfma_bf(a[ct], w, acc[ct]);
}

// A workgroup of NW waves owns 16 * NW * OT rows: wave w holds the 16-row tiles w, w + NW, .. (OT of them),
// so every fragment read from LDS serves OT own tiles and a chunk is staged once per 16 * NW * OT own rows
// (the kernels are bound by that staging, not by the MFMAs).  An own tile past the sequence has zero
// fragments and zero row constants: it computes finite values that are never stored.

// ---------------------------------------------------------------------------------------------
// own = queries (+ dO): streams K (both images) and V (row-major); dQ, and delta = rowsum(dO * O) for bwd-kv
template <int DH> struct SmemQ {
  __bf16 Kr[Img<DH>::NR], Vr[Img<DH>::NR], Kt[Img<DH>::NT];
};
template <int DH, int NW, int OT>
__global__ __launch_bounds__(64 * NW, (DH <= 64 && OT == 1) ? 4 : 1) void attn_bwd_q_bf16_kernel(
    const float* __restrict__ qkv, const float* __restrict__ O, const float* __restrict__ lse2, const float* __restrict__ dO,
    float* __restrict__ delta, float* __restrict__ dqkv, int nimg, int N, int H, int D, int out16) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SmemQ<DH>& sm = *reinterpret_cast<SmemQ<DH>*>(smem_raw);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, grp = lane >> 4;
  constexpr int ROWS = 16 * NW * OT;
  int rb, h, img;
  if (!attn_coords((N + ROWS - 1) / ROWS, H, nimg, rb, h, img)) return;
  const int ld = 3 * D;
  const float* base = qkv + (long long)img * N * ld;
  const float* Q = base + h * DH;
  const float* K = base + D + h * DH;
  const float* V = base + 2 * D + h * DH;
  const float* dOp = dO + (long long)img * N * D + h * DH;
  const float* Op = O + (long long)img * N * D + h * DH;
  const float scale = rsqrtf((float)DH);
  const float sc = scale * CORA_LOG2E;

  bf16x4 qf[OT][DH / 16], dof[OT][DH / 16];
  float dl[OT], l2[OT];
  f32x4 acc[OT][DH / 16];
#pragma unroll
  for (int o = 0; o < OT; ++o) {
    const int qrow = rb * ROWS + 16 * (w + NW * o) + c;
    load_frag_bf<DH>(qf[o], Q, ld, qrow, N, grp);
    float d = 0.f;
#pragma unroll
    for (int s = 0; s < DH / 16; ++s) {   // delta in fp32, from the fp32 dO and O
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      if (qrow < N) {
        a = *reinterpret_cast<const float4*>(dOp + (long long)qrow * D + 16 * s + 4 * grp);
        b = *reinterpret_cast<const float4*>(Op + (long long)qrow * D + 16 * s + 4 * grp);
      }
      d += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
      dof[o][s] = cvt4(a);
    }
    dl[o] = grp_sum(d);
    l2[o] = (qrow < N) ? lse2[((long long)img * N + qrow) * H + h] : 0.f;
    if (qrow < N && grp == 0) delta[((long long)img * N + qrow) * H + h] = dl[o];
#pragma unroll
    for (int ct = 0; ct < DH / 16; ++ct) acc[o][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const bool wave_active = rb * ROWS + 16 * w < N;   // its first tile; the later ones lie further on
  Stage2<DH, 64 * NW> stg;
  stg.load(K, ld, V, ld, 0, N);
  for (int k0 = 0; k0 < N; k0 += CHB) {
    __syncthreads();
    stg.store(sm.Kr, sm.Kt, sm.Vr, nullptr);
    __syncthreads();
    if (k0 + CHB < N) stg.load(K, ld, V, ld, k0 + CHB, N);
    if (!wave_active) continue;
    // keys past the sequence are staged as zeros: their K column of the accumulation is zero, no mask needed
    const int nvt = min(4, (N - k0 + 15) >> 4);
#pragma unroll 1
    for (int tt = 0; tt < nvt; ++tt) {
      bf16x4 ka[DH / 16], va[DH / 16], kt[DH / 16];
      row_frags<DH>(ka, sm.Kr, tt, c, grp);
      row_frags<DH>(va, sm.Vr, tt, c, grp);
      col_frags<DH>(kt, sm.Kt, tt, c, grp);
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        const f32x4 S = score_bf<DH>(ka, qf[o]);     // S[key][query]
        const f32x4 dP = score_bf<DH>(va, dof[o]);   // dP[key][query] = V[key] . dO[query]
        f32x4 dS;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) dS[reg] = __builtin_amdgcn_exp2f(S[reg] * sc - l2[o]) * (dP[reg] - dl[o]);
        accum_bf<DH>(acc[o], kt, cvt4(dS));          // dQ^T += K^T dS
      }
    }
  }
#pragma unroll
  for (int o = 0; o < OT; ++o) {
    const int qrow = rb * ROWS + 16 * (w + NW * o) + c;
    if (qrow < N) {
      const long long doff = ((long long)img * N + qrow) * ld + h * DH;
      float* drow = dqkv + doff;
#pragma unroll
      for (int ct = 0; ct < DH / 16; ++ct) {
        const float4 v = make_float4(acc[o][ct][0] * scale, acc[o][ct][1] * scale, acc[o][ct][2] * scale, acc[o][ct][3] * scale);
        if (out16) store_bf16x4(reinterpret_cast<__bf16*>(dqkv) + doff + 16 * ct + 4 * grp, v);
        else *reinterpret_cast<float4*>(drow + 16 * ct + 4 * grp) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// own = keys (+ V): streams Q and dO (both images each) with their row constants; dK, dV
template <int DH> struct SmemKV {
  __bf16 Qr[Img<DH>::NR], dOr[Img<DH>::NR], Qt[Img<DH>::NT], dOt[Img<DH>::NT];
  float rowc[2 * CHB];   // lse2, delta of the streamed query rows
};
template <int DH, int NW, int OT>
__global__ __launch_bounds__(64 * NW, OT == 1 ? (DH <= 48 ? 4 : DH <= 64 ? 3 : 1) : 1) void attn_bwd_kv_bf16_kernel(
    const float* __restrict__ qkv, const float* __restrict__ lse2, const float* __restrict__ dO, const float* __restrict__ delta,
    float* __restrict__ dqkv, int nimg, int N, int H, int D, int out16) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SmemKV<DH>& sm = *reinterpret_cast<SmemKV<DH>*>(smem_raw);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, grp = lane >> 4;
  constexpr int ROWS = 16 * NW * OT;
  int rb, h, img;
  if (!attn_coords((N + ROWS - 1) / ROWS, H, nimg, rb, h, img)) return;
  const int ld = 3 * D;
  const float* base = qkv + (long long)img * N * ld;
  const float* Q = base + h * DH;
  const float* K = base + D + h * DH;
  const float* V = base + 2 * D + h * DH;
  const float* dOp = dO + (long long)img * N * D + h * DH;
  const float scale = rsqrtf((float)DH);
  const float sc = scale * CORA_LOG2E;

  bf16x4 kf[OT][DH / 16], vf[OT][DH / 16];
  f32x4 dk[OT][DH / 16], dv[OT][DH / 16];
#pragma unroll
  for (int o = 0; o < OT; ++o) {
    const int krow = rb * ROWS + 16 * (w + NW * o) + c;
    load_frag_bf<DH>(kf[o], K, ld, krow, N, grp);
    load_frag_bf<DH>(vf[o], V, ld, krow, N, grp);
#pragma unroll
    for (int ct = 0; ct < DH / 16; ++ct) {
      dk[o][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dv[o][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }

  const bool wave_active = rb * ROWS + 16 * w < N;
  Stage2<DH, 64 * NW> stg;
  stg.load(Q, ld, dOp, D, 0, N);
  float rc_l = 0.f, rc_d = 0.f;          // row constants of the next chunk (threads 0..63)
  const long long rbase = (long long)img * N * H + h;
  if (threadIdx.x < CHB && (int)threadIdx.x < N) {
    rc_l = lse2[rbase + (long long)threadIdx.x * H];
    rc_d = delta[rbase + (long long)threadIdx.x * H];
  }
  for (int q0 = 0; q0 < N; q0 += CHB) {
    __syncthreads();
    stg.store(sm.Qr, sm.Qt, sm.dOr, sm.dOt);
    if (threadIdx.x < CHB) {
      sm.rowc[threadIdx.x] = rc_l;
      sm.rowc[CHB + threadIdx.x] = rc_d;
    }
    __syncthreads();
    if (q0 + CHB < N) {
      stg.load(Q, ld, dOp, D, q0 + CHB, N);
      if (threadIdx.x < CHB) {
        const int q = q0 + CHB + threadIdx.x;
        rc_l = (q < N) ? lse2[rbase + (long long)q * H] : 0.f;
        rc_d = (q < N) ? delta[rbase + (long long)q * H] : 0.f;
      }
    }
    if (!wave_active) continue;
    // query rows past the sequence are staged as zeros (Q, dO, row constants): P = 1 meets dO = 0 and
    // dS = 1 * (0 - 0), so they add nothing and need no mask
    const int nvt = min(4, (N - q0 + 15) >> 4);
#pragma unroll 1
    for (int tt = 0; tt < nvt; ++tt) {
      bf16x4 qa[DH / 16], da[DH / 16], qt[DH / 16], dt[DH / 16];
      row_frags<DH>(qa, sm.Qr, tt, c, grp);
      row_frags<DH>(da, sm.dOr, tt, c, grp);
      col_frags<DH>(qt, sm.Qt, tt, c, grp);
      col_frags<DH>(dt, sm.dOt, tt, c, grp);
      const f32x4 l2 = *reinterpret_cast<const f32x4*>(&sm.rowc[16 * tt + 4 * grp]);
      const f32x4 dl = *reinterpret_cast<const f32x4*>(&sm.rowc[CHB + 16 * tt + 4 * grp]);
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        const f32x4 S = score_bf<DH>(qa, kf[o]);     // S[q][key]
        const f32x4 dP = score_bf<DH>(da, vf[o]);    // dP[q][key] = dO[q] . V[key]
        f32x4 P, dS;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          P[reg] = __builtin_amdgcn_exp2f(S[reg] * sc - l2[reg]);
          dS[reg] = P[reg] * (dP[reg] - dl[reg]);
        }
        accum_bf<DH>(dv[o], dt, cvt4(P));     // dV^T += dO^T P
        accum_bf<DH>(dk[o], qt, cvt4(dS));    // dK^T += Q^T dS
      }
    }
  }
#pragma unroll
  for (int o = 0; o < OT; ++o) {
    const int krow = rb * ROWS + 16 * (w + NW * o) + c;
    if (krow < N) {
      const long long doff = ((long long)img * N + krow) * ld + h * DH;
      float* drow = dqkv + doff;
#pragma unroll
      for (int ct = 0; ct < DH / 16; ++ct) {
        const float4 vk = make_float4(dk[o][ct][0] * scale, dk[o][ct][1] * scale, dk[o][ct][2] * scale, dk[o][ct][3] * scale);
        const float4 vv = make_float4(dv[o][ct][0], dv[o][ct][1], dv[o][ct][2], dv[o][ct][3]);
        if (out16) {
          store_bf16x4(reinterpret_cast<__bf16*>(dqkv) + doff + D + 16 * ct + 4 * grp, vk);
          store_bf16x4(reinterpret_cast<__bf16*>(dqkv) + doff + 2 * D + 16 * ct + 4 * grp, vv);
        } else {
          *reinterpret_cast<float4*>(drow + D + 16 * ct + 4 * grp) = vk;
          *reinterpret_cast<float4*>(drow + 2 * D + 16 * ct + 4 * grp) = vv;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Forward with two-term split operands (x = hi + lo, three MFMAs per product: hi*hi + hi*lo + lo*hi, good to
// about 2^-16): its output feeds e = h - u, so plain bf16 is not enough here (see gemm_bf16.hip).  own =
// queries; streams K (row-major, hi and lo) and V (transposed, hi and lo); online softmax in fp32, base-2 LSE.
template <int DH> struct SmemF {
  __bf16 Krh[Img<DH>::NR], Krl[Img<DH>::NR], Vth[Img<DH>::NT], Vtl[Img<DH>::NT];
};
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, DH <= 48 ? 4 : DH <= 64 ? 3 : 1) void attn_fwd_bf16_kernel(
    const float* __restrict__ qkv, float* __restrict__ O, float* __restrict__ lse2, int nimg, int N, int H, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SmemF<DH>& sm = *reinterpret_cast<SmemF<DH>*>(smem_raw);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, grp = lane >> 4;
  constexpr int ROWS = 16 * NW;
  int rb, h, img;
  if (!attn_coords((N + ROWS - 1) / ROWS, H, nimg, rb, h, img)) return;
  const int ld = 3 * D;
  const float* base = qkv + (long long)img * N * ld;
  const float* Q = base + h * DH;
  const float* K = base + D + h * DH;
  const float* V = base + 2 * D + h * DH;
  const int qrow = rb * ROWS + 16 * w + c;
  const float sc = rsqrtf((float)DH) * CORA_LOG2E;  // base-2 logits

  bf16x4 qh[DH / 16], ql[DH / 16];
#pragma unroll
  for (int s = 0; s < DH / 16; ++s) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qrow < N) v = *reinterpret_cast<const float4*>(Q + (long long)qrow * ld + 16 * s + 4 * grp);
    qh[s] = cvt4(v);
    ql[s] = cvt4_lo(v, qh[s]);
  }
  f32x4 oacc[DH / 16];
#pragma unroll
  for (int ct = 0; ct < DH / 16; ++ct) oacc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, lsum = 0.f;

  const bool wave_active = rb * ROWS + 16 * w < N;
  Stage2<DH, 64 * NW> stg;
  stg.load(K, ld, V, ld, 0, N);
  for (int k0 = 0; k0 < N; k0 += CHB) {
    __syncthreads();
    stg.store_split(sm.Krh, sm.Krl, sm.Vth, sm.Vtl);
    __syncthreads();
    if (k0 + CHB < N) stg.load(K, ld, V, ld, k0 + CHB, N);
    if (!wave_active) continue;
    for_valid_tiles((N - k0 + 15) >> 4, [&](auto nv) {
      constexpr int NVT = decltype(nv)::value;
      f32x4 S[NVT];
      float mx = -INFINITY;
#pragma unroll
      for (int tt = 0; tt < NVT; ++tt) {
        bf16x4 kh[DH / 16], kl[DH / 16];
        row_frags<DH>(kh, sm.Krh, tt, c, grp);
        row_frags<DH>(kl, sm.Krl, tt, c, grp);
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < DH / 16; ++s) {
          a = mfma_bf(kl[s], qh[s], a);
          a = mfma_bf(kh[s], ql[s], a);
          a = mfma_bf(kh[s], qh[s], a);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int key = k0 + 16 * tt + 4 * grp + reg;
          a[reg] = (key < N) ? a[reg] * sc : -INFINITY;
          mx = fmaxf(mx, a[reg]);
        }
        S[tt] = a;
      }
      mx = grp_max(mx);
      const float mnew = fmaxf(m, mx);           // finite: every chunk holds at least one valid key
      const float alpha = __builtin_amdgcn_exp2f(m - mnew);
      float ps = 0.f;
#pragma unroll
      for (int tt = 0; tt < NVT; ++tt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          S[tt][reg] = __builtin_amdgcn_exp2f(S[tt][reg] - mnew);
          ps += S[tt][reg];
        }
      ps = grp_sum(ps);
      lsum = lsum * alpha + ps;
      m = mnew;
#pragma unroll
      for (int ct = 0; ct < DH / 16; ++ct) oacc[ct] *= alpha;
#pragma unroll
      for (int tt = 0; tt < NVT; ++tt) {
        const bf16x4 ph = cvt4(S[tt]);
        const bf16x4 pl = cvt4_lo(S[tt], ph);
        bf16x4 vh[DH / 16], vl[DH / 16];
        col_frags<DH>(vh, sm.Vth, tt, c, grp);
        col_frags<DH>(vl, sm.Vtl, tt, c, grp);
#pragma unroll
        for (int ct = 0; ct < DH / 16; ++ct) {
          oacc[ct] = mfma_bf(vl[ct], ph, oacc[ct]);
          oacc[ct] = mfma_bf(vh[ct], pl, oacc[ct]);
          oacc[ct] = mfma_bf(vh[ct], ph, oacc[ct]);
        }
      }
    });
  }
  if (qrow < N) {
    const float inv = 1.0f / lsum;
    float* orow = O + ((long long)img * N + qrow) * D + h * DH;
#pragma unroll
    for (int ct = 0; ct < DH / 16; ++ct)
      *reinterpret_cast<float4*>(orow + 16 * ct + 4 * grp) =
          make_float4(oacc[ct][0] * inv, oacc[ct][1] * inv, oacc[ct][2] * inv, oacc[ct][3] * inv);
    if (grp == 0) lse2[((long long)img * N + qrow) * H + h] = m + __builtin_amdgcn_logf(lsum);
  }
}

template <int DH, int NW>
int run_fwd_bf16(const float* qkv, float* O, float* lse2, int nimg, int N, int H, int D, hipStream_t st) {
  const size_t smf = sizeof(SmemF<DH>);
  static bool attr_set[64] = {};
  if (smf > 48 * 1024 && first_use_on_device(attr_set)) {
    if (hipFuncSetAttribute((const void*)attn_fwd_bf16_kernel<DH, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smf) != hipSuccess) return -1;
  }
  constexpr int ROWS = 16 * NW;
  hipLaunchKernelGGL((attn_fwd_bf16_kernel<DH, NW>), dim3(attn_grid((N + ROWS - 1) / ROWS, H, nimg)), dim3(64 * NW), smf, st,
                     qkv, O, lse2, nimg, N, H, D);
  return 0;
}
template <int DH>
int run_fwd_bf16_dh(const float* qkv, float* O, float* lse2, int nimg, int N, int H, int D, hipStream_t st) {
  // measured (cfg 4 shape / video16): 64 own rows 192 / 146 us, 128 rows 197 / 156 us; the fp32 kernel 308 / 269 us
  int rows = 64;
  if (const char* e = getenv("CORA_ATTN_BF16_ROWS")) { const int v = atoi(e); if (v == 64 || v == 128) rows = v; }
  return rows == 128 ? run_fwd_bf16<DH, 8>(qkv, O, lse2, nimg, N, H, D, st) : run_fwd_bf16<DH, 4>(qkv, O, lse2, nimg, N, H, D, st);
}

template <int DH, int NW, int OT>
int run_bwd_bf16(const float* qkv, const float* O, const float* lse2, const float* dO, float* delta,
                 float* dqkv, int nimg, int N, int H, int D, hipStream_t st, int out16) {
  const size_t smq = sizeof(SmemQ<DH>), smkv = sizeof(SmemKV<DH>);
  static bool attr_set[64] = {};
  if (smkv > 48 * 1024 && first_use_on_device(attr_set)) {
    if (hipFuncSetAttribute((const void*)attn_bwd_q_bf16_kernel<DH, NW, OT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smq) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void*)attn_bwd_kv_bf16_kernel<DH, NW, OT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smkv) != hipSuccess) return -1;
  }
  constexpr int ROWS = 16 * NW * OT;
  const dim3 grid(attn_grid((N + ROWS - 1) / ROWS, H, nimg));
  hipLaunchKernelGGL((attn_bwd_q_bf16_kernel<DH, NW, OT>), grid, dim3(64 * NW), smq, st, qkv, O, lse2, dO, delta, dqkv, nimg, N, H, D, out16);
  hipLaunchKernelGGL((attn_bwd_kv_bf16_kernel<DH, NW, OT>), grid, dim3(64 * NW), smkv, st, qkv, lse2, dO, delta, dqkv, nimg, N, H, D, out16);
  return 0;
}

// own rows per workgroup: 128 (8 waves) where that still leaves every CU two workgroups, else 64 (4 waves).
// Measured on the backward pair with the next chunk prefetched (cfg 4 shape / video16): 64 rows 385 / 217 us,
// 128 rows 355 / 202 us, 256 rows (8 waves x 2 own tiles, CORA_ATTN_BF16_ROWS=256 only) 390 / 225 us.
template <int DH>
int run_bwd_bf16_dh(const float* qkv, const float* O, const float* lse2, const float* dO, float* delta,
                    float* dqkv, int nimg, int N, int H, int D, hipStream_t st, int out16) {
  int rows = ((long long)((N + 127) / 128) * H * nimg >= 512) ? 128 : 64;
  if (const char* e = getenv("CORA_ATTN_BF16_ROWS")) { const int v = atoi(e); if (v == 64 || v == 128 || (v == 256 && DH <= 64)) rows = v; }
  if constexpr (DH <= 64) {
    if (rows == 256) return run_bwd_bf16<DH, 8, 2>(qkv, O, lse2, dO, delta, dqkv, nimg, N, H, D, st, out16);
  }
  if (rows == 128) return run_bwd_bf16<DH, 8, 1>(qkv, O, lse2, dO, delta, dqkv, nimg, N, H, D, st, out16);
  return run_bwd_bf16<DH, 4, 1>(qkv, O, lse2, dO, delta, dqkv, nimg, N, H, D, st, out16);
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
                         float* dqkv, int nimg, int N, int H, int D, hipStream_t st, int out16) {
  if (!attn_bwd_bf16_supported(N, H, D)) return -1;
#define CALL(X) run_bwd_bf16_dh<X>(qkv, O, lse2, dO, delta, dqkv, nimg, N, H, D, st, out16)
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

// forward on split bf16 operands: same shapes as the backward; CORA_BF16_FP32_ATTN_FWD keeps the fp32 kernel
bool attn_fwd_bf16_supported(int N, int H, int D) {
  if (getenv("CORA_BF16_FP32_ATTN_FWD")) return false;
  return attn_bwd_bf16_supported(N, H, D);
}
int launch_attn_fwd_bf16(const float* qkv, float* O, float* lse2, int nimg, int N, int H, int D, hipStream_t st) {
  if (!attn_fwd_bf16_supported(N, H, D)) return -1;
#define CALL(X) run_fwd_bf16_dh<X>(qkv, O, lse2, nimg, N, H, D, st)
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
