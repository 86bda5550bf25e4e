// fused_wide2.hip -- the fused per-(block, image) kernel for the D = 128 points of the 12-block sweep grid
// (N = 64 tokens, D = 128, 4 / 2 / 1 heads: /root/reference/sweeps/sweep_12block_comprehensive.yaml:8-47),
// built for TWO workgroups per CU.  Heads of 32 (the canonical point) are the native form; heads of 64 / 128 run
// on the same 32-column head tiles with the scores accumulated across a head's slices (wide2_attn_*_sliced).
//
// Same math and the same data layout "T" as fused_wide.hip (block /root/reference/pcx/nn/_layer.py:428-447,
// attention :359-402, LayerNorm :296-315, double residual /root/reference/src/decoder_transformer.py:238,
// hand-derived VJP of jax.value_and_grad(energy) w.r.t. the vodes, :396,445).  Why a second kernel: on
// gfx950 one wave per SIMD cannot overlap its own non-matrix instructions with its v_mfma_f32_16x16x4_f32
// stream -- measured (scripts/ubench/mfma_rate.hip) 32.3 cycles per MFMA bare, 34.9 with one ds_read_b128
// per 4 MFMAs, +5.5 cycles per VALU instruction in between -- so fused_wide.hip (one workgroup per CU,
// 141 KB of LDS, 512 registers) stalls at 40 cycles per MFMA in its GEMM stages.  Two waves per SIMD hide
// each other's LDS / VALU / global issue.  To fit two workgroups on a CU (<= 80 KB LDS, <= 256 VGPRs):
//  * weight tiles are D/4 = 32 output rows x D inputs (16.9 KB), two LDS slots, 96 tiles per workgroup;
//    stage k writes tile k+1 to LDS, fetches tile k+2 into the 4 staging float4 and multiplies with tile k,
//    the three interleaved group by group;
//  * K / V / Q / dO never sit in LDS whole: attention runs head by head on compact head tiles that each
//    wave fills with its own rows' slice of the head -- [64 tokens][32 + 4] for the operands of the
//    score-shaped products (contraction over the head dim), transposed [32][64 + 4] for the operands of
//    the products that sum over tokens (one ds_read_b128 per four MFMAs); four 9.2 KB slots in the
//    forward (K, V^T of two heads), six tiles in the backward (V, Q, dO and K^T, Q^T, dO^T of one head:
//    the four slots plus the weight slot that idles there); q, k, v (and P, and gelu'(a)) wait for the
//    backward in an L2-resident stash instead of LDS / registers;
//  * 76.8 KB of LDS per workgroup: 2 weight slots 33.8 KB, 4 head tiles 36.9 KB, LSE / delta rows 2 KB,
//    parameter block (b1, LN2 gamma / beta, b2, LN1 gamma) 4 KB.
// The same file holds the chain kernel (status "forward": INIT pass and reconstruction), the forward half
// of the block body walking one image through all L blocks.
#include "common.h"

#include "fused_wide_common.h"

namespace {

// acc[ot] (ot < 4: the 64 rows of a head tile) += sum over the head's NCH column groups of T[16 ot + i][c] xs[c]
template <int LDH, int NCH>
__device__ __forceinline__ void gemm_hd(const float* __restrict__ t, const float* xs, f32x4* acc, int i, int grp) {
  const lds_float* tl = opaque(t + i * LDH + 4 * grp);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    float4 a[4];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) a[ot] = lds_ld4(&tl[16 * ot * LDH + 16 * c]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = mfma16(a[ot].x, xs[4 * c + 0], acc[ot]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = mfma16(a[ot].y, xs[4 * c + 1], acc[ot]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = mfma16(a[ot].z, xs[4 * c + 2], acc[ot]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = mfma16(a[ot].w, xs[4 * c + 3], acc[ot]);
  }
}
// this wave's rows of a head slice (NCH column groups, 4 NCH registers) -> its rows of a head tile
template <int LDH, int NCH>
__device__ __forceinline__ void put_slice(const float* xs, float* tile, int w, int i, int grp) {
  lds_float* d = opaque_w(tile + (16 * w + i) * LDH + 4 * grp);
#pragma unroll
  for (int c = 0; c < NCH; ++c) lds_st4(d + 16 * c, make_float4(xs[4 * c], xs[4 * c + 1], xs[4 * c + 2], xs[4 * c + 3]));
}
// Transposed head tile T^T[col][token] (row stride LDT): the "sum over tokens" products read their A
// operand as one ds_read_b128 per 4 MFMAs instead of one ds_read_b32 per MFMA.
template <int LDT, int NCH>
__device__ __forceinline__ void put_slice_t(const float* xs, float* tile, int w, int i, int grp) {
  lds_float* d = opaque_w(tile + 4 * grp * LDT + 16 * w + i);
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) d[(16 * c + reg) * LDT] = xs[4 * c + reg];
}
// out[4 c + reg'] = scale * sum over the 64 tokens j of T^T[16 c + 4 grp + reg'][j] * pf[j]
template <int LDT, int NCH>
__device__ __forceinline__ void accum_t(const float* __restrict__ tt_, const float* pf, float* out, float scale, int i, int grp) {
  f32x4 acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const lds_float* tl = opaque(tt_ + i * LDT + 4 * grp);
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    float4 a[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) a[c] = lds_ld4(&tl[16 * c * LDT + 16 * tt]);
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = mfma16(a[c].x, pf[4 * tt + 0], acc[c]);
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = mfma16(a[c].y, pf[4 * tt + 1], acc[c]);
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = mfma16(a[c].z, pf[4 * tt + 2], acc[c]);
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = mfma16(a[c].w, pf[4 * tt + 3], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) out[4 * c + reg] = acc[c][reg] * scale;
}
template <int NCH>
__device__ __forceinline__ void load_slice(float* f, const float* base, int grp) {   // base = row pointer + 16 * first column group
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const float4 v = *reinterpret_cast<const float4*>(base + 16 * c + 4 * grp);
    f[4 * c] = v.x; f[4 * c + 1] = v.y; f[4 * c + 2] = v.z; f[4 * c + 3] = v.w;
  }
}
template <int NCH>
__device__ __forceinline__ void store_slice(const float* f, float* base, int grp) {
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    *reinterpret_cast<float4*>(base + 16 * c + 4 * grp) = make_float4(f[4 * c], f[4 * c + 1], f[4 * c + 2], f[4 * c + 3]);
}

// ---- heads wider than one 32-column slice (D = 128 with 2 or 1 heads) -----------------------------------
// The head tiles stay 32 columns wide; a head of NS slices accumulates its score-shaped products (S, dP: a
// contraction over the head dim) across the slices' tiles and runs the products that sum over tokens (O, dq,
// dK, dV: one 32-column slice of the output each) slice by slice.  A workgroup still walks through 24 head
// tiles in the backward and 8 in the forward whatever H is, so the LDS budget and the two workgroups per CU
// of the heads-of-32 form carry over.  Slice s = columns [32 s, 32 s + 32) = registers [8 s, 8 s + 8) of a row.
constexpr int W2_LDH = 36, W2_HT = 64 * W2_LDH, W2_LDT = 68;

// softmax over the 64 keys of S[key][q] (this lane: query i; keys 16 ot + 4 grp + reg): pf = P, returns the
// base-2 log-sum-exp of the scaled scores
__device__ __forceinline__ float softmax_keys(const f32x4* sa, float* pf, float sc) {
  unpack_at<4>(sa, pf, 0);
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 16; ++k) { pf[k] *= sc; mx = fmaxf(mx, pf[k]); }
  mx = grp_max(mx);
  float ls = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) { pf[k] = __builtin_amdgcn_exp2f(pf[k] - mx); ls += pf[k]; }
  ls = grp_sum(ls);
  const float inv = 1.0f / ls;
#pragma unroll
  for (int k = 0; k < 16; ++k) pf[k] *= inv;
  return mx + __builtin_amdgcn_logf(ls);
}
// P and dS of the key-owner pass from S[q][key] (sp) and dP[q][key] (dp): sp <- P, dp <- dS
__device__ __forceinline__ void p_ds_keys(float* sp, float* dp, const float* lse, const float* delta, float sc, int grp) {
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const float4 l4 = *reinterpret_cast<const float4*>(&lse[16 * tt + 4 * grp]);
    const float4 d4 = *reinterpret_cast<const float4*>(&delta[16 * tt + 4 * grp]);
    float pv;
    pv = __builtin_amdgcn_exp2f(sp[4 * tt + 0] * sc - l4.x); sp[4 * tt + 0] = pv; dp[4 * tt + 0] = pv * (dp[4 * tt + 0] - d4.x);
    pv = __builtin_amdgcn_exp2f(sp[4 * tt + 1] * sc - l4.y); sp[4 * tt + 1] = pv; dp[4 * tt + 1] = pv * (dp[4 * tt + 1] - d4.y);
    pv = __builtin_amdgcn_exp2f(sp[4 * tt + 2] * sc - l4.z); sp[4 * tt + 2] = pv; dp[4 * tt + 2] = pv * (dp[4 * tt + 2] - d4.z);
    pv = __builtin_amdgcn_exp2f(sp[4 * tt + 3] * sc - l4.w); sp[4 * tt + 3] = pv; dp[4 * tt + 3] = pv * (dp[4 * tt + 3] - d4.w);
  }
}

// forward attention, NS = 2 (two heads of 64) or 4 (one head of 128) slices per head; q / kk / v / of: the wave's
// rows (32 registers each); t[0..3]: the four head-tile slots; lse (LDS, [H][64]) and st_p (stash row, H x 64)
// may be null (chain kernel: nothing is kept for a backward)
template <int H>
__device__ __forceinline__ void wide2_attn_fwd_sliced(const float* q, const float* kk, const float* v, float* of, float* t0, float* t1,
                                                      float* t2, float* t3, float* lse, float* st_p, float sc, int w, int i, int grp) {
  constexpr int NS = 4 / H;
  static_assert(H == 1 || H == 2, "sliced form: D = 128 with heads of 64 or 128");
  if constexpr (NS == 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {      // one head per round: K_a, V_a^T, K_b, V_b^T
      const int a = 2 * h, b = 2 * h + 1;
      __syncthreads();                 // the previous round's tiles are no longer read
      put_slice<W2_LDH, 2>(kk + 8 * a, t0, w, i, grp);
      put_slice_t<W2_LDT, 2>(v + 8 * a, t1, w, i, grp);
      put_slice<W2_LDH, 2>(kk + 8 * b, t2, w, i, grp);
      put_slice_t<W2_LDT, 2>(v + 8 * b, t3, w, i, grp);
      __syncthreads();
      f32x4 sa[4];
      zero_acc<4>(sa);
      gemm_hd<W2_LDH, 2>(t0, q + 8 * a, sa, i, grp);          // S[key][q], summed over the head's two slices
      gemm_hd<W2_LDH, 2>(t2, q + 8 * b, sa, i, grp);
      float pf[16];
      const float l2 = softmax_keys(sa, pf, sc);
      if (lse && grp == 0) lse[64 * h + 16 * w + i] = l2;
      if (st_p) store_row<4>(pf, st_p + 64 * h, grp);
      accum_t<W2_LDT, 2>(t1, pf, of + 8 * a, 1.0f, i, grp);   // O, one slice of the head's columns at a time
      accum_t<W2_LDT, 2>(t3, pf, of + 8 * b, 1.0f, i, grp);
    }
  } else {
    float* const tt[4] = {t0, t1, t2, t3};
    __syncthreads();
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) put_slice<W2_LDH, 2>(kk + 8 * s_, tt[s_], w, i, grp);      // round 1: the four K slices
    __syncthreads();
    f32x4 sa[4];
    zero_acc<4>(sa);
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) gemm_hd<W2_LDH, 2>(tt[s_], q + 8 * s_, sa, i, grp);
    float pf[16];
    const float l2 = softmax_keys(sa, pf, sc);
    if (lse && grp == 0) lse[16 * w + i] = l2;
    if (st_p) store_row<4>(pf, st_p, grp);
    __syncthreads();                   // every wave is past its reads of the K tiles
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) put_slice_t<W2_LDT, 2>(v + 8 * s_, tt[s_], w, i, grp);     // round 2: the four V^T slices
    __syncthreads();
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) accum_t<W2_LDT, 2>(tt[s_], pf, of + 8 * s_, 1.0f, i, grp);
  }
}

// attention backward of the sliced form.  tz: six tile slots of W2_HT floats (weight slot 1, idle here, and the
// head-tile area behind it); dof: dO of the wave's rows; dq lands in dy; dk | dv go to the stash row st_dkv.
// Per head 6 NS tiles pass through the six slots in NS phases:
//   NS = 2:  [V_a V_b K_a^T K_b^T Q_a Q_b]  -> dP[key][q], delta, dS, dq ; S[q][key]
//            [dO_a dO_b dO_a^T dO_b^T Q_a^T Q_b^T] -> dP[q][key], P, dS ; dV, dK
//   NS = 4:  [V_0..3 K_0^T K_1^T] -> dP, delta, dS, dq_0, dq_1      [K_2^T K_3^T Q_0..3] -> dq_2, dq_3 ; S
//            [dO_0..3 dO_0^T dO_1^T] -> dP, P, dS ; dV_0, dV_1       [dO_2^T dO_3^T Q_0^T..Q_3^T] -> dV_2, dV_3 ; dK
template <int H>
__device__ __forceinline__ void wide2_attn_bwd_sliced(float* tz, const float* lse, float* delta, const float* st_q, const float* st_k,
                                                      const float* st_v, const float* st_p, float* st_dkv, const float* dof, float* dy,
                                                      float sc, float qscale, int w, int i, int grp) {
  constexpr int NS = 4 / H, D = 128;
  static_assert(H == 1 || H == 2, "sliced form: D = 128 with heads of 64 or 128");
  float* const z[6] = {tz, tz + W2_HT, tz + 2 * W2_HT, tz + 3 * W2_HT, tz + 4 * W2_HT, tz + 5 * W2_HT};
  if constexpr (NS == 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int a = 2 * h, b = 2 * h + 1;
      float ks[16], vs[16], qs[16], pf[16], dp[16], sp[16];
      load_slice<4>(ks, st_k + 64 * h, grp);
      load_slice<4>(vs, st_v + 64 * h, grp);
      load_slice<4>(qs, st_q + 64 * h, grp);
      load_row<4>(pf, st_p + 64 * h, grp);
      __syncthreads();                 // the previous head's tiles (and weight tile 21 NQ - 1) are no longer read
      put_slice<W2_LDH, 2>(vs, z[0], w, i, grp);
      put_slice<W2_LDH, 2>(vs + 8, z[1], w, i, grp);
      put_slice_t<W2_LDT, 2>(ks, z[2], w, i, grp);
      put_slice_t<W2_LDT, 2>(ks + 8, z[3], w, i, grp);
      put_slice<W2_LDH, 2>(qs, z[4], w, i, grp);
      put_slice<W2_LDH, 2>(qs + 8, z[5], w, i, grp);
      __syncthreads();
      {
        // own queries: dP[key][q] = V[key] . dO[q] over both slices, delta = sum_key P dP, dS, dq slice by slice
        f32x4 sa[4];
        zero_acc<4>(sa);
        gemm_hd<W2_LDH, 2>(z[0], dof + 8 * a, sa, i, grp);
        gemm_hd<W2_LDH, 2>(z[1], dof + 8 * b, sa, i, grp);
        unpack_at<4>(sa, dp, 0);
        float dl = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) dl += pf[k] * dp[k];
        dl = grp_sum(dl);
        if (grp == 0) delta[64 * h + 16 * w + i] = dl;
#pragma unroll
        for (int k = 0; k < 16; ++k) dp[k] = pf[k] * (dp[k] - dl);
        accum_t<W2_LDT, 2>(z[2], dp, dy + 8 * a, qscale, i, grp);
        accum_t<W2_LDT, 2>(z[3], dp, dy + 8 * b, qscale, i, grp);
        // own keys, first half: S[q][key] over both slices
        zero_acc<4>(sa);
        gemm_hd<W2_LDH, 2>(z[4], ks, sa, i, grp);
        gemm_hd<W2_LDH, 2>(z[5], ks + 8, sa, i, grp);
        unpack_at<4>(sa, sp, 0);
      }
      __syncthreads();                 // delta of all 64 queries visible; the first six tiles are no longer read
      put_slice<W2_LDH, 2>(dof + 8 * a, z[0], w, i, grp);
      put_slice<W2_LDH, 2>(dof + 8 * b, z[1], w, i, grp);
      put_slice_t<W2_LDT, 2>(dof + 8 * a, z[2], w, i, grp);
      put_slice_t<W2_LDT, 2>(dof + 8 * b, z[3], w, i, grp);
      put_slice_t<W2_LDT, 2>(qs, z[4], w, i, grp);
      put_slice_t<W2_LDT, 2>(qs + 8, z[5], w, i, grp);
      __syncthreads();
      {
        f32x4 sa[4];
        zero_acc<4>(sa);
        gemm_hd<W2_LDH, 2>(z[0], vs, sa, i, grp);               // dP[q][key]
        gemm_hd<W2_LDH, 2>(z[1], vs + 8, sa, i, grp);
        unpack_at<4>(sa, dp, 0);
        p_ds_keys(sp, dp, lse + 64 * h, delta + 64 * h, sc, grp);
        accum_t<W2_LDT, 2>(z[2], sp, vs, 1.0f, i, grp);          // dV = P^T dO, slice by slice (vs is dead as an operand)
        accum_t<W2_LDT, 2>(z[3], sp, vs + 8, 1.0f, i, grp);
        accum_t<W2_LDT, 2>(z[4], dp, ks, qscale, i, grp);        // dK = dS^T Q
        accum_t<W2_LDT, 2>(z[5], dp, ks + 8, qscale, i, grp);
        store_slice<4>(ks, st_dkv + 64 * h, grp);
        store_slice<4>(vs, st_dkv + D + 64 * h, grp);
      }
    }
  } else {
    float pf[16], dp[16], sp[16];
    {
      float vs[32], k2[16];
      load_row<8>(vs, st_v, grp);
      load_slice<4>(k2, st_k, grp);
      load_row<4>(pf, st_p, grp);
      __syncthreads();                 // weight tile 21 NQ - 1 is no longer read
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) put_slice<W2_LDH, 2>(vs + 8 * s_, z[s_], w, i, grp);
      put_slice_t<W2_LDT, 2>(k2, z[4], w, i, grp);
      put_slice_t<W2_LDT, 2>(k2 + 8, z[5], w, i, grp);
    }
    __syncthreads();
    {
      f32x4 sa[4];
      zero_acc<4>(sa);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) gemm_hd<W2_LDH, 2>(z[s_], dof + 8 * s_, sa, i, grp);     // dP[key][q]
      unpack_at<4>(sa, dp, 0);
      float dl = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) dl += pf[k] * dp[k];
      dl = grp_sum(dl);
      if (grp == 0) delta[16 * w + i] = dl;
#pragma unroll
      for (int k = 0; k < 16; ++k) dp[k] = pf[k] * (dp[k] - dl);
      accum_t<W2_LDT, 2>(z[4], dp, dy, qscale, i, grp);
      accum_t<W2_LDT, 2>(z[5], dp, dy + 8, qscale, i, grp);
    }
    {
      float k2[16], qs[32];
      load_slice<4>(k2, st_k + 64, grp);
      load_row<8>(qs, st_q, grp);
      __syncthreads();
      put_slice_t<W2_LDT, 2>(k2, z[0], w, i, grp);
      put_slice_t<W2_LDT, 2>(k2 + 8, z[1], w, i, grp);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) put_slice<W2_LDH, 2>(qs + 8 * s_, z[2 + s_], w, i, grp);
    }
    __syncthreads();
    {
      float ks[32];
      load_row<8>(ks, st_k, grp);
      accum_t<W2_LDT, 2>(z[0], dp, dy + 16, qscale, i, grp);
      accum_t<W2_LDT, 2>(z[1], dp, dy + 24, qscale, i, grp);
      f32x4 sa[4];
      zero_acc<4>(sa);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) gemm_hd<W2_LDH, 2>(z[2 + s_], ks + 8 * s_, sa, i, grp);  // S[q][key]
      unpack_at<4>(sa, sp, 0);
    }
    __syncthreads();                   // (delta of all queries has been visible since the second barrier)
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) put_slice<W2_LDH, 2>(dof + 8 * s_, z[s_], w, i, grp);
    put_slice_t<W2_LDT, 2>(dof, z[4], w, i, grp);
    put_slice_t<W2_LDT, 2>(dof + 8, z[5], w, i, grp);
    {
      float vs[32];
      load_row<8>(vs, st_v, grp);
      __syncthreads();
      f32x4 sa[4];
      zero_acc<4>(sa);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) gemm_hd<W2_LDH, 2>(z[s_], vs + 8 * s_, sa, i, grp);      // dP[q][key]
      unpack_at<4>(sa, dp, 0);
    }
    p_ds_keys(sp, dp, lse, delta, sc, grp);
    {
      float dv[16];
      accum_t<W2_LDT, 2>(z[4], sp, dv, 1.0f, i, grp);
      accum_t<W2_LDT, 2>(z[5], sp, dv + 8, 1.0f, i, grp);
      store_slice<4>(dv, st_dkv + D, grp);
    }
    {
      float qs[32];
      load_row<8>(qs, st_q, grp);
      __syncthreads();
      put_slice_t<W2_LDT, 2>(dof + 16, z[0], w, i, grp);
      put_slice_t<W2_LDT, 2>(dof + 24, z[1], w, i, grp);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) put_slice_t<W2_LDT, 2>(qs + 8 * s_, z[2 + s_], w, i, grp);
    }
    __syncthreads();
    {
      float dv[16];
      accum_t<W2_LDT, 2>(z[0], sp, dv, 1.0f, i, grp);
      accum_t<W2_LDT, 2>(z[1], sp, dv + 8, 1.0f, i, grp);
      store_slice<4>(dv, st_dkv + D + 64, grp);
      float dk[32];
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) accum_t<W2_LDT, 2>(z[2 + s_], dp, dk + 8 * s_, qscale, i, grp);
      store_row<8>(dk, st_dkv, grp);
    }
  }
}

// ---- D = 96: heads of 96 as three 32-column slices (NCS = 2), heads of 48 as three 16-column slices (NCS = 1) ----
// Same scheme with NS = 3; a tile slot is 64 x (16 NCS + 4) floats (its transposed image [16 NCS][68] is smaller),
// slots at stride TS from `t` (forward: 3 of them in the head-tile area) / `tz` (backward: 6, from weight slot 1 on).
// Row registers of slice s of head h: [RS (3 h + s), + RS), RS = 4 NCS.
template <int H, int NCS>
__device__ __forceinline__ void wide2_attn_fwd_3(const float* q, const float* kk, const float* v, float* of, float* t, float* lse,
                                                 float* st_p, float sc, int w, int i, int grp) {
  constexpr int RS = 4 * NCS, LDH = 16 * NCS + 4, TS = 64 * LDH;
#pragma unroll
  for (int h = 0; h < H; ++h) {
    __syncthreads();                   // the previous head's tiles are no longer read
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) put_slice<LDH, NCS>(kk + RS * (3 * h + s_), t + s_ * TS, w, i, grp);
    __syncthreads();
    f32x4 sa[4];
    zero_acc<4>(sa);
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) gemm_hd<LDH, NCS>(t + s_ * TS, q + RS * (3 * h + s_), sa, i, grp);     // S[key][q] over the slices
    float pf[16];
    const float l2 = softmax_keys(sa, pf, sc);
    if (lse && grp == 0) lse[64 * h + 16 * w + i] = l2;
    if (st_p) store_row<4>(pf, st_p + 64 * h, grp);
    __syncthreads();                   // every wave is past its reads of the K tiles
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) put_slice_t<W2_LDT, NCS>(v + RS * (3 * h + s_), t + s_ * TS, w, i, grp);
    __syncthreads();
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) accum_t<W2_LDT, NCS>(t + s_ * TS, pf, of + RS * (3 * h + s_), 1.0f, i, grp);
  }
}
// per head three phases of six tiles:  [V_0..2 K_0..2^T] -> dP[key][q], delta, dS, dq     [Q_0..2 dO_0..2] -> S, dP [q][key], P, dS
//                                      [dO_0..2^T Q_0..2^T] -> dV, dK
template <int H, int NCS>
__device__ __forceinline__ void wide2_attn_bwd_3(float* tz, const float* lse, float* delta, const float* st_q, const float* st_k,
                                                 const float* st_v, const float* st_p, float* st_dkv, const float* dof, float* dy,
                                                 float sc, float qscale, int w, int i, int grp) {
  constexpr int RS = 4 * NCS, LDH = 16 * NCS + 4, TS = 64 * LDH, DH = 48 * NCS, D = 96;
  static_assert(H * DH == D, "D = 96");
#pragma unroll
  for (int h = 0; h < H; ++h) {
    float ks[3 * RS], vs[3 * RS], qs[3 * RS], pf[16], dp[16], sp[16];
    load_slice<3 * NCS>(ks, st_k + DH * h, grp);
    load_slice<3 * NCS>(vs, st_v + DH * h, grp);
    load_slice<3 * NCS>(qs, st_q + DH * h, grp);
    load_row<4>(pf, st_p + 64 * h, grp);
    __syncthreads();                   // the previous head's tiles (and weight tile 21 NQ - 1) are no longer read
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {
      put_slice<LDH, NCS>(vs + RS * s_, tz + s_ * TS, w, i, grp);
      put_slice_t<W2_LDT, NCS>(ks + RS * s_, tz + (3 + s_) * TS, w, i, grp);
    }
    __syncthreads();
    {
      f32x4 sa[4];
      zero_acc<4>(sa);
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) gemm_hd<LDH, NCS>(tz + s_ * TS, dof + RS * (3 * h + s_), sa, i, grp);   // dP[key][q]
      unpack_at<4>(sa, dp, 0);
      float dl = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) dl += pf[k] * dp[k];
      dl = grp_sum(dl);
      if (grp == 0) delta[64 * h + 16 * w + i] = dl;
#pragma unroll
      for (int k = 0; k < 16; ++k) dp[k] = pf[k] * (dp[k] - dl);
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) accum_t<W2_LDT, NCS>(tz + (3 + s_) * TS, dp, dy + RS * (3 * h + s_), qscale, i, grp);   // dq
    }
    __syncthreads();
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {
      put_slice<LDH, NCS>(qs + RS * s_, tz + s_ * TS, w, i, grp);
      put_slice<LDH, NCS>(dof + RS * (3 * h + s_), tz + (3 + s_) * TS, w, i, grp);
    }
    __syncthreads();                   // (also: delta of all 64 queries visible)
    {
      f32x4 sa[4];
      zero_acc<4>(sa);
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) gemm_hd<LDH, NCS>(tz + s_ * TS, ks + RS * s_, sa, i, grp);              // S[q][key]
      unpack_at<4>(sa, sp, 0);
      zero_acc<4>(sa);
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) gemm_hd<LDH, NCS>(tz + (3 + s_) * TS, vs + RS * s_, sa, i, grp);        // dP[q][key]
      unpack_at<4>(sa, dp, 0);
      p_ds_keys(sp, dp, lse + 64 * h, delta + 64 * h, sc, grp);
    }
    __syncthreads();
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {
      put_slice_t<W2_LDT, NCS>(dof + RS * (3 * h + s_), tz + s_ * TS, w, i, grp);
      put_slice_t<W2_LDT, NCS>(qs + RS * s_, tz + (3 + s_) * TS, w, i, grp);
    }
    __syncthreads();
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) accum_t<W2_LDT, NCS>(tz + s_ * TS, sp, vs + RS * s_, 1.0f, i, grp);               // dV
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) accum_t<W2_LDT, NCS>(tz + (3 + s_) * TS, dp, ks + RS * s_, qscale, i, grp);      // dK
    store_slice<3 * NCS>(ks, st_dkv + DH * h, grp);
    store_slice<3 * NCS>(vs, st_dkv + D + DH * h, grp);
  }
}

// ---- D = 96, four heads of 24 ---------------------------------------------------------------------------------
// Head h lives in column groups c0 = h + h / 2 and c0 + 1: one whole group and HALF of the group it shares with its
// neighbour (even h: lane groups 0, 1 of the second group; odd h: lane groups 2, 3 of the first).  Its tiles are the
// 32-column slice of those two groups, as for heads of 32; the neighbour's 8 columns in them are harmless because
// every product either multiplies them with a REGISTER operand that is zeroed on the lanes that hold those columns
// (scores: S, dP) or produces results for them that are dropped lane by lane (O, dq, dK, dV).
__device__ __forceinline__ void h24_masked(const float* src8, float* dst8, int h, int grp) {
  const int pg = (h & 1) ? 0 : 1;
  const bool valid = (h & 1) ? (grp >= 2) : (grp < 2);
#pragma unroll
  for (int k = 0; k < 8; ++k) dst8[k] = ((k >> 2) == pg && !valid) ? 0.f : src8[k];
}
// results of a slice into the row registers: the whole group as it is, the shared group on this head's lanes only
// (an even head writes the shared group first, garbage on the other lanes; its odd neighbour then takes its lanes)
__device__ __forceinline__ void h24_merge(const float* out8, float* dst8, int h, int grp) {
  const int pg = (h & 1) ? 0 : 1;
  const bool valid = (h & 1) ? (grp >= 2) : (grp < 2);
#pragma unroll
  for (int k = 0; k < 8; ++k) dst8[k] = ((k >> 2) == pg && (h & 1) && !valid) ? dst8[k] : out8[k];
}
// ... and into a stash row: the shared group's float4 is stored by the lanes that own it
__device__ __forceinline__ void h24_store(const float* f8, float* base, int h, int grp) {   // base = row pointer + 16 c0
  const int pg = (h & 1) ? 0 : 1;
  const bool valid = (h & 1) ? (grp >= 2) : (grp < 2);
#pragma unroll
  for (int c = 0; c < 2; ++c)
    if (c != pg || valid) *reinterpret_cast<float4*>(base + 16 * c + 4 * grp) = make_float4(f8[4 * c], f8[4 * c + 1], f8[4 * c + 2], f8[4 * c + 3]);
}
__device__ __forceinline__ void wide2_attn_fwd_h24(const float* q, const float* kk, const float* v, float* of, float* tA, float* tB,
                                                   float* tC, float* tD, float* lse, float* st_p, float sc, int w, int i, int grp) {
#pragma unroll
  for (int hp = 0; hp < 4; hp += 2) {
    const int ca = hp + hp / 2, cb = (hp + 1) + (hp + 1) / 2;     // first column group of the two heads' slices
    __syncthreads();                   // the previous round's tiles are no longer read
    put_slice<W2_LDH, 2>(kk + 4 * ca, tA, w, i, grp);
    put_slice_t<W2_LDT, 2>(v + 4 * ca, tB, w, i, grp);
    put_slice<W2_LDH, 2>(kk + 4 * cb, tC, w, i, grp);
    put_slice_t<W2_LDT, 2>(v + 4 * cb, tD, w, i, grp);
    __syncthreads();
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int h = hp + hh, c0 = hh ? cb : ca;
      const float* tK = hh ? tC : tA;
      const float* tV = hh ? tD : tB;
      float qm[8], o8[8], pf[16];
      h24_masked(q + 4 * c0, qm, h, grp);
      f32x4 sa[4];
      zero_acc<4>(sa);
      gemm_hd<W2_LDH, 2>(tK, qm, sa, i, grp);                     // S[key][q] over this head's 24 columns
      const float l2 = softmax_keys(sa, pf, sc);
      if (lse && grp == 0) lse[64 * h + 16 * w + i] = l2;
      if (st_p) store_row<4>(pf, st_p + 64 * h, grp);
      accum_t<W2_LDT, 2>(tV, pf, o8, 1.0f, i, grp);
      h24_merge(o8, of + 4 * c0, h, grp);
    }
  }
}
// six tiles per head as for heads of 32 (V, Q, dO row-major; K, Q, dO transposed), at the addresses the caller passes
__device__ __forceinline__ void wide2_attn_bwd_h24(float* bV, float* bKT, float* bQ, float* bdO, float* bQT, float* bdOT, const float* lse,
                                                   float* delta, const float* st_q, const float* st_k, const float* st_v,
                                                   const float* st_p, float* st_dkv, const float* dof, float* dy, float sc,
                                                   float qscale, int w, int i, int grp) {
  constexpr int D = 96;
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const int c0 = h + h / 2;
    float ks[8], vs[8], pf[16], dp[16], dom[8];
    {
      float qs[8];
      load_slice<2>(ks, st_k + 16 * c0, grp);
      load_slice<2>(vs, st_v + 16 * c0, grp);
      load_slice<2>(qs, st_q + 16 * c0, grp);
      load_row<4>(pf, st_p + 64 * h, grp);
      __syncthreads();                 // the previous head's tiles (and weight tile 21 NQ - 1) are no longer read
      put_slice<W2_LDH, 2>(vs, bV, w, i, grp);
      put_slice_t<W2_LDT, 2>(ks, bKT, w, i, grp);
      put_slice<W2_LDH, 2>(qs, bQ, w, i, grp);
      put_slice<W2_LDH, 2>(dof + 4 * c0, bdO, w, i, grp);
      put_slice_t<W2_LDT, 2>(qs, bQT, w, i, grp);
      put_slice_t<W2_LDT, 2>(dof + 4 * c0, bdOT, w, i, grp);
    }
    __syncthreads();
    {
      // own queries: dP[key][q] = V[key] . dO[q] over the head's columns, delta, dS, dq
      h24_masked(dof + 4 * c0, dom, h, grp);
      f32x4 sa[4];
      zero_acc<4>(sa);
      gemm_hd<W2_LDH, 2>(bV, dom, sa, i, grp);
      unpack_at<4>(sa, dp, 0);
      float dl = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) dl += pf[k] * dp[k];
      dl = grp_sum(dl);
      if (grp == 0) delta[64 * h + 16 * w + i] = dl;
#pragma unroll
      for (int k = 0; k < 16; ++k) dp[k] = pf[k] * (dp[k] - dl);
      float o8[8];
      accum_t<W2_LDT, 2>(bKT, dp, o8, qscale, i, grp);
      h24_merge(o8, dy + 4 * c0, h, grp);                          // dq (into dy)
    }
    __syncthreads();                   // delta of all 64 queries visible
    {
      // own keys: S and dP recomputed with the key on the lane; dV = P^T dO, dK = dS^T Q
      float sp[16], km[8], vm[8], o8[8];
      h24_masked(ks, km, h, grp);
      h24_masked(vs, vm, h, grp);
      f32x4 sa[4];
      zero_acc<4>(sa); gemm_hd<W2_LDH, 2>(bQ, km, sa, i, grp); unpack_at<4>(sa, sp, 0);      // S[q][key]
      zero_acc<4>(sa); gemm_hd<W2_LDH, 2>(bdO, vm, sa, i, grp); unpack_at<4>(sa, dp, 0);     // dP[q][key]
      p_ds_keys(sp, dp, lse + 64 * h, delta + 64 * h, sc, grp);
      accum_t<W2_LDT, 2>(bdOT, sp, o8, 1.0f, i, grp);       // dV
      h24_store(o8, st_dkv + D + 16 * c0, h, grp);
      accum_t<W2_LDT, 2>(bQT, dp, o8, qscale, i, grp);      // dK
      h24_store(o8, st_dkv + 16 * c0, h, grp);
    }
  }
}

// LDS carve-up shared by the bodies and the launchers (floats): 2 weight slots | head-tile area | lse, delta | parameters
template <int NCT, int H>
struct W2Cfg {
  static constexpr int D = 16 * NCT, M = 4 * D, DH = D / H;
  static constexpr bool H24 = NCT == 6 && H == 4;                // heads of 24: lane-masked 32-column slices (wide2_attn_*_h24)
  static constexpr int SW = (NCT == 6 && H == 2) ? 16 : 32;      // slice width: 32 columns, 16 for heads of 48
  static constexpr int NS = H24 ? 1 : DH / SW, NCS = SW / 16;
  static constexpr int NOT = NCT == 6 ? 3 : 2;                  // 16-row output groups of a weight tile (32 rows; D = 96: 48)
  static constexpr int SLOT = 16 * NOT * (D + 4), HT = 64 * 36, TS = 64 * (SW + 4);
  static constexpr int FWD_END = 2 * SLOT + 4 * HT;              // forward: four head-tile slots behind the weight slots
  static constexpr int BWD_END = NS > 1 ? SLOT + 6 * TS : 0;     // sliced backward: six slots from weight slot 1 on
  static constexpr int TILE_END = FWD_END > BWD_END ? FWD_END : BWD_END;
  static constexpr int LDS_FLOATS = TILE_END + 2 * H * 64 + M + 4 * D;
  static_assert((H24 || NS * SW == DH) && (NCT == 8 ? (H == 4 || H == 2 || H == 1) : (NCT == 6 && (H == 4 || H == 2 || H == 1))), "shape not built");
  static_assert(2 * LDS_FLOATS * 4 <= 160 * 1024, "two workgroups per CU");
};

}  // namespace

#ifdef CORA_STAMPS
#define W2STAMP(K)                                                                                 \
  do {                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    if (p.dbg) {                                                                                   \
      unsigned long long t_;                                                                       \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                 \
      if (threadIdx.x == 0) p.dbg[(long long)blockIdx.x * 16 + (K)] = t_;                          \
    }                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)
#else
#define W2STAMP(K) do {} while (0)
#endif

// ---- weight tile pipeline (shared by the block body and the chain body below) -----------------------
// The macros name locals both bodies define: tile_src (tile index -> source pointer and row stride), the
// staging registers sr0..sr3, trow / tcol, wbuf, slot, NT (tiles this workgroup walks through) and the
// tile geometry constants.
// (D = 96: a tile row is 24 float4, so 192 threads -- waves 0..2 -- cover the 8 rows of a piece; wave 3 repeats wave 2's
// part, the same values to the same LDS addresses: a per-lane guard instead put exec-mask code around each of the 12
// pieces of a stage, between the MFMAs)
#define PF_ONE(J, REG, SRC, LDW) REG = *reinterpret_cast<const float4*>((SRC) + (long long)(8 * (J)) * (LDW) + (unsigned)(trow * (LDW) + tcol));
#define PB_ONE(J, REG, WB) lds_st4((WB) + 8 * (J) * LD, REG);
#define PREFETCH_ALL(K)                                                                           \
  do {                                                                                            \
    int ld_; const float* s_ = tile_src((K), ld_);                                                \
    PF_ONE(0, sr0, s_, ld_) PF_ONE(1, sr1, s_, ld_) PF_ONE(2, sr2, s_, ld_) PF_ONE(3, sr3, s_, ld_) \
    if constexpr (NPIECE > 4) { PF_ONE(4, sr4, s_, ld_) PF_ONE(5, sr5, s_, ld_) }                 \
  } while (0)
#define WRITE_ALL(S)                                                                              \
  do {                                                                                            \
    lds_float* wb_ = opaque_w(wbuf + (S) * SLOT + trow * LD + tcol);                              \
    PB_ONE(0, sr0, wb_) PB_ONE(1, sr1, wb_) PB_ONE(2, sr2, wb_) PB_ONE(3, sr3, wb_)               \
    if constexpr (NPIECE > 4) { PB_ONE(4, sr4, wb_) PB_ONE(5, sr5, wb_) }                         \
  } while (0)
  // column group C of the multiplication with tile k; every PSTEP-th group (8 groups: the even ones, 6 groups: the
  // first four) carries one staged float4
#define GEMM_CT(C, XF, ACC)                                                                       \
  {                                                                                               \
    if ((C) + 1 < NCT) {                                                                          \
      _Pragma("unroll") for (int ot = 0; ot < NOT; ++ot)                                          \
        a_[((C) + 1) & 1][ot] = lds_ld4(&tl_[16 * ot * LD + 16 * ((C) + 1)]);                     \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);   /* the reads of group C+1 stay AHEAD of group C's MFMAs */ \
    if ((C) == 0 * PSTEP) { if (do_w_) { PB_ONE(0, sr0, wb_) } if (do_p_) { PF_ONE(0, sr0, ns_, nld_) } } \
    if ((C) == 1 * PSTEP) { if (do_w_) { PB_ONE(1, sr1, wb_) } if (do_p_) { PF_ONE(1, sr1, ns_, nld_) } } \
    if ((C) == 2 * PSTEP) { if (do_w_) { PB_ONE(2, sr2, wb_) } if (do_p_) { PF_ONE(2, sr2, ns_, nld_) } } \
    if ((C) == 3 * PSTEP) { if (do_w_) { PB_ONE(3, sr3, wb_) } if (do_p_) { PF_ONE(3, sr3, ns_, nld_) } } \
    if constexpr (NPIECE > 4) {                                                                   \
      if ((C) == 4 * PSTEP) { if (do_w_) { PB_ONE(4, sr4, wb_) } if (do_p_) { PF_ONE(4, sr4, ns_, nld_) } } \
      if ((C) == 5 * PSTEP) { if (do_w_) { PB_ONE(5, sr5, wb_) } if (do_p_) { PF_ONE(5, sr5, ns_, nld_) } } \
    }                                                                                             \
    _Pragma("unroll") for (int ot = 0; ot < NOT; ++ot) (ACC)[ot] = mfma16(a_[(C) & 1][ot].x, (XF)[4 * (C) + 0], (ACC)[ot]); \
    _Pragma("unroll") for (int ot = 0; ot < NOT; ++ot) (ACC)[ot] = mfma16(a_[(C) & 1][ot].y, (XF)[4 * (C) + 1], (ACC)[ot]); \
    _Pragma("unroll") for (int ot = 0; ot < NOT; ++ot) (ACC)[ot] = mfma16(a_[(C) & 1][ot].z, (XF)[4 * (C) + 2], (ACC)[ot]); \
    _Pragma("unroll") for (int ot = 0; ot < NOT; ++ot) (ACC)[ot] = mfma16(a_[(C) & 1][ot].w, (XF)[4 * (C) + 3], (ACC)[ot]); \
    __builtin_amdgcn_sched_barrier(0);                                                            \
  }
#define STAGE(K, XF, ACC)                                                                         \
  do {                                                                                            \
    const int k_ = (K);                                                                           \
    const bool do_w_ = (k_ + 1 < NT);                                                             \
    const bool do_p_ = (k_ + 2 < NT);                                                             \
    int nld_ = D; const float* ns_ = do_p_ ? tile_src(k_ + 2, nld_) : nullptr;                    \
    __syncthreads();                                                                              \
    const lds_float* tl_ = opaque(wbuf + slot * SLOT + i * LD + 4 * grp);                         \
    lds_float* wb_ = opaque_w(wbuf + (slot ^ 1) * SLOT + trow * LD + tcol);                       \
    float4 a_[2][NOT];                                                                            \
    _Pragma("unroll") for (int ot = 0; ot < NOT; ++ot) a_[0][ot] = lds_ld4(&tl_[16 * ot * LD]);   \
    GEMM_CT(0, XF, ACC) GEMM_CT(1, XF, ACC) GEMM_CT(2, XF, ACC) GEMM_CT(3, XF, ACC)               \
    GEMM_CT(4, XF, ACC) GEMM_CT(5, XF, ACC)                                                       \
    if constexpr (NCT > 6) { GEMM_CT(6, XF, ACC) GEMM_CT(7, XF, ACC) }                            \
    slot ^= 1;                                                                                    \
    (void)wb_; (void)ns_; (void)nld_;                                                             \
  } while (0)
// one GEMM with D outputs: NQ stages, quarter j -> registers [8 j, 8 j + 8) of OUT
#define GEMM_D(K0, XF, OUT)                                                                       \
  _Pragma("unroll") for (int j_ = 0; j_ < NQ; ++j_) {                                             \
    f32x4 acc_[NOT]; zero_acc<NOT>(acc_);                                                         \
    STAGE((K0) + j_, XF, acc_);                                                                   \
    unpack_at<NOT>(acc_, OUT, NOT * j_);                                                          \
  }
// the same, accumulating into ACCD[NCT] over several K-chunks
#define GEMM_D_ACC(K0, XF, ACCD)                                                                  \
  _Pragma("unroll") for (int j_ = 0; j_ < NQ; ++j_) STAGE((K0) + j_, XF, (ACCD) + NOT * j_);


// LAZY (FusedP::lazy): the rows come in with the previous step's vode update applied -- the input vode's new
// value and momentum are written to the new arenas, the target's updated rows wait in this workgroup's slot of the error
// tensor until the forward needs them.
template <int NCT, int H, bool LAZY>
__device__ __forceinline__ void wide2_block_body(const FusedP& p, float* lds, int g, int b) {
  constexpr int D = 16 * NCT, M = 4 * D, R = 4 * NCT;
  constexpr int NOT = W2Cfg<NCT, H>::NOT, TR = 16 * NOT;   // weight tile: 32 output rows x D inputs (D = 96: 48 rows)
  constexpr int NPIECE = TR / 8;                  // staged float4 per thread and tile (8 tile rows per piece)
  constexpr int NQ = D / TR;                      // tiles per D outputs (4)
  constexpr int LD = D + 4, SLOT = TR * LD;
  // (four staged float4 per thread and tile: 8 tile rows per piece)
  using Cfg = W2Cfg<NCT, H>;
  constexpr int DH = D / H, NS = Cfg::NS;         // slices per head: 32 columns each (16 for heads of 48)
  constexpr int NCH = 2, RH = 8;                  // heads-of-32 form: a head = one slice of 2 column groups, 8 registers
  constexpr int LDH = 32 + 4, HT = 64 * LDH;      // head tile [64][36]
  constexpr int LDT = 64 + 4;                     // transposed head tile [32][68] (fits the same slot)
  static_assert(32 * LDT <= HT && LDH == W2_LDH && LDT == W2_LDT && SLOT == Cfg::SLOT && HT == Cfg::HT, "transposed tile fits a head-tile slot");
  constexpr int NT = 24 * NQ;                     // 96 tiles (D = 96: 72)
  constexpr int PSTEP = NCT / NPIECE;             // staged pieces ride on column groups 0, PSTEP, 2 PSTEP, ...
  constexpr int SLOT0 = (21 * NQ) & 1;            // first tile's slot: chosen so that tile 21 NQ sits in slot 0 and slot 1 --
                                                  // next to the head tiles -- idles during the attention backward
  static_assert(NCT == 8 || NCT == 6, "weight-tile pipeline: 8 or 6 column groups");
  const float qscale = rsqrtf((float)DH);
  const float sc = qscale * CORA_LOG2E;

  float* wbuf = lds;                     // 2 weight slots
  float* tA = lds + 2 * SLOT;            // head tiles: forward K_h V_h K_h' V_h' ; backward K_h V_h Q_h dO_h
  float* tB = tA + HT;
  float* tC = tB + HT;
  float* tD = tC + HT;
  float* rowc = lds + Cfg::TILE_END;     // lse2[H][64], delta[H][64]
  float* pb = rowc + 2 * H * 64;         // [b1 M | ln2 gamma D | ln2 beta D | b2 D | ln1 gamma D]
  constexpr int PB_LN2G = M, PB_LN2B = M + D, PB_B2 = M + 2 * D, PB_LN1G = M + 3 * D;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, grp = lane >> 4;
  const long long nl = (long long)p.B * 64 * D;
  const long long rowoff = ((long long)b * 64 + 16 * w + i) * D;
  const long long voff = (long long)g * nl + rowoff;
  const float* xrow = p.H1 + voff;
  const float* hnext = p.H1 + voff + nl;
  const float* xcur = LAZY ? p.Hn1 + voff : xrow;                          // where the later reloads of x come from
  const float* hncur = LAZY ? p.E2 + (long long)g * nl + rowoff : hnext;   // ... and the target rows
  const long long srow_g = ((long long)g * p.B + b) * 64 + 16 * w + i;
  float* st_p = p.S_stash + srow_g * (H * 64 + 3 * D);   // per token row: P (64 keys per head) | q | k | v
  float* st_q = st_p + H * 64;
  float* st_k = st_q + D;
  float* st_v = st_k + D;
  float* st_a = p.S_da + srow_g * M;                     // gelu'(a)
  float* st_dkv = p.S_dqkv + srow_g * 3 * D;             // dk | dv between the attention backward and the last GEMM

  const float* Wqkv = p.W + p.oWqkv + (long long)g * 3 * D * D;
  const float* Wo = p.W + p.oWo + (long long)g * D * D;
  const float* W1 = p.W + p.oW1 + (long long)g * M * D;
  const float* W2 = p.W + p.oW2 + (long long)g * D * M;
  const float* W2T = p.WT + p.tW2 + (long long)g * M * D;          // [M][D]
  const float* W1T = p.WT + p.tW1 + (long long)g * D * M;          // [D][M]
  const float* WoT = p.WT + p.tWo + (long long)g * D * D;          // [D][D]
  const float* WqkvT = p.WT + p.tWqkv + (long long)g * D * 3 * D;  // [D][3D]
  const float* ln1g = p.W + p.oLn1g + (long long)g * D;
  const float* ln1b = p.W + p.oLn1b + (long long)g * D;
  {
    constexpr int NP4 = (M + 4 * D) / 4;
    for (int f = tid; f < NP4; f += 256) {
      const int o = 4 * f;
      const float* src = o < M ? p.W + p.ob1 + (long long)g * M + o
                       : o < M + D ? p.W + p.oLn2g + (long long)g * D + (o - M)
                       : o < M + 2 * D ? p.W + p.oLn2b + (long long)g * D + (o - M - D)
                       : o < M + 3 * D ? p.W + p.ob2 + (long long)g * D + (o - M - 2 * D)
                                       : ln1g + (o - M - 3 * D);
      *reinterpret_cast<float4*>(pb + o) = *reinterpret_cast<const float4*>(src);   // visible after the first tile barrier
    }
  }

  // ---- weight tile pipeline (see fused_wide.hip) ---------------------------------------------------------
  // float4 #f of a tile = row f / (D/4), 16-byte column f % (D/4); thread t stages #t + 256 J: row
  // t / 32 + 8 J, so piece J sits 8 rows further in both images: a scalar base per piece plus one
  // per-thread offset per tile (global), immediates (LDS).
  auto tile_src = [&](int k, int& ldw) -> const float* {
    // forward: 0-11 Wqkv | 12-15 Wo | 16-47 MLP (per hidden chunk mt: 4 x W1 rows, 4 x W2 output quarters)
    // backward: 48-79 MLP (4 x W2^T rows, 4 x W1^T quarters per chunk) | 80-83 Wo^T | 84-95 Wqkv^T (dq, dk, dv x 4 quarters)
    if (k < 3 * NQ) { ldw = D; return Wqkv + (long long)k * TR * D; }
    if (k < 4 * NQ) { ldw = D; return Wo + (long long)(k - 3 * NQ) * TR * D; }
    if (k < 12 * NQ) {
      const int mt = (k - 4 * NQ) / (2 * NQ), j = (k - 4 * NQ) % (2 * NQ);
      if (j < NQ) { ldw = D; return W1 + ((long long)mt * D + j * TR) * D; }
      ldw = M; return W2 + (long long)(j - NQ) * TR * M + (long long)mt * D;
    }
    if (k < 20 * NQ) {
      const int mt = (k - 12 * NQ) / (2 * NQ), j = (k - 12 * NQ) % (2 * NQ);
      if (j < NQ) { ldw = D; return W2T + ((long long)mt * D + j * TR) * D; }
      ldw = M; return W1T + (long long)(j - NQ) * TR * M + (long long)mt * D;
    }
    if (k < 21 * NQ) { ldw = D; return WoT + (long long)(k - 20 * NQ) * TR * D; }
    const int j = k - 21 * NQ;
    ldw = 3 * D; return WqkvT + (long long)(j % NQ) * TR * 3 * D + (long long)(j / NQ) * D;
  };
  const int tst = (8 * (D / 4) >= 256 || tid < 8 * (D / 4)) ? tid : tid - 64;     // staging thread index (see PF_ONE)
  const int trow = tst / (D / 4), tcol = 4 * (tst % (D / 4));
  float4 sr0, sr1, sr2, sr3, sr4, sr5;
  (void)sr4; (void)sr5;
  int slot = SLOT0;

  PREFETCH_ALL(0);
  WRITE_ALL(SLOT0);                // tile 0 -> its slot (visible after stage 0's barrier)
  PREFETCH_ALL(1);
  W2STAMP(0);

  // ---- forward ---------------------------------------------------------------------------------------
  float y[R];
  float cn_lazy = 0.f, eta_lazy = 0.f;    // LAZY: multiplier of the target vode's update and the step size
  {
    float q[R], kk[R], v[R];
    {
      float xin[R];
      {
        float x[R], gm[R], bt[R], xh[R + 1];
        load_row<NCT>(x, xrow, grp);
        if (LAZY) {
          float cx, cn, c3;
          vode_multipliers_wave(p.up, g + 1, g + 2, g + 1, cx, cn, c3);
          const float eta = p.up.eta[p.up.t], mu = p.up.mu;
          cn_lazy = cn; eta_lazy = eta;
          float* gx = gm; float* mx = bt;      // (the LayerNorm parameters are loaded after)
          load_row<NCT>(gx, p.Go1 + voff, grp); load_row<NCT>(mx, p.Mo1 + voff, grp);
#pragma unroll
          for (int k = 0; k < R; ++k) { mx[k] = gx[k] * cx + mu * mx[k]; x[k] -= eta * mx[k]; }
          store_row<NCT>(mx, p.Mn1 + voff, grp);
          store_row<NCT>(x, p.Hn1 + voff, grp);
        }
        load_row<NCT>(gm, ln1g, grp);
        load_row<NCT>(bt, ln1b, grp);
        ln_row<NCT>(x, gm, bt, xh, xin);
      }
      W2STAMP(1);
      GEMM_D(0, xin, q)
      GEMM_D(NQ, xin, kk)
      GEMM_D(2 * NQ, xin, v)
    }
    store_row<NCT>(q, st_q, grp);       // q, k, v wait in the stash for the attention backward
    store_row<NCT>(kk, st_k, grp);
    store_row<NCT>(v, st_v, grp);
    W2STAMP(2);
    // attention, two heads per round: K_h, V_h slices of all 64 tokens in the head tiles
    float of[R];
    if constexpr (Cfg::H24) {
      wide2_attn_fwd_h24(q, kk, v, of, tA, tB, tC, tD, rowc, st_p, sc, w, i, grp);
    } else if constexpr (NS == 3) {
      wide2_attn_fwd_3<H, Cfg::NCS>(q, kk, v, of, tA, rowc, st_p, sc, w, i, grp);
    } else if constexpr (NS > 1) {
      wide2_attn_fwd_sliced<H>(q, kk, v, of, tA, tB, tC, tD, rowc, st_p, sc, w, i, grp);
    } else {
#pragma unroll
    for (int hp = 0; hp < H; hp += 2) {
      __syncthreads();                   // the previous round's tiles are no longer read
      put_slice<LDH, NCH>(kk + RH * hp, tA, w, i, grp);
      put_slice_t<LDT, NCH>(v + RH * hp, tB, w, i, grp);
      put_slice<LDH, NCH>(kk + RH * (hp + 1), tC, w, i, grp);
      put_slice_t<LDT, NCH>(v + RH * (hp + 1), tD, w, i, grp);
      __syncthreads();
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int h = hp + hh;
        const float* tK = hh ? tC : tA;
        const float* tV = hh ? tD : tB;
        f32x4 sa[4];
        zero_acc<4>(sa);
        gemm_hd<LDH, NCH>(tK, q + RH * h, sa, i, grp);           // S[key][q]
        float pf[16];
        unpack_at<4>(sa, pf, 0);
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 16; ++k) { pf[k] *= sc; mx = fmaxf(mx, pf[k]); }
        mx = grp_max(mx);
        float ls = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { pf[k] = __builtin_amdgcn_exp2f(pf[k] - mx); ls += pf[k]; }
        ls = grp_sum(ls);
        const float inv = 1.0f / ls;
#pragma unroll
        for (int k = 0; k < 16; ++k) pf[k] *= inv;
        if (grp == 0) rowc[64 * h + 16 * w + i] = mx + __builtin_amdgcn_logf(ls);
        store_row<4>(pf, st_p + 64 * h, grp);
        accum_t<LDT, NCH>(tV, pf, of + RH * h, 1.0f, i, grp);    // O, this head's columns (V_h transposed)
      }
    }
    }
    W2STAMP(3);
    if (LAZY) {   // the target vode's rows with the deferred update applied, parked in this workgroup's slot of the error
                  // tensor until the end of the MLP (here q, k, v are dead and only `of` is live)
      const float mu = p.up.mu;
      float hn[R], gn[R], mn[R];
      load_row<NCT>(hn, hnext, grp); load_row<NCT>(gn, p.Go1 + voff + nl, grp); load_row<NCT>(mn, p.Mo1 + voff + nl, grp);
#pragma unroll
      for (int k = 0; k < R; ++k) hn[k] -= eta_lazy * (gn[k] * cn_lazy + mu * mn[k]);
      store_row<NCT>(hn, p.E2 + (long long)g * nl + rowoff, grp);
    }
    // y = x + att Wo^T
    GEMM_D(3 * NQ, of, y)
    {
      float x[R];
      load_row<NCT>(x, xcur, grp);        // x again (L2): not kept live across the attention
#pragma unroll
      for (int k = 0; k < R; ++k) y[k] += x[k];
    }
  }
  float r[R];
  {
    float zn[R];
    {
      float gm[R], bt[R], xh[R + 1];
      load_row<NCT>(gm, pb + PB_LN2G, grp);
      load_row<NCT>(bt, pb + PB_LN2B, grp);
      ln_row<NCT>(y, gm, bt, xh, zn);
    }
    W2STAMP(4);
    // hidden chunk mt (D units): a = LN2(y) W1[mt]^T + b1, z += gelu(a) W2[:, mt]^T
    f32x4 zacc[NCT];
    zero_acc<NCT>(zacc);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float am[R];
      GEMM_D(4 * NQ + 2 * NQ * mt, zn, am)
      {
        float b1[R];
        load_row<NCT>(b1, pb + mt * D, grp);
#pragma unroll
        for (int k = 0; k < R; k += 2) {
          f32x2 g2, gp2;
          cora_gelu_both2((f32x2){am[k] + b1[k], am[k + 1] + b1[k + 1]}, g2, gp2);
          am[k] = g2.x; am[k + 1] = g2.y;
          b1[k] = gp2.x; b1[k + 1] = gp2.y;
        }
        store_row<NCT>(b1, st_a + mt * D, grp);     // gelu'(a) waits in the stash for the backward
      }
      if (mt == 3) {      // x again and the next vode's rows: their latency runs under the last four tiles
        load_row<NCT>(r, xcur, grp);
        load_row<NCT>(zn, hncur, grp);
      }
      GEMM_D_ACC(5 * NQ + 2 * NQ * mt, am, zacc)
    }
    W2STAMP(5);
    float z[R], b2[R];
    unpack_at<NCT>(zacc, z, 0);
    load_row<NCT>(b2, pb + PB_B2, grp);
#pragma unroll
    for (int k = 0; k < R; ++k) r[k] = zn[k] - (r[k] + y[k] + z[k] + b2[k]);          // e_{g+2} = h_next - u  (r held x, zn held h_next)
    store_row<NCT>(r, p.E2 + (long long)g * nl + rowoff, grp);
  }
  W2STAMP(6);

  // ---- backward (cotangent r) -----------------------------------------------------------------------
  float dy[R];
  {
    // hidden chunk mt: da = (r W2)[mt] * gelu'(a), dzn += da W1[mt]
    f32x4 dacc[NCT];
    zero_acc<NCT>(dacc);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float dg[R], gp[R];
      load_row<NCT>(gp, st_a + mt * D, grp);       // gelu'(a) of this chunk, back from the stash under the four tiles
      GEMM_D(12 * NQ + 2 * NQ * mt, r, dg)
#pragma unroll
      for (int k = 0; k < R; ++k) dg[k] *= gp[k];
      GEMM_D_ACC(13 * NQ + 2 * NQ * mt, dg, dacc)
    }
    W2STAMP(7);
    // dy = r + LN2^T(dzn); LN2's normalised input is recomputed from y (still in registers)
    float dzn[R], gm[R], xh[R + 1], tmp[R];
    unpack_at<NCT>(dacc, dzn, 0);
    load_row<NCT>(gm, pb + PB_LN2G, grp);
    ln_row<NCT>(y, gm, nullptr, xh, tmp);
    ln_bwd_row<NCT>(dzn, gm, xh, dy);
#pragma unroll
    for (int k = 0; k < R; ++k) dy[k] += r[k];
  }
  W2STAMP(8);
  {
    // dO = dy Wo; from here on r + dy is all that is needed of either
    float dof[R];
    GEMM_D(20 * NQ, dy, dof)
#pragma unroll
    for (int k = 0; k < R; ++k) r[k] += dy[k];
    W2STAMP(9);
    // attention backward, head by head.  Six head tiles: V_h, Q_h, dO_h row-major (the score-shaped products
    // contract over the head dim) and K_h, Q_h, dO_h transposed (dq, dK, dV sum over tokens: one 128-bit
    // LDS read per four MFMAs).  They fill the four forward tile slots plus the weight slot that is free
    // here (tile 21 NQ already sits in slot 0, tile 21 NQ + 1 in registers).  dq lands in dy, dk / dv go
    // to the stash until the last GEMM reads them.
    static_assert((21 * NQ + SLOT0) % 2 == 0 && (NS > 1 || 3 * HT + 3 * 32 * LDT <= SLOT + 4 * HT), "six backward head tiles fit slot 1 + the head tiles");
    float* bV = wbuf + SLOT;
    float* bKT = bV + HT;
    float* bQ = bKT + 32 * LDT;
    float* bdO = bQ + HT;
    float* bQT = bdO + HT;
    float* bdOT = bQT + 32 * LDT;
    float* delta = rowc + 64 * H;
    if constexpr (Cfg::H24) {
      wide2_attn_bwd_h24(bV, bKT, bQ, bdO, bQT, bdOT, rowc, delta, st_q, st_k, st_v, st_p, st_dkv, dof, dy, sc, qscale, w, i, grp);
    } else if constexpr (NS == 3) {
      wide2_attn_bwd_3<H, Cfg::NCS>(wbuf + SLOT, rowc, delta, st_q, st_k, st_v, st_p, st_dkv, dof, dy, sc, qscale, w, i, grp);
    } else if constexpr (NS > 1) {
      // heads of 64 / 128: the same tiles, 32-column slice by slice, scores accumulated across a head's slices
      wide2_attn_bwd_sliced<H>(wbuf + SLOT, rowc, delta, st_q, st_k, st_v, st_p, st_dkv, dof, dy, sc, qscale, w, i, grp);
    } else {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float ks[RH], vs[RH], pf[16], dp[16];
      {
        float qs[RH];
        load_slice<NCH>(ks, st_k + 16 * NCH * h, grp);
        load_slice<NCH>(vs, st_v + 16 * NCH * h, grp);
        load_slice<NCH>(qs, st_q + 16 * NCH * h, grp);
        load_row<4>(pf, st_p + 64 * h, grp);
        __syncthreads();                 // the previous head's tiles (and weight tile 21 NQ - 1) are no longer read
        put_slice<LDH, NCH>(vs, bV, w, i, grp);
        put_slice_t<LDT, NCH>(ks, bKT, w, i, grp);
        put_slice<LDH, NCH>(qs, bQ, w, i, grp);
        put_slice<LDH, NCH>(dof + RH * h, bdO, w, i, grp);
        put_slice_t<LDT, NCH>(qs, bQT, w, i, grp);
        put_slice_t<LDT, NCH>(dof + RH * h, bdOT, w, i, grp);
      }
      __syncthreads();
      {
        // pass A (own queries): dP[key][q] = V[key] . dO[q], delta = sum_key P dP, dS, dq
        f32x4 sa[4];
        zero_acc<4>(sa);
        gemm_hd<LDH, NCH>(bV, dof + RH * h, sa, i, grp);
        unpack_at<4>(sa, dp, 0);
        float dl = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) dl += pf[k] * dp[k];
        dl = grp_sum(dl);
        if (grp == 0) delta[64 * h + 16 * w + i] = dl;
#pragma unroll
        for (int k = 0; k < 16; ++k) dp[k] = pf[k] * (dp[k] - dl);    // dS
        accum_t<LDT, NCH>(bKT, dp, dy + RH * h, qscale, i, grp);      // dq (into dy)
      }
      __syncthreads();                   // delta of all 64 queries visible
      {
        // pass B (own keys): S and dP recomputed with the key on the lane; dV = P^T dO, dK = dS^T Q
        float sp[16];
        f32x4 sa[4];
        zero_acc<4>(sa); gemm_hd<LDH, NCH>(bQ, ks, sa, i, grp); unpack_at<4>(sa, sp, 0);      // S[q][key]
        zero_acc<4>(sa); gemm_hd<LDH, NCH>(bdO, vs, sa, i, grp); unpack_at<4>(sa, dp, 0);     // dP[q][key]
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const float4 l4 = *reinterpret_cast<const float4*>(&rowc[64 * h + 16 * tt + 4 * grp]);
          const float4 d4 = *reinterpret_cast<const float4*>(&delta[64 * h + 16 * tt + 4 * grp]);
          float pv;
          pv = __builtin_amdgcn_exp2f(sp[4 * tt + 0] * sc - l4.x); sp[4 * tt + 0] = pv; dp[4 * tt + 0] = pv * (dp[4 * tt + 0] - d4.x);
          pv = __builtin_amdgcn_exp2f(sp[4 * tt + 1] * sc - l4.y); sp[4 * tt + 1] = pv; dp[4 * tt + 1] = pv * (dp[4 * tt + 1] - d4.y);
          pv = __builtin_amdgcn_exp2f(sp[4 * tt + 2] * sc - l4.z); sp[4 * tt + 2] = pv; dp[4 * tt + 2] = pv * (dp[4 * tt + 2] - d4.z);
          pv = __builtin_amdgcn_exp2f(sp[4 * tt + 3] * sc - l4.w); sp[4 * tt + 3] = pv; dp[4 * tt + 3] = pv * (dp[4 * tt + 3] - d4.w);
        }
        accum_t<LDT, NCH>(bdOT, sp, vs, 1.0f, i, grp);       // dV (this head's columns)
        accum_t<LDT, NCH>(bQT, dp, ks, qscale, i, grp);      // dK
        store_slice<NCH>(ks, st_dkv + 16 * NCH * h, grp);
        store_slice<NCH>(vs, st_dkv + D + 16 * NCH * h, grp);
      }
    }
    }
  }
  W2STAMP(11);
  {
    // dxn = [dq | dk | dv] Wqkv: three K-chunks of Wqkv^T [D][3D]; dx = r + dy + LN1^T(dxn)
    f32x4 xacc[NCT];
    zero_acc<NCT>(xacc);
    GEMM_D_ACC(21 * NQ, dy, xacc)
    load_row<NCT>(dy, st_dkv, grp);             // dk (this wave wrote it itself: same lanes, no barrier needed)
    GEMM_D_ACC(22 * NQ, dy, xacc)
    load_row<NCT>(dy, st_dkv + D, grp);         // dv
    GEMM_D_ACC(23 * NQ, dy, xacc)
    W2STAMP(12);
    float dxn[R], gm[R], xh[R + 1], tmp[R], x[R];
    load_row<NCT>(x, xcur, grp);                 // LN1's normalised input is recomputed from x
    unpack_at<NCT>(xacc, dxn, 0);
    load_row<NCT>(gm, pb + PB_LN1G, grp);
    ln_row<NCT>(x, gm, nullptr, xh, tmp);
    ln_bwd_row<NCT>(dxn, gm, xh, dy);
#pragma unroll
    for (int k = 0; k < R; ++k) dy[k] += r[k];   // r holds r + dy
    store_row<NCT>(dy, p.BP1 + (long long)g * nl + rowoff, grp);
    W2STAMP(13);
  }
}

// ---- chain: status "forward" (/root/reference/src/decoder_transformer.py:232-241, 488-494) -------------
// One workgroup walks ONE image through all L blocks: x_{g+1} = u_{g+2}(x_g) = 2 x + MHA(LN1 x) + MLP(LN2(...)),
// the forward half of the block body on the same weight-tile pipeline (12 NQ = 48 tiles per block, the
// prefetch running across block boundaries).  Used for the INIT pass (every u stored as the next vode) and
// for the reconstruction after the weight update; replaces 7 launches per block on the per-op kernels
// (12 800-row GEMMs that fill the chip for 12..35 us each, 84 launches and their gaps: 1.8 ms at L = 12).
// Block g reads its rows from (g == 0 ? xin : xout + (g - 1) stride) and writes them to xout + g stride;
// with stride = 0 and xin == xout the walk is in place (a lane re-reads only rows it wrote itself).
template <int NCT, int H>
__device__ __forceinline__ void wide2_chain_body(const FusedP& p, float* lds, int b, const float* xin, float* xout, long long stride) {
  constexpr int D = 16 * NCT, M = 4 * D, R = 4 * NCT;
  constexpr int NOT = W2Cfg<NCT, H>::NOT, TR = 16 * NOT, NPIECE = TR / 8;
  constexpr int NQ = D / TR;
  constexpr int LD = D + 4, SLOT = TR * LD;
  using Cfg = W2Cfg<NCT, H>;
  constexpr int DH = D / H, NS = Cfg::NS, NCH = 2, RH = 8;
  constexpr int LDH = 32 + 4, HT = 64 * LDH;
  constexpr int LDT = 64 + 4;
  constexpr int TPB = 12 * NQ;                    // forward tiles per block (48; D = 96: 36)
  constexpr int PSTEP = NCT / NPIECE;
  const int NT = TPB * p.L;
  const float sc = rsqrtf((float)DH) * CORA_LOG2E;

  float* wbuf = lds;
  float* tA = lds + 2 * SLOT;
  float* tB = tA + HT;
  float* tC = tB + HT;
  float* tD = tC + HT;
  float* pb = lds + Cfg::TILE_END + 2 * H * 64;      // same carve-up as the block body: [b1 M | ln2 gamma D | ln2 beta D | b2 D]
  constexpr int PB_LN2G = M, PB_LN2B = M + D, PB_B2 = M + 2 * D;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, grp = lane >> 4;
  const long long rowoff = ((long long)b * 64 + 16 * w + i) * D;

  auto tile_src = [&](int k, int& ldw) -> const float* {
    const int gg = k / TPB, kk = k - gg * TPB;
    if (kk < 3 * NQ) { ldw = D; return p.W + p.oWqkv + (long long)gg * 3 * D * D + (long long)kk * TR * D; }
    if (kk < 4 * NQ) { ldw = D; return p.W + p.oWo + (long long)gg * D * D + (long long)(kk - 3 * NQ) * TR * D; }
    const int mt = (kk - 4 * NQ) / (2 * NQ), j = (kk - 4 * NQ) % (2 * NQ);
    if (j < NQ) { ldw = D; return p.W + p.oW1 + (long long)gg * M * D + ((long long)mt * D + j * TR) * D; }
    ldw = M; return p.W + p.oW2 + (long long)gg * D * M + (long long)(j - NQ) * TR * M + (long long)mt * D;
  };
  const int tst = (8 * (D / 4) >= 256 || tid < 8 * (D / 4)) ? tid : tid - 64;     // staging thread index (see PF_ONE)
  const int trow = tst / (D / 4), tcol = 4 * (tst % (D / 4));
  float4 sr0, sr1, sr2, sr3, sr4, sr5;
  (void)sr4; (void)sr5;
  int slot = 0;
  PREFETCH_ALL(0);
  WRITE_ALL(0);
  PREFETCH_ALL(1);

  for (int g = 0; g < p.L; ++g) {
    const float* xrow = (g == 0 ? xin : xout + (long long)(g - 1) * stride) + rowoff;
    float* urow = xout + (long long)g * stride + rowoff;
    const int k0 = g * TPB;
    __syncthreads();                     // the previous block's parameter vectors and head tiles are no longer read
    {
      constexpr int NP4 = (M + 3 * D) / 4;
      for (int f = tid; f < NP4; f += 256) {
        const int o = 4 * f;
        const float* src = o < M ? p.W + p.ob1 + (long long)g * M + o
                         : o < M + D ? p.W + p.oLn2g + (long long)g * D + (o - M)
                         : o < M + 2 * D ? p.W + p.oLn2b + (long long)g * D + (o - M - D)
                                         : p.W + p.ob2 + (long long)g * D + (o - M - 2 * D);
        *reinterpret_cast<float4*>(pb + o) = *reinterpret_cast<const float4*>(src);   // visible after the next tile barrier
      }
    }
    float y[R];
    {
      float q[R], kk[R], v[R];
      {
        float xn[R];
        {
          float x[R], gm[R], bt[R], xh[R + 1];
          load_row<NCT>(x, xrow, grp);
          load_row<NCT>(gm, p.W + p.oLn1g + (long long)g * D, grp);
          load_row<NCT>(bt, p.W + p.oLn1b + (long long)g * D, grp);
          ln_row<NCT>(x, gm, bt, xh, xn);
        }
        GEMM_D(k0, xn, q)
        GEMM_D(k0 + NQ, xn, kk)
        GEMM_D(k0 + 2 * NQ, xn, v)
      }
      float of[R];
      if constexpr (Cfg::H24) {
        wide2_attn_fwd_h24(q, kk, v, of, tA, tB, tC, tD, nullptr, nullptr, sc, w, i, grp);
      } else if constexpr (NS == 3) {
        wide2_attn_fwd_3<H, Cfg::NCS>(q, kk, v, of, tA, nullptr, nullptr, sc, w, i, grp);
      } else if constexpr (NS > 1) {
        wide2_attn_fwd_sliced<H>(q, kk, v, of, tA, tB, tC, tD, nullptr, nullptr, sc, w, i, grp);
      } else {
#pragma unroll
      for (int hp = 0; hp < H; hp += 2) {
        __syncthreads();                 // the previous round's tiles are no longer read
        put_slice<LDH, NCH>(kk + RH * hp, tA, w, i, grp);
        put_slice_t<LDT, NCH>(v + RH * hp, tB, w, i, grp);
        put_slice<LDH, NCH>(kk + RH * (hp + 1), tC, w, i, grp);
        put_slice_t<LDT, NCH>(v + RH * (hp + 1), tD, w, i, grp);
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int h = hp + hh;
          const float* tK = hh ? tC : tA;
          const float* tV = hh ? tD : tB;
          f32x4 sa[4];
          zero_acc<4>(sa);
          gemm_hd<LDH, NCH>(tK, q + RH * h, sa, i, grp);           // S[key][q]
          float pf[16];
          unpack_at<4>(sa, pf, 0);
          float mx = -INFINITY;
#pragma unroll
          for (int k = 0; k < 16; ++k) { pf[k] *= sc; mx = fmaxf(mx, pf[k]); }
          mx = grp_max(mx);
          float ls = 0.f;
#pragma unroll
          for (int k = 0; k < 16; ++k) { pf[k] = __builtin_amdgcn_exp2f(pf[k] - mx); ls += pf[k]; }
          ls = grp_sum(ls);
          const float inv = 1.0f / ls;
#pragma unroll
          for (int k = 0; k < 16; ++k) pf[k] *= inv;
          accum_t<LDT, NCH>(tV, pf, of + RH * h, 1.0f, i, grp);    // O, this head's columns
        }
      }
      }
      GEMM_D(k0 + 3 * NQ, of, y)           // y = x + att Wo^T
      {
        float x[R];
        load_row<NCT>(x, xrow, grp);
#pragma unroll
        for (int k = 0; k < R; ++k) y[k] += x[k];
      }
    }
    float zn[R], xr[R];
    {
      float gm[R], bt[R], xh[R + 1];
      load_row<NCT>(gm, pb + PB_LN2G, grp);
      load_row<NCT>(bt, pb + PB_LN2B, grp);
      ln_row<NCT>(y, gm, bt, xh, zn);
    }
    f32x4 zacc[NCT];
    zero_acc<NCT>(zacc);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float am[R];
      GEMM_D(k0 + 4 * NQ + 2 * NQ * mt, zn, am)
      {
        float b1[R];
        load_row<NCT>(b1, pb + mt * D, grp);
#pragma unroll
        for (int k = 0; k < R; k += 2) {
          f32x2 g2, gp2;
          cora_gelu_both2((f32x2){am[k] + b1[k], am[k + 1] + b1[k + 1]}, g2, gp2);
          am[k] = g2.x; am[k + 1] = g2.y;
        }
      }
      if (mt == 3) load_row<NCT>(xr, xrow, grp);     // x again: its latency runs under the last four tiles
      GEMM_D_ACC(k0 + 5 * NQ + 2 * NQ * mt, am, zacc)
    }
    {
      float z[R], b2[R];
      unpack_at<NCT>(zacc, z, 0);
      load_row<NCT>(b2, pb + PB_B2, grp);
#pragma unroll
      for (int k = 0; k < R; ++k) z[k] += xr[k] + y[k] + b2[k];      // u = x + y + MLP
      store_row<NCT>(z, urow, grp);
    }
  }
}

template <int NCT, int H>
__global__ __launch_bounds__(256, 1) void pc_chain_wide2_kernel(FusedP p, const float* xin, float* xout, long long stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  wide2_chain_body<NCT, H>(p, lds, (int)blockIdx.x, xin, xout, stride);
}

template <int NCT, int H, bool LAZY>
__global__ __launch_bounds__(256, 2) void pc_block_wide2_kernel(FusedP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x >= p.L * p.B) {   // the last B workgroups do the two edge layers of one image each
    edge_path_wide<NCT, LAZY>(p, lds, (int)blockIdx.x - p.L * p.B);
    return;
  }
  // XCD-aware order (see fused_block.hip): XCD x gets a contiguous eighth of the (block, image) list
  int j = (int)blockIdx.x;
  const int n = p.L * p.B;
  if ((n & 7) == 0) j = (j & 7) * (n >> 3) + (j >> 3);
  wide2_block_body<NCT, H, LAZY>(p, lds, j / p.B, j % p.B);
}

size_t fused_wide2_stash_floats_per_row(int D, int H) { return (size_t)H * 64 + 3 * (size_t)D; }
// P == 48: the edge layers (edge_path_wide, fused_wide_common.h) are written for the 4x4x3 patch of the sweep configs.
// D = 128: H = 4 (heads of 32, one slice per head), H = 2 / 1 (heads of 64 / 128 as two / four 32-column slices);
// D = 96: H = 1 (three 32-column slices), H = 2 (heads of 48 as three 16-column slices), H = 4 (heads of 24: one and a
// half column groups, lane-masked 32-column slices).
bool fused_wide2_supported(int N, int P, int D, int H, int M) {
  return N == 64 && M == 4 * D && P == 48 && (D == 128 || D == 96) && (H == 4 || H == 2 || H == 1);
}

namespace {
template <int NCT, int H>
int launch_fused_wide2_h(const FusedP& p, int L, hipStream_t st) {
  static bool attr[64] = {};
  constexpr size_t sm = (size_t)W2Cfg<NCT, H>::LDS_FLOATS * sizeof(float);
  if (first_use_on_device(attr)) {
    if (hipFuncSetAttribute((const void*)pc_block_wide2_kernel<NCT, H, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void*)pc_block_wide2_kernel<NCT, H, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) return -1;
  }
  if (p.lazy) hipLaunchKernelGGL((pc_block_wide2_kernel<NCT, H, true>), dim3((L + 1) * p.B), dim3(256), sm, st, p);
  else hipLaunchKernelGGL((pc_block_wide2_kernel<NCT, H, false>), dim3((L + 1) * p.B), dim3(256), sm, st, p);
  return 0;
}
template <int NCT, int H>
int launch_chain_wide2_h(const FusedP& p, const float* xin, float* xout, long long stride, hipStream_t st) {
  static bool attr[64] = {};
  constexpr size_t sm = (size_t)W2Cfg<NCT, H>::LDS_FLOATS * sizeof(float);
  if (first_use_on_device(attr)) {
    if (hipFuncSetAttribute((const void*)pc_chain_wide2_kernel<NCT, H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) return -1;
  }
  hipLaunchKernelGGL((pc_chain_wide2_kernel<NCT, H>), dim3(p.B), dim3(256), sm, st, p, xin, xout, stride);
  return 0;
}
}  // namespace

int launch_fused_wide2(const FusedP& p, int L, int D, hipStream_t st) {
  switch (D * 8 + p.H) {
    case 128 * 8 + 4: return launch_fused_wide2_h<8, 4>(p, L, st);
    case 128 * 8 + 2: return launch_fused_wide2_h<8, 2>(p, L, st);
    case 128 * 8 + 1: return launch_fused_wide2_h<8, 1>(p, L, st);
    case 96 * 8 + 4: return launch_fused_wide2_h<6, 4>(p, L, st);
    case 96 * 8 + 2: return launch_fused_wide2_h<6, 2>(p, L, st);
    case 96 * 8 + 1: return launch_fused_wide2_h<6, 1>(p, L, st);
  }
  return -1;
}

// the L blocks of the status-"forward" chain for every image (see wide2_chain_body)
int launch_chain_wide2(const FusedP& p, int L, int D, const float* xin, float* xout, long long stride, hipStream_t st) {
  if (L < 1 || p.L != L) return -1;
  switch (D * 8 + p.H) {
    case 128 * 8 + 4: return launch_chain_wide2_h<8, 4>(p, xin, xout, stride, st);
    case 128 * 8 + 2: return launch_chain_wide2_h<8, 2>(p, xin, xout, stride, st);
    case 128 * 8 + 1: return launch_chain_wide2_h<8, 1>(p, xin, xout, stride, st);
    case 96 * 8 + 4: return launch_chain_wide2_h<6, 4>(p, xin, xout, stride, st);
    case 96 * 8 + 2: return launch_chain_wide2_h<6, 2>(p, xin, xout, stride, st);
    case 96 * 8 + 1: return launch_chain_wide2_h<6, 1>(p, xin, xout, stride, st);
  }
  return -1;
}
#undef STAGE
#undef GEMM_CT
#undef GEMM_D
#undef GEMM_D_ACC
#undef PREFETCH_ALL
#undef WRITE_ALL
#undef PF_ONE
#undef PB_ONE
