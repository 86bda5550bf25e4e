// layout_t.h -- "layout T" device helpers shared by the D = 64 fused kernels (fused_block.hip: one workgroup
// per (block, image); fused_rows.hip: one workgroup per 64-row tile of a block, any sequence length).
//
// Layout T (per wave = 16 token rows): lane l = (r = l & 15, grp = l >> 4) holds X[row r][16 ct + 4 grp + reg],
// ct < D/16, reg < 4 -- the B operand of v_mfma_f32_16x16x4_f32 for k-step (ct, reg) AND what the instruction
// returns when the weight is the A operand, so a chain of linear layers needs no shuffles and no LDS round trip.
#pragma once
#include "common.h"

namespace {

constexpr int LD = 68;            // LDS row stride (floats): 64 + 4, keeps b128 alignment, conflict-free
constexpr int TILE = 64 * LD;     // one 64x64 tile

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// acc[ot] += sum_k T[16 ot + i][k] * xf[k]  for the 16*NOT rows x 16*NCT columns of an LDS tile
// (weights, K, Q, V or dO rows; row stride LDT) -> acc[ot][reg'] = Y[row r][16 ot + 4 grp + reg']
template <int NOT, int NCT, int LDT>
__device__ __forceinline__ void gemm_tile_g(const float* __restrict__ t, const float* xf, f32x4* acc, int i, int grp) {
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    float4 a[NOT];
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) a[ot] = *reinterpret_cast<const float4*>(&t[(16 * ot + i) * LDT + 16 * ct + 4 * grp]);
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) acc[ot] = mfma16(a[ot].x, xf[4 * ct + 0], acc[ot]);
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) acc[ot] = mfma16(a[ot].y, xf[4 * ct + 1], acc[ot]);
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) acc[ot] = mfma16(a[ot].z, xf[4 * ct + 2], acc[ot]);
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) acc[ot] = mfma16(a[ot].w, xf[4 * ct + 3], acc[ot]);
  }
}
__device__ __forceinline__ void gemm_tile(const float* __restrict__ t, const float* xf, f32x4* acc, int i, int grp) {
  gemm_tile_g<4, 4, LD>(t, xf, acc, i, grp);
}

// acc[ct][reg'] += sum over the 64 tile rows j of T[j][16 ct + 4 grp + reg'] * pf[j -> (tt, reg)]
// where pf[4 tt + reg] belongs to tile row 16 tt + 4 grp + reg (the layout gemm_tile returns)
__device__ __forceinline__ void accum_tile(const float* __restrict__ t, const float* pf, f32x4* acc, int i, int grp) {
#pragma unroll
  for (int tt = 0; tt < 4; ++tt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const float* row = &t[(16 * tt + 4 * grp + reg) * LD + i];
      float b = pf[4 * tt + reg];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = mfma16(row[16 * ct], b, acc[ct]);
    }
}

// Head-restricted variants (H heads of width 64 / H = 16 NC columns): the contraction of gemm_tile_h
// and the output columns of accum_tile_h cover column groups ct0 .. ct0 + NC - 1 only.  xf / the
// caller's result stay indexed by the global column group, so a head fills its own slice of a row.
template <int NC>
__device__ __forceinline__ void gemm_tile_h(const float* __restrict__ t, const float* xf, f32x4* acc, int i, int grp, int ct0) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ct = ct0 + c;
    float4 a[4];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) a[ot] = *reinterpret_cast<const float4*>(&t[(16 * ot + i) * LD + 16 * ct + 4 * grp]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = mfma16(a[ot].x, xf[4 * ct + 0], acc[ot]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = mfma16(a[ot].y, xf[4 * ct + 1], acc[ot]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = mfma16(a[ot].z, xf[4 * ct + 2], acc[ot]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = mfma16(a[ot].w, xf[4 * ct + 3], acc[ot]);
  }
}
template <int NC>
__device__ __forceinline__ void accum_tile_h(const float* __restrict__ t, const float* pf, f32x4* acc, int i, int grp, int ct0) {
#pragma unroll
  for (int tt = 0; tt < 4; ++tt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const float* row = &t[(16 * tt + 4 * grp + reg) * LD + i + 16 * ct0];
      float b = pf[4 * tt + reg];
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = mfma16(row[16 * c], b, acc[c]);
    }
}

__device__ __forceinline__ float grp_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float grp_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

__device__ __forceinline__ void zero4(f32x4* a) {
#pragma unroll
  for (int k = 0; k < 4; ++k) a[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void unpack(const f32x4* a, float* f) {
#pragma unroll
  for (int k = 0; k < 4; ++k) { f[4 * k] = a[k][0]; f[4 * k + 1] = a[k][1]; f[4 * k + 2] = a[k][2]; f[4 * k + 3] = a[k][3]; }
}

// layout-T row I/O: 16 values of row r <-> 4 x float4 at columns 16 ct + 4 grp
__device__ __forceinline__ void load_rowT(float* f, const float* base, int ld, int grp) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    float4 v = *reinterpret_cast<const float4*>(base + 16 * ct + 4 * grp);
    f[4 * ct] = v.x; f[4 * ct + 1] = v.y; f[4 * ct + 2] = v.z; f[4 * ct + 3] = v.w;
  }
  (void)ld;
}
__device__ __forceinline__ void store_rowT(const float* f, float* base, int grp) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
    *reinterpret_cast<float4*>(base + 16 * ct + 4 * grp) = make_float4(f[4 * ct], f[4 * ct + 1], f[4 * ct + 2], f[4 * ct + 3]);
}

// Bias / LayerNorm-parameter gradients (dE/dtheta = -sum_rows ...): column sums over the 16 rows of
// a wave (lanes with equal grp).  The workgroups do not add into dW directly (800 waves per
// address: the float atomics doubled the STORE pass): each wave leaves its sums in its own slice of a scratch
// table, colpart[(g * B + b) * 4 + w][COLPART], reduced over the 4 B slices by colpart_reduce_kernel.
constexpr int CP_B2 = 0, CP_B1 = 64, CP_LN2B = 320, CP_LN2G = 384, CP_LN1B = 448, CP_LN1G = 512, COLPART = 576;
constexpr int CP_EDGE = 128;   // edge workgroups: [bp 64 | bout 48 | pad], table appended after the block slices
template <int NCT>
__device__ __forceinline__ void colsum_store(const float* f, float* dst, int i, int grp) {
#pragma unroll
  for (int k = 0; k < 4 * NCT; ++k) {
    float v = f[k];
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    if (i == 0) dst[16 * (k >> 2) + 4 * grp + (k & 3)] = v;
  }
}

// LayerNorm of a layout-T row (D = 64): xh = (x - mean) * rstd (kept), out = xh * gamma + beta
__device__ __forceinline__ void ln_rowT(const float* x, const float* gam, const float* bet, float* xh, float* out) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += x[k];
  const float mean = grp_sum(s) * (1.0f / 64.0f);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) { float d = x[k] - mean; q += d * d; }
  const float rstd = rsqrtf(fmaxf(grp_sum(q) * (1.0f / 64.0f), 0.f) + 1e-5f);
#pragma unroll
  for (int k = 0; k < 16; ++k) { xh[k] = (x[k] - mean) * rstd; out[k] = xh[k] * gam[k] + bet[k]; }
  xh[16] = rstd;
}
// dx = rstd * (dxh - mean(dxh) - xh * mean(dxh * xh)),  dxh = dout * gamma
__device__ __forceinline__ void ln_bwd_rowT(const float* dout, const float* gam, const float* xh, float* dx) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) { float d = dout[k] * gam[k]; dx[k] = d; s1 += d; s2 += d * xh[k]; }
  const float m1 = grp_sum(s1) * (1.0f / 64.0f), m2 = grp_sum(s2) * (1.0f / 64.0f);
  const float rstd = xh[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) dx[k] = rstd * (dx[k] - m1 - xh[k] * m2);
}

// ---- rows of other widths (the 48-wide patch vectors of the edge layers) and whole small matrices into LDS ----
__device__ __forceinline__ void copy_to_lds(float* dst, const float* src, int rows, int cols, int ldt, int tid) {
  const int c4n = cols / 4;
  for (int f = tid; f < rows * c4n; f += 256) {
    int rr = f / c4n, c4 = f % c4n;
    *reinterpret_cast<float4*>(&dst[rr * ldt + 4 * c4]) = *reinterpret_cast<const float4*>(src + (long long)rr * cols + 4 * c4);
  }
}
template <int NCT>
__device__ __forceinline__ void load_rowT_n(float* f, const float* base, int grp) {
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    float4 v = *reinterpret_cast<const float4*>(base + 16 * ct + 4 * grp);
    f[4 * ct] = v.x; f[4 * ct + 1] = v.y; f[4 * ct + 2] = v.z; f[4 * ct + 3] = v.w;
  }
}
template <int NCT>
__device__ __forceinline__ void store_rowT_n(const float* f, float* base, int grp) {
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
    *reinterpret_cast<float4*>(base + 16 * ct + 4 * grp) = make_float4(f[4 * ct], f[4 * ct + 1], f[4 * ct + 2], f[4 * ct + 3]);
}

}  // namespace
