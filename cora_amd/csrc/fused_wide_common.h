// fused_wide_common.h -- device helpers shared by the wide fused kernels (fused_wide.hip: one workgroup per
// CU, any head width; fused_wide2.hip: two workgroups per CU, heads of 32): layout-T row I/O, LayerNorm
// rows, MFMA tile products, opaque LDS pointers.  gfx950 only.
#pragma once
#include "common.h"
#include "vode_mult.h"

namespace {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
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

// LDS spans 141 KB here but ds_read / ds_write immediates reach 64 KB: an address the compiler can see as
// (lane offset + large constant) gets a register of its own per access, hoisted and then spilled by the
// hundred.  Passing each buffer's per-lane base through an empty asm makes it opaque: one base register
// per buffer and use, all tile-internal offsets as immediates.
// The laundered pointer keeps its LDS address space (a generic pointer here would turn every read into flat_load).
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
__device__ __forceinline__ const lds_float* opaque(const float* q) {
  const lds_float* l = (const lds_float*)q;
  asm volatile("" : "+v"(l));
  return l;
}
__device__ __forceinline__ lds_float* opaque_w(float* q) {
  lds_float* l = (lds_float*)q;
  asm volatile("" : "+v"(l));
  return l;
}
__device__ __forceinline__ float4 lds_ld4(const lds_float* q) {
  const f32x4 v = *(const lds_f32x4*)q;
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void lds_st4(lds_float* q, float4 v) {
  *(lds_f32x4*)q = (f32x4){v.x, v.y, v.z, v.w};
}

template <int NCT>
struct WideCfg {
  static constexpr int D = 16 * NCT;        // hidden size
  static constexpr int M = 4 * D;           // MLP hidden
  static constexpr int R = 4 * NCT;         // registers per lane of a layout-T row
  static constexpr int NOT = NCT / 2;       // 16-row output groups of a weight tile
  static constexpr int TR = 16 * NOT;       // weight-tile rows (outputs) = D / 2
  static constexpr int LD = D + 4;          // LDS row stride: b128-aligned, conflict-free in 16-lane groups
  static constexpr int SLOT = 64 * LD;      // one LDS slot: 64 rows (K, V, Q, dO; a weight tile uses TR <= 64 of them)
  static constexpr int NF4 = TR * D / 4;    // float4s of a weight tile
  static constexpr int NST = (NF4 + 255) / 256;   // float4s staged per thread
  static constexpr int PB = M + 4 * D;      // parameter block: b1 | ln2 gamma | ln2 beta | b2 | ln1 gamma
  static constexpr int LDS_FLOATS = 4 * SLOT + 512 + PB;
};

// acc[ot] += sum_k T[16 ot + i][k] xf[k] over the NOT x NCT groups of an LDS tile (rows = outputs)
// The LDS operand reads run ONE column group ahead of the MFMAs that consume them and a scheduling
// barrier per group keeps the compiler from hoisting more (left alone it lifts all 4 NOT NCT reads of a
// tile to the top of the stage -- 128 registers at D = 128 -- and the kernel spills).
template <int NOT, int NCT, int LD>
__device__ __forceinline__ void gemm_tile_w(const float* __restrict__ t, const float* xf, f32x4* acc, int i, int grp) {
  float4 a[2][NOT];
  const lds_float* tl = opaque(t + i * LD + 4 * grp);
#pragma unroll
  for (int ot = 0; ot < NOT; ++ot) a[0][ot] = lds_ld4(&tl[16 * ot * LD]);
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    if (ct + 1 < NCT) {
#pragma unroll
      for (int ot = 0; ot < NOT; ++ot) a[(ct + 1) & 1][ot] = lds_ld4(&tl[16 * ot * LD + 16 * (ct + 1)]);
    }
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) acc[ot] = mfma16(a[ct & 1][ot].x, xf[4 * ct + 0], acc[ot]);
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) acc[ot] = mfma16(a[ct & 1][ot].y, xf[4 * ct + 1], acc[ot]);
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) acc[ot] = mfma16(a[ct & 1][ot].z, xf[4 * ct + 2], acc[ot]);
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) acc[ot] = mfma16(a[ct & 1][ot].w, xf[4 * ct + 3], acc[ot]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Head-restricted contraction over the 64 rows of an LDS slot (K, V, Q or dO rows):
//   acc[ot][reg'] = sum over the head's columns c of T[16 ot + i][c] * xf[c]   (ot < 4: 64 rows)
// The head is the quad range [Q0, Q1) (quad = 4 columns); lanes whose quad of a shared column group
// belongs to another head contribute zeros.
template <int LD, int Q0, int Q1>
__device__ __forceinline__ void gemm_head(const float* __restrict__ t, const float* xf, f32x4* acc, int i, int grp) {
  constexpr int CT0 = Q0 / 4, CT1 = (Q1 + 3) / 4;
  const lds_float* tl = opaque(t + i * LD + 4 * grp);
  float4 a[2][4];
#pragma unroll
  for (int ot = 0; ot < 4; ++ot) a[CT0 & 1][ot] = lds_ld4(&tl[16 * ot * LD + 16 * CT0]);
#pragma unroll
  for (int ct = CT0; ct < CT1; ++ct) {
    const bool whole = (4 * ct >= Q0) && (4 * ct + 4 <= Q1);       // compile-time after unrolling
    const bool mine = whole || ((4 * ct + grp >= Q0) && (4 * ct + grp < Q1));
    if (ct + 1 < CT1) {
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) a[(ct + 1) & 1][ot] = lds_ld4(&tl[16 * ot * LD + 16 * (ct + 1)]);
    }
    const float b0 = mine ? xf[4 * ct + 0] : 0.f, b1 = mine ? xf[4 * ct + 1] : 0.f;
    const float b2 = mine ? xf[4 * ct + 2] : 0.f, b3 = mine ? xf[4 * ct + 3] : 0.f;
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = mfma16(a[ct & 1][ot].x, b0, acc[ot]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = mfma16(a[ct & 1][ot].y, b1, acc[ot]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = mfma16(a[ct & 1][ot].z, b2, acc[ot]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = mfma16(a[ct & 1][ot].w, b3, acc[ot]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// out[4 ct + reg'] (only the quads of the head) = sum over the 64 slot rows j of T[j][col] * pf[j],
// pf[4 tt + reg] belonging to slot row 16 tt + 4 grp + reg (the layout gemm_head returns)
template <int LD, int Q0, int Q1>
__device__ __forceinline__ void accum_head(const float* __restrict__ t, const float* pf, float* out, float scale, int i, int grp) {
  constexpr int CT0 = Q0 / 4, CT1 = (Q1 + 3) / 4, NC = CT1 - CT0;
  f32x4 acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const lds_float* tl = opaque(t + 4 * grp * LD + i);
#pragma unroll
  for (int tt = 0; tt < 4; ++tt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const lds_float* row = &tl[(16 * tt + reg) * LD + 16 * CT0];
      const float b = pf[4 * tt + reg];
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = mfma16(row[16 * c], b, acc[c]);
      if (reg == 3) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ct = CT0 + c;
    const bool whole = (4 * ct >= Q0) && (4 * ct + 4 <= Q1);
    const bool mine = whole || ((4 * ct + grp >= Q0) && (4 * ct + grp < Q1));
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) if (mine) out[4 * ct + reg] = acc[c][reg] * scale;
  }
}

template <int NG>
__device__ __forceinline__ void zero_acc(f32x4* a) {
#pragma unroll
  for (int k = 0; k < NG; ++k) a[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
}
// accumulator groups -> row registers, starting at column group ct0
template <int NG>
__device__ __forceinline__ void unpack_at(const f32x4* a, float* f, int ct0) {
#pragma unroll
  for (int k = 0; k < NG; ++k) {
    f[4 * (ct0 + k)] = a[k][0]; f[4 * (ct0 + k) + 1] = a[k][1]; f[4 * (ct0 + k) + 2] = a[k][2]; f[4 * (ct0 + k) + 3] = a[k][3];
  }
}
template <int NCT>
__device__ __forceinline__ void load_row(float* f, const float* base, int grp) {
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    float4 v = *reinterpret_cast<const float4*>(base + 16 * ct + 4 * grp);
    f[4 * ct] = v.x; f[4 * ct + 1] = v.y; f[4 * ct + 2] = v.z; f[4 * ct + 3] = v.w;
  }
}
template <int NCT>
__device__ __forceinline__ void store_row(const float* f, float* base, int grp) {
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
    *reinterpret_cast<float4*>(base + 16 * ct + 4 * grp) = make_float4(f[4 * ct], f[4 * ct + 1], f[4 * ct + 2], f[4 * ct + 3]);
}

// LayerNorm of a layout-T row: xh = (x - mean) * rstd (kept, rstd in xh[R]), out = xh * gamma + beta
template <int NCT>
__device__ __forceinline__ void ln_row(const float* x, const float* gam, const float* bet, float* xh, float* out) {
  constexpr int R = 4 * NCT;
  constexpr float invD = 1.0f / (16.0f * NCT);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < R; ++k) s += x[k];
  const float mean = grp_sum(s) * invD;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < R; ++k) { float d = x[k] - mean; q += d * d; }
  const float rstd = rsqrtf(fmaxf(grp_sum(q) * invD, 0.f) + 1e-5f);
#pragma unroll
  for (int k = 0; k < R; ++k) { xh[k] = (x[k] - mean) * rstd; out[k] = xh[k] * gam[k] + (bet ? bet[k] : 0.f); }
  xh[R] = rstd;
}
// dx = rstd * (dxh - mean(dxh) - xh * mean(dxh * xh)),  dxh = dout * gamma
template <int NCT>
__device__ __forceinline__ void ln_bwd_row(const float* dout, const float* gam, const float* xh, float* dx) {
  constexpr int R = 4 * NCT;
  constexpr float invD = 1.0f / (16.0f * NCT);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < R; ++k) { float d = dout[k] * gam[k]; dx[k] = d; s1 += d; s2 += d * xh[k]; }
  const float m1 = grp_sum(s1) * invD, m2 = grp_sum(s2) * invD;
  const float rstd = xh[R];
#pragma unroll
  for (int k = 0; k < R; ++k) dx[k] = rstd * (dx[k] - m1 - xh[k] * m2);
}

// ---- attention pieces of one head (Q0..Q1 = its quad range), layout as in fused_block.hip ---------
// forward: S[key][q] -> softmax over keys -> O (this head's columns of `of`), P to the stash, LSE to rowc
template <int LD, int Q0, int Q1>
__device__ __forceinline__ void head_forward(const float* bufK, const float* bufV, const float* qf, float* of, float* st_p_h,
                                             float* lse_h, float sc, int i, int grp, int w) {
  f32x4 sa[4];
  zero_acc<4>(sa);
  gemm_head<LD, Q0, Q1>(bufK, qf, sa, i, grp);
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
  if (grp == 0) lse_h[16 * w + i] = mx + __builtin_amdgcn_logf(ls);
  store_row<4>(pf, st_p_h, grp);
  __builtin_amdgcn_sched_barrier(0);
  accum_head<LD, Q0, Q1>(bufV, pf, of, 1.0f, i, grp);
  __builtin_amdgcn_sched_barrier(0);
}
// backward pass A (own queries): dP, delta, dS, dq (this head's columns of `dq`); P / dS returned in pf / dp
template <int LD, int Q0, int Q1>
__device__ __forceinline__ void head_backward_a(const float* bufK, const float* bufV, const float* dof, float* dq, const float* st_p_h,
                                                float* delta_h, float* pf, float* dp, float qscale, int i, int grp, int w) {
  load_row<4>(pf, st_p_h, grp);
  f32x4 sa[4];
  zero_acc<4>(sa);
  gemm_head<LD, Q0, Q1>(bufV, dof, sa, i, grp);                 // dP[key][q] = V[key] . dO[q]
  unpack_at<4>(sa, dp, 0);
  float dl = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) dl += pf[k] * dp[k];
  dl = grp_sum(dl);                                              // delta = sum_key P dP
  if (grp == 0) delta_h[16 * w + i] = dl;
#pragma unroll
  for (int k = 0; k < 16; ++k) dp[k] = pf[k] * (dp[k] - dl);    // dS
  __builtin_amdgcn_sched_barrier(0);
  accum_head<LD, Q0, Q1>(bufK, dp, dq, qscale, i, grp);
  __builtin_amdgcn_sched_barrier(0);
}
// backward pass B (own keys), several heads: S and dP recomputed with the key on the lane; dv, dk
template <int LD, int Q0, int Q1>
__device__ __forceinline__ void head_backward_b(const float* bufQ, const float* bufDO, const float* kf, const float* vf, float* dk, float* dv,
                                                const float* lse_h, const float* delta_h, float sc, float qscale, int i, int grp) {
  float sp[16], dp[16];
  f32x4 sa[4];
  zero_acc<4>(sa); gemm_head<LD, Q0, Q1>(bufQ, kf, sa, i, grp); unpack_at<4>(sa, sp, 0);     // S[q][key]
  zero_acc<4>(sa); gemm_head<LD, Q0, Q1>(bufDO, vf, sa, i, grp); unpack_at<4>(sa, dp, 0);    // dP[q][key]
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    float4 l4 = *reinterpret_cast<const float4*>(&lse_h[16 * tt + 4 * grp]);
    float4 d4 = *reinterpret_cast<const float4*>(&delta_h[16 * tt + 4 * grp]);
    float pv;
    pv = __builtin_amdgcn_exp2f(sp[4 * tt + 0] * sc - l4.x); sp[4 * tt + 0] = pv; dp[4 * tt + 0] = pv * (dp[4 * tt + 0] - d4.x);
    pv = __builtin_amdgcn_exp2f(sp[4 * tt + 1] * sc - l4.y); sp[4 * tt + 1] = pv; dp[4 * tt + 1] = pv * (dp[4 * tt + 1] - d4.y);
    pv = __builtin_amdgcn_exp2f(sp[4 * tt + 2] * sc - l4.z); sp[4 * tt + 2] = pv; dp[4 * tt + 2] = pv * (dp[4 * tt + 2] - d4.z);
    pv = __builtin_amdgcn_exp2f(sp[4 * tt + 3] * sc - l4.w); sp[4 * tt + 3] = pv; dp[4 * tt + 3] = pv * (dp[4 * tt + 3] - d4.w);
  }
  __builtin_amdgcn_sched_barrier(0);
  accum_head<LD, Q0, Q1>(bufDO, sp, dv, 1.0f, i, grp);       // dV = P^T dO
  accum_head<LD, Q0, Q1>(bufQ, dp, dk, qscale, i, grp);      // dK = dS^T Q
  __builtin_amdgcn_sched_barrier(0);
}


// ---- edge layers of one image (one extra workgroup per image in the wide kernels' grids) ------------
//   e_1 = h_1 - (h_0 Wp^T + bp + pos)          g_0 = -s_0 e_1 Wp              (vode 0 has no own energy)
//   e_img = x_patch - (h_{L+1} Wout^T + bout)   bp_{L+1} = e_img Wout          (finished by vode_grad_kernel)
// (/root/reference/src/decoder_transformer.py:222-228, 244-256, 143-153).  P = 48 (patch 4 x 4 x 3).
// The two halves run one after the other through the same LDS (Wp + Wp^T, then Wout + Wout^T: 52 KB).
__device__ __forceinline__ void edge_copy_to_lds(float* dst, const float* src, int rows, int cols, int ldt, int tid) {
  const int c4n = cols / 4;
  for (int f = tid; f < rows * c4n; f += 256) {
    const int rr = f / c4n, c4 = f % c4n;
    *reinterpret_cast<float4*>(&dst[rr * ldt + 4 * c4]) = *reinterpret_cast<const float4*>(src + (long long)rr * cols + 4 * c4);
  }
}
template <int NCT, bool LAZY = false>
__device__ __forceinline__ void edge_path_wide(const FusedP& p, float* lds, int b) {
  constexpr int D = 16 * NCT, P = 48, LP = P + 4, LDD = D + 4, R = 4 * NCT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, grp = lane >> 4;
  float* sA = lds;                   // [D][LP]    Wp   [D][P]   /  Wout^T [D][P]
  float* sB = lds + D * LP;          // [P][LDD]   Wp^T [P][D]   /  Wout   [P][D]
  const long long tok = (long long)b * 64 + 16 * w + i;     // token row of this lane
  const long long nl = (long long)p.B * 64 * D;
  edge_copy_to_lds(sA, p.W + p.oWp, D, P, LP, tid);
  edge_copy_to_lds(sB, p.WT + p.tWp, P, D, LDD, tid);
  float e1[R];
  float c0 = 0.f, c1 = 0.f, cl = 0.f;     // LAZY (FusedP::lazy): multipliers of the update this launch applies to vodes 0, 1, L+1
  if (LAZY) vode_multipliers_wave(p.up, 0, 1, p.L + 1, c0, c1, cl);
  {
    float h0[12];
    load_row<3>(h0, p.H0 + tok * P, grp);
    load_row<NCT>(e1, p.H1 + tok * D, grp);                  // h_1
    if (LAZY) {   // vode 0 is this workgroup's input (it writes the new value and momentum), vode 1 only its target
      const float eta = p.up.eta[p.up.t], mu = p.up.mu;
      float g0[12], m0[12];
      load_row<3>(g0, p.G0o + tok * P, grp); load_row<3>(m0, p.M0o + tok * P, grp);
#pragma unroll
      for (int k = 0; k < 12; ++k) { m0[k] = g0[k] * c0 + mu * m0[k]; h0[k] -= eta * m0[k]; }
      store_row<3>(h0, p.H0n + tok * P, grp); store_row<3>(m0, p.M0n + tok * P, grp);
      float g1[R], m1[R];
      load_row<NCT>(g1, p.Go1 + tok * D, grp); load_row<NCT>(m1, p.Mo1 + tok * D, grp);
#pragma unroll
      for (int k = 0; k < R; ++k) e1[k] -= eta * (g1[k] * c1 + mu * m1[k]);
    }
    __syncthreads();
    f32x4 acc[NCT];
    zero_acc<NCT>(acc);
    gemm_tile_w<NCT, 3, LP>(sA, h0, acc, i, grp);
    float u[R], bp[R], ps[R];
    unpack_at<NCT>(acc, u, 0);
    load_row<NCT>(bp, p.W + p.obp, grp);
    load_row<NCT>(ps, p.pos + (long long)(16 * w + i) * D, grp);
#pragma unroll
    for (int k = 0; k < R; ++k) e1[k] = e1[k] - (u[k] + bp[k] + ps[k]);
    store_row<NCT>(e1, p.E1 + tok * D, grp);
  }
  {
    f32x4 acc[3];
    zero_acc<3>(acc);
    gemm_tile_w<3, NCT, LDD>(sB, e1, acc, i, grp);
    float g0[12];
    double ss = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float v = -p.s0 * acc[k][j]; g0[4 * k + j] = v; ss += (double)v * v; }
    store_row<3>(g0, p.G0 + tok * P, grp);
    ss = wave_sum_d(ss);
    if (lane == 0) atomicAdd(&p.sumsq0[(b * 4 + w) & 7], ss);
  }
  __syncthreads();                   // both tiles are no longer read
  edge_copy_to_lds(sB, p.W + p.oWout, P, D, LDD, tid);
  edge_copy_to_lds(sA, p.WT + p.tWout, D, P, LP, tid);
  {
    float hl[R], xp[12], bo[12], ei[12];
    load_row<NCT>(hl, p.H1 + (long long)p.L * nl + tok * D, grp);
    load_row<3>(xp, p.XP + tok * P, grp);
    load_row<3>(bo, p.W + p.obout, grp);
    if (LAZY) {   // vode L+1: input of the output projection, so written here
      const float eta = p.up.eta[p.up.t], mu = p.up.mu;
      const long long oL = (long long)p.L * nl + tok * D;
      float gl[R], ml[R];
      load_row<NCT>(gl, p.Go1 + oL, grp); load_row<NCT>(ml, p.Mo1 + oL, grp);
#pragma unroll
      for (int k = 0; k < R; ++k) { ml[k] = gl[k] * cl + mu * ml[k]; hl[k] -= eta * ml[k]; }
      store_row<NCT>(hl, p.Hn1 + oL, grp); store_row<NCT>(ml, p.Mn1 + oL, grp);
    }
    __syncthreads();
    f32x4 acc[3];
    zero_acc<3>(acc);
    gemm_tile_w<3, NCT, LDD>(sB, hl, acc, i, grp);
    float en = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float e = xp[4 * k + j] - (acc[k][j] + bo[4 * k + j]); ei[4 * k + j] = e; en += 0.5f * e * e; }
    store_row<3>(ei, p.EIMG + tok * P, grp);
    const double tot = wave_sum_d((double)en);
    if (lane == 0) atomicAdd(&p.esens[(b * 4 + w) & 7], tot);          // vode L+2 (sensory): plain se_energy
    f32x4 acc2[NCT];
    zero_acc<NCT>(acc2);
    gemm_tile_w<NCT, 3, LP>(sA, ei, acc2, i, grp);
    float bpl[R];
    unpack_at<NCT>(acc2, bpl, 0);
    store_row<NCT>(bpl, p.BP1 + (long long)p.L * nl + tok * D, grp);
  }
}

}  // namespace
