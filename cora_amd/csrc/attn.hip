// attn.hip -- fp32 self-attention forward / backward for the PC-ViT block on
// v_mfma_f32_16x16x4_f32 (gfx950).  Replaces eqx.nn.MultiheadAttention as wrapped by
// pxnn.MultiHeadAttention (/root/reference/pcx/nn/_layer.py:359-402): no biases, no mask,
// no dropout, logits = (q/sqrt(d)) k^T, softmax over keys, per-head column blocks.
//
// One workgroup (4 waves) per (image, head, 64 "own" rows); each wave owns 16 rows held as MFMA
// B-operand fragments in registers, and the workgroup streams the other side through LDS in
// 64-row chunks (one chunk = the whole sequence at N = 64, so K/V are LDS-resident there).
//   forward : own = queries, stream keys/values, online softmax, writes O and base-2 LSE
//   bwd-q   : own = queries (+dO), stream K,V : S, dP recomputed, dQ accumulated; also delta=rowsum(dO*O)
//   bwd-kv  : own = keys (+V),   stream Q,dO : S, dP recomputed, dK, dV accumulated
// Nothing is summed across workgroups, so the results are deterministic (no atomics).
//
// MFMA operand maps used throughout (lane l, c = l & 15, grp = l >> 4):
//   score   tile  D[i][j] = sum_dd X[i][dd] Y[j][dd] : A = X[i=c][4s+grp], B = Y[j=c][4s+grp];
//                 lane holds D[i = 4grp+reg][j = c]  (the own row j stays on the lane)
//   accum   tile  D[cc][j] += sum_i Z[i][cc] W[i][j] : A = Z[i = 4grp+reg][cc = c], B = score regs
#include "common.h"
#include "attn_map.h"

#define CH 64  // streamed rows per chunk
// 16-column output tiles per head; a head dim that is not a multiple of 16 (24 = 96 / 4 of the
// 12-block sweep grid) computes a padded last tile -- MFMA output columns are independent, so the
// extra columns read neighbouring LDS words and are simply never stored
#define NT16(dh) (((dh) + 15) / 16)

template <int DH>
struct AttnSmem {
  static constexpr int LD = DH + 4;
  float s1[CH * LD];
  float s2[CH * LD];
  float rowc[2 * CH];  // per streamed row constants (lse2, delta) for bwd-kv
};
// forward and bwd-q carry no row constants: at head dim 48 the 512 bytes decide between five and six
// workgroups per CU (6 x 26 624 B = 159 744 B of the 160 KB).  The kernels' launch bounds ask for 6 / 5 / 4
// waves per SIMD (forward / bwd-q / bwd-kv) at head dims <= 48, where that costs no spills; these kernels
// hide their staging latency with co-resident workgroups, not inside one
template <int DH>
struct AttnSmem2 {
  static constexpr int LD = DH + 4;
  float s1[CH * LD];
  float s2[CH * LD];
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// stage rows [r0, r0+CH) x [0,DH) of a [*, ld] matrix (column offset already applied) into LDS
template <int DH, int NTH = 256>
__device__ __forceinline__ void stage_rows(float* dst, const float* src, int ld, int r0, int nrows) {
  constexpr int LD = DH + 4;
  constexpr int V = DH / 4;  // float4 per row
  for (int f = threadIdx.x; f < CH * V; f += NTH) {
    int row = f / V, c4 = f % V;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + row < nrows) v = *reinterpret_cast<const float4*>(src + (long long)(r0 + row) * ld + 4 * c4);
    *reinterpret_cast<float4*>(&dst[row * LD + 4 * c4]) = v;
  }
}

// the same staging in two halves (registers in between) for the 8-wave workgroups: the next chunk's global
// loads are in flight while the workgroup computes on the current one; 512 threads make it 3 float4 per
// thread and matrix pair at head dim 48, which the 4-wave form could not afford (it lost occupancy)
template <int DH, int NTH>
struct StageF {
  static constexpr int LD = DH + 4, V = DH / 4, NF = CH * V, NI = (2 * NF + NTH - 1) / NTH;
  float4 r[NI];
  __device__ __forceinline__ void load(const float* src0, int ld0, const float* src1, int ld1, int r0, int nrows) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int f2 = it * NTH + (int)threadIdx.x;
      r[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f2 >= 2 * NF) continue;
      const int m = f2 >= NF, f = m ? f2 - NF : f2;
      const int row = r0 + f / V, c4 = f % V;
      const float* src = m ? src1 : src0;
      const int ld = m ? ld1 : ld0;
      if (row < nrows) r[it] = *reinterpret_cast<const float4*>(src + (long long)row * ld + 4 * c4);
    }
  }
  __device__ __forceinline__ void store(float* dst0, float* dst1) const {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int f2 = it * NTH + (int)threadIdx.x;
      if (f2 >= 2 * NF) continue;
      const int m = f2 >= NF, f = m ? f2 - NF : f2;
      float* dst = m ? dst1 : dst0;
      *reinterpret_cast<float4*>(&dst[(f / V) * LD + 4 * (f % V)]) = r[it];
    }
  }
};

// own-row fragment.  Which k an MFMA step contracts is free as long as the A lane and the B lane agree,
// so for head dims that are a multiple of 16 step (j, t) takes k = 16 j + 4 grp + t: a lane's fragment is
// DH/16 contiguous float4 (f[4 j + t] = X[row][16 j + 4 grp + t]) and the streamed side reads one
// ds_read_b128 per four MFMAs.  Other head dims (24) keep k = 4 s + grp (f[s] = X[row][4 s + grp]).
template <int DH>
__device__ __forceinline__ void load_frag(float* f, const float* src, int ld, int row, int nrows, int grp) {
  if constexpr (DH % 16 == 0) {
#pragma unroll
    for (int j = 0; j < DH / 16; ++j) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < nrows) v = *reinterpret_cast<const float4*>(src + (long long)row * ld + 16 * j + 4 * grp);
      f[4 * j] = v.x; f[4 * j + 1] = v.y; f[4 * j + 2] = v.z; f[4 * j + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int s = 0; s < DH / 4; ++s) f[s] = (row < nrows) ? src[(long long)row * ld + 4 * s + grp] : 0.f;
  }
}
// the same fragment from an LDS tile (row stride DH + 4)
template <int DH>
__device__ __forceinline__ void lds_frag(float* f, const float* tile, int row, int grp) {
  constexpr int LD = DH + 4;
  if constexpr (DH % 16 == 0) {
#pragma unroll
    for (int j = 0; j < DH / 16; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(&tile[row * LD + 16 * j + 4 * grp]);
      f[4 * j] = v.x; f[4 * j + 1] = v.y; f[4 * j + 2] = v.z; f[4 * j + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int s = 0; s < DH / 4; ++s) f[s] = tile[row * LD + 4 * s + grp];
  }
}

// S[tt] (tt = 0..NVT-1) = streamed rows 16tt..16tt+15 (from LDS) x own fragment.  NVT = 16-row tiles of
// the chunk that hold a valid row: the ragged last chunk of a sequence (N = 196: 4 of 64 rows) skips
// the MFMAs of its empty tiles instead of multiplying padding.
template <int DH, int NVT = 4>
__device__ __forceinline__ void score_tiles(f32x4* S, const float* lds, const float* frag, int c, int grp) {
  constexpr int LD = DH + 4;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) S[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (DH % 16 == 0) {
#pragma unroll
    for (int j = 0; j < DH / 16; ++j) {
      float4 a[NVT];
#pragma unroll
      for (int tt = 0; tt < NVT; ++tt) a[tt] = *reinterpret_cast<const float4*>(&lds[(16 * tt + c) * LD + 16 * j + 4 * grp]);
#pragma unroll
      for (int tt = 0; tt < NVT; ++tt) S[tt] = mfma16(a[tt].x, frag[4 * j + 0], S[tt]);
#pragma unroll
      for (int tt = 0; tt < NVT; ++tt) S[tt] = mfma16(a[tt].y, frag[4 * j + 1], S[tt]);
#pragma unroll
      for (int tt = 0; tt < NVT; ++tt) S[tt] = mfma16(a[tt].z, frag[4 * j + 2], S[tt]);
#pragma unroll
      for (int tt = 0; tt < NVT; ++tt) S[tt] = mfma16(a[tt].w, frag[4 * j + 3], S[tt]);
    }
  } else {
#pragma unroll
    for (int s = 0; s < DH / 4; ++s) {
      float b = frag[s];
#pragma unroll
      for (int tt = 0; tt < NVT; ++tt) S[tt] = mfma16(lds[(16 * tt + c) * LD + 4 * s + grp], b, S[tt]);
    }
  }
}

// acc[ct] (ct = 0..DH/16-1) += sum over the 64 streamed rows of Z[row][16ct + c] * Pv[row][own]
template <int DH, int NVT = 4>
__device__ __forceinline__ void accum_tiles(f32x4* acc, const float* lds, const f32x4* Pv, int c, int grp) {
  constexpr int LD = DH + 4;
#pragma unroll
  for (int tt = 0; tt < NVT; ++tt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      float b = Pv[tt][reg];
      const float* zr = &lds[(16 * tt + 4 * grp + reg) * LD + c];
#pragma unroll
      for (int ct = 0; ct < NT16(DH); ++ct) acc[ct] = mfma16(zr[16 * ct], b, acc[ct]);
    }
}

// ---------------------------------------------------------------------------------------------
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, DH <= 48 ? (NW == 8 ? 5 : 6) : 1) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ O,
                                                       float* __restrict__ lse2, int nimg, int N, int H, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  AttnSmem2<DH>& sm = *reinterpret_cast<AttnSmem2<DH>*>(smem_raw);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, grp = lane >> 4;
  int rb, h, img;
  if (!attn_coords((N + 16 * NW - 1) / (16 * NW), H, nimg, rb, h, img)) return;
  const int ld = 3 * D;
  const float* base = qkv + (long long)img * N * ld;
  const float* Q = base + h * DH;
  const float* K = base + D + h * DH;
  const float* V = base + 2 * D + h * DH;
  const int qrow = rb * 16 * NW + 16 * w + c;
  const float sc = rsqrtf((float)DH) * CORA_LOG2E;  // base-2 logits

  float qf[DH / 4];
  load_frag<DH>(qf, Q, ld, qrow, N, grp);
  f32x4 oacc[NT16(DH)];
#pragma unroll
  for (int ct = 0; ct < NT16(DH); ++ct) oacc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, lsum = 0.f;

  // a wave whose 16 query rows all lie past the sequence (the last row block of N = 196 holds 4 rows)
  // only helps staging; empty key tiles of the last chunk are skipped (for_valid_tiles)
  const bool wave_active = rb * 16 * NW + 16 * w < N;
  constexpr bool PRE = (NW == 8);
  StageF<DH, 64 * NW> stg;
  if (PRE) stg.load(K, ld, V, ld, 0, N);
  for (int k0 = 0; k0 < N; k0 += CH) {
    __syncthreads();
    if (PRE) stg.store(sm.s1, sm.s2);
    else {
      stage_rows<DH, 64 * NW>(sm.s1, K, ld, k0, N);
      stage_rows<DH, 64 * NW>(sm.s2, V, ld, k0, N);
    }
    __syncthreads();
    if (PRE && k0 + CH < N) stg.load(K, ld, V, ld, k0 + CH, N);
    if (!wave_active) continue;
    for_valid_tiles((N - k0 + 15) >> 4, [&](auto nv) {
      constexpr int NVT = decltype(nv)::value;
      f32x4 S[4];
      score_tiles<DH, NVT>(S, sm.s1, qf, c, grp);
      float mx = -INFINITY;
#pragma unroll
      for (int tt = 0; tt < NVT; ++tt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          int key = k0 + 16 * tt + 4 * grp + reg;
          float sv = (key < N) ? S[tt][reg] * sc : -INFINITY;
          S[tt][reg] = sv;
          mx = fmaxf(mx, sv);
        }
      mx = grp_max(mx);
      float mnew = fmaxf(m, mx);               // finite: every chunk holds at least one valid key
      float alpha = __builtin_amdgcn_exp2f(m - mnew);
      float ps = 0.f;
#pragma unroll
      for (int tt = 0; tt < NVT; ++tt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          float pv = __builtin_amdgcn_exp2f(S[tt][reg] - mnew);
          S[tt][reg] = pv;
          ps += pv;
        }
      ps = grp_sum(ps);
      lsum = lsum * alpha + ps;
      m = mnew;
#pragma unroll
      for (int ct = 0; ct < NT16(DH); ++ct) oacc[ct] *= alpha;
      accum_tiles<DH, NVT>(oacc, sm.s2, S, c, grp);
    });
  }
  if (qrow < N) {
    float inv = 1.0f / lsum;
    float* orow = O + ((long long)img * N + qrow) * D + h * DH;
#pragma unroll
    for (int ct = 0; ct < NT16(DH); ++ct) {
      float4 v = make_float4(oacc[ct][0] * inv, oacc[ct][1] * inv, oacc[ct][2] * inv, oacc[ct][3] * inv);
      if (16 * ct + 4 * grp < DH) *reinterpret_cast<float4*>(orow + 16 * ct + 4 * grp) = v;
    }
    if (grp == 0) lse2[((long long)img * N + qrow) * H + h] = m + __builtin_amdgcn_logf(lsum);
  }
}

// ---------------------------------------------------------------------------------------------
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, DH <= 48 ? (NW == 8 ? 4 : 5) : 1) void attn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ O,
                                                         const float* __restrict__ lse2, const float* __restrict__ dO,
                                                         float* __restrict__ delta, float* __restrict__ dqkv,
                                                         int nimg, int N, int H, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  AttnSmem2<DH>& sm = *reinterpret_cast<AttnSmem2<DH>*>(smem_raw);
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

  float qf[DH / 4], dof[DH / 4];
  load_frag<DH>(qf, Q, ld, qrow, N, grp);
  load_frag<DH>(dof, dOp, D, qrow, N, grp);
  float dl = 0.f;
  {
    float of_[DH / 4];
    load_frag<DH>(of_, Op, D, qrow, N, grp);     // same k map as dof
#pragma unroll
    for (int s = 0; s < DH / 4; ++s) dl += dof[s] * of_[s];
  }
  dl = grp_sum(dl);
  const float l2 = (qrow < N) ? lse2[((long long)img * N + qrow) * H + h] : 0.f;
  if (qrow < N && grp == 0) delta[((long long)img * N + qrow) * H + h] = dl;

  f32x4 acc[NT16(DH)];
#pragma unroll
  for (int ct = 0; ct < NT16(DH); ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool wave_active = rb * 16 * NW + 16 * w < N;
  constexpr bool PRE = (NW == 8);
  StageF<DH, 64 * NW> stg;
  if (PRE) stg.load(K, ld, V, ld, 0, N);
  for (int k0 = 0; k0 < N; k0 += CH) {
    __syncthreads();
    if (PRE) stg.store(sm.s1, sm.s2);
    else {
      stage_rows<DH, 64 * NW>(sm.s1, K, ld, k0, N);
      stage_rows<DH, 64 * NW>(sm.s2, V, ld, k0, N);
    }
    __syncthreads();
    if (PRE && k0 + CH < N) stg.load(K, ld, V, ld, k0 + CH, N);
    if (!wave_active) continue;
    for_valid_tiles((N - k0 + 15) >> 4, [&](auto nv) {
      constexpr int NVT = decltype(nv)::value;
      f32x4 S[4], dP[4];
      score_tiles<DH, NVT>(S, sm.s1, qf, c, grp);
      score_tiles<DH, NVT>(dP, sm.s2, dof, c, grp);
#pragma unroll
      for (int tt = 0; tt < NVT; ++tt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          int key = k0 + 16 * tt + 4 * grp + reg;
          float pv = (key < N) ? __builtin_amdgcn_exp2f(S[tt][reg] * sc - l2) : 0.f;
          S[tt][reg] = pv * (dP[tt][reg] - dl);  // dS
        }
      accum_tiles<DH, NVT>(acc, sm.s1, S, c, grp);  // dQ^T += K^T dS^T
    });
  }
  if (qrow < N) {
    float* drow = dqkv + ((long long)img * N + qrow) * ld + h * DH;
#pragma unroll
    for (int ct = 0; ct < NT16(DH); ++ct) {
      float4 v = make_float4(acc[ct][0] * scale, acc[ct][1] * scale, acc[ct][2] * scale, acc[ct][3] * scale);
      if (16 * ct + 4 * grp < DH) *reinterpret_cast<float4*>(drow + 16 * ct + 4 * grp) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, DH <= 48 ? (NW == 8 ? 3 : 4) : 1) void attn_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ lse2,
                                                          const float* __restrict__ dO, const float* __restrict__ delta,
                                                          float* __restrict__ dqkv, int nimg, int N, int H, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  AttnSmem<DH>& sm = *reinterpret_cast<AttnSmem<DH>*>(smem_raw);
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

  float kf[DH / 4], vf[DH / 4];
  load_frag<DH>(kf, K, ld, krow, N, grp);
  load_frag<DH>(vf, V, ld, krow, N, grp);
  f32x4 dk[NT16(DH)], dv[NT16(DH)];
#pragma unroll
  for (int ct = 0; ct < NT16(DH); ++ct) {
    dk[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dv[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const bool wave_active = rb * 16 * NW + 16 * w < N;
  constexpr bool PRE = (NW == 8);
  float rc_l = 0.f, rc_d = 0.f;            // PRE: lse2 / delta of the next chunk's queries, prefetched with its rows
  auto load_rowc = [&](int qn) {
    if (threadIdx.x < CH) {
      const int q = qn + threadIdx.x;
      const long long ri = ((long long)img * N + q) * H + h;
      rc_l = (q < N) ? lse2[ri] : 0.f;
      rc_d = (q < N) ? delta[ri] : 0.f;
    }
  };
  StageF<DH, 64 * NW> stg;
  if (PRE) { stg.load(Q, ld, dOp, D, 0, N); load_rowc(0); }
  for (int q0 = 0; q0 < N; q0 += CH) {
    __syncthreads();
    if (PRE) stg.store(sm.s1, sm.s2);
    else {
      stage_rows<DH, 64 * NW>(sm.s1, Q, ld, q0, N);
      stage_rows<DH, 64 * NW>(sm.s2, dOp, D, q0, N);
      load_rowc(q0);
    }
    if (threadIdx.x < CH) {
      sm.rowc[threadIdx.x] = rc_l;
      sm.rowc[CH + threadIdx.x] = rc_d;
    }
    __syncthreads();
    if (PRE && q0 + CH < N) { stg.load(Q, ld, dOp, D, q0 + CH, N); load_rowc(q0 + CH); }
    if (!wave_active) continue;
    for_valid_tiles((N - q0 + 15) >> 4, [&](auto nv) {
      constexpr int NVT = decltype(nv)::value;
      f32x4 S[4], dP[4];
      score_tiles<DH, NVT>(S, sm.s1, kf, c, grp);   // S[q][key]
      score_tiles<DH, NVT>(dP, sm.s2, vf, c, grp);  // dP[q][key] = dO[q] . V[key]
#pragma unroll
      for (int tt = 0; tt < NVT; ++tt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          int ql = 16 * tt + 4 * grp + reg;
          float pv = (q0 + ql < N) ? __builtin_amdgcn_exp2f(S[tt][reg] * sc - sm.rowc[ql]) : 0.f;
          S[tt][reg] = pv;                                        // P
          dP[tt][reg] = pv * (dP[tt][reg] - sm.rowc[CH + ql]);    // dS
        }
      accum_tiles<DH, NVT>(dv, sm.s2, S, c, grp);   // dV^T += dO^T P
      accum_tiles<DH, NVT>(dk, sm.s1, dP, c, grp);  // dK^T += Q^T dS
    });
  }
  if (krow < N) {
    float* drow = dqkv + ((long long)img * N + krow) * ld + h * DH;
#pragma unroll
    for (int ct = 0; ct < NT16(DH); ++ct) {
      float4 a = make_float4(dk[ct][0] * scale, dk[ct][1] * scale, dk[ct][2] * scale, dk[ct][3] * scale);
      float4 b = make_float4(dv[ct][0], dv[ct][1], dv[ct][2], dv[ct][3]);
      if (16 * ct + 4 * grp < DH) {
        *reinterpret_cast<float4*>(drow + D + 16 * ct + 4 * grp) = a;
        *reinterpret_cast<float4*>(drow + 2 * D + 16 * ct + 4 * grp) = b;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// N <= 64 (the whole sequence is one chunk): both backward passes in one workgroup per (image, head).
// Q, K, V and dO are staged once and shared by the query-owner pass (dQ, delta) and the key-owner pass
// (dK, dV); delta = sum_key P dP crosses from one to the other through LDS, so neither O nor a delta
// array in HBM is read.  Halves the HBM traffic of the two-kernel path, which is what bounds it here.
template <int DH>
struct AttnSmem4 {
  static constexpr int LD = DH + 4;
  float sK[CH * LD], sV[CH * LD], sQ[CH * LD], sdO[CH * LD];
  float rowc[2 * CH];   // lse2 and delta of the 64 query rows
};

template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_small_kernel(const float* __restrict__ qkv, const float* __restrict__ lse2,
                                                             const float* __restrict__ dO, float* __restrict__ dqkv,
                                                             int N, int H, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  AttnSmem4<DH>& sm = *reinterpret_cast<AttnSmem4<DH>*>(smem_raw);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, grp = lane >> 4;
  const int img = blockIdx.z, h = blockIdx.y;
  const int ld = 3 * D;
  const float* base = qkv + (long long)img * N * ld;
  const float scale = rsqrtf((float)DH);
  const float sc = scale * CORA_LOG2E;
  stage_rows<DH>(sm.sQ, base + h * DH, ld, 0, N);
  stage_rows<DH>(sm.sK, base + D + h * DH, ld, 0, N);
  stage_rows<DH>(sm.sV, base + 2 * D + h * DH, ld, 0, N);
  stage_rows<DH>(sm.sdO, dO + (long long)img * N * D + h * DH, D, 0, N);
  if (threadIdx.x < CH) sm.rowc[threadIdx.x] = ((int)threadIdx.x < N) ? lse2[((long long)img * N + threadIdx.x) * H + h] : 0.f;
  __syncthreads();
  const int row = 16 * w + c;                    // this lane's own row in both passes
  float* drow = dqkv + ((long long)img * N + row) * ld + h * DH;
  {  // ---- pass 1: own queries
    float qf[DH / 4], dof[DH / 4];
    lds_frag<DH>(qf, sm.sQ, row, grp);
    lds_frag<DH>(dof, sm.sdO, row, grp);
    f32x4 S[4], dP[4];
    score_tiles<DH>(S, sm.sK, qf, c, grp);
    score_tiles<DH>(dP, sm.sV, dof, c, grp);
    const float l2 = sm.rowc[row];
    float dl = 0.f;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int key = 16 * tt + 4 * grp + reg;
        const float pv = (key < N && row < N) ? __builtin_amdgcn_exp2f(S[tt][reg] * sc - l2) : 0.f;
        S[tt][reg] = pv;
        dl += pv * dP[tt][reg];
      }
    dl = grp_sum(dl);                            // delta = sum_key P dP  (= dO . O)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) S[tt][reg] *= dP[tt][reg] - dl;   // dS
    f32x4 acc[NT16(DH)];
#pragma unroll
    for (int ct = 0; ct < NT16(DH); ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    accum_tiles<DH>(acc, sm.sK, S, c, grp);      // dQ^T += K^T dS^T
    if (row < N) {
#pragma unroll
      for (int ct = 0; ct < NT16(DH); ++ct)
        if (16 * ct + 4 * grp < DH)
          *reinterpret_cast<float4*>(drow + 16 * ct + 4 * grp) =
              make_float4(acc[ct][0] * scale, acc[ct][1] * scale, acc[ct][2] * scale, acc[ct][3] * scale);
    }
    if (grp == 0) sm.rowc[CH + row] = dl;
  }
  __syncthreads();
  {  // ---- pass 2: own keys
    float kf[DH / 4], vf[DH / 4];
    lds_frag<DH>(kf, sm.sK, row, grp);
    lds_frag<DH>(vf, sm.sV, row, grp);
    f32x4 S[4], dP[4];
    score_tiles<DH>(S, sm.sQ, kf, c, grp);       // S[q][key]
    score_tiles<DH>(dP, sm.sdO, vf, c, grp);     // dP[q][key] = dO[q] . V[key]
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int ql = 16 * tt + 4 * grp + reg;
        const float pv = (ql < N) ? __builtin_amdgcn_exp2f(S[tt][reg] * sc - sm.rowc[ql]) : 0.f;
        S[tt][reg] = pv;                                          // P
        dP[tt][reg] = pv * (dP[tt][reg] - sm.rowc[CH + ql]);      // dS
      }
    f32x4 dk[NT16(DH)], dv[NT16(DH)];
#pragma unroll
    for (int ct = 0; ct < NT16(DH); ++ct) { dk[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    accum_tiles<DH>(dv, sm.sdO, S, c, grp);      // dV^T += dO^T P
    accum_tiles<DH>(dk, sm.sQ, dP, c, grp);      // dK^T += Q^T dS
    if (row < N) {
#pragma unroll
      for (int ct = 0; ct < NT16(DH); ++ct)
        if (16 * ct + 4 * grp < DH) {
          *reinterpret_cast<float4*>(drow + D + 16 * ct + 4 * grp) =
              make_float4(dk[ct][0] * scale, dk[ct][1] * scale, dk[ct][2] * scale, dk[ct][3] * scale);
          *reinterpret_cast<float4*>(drow + 2 * D + 16 * ct + 4 * grp) = make_float4(dv[ct][0], dv[ct][1], dv[ct][2], dv[ct][3]);
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Single-pass backward for long sequences (N a multiple of 64, head dims 16 / 32 / 64; the row-tile path of
// fused_rows.hip, which supplies delta = rowsum(dO * O) from its own registers): one workgroup owns 16 NW keys
// and streams Q / dO chunks ONCE -- S and dP are computed once (5 products per (own, streamed) tile pair instead
// of the 7 of the two-kernel form above).  dV and dK contract over the streamed queries, which is what the score
// registers are laid out for; dQ contracts over the OWN keys (the lane index), so dS crosses the workgroup
// through LDS, transposed -- sT[key][q] -- and each wave multiplies a 16-query x 16-column slice of
// dQ = dS K_own against the workgroup's own K rows (staged once).  The N / (16 NW) key blocks of one
// (image, head) each hold a partial dQ: key block rb writes its own copy, dqa + rb * pstride, with plain stores
// (every element of every copy is written exactly once: nothing to zero, deterministic) and the consumer
// (rows_out) adds the copies up.  Summing them with float atomics into one array instead cost 0.12 of the
// kernel's 0.74 ms at the video shape (50 M uncoalesced atomics per launch).
template <int DH, int NW>
struct AttnSmemF {
  static constexpr int LD = DH + 4, LT = CH + 4;
  float s1[CH * LD];             // Q chunk
  float s2[CH * LD];             // dO chunk
  float sK[16 * NW * LD];        // own K rows
  float sT[16 * NW * LT];        // dS^T [own key][query of the chunk], already scaled by 1/sqrt(d)
  float rowc[2 * CH];            // lse2, delta of the chunk's queries
};

template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, 1) void attn_bwd_fused_kernel(const float* __restrict__ qkv, const float* __restrict__ lse2,
                                                                    const float* __restrict__ dO, const float* __restrict__ delta,
                                                                    float* __restrict__ dqa, long long pstride, float* __restrict__ dqkv,
                                                                    int nimg, int N, int H, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  AttnSmemF<DH, NW>& sm = *reinterpret_cast<AttnSmemF<DH, NW>*>(smem_raw);
  constexpr int LD = DH + 4, LT = CH + 4, NCT = DH / 16;
  constexpr int NTILE = 4 * NCT;                         // 16 x 16 tiles of the chunk's dQ^T [DH][64]
  constexpr int TPW = (NTILE + NW - 1) / NW;             // ... per wave
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, grp = lane >> 4;
  int rb, h, img;
  if (!attn_coords((N + 16 * NW - 1) / (16 * NW), H, nimg, rb, h, img)) return;
  const int ld = 3 * D;
  const float* base = qkv + (long long)img * N * ld;
  const float* Q = base + h * DH;
  const float* K = base + D + h * DH;
  const float* V = base + 2 * D + h * DH;
  const float* dOp = dO + (long long)img * N * D + h * DH;
  const int krow0 = rb * 16 * NW;
  const int krow = krow0 + 16 * w + c;
  const float scale = rsqrtf((float)DH);
  const float sc = scale * CORA_LOG2E;

  float kf[DH / 4], vf[DH / 4];
  load_frag<DH>(kf, K, ld, krow, N, grp);
  load_frag<DH>(vf, V, ld, krow, N, grp);
  for (int f = threadIdx.x; f < 16 * NW * (DH / 4); f += 64 * NW) {      // own K rows, once
    const int row = f / (DH / 4), c4 = f % (DH / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (krow0 + row < N) v = *reinterpret_cast<const float4*>(K + (long long)(krow0 + row) * ld + 4 * c4);
    *reinterpret_cast<float4*>(&sm.sK[row * LD + 4 * c4]) = v;
  }
  f32x4 dk[NCT], dv[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) { dk[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const bool key_ok = krow < N;

  // lse2 / delta of a chunk's 64 queries travel with its Q / dO rows: fetched into two registers of wave 0 a chunk ahead.
  // (Loaded between the two barriers they left all eight waves waiting for one global-load latency per chunk.)
  float rc_l = 0.f, rc_d = 0.f;
  auto load_rowc = [&](int qn) {
    if (threadIdx.x < CH) {
      const long long ri = ((long long)img * N + qn + threadIdx.x) * H + h;
      rc_l = lse2[ri]; rc_d = delta[ri];
    }
  };
  StageF<DH, 64 * NW> stg;
  stg.load(Q, ld, dOp, D, 0, N);
  load_rowc(0);
  for (int q0 = 0; q0 < N; q0 += CH) {
    __syncthreads();                         // the previous chunk's s1 / s2 / sT are no longer read
    stg.store(sm.s1, sm.s2);
    if (threadIdx.x < CH) {
      sm.rowc[threadIdx.x] = rc_l;
      sm.rowc[CH + threadIdx.x] = rc_d;
    }
    __syncthreads();
    if (q0 + CH < N) { stg.load(Q, ld, dOp, D, q0 + CH, N); load_rowc(q0 + CH); }
    {
      f32x4 S[4], dP[4];
      score_tiles<DH>(S, sm.s1, kf, c, grp);     // S[q][key]
      score_tiles<DH>(dP, sm.s2, vf, c, grp);    // dP[q][key] = dO[q] . V[key]
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int ql = 16 * tt + 4 * grp + reg;
          const float pv = key_ok ? __builtin_amdgcn_exp2f(S[tt][reg] * sc - sm.rowc[ql]) : 0.f;
          S[tt][reg] = pv;                                        // P
          dP[tt][reg] = pv * (dP[tt][reg] - sm.rowc[CH + ql]);    // dS
        }
      accum_tiles<DH>(dv, sm.s2, S, c, grp);     // dV^T += dO^T P
      accum_tiles<DH>(dk, sm.s1, dP, c, grp);    // dK^T += Q^T dS
      float* trow = &sm.sT[(16 * w + c) * LT + 4 * grp];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
        *reinterpret_cast<float4*>(trow + 16 * tt) =
            make_float4(dP[tt][0] * scale, dP[tt][1] * scale, dP[tt][2] * scale, dP[tt][3] * scale);
    }
    __syncthreads();                         // dS^T of all 16 NW own keys visible
    if (w * TPW < NTILE) {                   // dQ^T[dd][q] += sum_key K[key][dd] dS[q][key] over the workgroup's keys
      const int qt = (w * TPW) / NCT;
      f32x4 acc[TPW];
#pragma unroll
      for (int j = 0; j < TPW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int kt = 0; kt < NW; ++kt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int key = 16 * kt + 4 * grp + reg;
          const float b = sm.sT[key * LT + 16 * qt + c];
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            const int ct = (w * TPW + j) % NCT;
            acc[j] = mfma16(sm.sK[key * LD + 16 * ct + c], b, acc[j]);
          }
        }
      float* qrow = dqa + (long long)rb * pstride + ((long long)img * N + q0 + 16 * qt + c) * D + h * DH + 4 * grp;
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const int ct = (w * TPW + j) % NCT;
        *reinterpret_cast<float4*>(qrow + 16 * ct) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
      }
    }
  }
  if (key_ok) {
    float* drow = dqkv + ((long long)img * N + krow) * ld + h * DH;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      *reinterpret_cast<float4*>(drow + D + 16 * ct + 4 * grp) = make_float4(dk[ct][0] * scale, dk[ct][1] * scale, dk[ct][2] * scale, dk[ct][3] * scale);
      *reinterpret_cast<float4*>(drow + 2 * D + 16 * ct + 4 * grp) = make_float4(dv[ct][0], dv[ct][1], dv[ct][2], dv[ct][3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
template <int DH, int NW>
static int run_fwd_w(const float* qkv, float* O, float* lse2, int nimg, int N, int H, int D, hipStream_t st) {
  size_t sm = sizeof(AttnSmem2<DH>);
  static bool attr_set[64] = {};
  if (sm > 48 * 1024 && first_use_on_device(attr_set)) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<DH, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  }
  const int nrb = (N + 16 * NW - 1) / (16 * NW);
  hipLaunchKernelGGL((attn_fwd_kernel<DH, NW>), dim3(attn_grid(nrb, H, nimg)), dim3(64 * NW), sm, st, qkv, O, lse2, nimg, N, H, D);
  return 0;
}
template <int DH>
static int run_fwd(const float* qkv, float* O, float* lse2, int nimg, int N, int H, int D, hipStream_t st) {
  return attn_waves(nimg, N, H, false) == 8 ? run_fwd_w<DH, 8>(qkv, O, lse2, nimg, N, H, D, st)
                                     : run_fwd_w<DH, 4>(qkv, O, lse2, nimg, N, H, D, st);
}
template <int DH, int NW>
static int run_bwd_w(const float* qkv, const float* O, const float* lse2, const float* dO, float* delta,
                     float* dqkv, int nimg, int N, int H, int D, hipStream_t st) {
  size_t sm = sizeof(AttnSmem<DH>);
  static bool attr_set[64] = {};
  if (sm > 48 * 1024 && first_use_on_device(attr_set)) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_q_kernel<DH, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    (void)hipFuncSetAttribute((const void*)attn_bwd_kv_kernel<DH, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  }
  const int nrb = (N + 16 * NW - 1) / (16 * NW);
  const dim3 grid(attn_grid(nrb, H, nimg));
  hipLaunchKernelGGL((attn_bwd_q_kernel<DH, NW>), grid, dim3(64 * NW), sizeof(AttnSmem2<DH>), st, qkv, O, lse2, dO, delta, dqkv, nimg, N, H, D);
  hipLaunchKernelGGL((attn_bwd_kv_kernel<DH, NW>), grid, dim3(64 * NW), sm, st, qkv, lse2, dO, delta, dqkv, nimg, N, H, D);
  return 0;
}
template <int DH>
static int run_bwd(const float* qkv, const float* O, const float* lse2, const float* dO, float* delta,
                   float* dqkv, int nimg, int N, int H, int D, hipStream_t st) {
  if (N <= CH && DH <= 64) {   // one chunk: fused backward
    size_t sm4 = sizeof(AttnSmem4<DH>);
    static bool attr4[64] = {};
    if (sm4 > 48 * 1024 && first_use_on_device(attr4)) {
      if (hipFuncSetAttribute((const void*)attn_bwd_small_kernel<DH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm4) != hipSuccess) return -1;
    }
    hipLaunchKernelGGL((attn_bwd_small_kernel<DH>), dim3(1, H, nimg), dim3(256), sm4, st, qkv, lse2, dO, dqkv, N, H, D);
    return 0;
  }
  return attn_waves(nimg, N, H, true) == 8 ? run_bwd_w<DH, 8>(qkv, O, lse2, dO, delta, dqkv, nimg, N, H, D, st)
                                     : run_bwd_w<DH, 4>(qkv, O, lse2, dO, delta, dqkv, nimg, N, H, D, st);
}

#define DISPATCH_DH(dh, CALL)        \
  switch (dh) {                      \
    case 16: return CALL(16);        \
    case 24: return CALL(24);        \
    case 32: return CALL(32);        \
    case 48: return CALL(48);        \
    case 64: return CALL(64);        \
    case 96: return CALL(96);        \
    case 128: return CALL(128);      \
    default: return -1;              \
  }

template <int DH, int NW>
static int run_bwd_fused(const float* qkv, const float* lse2, const float* dO, const float* delta, float* dqa, long long pstride,
                         float* dqkv, int nimg, int N, int H, int D, hipStream_t st) {
  const size_t sm = sizeof(AttnSmemF<DH, NW>);
  static bool attr[64] = {};
  if (first_use_on_device(attr)) {
    if (hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<DH, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) return -1;
  }
  const int nrb = (N + 16 * NW - 1) / (16 * NW);
  hipLaunchKernelGGL((attn_bwd_fused_kernel<DH, NW>), dim3(attn_grid(nrb, H, nimg)), dim3(64 * NW), sm, st, qkv, lse2, dO, delta, dqa,
                     pstride, dqkv, nimg, N, H, D);
  return 0;
}
bool attn_bwd_fused_supported(int N, int H, int D) {
  const int dh = D / H;
  return (dh == 16 || dh == 32 || dh == 64) && N > CH && N % CH == 0;
}
// dq leaves as attn_bwd_fused_parts(N) partial copies (one per block of 128 own keys), copy k at dqa + k * pstride, each
// [nimg * N, D]; their sum is dq.  dqkv receives dk | dv only.  (A 4-wave form -- 64 own keys, two workgroups per CU --
// was measured at the video shape: 1.33 ms against 0.74 ms; the streamed chunks are staged twice as often.)
int attn_bwd_fused_parts(int N) { return (N + 127) / 128; }
int launch_attn_bwd_fused(const float* qkv, const float* lse2, const float* dO, const float* delta, float* dqa, long long pstride,
                          float* dqkv, int nimg, int N, int H, int D, hipStream_t st) {
  if (!attn_bwd_fused_supported(N, H, D)) return -1;
  switch (D / H) {
    case 16: return run_bwd_fused<16, 8>(qkv, lse2, dO, delta, dqa, pstride, dqkv, nimg, N, H, D, st);
    case 32: return run_bwd_fused<32, 8>(qkv, lse2, dO, delta, dqa, pstride, dqkv, nimg, N, H, D, st);
    case 64: return run_bwd_fused<64, 8>(qkv, lse2, dO, delta, dqa, pstride, dqkv, nimg, N, H, D, st);
  }
  return -1;
}

int launch_attn_fwd(const float* qkv, float* O, float* lse2, int nimg, int N, int H, int D, hipStream_t st) {
#define CALL(X) run_fwd<X>(qkv, O, lse2, nimg, N, H, D, st)
  DISPATCH_DH(D / H, CALL)
#undef CALL
}
int launch_attn_bwd(const float* qkv, const float* O, const float* lse2, const float* dO, float* delta,
                    float* dqkv, int nimg, int N, int H, int D, hipStream_t st) {
#define CALL(X) run_bwd<X>(qkv, O, lse2, dO, delta, dqkv, nimg, N, H, D, st)
  DISPATCH_DH(D / H, CALL)
#undef CALL
}
