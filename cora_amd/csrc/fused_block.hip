// fused_block.hip -- one workgroup = one (transformer block, image): the block's prediction
// u = 2x + MHA(LN1 x) + MLP(LN2(x + MHA(LN1 x))), the next vode's error e = h_next - u with its
// energy, and the hand-derived VJP dx = J^T e, all without leaving the CU (gfx950, fp32 MFMA).
//
// Replaces, for N = 64 tokens / one head, the 13 launches of the general path
// (/root/reference/pcx/nn/_layer.py:428-447 block, :359-402 attention, :296-315 LayerNorm;
// double residual /root/reference/src/decoder_transformer.py:238; VJP = what
// jax.value_and_grad(energy) w.r.t. the vodes computes, :396,445).
//
// Data layout "T" (per wave = 16 token rows): lane l = (r = l & 15, grp = l >> 4) holds
// X[row r][16 ct + 4 grp + reg], ct < D/16, reg < 4  -- i.e. exactly the B operand of
// v_mfma_f32_16x16x4_f32 for k-step (ct, reg), AND exactly what the instruction returns when the
// weight is the A operand (D[i = out][j = row]).  A chain of linear layers therefore needs no
// register shuffles and no LDS round trip for activations: only the weights stream through LDS
// (64x64 tiles, double buffered, shared by the 4 waves, read with ds_read_b128), and K/V/Q/dO
// cross waves through LDS for attention.  Row statistics (LayerNorm, softmax) are 16 in-register
// adds + two shuffles.
#include "common.h"

namespace {

constexpr int LD = 68;            // LDS row stride (floats): 64 + 4, keeps b128 alignment, conflict-free
constexpr int TILE = 64 * LD;     // one 64x64 tile

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// acc[ot] += sum_k T[16 ot + i][k] * xf[k]  for the 64 rows of an LDS tile (weights, K, Q, V or dO rows)
// -> acc[ot][reg'] = Y[row r][16 ot + 4 grp + reg']
__device__ __forceinline__ void gemm_tile(const float* __restrict__ t, const float* xf, f32x4* acc, int i, int grp) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
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

}  // namespace

// D = 64, M = 256, N = 64, H = 1
__global__ __launch_bounds__(256, 1) void pc_block_fused_kernel(FusedP p) {
  constexpr int D = 64, M = 256;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* bufK = lds;
  float* bufV = lds + TILE;
  float* bufQ = lds + 2 * TILE;
  float* bufDO = lds + 3 * TILE;
  float* wbuf = lds + 4 * TILE;          // 2 tiles
  float* rowc = lds + 6 * TILE;          // lse2[64], delta[64]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, grp = lane >> 4;
  const int g = blockIdx.x / p.B, b = blockIdx.x % p.B;
  const long long nl = (long long)p.B * 64 * D;
  const long long rowoff = ((long long)b * 64 + 16 * w + i) * D;   // this lane's token row
  const float* xrow = p.H1 + (long long)g * nl + rowoff;
  const float* hnext = p.H1 + (long long)(g + 1) * nl + rowoff;

  const float* Wqkv = p.W + p.oWqkv + (long long)g * 3 * D * D;
  const float* Wo = p.W + p.oWo + (long long)g * D * D;
  const float* W1 = p.W + p.oW1 + (long long)g * M * D;
  const float* W2 = p.W + p.oW2 + (long long)g * D * M;
  const float* W2T = p.WT + p.tW2 + (long long)g * M * D;      // [M][D]
  const float* W1T = p.WT + p.tW1 + (long long)g * D * M;      // [D][M]
  const float* WoT = p.WT + p.tWo + (long long)g * D * D;      // [D][D]
  const float* WqkvT = p.WT + p.tWqkv + (long long)g * D * 3 * D;  // [D][3D]

  // ---- weight tile pipeline: global -> registers (prefetch) -> LDS (double buffered) ------------
  float4 sr[4];
  int cur = 1;
  auto prefetch = [&](const float* src, int ldw) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int f = tid + 256 * k;
      sr[k] = *reinterpret_cast<const float4*>(src + (long long)(f >> 4) * ldw + 4 * (f & 15));
    }
  };
  // publish the prefetched tile; returns the LDS tile to read.  One barrier per tile: the buffer
  // being overwritten was last read two tiles ago, i.e. before the previous barrier.
  auto publish = [&]() -> const float* {
    cur ^= 1;
    float* dst = wbuf + cur * TILE;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int f = tid + 256 * k;
      *reinterpret_cast<float4*>(&dst[(f >> 4) * LD + 4 * (f & 15)]) = sr[k];
    }
    __syncthreads();
    return dst;
  };

  prefetch(Wqkv, D);

  // ---- forward ---------------------------------------------------------------------------------------
  float x[16], xh1[17], xin[16];
  load_rowT(x, xrow, D, grp);
  {
    float gm[16], bt[16];
    load_rowT(gm, p.W + p.oLn1g + (long long)g * D, D, grp);
    load_rowT(bt, p.W + p.oLn1b + (long long)g * D, D, grp);
    ln_rowT(x, gm, bt, xh1, xin);
  }
  f32x4 acc[4];
  float qf[16];
  {  // q, k, v = LN1(x) Wq^T, Wk^T, Wv^T
    const float* t = publish(); prefetch(Wqkv + 64 * D, D);
    zero4(acc); gemm_tile(t, xin, acc, i, grp); unpack(acc, qf);
    store_rowT(qf, &bufQ[(16 * w + i) * LD], grp);
    float kf[16];
    t = publish(); prefetch(Wqkv + 128 * D, D);
    zero4(acc); gemm_tile(t, xin, acc, i, grp); unpack(acc, kf);
    store_rowT(kf, &bufK[(16 * w + i) * LD], grp);
    t = publish(); prefetch(Wo, D);
    zero4(acc); gemm_tile(t, xin, acc, i, grp); unpack(acc, kf);
    store_rowT(kf, &bufV[(16 * w + i) * LD], grp);
  }
  __syncthreads();   // K, V, Q of all 64 tokens visible
  // attention: S[key][q] -> softmax over keys -> O
  float pf[16];
  const float sc = 0.125f * CORA_LOG2E;   // 1/sqrt(64) in base-2 units
  {
    zero4(acc); gemm_tile(bufK, qf, acc, i, grp); unpack(acc, pf);
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
    if (grp == 0) rowc[16 * w + i] = mx + __builtin_amdgcn_logf(ls);
  }
  float y[16];
  {
    float of[16];
    zero4(acc); accum_tile(bufV, pf, acc, i, grp); unpack(acc, of);
    const float* t = publish(); prefetch(W1, D);       // Wo
    zero4(acc); gemm_tile(t, of, acc, i, grp); unpack(acc, y);
#pragma unroll
    for (int k = 0; k < 16; ++k) y[k] += x[k];
  }
  float xh2[17], a[64], r[16];
  {
    float gm[16], bt[16], zn[16];
    load_rowT(gm, p.W + p.oLn2g + (long long)g * D, D, grp);
    load_rowT(bt, p.W + p.oLn2b + (long long)g * D, D, grp);
    ln_rowT(y, gm, bt, xh2, zn);
    // a = LN2(y) W1^T + b1   (4 row tiles of W1)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const float* t = publish();
      if (mt < 3) prefetch(W1 + (mt + 1) * 64 * D, D); else prefetch(W2, M);
      zero4(acc); gemm_tile(t, zn, acc, i, grp); unpack(acc, &a[16 * mt]);
      float b1[16];
      load_rowT(b1, p.W + p.ob1 + (long long)g * M + 64 * mt, M, grp);
#pragma unroll
      for (int k = 0; k < 16; ++k) a[16 * mt + k] += b1[k];
    }
    // z = gelu(a) W2^T + b2  (4 K-chunks of W2), u = x + y + z, r = h_next - u
    zero4(acc);
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const float* t = publish();
      if (kc < 3) prefetch(W2 + (kc + 1) * 64, M); else prefetch(W2T, D);
      float gl[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) gl[k] = cora_gelu(a[16 * kc + k]);
      gemm_tile(t, gl, acc, i, grp);
    }
    float z[16], b2[16], hn[16];
    unpack(acc, z);
    load_rowT(b2, p.W + p.ob2 + (long long)g * D, D, grp);
    load_rowT(hn, hnext, D, grp);
    float en = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float u = x[k] + y[k] + z[k] + b2[k];
      r[k] = hn[k] - u;
      en += 0.5f * r[k] * r[k] + p.l1 * fabsf(hn[k]) + p.l2 * hn[k] * hn[k];
    }
    store_rowT(r, p.E2 + (long long)g * nl + rowoff, grp);
    double tot = wave_sum_d((double)en);
    if (lane == 0) atomicAdd(&p.energy[g], tot);
  }

  // ---- backward (cotangent r) -----------------------------------------------------------------------
  // da = (r W2) * gelu'(a), in place over a   (4 row tiles of W2^T)
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const float* t = publish();
    if (mt < 3) prefetch(W2T + (mt + 1) * 64 * D, D); else prefetch(W1T, M);
    zero4(acc); gemm_tile(t, r, acc, i, grp);
    float dg[16];
    unpack(acc, dg);
#pragma unroll
    for (int k = 0; k < 16; ++k) a[16 * mt + k] = dg[k] * cora_gelu_grad(a[16 * mt + k]);
  }
  float dy[16];
  {  // dzn = da W1 (4 K-chunks of W1^T), dy = r + LN2^T(dzn)
    zero4(acc);
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const float* t = publish();
      if (kc < 3) prefetch(W1T + (kc + 1) * 64, M); else prefetch(WoT, D);
      gemm_tile(t, &a[16 * kc], acc, i, grp);
    }
    float dzn[16], gm[16];
    unpack(acc, dzn);
    load_rowT(gm, p.W + p.oLn2g + (long long)g * D, D, grp);
    ln_bwd_rowT(dzn, gm, xh2, dy);
#pragma unroll
    for (int k = 0; k < 16; ++k) dy[k] += r[k];
  }
  float dof[16];
  {  // dO = dy Wo
    const float* t = publish(); prefetch(WqkvT, 3 * D);
    zero4(acc); gemm_tile(t, dy, acc, i, grp); unpack(acc, dof);
    store_rowT(dof, &bufDO[(16 * w + i) * LD], grp);
  }
  // attention backward, pass A (own queries): dP, delta, dS, dQ
  float dq[16];
  {
    float dp[16];
    zero4(acc); gemm_tile(bufV, dof, acc, i, grp); unpack(acc, dp);     // dP[key][q] = V[key] . dO[q]
    float dl = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) dl += pf[k] * dp[k];
    dl = grp_sum(dl);                                                    // delta = sum_key P dP = dO . O
    if (grp == 0) rowc[64 + 16 * w + i] = dl;
#pragma unroll
    for (int k = 0; k < 16; ++k) dp[k] = pf[k] * (dp[k] - dl);           // dS
    zero4(acc); accum_tile(bufK, dp, acc, i, grp); unpack(acc, dq);
#pragma unroll
    for (int k = 0; k < 16; ++k) dq[k] *= 0.125f;
  }
  __syncthreads();   // dO rows and delta of all tokens visible
  // pass B (own keys): S, dP recomputed with the key on the lane; dV, dK
  float dk[16], dv[16];
  {
    float kf[16], vf[16], sp[16], dp[16];
    load_rowT(kf, &bufK[(16 * w + i) * LD], LD, grp);
    load_rowT(vf, &bufV[(16 * w + i) * LD], LD, grp);
    zero4(acc); gemm_tile(bufQ, kf, acc, i, grp); unpack(acc, sp);      // S[q][key]
    zero4(acc); gemm_tile(bufDO, vf, acc, i, grp); unpack(acc, dp);     // dP[q][key]
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      float4 l4 = *reinterpret_cast<const float4*>(&rowc[16 * tt + 4 * grp]);
      float4 d4 = *reinterpret_cast<const float4*>(&rowc[64 + 16 * tt + 4 * grp]);
      float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        float pv = __builtin_amdgcn_exp2f(sp[4 * tt + reg] * sc - lv[reg]);
        sp[4 * tt + reg] = pv;
        dp[4 * tt + reg] = pv * (dp[4 * tt + reg] - dvv[reg]);
      }
    }
    zero4(acc); accum_tile(bufDO, sp, acc, i, grp); unpack(acc, dv);    // dV = P^T dO
    zero4(acc); accum_tile(bufQ, dp, acc, i, grp); unpack(acc, dk);     // dK = dS^T Q
#pragma unroll
    for (int k = 0; k < 16; ++k) dk[k] *= 0.125f;
  }
  {  // dxn = [dq | dk | dv] Wqkv (3 K-chunks of Wqkv^T), dx = r + dy + LN1^T(dxn)
    zero4(acc);
    const float* t = publish(); prefetch(WqkvT + 64, 3 * D);
    gemm_tile(t, dq, acc, i, grp);
    t = publish(); prefetch(WqkvT + 128, 3 * D);
    gemm_tile(t, dk, acc, i, grp);
    t = publish();
    gemm_tile(t, dv, acc, i, grp);
    float dxn[16], gm[16], dx[16];
    unpack(acc, dxn);
    load_rowT(gm, p.W + p.oLn1g + (long long)g * D, D, grp);
    ln_bwd_rowT(dxn, gm, xh1, dx);
#pragma unroll
    for (int k = 0; k < 16; ++k) dx[k] += r[k] + dy[k];
    store_rowT(dx, p.BP1 + (long long)g * nl + rowoff, grp);
  }
}

size_t fused_block_lds_bytes() { return (size_t)(6 * TILE + 128) * sizeof(float); }

int launch_fused_block(const FusedP& p, int L, hipStream_t st) {
  static bool attr = false;
  size_t sm = fused_block_lds_bytes();
  if (!attr) {
    if (hipFuncSetAttribute((const void*)pc_block_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) return -1;
    attr = true;
  }
  hipLaunchKernelGGL(pc_block_fused_kernel, dim3(L * p.B), dim3(256), sm, st, p);
  return 0;
}

// ---- transposed weight copies: dst[c][r] = src[r][c] per group -----------------------------------
__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
  __shared__ float tile[32][33];
  const float* s = src + (long long)blockIdx.z * rows * cols;
  float* d = dst + (long long)blockIdx.z * rows * cols;
  int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int k = threadIdx.y; k < 32; k += 8) {
    int rr = r0 + k, cc = c0 + threadIdx.x;
    tile[k][threadIdx.x] = (rr < rows && cc < cols) ? s[(long long)rr * cols + cc] : 0.f;
  }
  __syncthreads();
  for (int k = threadIdx.y; k < 32; k += 8) {
    int cc = c0 + k, rr = r0 + threadIdx.x;
    if (rr < rows && cc < cols) d[(long long)cc * rows + rr] = tile[threadIdx.x][k];
  }
}
void launch_transpose(const float* src, float* dst, int rows, int cols, int groups, hipStream_t st) {
  dim3 grid((cols + 31) / 32, (rows + 31) / 32, groups);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, st, src, dst, rows, cols);
}

// ---- g_l = s_l * (e_l + l1 sign(h_l) + 2 l2 h_l - bp_l) in place over bp, layers 1..L, + sum of squares ----
__global__ __launch_bounds__(256) void grad_finish_kernel(float* __restrict__ gbp, const float* __restrict__ e,
                                                          const float* __restrict__ h, const float* __restrict__ gscale,
                                                          long long nl4, float l1, float l2, double* __restrict__ red) {
  __shared__ double sh[4];
  const int layer = blockIdx.y;
  const float s = gscale[layer];
  const long long base = (long long)layer * nl4;
  double acc = 0.0;
  for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nl4; k += (long long)gridDim.x * blockDim.x) {
    float4 bp = reinterpret_cast<const float4*>(gbp)[base + k];
    float4 ev = reinterpret_cast<const float4*>(e)[base + k];
    float o[4] = {ev.x, ev.y, ev.z, ev.w};
    if (l1 != 0.f || l2 != 0.f) {
      float4 hv = reinterpret_cast<const float4*>(h)[base + k];
      float hh[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] += l1 * ((hh[j] > 0.f) ? 1.f : ((hh[j] < 0.f) ? -1.f : 0.f)) + 2.f * l2 * hh[j];
    }
    float4 gv = make_float4(s * (o[0] - bp.x), s * (o[1] - bp.y), s * (o[2] - bp.z), s * (o[3] - bp.w));
    reinterpret_cast<float4*>(gbp)[base + k] = gv;
    acc += (double)gv.x * gv.x + (double)gv.y * gv.y + (double)gv.z * gv.z + (double)gv.w * gv.w;
  }
  double tot = block_sum_256(acc, sh);
  if (threadIdx.x == 0) atomicAdd(&red[layer * 8 + (blockIdx.x & 7)], tot);
}
void launch_grad_finish(float* gbp, const float* e, const float* h, const float* gscale, long long nl, int L,
                        float l1, float l2, double* red, hipStream_t st) {
  long long nl4 = nl / 4;
  long long bx = (nl4 + 255) / 256;
  if (bx > 512) bx = 512;
  hipLaunchKernelGGL(grad_finish_kernel, dim3((unsigned)bx, L), dim3(256), 0, st, gbp, e, h, gscale, nl4, l1, l2, red);
}
