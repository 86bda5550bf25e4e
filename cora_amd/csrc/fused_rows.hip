// fused_rows.hip -- the row-local phases of a D = 64 transformer block as three fused kernels, one
// workgroup per 64-row tile of one block, for ANY sequence length (BASELINE config 5: 16 frames x 64
// patches = 1024 tokens; also 64-wide models the per-(block, image) kernel of fused_block.hip does not take).
// Attention itself -- the only part of the block that couples rows -- stays in attn.hip between them:
//
//   rows_qkv : x -> LN1 -> q | k | v                                               (3 weight tiles)
//   attn_fwd : q, k, v -> att, lse                                                  (attn.hip)
//   rows_mid : att -> Wo + x = y -> LN2 -> W1 -> gelu -> W2 -> u = x + y + z + b2 -> e = h_next - u, energy,
//              then the transposes  da = (e W2) gelu'(a) -> dzn = da W1 -> dy = e + LN2^T(dzn) -> dO = dy Wo
//              (18 weight tiles; leaves e, dO and r + dy)
//   attn_bwd : q, k, v, dO -> dq | dk | dv                                          (attn.hip)
//   rows_out : [dq | dk | dv] Wqkv -> LN1^T -> J^T e = (r + dy) + LN1^T(..) -> g = s (e_own + reg - J^T e), sum g^2
//              (3 weight tiles; the vode-gradient assembly of the per-op path's last ln_bwd, fused)
//
// Same math as /root/reference/pcx/nn/_layer.py:428-447 (block), :296-315 (LayerNorm), the double residual of
// /root/reference/src/decoder_transformer.py:238 and the VJP jax.value_and_grad(energy) takes w.r.t. the vodes
// (:396,445); same layout T, weight-tile pipeline and register-resident MLP as fused_block.hip (layout_t.h).
// Replaces, per inference step, 4 LayerNorm launches and 8 GEMM launches whose operands all went through HBM
// (xn1, y, xn2, a [256 wide], da [256 wide], dzn, dy, dxn) by three launches that read x, att, h_next, dq|dk|dv and
// write q|k|v, e, dO, r + dy and g.
#include "common.h"
#include "layout_t.h"

namespace {

// ---- weight tile pipeline (as fused_block.hip): global -> registers one tile ahead -> LDS double buffer ----
#define PREFETCH(SRC, LDW)                                                                        \
  do {                                                                                            \
    const float* s_ = (SRC) + (long long)srow * (LDW) + scol;                                     \
    sr0 = *reinterpret_cast<const float4*>(s_);                                                   \
    sr1 = *reinterpret_cast<const float4*>(s_ + 16LL * (LDW));                                    \
    sr2 = *reinterpret_cast<const float4*>(s_ + 32LL * (LDW));                                    \
    sr3 = *reinterpret_cast<const float4*>(s_ + 48LL * (LDW));                                    \
  } while (0)
#define PUBLISH(S, TPTR)                                                                          \
  do {                                                                                            \
    float* d_ = wbuf + S * TILE + srow * LD + scol;                                               \
    *reinterpret_cast<float4*>(d_) = sr0;                                                         \
    *reinterpret_cast<float4*>(d_ + 16 * LD) = sr1;                                               \
    *reinterpret_cast<float4*>(d_ + 32 * LD) = sr2;                                               \
    *reinterpret_cast<float4*>(d_ + 48 * LD) = sr3;                                               \
    __syncthreads();                                                                              \
    TPTR = wbuf + S * TILE;                                                                       \
  } while (0)

// workgroup id -> (block g, 64-row tile): XCD x gets a contiguous eighth of the (block, tile) list, so its
// L2 sees the weights of at most two blocks (the hardware deals workgroup ids round-robin over the 8 XCDs)
__device__ __forceinline__ void rows_coords(const RowsP& p, int& g, long long& tile) {
  const long long ntile = p.rows / 64, n = (long long)p.L * ntile;
  long long j = blockIdx.x;
  if ((n & 7) == 0) j = (j & 7) * (n >> 3) + (j >> 3);
  g = (int)(j / ntile);
  tile = j % ntile;
}

// RMODE_CHAIN: status "forward" walk (INIT pass, reconstruction): one block per launch, rows_qkv as it is, rows_mid stops
// at u = x + y + MLP and writes it where the next block reads (p.E2), no backward
enum { RMODE_INFER = 0, RMODE_STORE = 1, RMODE_CHAIN = 2 };

// ---- edge layers as extra workgroups of the inference launches (P = 48 only, like fused_block.hip: edge_path) ----
//   top    (rows_qkv launch): e_1 = h_1 - (h_0 Wp^T + bp + pos), energy of vode 1, g_0 = -s_0 e_1 Wp and its sum of squares
//   bottom (rows_out launch): e_img = x_patch - (h_{L+1} Wout^T + bout), sensory energy,
//                             g_{L+1} = s_{L+1} (e_{L+1} + reg - e_img Wout) and its sum of squares
// (/root/reference/src/decoder_transformer.py:222-228, 244-256, 143-153).  e_1 must exist before rows_out reads it as
// the own error of vode 1, and e_{L+1} is written by rows_mid: hence one edge per launch.  Four GEMM launches of
// 16 k rows x K = 48 / 64 less per inference step (launch-latency bound: 82 of 1 320 us at the video shape).
__device__ __forceinline__ void rows_edge_top(const RowsP& p, float* lds, long long tile) {
  constexpr int D = 64, P = 48, LP = 52;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, grp = lane >> 4;
  float* sWp = lds;                  // [64][LP]   Wp   [D][P]
  float* sWpT = sWp + 64 * LP;       // [48][LD]   Wp^T [P][D]
  double* red = reinterpret_cast<double*>(lds + 2 * TILE);     // 8 doubles
  copy_to_lds(sWp, p.W + p.oWp, 64, P, LP, tid);
  copy_to_lds(sWpT, p.WT + p.tWp, P, D, LD, tid);
  const long long row = tile * 64 + 16 * w + i;
  float h0[12], h1[16], bp[16], ps[16];
  load_rowT_n<3>(h0, p.H0 + row * P, grp);
  load_rowT_n<4>(h1, p.X + row * D, grp);
  load_rowT_n<4>(bp, p.W + p.obp, grp);
  load_rowT_n<4>(ps, p.pos + (row % p.N) * D, grp);
  __syncthreads();
  f32x4 acc[4];
  float e1[16], u[16];
  zero4(acc); gemm_tile_g<4, 3, LP>(sWp, h0, acc, i, grp); unpack(acc, u);
  float en = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    e1[k] = h1[k] - (u[k] + bp[k] + ps[k]);
    en += 0.5f * e1[k] * e1[k] + p.l1 * fabsf(h1[k]) + p.l2 * h1[k] * h1[k];
  }
  store_rowT_n<4>(e1, const_cast<float*>(p.E1) + row * D, grp);
  zero4(acc); gemm_tile_g<3, 4, LD>(sWpT, e1, acc, i, grp);
  float g0[12];
  double ss = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) { float v = -p.s0 * acc[k][j]; g0[4 * k + j] = v; ss += (double)v * v; }
  store_rowT_n<3>(g0, p.G0 + row * P, grp);
  const double et = wave_sum_d((double)en);
  ss = wave_sum_d(ss);
  if (lane == 0) { red[w] = et; red[4 + w] = ss; }
  __syncthreads();
  if (tid == 0) {
    atomicAdd(&p.energy[-1], red[0] + red[1] + red[2] + red[3]);                      // energy[] points at vode 2
    atomicAdd(&p.sumsq[-8 + (int)(tile & 7)], red[4] + red[5] + red[6] + red[7]);     // sumsq[] points at vode 1's slots
  }
}

__device__ __forceinline__ void rows_edge_bottom(const RowsP& p, float* lds, long long tile) {
  constexpr int D = 64, P = 48, LP = 52;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, grp = lane >> 4;
  float* sWo = lds;                  // [48][LD]   Wout [P][D]
  float* sWoT = sWo + 48 * LD;       // [64][LP]   Wout^T [D][P]
  double* red = reinterpret_cast<double*>(lds + 2 * TILE);
  copy_to_lds(sWo, p.W + p.oWout, P, D, LD, tid);
  copy_to_lds(sWoT, p.WT + p.tWout, D, P, LP, tid);
  const long long row = tile * 64 + 16 * w + i;
  const long long nl = p.rows * D;
  float hl[16], xp[12], bo[12], eL[16];
  load_rowT_n<4>(hl, p.X + (long long)p.L * nl + row * D, grp);
  load_rowT_n<3>(xp, p.XP + row * P, grp);
  load_rowT_n<3>(bo, p.W + p.obout, grp);
  load_rowT_n<4>(eL, p.E2 + (long long)(p.L - 1) * nl + row * D, grp);       // e_{L+1}, written by rows_mid
  __syncthreads();
  f32x4 acc[4];
  float ei[12];
  zero4(acc); gemm_tile_g<3, 4, LD>(sWo, hl, acc, i, grp);
  float en = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) { float e = xp[4 * k + j] - (acc[k][j] + bo[4 * k + j]); ei[4 * k + j] = e; en += 0.5f * e * e; }
  store_rowT_n<3>(ei, p.EIMG + row * P, grp);
  zero4(acc); gemm_tile_g<4, 3, LP>(sWoT, ei, acc, i, grp);
  float bpl[16];
  unpack(acc, bpl);
  const float s = p.gscale[p.L];
  double ss = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float h = hl[k];
    const float sg = (h > 0.f) ? 1.f : ((h < 0.f) ? -1.f : 0.f);
    const float gv = s * (eL[k] + p.l1 * sg + 2.f * p.l2 * h - bpl[k]);
    bpl[k] = gv;
    ss += (double)gv * (double)gv;
  }
  store_rowT_n<4>(bpl, p.G1 + (long long)p.L * nl + row * D, grp);
  const double et = wave_sum_d((double)en);
  ss = wave_sum_d(ss);
  if (lane == 0) { red[w] = et; red[4 + w] = ss; }
  __syncthreads();
  if (tid == 0) {
    atomicAdd(p.esens, red[0] + red[1] + red[2] + red[3]);
    atomicAdd(&p.sumsq[(long long)p.L * 8 + (int)(tile & 7)], red[4] + red[5] + red[6] + red[7]);
  }
}

}  // namespace

// ---- rows_qkv: q | k | v = LN1(x) [Wq | Wk | Wv]^T --------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void rows_qkv_kernel(RowsP p) {
  constexpr int D = 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wbuf = lds;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, grp = lane >> 4;
  if (MODE == RMODE_INFER && (long long)blockIdx.x >= (long long)p.L * (p.rows / 64)) {   // the last rows / 64 workgroups: top edge
    rows_edge_top(p, lds, (long long)blockIdx.x - (long long)p.L * (p.rows / 64));
    return;
  }
  int g; long long tile;
  rows_coords(p, g, tile);
  const long long nl = p.rows * D;
  const long long row = tile * 64 + 16 * w + i;
  const long long srow_g = (long long)g * p.rows + row;
  const float* Wqkv = p.W + p.oWqkv + (long long)g * 3 * D * D;
  float4 sr0, sr1, sr2, sr3;
  const int srow = tid >> 4, scol = 4 * (tid & 15);
  PREFETCH(Wqkv, D);
  float x[16], xin[16];
  load_rowT(x, p.X + (long long)g * nl + row * D, D, grp);
  {
    float gm[16], bt[16], xh[17];
    load_rowT(gm, p.W + p.oLn1g + (long long)g * D, D, grp);
    load_rowT(bt, p.W + p.oLn1b + (long long)g * D, D, grp);
    ln_rowT(x, gm, bt, xh, xin);
    if (MODE == RMODE_STORE) store_rowT(xin, p.S_xn1 + srow_g * D, grp);
  }
  const float* t;
  f32x4 acc[4];
  float o[16];
  float* dst = p.qkv + srow_g * 3 * D;
  PUBLISH(0, t); PREFETCH(Wqkv + 64 * D, D);
  zero4(acc); gemm_tile(t, xin, acc, i, grp); unpack(acc, o);
  store_rowT(o, dst, grp);
  PUBLISH(1, t); PREFETCH(Wqkv + 128 * D, D);
  zero4(acc); gemm_tile(t, xin, acc, i, grp); unpack(acc, o);
  store_rowT(o, dst + D, grp);
  PUBLISH(0, t);
  zero4(acc); gemm_tile(t, xin, acc, i, grp); unpack(acc, o);
  store_rowT(o, dst + 2 * D, grp);
}

// ---- rows_mid: Wo + residual, LN2, MLP, error + energy, and the transposes back to dO ------------------------
template <int MODE>
__global__ __launch_bounds__(256, 2) void rows_mid_kernel(RowsP p) {
  constexpr int D = 64, M = 256;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wbuf = lds;                     // 2 tiles
  float* pb = lds + 2 * TILE;            // [b1 256 | ln2 gamma 64 | ln2 beta 64 | b2 64]
  double* esh = reinterpret_cast<double*>(lds + 2 * TILE + 448);   // 4 doubles
  constexpr int PB_LN2G = 256, PB_LN2B = 320, PB_B2 = 384;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, grp = lane >> 4;
  int g; long long tile;
  rows_coords(p, g, tile);
  const long long nl = p.rows * D;
  const long long row = tile * 64 + 16 * w + i;
  const long long srow_g = (long long)g * p.rows + row;
  const long long voff = (long long)g * nl + row * D;
  float* cpart = (MODE == RMODE_STORE) ? p.colpart + ((long long)(g * (p.rows / 64) + tile) * 4 + w) * COLPART : nullptr;

  const float* Wo = p.W + p.oWo + (long long)g * D * D;
  const float* W1 = p.W + p.oW1 + (long long)g * M * D;
  const float* W2 = p.W + p.oW2 + (long long)g * D * M;
  const float* W2T = p.WT + p.tW2 + (long long)g * M * D;      // [M][D]
  const float* W1T = p.WT + p.tW1 + (long long)g * D * M;      // [D][M]
  const float* WoT = p.WT + p.tWo + (long long)g * D * D;      // [D][D]
  if (tid < 112) {
    const float* src = tid < 64 ? p.W + p.ob1 + (long long)g * M + 4 * tid
                     : tid < 80 ? p.W + p.oLn2g + (long long)g * D + 4 * (tid - 64)
                     : tid < 96 ? p.W + p.oLn2b + (long long)g * D + 4 * (tid - 80)
                                : p.W + p.ob2 + (long long)g * D + 4 * (tid - 96);
    *reinterpret_cast<float4*>(pb + 4 * tid) = *reinterpret_cast<const float4*>(src);   // visible after the first tile barrier
  }
  float4 sr0, sr1, sr2, sr3;
  const int srow = tid >> 4, scol = 4 * (tid & 15);
  PREFETCH(Wo, D);
  const float* t;
  f32x4 acc[4];

  // ---- forward
  float x[16], y[16];
  float a_keep[MODE == RMODE_INFER ? 64 : 1];    // gelu'(a), kept for the backward (the STORE pass parks it in S_da)
  load_rowT(x, p.X + voff, D, grp);
  {
    float of[16];
    load_rowT(of, p.att + srow_g * D, D, grp);
    PUBLISH(1, t); PREFETCH(W1, D);               // Wo; -> W1[0]
    zero4(acc); gemm_tile(t, of, acc, i, grp); unpack(acc, y);
#pragma unroll
    for (int k = 0; k < 16; ++k) y[k] += x[k];
  }
  float hn[16];
  float en = 0.f;
  {
    float zn[16];
    {
      float gm[16], bt[16], xh[17];
      load_rowT(gm, pb + PB_LN2G, D, grp);
      load_rowT(bt, pb + PB_LN2B, D, grp);
      ln_rowT(y, gm, bt, xh, zn);
      if (MODE == RMODE_STORE) store_rowT(zn, p.S_xn2 + srow_g * D, grp);
    }
    if (MODE != RMODE_CHAIN) load_rowT(hn, p.X + voff + nl, D, grp);     // the next vode's rows: the HBM latency runs under the MLP
    f32x4 zacc[4];
    zero4(zacc);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      PUBLISH(0, t);                                        // W1[mt]
      PREFETCH(W2 + mt * 64, M);                            // -> W2[:, mt]
      float am[16];
      {
        float b1[16];
        load_rowT(b1, pb + 64 * mt, M, grp);
        zero4(acc); gemm_tile(t, zn, acc, i, grp); unpack(acc, am);
#pragma unroll
        for (int k = 0; k < 16; ++k) am[k] += b1[k];
      }
      if (MODE == RMODE_INFER) {
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
          f32x2 g2, gp2;
          cora_gelu_both2((f32x2){am[k], am[k + 1]}, g2, gp2);
          am[k] = g2.x; am[k + 1] = g2.y;
          a_keep[16 * mt + k] = gp2.x; a_keep[16 * mt + k + 1] = gp2.y;
        }
      } else if (MODE == RMODE_CHAIN) {
#pragma unroll
        for (int k = 0; k < 16; ++k) am[k] = cora_gelu(am[k]);
      } else {
        float gpv[16];
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
          f32x2 g2, gp2;
          cora_gelu_both2((f32x2){am[k], am[k + 1]}, g2, gp2);
          am[k] = g2.x; am[k + 1] = g2.y;
          gpv[k] = gp2.x; gpv[k + 1] = gp2.y;
        }
        store_rowT(am, p.S_a + srow_g * M + 64 * mt, grp);       // gelu(a): operand of dW2
        store_rowT(gpv, p.S_da + srow_g * M + 64 * mt, grp);     // gelu'(a) waits where da will land
      }
      PUBLISH(1, t);                                        // W2[:, mt]
      if (mt < 3) PREFETCH(W1 + (mt + 1) * 64 * D, D);
      else if (MODE != RMODE_CHAIN) PREFETCH(W2T, D);       // -> W2^T[0]
      gemm_tile(t, am, zacc, i, grp);
    }
    float z[16], b2[16];
    unpack(zacc, z);
    load_rowT(b2, pb + PB_B2, D, grp);
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = x[k] + y[k] + z[k] + b2[k];     // x now holds u
  }
  if (MODE == RMODE_CHAIN) {   // the next block's input rows
    store_rowT(x, p.E2 + voff, grp);
    return;
  }
  float r[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    r[k] = hn[k] - x[k];
    en += 0.5f * r[k] * r[k] + p.l1 * fabsf(hn[k]) + p.l2 * hn[k] * hn[k];
  }
  store_rowT(r, p.E2 + voff, grp);
  if (MODE == RMODE_STORE) colsum_store<4>(r, cpart + CP_B2, i, grp);

  // ---- backward (cotangent r)
  float dy[16];
  {
    f32x4 dacc[4];
    zero4(dacc);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      PUBLISH(0, t);                                        // W2^T[mt]
      PREFETCH(W1T + mt * 64, M);                           // -> W1^T[:, mt]
      float am[16], dg[16];     // am: gelu'(a)
      if (MODE == RMODE_INFER) {
#pragma unroll
        for (int k = 0; k < 16; ++k) am[k] = a_keep[16 * mt + k];
      } else {
        load_rowT(am, p.S_da + srow_g * M + 64 * mt, M, grp);
      }
      zero4(acc); gemm_tile(t, r, acc, i, grp); unpack(acc, dg);
#pragma unroll
      for (int k = 0; k < 16; ++k) dg[k] *= am[k];
      if (MODE == RMODE_STORE) {
        store_rowT(dg, p.S_da + srow_g * M + 64 * mt, grp);
        colsum_store<4>(dg, cpart + CP_B1 + 64 * mt, i, grp);
      }
      PUBLISH(1, t);                                        // W1^T[:, mt]
      if (mt < 3) PREFETCH(W2T + (mt + 1) * 64 * D, D); else PREFETCH(WoT, D);
      gemm_tile(t, dg, dacc, i, grp);
    }
    float dzn[16], gm[16], bt[16], xh[17], tmp[16];
    unpack(dacc, dzn);
    load_rowT(gm, pb + PB_LN2G, D, grp);
#pragma unroll
    for (int k = 0; k < 16; ++k) bt[k] = 0.f;
    ln_rowT(y, gm, bt, xh, tmp);           // LN2's normalised input recomputed from y (still in registers)
    ln_bwd_rowT(dzn, gm, xh, dy);
#pragma unroll
    for (int k = 0; k < 16; ++k) dy[k] += r[k];
    if (MODE == RMODE_STORE) {
      store_rowT(dy, p.S_dy + srow_g * D, grp);
      colsum_store<4>(dzn, cpart + CP_LN2B, i, grp);
#pragma unroll
      for (int k = 0; k < 16; ++k) dzn[k] *= xh[k];
      colsum_store<4>(dzn, cpart + CP_LN2G, i, grp);
    }
  }
  {  // dO = dy Wo ; r + dy goes on to rows_out
    float dof[16], of[16];
    if (p.delta) load_rowT(of, p.att + srow_g * D, D, grp);   // again (16 registers through the MLP would cost more)
    PUBLISH(0, t);                                          // Wo^T
    zero4(acc); gemm_tile(t, dy, acc, i, grp); unpack(acc, dof);
    store_rowT(dof, p.dO + srow_g * D, grp);
    if (p.delta) {   // delta[row, head] = dO . att over the head's columns (what the attention backward subtracts from dP)
      const int nch = 4 / p.H;                              // 16-column groups per head (H = 1, 2, 4)
      for (int hh = 0; hh < p.H; ++hh) {
        float dl = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) dl += ((k >> 2) / nch == hh) ? dof[k] * of[k] : 0.f;
        dl = grp_sum(dl);
        if (grp == 0) p.delta[srow_g * p.H + hh] = dl;
      }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] += dy[k];
    store_rowT(r, p.rdy + srow_g * D, grp);
  }
  // energy of vode g + 2: one atomic per workgroup, issued last (an atomic in the middle of the kernel holds
  // every later load of the wave behind it in the in-order memory queue)
  {
    double tot = wave_sum_d((double)en);
    if (lane == 0) esh[w] = tot;
    __syncthreads();
    if (tid == 0) atomicAdd(&p.energy[g], esh[0] + esh[1] + esh[2] + esh[3]);
  }
}

// ---- rows_out: Wqkv^T, LN1^T, residual sum, vode-gradient assembly -----------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void rows_out_kernel(RowsP p) {
  constexpr int D = 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wbuf = lds;
  double* ssh = reinterpret_cast<double*>(lds + 2 * TILE);   // 4 doubles
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, grp = lane >> 4;
  if (MODE == RMODE_INFER && (long long)blockIdx.x >= (long long)p.L * (p.rows / 64)) {   // the last rows / 64 workgroups: bottom edge
    rows_edge_bottom(p, lds, (long long)blockIdx.x - (long long)p.L * (p.rows / 64));
    return;
  }
  int g; long long tile;
  rows_coords(p, g, tile);
  const long long nl = p.rows * D;
  const long long row = tile * 64 + 16 * w + i;
  const long long srow_g = (long long)g * p.rows + row;
  const long long voff = (long long)g * nl + row * D;
  const float* WqkvT = p.WT + p.tWqkv + (long long)g * D * 3 * D;  // [D][3D]
  float4 sr0, sr1, sr2, sr3;
  const int srow = tid >> 4, scol = 4 * (tid & 15);
  PREFETCH(WqkvT, 3 * D);
  float dq[16], dk[16], dv[16];
  float* src = p.dqkv + srow_g * 3 * D;
  if (p.dqa) {   // single-pass attention backward: dq = sum of its per-key-block partial copies
    load_rowT(dq, p.dqa + srow_g * D, D, grp);
    for (int kb = 1; kb < p.ndq; ++kb) {
      float part[16];
      load_rowT(part, p.dqa + kb * p.dq_stride + srow_g * D, D, grp);
#pragma unroll
      for (int k = 0; k < 16; ++k) dq[k] += part[k];
    }
    if (MODE == RMODE_STORE) store_rowT(dq, src, grp);     // dWqkv contracts the contiguous [dq | dk | dv] rows
  } else {
    load_rowT(dq, src, D, grp);
  }
  load_rowT(dk, src + D, D, grp);
  load_rowT(dv, src + 2 * D, D, grp);
  const float* t;
  f32x4 acc[4];
  zero4(acc);
  PUBLISH(0, t); PREFETCH(WqkvT + 64, 3 * D);
  gemm_tile(t, dq, acc, i, grp);
  PUBLISH(1, t); PREFETCH(WqkvT + 128, 3 * D);
  gemm_tile(t, dk, acc, i, grp);
  float xv[16], rd[16], eo[16];
  load_rowT(xv, p.X + voff, D, grp);               // these three arrive under the last tile
  load_rowT(rd, p.rdy + srow_g * D, D, grp);
  if (MODE == RMODE_INFER) load_rowT(eo, p.E1 + voff, D, grp);
  PUBLISH(0, t);
  gemm_tile(t, dv, acc, i, grp);
  float dxn[16], gm[16], bt[16], xh[17], tmp[16];
  unpack(acc, dxn);
  load_rowT(gm, p.W + p.oLn1g + (long long)g * D, D, grp);
#pragma unroll
  for (int k = 0; k < 16; ++k) bt[k] = 0.f;
  ln_rowT(xv, gm, bt, xh, tmp);                    // LN1's normalised input recomputed from x
  ln_bwd_rowT(dxn, gm, xh, dk);
  if (MODE == RMODE_STORE) {
    float* cpart = p.colpart + ((long long)(g * (p.rows / 64) + tile) * 4 + w) * COLPART;
    colsum_store<4>(dxn, cpart + CP_LN1B, i, grp);
#pragma unroll
    for (int k = 0; k < 16; ++k) dxn[k] *= xh[k];
    colsum_store<4>(dxn, cpart + CP_LN1G, i, grp);
    return;                                        // the learning pass needs no vode gradient
  }
  // g_{g+1} = s (e_{g+1} + l1 sign(h) + 2 l2 h - J^T e_{g+2}),  J^T e = (r + dy) + LN1^T(dxn)
  const float s = p.gscale[g];
  double ss = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float h = xv[k];
    const float sg = (h > 0.f) ? 1.f : ((h < 0.f) ? -1.f : 0.f);
    const float gv = s * (eo[k] + p.l1 * sg + 2.f * p.l2 * h - (rd[k] + dk[k]));
    dq[k] = gv;
    ss += (double)gv * (double)gv;
  }
  store_rowT(dq, p.G1 + voff, grp);
  ss = wave_sum_d(ss);
  if (lane == 0) ssh[w] = ss;
  __syncthreads();
  if (tid == 0) atomicAdd(&p.sumsq[(long long)g * 8 + (blockIdx.x & 7)], ssh[0] + ssh[1] + ssh[2] + ssh[3]);
}

#undef PREFETCH
#undef PUBLISH

// ---- dW[bias / LayerNorm parameters] -= sum over the wave slices of colpart (block groups only: the edge
// layers of this path are per-op GEMMs with their own column sums) ------------------------------------------
__global__ __launch_bounds__(256) void rows_colpart_reduce_kernel(RowsP p, float* dW) {
  __shared__ float sh[4][64];
  const int g = blockIdx.x, c0 = 64 * blockIdx.y;
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const long long slices = 4 * (p.rows / 64);
  const float* src = p.colpart + (long long)g * slices * COLPART + c0 + cx;
  const long long per = (slices + gridDim.z - 1) / gridDim.z;
  const long long r0 = blockIdx.z * per, r1 = min(slices, r0 + per);
  float s = 0.f;
  for (long long rr = r0 + ry; rr < r1; rr += 4) s += src[rr * COLPART];
  sh[ry][cx] = s;
  __syncthreads();
  if (ry == 0) {
    s = sh[0][cx] + sh[1][cx] + sh[2][cx] + sh[3][cx];
    long long off;
    if (c0 < CP_B1) off = p.ob2 + (long long)g * 64;
    else if (c0 < CP_LN2B) off = p.ob1 + (long long)g * 256 + (c0 - CP_B1);
    else if (c0 < CP_LN2G) off = p.oLn2b + (long long)g * 64;
    else if (c0 < CP_LN1B) off = p.oLn2g + (long long)g * 64;
    else if (c0 < CP_LN1G) off = p.oLn1b + (long long)g * 64;
    else off = p.oLn1g + (long long)g * 64;
    atomicAdd(&dW[off + cx], -s);
  }
}

// ---- host side ------------------------------------------------------------------------------------------------
bool fused_rows_supported(long long rows, int D, int M) { return D == 64 && M == 256 && rows > 0 && rows % 64 == 0; }
size_t fused_rows_colpart_floats(int L, long long rows) { return (size_t)L * (size_t)(rows / 64) * 4 * COLPART; }

static size_t rows_lds_small() { return (size_t)(2 * TILE + 16) * sizeof(float); }   // edge workgroups: 6 592 floats + 8 doubles at 2 TILE
static size_t rows_lds_mid() { return (size_t)(2 * TILE + 448 + 16) * sizeof(float); }

static int rows_attrs() {
  static bool attr[64] = {};
  if (!first_use_on_device(attr)) return 0;
  const int sm = (int)rows_lds_mid();
  if (hipFuncSetAttribute((const void*)rows_mid_kernel<RMODE_INFER>, hipFuncAttributeMaxDynamicSharedMemorySize, sm) != hipSuccess) return -1;
  if (hipFuncSetAttribute((const void*)rows_mid_kernel<RMODE_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, sm) != hipSuccess) return -1;
  if (hipFuncSetAttribute((const void*)rows_mid_kernel<RMODE_CHAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, sm) != hipSuccess) return -1;
  return 0;
}

static bool rows_ok(const RowsP& p) { return p.L > 0 && p.rows > 0 && p.rows % 64 == 0 && (long long)p.L * (p.rows / 64) < (1LL << 31); }

int launch_rows_qkv(const RowsP& p, int store, hipStream_t st) {
  if (!rows_ok(p)) return -1;
  const dim3 grid((unsigned)((p.L + ((p.edge && !store) ? 1 : 0)) * (p.rows / 64)));
  if (store == RMODE_CHAIN) hipLaunchKernelGGL(rows_qkv_kernel<RMODE_CHAIN>, grid, dim3(256), rows_lds_small(), st, p);
  else if (store) hipLaunchKernelGGL(rows_qkv_kernel<RMODE_STORE>, grid, dim3(256), rows_lds_small(), st, p);
  else hipLaunchKernelGGL(rows_qkv_kernel<RMODE_INFER>, grid, dim3(256), rows_lds_small(), st, p);
  return 0;
}
int launch_rows_mid(const RowsP& p, int store, hipStream_t st) {
  if (!rows_ok(p) || rows_attrs()) return -1;
  const dim3 grid((unsigned)(p.L * (p.rows / 64)));
  if (store == RMODE_CHAIN) hipLaunchKernelGGL(rows_mid_kernel<RMODE_CHAIN>, grid, dim3(256), rows_lds_mid(), st, p);
  else if (store) hipLaunchKernelGGL(rows_mid_kernel<RMODE_STORE>, grid, dim3(256), rows_lds_mid(), st, p);
  else hipLaunchKernelGGL(rows_mid_kernel<RMODE_INFER>, grid, dim3(256), rows_lds_mid(), st, p);
  return 0;
}
int launch_rows_out(const RowsP& p, int store, hipStream_t st) {
  if (!rows_ok(p)) return -1;
  const dim3 grid((unsigned)((p.L + ((p.edge && !store) ? 1 : 0)) * (p.rows / 64)));
  if (store) hipLaunchKernelGGL(rows_out_kernel<RMODE_STORE>, grid, dim3(256), rows_lds_small(), st, p);
  else hipLaunchKernelGGL(rows_out_kernel<RMODE_INFER>, grid, dim3(256), rows_lds_small(), st, p);
  return 0;
}
void launch_rows_colpart_reduce(const RowsP& p, float* dW, hipStream_t st) {
  hipLaunchKernelGGL(rows_colpart_reduce_kernel, dim3(p.L, COLPART / 64, 16), dim3(256), 0, st, p, dW);
}
