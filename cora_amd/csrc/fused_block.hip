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

#include "layout_t.h"
#include "vode_mult.h"



// ---- edge layers of one image: vode 0 <-> vode 1 (patch projection) and vode L+1 <-> sensory ------
//   e_1 = h_1 - (h_0 Wp^T + bp + pos)          g_0 = -s_0 e_1 Wp              (vode 0 has no own energy)
//   e_img = x_patch - (h_{L+1} Wout^T + bout)   bp_{L+1} = e_img Wout          (finished by vode_grad_kernel)
// (/root/reference/src/decoder_transformer.py:222-228, 244-256, 143-153)
// LAZY: the rows come in as h - eta (mult g + mu m) of the previous step (FusedP::lazy); this workgroup is the one that
// reads vode 0 and vode L+1 as INPUTS, so it writes their new values and momenta; vode 1 it only reads (as a target).
template <bool LAZY>
__device__ __forceinline__ void edge_path(const FusedP& p, float* lds, int b) {
  constexpr int D = 64, P = 48, LP = 52;   // P = 48 patch dim; LP = padded stride of 48-wide tiles
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, grp = lane >> 4;
  float* sWp = lds;                  // [64][LP]   Wp   [D][P]
  float* sWpT = sWp + 64 * LP;       // [48][LD]   Wp^T [P][D]
  float* sWo = sWpT + 48 * LD;       // [48][LD]   Wout [P][D]
  float* sWoT = sWo + 48 * LD;       // [64][LP]   Wout^T [D][P]
  copy_to_lds(sWp, p.W + p.oWp, 64, P, LP, tid);
  copy_to_lds(sWpT, p.WT + p.tWp, P, D, LD, tid);
  copy_to_lds(sWo, p.W + p.oWout, P, D, LD, tid);
  copy_to_lds(sWoT, p.WT + p.tWout, D, P, LP, tid);
  const long long tok = (long long)b * 64 + 16 * w + i;     // token row of this lane
  float h0[12], h1[16], hl[16], xp[12];
  load_rowT_n<3>(h0, p.H0 + tok * P, grp);
  load_rowT_n<4>(h1, p.H1 + tok * D, grp);
  load_rowT_n<4>(hl, p.H1 + (long long)p.L * p.B * 64 * D + tok * D, grp);
  load_rowT_n<3>(xp, p.XP + tok * P, grp);
  if (LAZY) {
    const long long oL = (long long)p.L * p.B * 64 * D + tok * D;
    float g0[12], m0[12], g1[16], m1[16], gl[16], ml[16];
    load_rowT_n<3>(g0, p.G0o + tok * P, grp); load_rowT_n<3>(m0, p.M0o + tok * P, grp);
    load_rowT_n<4>(g1, p.Go1 + tok * D, grp); load_rowT_n<4>(m1, p.Mo1 + tok * D, grp);
    load_rowT_n<4>(gl, p.Go1 + oL, grp); load_rowT_n<4>(ml, p.Mo1 + oL, grp);
    float c0, c1, cl;
    vode_multipliers_wave(p.up, 0, 1, p.L + 1, c0, c1, cl);
    const float eta = p.up.eta[p.up.t], mu = p.up.mu;
#pragma unroll
    for (int k = 0; k < 12; ++k) { m0[k] = g0[k] * c0 + mu * m0[k]; h0[k] -= eta * m0[k]; }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      m1[k] = g1[k] * c1 + mu * m1[k]; h1[k] -= eta * m1[k];
      ml[k] = gl[k] * cl + mu * ml[k]; hl[k] -= eta * ml[k];
    }
    store_rowT_n<3>(h0, p.H0n + tok * P, grp); store_rowT_n<3>(m0, p.M0n + tok * P, grp);
    store_rowT_n<4>(hl, p.Hn1 + oL, grp); store_rowT_n<4>(ml, p.Mn1 + oL, grp);
  }
  __syncthreads();
  f32x4 acc[4];
  // ---- top: e_1, energy of vode 1, g_0
  float e1[16];
  {
    float bp[16], ps[16], u[16];
    load_rowT_n<4>(bp, p.W + p.obp, grp);
    load_rowT_n<4>(ps, p.pos + (long long)(16 * w + i) * D, grp);
    zero4(acc); gemm_tile_g<4, 3, LP>(sWp, h0, acc, i, grp); unpack(acc, u);
    float en = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      e1[k] = h1[k] - (u[k] + bp[k] + ps[k]);
      en += 0.5f * e1[k] * e1[k] + p.l1 * fabsf(h1[k]) + p.l2 * h1[k] * h1[k];
    }
    store_rowT_n<4>(e1, p.E1 + tok * D, grp);
    if (p.dW) colsum_store<4>(e1, p.colpart + (long long)p.L * p.B * 4 * COLPART + ((long long)b * 4 + w) * CP_EDGE, i, grp);
    double tot = wave_sum_d((double)en);
    if (p.energy && lane == 0) atomicAdd(&p.energy[-1], tot);   // energy[] points at vode 2; vode 1 is [-1]
  }
  {
    zero4(acc); gemm_tile_g<3, 4, LD>(sWpT, e1, acc, i, grp);
    float g0[12];
    double ss = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) { float v = -p.s0 * acc[k][j]; g0[4 * k + j] = v; ss += (double)v * v; }
    store_rowT_n<3>(g0, p.G0 + tok * P, grp);
    ss = wave_sum_d(ss);
    if (lane == 0) atomicAdd(&p.sumsq0[(b * 4 + w) & 7], ss);
  }
  // ---- bottom: e_img, energy of the sensory vode, bp_{L+1}
  {
    float bo[12], ei[12];
    load_rowT_n<3>(bo, p.W + p.obout, grp);
    zero4(acc); gemm_tile_g<3, 4, LD>(sWo, hl, acc, i, grp);
    float en = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) { float e = xp[4 * k + j] - (acc[k][j] + bo[4 * k + j]); ei[4 * k + j] = e; en += 0.5f * e * e; }
    store_rowT_n<3>(ei, p.EIMG + tok * P, grp);
    if (p.dW) colsum_store<3>(ei, p.colpart + (long long)p.L * p.B * 4 * COLPART + ((long long)b * 4 + w) * CP_EDGE + 64, i, grp);
    double tot = wave_sum_d((double)en);
    if (lane == 0) {                                         // vode L+2 (sensory)
      if (p.energy) atomicAdd(&p.energy[p.L], tot);
      else atomicAdd(&p.esens[(b * 4 + w) & 7], tot);
    }
    zero4(acc); gemm_tile_g<4, 3, LP>(sWoT, ei, acc, i, grp);
    float bpl[16];
    unpack(acc, bpl);
    store_rowT_n<4>(bpl, p.BP1 + (long long)p.L * p.B * 64 * D + tok * D, grp);
  }
}

// One transformer block of one image (D = 64, M = 256, N = 64, H = 1, 2 or 4 heads).
//   MODE_INFER: x = h_{g+1} from HBM; writes e_{g+2}, its energy and J^T e (the inference step)
//   MODE_STORE: the same, and also leaves every tensor the weight-gradient GEMMs need in HBM
//   MODE_CHAIN: x comes in registers (xio), forward only, the prediction u goes back in xio
//   MODE_LAZY:  MODE_INFER that first applies the previous step's vode update to the rows it loads (FusedP::lazy)
enum { MODE_INFER = 0, MODE_STORE = 1, MODE_CHAIN = 2, MODE_LAZY = 3 };

// Diagnostic build only (-DCORA_STAMPS): wave 0 of each workgroup records s_memtime at phase
// boundaries into p.dbg[blockIdx.x * 16 + k].  Never compiled into the shipped library.
#ifdef CORA_STAMPS
#define STAMP(K) STAMP_IF(CORA_STAMPS == 1 || (K) <= 1, K)
#if CORA_STAMPS == 2
#define SUB(K) STAMP_IF(1, K)      /* -DCORA_STAMPS=2: zoom into the first two MLP-backward tile pairs */
#else
#define SUB(K) do {} while (0)
#endif
#define STAMP_IF(COND, K)                                                                          \
  do {                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    if ((COND) && MODE == MODE_INFER && p.dbg) {                                                             \
      unsigned long long t_;                                                                       \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                 \
      if (threadIdx.x == 0) p.dbg[(long long)blockIdx.x * 16 + (K)] = t_;                          \
    }                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)
#else
#define STAMP(K) do {} while (0)
#define SUB(K) do {} while (0)
#endif


template <int MODE, int H>
__device__ __forceinline__ void block_body(const FusedP& p, float* lds, int g, int b, float* xio) {
  constexpr int D = 64, M = 256;
  constexpr bool INFER = (MODE == MODE_INFER || MODE == MODE_LAZY);
  constexpr int NCH = 4 / H;                 // 16-column groups per head (head dim 64 / H)
  constexpr float qscale = (H == 1) ? 0.125f : (H == 2 ? 0.17677669529663687f : 0.25f);   // 1 / sqrt(head dim)
  // 4 tiles (69.6 KB) so two workgroups share a CU: K and V persist from the forward into the
  // backward; the weight double buffer is only live during GEMM phases and is reused for Q and dO
  // (all 64 tokens) during the attention backward.
  float* bufK = lds;
  float* bufV = lds + TILE;
  float* wbuf = lds + 2 * TILE;          // 2 tiles
  float* bufQ = lds + 2 * TILE;          // aliases wbuf[0]
  float* bufDO = lds + 3 * TILE;         // aliases wbuf[1]
  float* rowc = lds + 4 * TILE;          // lse2[H][64], delta[H][64]
  // small parameters of this block, copied once: fetched from global at their point of use they sat
  // behind a barrier and cost an exposed L2 latency each (b1 four times per block)
  float* pb = lds + 4 * TILE + 512;      // [b1 256 | ln2 gamma 64 | ln2 beta 64 | b2 64 | ln1 gamma 64]
  constexpr int PB_LN2G = 256, PB_LN2B = 320, PB_B2 = 384, PB_LN1G = 448;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, grp = lane >> 4;
  const long long nl = (long long)p.B * 64 * D;
  const long long rowoff = ((long long)b * 64 + 16 * w + i) * D;   // this lane's token row
  const long long voff = (long long)g * nl + rowoff;               // into the vode-1-based arenas
  const float* xrow = p.H1 + voff;
  const float* hnext = p.H1 + voff + nl;
  const long long srow_g = ((long long)g * p.B + b) * 64 + 16 * w + i;   // row in the [L*B*N, .] scratch tensors
  float* cpart = (MODE == MODE_STORE) ? p.colpart + (((long long)g * p.B + b) * 4 + w) * COLPART : nullptr;
  // Register diet (the kernel must fit 256 VGPRs for two waves per SIMD): tensors that are produced
  // in the forward and only needed again in the backward (P; in the STORE pass also a) are
  // parked in an L2-resident stash instead of registers and reloaded at their point of use.
  float* st_p = p.S_stash + srow_g * (H * D);   // H x 64: one row of P per head
  float* st_a = p.S_a + srow_g * M;

  const float* Wqkv = p.W + p.oWqkv + (long long)g * 3 * D * D;
  const float* Wo = p.W + p.oWo + (long long)g * D * D;
  const float* W1 = p.W + p.oW1 + (long long)g * M * D;
  const float* W2 = p.W + p.oW2 + (long long)g * D * M;
  const float* W2T = p.WT + p.tW2 + (long long)g * M * D;      // [M][D]
  const float* W1T = p.WT + p.tW1 + (long long)g * D * M;      // [D][M]
  const float* WoT = p.WT + p.tWo + (long long)g * D * D;      // [D][D]
  const float* WqkvT = p.WT + p.tWqkv + (long long)g * D * 3 * D;  // [D][3D]
  const float* ln1g = p.W + p.oLn1g + (long long)g * D;
  const float* ln1b = p.W + p.oLn1b + (long long)g * D;
  // (MODE_LAZY) the clip multipliers' sums of squares: this wave's first load, so that waiting for it waits for nothing else
  double ss_first = 0.0;
  float eta_lazy = 0.f;      // ... and the step size: loaded behind the rows, its wait would wait for all of them
  if (MODE == MODE_LAZY) { ss_first = vode_multipliers_first_load(p.up); eta_lazy = p.up.eta[p.up.t]; }
  __builtin_amdgcn_sched_barrier(0);
  // the parameter block: loaded here, written to LDS just before the first tile barrier -- written at once it put a
  // full memory latency in front of everything else the first two waves issue
  float4 pbv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 128) {
    const long long off = tid < 64 ? (long long)p.ob1 + (long long)g * M + 4 * tid
                        : tid < 80 ? (long long)p.oLn2g + (long long)g * D + 4 * (tid - 64)
                        : tid < 96 ? (long long)p.oLn2b + (long long)g * D + 4 * (tid - 80)
                        : tid < 112 ? (long long)p.ob2 + (long long)g * D + 4 * (tid - 96)
                                    : (long long)p.oLn1g + (long long)g * D + 4 * (tid - 112);
    pbv = *reinterpret_cast<const float4*>(p.W + off);
  }

  // ---- weight tile pipeline: global -> registers (prefetched one tile ahead) -> LDS (double buffered)
  // 24 tiles per block, tile k goes to LDS buffer k&1.  While tile k is computed, tile k+1 is in
  // flight into the staging registers (a second staging set, two tiles ahead, bought nothing and
  // cost the 16 registers that now hold part of the MLP pre-activation).  Named registers + macros (not lambdas over an
  // array: hipcc then keeps the staging registers in scratch).  Thread t stages float4 #t, t+256,
  // t+512, t+768 of the 64x64 tile: row = f >> 4, 16-byte column = f & 15.
  float4 sr0, sr1, sr2, sr3;
  const int srow = tid >> 4, scol = 4 * (tid & 15);
#define PREFETCH(SRC, LDW)                                                                        \
  do {                                                                                            \
    const float* s_ = (SRC) + (long long)srow * (LDW) + scol;                                     \
    sr0 = *reinterpret_cast<const float4*>(s_);                                                   \
    sr1 = *reinterpret_cast<const float4*>(s_ + 16LL * (LDW));                                    \
    sr2 = *reinterpret_cast<const float4*>(s_ + 32LL * (LDW));                                    \
    sr3 = *reinterpret_cast<const float4*>(s_ + 48LL * (LDW));                                    \
  } while (0)
  // publish the staged tile into LDS buffer S.  One barrier per tile: the buffer being overwritten
  // was last read two tiles ago, i.e. before the previous barrier.
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

  PREFETCH(Wqkv, D);               // tile 0: Wq
  const float* t;
  f32x4 acc[4];
  const float sc = qscale * CORA_LOG2E;   // 1/sqrt(head dim) in base-2 units

  STAMP(0);
  // ---- forward ---------------------------------------------------------------------------------------
  float x[16], y[16];
  float hn[16];                               // the next vode's rows (target of this block's prediction)
  float gn_lazy[MODE == MODE_LAZY ? 16 : 1], mn_lazy[MODE == MODE_LAZY ? 16 : 1], cn_lazy = 0.f;
  float gm1[16], bt1[16];                     // LayerNorm-1 gamma / beta
  float qkeep[MODE == MODE_CHAIN ? 1 : 16];   // q of the forward, kept for the attention backward
  // gelu'(a) of the MLP pre-activation a (64 values per lane): the inference step has the registers to keep it for
  // the backward; the STORE pass writes it to HBM anyway (weight-gradient GEMM operand) and reloads it.
  float a_keep[INFER ? 64 : 1];
  if (MODE == MODE_CHAIN) {
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = xio[k];
  } else if (MODE == MODE_LAZY) {
    // the update of the previous step, applied on the way in: x is this block's INPUT vode, so this workgroup writes its
    // new value and momentum (new arenas: the old ones are still being read by the block below, whose target it is);
    // hn, the TARGET vode, is only read -- the block above writes it.  Its three loads go out behind the v tile and are
    // consumed after the attention (swept: with the input rows at the head of the workgroup -1.5 %, any earlier point in
    // between proportionally; later -- at the MLP, where the plain kernel loads hn -- the three rows spill), and the
    // updated rows wait in this lane's slot of the error tensor (written at the end of the forward anyway) instead of
    // 16 registers across the MLP's neighbours.
    float gx[16], mx[16];
    load_rowT(x, xrow, D, grp);
    load_rowT(gx, p.Go1 + voff, D, grp);
    load_rowT(mx, p.Mo1 + voff, D, grp);
    load_rowT(gm1, ln1g, D, grp);                     // LayerNorm-1 parameters: behind the update they would cost a latency of their own
    load_rowT(bt1, ln1b, D, grp);
    __builtin_amdgcn_sched_barrier(0);                // everything above is in flight while the multipliers are worked out
    float cx, cn_, c3_;
    vode_multipliers_wave(p.up, g + 1, g + 2, g + 1, cx, cn_, c3_, ss_first);
    cn_lazy = cn_;
    const float eta = eta_lazy, mu = p.up.mu;
#pragma unroll
    for (int k = 0; k < 16; ++k) { mx[k] = gx[k] * cx + mu * mx[k]; x[k] -= eta * mx[k]; }
    store_rowT(mx, p.Mn1 + voff, grp);
    store_rowT(x, p.Hn1 + voff, grp);
  } else {
    load_rowT(x, xrow, D, grp);
  }
  {
    float xin[16];
    {
      float xh[17];
      if (MODE != MODE_LAZY) {
        load_rowT(gm1, ln1g, D, grp);
        load_rowT(bt1, ln1b, D, grp);
      }
      ln_rowT(x, gm1, bt1, xh, xin);
      if (MODE == MODE_STORE) store_rowT(xin, p.S_xn1 + srow_g * D, grp);
    }
    STAMP(1);
    float qf[16];
    if (tid < 128) *reinterpret_cast<float4*>(pb + 4 * tid) = pbv;   // visible after the first tile barrier
    {  // q, k, v = LN1(x) Wq^T, Wk^T, Wv^T
      PUBLISH(0, t); PREFETCH(Wqkv + 64 * D, D);    // -> tile 1: Wk
     
      zero4(acc); gemm_tile(t, xin, acc, i, grp); unpack(acc, qf);
     
      if (MODE != MODE_CHAIN) {
#pragma unroll
        for (int k = 0; k < 16; ++k) qkeep[k] = qf[k];
      }
      float kf[16];
     
      PUBLISH(1, t); PREFETCH(Wqkv + 128 * D, D);   // -> tile 2: Wv
     
      zero4(acc); gemm_tile(t, xin, acc, i, grp); unpack(acc, kf);
     
      store_rowT(kf, &bufK[(16 * w + i) * LD], grp);
     
      PUBLISH(0, t); PREFETCH(Wo, D);               // -> tile 3: Wo
     
      zero4(acc); gemm_tile(t, xin, acc, i, grp); unpack(acc, kf);
      if (MODE == MODE_LAZY) {   // the target vode's rows, gradient and momentum (see the head of the function)
        load_rowT(hn, hnext, D, grp);
        load_rowT(gn_lazy, p.Go1 + voff + nl, D, grp);
        load_rowT(mn_lazy, p.Mo1 + voff + nl, D, grp);
      }
     
      store_rowT(kf, &bufV[(16 * w + i) * LD], grp);
    }
    __syncthreads();   // K, V of all 64 tokens visible
   
    STAMP(2);
    // attention: S[key][q] -> softmax over keys -> O
    float of[16];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float pf[16];
      zero4(acc); gemm_tile_h<NCH>(bufK, qf, acc, i, grp, NCH * h); unpack(acc, pf);
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
      if (MODE != MODE_CHAIN) store_rowT(pf, st_p + 64 * h, grp);
      f32x4 oa[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) oa[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      accum_tile_h<NCH>(bufV, pf, oa, i, grp, NCH * h);
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) of[4 * (NCH * h + c) + reg] = oa[c][reg];
    }
    if (MODE == MODE_STORE) store_rowT(of, p.S_att + srow_g * D, grp);
    if (MODE == MODE_LAZY) {
      const float eta = eta_lazy, mu = p.up.mu;
#pragma unroll
      for (int k = 0; k < 16; ++k) hn[k] -= eta * (gn_lazy[k] * cn_lazy + mu * mn_lazy[k]);
      store_rowT(hn, p.E2 + (long long)g * nl + rowoff, grp);
    }
    STAMP(3);
    PUBLISH(1, t); PREFETCH(W1, D);   // Wo; -> tile 4: W1[0]
    zero4(acc); gemm_tile(t, of, acc, i, grp); unpack(acc, y);
#pragma unroll
    for (int k = 0; k < 16; ++k) y[k] += x[k];
  }
  {
    float zn[16];
    {
      float gm[16], bt[16], xh[17];
      load_rowT(gm, pb + PB_LN2G, D, grp);
      load_rowT(bt, pb + PB_LN2B, D, grp);
      ln_rowT(y, gm, bt, xh, zn);
      if (MODE == MODE_STORE) store_rowT(zn, p.S_xn2 + srow_g * D, grp);
    }
    STAMP(4);
    // the next vode's rows (target of this block's prediction): fetched here so the HBM latency runs
    // under the MLP instead of in front of the error
    if (MODE != MODE_CHAIN) load_rowT(hn, MODE == MODE_LAZY ? p.E2 + (long long)g * nl + rowoff : hnext, D, grp);
    // a_mt = LN2(y) W1[mt]^T + b1 ; z += gelu(a_mt) W2[:, mt]^T   (row tile of W1, K-chunk of W2, interleaved)
    f32x4 zacc[4];
    zero4(zacc);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      PUBLISH(0, t);                                        // tile 4+2mt: W1[mt]
      PREFETCH(W2 + mt * 64, M);                            // -> tile 5+2mt: W2[:, mt]
      float am[16];
      {
        float b1[16];
        load_rowT(b1, pb + 64 * mt, M, grp);
        zero4(acc); gemm_tile(t, zn, acc, i, grp); unpack(acc, am);
#pragma unroll
        for (int k = 0; k < 16; ++k) am[k] += b1[k];
      }
      if (INFER) {   // keep gelu'(a) for the backward: it shares the exp2 / rcp with gelu(a)
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
          f32x2 g2, gp2;
          cora_gelu_both2((f32x2){am[k], am[k + 1]}, g2, gp2);
          am[k] = g2.x; am[k + 1] = g2.y;
          a_keep[16 * mt + k] = gp2.x; a_keep[16 * mt + k + 1] = gp2.y;
        }
      } else if (MODE == MODE_STORE) {
        // the weight-gradient GEMM wants gelu(a) (S_a); gelu'(a) waits in the S_da slot that da overwrites in the backward
        float gpv[16];
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
          f32x2 g2, gp2;
          cora_gelu_both2((f32x2){am[k], am[k + 1]}, g2, gp2);
          am[k] = g2.x; am[k + 1] = g2.y;
          gpv[k] = gp2.x; gpv[k + 1] = gp2.y;
        }
        store_rowT(am, st_a + 64 * mt, grp);
        store_rowT(gpv, p.S_da + srow_g * M + 64 * mt, grp);
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) am[k] = cora_gelu(am[k]);
      }
      PUBLISH(1, t);                                        // tile 5+2mt: W2[:, mt]
      if (mt < 3) PREFETCH(W1 + (mt + 1) * 64 * D, D);      // -> tile 6+2mt
      else if (MODE != MODE_CHAIN) PREFETCH(W2T, D);        // -> tile 12: W2^T[0]
      gemm_tile(t, am, zacc, i, grp);
    }
    STAMP(5);
    float z[16], b2[16];
    unpack(zacc, z);
    load_rowT(b2, pb + PB_B2, D, grp);
    if (MODE == MODE_CHAIN) {
#pragma unroll
      for (int k = 0; k < 16; ++k) xio[k] = x[k] + y[k] + z[k] + b2[k];
      return;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = x[k] + y[k] + z[k] + b2[k];     // x now holds u; y stays for LN2's backward
  }
  float r[16];
  {
    float en = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      r[k] = hn[k] - x[k];
      en += 0.5f * r[k] * r[k] + p.l1 * fabsf(hn[k]) + p.l2 * hn[k] * hn[k];
    }
    store_rowT(r, p.E2 + (long long)g * nl + rowoff, grp);
    double tot = wave_sum_d((double)en);
    if (MODE == MODE_STORE && lane == 0) atomicAdd(&p.energy[g], tot);
    if (MODE == MODE_STORE) colsum_store<4>(r, cpart + CP_B2, i, grp);
  }

  STAMP(6);
  // ---- backward (cotangent r) -----------------------------------------------------------------------
  // da_mt = (r W2)[mt] * gelu'(a_mt) ; dzn += da_mt W1[mt]   (row tile of W2^T, K-chunk of W1^T, interleaved)
  float dy[16];
  {
    f32x4 dacc[4];
    zero4(dacc);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (mt < 2) SUB(2 + 6 * mt);
      PUBLISH(0, t);                                        // tile 12+2mt: W2^T[mt]
      if (mt < 2) SUB(3 + 6 * mt);
      PREFETCH(W1T + mt * 64, M);                           // -> tile 13+2mt: W1^T[:, mt]
      float am[16], dg[16];     // am: gelu'(a)
      if (INFER) {
#pragma unroll
        for (int k = 0; k < 16; ++k) am[k] = a_keep[16 * mt + k];
      } else {
        load_rowT(am, p.S_da + srow_g * M + 64 * mt, M, grp);   // gelu'(a), parked by the forward
      }
      if (mt < 2) SUB(4 + 6 * mt);
      zero4(acc); gemm_tile(t, r, acc, i, grp); unpack(acc, dg);
#pragma unroll
      for (int k = 0; k < 16; ++k) dg[k] *= am[k];
      if (MODE == MODE_STORE) {
        store_rowT(dg, p.S_da + srow_g * M + 64 * mt, grp);
        colsum_store<4>(dg, cpart + CP_B1 + 64 * mt, i, grp);
      }
      if (mt < 2) SUB(5 + 6 * mt);
      PUBLISH(1, t);                                        // tile 13+2mt: W1^T[:, mt]
      if (mt < 2) SUB(6 + 6 * mt);
      if (mt < 3) PREFETCH(W2T + (mt + 1) * 64 * D, D); else PREFETCH(WoT, D);   // -> tile 14+2mt / 20
      gemm_tile(t, dg, dacc, i, grp);
      if (mt < 2) SUB(7 + 6 * mt);
    }
    STAMP(7);
    // dy = r + LN2^T(dzn); LN2's normalised input is recomputed from y (still in registers)
    float dzn[16], gm[16], bt[16], xh[17], tmp[16];
    const float* yv = y;
    unpack(dacc, dzn);
    load_rowT(gm, pb + PB_LN2G, D, grp);
#pragma unroll
    for (int k = 0; k < 16; ++k) bt[k] = 0.f;
    ln_rowT(yv, gm, bt, xh, tmp);
    ln_bwd_rowT(dzn, gm, xh, dy);
#pragma unroll
    for (int k = 0; k < 16; ++k) dy[k] += r[k];
    if (MODE == MODE_STORE) {
      store_rowT(dy, p.S_dy + srow_g * D, grp);
      colsum_store<4>(dzn, cpart + CP_LN2B, i, grp);
#pragma unroll
      for (int k = 0; k < 16; ++k) dzn[k] *= xh[k];
      colsum_store<4>(dzn, cpart + CP_LN2G, i, grp);
    }
  }
  STAMP(8);
  float pT[H == 1 ? 16 : 1], dsT[H == 1 ? 16 : 1];
  {  // dO = dy Wo ; from here on r + dy is all that is needed of either
    float dof[16];
    if (H == 1) load_rowT(pT, st_p, D, grp);        // P back from the stash under the Wo^T tile (the MLP's registers are free again)
    PUBLISH(0, t); PREFETCH(WqkvT, 3 * D);          // tile 20: Wo^T; -> tile 21 (waits in registers through the attention backward)
    zero4(acc); gemm_tile(t, dy, acc, i, grp); unpack(acc, dof);
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] += dy[k];
    __syncthreads();   // every wave is done with the weight buffers: reuse them for Q and dO
    store_rowT(dof, &bufDO[(16 * w + i) * LD], grp);
    store_rowT(qkeep, &bufQ[(16 * w + i) * LD], grp);
    STAMP(9);
    // attention backward, pass A (own queries): dP, delta, dS, dQ -- head by head
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float pf_[16], dp_[16];
      float* pf = (H == 1) ? pT : pf_;     // one head: P and dS outlive the pass (handed to pass B through LDS)
      float* dp = (H == 1) ? dsT : dp_;
      if (H > 1) load_rowT(pf, st_p + 64 * h, D, grp);
      zero4(acc); gemm_tile_h<NCH>(bufV, dof, acc, i, grp, NCH * h); unpack(acc, dp);   // dP[key][q] = V[key] . dO[q]
      float dl = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) dl += pf[k] * dp[k];
      dl = grp_sum(dl);                                                    // delta = sum_key P dP = dO . O
      if (grp == 0) rowc[64 * H + 64 * h + 16 * w + i] = dl;
#pragma unroll
      for (int k = 0; k < 16; ++k) dp[k] = pf[k] * (dp[k] - dl);           // dS
      f32x4 qa[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) qa[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      accum_tile_h<NCH>(bufK, dp, qa, i, grp, NCH * h);                    // dy reused: dq
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) dy[4 * (NCH * h + c) + reg] = qa[c][reg] * qscale;
    }
  }
  __syncthreads();   // Q, dO rows and delta of all tokens visible; K and V no longer read (one head)
  STAMP(10);
  float dk[16], dv[16];
  if (H == 1) {
    // pass B (own keys), one head: P and dS cross from the query owners to the key owners through the
    // K / V tiles, which are dead now -- rows = queries as written, each lane gathers its key's column.
    // 32 LDS reads instead of 128 MFMAs + 16 exp2 per wave to recompute S and dP.
    store_rowT(pT, &bufK[(16 * w + i) * LD], grp);
    store_rowT(dsT, &bufV[(16 * w + i) * LD], grp);
    __syncthreads();
    float sp[16], dp[16];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        sp[4 * tt + reg] = bufK[(16 * tt + 4 * grp + reg) * LD + 16 * w + i];
        dp[4 * tt + reg] = bufV[(16 * tt + 4 * grp + reg) * LD + 16 * w + i];
      }
    zero4(acc); accum_tile(bufDO, sp, acc, i, grp); unpack(acc, dv);    // dV = P^T dO
    zero4(acc); accum_tile(bufQ, dp, acc, i, grp); unpack(acc, dk);     // dK = dS^T Q
#pragma unroll
    for (int k = 0; k < 16; ++k) dk[k] *= qscale;
  }
  // pass B (own keys), several heads: S, dP recomputed with the key on the lane; dV, dK
#pragma unroll
  for (int h = 0; h < (H == 1 ? 0 : H); ++h) {
    float sp[16], dp[16];
    {
      float kf[16], vf[16];   // only this head's column groups are loaded (and read)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ct = NCH * h + c;
        float4 kv = *reinterpret_cast<const float4*>(&bufK[(16 * w + i) * LD + 16 * ct + 4 * grp]);
        float4 vv = *reinterpret_cast<const float4*>(&bufV[(16 * w + i) * LD + 16 * ct + 4 * grp]);
        kf[4 * ct] = kv.x; kf[4 * ct + 1] = kv.y; kf[4 * ct + 2] = kv.z; kf[4 * ct + 3] = kv.w;
        vf[4 * ct] = vv.x; vf[4 * ct + 1] = vv.y; vf[4 * ct + 2] = vv.z; vf[4 * ct + 3] = vv.w;
      }
      zero4(acc); gemm_tile_h<NCH>(bufQ, kf, acc, i, grp, NCH * h); unpack(acc, sp);      // S[q][key]
      zero4(acc); gemm_tile_h<NCH>(bufDO, vf, acc, i, grp, NCH * h); unpack(acc, dp);     // dP[q][key]
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      float4 l4 = *reinterpret_cast<const float4*>(&rowc[64 * h + 16 * tt + 4 * grp]);
      float4 d4 = *reinterpret_cast<const float4*>(&rowc[64 * H + 64 * h + 16 * tt + 4 * grp]);
      float pv;
      pv = __builtin_amdgcn_exp2f(sp[4 * tt + 0] * sc - l4.x); sp[4 * tt + 0] = pv; dp[4 * tt + 0] = pv * (dp[4 * tt + 0] - d4.x);
      pv = __builtin_amdgcn_exp2f(sp[4 * tt + 1] * sc - l4.y); sp[4 * tt + 1] = pv; dp[4 * tt + 1] = pv * (dp[4 * tt + 1] - d4.y);
      pv = __builtin_amdgcn_exp2f(sp[4 * tt + 2] * sc - l4.z); sp[4 * tt + 2] = pv; dp[4 * tt + 2] = pv * (dp[4 * tt + 2] - d4.z);
      pv = __builtin_amdgcn_exp2f(sp[4 * tt + 3] * sc - l4.w); sp[4 * tt + 3] = pv; dp[4 * tt + 3] = pv * (dp[4 * tt + 3] - d4.w);
    }
    f32x4 va[NCH], ka[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { va[c] = (f32x4){0.f, 0.f, 0.f, 0.f}; ka[c] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    accum_tile_h<NCH>(bufDO, sp, va, i, grp, NCH * h);    // dV = P^T dO
    accum_tile_h<NCH>(bufQ, dp, ka, i, grp, NCH * h);     // dK = dS^T Q
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        dv[4 * (NCH * h + c) + reg] = va[c][reg];
        dk[4 * (NCH * h + c) + reg] = ka[c][reg] * qscale;
      }
  }
  if (MODE == MODE_STORE) {
    store_rowT(dy, p.S_dqkv + srow_g * 3 * D, grp);
    store_rowT(dk, p.S_dqkv + srow_g * 3 * D + D, grp);
    store_rowT(dv, p.S_dqkv + srow_g * 3 * D + 2 * D, grp);
  }
  __syncthreads();     // Q / dO no longer read: the buffers go back to the weight pipeline
  STAMP(11);
  {  // dxn = [dq | dk | dv] Wqkv (3 K-chunks of Wqkv^T), dx = r + dy + LN1^T(dxn)
    zero4(acc);
    PUBLISH(1, t); PREFETCH(WqkvT + 64, 3 * D);     // tile 21; -> tile 22
    gemm_tile(t, dy, acc, i, grp);
    PUBLISH(0, t); PREFETCH(WqkvT + 128, 3 * D);    // tile 22; -> tile 23
    gemm_tile(t, dk, acc, i, grp);
    PUBLISH(1, t);                                     // tile 23
    gemm_tile(t, dv, acc, i, grp);
    STAMP(12);
    float dxn[16], gm[16], bt[16], xh[17], xv[16], tmp[16];
    unpack(acc, dxn);
    load_rowT(xv, MODE == MODE_LAZY ? p.Hn1 + voff : xrow, D, grp);          // LN1's normalised input is recomputed from x (LAZY: the row this lane stored on the way in)
    load_rowT(gm, pb + PB_LN1G, D, grp);
#pragma unroll
    for (int k = 0; k < 16; ++k) bt[k] = 0.f;
    ln_rowT(xv, gm, bt, xh, tmp);
    ln_bwd_rowT(dxn, gm, xh, dk);
    if (MODE == MODE_STORE) {
      colsum_store<4>(dxn, cpart + CP_LN1B, i, grp);
#pragma unroll
      for (int k = 0; k < 16; ++k) dxn[k] *= xh[k];
      colsum_store<4>(dxn, cpart + CP_LN1G, i, grp);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) dk[k] += r[k];        // r holds r + dy
    store_rowT(dk, p.BP1 + (long long)g * nl + rowoff, grp);
    STAMP(13);
  }
}

#undef PREFETCH
#undef PUBLISH


// ---- experimental small-batch form of the inference step: one (block, image) on EIGHT waves -----------------
// With fewer than ~2 x 256 workgroups the 4-wave kernel leaves one wave per SIMD and every MFMA -> LDS -> VALU
// dependency is exposed (B = 25 per GPU is what an 8-GPU run of the reference's batch of 200 hands each rank).
// This form puts two waves on each SIMD inside one workgroup.  It did NOT pay (see fused_pair_wanted below).
// Here wave W = (row group w = W & 3, half hf = W >> 2): both halves hold the same 16 token rows, and every GEMM is
// split between them the way tensor parallelism splits a transformer block:
//   column-parallel (input whole, each half produces ITS 32 output columns): Wq|Wk|Wv, W1, W2^T, Wo^T
//   row-parallel    (each half contracts over ITS 32 input columns, partial sums added):  Wo, W2, W1^T, Wqkv^T
// so a column-parallel product feeds a row-parallel one with no exchange (q/k/v -> attention of the half's own
// heads -> Wo; W1 -> gelu -> W2; W2^T -> gelu' -> W1^T; Wo^T -> attention backward -> Wqkv^T) and the halves meet
// only at the five row-wide points (y, z, dzn, dxn; with one head also S and dP, whose 64 columns span both halves):
// PAIR_SUM = both write their 16 partial values per lane to LDS, one barrier, each adds the partner's.  a + b and
// b + a round alike, so both halves carry bit-identical rows through the duplicated LayerNorm / softmax work.
// Weight tiles are staged once per workgroup (512 threads, two float4 each).  MODE_INFER only.
template <int NC>
__device__ __forceinline__ void gemm_tile_hc(const float* __restrict__ t, const float* xf, f32x4* acc, int i, int grp, int ct0) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {   // xf is compact: xf[4 c + reg] belongs to column 16 (ct0 + c) + 4 grp + reg
    const int ct = ct0 + c;
    float4 a[4];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) a[ot] = *reinterpret_cast<const float4*>(&t[(16 * ot + i) * LD + 16 * ct + 4 * grp]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = mfma16(a[ot].x, xf[4 * c + 0], acc[ot]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = mfma16(a[ot].y, xf[4 * c + 1], acc[ot]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = mfma16(a[ot].z, xf[4 * c + 2], acc[ot]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = mfma16(a[ot].w, xf[4 * c + 3], acc[ot]);
  }
}
// the half's 32 output columns of a 64x64 weight tile, whole 64-wide input
__device__ __forceinline__ void gemm_cols_half(const float* __restrict__ t, const float* xf, float* out8, int i, int grp, int hf) {
  f32x4 a2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  gemm_tile_g<2, 4, LD>(t + 32 * hf * LD, xf, a2, i, grp);
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) out8[4 * c + reg] = a2[c][reg];
}
__device__ __forceinline__ void load_half(float* f8, const float* base, int grp, int hf) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    float4 v = *reinterpret_cast<const float4*>(base + 16 * (2 * hf + c) + 4 * grp);
    f8[4 * c] = v.x; f8[4 * c + 1] = v.y; f8[4 * c + 2] = v.z; f8[4 * c + 3] = v.w;
  }
}
__device__ __forceinline__ void store_half(const float* f8, float* base, int grp, int hf) {
#pragma unroll
  for (int c = 0; c < 2; ++c)
    *reinterpret_cast<float4*>(base + 16 * (2 * hf + c) + 4 * grp) = make_float4(f8[4 * c], f8[4 * c + 1], f8[4 * c + 2], f8[4 * c + 3]);
}
// the half's own 8 entries of a whole row held in registers (wave-uniform select, no dynamic register index)
__device__ __forceinline__ void pick_half(const float* f16, float* f8, int hf) {
#pragma unroll
  for (int k = 0; k < 8; ++k) f8[k] = hf ? f16[8 + k] : f16[k];
}

template <int H>
__device__ __forceinline__ void block_body_pair(const FusedP& p, float* lds, int g, int b) {
  constexpr int D = 64, M = 256;
  constexpr int HH = (H == 1) ? 1 : H / 2;   // heads a half works on (one head: both halves, 32 of its 64 columns each)
  constexpr int NCL = 2 / HH;                // 16-column groups of one head inside a half
  constexpr float qscale = (H == 1) ? 0.125f : (H == 2 ? 0.17677669529663687f : 0.25f);
  float* bufK = lds;
  float* bufV = lds + TILE;
  float* wbuf = lds + 2 * TILE;          // 2 tiles; Q and dO of the attention backward alias them
  float* bufQ = lds + 2 * TILE;
  float* bufDO = lds + 3 * TILE;
  float* xch = lds + 4 * TILE;           // 2 tiles: the halves' partial rows for PAIR_SUM
  float* rowc = lds + 6 * TILE;          // lse2[H][64], delta[H][64]
  float* pb = rowc + 512;                // [b1 256 | ln2 gamma 64 | ln2 beta 64 | b2 64 | ln1 gamma 64]
  constexpr int PB_LN2G = 256, PB_LN2B = 320, PB_B2 = 384, PB_LN1G = 448;

  const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) & 3, hf = tid >> 8;
  const int i = lane & 15, grp = lane >> 4;
  const int ct0 = 2 * hf;                // first of the half's two 16-column groups
  const int lrow = 16 * w + i;
  const long long nl = (long long)p.B * 64 * D;
  const long long rowoff = ((long long)b * 64 + lrow) * D;
  const long long voff = (long long)g * nl + rowoff;
  const float* xrow = p.H1 + voff;
  const float* hnext = p.H1 + voff + nl;

  const float* Wqkv = p.W + p.oWqkv + (long long)g * 3 * D * D;
  const float* Wo = p.W + p.oWo + (long long)g * D * D;
  const float* W1 = p.W + p.oW1 + (long long)g * M * D;
  const float* W2 = p.W + p.oW2 + (long long)g * D * M;
  const float* W2T = p.WT + p.tW2 + (long long)g * M * D;
  const float* W1T = p.WT + p.tW1 + (long long)g * D * M;
  const float* WoT = p.WT + p.tWo + (long long)g * D * D;
  const float* WqkvT = p.WT + p.tWqkv + (long long)g * D * 3 * D;
  const float* ln1g = p.W + p.oLn1g + (long long)g * D;
  const float* ln1b = p.W + p.oLn1b + (long long)g * D;
  if (tid < 128) {
    const float* src = tid < 64 ? p.W + p.ob1 + (long long)g * M + 4 * tid
                     : tid < 80 ? p.W + p.oLn2g + (long long)g * D + 4 * (tid - 64)
                     : tid < 96 ? p.W + p.oLn2b + (long long)g * D + 4 * (tid - 80)
                     : tid < 112 ? p.W + p.ob2 + (long long)g * D + 4 * (tid - 96)
                                 : ln1g + 4 * (tid - 112);
    *reinterpret_cast<float4*>(pb + 4 * tid) = *reinterpret_cast<const float4*>(src);
  }

  // weight tiles: thread t stages float4 #t and #t+512 of the 64x64 tile (rows t >> 4 and 32 + (t >> 4))
  float4 sr0, sr1;
  const int srow = tid >> 4, scol = 4 * (tid & 15);
#define PREFETCH2(SRC, LDW)                                                                       \
  do {                                                                                            \
    const float* s_ = (SRC) + (long long)srow * (LDW) + scol;                                     \
    sr0 = *reinterpret_cast<const float4*>(s_);                                                   \
    sr1 = *reinterpret_cast<const float4*>(s_ + 32LL * (LDW));                                    \
  } while (0)
#define PUBLISH2(S, TPTR)                                                                         \
  do {                                                                                            \
    float* d_ = wbuf + S * TILE + srow * LD + scol;                                               \
    *reinterpret_cast<float4*>(d_) = sr0;                                                         \
    *reinterpret_cast<float4*>(d_ + 32 * LD) = sr1;                                               \
    __syncthreads();                                                                              \
    TPTR = wbuf + S * TILE;                                                                       \
  } while (0)
  // V (16 partial values per lane, the same row in both halves) <- own + partner's.  The exchange tile of a half is
  // rewritten only after a later barrier (every pair of PAIR_SUMs has a weight-tile barrier in between).
#define PAIR_SUM(V)                                                                               \
  do {                                                                                            \
    store_rowT(V, xch + hf * TILE + lrow * LD, grp);                                              \
    __syncthreads();                                                                              \
    float o_[16];                                                                                 \
    load_rowT(o_, xch + (hf ^ 1) * TILE + lrow * LD, D, grp);                                     \
    _Pragma("unroll") for (int k_ = 0; k_ < 16; ++k_) V[k_] += o_[k_];                            \
  } while (0)

  PREFETCH2(Wqkv, D);
  const float* t;
  f32x4 acc[4];
  const float sc = qscale * CORA_LOG2E;

  // ---- forward ---------------------------------------------------------------------------------------
  float x[16], y[16];
  float q8[8];                   // own columns of q, kept for the attention backward
  float a_keep[32];              // gelu'(a) of the half's 4 x 32 hidden units
  float pkeep[16 * HH];          // P of the half's head(s): registers, not the stash (half the gelu' leaves the room)
  load_rowT(x, xrow, D, grp);
  {
    float xin[16];
    {
      float gm[16], bt[16], xh[17];
      load_rowT(gm, ln1g, D, grp);
      load_rowT(bt, ln1b, D, grp);
      ln_rowT(x, gm, bt, xh, xin);
    }
    float o8[8];
    {
      float k8[8];
      PUBLISH2(0, t); PREFETCH2(Wqkv + 64 * D, D);
      gemm_cols_half(t, xin, q8, i, grp, hf);
      PUBLISH2(1, t); PREFETCH2(Wqkv + 128 * D, D);
      gemm_cols_half(t, xin, k8, i, grp, hf);
      store_half(k8, &bufK[lrow * LD], grp, hf);
      PUBLISH2(0, t); PREFETCH2(Wo, D);
      gemm_cols_half(t, xin, k8, i, grp, hf);
      store_half(k8, &bufV[lrow * LD], grp, hf);
    }
    __syncthreads();   // K, V of all 64 tokens visible
#pragma unroll
    for (int hh = 0; hh < HH; ++hh) {
      float pf[16];
      zero4(acc); gemm_tile_hc<NCL>(bufK, q8 + 4 * NCL * hh, acc, i, grp, ct0 + NCL * hh); unpack(acc, pf);
      if (H == 1) PAIR_SUM(pf);
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
      for (int k = 0; k < 16; ++k) { pf[k] *= inv; pkeep[16 * hh + k] = pf[k]; }
      if (H > 1 && grp == 0) rowc[64 * (HH * hf + hh) + lrow] = mx + __builtin_amdgcn_logf(ls);
      f32x4 oa[NCL];
#pragma unroll
      for (int c = 0; c < NCL; ++c) oa[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      accum_tile_h<NCL>(bufV, pf, oa, i, grp, ct0 + NCL * hh);
#pragma unroll
      for (int c = 0; c < NCL; ++c)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) o8[4 * (NCL * hh + c) + reg] = oa[c][reg];
    }
    PUBLISH2(1, t); PREFETCH2(W1, D);   // Wo
    zero4(acc); gemm_tile_hc<2>(t, o8, acc, i, grp, ct0); unpack(acc, y);
    PAIR_SUM(y);
#pragma unroll
    for (int k = 0; k < 16; ++k) y[k] += x[k];
  }
  float hn[16];
  {
    float zn[16];
    {
      float gm[16], bt[16], xh[17];
      load_rowT(gm, pb + PB_LN2G, D, grp);
      load_rowT(bt, pb + PB_LN2B, D, grp);
      ln_rowT(y, gm, bt, xh, zn);
    }
    load_rowT(hn, hnext, D, grp);
    f32x4 zacc[4];
    zero4(zacc);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      PUBLISH2(0, t);                                        // W1[mt]
      PREFETCH2(W2 + mt * 64, M);
      float am[8];
      {
        float b1[8];
        load_half(b1, pb + 64 * mt, grp, hf);
        gemm_cols_half(t, zn, am, i, grp, hf);
#pragma unroll
        for (int k = 0; k < 8; ++k) am[k] += b1[k];
      }
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        f32x2 g2, gp2;
        cora_gelu_both2((f32x2){am[k], am[k + 1]}, g2, gp2);
        am[k] = g2.x; am[k + 1] = g2.y;
        a_keep[8 * mt + k] = gp2.x; a_keep[8 * mt + k + 1] = gp2.y;
      }
      PUBLISH2(1, t);                                        // W2[:, mt]
      if (mt < 3) PREFETCH2(W1 + (mt + 1) * 64 * D, D);
      else PREFETCH2(W2T, D);
      gemm_tile_hc<2>(t, am, zacc, i, grp, ct0);
    }
    float z[16], b2[16];
    unpack(zacc, z);
    PAIR_SUM(z);
    load_rowT(b2, pb + PB_B2, D, grp);
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = x[k] + y[k] + z[k] + b2[k];     // u
  }
  float r[16];
  {
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = hn[k] - x[k];
    float r8[8];
    pick_half(r, r8, hf);
    store_half(r8, p.E2 + (long long)g * nl + rowoff, grp, hf);
  }

  // ---- backward (cotangent r) -----------------------------------------------------------------------
  float dy[16];
  {
    f32x4 dacc[4];
    zero4(dacc);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      PUBLISH2(0, t);                                        // W2^T[mt]
      PREFETCH2(W1T + mt * 64, M);
      float dg[8];
      gemm_cols_half(t, r, dg, i, grp, hf);
#pragma unroll
      for (int k = 0; k < 8; ++k) dg[k] *= a_keep[8 * mt + k];
      PUBLISH2(1, t);                                        // W1^T[:, mt]
      if (mt < 3) PREFETCH2(W2T + (mt + 1) * 64 * D, D); else PREFETCH2(WoT, D);
      gemm_tile_hc<2>(t, dg, dacc, i, grp, ct0);
    }
    float dzn[16], gm[16], bt[16], xh[17], tmp[16];
    unpack(dacc, dzn);
    PAIR_SUM(dzn);
    load_rowT(gm, pb + PB_LN2G, D, grp);
#pragma unroll
    for (int k = 0; k < 16; ++k) bt[k] = 0.f;
    ln_rowT(y, gm, bt, xh, tmp);
    ln_bwd_rowT(dzn, gm, xh, dy);
#pragma unroll
    for (int k = 0; k < 16; ++k) dy[k] += r[k];
  }
  float dq8[8], dk8[8], dv8[8];
  float dsk[H == 1 ? 16 : 1];
  {
    float dof[8];
    PUBLISH2(0, t); PREFETCH2(WqkvT, 3 * D);         // Wo^T; the next tile waits in registers through the attention backward
    gemm_cols_half(t, dy, dof, i, grp, hf);
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] += dy[k];
    __syncthreads();   // every wave is done with the weight buffers: reuse them for Q and dO
    store_half(dof, &bufDO[lrow * LD], grp, hf);
    store_half(q8, &bufQ[lrow * LD], grp, hf);
    // pass A (own queries): dP, delta, dS, dQ for the half's head(s)
#pragma unroll
    for (int hh = 0; hh < HH; ++hh) {
      float dp[16];
      const float* pf = pkeep + 16 * hh;
      zero4(acc); gemm_tile_hc<NCL>(bufV, dof + 4 * NCL * hh, acc, i, grp, ct0 + NCL * hh); unpack(acc, dp);
      if (H == 1) PAIR_SUM(dp);
      float dl = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) dl += pf[k] * dp[k];
      dl = grp_sum(dl);
      if (H > 1 && grp == 0) rowc[64 * H + 64 * (HH * hf + hh) + lrow] = dl;
#pragma unroll
      for (int k = 0; k < 16; ++k) dp[k] = pf[k] * (dp[k] - dl);
      f32x4 qa[NCL];
#pragma unroll
      for (int c = 0; c < NCL; ++c) qa[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      accum_tile_h<NCL>(bufK, dp, qa, i, grp, ct0 + NCL * hh);
#pragma unroll
      for (int c = 0; c < NCL; ++c)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) dq8[4 * (NCL * hh + c) + reg] = qa[c][reg] * qscale;
      if (H == 1) {
#pragma unroll
        for (int k = 0; k < 16; ++k) dsk[k] = dp[k];
      }
    }
  }
  __syncthreads();   // Q, dO rows and delta of all tokens visible; K and V no longer read (one head)
  if (H == 1) {
    // pass B, one head: P (written by half 0) and dS (half 1) cross from the query owners to the key owners through
    // the dead K / V tiles; each half then produces its 32 columns of dV = P^T dO and dK = dS^T Q
    if (hf == 0) store_rowT(pkeep, &bufK[lrow * LD], grp);
    else store_rowT(dsk, &bufV[lrow * LD], grp);
    __syncthreads();
    float sp[16], dp[16];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        sp[4 * tt + reg] = bufK[(16 * tt + 4 * grp + reg) * LD + lrow];
        dp[4 * tt + reg] = bufV[(16 * tt + 4 * grp + reg) * LD + lrow];
      }
    f32x4 va[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, ka[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    accum_tile_h<2>(bufDO, sp, va, i, grp, ct0);
    accum_tile_h<2>(bufQ, dp, ka, i, grp, ct0);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) { dv8[4 * c + reg] = va[c][reg]; dk8[4 * c + reg] = ka[c][reg] * qscale; }
  }
#pragma unroll
  for (int hh = 0; hh < (H == 1 ? 0 : HH); ++hh) {
    // pass B, several heads: S, dP of the half's head recomputed with the key on the lane; dV, dK
    const int hg = HH * hf + hh, ch = ct0 + NCL * hh;
    float sp[16], dp[16];
    {
      float kf[4 * NCL], vf[4 * NCL];
#pragma unroll
      for (int c = 0; c < NCL; ++c) {
        float4 kv = *reinterpret_cast<const float4*>(&bufK[lrow * LD + 16 * (ch + c) + 4 * grp]);
        float4 vv = *reinterpret_cast<const float4*>(&bufV[lrow * LD + 16 * (ch + c) + 4 * grp]);
        kf[4 * c] = kv.x; kf[4 * c + 1] = kv.y; kf[4 * c + 2] = kv.z; kf[4 * c + 3] = kv.w;
        vf[4 * c] = vv.x; vf[4 * c + 1] = vv.y; vf[4 * c + 2] = vv.z; vf[4 * c + 3] = vv.w;
      }
      zero4(acc); gemm_tile_hc<NCL>(bufQ, kf, acc, i, grp, ch); unpack(acc, sp);      // S[q][key]
      zero4(acc); gemm_tile_hc<NCL>(bufDO, vf, acc, i, grp, ch); unpack(acc, dp);     // dP[q][key]
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      float4 l4 = *reinterpret_cast<const float4*>(&rowc[64 * hg + 16 * tt + 4 * grp]);
      float4 d4 = *reinterpret_cast<const float4*>(&rowc[64 * H + 64 * hg + 16 * tt + 4 * grp]);
      float pv;
      pv = __builtin_amdgcn_exp2f(sp[4 * tt + 0] * sc - l4.x); sp[4 * tt + 0] = pv; dp[4 * tt + 0] = pv * (dp[4 * tt + 0] - d4.x);
      pv = __builtin_amdgcn_exp2f(sp[4 * tt + 1] * sc - l4.y); sp[4 * tt + 1] = pv; dp[4 * tt + 1] = pv * (dp[4 * tt + 1] - d4.y);
      pv = __builtin_amdgcn_exp2f(sp[4 * tt + 2] * sc - l4.z); sp[4 * tt + 2] = pv; dp[4 * tt + 2] = pv * (dp[4 * tt + 2] - d4.z);
      pv = __builtin_amdgcn_exp2f(sp[4 * tt + 3] * sc - l4.w); sp[4 * tt + 3] = pv; dp[4 * tt + 3] = pv * (dp[4 * tt + 3] - d4.w);
    }
    f32x4 va[NCL], ka[NCL];
#pragma unroll
    for (int c = 0; c < NCL; ++c) { va[c] = (f32x4){0.f, 0.f, 0.f, 0.f}; ka[c] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    accum_tile_h<NCL>(bufDO, sp, va, i, grp, ch);
    accum_tile_h<NCL>(bufQ, dp, ka, i, grp, ch);
#pragma unroll
    for (int c = 0; c < NCL; ++c)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        dv8[4 * (NCL * hh + c) + reg] = va[c][reg];
        dk8[4 * (NCL * hh + c) + reg] = ka[c][reg] * qscale;
      }
  }
  __syncthreads();     // Q / dO no longer read: the buffers go back to the weight pipeline
  {  // dxn = [dq | dk | dv] Wqkv (each half its own columns of the three K-chunks), dx = r + dy + LN1^T(dxn)
    zero4(acc);
    PUBLISH2(1, t); PREFETCH2(WqkvT + 64, 3 * D);
    gemm_tile_hc<2>(t, dq8, acc, i, grp, ct0);
    PUBLISH2(0, t); PREFETCH2(WqkvT + 128, 3 * D);
    gemm_tile_hc<2>(t, dk8, acc, i, grp, ct0);
    PUBLISH2(1, t);
    gemm_tile_hc<2>(t, dv8, acc, i, grp, ct0);
    float dxn[16], gm[16], bt[16], xh[17], xv[16], tmp[16], dx[16];
    unpack(acc, dxn);
    PAIR_SUM(dxn);
    load_rowT(xv, xrow, D, grp);
    load_rowT(gm, pb + PB_LN1G, D, grp);
#pragma unroll
    for (int k = 0; k < 16; ++k) bt[k] = 0.f;
    ln_rowT(xv, gm, bt, xh, tmp);
    ln_bwd_rowT(dxn, gm, xh, dx);
#pragma unroll
    for (int k = 0; k < 16; ++k) dx[k] += r[k];        // r holds r + dy
    float dx8[8];
    pick_half(dx, dx8, hf);
    store_half(dx8, p.BP1 + (long long)g * nl + rowoff, grp, hf);
  }
}
#undef PREFETCH2
#undef PUBLISH2
#undef PAIR_SUM

template <int H>
__global__ __launch_bounds__(512) void pc_block_pair_kernel(FusedP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x >= p.L * p.B) {   // edge layers of one image: the first four waves, as in the 4-wave kernel
    if (threadIdx.x >= 256) return;
    edge_path<false>(p, lds, (int)blockIdx.x - p.L * p.B);
    return;
  }
  int j = (int)blockIdx.x;
  const int n = p.L * p.B;
  if ((n & 7) == 0) j = (j & 7) * (n >> 3) + (j >> 3);
  block_body_pair<H>(p, lds, j / p.B, j % p.B);
}

template <int MODE, int H>
__global__ __launch_bounds__(256, 2) void pc_block_fused_kernel(FusedP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x >= p.L * p.B) {   // the last B workgroups do the two edge layers of one image each
    edge_path<MODE == MODE_LAZY>(p, lds, (int)blockIdx.x - p.L * p.B);
    return;
  }
  // XCD-aware order: the hardware deals workgroups round-robin over the 8 XCDs (id % 8), each with its
  // own 4 MB L2.  Give XCD x the contiguous slice [x n/8, (x+1) n/8) of the (block, image) list so it
  // touches the weights of at most two blocks instead of all L, and the L2 keeps the stash instead.
  int j = (int)blockIdx.x;
  const int n = p.L * p.B;
  if ((n & 7) == 0) j = (j & 7) * (n >> 3) + (j >> 3);
  block_body<MODE, H>(p, lds, j / p.B, j % p.B, nullptr);
}

// Status "forward" chain (rule 'h -> u', /root/reference/src/decoder_transformer.py:167-168,175-176,
// 488-489): u_1 from h_0, every block reads the previous block's prediction; one workgroup walks
// one image through all L blocks.  store_h: also write u_l into h_l (INIT rule 'h, u <- u',
// /root/reference/pcx/predictive_coding/_vode.py:201).  Writes z = u_{L+1} Wout^T + bout to p.Z and
// accumulates the clipped reconstruction error (:491-494) into p.mse.
template <int H>
__global__ __launch_bounds__(256, 1) void pc_chain_kernel(FusedP p, int store_h) {
  constexpr int D = 64, P = 48, LP = 52;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, grp = lane >> 4;
  const int b = blockIdx.x;
  const long long tok = (long long)b * 64 + 16 * w + i;
  const long long nl = (long long)p.B * 64 * D;
  float x[16];
  f32x4 acc[4];
  {
    copy_to_lds(lds, p.W + p.oWp, 64, P, LP, tid);
    float h0[12], bp[16], ps[16];
    load_rowT_n<3>(h0, p.H0 + tok * P, grp);
    load_rowT_n<4>(bp, p.W + p.obp, grp);
    load_rowT_n<4>(ps, p.pos + (long long)(16 * w + i) * D, grp);
    __syncthreads();
    zero4(acc); gemm_tile_g<4, 3, LP>(lds, h0, acc, i, grp); unpack(acc, x);
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] += bp[k] + ps[k];
    if (store_h) store_rowT(x, const_cast<float*>(p.H1) + tok * D, grp);
  }
  for (int g = 0; g < p.L; ++g) {
    __syncthreads();   // LDS of the previous stage is no longer read
    block_body<MODE_CHAIN, H>(p, lds, g, b, x);
    if (store_h) store_rowT(x, const_cast<float*>(p.H1) + (long long)(g + 1) * nl + tok * D, grp);
  }
  __syncthreads();
  copy_to_lds(lds, p.W + p.oWout, P, D, LD, tid);
  float bo[12], xp[12];
  load_rowT_n<3>(bo, p.W + p.obout, grp);
  load_rowT_n<3>(xp, p.XP + tok * P, grp);
  __syncthreads();
  zero4(acc); gemm_tile_g<3, 4, LD>(lds, x, acc, i, grp);
  float z[12];
  double se = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = acc[k][j] + bo[4 * k + j];
      z[4 * k + j] = v;
      float dd = fminf(fmaxf(v, 0.f), 1.f) - fminf(fmaxf(xp[4 * k + j], 0.f), 1.f);
      se += (double)dd * dd;
    }
  store_rowT_n<3>(z, p.Z + tok * P, grp);
  if (p.mse) {
    se = wave_sum_d(se);
    if (lane == 0) atomicAdd(p.mse, se);
  }
}


size_t fused_block_lds_bytes() { return (size_t)(4 * TILE + 128 * 4 + 512) * sizeof(float); }   // rowc sized for 4 heads + parameter block
static size_t fused_pair_lds_bytes() { return (size_t)(6 * TILE + 128 * 4 + 512) * sizeof(float); }

// Opt-in (CORA_PAIR=1), off by default: measured on MI355X (profiles/r3_pair_sweep.txt) the 8-wave form gains 1.7 % at
// B = 25 (6 blocks: 150 workgroups, one per CU either way) and loses 7-13 % from B = 42 up.  The two halves of a row
// group sit on the same SIMD in lockstep -- same barriers, MFMA and VALU phases at the same time -- so they hide
// little of each other's latency, and the row-wide work they both carry (LayerNorm, one-head softmax, the exchanges)
// is paid twice on that SIMD.  Kept, with its parity tests, as the reproducible record of that experiment.
bool fused_pair_wanted(int L, int B) {
  (void)L; (void)B;
  const char* e = getenv("CORA_PAIR");   // read per launch (launches are captured into graphs, so this is off the hot path)
  return e && *e && atoi(e) != 0;
}

template <int H>
static int launch_fused_h(const FusedP& p, int L, int store, hipStream_t st) {
  static bool attr[64] = {};
  size_t sm = fused_block_lds_bytes();
  if (first_use_on_device(attr)) {
    if (hipFuncSetAttribute((const void*)pc_block_fused_kernel<MODE_INFER, H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void*)pc_block_fused_kernel<MODE_STORE, H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void*)pc_block_fused_kernel<MODE_LAZY, H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void*)pc_chain_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void*)pc_block_pair_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_pair_lds_bytes()) != hipSuccess) return -1;
  }
  if (L < 0) return 0;   // attribute set-up only
  if (!store && p.lazy) {
    hipLaunchKernelGGL((pc_block_fused_kernel<MODE_LAZY, H>), dim3((L + 1) * p.B), dim3(256), sm, st, p);
    return 0;
  }
  if (!store && fused_pair_wanted(L, p.B)) {
    hipLaunchKernelGGL(pc_block_pair_kernel<H>, dim3((L + 1) * p.B), dim3(512), fused_pair_lds_bytes(), st, p);
    return 0;
  }
  if (store == 2) hipLaunchKernelGGL(pc_chain_kernel<H>, dim3(p.B), dim3(256), sm, st, p, L /* store_h */);
  else if (store) hipLaunchKernelGGL((pc_block_fused_kernel<MODE_STORE, H>), dim3((L + 1) * p.B), dim3(256), sm, st, p);
  else hipLaunchKernelGGL((pc_block_fused_kernel<MODE_INFER, H>), dim3((L + 1) * p.B), dim3(256), sm, st, p);
  return 0;
}

int launch_fused_block(const FusedP& p, int L, int store, hipStream_t st) {
  switch (p.H) {
    case 1: return launch_fused_h<1>(p, L, store ? 1 : 0, st);
    case 2: return launch_fused_h<2>(p, L, store ? 1 : 0, st);
    case 4: return launch_fused_h<4>(p, L, store ? 1 : 0, st);
  }
  return -1;
}

int launch_chain(const FusedP& p, int store_h, hipStream_t st) {
  int rc = -1;
  switch (p.H) {   // attributes first (a chain launch may be the first use of this head count)
    case 1: rc = launch_fused_h<1>(p, -1, 0, st); if (!rc) rc = launch_fused_h<1>(p, store_h ? 1 : 0, 2, st); break;
    case 2: rc = launch_fused_h<2>(p, -1, 0, st); if (!rc) rc = launch_fused_h<2>(p, store_h ? 1 : 0, 2, st); break;
    case 4: rc = launch_fused_h<4>(p, -1, 0, st); if (!rc) rc = launch_fused_h<4>(p, store_h ? 1 : 0, 2, st); break;
  }
  return rc;
}

// ---- transposed weight copies: dst[c][r] = src[r][c] per group; up to 6 matrices per launch -----------
__global__ void transpose_kernel(TransposeSet t) {
  __shared__ float tile[32][33];
  int z = blockIdx.z, q = 0;
#pragma unroll
  for (int k = 1; k < 6; ++k) if (k < t.n && z >= t.zstart[k]) q = k;
  z -= t.zstart[q];
  const int rows = t.rows[q], cols = t.cols[q];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  if (c0 >= cols || r0 >= rows) return;      // the grid is sized for the largest matrix of the set
  const float* s = t.src[q] + (long long)z * rows * cols;
  float* d = t.dst[q] + (long long)z * rows * cols;
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
void launch_transpose_set(const TransposeSet& t_in, hipStream_t st) {
  TransposeSet t = t_in;
  int z = 0, mx = 1, my = 1;
  for (int k = 0; k < t.n; ++k) {
    t.zstart[k] = z; z += t.groups[k];
    mx = max(mx, (t.cols[k] + 31) / 32); my = max(my, (t.rows[k] + 31) / 32);
  }
  hipLaunchKernelGGL(transpose_kernel, dim3(mx, my, z), dim3(32, 8), 0, st, t);
}
void launch_transpose(const float* src, float* dst, int rows, int cols, int groups, hipStream_t st) {
  TransposeSet t = {};
  t.n = 1; t.src[0] = src; t.dst[0] = dst; t.rows[0] = rows; t.cols[0] = cols; t.groups[0] = groups;
  launch_transpose_set(t, st);
}

// ---- dW[bias / LayerNorm parameters] -= sum over the 4 B wave slices of colpart (dW zeroed before) ------------
// grid (L + 1, COLPART / 64, row chunks): group L is the edge table (patch / output projection biases)
__global__ __launch_bounds__(256) void colpart_reduce_kernel(FusedP p) {
  __shared__ float sh[4][64];
  const int g = blockIdx.x, c0 = 64 * blockIdx.y;
  const bool edge = (g == p.L);
  if (edge && c0 >= CP_EDGE) return;
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int stride = edge ? CP_EDGE : COLPART;
  const float* src = p.colpart + (edge ? (long long)p.L * p.B * 4 * COLPART : (long long)g * p.B * 4 * COLPART) + c0 + cx;
  const int rows = 4 * p.B, per = (rows + gridDim.z - 1) / gridDim.z;
  const int r0 = blockIdx.z * per, r1 = min(rows, r0 + per);
  float s = 0.f;
  for (int rr = r0 + ry; rr < r1; rr += 4) s += src[(long long)rr * stride];
  sh[ry][cx] = s;
  __syncthreads();
  if (ry == 0) {
    s = sh[0][cx] + sh[1][cx] + sh[2][cx] + sh[3][cx];
    long long off;
    if (edge) {
      if (c0 == 0) off = p.obp; else { if (cx >= 48) return; off = p.obout; }
    }
    else if (c0 < CP_B1) off = p.ob2 + (long long)g * 64;
    else if (c0 < CP_LN2B) off = p.ob1 + (long long)g * 256 + (c0 - CP_B1);
    else if (c0 < CP_LN2G) off = p.oLn2b + (long long)g * 64;
    else if (c0 < CP_LN1B) off = p.oLn2g + (long long)g * 64;
    else if (c0 < CP_LN1G) off = p.oLn1b + (long long)g * 64;
    else off = p.oLn1g + (long long)g * 64;
    atomicAdd(&p.dW[off + cx], -s);
  }
}
void launch_colpart_reduce(const FusedP& p, int L, hipStream_t st) {
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3(L + 1, COLPART / 64, 16), dim3(256), 0, st, p);
}
size_t fused_colpart_floats(int L, int B) { return (size_t)L * B * 4 * COLPART + (size_t)B * 4 * CP_EDGE; }
