// fused_wide.hip -- the fused per-(block, image) kernel for the wide sweep points (N = 64 tokens,
// hidden size D = 96 or 128, any head count whose head dim is a multiple of 4): the block's
// prediction u = 2x + MHA(LN1 x) + MLP(LN2(x + MHA(LN1 x))), the next vode's error e = h_next - u
// and the hand-derived VJP dx = J^T e in ONE launch, one workgroup per (block, image).
//
// Replaces the 13 per-op launches of the general path for config 3 of BASELINE.json
// (/root/reference/sweeps/sweep_12block_comprehensive.yaml:8-47: hidden_size 64/96/128, heads 1/2/4);
// math: /root/reference/pcx/nn/_layer.py:428-447 (block), :359-402 (attention), :296-315 (LayerNorm),
// double residual /root/reference/src/decoder_transformer.py:238, VJP = jax.value_and_grad(energy)
// w.r.t. the vodes (:396,445).
//
// Same data layout "T" as fused_block.hip (lane (r = l & 15, grp = l >> 4) of a wave holds
// X[row r][16 ct + 4 grp + reg]: the B operand of v_mfma_f32_16x16x4_f32 AND what it returns when the
// weight is the A operand), generalised to NCT = D / 16 column groups.  What changes with the width:
//  * a row tensor is 4 NCT = 24 / 32 registers per lane and the MLP's gelu'(a) alone is 16 NCT, so
//    the kernel is built for ONE workgroup per CU (one wave per SIMD, up to 512 VGPRs): y and all of
//    gelu'(a) stay in registers from the forward to the backward -- no recomputation; q and P go
//    through an L2-resident stash, x (an input) is read again where LayerNorm-1's backward needs it;
//  * weights stream through LDS as tiles of D/2 output rows x D inputs (row stride D + 4), double
//    buffered, prefetched global -> registers one tile ahead; every GEMM of the block is a whole number
//    of such tiles (all its dimensions are multiples of D): 48 weight tiles per workgroup;
//  * K / V of all 64 tokens persist in LDS from the forward into the backward (2 x 64 x (D + 4) floats);
//    Q / dO alias the weight double buffer during the attention backward.  D = 128: 141 KB of LDS;
//  * heads are ranges of 4-column quads [q0, q1) of the layout: a head whose width is not a multiple
//    of 16 (D = 96, 4 heads of 24) shares a column group with its neighbour, and the lanes of the other
//    head feed zeros into the contraction (one v_cndmask per MFMA operand).
#include "common.h"

#include <type_traits>

#include "fused_wide_common.h"

// Diagnostic build only (-DCORA_STAMPS): wave 0 of each workgroup records s_memtime at phase boundaries
// into p.dbg[blockIdx.x * 16 + k].  Never compiled into the shipped library.
#if defined(CORA_STAMPS) && CORA_STAMPS == 2     /* zoom: stamps inside stages 2 and 3 only */
#define SSTAMP(J)                                                                                  \
  do {                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    if (p.dbg && (k_ == 2 || k_ == 3)) {                                                           \
      unsigned long long t_;                                                                       \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                 \
      if (threadIdx.x == 0) p.dbg[(long long)blockIdx.x * 16 + 5 * (k_ - 2) + (J)] = t_;           \
    }                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)
#define WSTAMP(K) do {} while (0)
#elif defined(CORA_STAMPS)
#define SSTAMP(J) do {} while (0)
#define WSTAMP(K)                                                                                  \
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
#define WSTAMP(K) do {} while (0)
#define SSTAMP(J) do {} while (0)
#endif

// ---- weight tile pipeline macros (shared by the block body and the chain body) ----------------------------
// They name locals both bodies define: tile_src, the staging registers sr0..sr7, tid / i / grp, wbuf, slot,
// NT (tiles the workgroup walks through), HOLD (a stage that neither publishes nor prefetches, or -1) and the
// WideCfg constants.
#define PF_ONE(J, REG, SRC, LDW)                                                                  \
  if constexpr (NST > (J)) {                                                                      \
    const int f_ = tid + 256 * (J);                                                               \
    if (NF4 % 256 == 0 || f_ < NF4)                                                               \
      REG = *reinterpret_cast<const float4*>((SRC) + (long long)(f_ / (D / 4)) * (LDW) + 4 * (f_ % (D / 4))); \
  }
#define PB_ONE(J, REG, WB)                                                                        \
  if constexpr (NST > (J)) {                                                                      \
    const int f_ = tid + 256 * (J);                                                               \
    if (NF4 % 256 == 0 || f_ < NF4)                                                               \
      lds_st4((WB) + (f_ / (D / 4)) * LD + 4 * (f_ % (D / 4)), REG);                              \
  }
  // whole-tile forms (prologue, and the restart after the attention backward)
#define PREFETCH_ALL(K)                                                                           \
  do {                                                                                            \
    int ld_; const float* s_ = tile_src((K), ld_);                                                \
    PF_ONE(0, sr0, s_, ld_) PF_ONE(1, sr1, s_, ld_) PF_ONE(2, sr2, s_, ld_) PF_ONE(3, sr3, s_, ld_) \
    PF_ONE(4, sr4, s_, ld_) PF_ONE(5, sr5, s_, ld_) PF_ONE(6, sr6, s_, ld_) PF_ONE(7, sr7, s_, ld_) \
  } while (0)
#define WRITE_ALL(S)                                                                              \
  do {                                                                                            \
    lds_float* wb_ = opaque_w(wbuf + (S) * SLOT);                                                 \
    PB_ONE(0, sr0, wb_) PB_ONE(1, sr1, wb_) PB_ONE(2, sr2, wb_) PB_ONE(3, sr3, wb_)               \
    PB_ONE(4, sr4, wb_) PB_ONE(5, sr5, wb_) PB_ONE(6, sr6, wb_) PB_ONE(7, sr7, wb_)               \
  } while (0)
  // column group C of the multiplication with tile k, with its share of the staging traffic
#define GEMM_CT(C, REG, XF, ACC)                                                                  \
  if constexpr (NCT > (C)) {                                                                      \
    if ((C) + 1 < NCT) {                                                                          \
      _Pragma("unroll") for (int ot = 0; ot < NOT; ++ot)                                          \
        a_[((C) + 1) & 1][ot] = lds_ld4(&tl_[16 * ot * LD + 16 * ((C) + 1)]);                     \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);   /* the reads of group C+1 stay AHEAD of group C's MFMAs */ \
    if (do_w_) { PB_ONE(C, REG, wb_) }                                                            \
    if (do_p_) { PF_ONE(C, REG, ns_, nld_) }                                                      \
    _Pragma("unroll") for (int ot = 0; ot < NOT; ++ot) (ACC)[ot] = mfma16(a_[(C) & 1][ot].x, (XF)[4 * (C) + 0], (ACC)[ot]); \
    _Pragma("unroll") for (int ot = 0; ot < NOT; ++ot) (ACC)[ot] = mfma16(a_[(C) & 1][ot].y, (XF)[4 * (C) + 1], (ACC)[ot]); \
    _Pragma("unroll") for (int ot = 0; ot < NOT; ++ot) (ACC)[ot] = mfma16(a_[(C) & 1][ot].z, (XF)[4 * (C) + 2], (ACC)[ot]); \
    _Pragma("unroll") for (int ot = 0; ot < NOT; ++ot) (ACC)[ot] = mfma16(a_[(C) & 1][ot].w, (XF)[4 * (C) + 3], (ACC)[ot]); \
    __builtin_amdgcn_sched_barrier(0);                                                            \
  }
#define STAGE(K, XF, ACC)                                                                         \
  do {                                                                                            \
    const int k_ = (K);                                                                           \
    const bool do_w_ = (k_ + 1 < NT) && (k_ != HOLD);                                               \
    const bool do_p_ = (k_ + 2 < NT) && (k_ != HOLD);                                               \
    int nld_ = D; const float* ns_ = do_p_ ? tile_src(k_ + 2, nld_) : nullptr;                    \
    SSTAMP(0);                                                                                    \
    __syncthreads();                                                                              \
    SSTAMP(1);                                                                                    \
    const lds_float* tl_ = opaque(wbuf + slot * SLOT + i * LD + 4 * grp);                         \
    lds_float* wb_ = opaque_w(wbuf + (slot ^ 1) * SLOT);                                          \
    float4 a_[2][NOT];                                                                            \
    _Pragma("unroll") for (int ot = 0; ot < NOT; ++ot) a_[0][ot] = lds_ld4(&tl_[16 * ot * LD]);   \
    GEMM_CT(0, sr0, XF, ACC) SSTAMP(2); GEMM_CT(1, sr1, XF, ACC) GEMM_CT(2, sr2, XF, ACC) GEMM_CT(3, sr3, XF, ACC) \
    SSTAMP(3); GEMM_CT(4, sr4, XF, ACC) GEMM_CT(5, sr5, XF, ACC) GEMM_CT(6, sr6, XF, ACC) GEMM_CT(7, sr7, XF, ACC) \
    SSTAMP(4);                                                                                    \
    slot ^= 1;                                                                                    \
    (void)wb_; (void)ns_; (void)nld_;                                                             \
  } while (0)


// One transformer block of one image, inference step: x = h_{g+1} from HBM; writes e_{g+2} and J^T e.
template <int NCT, int H>
__device__ __forceinline__ void wide_block_body(const FusedP& p, float* lds, int g, int b) {
  using Cf = WideCfg<NCT>;
  constexpr int D = Cf::D, M = Cf::M, R = Cf::R, NOT = Cf::NOT, TR = Cf::TR, LD = Cf::LD, SLOT = Cf::SLOT;
  constexpr int NST = Cf::NST, NF4 = Cf::NF4;
  constexpr int DH = D / H, QH = DH / 4;        // head dim, quads per head
  static_assert(D % H == 0 && DH % 4 == 0, "head dim must be a multiple of 4");
  const float qscale = rsqrtf((float)DH);
  const float sc = qscale * CORA_LOG2E;

  float* bufK = lds;
  float* bufV = lds + SLOT;
  float* wbuf = lds + 2 * SLOT;          // 2 slots
  float* bufQ = lds + 2 * SLOT;          // aliases wbuf[0] during the attention backward
  float* bufDO = lds + 3 * SLOT;         // aliases wbuf[1]
  float* rowc = lds + 4 * SLOT;          // lse2[H][64], delta[H][64]
  float* pb = lds + 4 * SLOT + 512;      // [b1 M | ln2 gamma D | ln2 beta D | b2 D | ln1 gamma D]
  constexpr int PB_LN2G = M, PB_LN2B = M + D, PB_B2 = M + 2 * D, PB_LN1G = M + 3 * D;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, grp = lane >> 4;
  const long long nl = (long long)p.B * 64 * D;
  const long long rowoff = ((long long)b * 64 + 16 * w + i) * D;
  const long long voff = (long long)g * nl + rowoff;
  const float* xrow = p.H1 + voff;
  const float* hnext = p.H1 + voff + nl;
  const long long srow_g = ((long long)g * p.B + b) * 64 + 16 * w + i;
  float* st_p = p.S_stash + srow_g * (H * 64 + D);   // H x 64: one row of P per head, then q (D)
  float* st_q = st_p + H * 64;

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
    // parameter block into LDS, float4 per thread: visible after the first tile barrier
    constexpr int NP4 = Cf::PB / 4;
    for (int f = tid; f < NP4; f += 256) {
      const int o = 4 * f;
      const float* src = o < M ? p.W + p.ob1 + (long long)g * M + o
                       : o < M + D ? p.W + p.oLn2g + (long long)g * D + (o - M)
                       : o < M + 2 * D ? p.W + p.oLn2b + (long long)g * D + (o - M - D)
                       : o < M + 3 * D ? p.W + p.ob2 + (long long)g * D + (o - M - 2 * D)
                                       : ln1g + (o - M - 3 * D);
      *reinterpret_cast<float4*>(pb + o) = *reinterpret_cast<const float4*>(src);
    }
  }

  // ---- weight tile pipeline ----------------------------------------------------------------------------
  // 48 tiles of TR rows x D floats (tile_src below gives tile k's address and row stride), two LDS slots,
  // one register staging set.  Stage k, between two barriers:
  //     write tile k+1 (staged in registers during stage k-1) into the slot stage k-1 read,
  //     fetch tile k+2 from global into the staging registers,
  //     multiply with tile k
  // and the three are INTERLEAVED: in column group c of the multiplication one staged float4 goes to LDS
  // (ds_write_b128), the same register is refilled from global and the LDS operands of group c+1 are
  // read, then the 4 NOT MFMAs of group c issue.  One wave per SIMD has nothing else to hide the LDS
  // write pass and the loads behind (written before the barrier, tile by tile, the kernel ran at 46 %
  // of the MFMA rate).  float4 #f of a tile = row f / (D/4), 16-byte column f % (D/4); thread t stages
  // #t, t + 256, ...  Named staging registers + macros (an array, even fully unrolled, ends up in scratch).
  float4 sr0, sr1, sr2, sr3, sr4, sr5, sr6, sr7;
  static_assert(NST <= 8 && NST <= NCT, "one staged float4 per column group");
  auto tile_src = [&](int k, int& ldw) -> const float* {
    // forward: 0-5 Wqkv | 6-7 Wo | 8-23 MLP (per hidden chunk mt: W1 half 0, W1 half 1, W2 half 0, W2 half 1)
    // backward: 24-39 MLP (W2^T half 0/1, W1^T half 0/1 per chunk) | 40-41 Wo^T | 42-47 Wqkv^T (half, dq/dk/dv)
    if (k < 6) { ldw = D; return Wqkv + (long long)k * TR * D; }
    if (k < 8) { ldw = D; return Wo + (long long)(k - 6) * TR * D; }
    if (k < 24) {
      const int mt = (k - 8) >> 2, j = (k - 8) & 3;
      if (j < 2) { ldw = D; return W1 + ((long long)mt * D + j * TR) * D; }
      ldw = M; return W2 + (long long)(j - 2) * TR * M + (long long)mt * D;
    }
    if (k < 40) {
      const int mt = (k - 24) >> 2, j = (k - 24) & 3;
      if (j < 2) { ldw = D; return W2T + ((long long)mt * D + j * TR) * D; }
      ldw = M; return W1T + (long long)(j - 2) * TR * M + (long long)mt * D;
    }
    if (k < 42) { ldw = D; return WoT + (long long)(k - 40) * TR * D; }
    const int j = k - 42;
    ldw = 3 * D; return WqkvT + (long long)(j & 1) * TR * 3 * D + (long long)(j >> 1) * D;
  };
  // tiles 41 -> 42 straddle the attention backward, whose Q / dO rows take both slots: stage HOLD = 41
  // writes nothing (tile 42 waits in the staging registers) and the restart publishes it afterwards
  constexpr int NT = 48, HOLD = 41;
  int slot = 0;
  f32x4 acc[NOT];
  PREFETCH_ALL(0);
  WRITE_ALL(0);                    // tile 0 -> slot 0 (visible after stage 0's barrier)
  PREFETCH_ALL(1);
  WSTAMP(0);

  // ---- forward ---------------------------------------------------------------------------------------
  float y[R];
  // gelu'(a) of the MLP pre-activation (M values per row) waits for the backward in an L2-resident stash:
  // kept in registers (16 NCT per lane) the kernel lives beyond the 256 architectural VGPRs and the
  // compiler pays for it with ~1 000 scratch reloads per workgroup
  float* st_a = p.S_da + srow_g * M;
  {
    float x[R], qkeep[R];
    load_row<NCT>(x, xrow, grp);
    float xin[R];
    {
      float gm[R], bt[R], xh[R + 1];
      load_row<NCT>(gm, ln1g, grp);
      load_row<NCT>(bt, ln1b, grp);
      ln_row<NCT>(x, gm, bt, xh, xin);
    }
    WSTAMP(1);
    float kf[R];
    // q, k, v = LN1(x) Wq^T, Wk^T, Wv^T: six tiles (two output halves each)
    zero_acc<NOT>(acc); STAGE(0, xin, acc); unpack_at<NOT>(acc, qkeep, 0);
    zero_acc<NOT>(acc); STAGE(1, xin, acc); unpack_at<NOT>(acc, qkeep, NOT);
    zero_acc<NOT>(acc); STAGE(2, xin, acc); unpack_at<NOT>(acc, kf, 0);
    zero_acc<NOT>(acc); STAGE(3, xin, acc); unpack_at<NOT>(acc, kf, NOT);
    store_row<NCT>(kf, &bufK[(16 * w + i) * LD], grp);
    zero_acc<NOT>(acc); STAGE(4, xin, acc); unpack_at<NOT>(acc, kf, 0);
    zero_acc<NOT>(acc); STAGE(5, xin, acc); unpack_at<NOT>(acc, kf, NOT);
    store_row<NCT>(kf, &bufV[(16 * w + i) * LD], grp);
    store_row<NCT>(qkeep, st_q, grp);     // q waits in the stash for the attention backward
    __syncthreads();   // K, V of all 64 tokens visible
    WSTAMP(2);
    // attention: S[key][q] -> softmax over keys -> O, head by head
    float of[R];
#pragma unroll
    for (int k = 0; k < R; ++k) of[k] = 0.f;
    head_forward<LD, 0, QH>(bufK, bufV, qkeep, of, st_p, rowc, sc, i, grp, w);
    if constexpr (H > 1) head_forward<LD, QH, 2 * QH>(bufK, bufV, qkeep, of, st_p + 64, rowc + 64, sc, i, grp, w);
    if constexpr (H > 2) head_forward<LD, 2 * QH, 3 * QH>(bufK, bufV, qkeep, of, st_p + 128, rowc + 128, sc, i, grp, w);
    if constexpr (H > 3) head_forward<LD, 3 * QH, 4 * QH>(bufK, bufV, qkeep, of, st_p + 192, rowc + 192, sc, i, grp, w);
    static_assert(H <= 4, "at most 4 heads");
    WSTAMP(3);
    // y = x + att Wo^T
    zero_acc<NOT>(acc); STAGE(6, of, acc); unpack_at<NOT>(acc, y, 0);
    zero_acc<NOT>(acc); STAGE(7, of, acc); unpack_at<NOT>(acc, y, NOT);
    load_row<NCT>(x, xrow, grp);          // x again (L2): not kept live across the attention
#pragma unroll
    for (int k = 0; k < R; ++k) y[k] += x[k];
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
    WSTAMP(4);
    // hidden chunk mt (D units): a = LN2(y) W1[mt]^T + b1 (two tiles), z += gelu(a) W2[:, mt]^T (two tiles)
    f32x4 zacc[NCT];
    zero_acc<NCT>(zacc);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float am[R];
      zero_acc<NOT>(acc); STAGE(8 + 4 * mt, zn, acc); unpack_at<NOT>(acc, am, 0);
      zero_acc<NOT>(acc); STAGE(9 + 4 * mt, zn, acc); unpack_at<NOT>(acc, am, NOT);
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
        store_row<NCT>(b1, st_a + mt * D, grp);
      }
      if (mt == 3) {      // x again and the next vode's rows: their latency runs under the last two tiles
        load_row<NCT>(r, xrow, grp);
        load_row<NCT>(zn, hnext, grp);
      }
      STAGE(10 + 4 * mt, am, zacc);             // outputs [0, TR)
      STAGE(11 + 4 * mt, am, zacc + NOT);       // outputs [TR, D)
    }
    WSTAMP(5);
    float z[R], b2[R];
    unpack_at<NCT>(zacc, z, 0);
    load_row<NCT>(b2, pb + PB_B2, grp);
#pragma unroll
    for (int k = 0; k < R; ++k) r[k] = zn[k] - (r[k] + y[k] + z[k] + b2[k]);          // e_{g+2} = h_next - u  (r held x, zn held h_next)
    store_row<NCT>(r, p.E2 + (long long)g * nl + rowoff, grp);
  }

  WSTAMP(6);
  // ---- backward (cotangent r) -----------------------------------------------------------------------
  float dy[R];
  {
    // hidden chunk mt: da = (r W2)[mt] * gelu'(a) (two tiles of W2^T), dzn += da W1[mt] (two tiles of W1^T)
    f32x4 dacc[NCT];
    zero_acc<NCT>(dacc);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float dg[R], gp[R];
      load_row<NCT>(gp, st_a + mt * D, grp);       // gelu'(a) of this chunk, back from the stash under the two tiles
      zero_acc<NOT>(acc); STAGE(24 + 4 * mt, r, acc); unpack_at<NOT>(acc, dg, 0);
      zero_acc<NOT>(acc); STAGE(25 + 4 * mt, r, acc); unpack_at<NOT>(acc, dg, NOT);
#pragma unroll
      for (int k = 0; k < R; ++k) dg[k] *= gp[k];
      STAGE(26 + 4 * mt, dg, dacc);
      STAGE(27 + 4 * mt, dg, dacc + NOT);
    }
    WSTAMP(7);
    // dy = r + LN2^T(dzn); LN2's normalised input is recomputed from y (still in registers)
    float dzn[R], gm[R], xh[R + 1], tmp[R];
    unpack_at<NCT>(dacc, dzn, 0);
    load_row<NCT>(gm, pb + PB_LN2G, grp);
    ln_row<NCT>(y, gm, nullptr, xh, tmp);
    ln_bwd_row<NCT>(dzn, gm, xh, dy);
#pragma unroll
    for (int k = 0; k < R; ++k) dy[k] += r[k];
  }
  WSTAMP(8);
  float dk[R], dv[R];
  {
    // dO = dy Wo (two tiles of Wo^T); from here on r + dy is all that is needed of either
    float dof[R];
    zero_acc<NOT>(acc); STAGE(40, dy, acc); unpack_at<NOT>(acc, dof, 0);
    zero_acc<NOT>(acc); STAGE(41, dy, acc); unpack_at<NOT>(acc, dof, NOT);   // tile 42 waits in the staging registers through the attention backward
#pragma unroll
    for (int k = 0; k < R; ++k) r[k] += dy[k];
    __syncthreads();   // every wave is done with the weight buffers: reuse them for Q and dO
    store_row<NCT>(dof, &bufDO[(16 * w + i) * LD], grp);
    {
      float qk[R];
      load_row<NCT>(qk, st_q, grp);
      store_row<NCT>(qk, &bufQ[(16 * w + i) * LD], grp);
    }
    WSTAMP(9);
    // pass A (own queries): dP, delta, dS, dQ -- head by head; dq lands in dy
    float pT[16], dsT[16];      // one head: P and dS outlive the pass (handed to pass B through LDS)
    {
      float* dl = rowc + 64 * H;
      head_backward_a<LD, 0, QH>(bufK, bufV, dof, dy, st_p, dl, pT, dsT, qscale, i, grp, w);
      if constexpr (H > 1) { float pf_[16], dp_[16]; head_backward_a<LD, QH, 2 * QH>(bufK, bufV, dof, dy, st_p + 64, dl + 64, pf_, dp_, qscale, i, grp, w); }
      if constexpr (H > 2) { float pf_[16], dp_[16]; head_backward_a<LD, 2 * QH, 3 * QH>(bufK, bufV, dof, dy, st_p + 128, dl + 128, pf_, dp_, qscale, i, grp, w); }
      if constexpr (H > 3) { float pf_[16], dp_[16]; head_backward_a<LD, 3 * QH, 4 * QH>(bufK, bufV, dof, dy, st_p + 192, dl + 192, pf_, dp_, qscale, i, grp, w); }
    }
    __syncthreads();   // Q, dO rows and delta of all tokens visible; K and V no longer read (one head)
    WSTAMP(10);
    if constexpr (H == 1) {
      // pass B (own keys), one head: P and dS cross from the query owners to the key owners through the
      // K / V slots, which are dead now -- rows = queries as written, each lane gathers its key's column
      store_row<4>(pT, &bufK[(16 * w + i) * LD], grp);
      store_row<4>(dsT, &bufV[(16 * w + i) * LD], grp);
      __syncthreads();
      float sp[16], dp[16];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          sp[4 * tt + reg] = bufK[(16 * tt + 4 * grp + reg) * LD + 16 * w + i];
          dp[4 * tt + reg] = bufV[(16 * tt + 4 * grp + reg) * LD + 16 * w + i];
        }
      accum_head<LD, 0, QH>(bufDO, sp, dv, 1.0f, i, grp);          // dV = P^T dO
      accum_head<LD, 0, QH>(bufQ, dp, dk, qscale, i, grp);         // dK = dS^T Q
    } else {
      // pass B (own keys), several heads: S and dP recomputed with the key on the lane; dV, dK
      float kf[R], vf[R];
      load_row<NCT>(kf, &bufK[(16 * w + i) * LD], grp);
      load_row<NCT>(vf, &bufV[(16 * w + i) * LD], grp);
      const float* dl = rowc + 64 * H;
      head_backward_b<LD, 0, QH>(bufQ, bufDO, kf, vf, dk, dv, rowc, dl, sc, qscale, i, grp);
      if constexpr (H > 1) head_backward_b<LD, QH, 2 * QH>(bufQ, bufDO, kf, vf, dk, dv, rowc + 64, dl + 64, sc, qscale, i, grp);
      if constexpr (H > 2) head_backward_b<LD, 2 * QH, 3 * QH>(bufQ, bufDO, kf, vf, dk, dv, rowc + 128, dl + 128, sc, qscale, i, grp);
      if constexpr (H > 3) head_backward_b<LD, 3 * QH, 4 * QH>(bufQ, bufDO, kf, vf, dk, dv, rowc + 192, dl + 192, sc, qscale, i, grp);
    }
  }
  __syncthreads();     // Q / dO no longer read: the slots go back to the weight pipeline
  WSTAMP(11);
  {
    // dxn = [dq | dk | dv] Wqkv: output halves x three K-chunks of Wqkv^T [D][3D]; dx = r + dy + LN1^T(dxn)
    f32x4 xacc[NCT];
    zero_acc<NCT>(xacc);
    slot = 0;
    WRITE_ALL(0);                // tile 42 (held in registers across the attention backward) -> slot 0
    PREFETCH_ALL(43);
    STAGE(42, dy, xacc);         // (half 0, dq)
    STAGE(43, dy, xacc + NOT);   // (half 1, dq)
    STAGE(44, dk, xacc);
    STAGE(45, dk, xacc + NOT);
    STAGE(46, dv, xacc);
    STAGE(47, dv, xacc + NOT);
    WSTAMP(12);
    float dxn[R], gm[R], xh[R + 1], tmp[R], x[R];
    load_row<NCT>(x, xrow, grp);                 // LN1's normalised input is recomputed from x
    unpack_at<NCT>(xacc, dxn, 0);
    load_row<NCT>(gm, pb + PB_LN1G, grp);
    ln_row<NCT>(x, gm, nullptr, xh, tmp);
    ln_bwd_row<NCT>(dxn, gm, xh, dk);
#pragma unroll
    for (int k = 0; k < R; ++k) dk[k] += r[k];   // r holds r + dy
    store_row<NCT>(dk, p.BP1 + (long long)g * nl + rowoff, grp);
    WSTAMP(13);
  }
}

// ---- chain: status "forward" (/root/reference/src/decoder_transformer.py:232-241, 488-494) -------------
// One workgroup walks ONE image through all L blocks on the forward half of the block body (24 tiles per
// block, the prefetch running across block boundaries): INIT pass and the reconstruction after the weight
// update, instead of 7 per-op launches per block.  Block g reads its rows from (g == 0 ? xin : xout +
// (g - 1) stride) and writes them to xout + g stride; stride = 0 with xin == xout walks in place (a lane
// re-reads only rows it wrote itself).  P / LSE still go to the stash / LDS rows (head_forward writes them;
// nobody reads them here).
template <int NCT, int H>
__device__ __forceinline__ void wide_chain_body(const FusedP& p, float* lds, int b, const float* xin, float* xout, long long stride) {
  using Cf = WideCfg<NCT>;
  constexpr int D = Cf::D, M = Cf::M, R = Cf::R, NOT = Cf::NOT, TR = Cf::TR, LD = Cf::LD, SLOT = Cf::SLOT;
  constexpr int NST = Cf::NST, NF4 = Cf::NF4;
  constexpr int DH = D / H, QH = DH / 4;
  constexpr int TPB = 24;                 // forward tiles per block
  constexpr int HOLD = -1;
  const int NT = TPB * p.L;
  const float sc = rsqrtf((float)DH) * CORA_LOG2E;

  float* bufK = lds;
  float* bufV = lds + SLOT;
  float* wbuf = lds + 2 * SLOT;
  float* rowc = lds + 4 * SLOT;
  float* pb = lds + 4 * SLOT + 512;      // [b1 M | ln2 gamma D | ln2 beta D | b2 D]
  constexpr int PB_LN2G = M, PB_LN2B = M + D, PB_B2 = M + 2 * D;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, grp = lane >> 4;
  const long long rowoff = ((long long)b * 64 + 16 * w + i) * D;
  float* st_p = p.S_stash + ((long long)b * 64 + 16 * w + i) * (H * 64 + D);   // scratch rows of block 0's stash

  auto tile_src = [&](int k, int& ldw) -> const float* {
    const int gg = k / TPB, kk = k - gg * TPB;
    if (kk < 6) { ldw = D; return p.W + p.oWqkv + (long long)gg * 3 * D * D + (long long)kk * TR * D; }
    if (kk < 8) { ldw = D; return p.W + p.oWo + (long long)gg * D * D + (long long)(kk - 6) * TR * D; }
    const int mt = (kk - 8) >> 2, j = (kk - 8) & 3;
    if (j < 2) { ldw = D; return p.W + p.oW1 + (long long)gg * M * D + ((long long)mt * D + j * TR) * D; }
    ldw = M; return p.W + p.oW2 + (long long)gg * D * M + (long long)(j - 2) * TR * M + (long long)mt * D;
  };
  float4 sr0, sr1, sr2, sr3, sr4, sr5, sr6, sr7;
  int slot = 0;
  f32x4 acc[NOT];
  PREFETCH_ALL(0);
  WRITE_ALL(0);
  PREFETCH_ALL(1);

  for (int g = 0; g < p.L; ++g) {
    const float* xrow = (g == 0 ? xin : xout + (long long)(g - 1) * stride) + rowoff;
    float* urow = xout + (long long)g * stride + rowoff;
    const int k0 = g * TPB;
    __syncthreads();                     // the previous block's parameter vectors and K / V are no longer read
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
      float x[R], qkeep[R];
      load_row<NCT>(x, xrow, grp);
      float xn[R];
      {
        float gm[R], bt[R], xh[R + 1];
        load_row<NCT>(gm, p.W + p.oLn1g + (long long)g * D, grp);
        load_row<NCT>(bt, p.W + p.oLn1b + (long long)g * D, grp);
        ln_row<NCT>(x, gm, bt, xh, xn);
      }
      float kf[R];
      zero_acc<NOT>(acc); STAGE(k0 + 0, xn, acc); unpack_at<NOT>(acc, qkeep, 0);
      zero_acc<NOT>(acc); STAGE(k0 + 1, xn, acc); unpack_at<NOT>(acc, qkeep, NOT);
      zero_acc<NOT>(acc); STAGE(k0 + 2, xn, acc); unpack_at<NOT>(acc, kf, 0);
      zero_acc<NOT>(acc); STAGE(k0 + 3, xn, acc); unpack_at<NOT>(acc, kf, NOT);
      store_row<NCT>(kf, &bufK[(16 * w + i) * LD], grp);
      zero_acc<NOT>(acc); STAGE(k0 + 4, xn, acc); unpack_at<NOT>(acc, kf, 0);
      zero_acc<NOT>(acc); STAGE(k0 + 5, xn, acc); unpack_at<NOT>(acc, kf, NOT);
      store_row<NCT>(kf, &bufV[(16 * w + i) * LD], grp);
      __syncthreads();   // K, V of all 64 tokens visible
      float of[R];
#pragma unroll
      for (int k = 0; k < R; ++k) of[k] = 0.f;
      head_forward<LD, 0, QH>(bufK, bufV, qkeep, of, st_p, rowc, sc, i, grp, w);
      if constexpr (H > 1) head_forward<LD, QH, 2 * QH>(bufK, bufV, qkeep, of, st_p + 64, rowc + 64, sc, i, grp, w);
      if constexpr (H > 2) head_forward<LD, 2 * QH, 3 * QH>(bufK, bufV, qkeep, of, st_p + 128, rowc + 128, sc, i, grp, w);
      if constexpr (H > 3) head_forward<LD, 3 * QH, 4 * QH>(bufK, bufV, qkeep, of, st_p + 192, rowc + 192, sc, i, grp, w);
      zero_acc<NOT>(acc); STAGE(k0 + 6, of, acc); unpack_at<NOT>(acc, y, 0);
      zero_acc<NOT>(acc); STAGE(k0 + 7, of, acc); unpack_at<NOT>(acc, y, NOT);
      load_row<NCT>(x, xrow, grp);
#pragma unroll
      for (int k = 0; k < R; ++k) y[k] += x[k];
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
      zero_acc<NOT>(acc); STAGE(k0 + 8 + 4 * mt, zn, acc); unpack_at<NOT>(acc, am, 0);
      zero_acc<NOT>(acc); STAGE(k0 + 9 + 4 * mt, zn, acc); unpack_at<NOT>(acc, am, NOT);
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
      if (mt == 3) load_row<NCT>(xr, xrow, grp);     // x again: its latency runs under the last two tiles
      STAGE(k0 + 10 + 4 * mt, am, zacc);
      STAGE(k0 + 11 + 4 * mt, am, zacc + NOT);
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
__global__ __launch_bounds__(256, 1) void pc_chain_wide_kernel(FusedP p, const float* xin, float* xout, long long stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  wide_chain_body<NCT, H>(p, lds, (int)blockIdx.x, xin, xout, stride);
}

template <int NCT, int H>
__global__ __launch_bounds__(256, 1) void pc_block_wide_kernel(FusedP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x >= p.L * p.B) {   // the last B workgroups do the two edge layers of one image each
    edge_path_wide<NCT>(p, lds, (int)blockIdx.x - p.L * p.B);
    return;
  }
  // XCD-aware order (see fused_block.hip): XCD x gets a contiguous eighth of the (block, image) list
  int j = (int)blockIdx.x;
  const int n = p.L * p.B;
  if ((n & 7) == 0) j = (j & 7) * (n >> 3) + (j >> 3);
  wide_block_body<NCT, H>(p, lds, j / p.B, j % p.B);
}

template <int NCT, int H>
static int launch_wide_t(const FusedP& p, int L, hipStream_t st) {
  static bool attr[64] = {};
  const size_t sm = (size_t)WideCfg<NCT>::LDS_FLOATS * sizeof(float);
  if (first_use_on_device(attr)) {
    if (hipFuncSetAttribute((const void*)pc_block_wide_kernel<NCT, H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) return -1;
  }
  hipLaunchKernelGGL((pc_block_wide_kernel<NCT, H>), dim3((L + 1) * p.B), dim3(256), sm, st, p);
  return 0;
}

bool fused_wide_supported(int N, int P, int D, int H, int M) {
  // P == 48: the edge layers (edge_path_wide, fused_wide_common.h) hard-code the 4x4x3 patch of the sweep configs
  if (N != 64 || M != 4 * D || P != 48) return false;
  if (D == 128) return H == 1 || H == 2 || H == 4;
  if (D == 96) return H == 1 || H == 2 || H == 4;
  return false;
}

int launch_fused_wide(const FusedP& p, int L, int D, hipStream_t st) {
#ifdef WIDE_ONLY_84       /* compile-time experiments: one instantiation */
  return launch_wide_t<8, 4>(p, L, st);
#else
  if (D == 128) {
    switch (p.H) {
      case 1: return launch_wide_t<8, 1>(p, L, st);
      case 2: return launch_wide_t<8, 2>(p, L, st);
      case 4: return launch_wide_t<8, 4>(p, L, st);
    }
  } else if (D == 96) {
    switch (p.H) {
      case 1: return launch_wide_t<6, 1>(p, L, st);
      case 2: return launch_wide_t<6, 2>(p, L, st);
      case 4: return launch_wide_t<6, 4>(p, L, st);
    }
  }
  return -1;
#endif
}

template <int NCT, int H>
static int launch_chain_wide_t(const FusedP& p, const float* xin, float* xout, long long stride, hipStream_t st) {
  static bool attr[64] = {};
  const size_t sm = (size_t)WideCfg<NCT>::LDS_FLOATS * sizeof(float);
  if (first_use_on_device(attr)) {
    if (hipFuncSetAttribute((const void*)pc_chain_wide_kernel<NCT, H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) return -1;
  }
  hipLaunchKernelGGL((pc_chain_wide_kernel<NCT, H>), dim3(p.B), dim3(256), sm, st, p, xin, xout, stride);
  return 0;
}

// the L blocks of the status-"forward" chain for every image (see wide_chain_body)
int launch_chain_wide(const FusedP& p, int L, int D, const float* xin, float* xout, long long stride, hipStream_t st) {
  if (L < 1 || p.L != L) return -1;
  if (D == 128) {
    switch (p.H) {
      case 1: return launch_chain_wide_t<8, 1>(p, xin, xout, stride, st);
      case 2: return launch_chain_wide_t<8, 2>(p, xin, xout, stride, st);
      case 4: return launch_chain_wide_t<8, 4>(p, xin, xout, stride, st);
    }
  } else if (D == 96) {
    switch (p.H) {
      case 1: return launch_chain_wide_t<6, 1>(p, xin, xout, stride, st);
      case 2: return launch_chain_wide_t<6, 2>(p, xin, xout, stride, st);
      case 4: return launch_chain_wide_t<6, 4>(p, xin, xout, stride, st);
    }
  }
  return -1;
}
#undef STAGE
#undef GEMM_CT
#undef PREFETCH_ALL
#undef WRITE_ALL
#undef PF_ONE
#undef PB_ONE
