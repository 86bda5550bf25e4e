// gemm_bf16.hip -- the bf16 mode's grouped GEMM (gfx950): C[R,O] = alpha * A[R,K] * B[O,K]^T + epilogue.
//
// Opt-in path (cora_config.dtype_mode = 1; the fp32 kernel of gemm.hip is the parity path).  Everything
// in HBM stays fp32; operands are rounded to bf16 ONCE, on their way from the staging registers into
// LDS, so a k-tile costs half the LDS bytes of the fp32 kernel, its fragments are one ds_read_b128 per
// v_mfma_f32_32x32x16_bf16 operand (8 k values per lane) and no conversion work sits between the
// MFMAs.  SPLIT = 1 keeps the two-term split x = hi + lo of both operands (three MFMAs, products good
// to ~2^-16): the forward predictions feed e = h - u, which cancels most of u, and plain bf16 there
// costs 2.5e-2 on the 6-block vodes (gemm.hip).  SPLIT = 2 (dtype_mode = 2, "bf16x3") keeps three terms
// x = hi + mid + lo of both operands and the six products whose weight is above 2^-24 (hi*hi, hi*mid, mid*hi,
// mid*mid, hi*lo, lo*hi): fp32-grade products from the bf16 pipe, which at six MFMAs still has 2.7 x the
// rate of v_mfma_f32_32x32x2_f32.
//
// Both operands are K-inner ("NT"): activations [row][k] and weights [out][in].  The backward-data
// GEMMs, whose weight is K-outer as stored, read the transposed weight copies the context keeps in
// bf16 mode (ctx.hip: refresh_wt), so one operand form serves every GEMM of the inference step; the
// weight-gradient contractions (K-outer on both sides, once per batch) stay on gemm.hip.  BPRE: the weight
// operand comes from bf16 copies the context refreshes with every weight update (hi, and lo for the split
// form) and goes to LDS as 16-byte pieces without conversion -- bit-identical products, -8 % on the cfg 4
// GEMM set.
//
// Tile 128 x 128 x 32, 256 threads = 2 x 2 waves of 64 x 64 (2 x 2 MFMA tiles, 64 accumulator
// registers): per 16-deep k step a wave reads 2 + 2 fragments for 4 MFMAs (split: 4 + 4 for 12).
// LDS rows are 32 bf16 + 8 pad = 80 B: the 16-lane groups of ds_read_b128 (MI355X_MICROARCH.md, LDS)
// then touch 16 distinct 16-byte slots of the 256-byte bank row.  Workgroup ids are dealt so that each
// XCD gets a contiguous range with the column tiles of one row tile adjacent: a row tile of A is
// fetched from HBM once and re-read from that XCD's L2 for the other column tiles.  The output and the
// read-once epilogue operands are streamed (nontemporal: -5 % over the cfg 4 GEMM set).
//
// Measured on the cfg 4 shapes (scripts/gemm_bf16_bench.hip, 8 GEMMs of one inference step, 1 986 us):
// the MFMA + staging chain alone takes 917 us, global loads add ~580 and the epilogue ~490, largely
// additively -- with one staging register set a wave that waits for its loads is not multiplying.
// Tried and dropped: 64-deep k slabs (occupancy 3 -> 2: 2 171 us), 8 waves of 64 x 32 (3 065 us, spills
// under the 128-register cap), and a producer / consumer split of a 512-thread persistent workgroup
// with three slabs in flight per producer (2 324 us: the compiler's vmcnt bookkeeping drains every
// outstanding load at the first use once loads from several loop iterations are in flight, so the
// deeper prefetch never materialises without hand-written waits).
#include "common.h"
#include "gemm_epilogue.h"
#include <cstdio>
#include <cstdlib>

#ifndef EP_STREAM
#define EP_STREAM true
#endif
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#ifndef BF_TK
#define BF_TK 32
#endif
#ifndef BF_TK_A16
#define BF_TK_A16 64
#endif
#ifndef BF_TM
#define BF_TM 128
#endif
#ifndef BF_TN
#define BF_TN 128
#endif
constexpr int TM = BF_TM, TN = BF_TN;     // the slab depth TK and what follows from it are set per kernel form (below)
#ifndef BF_THREADS
#define BF_THREADS 256
#endif
constexpr int NTHR = BF_THREADS, NWC = NTHR / 128;      // waves across the 128 columns (2 or 4); 2 wave rows of 64
constexpr int WN = TN / NWC / 32;                        // 32-column MFMA tiles per wave (2 or 1)
constexpr int WM = TM / 2 / 32;                          // 32-row MFMA tiles per wave (2 wave rows)

__device__ __forceinline__ bf16x4 to_bf16(const float4 v) {
  bf16x4 h;
  h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
  return h;
}
__device__ __forceinline__ bf16x4 to_bf16_lo(const float4 v, const bf16x4 h) {
  bf16x4 l;
  l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]);
  l[2] = (__bf16)(v.z - (float)h[2]); l[3] = (__bf16)(v.w - (float)h[3]);
  return l;
}
// third term: bf16(x - hi - mid); both subtractions are exact in fp32
__device__ __forceinline__ bf16x4 to_bf16_3(const float4 v, const bf16x4 h, const bf16x4 m) {
  bf16x4 l;
  l[0] = (__bf16)((v.x - (float)h[0]) - (float)m[0]); l[1] = (__bf16)((v.y - (float)h[1]) - (float)m[1]);
  l[2] = (__bf16)((v.z - (float)h[2]) - (float)m[2]); l[3] = (__bf16)((v.w - (float)h[3]) - (float)m[3]);
  return l;
}

#ifndef BF_OCC
#define BF_OCC 2
#endif
#ifndef BF_OCC_A16
#define BF_OCC_A16 3   /* bf16 A operand with 64-deep slabs: 163 registers, no spills (at 32-deep slabs it fits four workgroups per CU
                         with 5 spilled registers: W1^T 111 us against 102 us for this form; 64-deep at four spills 141) */
#endif
template <int SPLIT, bool BPRE, bool A16 = false>
__global__ __launch_bounds__(NTHR, NTHR / 256 * (A16 ? BF_OCC_A16 : BF_OCC)) void gemm_bf16_nt_kernel(const GemmP p, const int tx, const int ty, const int ntiles) {
  // slab depth: 32 for fp32 A operands (64 costs them a workgroup per CU); 64 for the bf16-A form, whose rows are then whole
  // 128-byte lines per slab on both operands and whose staging still fits 163 registers (three workgroups per CU):
  // W1^T 120 -> 102 us (866 TFLOP/s), Wqkv^T 89 -> 80 us on the cfg 4 shapes.
  constexpr int TK = A16 ? BF_TK_A16 : BF_TK;
  constexpr int NVB = TN * TK / 4 / NTHR, NPB = TN * TK / 8 / NTHR;
  constexpr int KQ = TK / 4, NV = TM * TK / 4 / NTHR, RSTEP = NTHR / KQ;   // staging: float4 per row, per thread and operand, row step
  constexpr int LDR = TK + 8;          // bf16 per LDS row
  __shared__ __attribute__((aligned(16))) __bf16 Ah[TM * LDR];
  __shared__ __attribute__((aligned(16))) __bf16 Bh[TN * LDR];
  __shared__ __attribute__((aligned(16))) __bf16 Al[SPLIT ? TM * LDR : 8];
  __shared__ __attribute__((aligned(16))) __bf16 Bl[SPLIT ? TN * LDR : 8];
  __shared__ __attribute__((aligned(16))) __bf16 A3[SPLIT == 2 ? TM * LDR : 8];
  __shared__ __attribute__((aligned(16))) __bf16 B3[SPLIT == 2 ? TN * LDR : 8];

  // hardware deals consecutive workgroup ids round-robin over the 8 XCDs: XCD x works through the
  // contiguous range [x * per, (x + 1) * per) of the tile list (column tile fastest)
  const int per = (ntiles + 7) >> 3;
  const int lin = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (lin >= ntiles) return;
  const int by = lin % ty, bx = (lin / ty) % tx, g = lin / (ty * tx);

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, hi = lane >> 5;
  const int wr = w / NWC, wc = w % NWC;
  const int row0 = bx * TM, col0 = by * TN;
  const float* A = p.A + (long long)g * p.sA;
  const float* Bw = p.Bw + (long long)g * p.sB;

  // staging: float4 #f of a 128 x TK tile = row f / KQ, k quad f % KQ; thread t takes #t + 256 i (row t / KQ + RSTEP i)
  const int srow = tid / KQ, skq = tid % KQ;
  constexpr int NPA = TM * TK / 8 / 256;     // 16-byte pieces of a bf16 A tile per thread
  float4 ra[A16 ? 1 : NV], rb[BPRE ? 1 : NVB];
  uint4 pbh[NPB], pbl[NPB], pb3[NPB], pa16[A16 ? NPA : 1];
  static_assert(!A16 || (SPLIT == 0 && BPRE && NTHR == 256), "bf16 A operand: plain form, pre-converted B");
  constexpr int PPR = TK / 8;                 // 16-byte pieces per tile row
  auto load_tiles = [&](int k0) {
    const int gk = k0 + 4 * skq;
    const bool kok = gk < p.K;
    if (A16) {
      // A already bf16 in memory (backward-data scratch written as bf16 by its producer): 16-byte pieces of 8 bf16 go to
      // LDS as they are, like the pre-converted weights -- half the bytes from HBM / L2 and no conversion work
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        const int f = tid + 256 * i;
        const int gr = row0 + f / PPR, gkb = k0 + 8 * (f % PPR);
        const bool ok = gkb < p.K && gr < p.R;
        pa16[i] = ok ? *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p.A) + (long long)g * p.sA + (long long)gr * p.lda + gkb)
                     : make_uint4(0u, 0u, 0u, 0u);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int gr = row0 + srow + RSTEP * i;
        ra[i] = (kok && gr < p.R) ? *reinterpret_cast<const float4*>(A + (long long)gr * p.lda + gk) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (BPRE) {
      // weights converted once per weight update (ctx.hip: refresh_wt): 16-byte pieces of 8 bf16 go to LDS as
      // they are -- piece #f of the 128 x 32 tile = row f / 4, k octet f % 4; thread t takes #t and #t + 256
      static_assert(!BPRE || NTHR == 256, "pre-converted B staging is written for 256 threads");
#pragma unroll
      for (int i = 0; i < NPB; ++i) {
        const int f = tid + 256 * i;
        const int gc = col0 + f / PPR, gkb = k0 + 8 * (f % PPR);
        const bool ok = gkb < p.K && gc < p.O;
        const long long off = (long long)g * p.sB + (long long)gc * p.ldb + gkb;
        pbh[i] = ok ? *reinterpret_cast<const uint4*>(p.Bh + off) : make_uint4(0u, 0u, 0u, 0u);
        if (SPLIT) pbl[i] = ok ? *reinterpret_cast<const uint4*>(p.Bl + off) : make_uint4(0u, 0u, 0u, 0u);
        if (SPLIT == 2) pb3[i] = ok ? *reinterpret_cast<const uint4*>(p.B3 + off) : make_uint4(0u, 0u, 0u, 0u);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NVB; ++i) {
        const int gc = col0 + srow + RSTEP * i;
        rb[i] = (kok && gc < p.O) ? *reinterpret_cast<const float4*>(Bw + (long long)gc * p.ldb + gk) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto store_tiles = [&]() {
    if (A16) {
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        const int f = tid + 256 * i;
        *reinterpret_cast<uint4*>(&Ah[(f / PPR) * LDR + 8 * (f % PPR)]) = pa16[i];
      }
    }
#pragma unroll
    for (int i = 0; i < (A16 ? 0 : NV); ++i) {
      float4 v = ra[i];
      const int o = (srow + RSTEP * i) * LDR + 4 * skq;
#ifdef EXP_NOCVT   /* timing experiment (scripts/gemm_bf16_bench.hip): A as if pre-split by its producer -- no VALU between load and LDS */
      *reinterpret_cast<uint2*>(&Ah[o]) = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
      if (SPLIT) *reinterpret_cast<uint2*>(&Al[o]) = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w));
      continue;
#endif
      if (p.a_gelu) { v.x = cora_gelu(v.x); v.y = cora_gelu(v.y); v.z = cora_gelu(v.z); v.w = cora_gelu(v.w); }
      const bf16x4 h = to_bf16(v);
      *reinterpret_cast<bf16x4*>(&Ah[o]) = h;
      if (SPLIT) {
        const bf16x4 m = to_bf16_lo(v, h);
        *reinterpret_cast<bf16x4*>(&Al[o]) = m;
        if (SPLIT == 2) *reinterpret_cast<bf16x4*>(&A3[o]) = to_bf16_3(v, h, m);
      }
    }
    if (BPRE) {
#pragma unroll
      for (int i = 0; i < NPB; ++i) {
        const int f = tid + 256 * i;
        const int o = (f / PPR) * LDR + 8 * (f % PPR);
        *reinterpret_cast<uint4*>(&Bh[o]) = pbh[i];
        if (SPLIT) *reinterpret_cast<uint4*>(&Bl[o]) = pbl[i];
        if (SPLIT == 2) *reinterpret_cast<uint4*>(&B3[o]) = pb3[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < NVB; ++i) {
        const float4 v = rb[i];
        const int o = (srow + RSTEP * i) * LDR + 4 * skq;
        const bf16x4 h = to_bf16(v);
        *reinterpret_cast<bf16x4*>(&Bh[o]) = h;
        if (SPLIT) {
          const bf16x4 m = to_bf16_lo(v, h);
          *reinterpret_cast<bf16x4*>(&Bl[o]) = m;
          if (SPLIT == 2) *reinterpret_cast<bf16x4*>(&B3[o]) = to_bf16_3(v, h, m);
        }
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  // lane (r, hi) holds k = 16 kk + 8 hi .. + 7 of its row (A) / column (B): 16 bytes, one ds_read_b128
  const int aoff = (32 * WM * wr + r) * LDR + 8 * hi;
  const int boff = (32 * WN * wc + r) * LDR + 8 * hi;

  load_tiles(0);
  for (int k0 = 0; k0 < p.K; k0 += TK) {
    __syncthreads();            // the previous tile is fully consumed
    store_tiles();
    __syncthreads();
    if (k0 + TK < p.K) load_tiles(k0 + TK);     // next tile into registers under the MFMAs
#pragma unroll
    for (int kk = 0; kk < TK / 16; ++kk) {
      bf16x8 ah[WM], bh[WN], al[WM], bl[WN];
#pragma unroll
      for (int m = 0; m < WM; ++m) ah[m] = *reinterpret_cast<const bf16x8*>(&Ah[aoff + 32 * m * LDR + 16 * kk]);
#pragma unroll
      for (int n = 0; n < WN; ++n) bh[n] = *reinterpret_cast<const bf16x8*>(&Bh[boff + 32 * n * LDR + 16 * kk]);
      if (SPLIT == 2) {   // smallest terms first: hi * lo, lo * hi, mid * mid
        bf16x8 a3[WM], b3[WN];
#pragma unroll
        for (int m = 0; m < WM; ++m) a3[m] = *reinterpret_cast<const bf16x8*>(&A3[aoff + 32 * m * LDR + 16 * kk]);
#pragma unroll
        for (int n = 0; n < WN; ++n) b3[n] = *reinterpret_cast<const bf16x8*>(&B3[boff + 32 * n * LDR + 16 * kk]);
#pragma unroll
        for (int m = 0; m < WM; ++m) al[m] = *reinterpret_cast<const bf16x8*>(&Al[aoff + 32 * m * LDR + 16 * kk]);
#pragma unroll
        for (int n = 0; n < WN; ++n) bl[n] = *reinterpret_cast<const bf16x8*>(&Bl[boff + 32 * n * LDR + 16 * kk]);
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
          for (int n = 0; n < WN; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[m], bh[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], b3[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bl[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
          }
      } else if (SPLIT) {
#pragma unroll
        for (int m = 0; m < WM; ++m) al[m] = *reinterpret_cast<const bf16x8*>(&Al[aoff + 32 * m * LDR + 16 * kk]);
#pragma unroll
        for (int n = 0; n < WN; ++n) bl[n] = *reinterpret_cast<const bf16x8*>(&Bl[boff + 32 * n * LDR + 16 * kk]);
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
          for (int n = 0; n < WN; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);   // small terms first
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
          }
      }
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
    }
  }

  const GemmEp q = gemm_ep_ptrs(p, g);
  double red = 0.0;
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n)
      gemm_ep_tile<EP_STREAM>(p, q, acc[m][n], row0 + 32 * WM * wr + 32 * m + 4 * hi, col0 + 32 * WN * wc + 32 * n + r, red);
  if (p.red) {      // energy / sum of squares of the whole workgroup (OUT_GRAD: 8 slots per vode, folded on the host)
    double* red_sh = reinterpret_cast<double*>(Ah);      // the tiles are dead: their memory carries the wave sums
    const double wsum = wave_sum_d(red);
    __syncthreads();                                      // every wave is past its last fragment read
    if (lane == 0) red_sh[w] = wsum;
    __syncthreads();
    if (tid == 0) {
      double tot = 0.0;
#pragma unroll
      for (int k = 0; k < NTHR / 64; ++k) tot += red_sh[k];
      if (p.mode == OUT_GRAD) atomicAdd(&p.red[((long long)g * p.red_stride) * 8 + (bx & 7)], tot);
      else atomicAdd(&p.red[(long long)g * p.red_stride], tot);
    }
  }
}

}  // namespace

bool gemm_bf16_nt_supported(const GemmP& p) {
  return p.bf16 != 0 && p.splitk == 1 && !p.b_gelu && p.K % 4 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0 &&
         (p.sA % 4) == 0 && (p.sB % 4) == 0;
}

void launch_gemm_bf16_nt(const GemmP& p, hipStream_t st) {
  const int tx = (p.R + TM - 1) / TM, ty = (p.O + TN - 1) / TN;
  const int ntiles = tx * ty * p.groups;
  const dim3 grid(((ntiles + 7) / 8) * 8);
  // pre-converted weights need 16-byte aligned octets of k
  const int terms = p.bf16;     // 1 plain, 2 two-term split, 3 three-term split
  const bool pre = p.Bh != nullptr && (terms < 2 || p.Bl != nullptr) && (terms < 3 || p.B3 != nullptr) && p.K % 8 == 0 &&
                   p.ldb % 8 == 0 && p.sB % 8 == 0 && ((uintptr_t)p.Bh % 16) == 0 && (terms < 2 || ((uintptr_t)p.Bl % 16) == 0) &&
                   (terms < 3 || ((uintptr_t)p.B3 % 16) == 0);
  if (p.a16 || p.c16) {
    // set by ctx.hip only where these hold (checked at cora_create); anything else is a programming error
    const bool ok16 = terms == 1 && pre && p.K % 8 == 0 && p.lda % 8 == 0 && p.sA % 8 == 0 && ((uintptr_t)p.A % 16) == 0 && !p.a_gelu &&
                      (!p.c16 || (p.mode == OUT_GELUGRAD && p.O % 2 == 0 && p.ldc % 2 == 0 && p.sC % 2 == 0));
    if (!ok16) { fprintf(stderr, "gemm_bf16: bf16 scratch operand requested for a GEMM that cannot take it\n"); abort(); }
  }
  if (p.a16) {
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<0, true, true>), grid, dim3(NTHR), 0, st, p, tx, ty, ntiles);
    return;
  }
  if (terms == 3) {
    if (pre) hipLaunchKernelGGL((gemm_bf16_nt_kernel<2, true>), grid, dim3(NTHR), 0, st, p, tx, ty, ntiles);
    else hipLaunchKernelGGL((gemm_bf16_nt_kernel<2, false>), grid, dim3(NTHR), 0, st, p, tx, ty, ntiles);
  } else if (terms == 2) {
    if (pre) hipLaunchKernelGGL((gemm_bf16_nt_kernel<1, true>), grid, dim3(NTHR), 0, st, p, tx, ty, ntiles);
    else hipLaunchKernelGGL((gemm_bf16_nt_kernel<1, false>), grid, dim3(NTHR), 0, st, p, tx, ty, ntiles);
  } else {
    if (pre) hipLaunchKernelGGL((gemm_bf16_nt_kernel<0, true>), grid, dim3(NTHR), 0, st, p, tx, ty, ntiles);
    else hipLaunchKernelGGL((gemm_bf16_nt_kernel<0, false>), grid, dim3(NTHR), 0, st, p, tx, ty, ntiles);
  }
}
