// gemm.hip -- grouped GEMM with fused PC epilogues (gfx950): fp32 on v_mfma_f32_32x32x2_f32 (parity
// path) or, opt-in (GemmP::bf16), operands rounded to bf16 on the way from LDS to the matrix core
// and v_mfma_f32_32x32x16_bf16 with fp32 accumulation (plain, or the two-term split for the
// forward predictions); everything in HBM / LDS stays fp32.
//
// One kernel template serves every dense contraction of the PC-ViT step
// (K1 patch projection, K3 QKV, K5 Wo, K6 MLP, K7 output projection, K9/K11/K12
// backward-data, K14 weight gradients).  C[R,O] = alpha * op(A) * op(B) (+ epilogue), per group
// (group = transformer block: every block reads its own vode h_l, so all L blocks of one
// inference step run as ONE grouped launch -- src/decoder_transformer.py:236-239 +
// pcx/predictive_coding/_vode.py:223-236).
//
//   A K-inner : A[row][k]  (activations, row-major)        A K-outer : A[k][row]  (weight-grad: dY^T)
//   B K-inner : B[col][k]  (eqx Linear weight [out,in])    B K-outer : B[k][col]  (backward-data: W as is)
//
// Tile 128x64xBK, 256 threads = 4 waves, wave w owns rows 32w..32w+31 x 64 cols = 2 MFMA tiles.
// k is permuted inside each 8-deep group so K-inner operands are fetched with one ds_read_b128:
// MFMA t of group j sees k = 8j + 4*(lane>>5) + t on both operands.
#include "common.h"
#include "gemm_epilogue.h"

#ifndef GEMM_EP_STREAM
#define GEMM_EP_STREAM false   /* streamed C was measured on the fp32 kernel too: -1.5 % on the cfg 4 GEMMs, within noise on the step */
#endif
#define BM 128
#define BN 64

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 pack_bf16(const float* lo, const float* hi4) {
  bf16x8 v;
#pragma unroll
  for (int t = 0; t < 4; ++t) { v[t] = (__bf16)lo[t]; v[4 + t] = (__bf16)hi4[t]; }
  return v;
}

// lo part of the two-term bf16 split x = hi + lo (+ O(2^-17 |x|)): what hi's rounding dropped
__device__ __forceinline__ bf16x8 pack_bf16_lo(const float* lo, const float* hi4, bf16x8 h) {
  bf16x8 v;
#pragma unroll
  for (int t = 0; t < 4; ++t) { v[t] = (__bf16)(lo[t] - (float)h[t]); v[4 + t] = (__bf16)(hi4[t] - (float)h[4 + t]); }
  return v;
}

// BF16: 0 = fp32 MFMA; 1 = bf16 operands; 2 = split bf16 (A = Ah + Al, B = Bh + Bl, three bf16 MFMAs
// Ah Bh + Ah Bl + Al Bh): products good to ~2^-16, for the forward predictions whose error
// e = h - u cancels most leading bits of u
template <int BK, bool AKO, bool BKO, int BF16>
__device__ __forceinline__ void gemm_body(const GemmP& p, const int bx, const int by, const int bz) {
  constexpr int LDA_S = AKO ? (BM + 4) : (BK + 4);
  constexpr int LDB_S = BKO ? (BN + 4) : (BK + 4);
  constexpr int A_ELEMS = AKO ? BK * (BM + 4) : BM * (BK + 4);
  constexpr int B_ELEMS = BKO ? BK * (BN + 4) : BN * (BK + 4);
  __shared__ __attribute__((aligned(16))) float As[A_ELEMS];
  __shared__ __attribute__((aligned(16))) float Bs[B_ELEMS];
  __shared__ double red_sh[4];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, hi = lane >> 5;
  const int g = bz / p.splitk, sk = bz % p.splitk;
  const int row0 = bx * BM, col0 = by * BN;

  const float* A = p.A + (long long)g * p.sA;
  const float* Bw = p.Bw + (long long)g * p.sB;

  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    int per = (p.K + p.splitk - 1) / p.splitk;
    per = (per + BK - 1) / BK * BK;
    kbeg = sk * per;
    kend = min(p.K, kbeg + per);
  }

  constexpr int A_V = BM * BK / 4 / 256;  // float4 per thread
  constexpr int B_V = BN * BK / 4 / 256;
  static_assert(A_V >= 1 && B_V >= 1, "tile too small");
  float4 ra[A_V], rb[B_V];

  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_V; ++i) {
      int f = tid + 256 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!AKO) {
        int row = f / (BK / 4), kq = f % (BK / 4);
        int gr = row0 + row, gk = k0 + 4 * kq;
        if (gr < p.R && gk < kend) v = *reinterpret_cast<const float4*>(A + (long long)gr * p.lda + gk);
      } else {
        int k = f / (BM / 4), c4 = f % (BM / 4);
        int gk = k0 + k, gr = row0 + 4 * c4;
        if (gk < kend && gr < p.R) v = *reinterpret_cast<const float4*>(A + (long long)gk * p.lda + gr);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_V; ++i) {
      int f = tid + 256 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!BKO) {
        int col = f / (BK / 4), kq = f % (BK / 4);
        int gc = col0 + col, gk = k0 + 4 * kq;
        if (gc < p.O && gk < kend) v = *reinterpret_cast<const float4*>(Bw + (long long)gc * p.ldb + gk);
      } else {
        int k = f / (BN / 4), c4 = f % (BN / 4);
        int gk = k0 + k, gc = col0 + 4 * c4;
        if (gk < kend && gc < p.O) v = *reinterpret_cast<const float4*>(Bw + (long long)gk * p.ldb + gc);
      }
      rb[i] = v;
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < A_V; ++i) {
      int f = tid + 256 * i;
      float4 v = ra[i];
      if (!AKO) {
        if (p.a_gelu) { v.x = cora_gelu(v.x); v.y = cora_gelu(v.y); v.z = cora_gelu(v.z); v.w = cora_gelu(v.w); }
        int row = f / (BK / 4), kq = f % (BK / 4);
        *reinterpret_cast<float4*>(&As[row * LDA_S + 4 * kq]) = v;
      } else {
        int k = f / (BM / 4), c4 = f % (BM / 4);
        *reinterpret_cast<float4*>(&As[k * LDA_S + 4 * c4]) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < B_V; ++i) {
      int f = tid + 256 * i;
      float4 v = rb[i];
      if (p.b_gelu) { v.x = cora_gelu(v.x); v.y = cora_gelu(v.y); v.z = cora_gelu(v.z); v.w = cora_gelu(v.w); }
      if (!BKO) {
        int col = f / (BK / 4), kq = f % (BK / 4);
        *reinterpret_cast<float4*>(&Bs[col * LDB_S + 4 * kq]) = v;
      } else {
        int k = f / (BN / 4), c4 = f % (BN / 4);
        *reinterpret_cast<float4*>(&Bs[k * LDB_S + 4 * c4]) = v;
      }
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;

  if (kbeg < kend) load_tiles(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();  // previous tile fully consumed
    store_tiles();
    __syncthreads();
    if (k0 + BK < kend) load_tiles(k0 + BK);  // prefetch into registers under the MFMAs
    // fragments of one 8-deep k group: lane (r, hi) holds k = 8j + 4hi + t, t < 4, of its row / column
    auto frag = [&](int j, float* a4, float (*b4)[4]) {
      if (!AKO) {
        float4 v = *reinterpret_cast<const float4*>(&As[(32 * w + r) * LDA_S + 8 * j + 4 * hi]);
        a4[0] = v.x; a4[1] = v.y; a4[2] = v.z; a4[3] = v.w;
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) a4[t] = As[(8 * j + 4 * hi + t) * LDA_S + 32 * w + r];
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        if (!BKO) {
          float4 v = *reinterpret_cast<const float4*>(&Bs[(32 * n + r) * LDB_S + 8 * j + 4 * hi]);
          b4[n][0] = v.x; b4[n][1] = v.y; b4[n][2] = v.z; b4[n][3] = v.w;
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) b4[n][t] = Bs[(8 * j + 4 * hi + t) * LDB_S + 32 * n + r];
        }
      }
    };
    if (!BF16) {
#pragma unroll
      for (int j = 0; j < BK / 8; ++j) {
        float a4[4], b4[2][4];
        frag(j, a4, b4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t], b4[0][t], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t], b4[1][t], acc[1], 0, 0, 0);
        }
      }
    } else {
      // two k groups make one 32x32x16 bf16 MFMA: the instruction wants 8 k values per lane, and
      // WHICH 8 is free as long as the A lane and the B lane of the same half agree -- they do
#pragma unroll
      for (int j = 0; j < BK / 8; j += 2) {
        float a4[2][4], b4[2][2][4];
        frag(j, a4[0], b4[0]);
        frag(j + 1, a4[1], b4[1]);
        bf16x8 av = pack_bf16(a4[0], a4[1]);
        if (BF16 == 1) {
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, pack_bf16(b4[0][n], b4[1][n]), acc[n], 0, 0, 0);
        } else {
          bf16x8 al = pack_bf16_lo(a4[0], a4[1], av);
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            bf16x8 bv = pack_bf16(b4[0][n], b4[1][n]);
            bf16x8 bl = pack_bf16_lo(b4[0][n], b4[1][n], bv);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bv, acc[n], 0, 0, 0);   // small terms first
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bl, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[n], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- epilogue (gemm_epilogue.h) -------------------------------------------------------------
  const GemmEp q = gemm_ep_ptrs(p, g);
  double red = 0.0;
#pragma unroll
  for (int n = 0; n < 2; ++n) gemm_ep_tile<GEMM_EP_STREAM>(p, q, acc[n], row0 + 32 * w + 4 * hi, col0 + 32 * n + r, red);
  gemm_ep_reduce(p, g, bx, red, red_sh);
}

#ifndef GEMM_WAVES
#define GEMM_WAVES 4
#endif
// Tile order: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs (one L2 each), so the
// 1-D grid is cut into 8 contiguous runs of the tile list, one per XCD, with the COLUMN tile fastest: the
// O / 64 workgroups that read one 128-row tile of A run back to back on one XCD and all but the first find it
// in that L2 (with the row tile fastest they were R / 128 launches apart and A came back from HBM / MALL once
// per column tile).  tx, ty = row / column tiles, nz = groups * splitk; surplus workgroups of the rounded-up
// grid return at once.
template <int BK, bool AKO, bool BKO, int BF16>
__global__ __launch_bounds__(256, GEMM_WAVES) void gemm_kernel(GemmP p, int tx, int ty, int ntiles) {
  const int per = (ntiles + 7) >> 3;
  const int lin = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (lin >= ntiles) return;
  gemm_body<BK, AKO, BKO, BF16>(p, (lin / ty) % tx, lin % ty, lin / (ty * tx));
}

// Several independent TN problems (the weight-gradient contractions of one batch: different shapes,
// each too small to fill the chip) in ONE launch: block -> (problem, tile) through a prefix table.
template <int BF16>
__global__ __launch_bounds__(256, 5) void gemm_multi_tn_kernel(GemmMulti m) {
  int q = 0;
#pragma unroll
  for (int k = 1; k < GEMM_MULTI_MAX; ++k) if (k < m.n && (int)blockIdx.x >= m.start[k]) q = k;
  const GemmP& p = m.p[q];
  int rel = (int)blockIdx.x - m.start[q];
  const int tx = (p.R + BM - 1) / BM, ty = (p.O + BN - 1) / BN;
  const int bx = rel % tx; rel /= tx;
  const int by = rel % ty; rel /= ty;
  gemm_body<16, true, true, BF16>(p, bx, by, rel);
}

template <int BK, int BF16>
static void launch_bk(const GemmP& p, bool ako, bool bko, dim3 g3, hipStream_t st) {
  const int tx = (int)g3.x, ty = (int)g3.y, ntiles = tx * ty * (int)g3.z;
  const dim3 grid((unsigned)(((ntiles + 7) / 8) * 8));
  if (!ako && !bko) hipLaunchKernelGGL((gemm_kernel<BK, false, false, BF16>), grid, dim3(256), 0, st, p, tx, ty, ntiles);
  else if (!ako && bko) hipLaunchKernelGGL((gemm_kernel<BK, false, true, BF16>), grid, dim3(256), 0, st, p, tx, ty, ntiles);
  else if (ako && bko) hipLaunchKernelGGL((gemm_kernel<BK, true, true, BF16>), grid, dim3(256), 0, st, p, tx, ty, ntiles);
  else hipLaunchKernelGGL((gemm_kernel<BK, true, false, BF16>), grid, dim3(256), 0, st, p, tx, ty, ntiles);
}

void launch_gemm(const GemmP& p, bool a_kouter, bool b_kouter, hipStream_t st) {
  // bf16 mode, both operands K-inner: the bf16-staged kernel (CORA_BF16_INREG=1 keeps the in-register
  // conversion of this file, for A/B measurements)
  static const bool inreg = getenv("CORA_BF16_INREG") != nullptr;
  if (!a_kouter && !b_kouter && !inreg && gemm_bf16_nt_supported(p)) { launch_gemm_bf16_nt(p, st); return; }
  dim3 grid((p.R + BM - 1) / BM, (p.O + BN - 1) / BN, p.groups * p.splitk);
  const bool bk32 = (p.K % 32 == 0 && p.splitk == 1);
  if (p.bf16 == 3) {            // three-term mode outside the bf16-staged kernel: plain fp32 MFMA
    if (bk32) launch_bk<32, 0>(p, a_kouter, b_kouter, grid, st);
    else launch_bk<16, 0>(p, a_kouter, b_kouter, grid, st);
  } else if (p.bf16 == 2) {
    if (bk32) launch_bk<32, 2>(p, a_kouter, b_kouter, grid, st);
    else launch_bk<16, 2>(p, a_kouter, b_kouter, grid, st);
  } else if (p.bf16 == 1) {
    if (bk32) launch_bk<32, 1>(p, a_kouter, b_kouter, grid, st);
    else launch_bk<16, 1>(p, a_kouter, b_kouter, grid, st);
  } else {
    if (bk32) launch_bk<32, 0>(p, a_kouter, b_kouter, grid, st);
    else launch_bk<16, 0>(p, a_kouter, b_kouter, grid, st);
  }
}

void launch_gemm_multi_tn(const GemmMulti& m_in, hipStream_t st) {
  GemmMulti m = m_in;
  int total = 0;
  for (int k = 0; k < m.n; ++k) {
    m.start[k] = total;
    const GemmP& p = m.p[k];
    total += ((p.R + BM - 1) / BM) * ((p.O + BN - 1) / BN) * p.groups * p.splitk;
  }
  m.start[m.n] = total;
  if (m.p[0].bf16 == 1 || m.p[0].bf16 == 2) hipLaunchKernelGGL((gemm_multi_tn_kernel<1>), dim3(total), dim3(256), 0, st, m);
  else hipLaunchKernelGGL((gemm_multi_tn_kernel<0>), dim3(total), dim3(256), 0, st, m);
}
