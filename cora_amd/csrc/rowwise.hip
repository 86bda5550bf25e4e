// rowwise.hip -- HBM-bound row-wise and element-wise kernels of the PC step (gfx950):
// LayerNorm forward/backward (eqx.nn.LayerNorm, /root/reference/pcx/nn/_layer.py:296-315),
// patchify/unpatchify (/root/reference/src/decoder_transformer.py:259-294), the fused
// scale/clip/momentum vode update (optax clip_by_global_norm + sgd trace,
// /root/reference/examples/debug_transformer_wandb.py:1126,1134-1140 via
// /root/reference/pcx/utils/_optim.py:53-148), bias / LayerNorm parameter gradient column sums,
// clip + AdamW (/root/reference/examples/debug_transformer_wandb.py:1129,1153-1159) and the
// clipped reconstruction MSE (/root/reference/src/decoder_transformer.py:491-494).
// All loads/stores are 16 B per lane; reductions are wave shuffles -> LDS -> one f64 atomic per block.
#include "common.h"
#include "vode_mult.h"

#define LN_EPS 1e-5f

// 16 lanes per row, float4 per lane, NV float4 per lane (D <= 64*NV)
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, long long sParam,
                                                     float* __restrict__ xn, float* __restrict__ stats,
                                                     int rows_per_group, long long rows, int D) {
  const int sub = threadIdx.x & 15;
  const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = row < rows;
  const int g = live ? (int)(row / rows_per_group) : 0;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int cidx = 64 * i + 4 * sub;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && cidx < D) v[i] = *reinterpret_cast<const float4*>(x + row * D + cidx);
    s += v[i].x + v[i].y + v[i].z + v[i].w;
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int cidx = 64 * i + 4 * sub;
    if (cidx < D) {
      float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += a * a + b * b + c * c + d * d;
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(fmaxf(q / (float)D, 0.f) + LN_EPS);
  if (!live) return;
  const float* gm = gamma + (long long)g * sParam;
  const float* bt = beta + (long long)g * sParam;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int cidx = 64 * i + 4 * sub;
    if (cidx < D) {
      float4 gg = *reinterpret_cast<const float4*>(gm + cidx);
      float4 bb = *reinterpret_cast<const float4*>(bt + cidx);
      float4 o;
      o.x = (v[i].x - mean) * rstd * gg.x + bb.x;
      o.y = (v[i].y - mean) * rstd * gg.y + bb.y;
      o.z = (v[i].z - mean) * rstd * gg.z + bb.z;
      o.w = (v[i].w - mean) * rstd * gg.w + bb.w;
      *reinterpret_cast<float4*>(xn + row * D + cidx) = o;
    }
  }
  if (sub == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
}

template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdP p) {
  __shared__ double red_sh[4];
  const int sub = threadIdx.x & 15;
  const long long rows = (long long)p.rows_per_group * p.groups;
  const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = row < rows;
  // all 16 rows of a block belong to one group when rows_per_group % 16 == 0; otherwise handle per row
  const int g = live ? (int)(row / p.rows_per_group) : 0;
  const int D = p.D;
  const float* gm = p.gamma + (long long)g * p.sParam;
  float mean = 0.f, rstd = 0.f;
  if (live) { mean = p.stats[2 * row]; rstd = p.stats[2 * row + 1]; }
  float4 dh[NV], xh[NV];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int cidx = 64 * i + 4 * sub;
    dh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && cidx < D) {
      float4 d = *reinterpret_cast<const float4*>(p.dxn + row * D + cidx);
      float4 xv = *reinterpret_cast<const float4*>(p.x + row * D + cidx);
      float4 gg = *reinterpret_cast<const float4*>(gm + cidx);
      dh[i] = make_float4(d.x * gg.x, d.y * gg.y, d.z * gg.z, d.w * gg.w);
      xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
    }
    s1 += dh[i].x + dh[i].y + dh[i].z + dh[i].w;
    s2 += dh[i].x * xh[i].x + dh[i].y * xh[i].y + dh[i].z * xh[i].z + dh[i].w * xh[i].w;
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  const float m1 = s1 / (float)D, m2 = s2 / (float)D;
  double red = 0.0;
  const float gs = (p.mode == 1 && p.gscale) ? p.gscale[g] : 1.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int cidx = 64 * i + 4 * sub;
    if (live && cidx < D) {
      const long long idx = row * D + cidx;
      float4 rr = p.r ? *reinterpret_cast<const float4*>(p.r + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
      float dx[4] = {rstd * (dh[i].x - m1 - xh[i].x * m2), rstd * (dh[i].y - m1 - xh[i].y * m2),
                     rstd * (dh[i].z - m1 - xh[i].z * m2), rstd * (dh[i].w - m1 - xh[i].w * m2)};
      float rv[4] = {rr.x, rr.y, rr.z, rr.w};
      float out[4];
      if (p.mode == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = rv[j] + dx[j];
      } else if (p.mode == 2) {   // raw cotangent r + dy + LN^T(dxn), no gradient assembly
        float4 dy4 = *reinterpret_cast<const float4*>(p.dy + idx);
        out[0] = rv[0] + dy4.x + dx[0]; out[1] = rv[1] + dy4.y + dx[1];
        out[2] = rv[2] + dy4.z + dx[2]; out[3] = rv[3] + dy4.w + dx[3];
      } else {
        float4 dy4 = p.dy ? *reinterpret_cast<const float4*>(p.dy + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 ow4 = *reinterpret_cast<const float4*>(p.own + idx);
        float4 xv4 = *reinterpret_cast<const float4*>(p.x + idx);
        float dyv[4] = {dy4.x, dy4.y, dy4.z, dy4.w};
        float ow[4] = {ow4.x, ow4.y, ow4.z, ow4.w};
        float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float sg = (xv[j] > 0.f) ? 1.f : ((xv[j] < 0.f) ? -1.f : 0.f);
          float own = ow[j] + p.l1 * sg + 2.f * p.l2 * xv[j];
          float gv = gs * (own - (rv[j] + dyv[j] + dx[j]));
          out[j] = gv;
          red += (double)gv * (double)gv;
        }
      }
      *reinterpret_cast<float4*>(p.out + idx) = make_float4(out[0], out[1], out[2], out[3]);
    }
  }
  if (p.mode == 1 && p.red) {
    // a wave holds 4 consecutive rows; 8 slots per layer spread the same-address atomic traffic (one word takes
    // ~88 atomics/us).  The 4 rows share a group (= vode) unless rows_per_group % 4 != 0 puts a group boundary inside
    // the wave: then each row's 16 lanes reduce and add on their own (wave-uniform choice).
    const long long wr0 = (long long)blockIdx.x * 16 + ((threadIdx.x >> 6) << 2);
    const long long wr3 = (wr0 + 3 < rows) ? wr0 + 3 : rows - 1;
    const int slot = (blockIdx.x * 4 + (threadIdx.x >> 6)) & 7;
    if (wr0 >= rows || wr0 / p.rows_per_group == wr3 / p.rows_per_group) {
      double tot = wave_sum_d(red);
      if ((threadIdx.x & 63) == 0 && live) atomicAdd(&p.red[((long long)g * p.red_stride) * 8 + slot], tot);
    } else {
      double tot = red;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
      if (sub == 0 && live) atomicAdd(&p.red[((long long)g * p.red_stride) * 8 + slot], tot);
    }
    (void)red_sh;
  }
}

int launch_ln_fwd(const float* x, const float* gamma, const float* beta, long long sParam, float* xn,
                  float* stats, int rows_per_group, int groups, int D, hipStream_t st) {
  long long rows = (long long)rows_per_group * groups;
  dim3 grid((unsigned)((rows + 15) / 16));
  int nv = (D + 63) / 64;
#define LNF(NVV) hipLaunchKernelGGL((ln_fwd_kernel<NVV>), grid, dim3(256), 0, st, x, gamma, beta, sParam, xn, stats, rows_per_group, rows, D)
  if (nv == 1) LNF(1); else if (nv == 2) LNF(2); else if (nv <= 4) LNF(4); else if (nv <= 6) LNF(6);
  else if (nv <= 12) LNF(12); else if (nv <= 16) LNF(16); else return -1;
#undef LNF
  return 0;
}

int launch_ln_bwd(const LnBwdP& p, hipStream_t st) {
  long long rows = (long long)p.rows_per_group * p.groups;
  dim3 grid((unsigned)((rows + 15) / 16));
  int nv = (p.D + 63) / 64;
#define LNB(NVV) hipLaunchKernelGGL((ln_bwd_kernel<NVV>), grid, dim3(256), 0, st, p)
  if (nv == 1) LNB(1); else if (nv == 2) LNB(2); else if (nv <= 4) LNB(4); else if (nv <= 6) LNB(6);
  else if (nv <= 12) LNB(12); else if (nv <= 16) LNB(16); else return -1;
#undef LNB
  return 0;
}

// ---- patchify: img [B,F,C,H,W] <-> xp [B, N, P], P index = (a*p + b)*C + c  ------------------
__global__ void patchify_kernel(const float* __restrict__ src, float* __restrict__ dst, long long total, int F,
                                int C, int Himg, int Wimg, int p, int to_image) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  // i indexes the image tensor [B,F,C,H,W]
  int wx = (int)(i % Wimg);
  long long t = i / Wimg;
  int hy = (int)(t % Himg); t /= Himg;
  int c = (int)(t % C); t /= C;
  int f = (int)(t % F);
  long long b = t / F;
  int hp = Himg / p, wp = Wimg / p;
  int pi = hy / p, a = hy % p, pj = wx / p, bb = wx % p;
  long long n = ((long long)f * hp + pi) * wp + pj;
  long long N = (long long)F * hp * wp;
  int P = p * p * C;
  long long j = (b * N + n) * P + (a * p + bb) * C + c;
  if (to_image) dst[i] = src[j]; else dst[j] = src[i];
}
void launch_patchify(const float* img, float* xp, int B, int F, int C, int Himg, int Wimg, int p, int to_image,
                     hipStream_t st) {
  long long total = (long long)B * F * C * Himg * Wimg;
  dim3 grid((unsigned)((total + 255) / 256));
  if (to_image) hipLaunchKernelGGL(patchify_kernel, grid, dim3(256), 0, st, xp, (float*)img, total, F, C, Himg, Wimg, p, 1);
  else hipLaunchKernelGGL(patchify_kernel, grid, dim3(256), 0, st, img, xp, total, F, C, Himg, Wimg, p, 0);
}

// ---- vode update (per-leaf multipliers: vode_mult.h) ------------------------------------------------
__global__ __launch_bounds__(256) void update_kernel(UpdateP p) {
  __shared__ float mult_sh[MULT_MAX_LAYERS + 1];
  __shared__ double ss_sh[MULT_MAX_LAYERS * 8], se_sh[8];
  const long long n4 = (p.n0 + (long long)(p.nlayers - 1) * p.nl) / 4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  constexpr int NS = 4;   // independent float4 triples in flight per thread (swept with the grid: 4 x 1024 blocks is the fastest at cfg 2)
  float4 g[NS], m[NS], h[NS];
  long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  auto fetch = [&]() {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const long long e = 4 * (i0 + u * stride);
      if (i0 + u * stride < n4) {
        g[u] = *reinterpret_cast<const float4*>(p.g + e);
        m[u] = *reinterpret_cast<const float4*>(p.m + e);
        h[u] = *reinterpret_cast<const float4*>(p.h + e);
      }
    }
  };
  // the first (usually only) batch of loads is issued BEFORE the per-leaf multipliers are worked out by
  // thread 0: they do not depend on them, and the block's start-up then costs one memory latency, not two
  fetch();
  stage_multiplier_inputs(p, ss_sh, se_sh);
  __syncthreads();
  if (threadIdx.x == 0) vode_multipliers(p, mult_sh, ss_sh, se_sh);
  __syncthreads();
  const float eta = p.eta[p.t];
  float* const hdst = p.h_out ? p.h_out : p.h;
  float* const mdst = p.m_out ? p.m_out : p.m;
  while (i0 < n4) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      if (i0 + u * stride >= n4) break;
      const long long e = 4 * (i0 + u * stride);
      const int l = (e < p.n0) ? 0 : 1 + (int)((e - p.n0) / p.nl);
      const float mul = mult_sh[l];
      m[u].x = g[u].x * mul + p.mu * m[u].x; m[u].y = g[u].y * mul + p.mu * m[u].y;
      m[u].z = g[u].z * mul + p.mu * m[u].z; m[u].w = g[u].w * mul + p.mu * m[u].w;
      h[u].x -= eta * m[u].x; h[u].y -= eta * m[u].y; h[u].z -= eta * m[u].z; h[u].w -= eta * m[u].w;
      if (p.g_out) *reinterpret_cast<float4*>(p.g_out + e) = g[u];
      *reinterpret_cast<float4*>(mdst + e) = m[u];
      *reinterpret_cast<float4*>(hdst + e) = h[u];
    }
    i0 += NS * stride;
    if (i0 < n4) fetch();
  }
}
__global__ __launch_bounds__(256) void sens_update_kernel(UpdateP p, float* __restrict__ hs, float* __restrict__ ms,
                                                          const float* __restrict__ e_img, long long n4) {
  __shared__ float mult_sh[MULT_MAX_LAYERS + 1];
  __shared__ double ss_sh[MULT_MAX_LAYERS * 8], se_sh[8];
  stage_multiplier_inputs(p, ss_sh, se_sh);
  __syncthreads();
  if (threadIdx.x == 0) vode_multipliers(p, mult_sh, ss_sh, se_sh);
  __syncthreads();
  const float mul = mult_sh[p.nlayers], eta = p.eta[p.t];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 g = reinterpret_cast<const float4*>(e_img)[i];
    float4 m = reinterpret_cast<float4*>(ms)[i];
    float4 h = reinterpret_cast<float4*>(hs)[i];
    m.x = g.x * mul + p.mu * m.x; m.y = g.y * mul + p.mu * m.y;
    m.z = g.z * mul + p.mu * m.z; m.w = g.w * mul + p.mu * m.w;
    h.x -= eta * m.x; h.y -= eta * m.y; h.z -= eta * m.z; h.w -= eta * m.w;
    reinterpret_cast<float4*>(ms)[i] = m;
    reinterpret_cast<float4*>(hs)[i] = h;
  }
}
__global__ __launch_bounds__(256) void sens_clamp_kernel(float* __restrict__ hs, const float* __restrict__ xt,
                                                         const float* __restrict__ mask, float alpha, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float cur = hs[i];
    float target = (mask[i] == 0.f) ? xt[i] : cur;
    hs[i] = alpha * target + (1.0f - alpha) * cur;
  }
}
void launch_update(const UpdateP& p, hipStream_t st) {
  long long n4 = (p.n0 + (long long)(p.nlayers - 1) * p.nl) / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 1024) blocks = 1024;   // four resident blocks per CU, each thread streams several batches
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(update_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
}

void launch_sens_update(const UpdateP& p, float* hs, float* ms, const float* e_img, long long n, hipStream_t st) {
  long long n4 = n / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sens_update_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, hs, ms, e_img, n4);
}
void launch_sens_clamp(float* hs, const float* xt, const float* mask, float alpha, long long n, hipStream_t st) {
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sens_clamp_kernel, dim3((unsigned)blocks), dim3(256), 0, st, hs, xt, mask, alpha, n);
}

// ---- fused path, pass 1 of the vode step (see VodeGradP) ---------------------------------------------
__device__ __forceinline__ float sgnf(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void vode_grad_kernel(VodeGradP p) {
  __shared__ double sh[4];
  const int v = blockIdx.y;
  const float s = p.gscale[v];
  const long long nl4 = p.nl / 4;
  const long long base = (long long)v * nl4;
  const bool reg = (p.l1 != 0.f || p.l2 != 0.f);
  const float4* __restrict__ bpp = reinterpret_cast<const float4*>(p.bp) + base;
  const float4* __restrict__ ep = reinterpret_cast<const float4*>(p.e) + base;
  const float4* __restrict__ hp = reinterpret_cast<const float4*>(p.h) + base;
  float4* gp = reinterpret_cast<float4*>(p.g) + base;
  double ss = 0.0;
  float en = 0.f;
  const long long stride = (long long)gridDim.x * blockDim.x;
  // NS independent float4 streams per thread and iteration keep enough loads in flight
  constexpr int NS = 4;
  for (long long k0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; k0 < nl4; k0 += NS * stride) {
    float4 bq[NS], eq[NS], hq[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const long long k = k0 + u * stride;
      if (k < nl4) { bq[u] = bpp[k]; eq[u] = ep[k]; if (reg) hq[u] = hp[k]; }
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const long long k = k0 + u * stride;
      if (k >= nl4) break;
      float bp[4] = {bq[u].x, bq[u].y, bq[u].z, bq[u].w}, e[4] = {eq[u].x, eq[u].y, eq[u].z, eq[u].w}, o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[j] = e[j]; en += 0.5f * e[j] * e[j]; }
      if (reg) {
        float h[4] = {hq[u].x, hq[u].y, hq[u].z, hq[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] += p.l1 * sgnf(h[j]) + 2.f * p.l2 * h[j];
          en += p.l1 * fabsf(h[j]) + p.l2 * h[j] * h[j];
        }
      }
      float4 gv = make_float4(s * (o[0] - bp[0]), s * (o[1] - bp[1]), s * (o[2] - bp[2]), s * (o[3] - bp[3]));
      gp[k] = gv;
      ss += (double)(gv.x * gv.x + gv.y * gv.y) + (double)(gv.z * gv.z + gv.w * gv.w);
    }
  }
  // both sums through the wave shuffles together, one barrier for the pair
  __shared__ double sh2[8];
  double a = wave_sum_d(ss), b = wave_sum_d((double)en);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh2[w] = a; sh2[4 + w] = b; }
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&p.sumsq[v * 8 + (blockIdx.x & 7)], sh2[0] + sh2[1] + sh2[2] + sh2[3]);
  if (threadIdx.x == 64) atomicAdd(&p.esum[v * 8 + (blockIdx.x & 7)], sh2[4] + sh2[5] + sh2[6] + sh2[7]);
  (void)sh;
}
void launch_vode_grad(const VodeGradP& p, hipStream_t st) {
  long long nl4 = p.nl / 4;
  long long bx = (nl4 + 4095) / 4096;   // four float4 per thread and iteration, four iterations per thread (swept: 14.5 us against 16.1 at one)
  if (bx > 1024) bx = 1024;
  if (bx < 1) bx = 1;
  if (bx * p.nlay < 256) {   // small batches: fill the chip (>= 2 blocks per CU) before giving a thread a second iteration
    const long long fine = (nl4 + 255) / 256, want = (512 + p.nlay - 1) / p.nlay;
    bx = fine < want ? fine : want;
    if (bx < 1) bx = 1;
  }
  hipLaunchKernelGGL(vode_grad_kernel, dim3((unsigned)bx, p.nlay), dim3(256), 0, st, p);
}

// ---- column sums (bias / LayerNorm parameter gradients) ------------------------------------------
// out[col] += alpha * sum_rows a[row][col] * (b ? bhat[row][col] : 1), bhat = b or its normalised value;
// out2 (optional, with b): the plain column sums of a in the same pass (d gamma and d beta of one LayerNorm
// read the cotangent once).  grid: (ceil(cols/64), row_chunks, groups); block 256 = 16 row lanes x 16 column
// quads: float4 loads, four rows in flight per thread (the scalar, one-row-at-a-time form ran at 2.9 TB/s).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ stats, float* __restrict__ out,
                                                     float* __restrict__ out2, long long sOut, int rows_per_group,
                                                     int cols, float alpha, int rows_per_chunk) {
  __shared__ float4 sh[16][16];
  __shared__ float4 sh2[16][16];
  const int cq = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int col = blockIdx.x * 64 + 4 * cq;
  const int g = blockIdx.z;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(rows_per_group, r0 + rows_per_chunk);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), s2 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < cols) {     // cols % 4 == 0 (launcher): a quad is inside or outside as a whole
    const float* ap = a + (long long)g * rows_per_group * cols + col;
    const float* bp = b ? b + (long long)g * rows_per_group * cols + col : nullptr;
    const float* sp = stats ? stats + 2LL * g * rows_per_group : nullptr;
    for (int rb = r0 + ry; rb < r1; rb += 64) {
      float4 av[4], bv[4];
      float mu[4], rs[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = rb + 16 * u;
        const bool ok = rr < r1;
        av[u] = ok ? *reinterpret_cast<const float4*>(ap + (long long)rr * cols) : make_float4(0.f, 0.f, 0.f, 0.f);
        bv[u] = (ok && bp) ? *reinterpret_cast<const float4*>(bp + (long long)rr * cols) : make_float4(1.f, 1.f, 1.f, 1.f);
        mu[u] = (ok && sp) ? sp[2 * rr] : 0.f;
        rs[u] = (ok && sp) ? sp[2 * rr + 1] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s2.x += av[u].x; s2.y += av[u].y; s2.z += av[u].z; s2.w += av[u].w;
        s.x += av[u].x * ((bv[u].x - mu[u]) * rs[u]);
        s.y += av[u].y * ((bv[u].y - mu[u]) * rs[u]);
        s.z += av[u].z * ((bv[u].z - mu[u]) * rs[u]);
        s.w += av[u].w * ((bv[u].w - mu[u]) * rs[u]);
      }
    }
  }
  sh[ry][cq] = s;
  sh2[ry][cq] = s2;
  __syncthreads();
  if (ry == 0 && col < cols) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f), t2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      t.x += sh[k][cq].x; t.y += sh[k][cq].y; t.z += sh[k][cq].z; t.w += sh[k][cq].w;
      t2.x += sh2[k][cq].x; t2.y += sh2[k][cq].y; t2.z += sh2[k][cq].z; t2.w += sh2[k][cq].w;
    }
    float* o = out + (long long)g * sOut + col;
    atomicAdd(o + 0, alpha * t.x); atomicAdd(o + 1, alpha * t.y); atomicAdd(o + 2, alpha * t.z); atomicAdd(o + 3, alpha * t.w);
    if (out2) {
      float* o2 = out2 + (long long)g * sOut + col;
      atomicAdd(o2 + 0, alpha * t2.x); atomicAdd(o2 + 1, alpha * t2.y); atomicAdd(o2 + 2, alpha * t2.z); atomicAdd(o2 + 3, alpha * t2.w);
    }
  }
}
void launch_colsum(const float* a, const float* b, const float* stats, float* out, long long sOut,
                   int rows_per_group, int groups, int cols, float alpha, hipStream_t st, float* out2) {
  int chunk = 256;
  dim3 grid((cols + 63) / 64, (rows_per_group + chunk - 1) / chunk, groups);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, st, a, b, stats, out, out2, sOut, rows_per_group, cols, alpha, chunk);
}

// ---- norms, AdamW, MSE -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ a, long long n, float scale, double* out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    double v = (double)(a[i] * scale);
    s += v * v;
  }
  double tot = block_sum_256(s, sh);
  if (threadIdx.x == 0) atomicAdd(out, tot);
}
void launch_sumsq(const float* a, long long n, float scale, double* out, hipStream_t st) {
  long long blocks = (n + 255) / 256;
  if (blocks > 128) blocks = 128;   // one f64 atomic per block on ONE address: keep the fan-in small
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, n, scale, out);
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamP p) {
  float cm = 1.f;
  if (p.clip > 0.f) {
    double gn = sqrt(p.sumsq[0]);
    if (!(gn < (double)p.clip)) cm = (float)((double)p.clip / gn);
  }
  const float gmul = p.gscale * cm;
  if (p.hyp) { p.lr = p.hyp[0]; p.bc1 = p.hyp[1]; p.bc2 = p.hyp[2]; }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (long long)gridDim.x * blockDim.x) {
    float g = p.g[i] * gmul;
    float mu = p.b1 * p.mu[i] + (1.f - p.b1) * g;
    float nu = p.b2 * p.nu[i] + (1.f - p.b2) * g * g;
    p.mu[i] = mu;
    p.nu[i] = nu;
    float mh = mu / p.bc1, nh = nu / p.bc2;
    float w = p.w[i];
    float upd = mh / (sqrtf(nh) + p.eps) + p.wd * w;
    p.w[i] = w - p.lr * upd;
  }
}
void launch_adamw(const AdamP& p, hipStream_t st) {
  long long blocks = (p.n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
}

__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                  double* out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float x = fminf(fmaxf(a[i], 0.f), 1.f), y = fminf(fmaxf(b[i], 0.f), 1.f);
    double d = (double)(x - y);
    s += d * d;
  }
  double tot = block_sum_256(s, sh);
  if (threadIdx.x == 0) atomicAdd(out, tot);
}
void launch_mse(const float* a, const float* b, long long n, double* out, hipStream_t st) {
  long long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(mse_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, b, n, out);
}

// ---- zero fill as a kernel (used inside captured graphs instead of memset nodes) --------------------
__global__ __launch_bounds__(256) void zero_kernel(float4* __restrict__ p, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
    p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
void launch_zero(void* ptr, size_t bytes, hipStream_t st) {   // bytes % 16 == 0, ptr 16-byte aligned
  long long n4 = (long long)(bytes / 16);
  long long blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (float4*)ptr, n4);
}

// ---- bf16 copies of a weight arena (bf16 mode, gemm_bf16.hip) ---------------------------------------
// hi = bf16(x) (round to nearest even), lo = bf16(x - hi): the same two conversions the GEMM kernel applies
// to an operand it converts itself while staging, so both forms give bit-identical products.
__global__ __launch_bounds__(256) void cvt_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ hi,
                                                       unsigned short* __restrict__ lo, unsigned short* __restrict__ t3, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float x = src[i];
    const __bf16 h = (__bf16)x;
    hi[i] = __builtin_bit_cast(unsigned short, h);
    if (lo) {
      const __bf16 l = (__bf16)(x - (float)h);
      lo[i] = __builtin_bit_cast(unsigned short, l);
      if (t3) {
        const __bf16 t = (__bf16)((x - (float)h) - (float)l);
        t3[i] = __builtin_bit_cast(unsigned short, t);
      }
    }
  }
}
void launch_cvt_bf16(const float* src, unsigned short* hi, unsigned short* lo, long long n, hipStream_t st, unsigned short* t3) {
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cvt_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, hi, lo, t3, n);
}
