// vode_mult.h -- per-leaf multipliers of the vode update (optional per-leaf normalisation, then the global-norm clip):
// shared by update_kernel / sens_update_kernel (rowwise.hip) and the inference kernel that applies the previous step's
// update while it loads its rows (fused_block.hip, MODE_LAZY).
// /root/reference/examples/debug_transformer_wandb.py:1137-1140 (optax.clip_by_global_norm over all vode leaves),
// /root/reference/src/decoder_transformer.py:353-389 (normalize_vode_h_gradients)
#pragma once
#include "common.h"

// per-leaf multipliers: optional per-leaf normalisation, then the global-norm clip over all leaves
// (mult[nlayers] is the un-frozen sensory leaf's, layer scale included).  One thread, reading the slots from the
// block's LDS copy (ss: [nlayers][8], se: [sens_n]) -- straight from global memory the layer loop was a chain of
// nlayers dependent L2 round trips at the head of every block (a third of the kernel at small batches).
constexpr int MULT_MAX_LAYERS = 65;
__device__ inline void stage_multiplier_inputs(const UpdateP& p, double* ss, double* se) {
  for (int j = threadIdx.x; j < p.nlayers * 8; j += blockDim.x) ss[j] = p.sumsq[j];
  if (p.sens_e && (int)threadIdx.x < p.sens_n) se[threadIdx.x] = p.sens_e[threadIdx.x];
}
__device__ inline void vode_multipliers(const UpdateP& p, float* mult, const double* ss, const double* se) {
  double tot = 0.0;
  for (int l = 0; l < p.nlayers; ++l) {
    double n2 = 0.0;
    for (int k = 0; k < 8; ++k) n2 += ss[l * 8 + k];
    float m = 1.f;
    if (p.use_grad_norm) {
      double nrm = sqrt(n2);
      m = (float)((double)p.grad_norm_target / (nrm + 1e-8));
      n2 = n2 * (double)m * (double)m;
    }
    mult[l] = m;
    tot += n2;
  }
  mult[p.nlayers] = 0.f;
  if (p.sens_e) {
    double es = 0.0;
    for (int k = 0; k < p.sens_n; ++k) es += se[k];
    double n2 = 2.0 * (double)p.sens_s * (double)p.sens_s * es;
    float m = 1.f;
    if (p.use_grad_norm) {
      double nrm = sqrt(n2);
      m = (float)((double)p.grad_norm_target / (nrm + 1e-8));
      n2 = n2 * (double)m * (double)m;
    }
    mult[p.nlayers] = m * p.sens_s;
    tot += n2;
  }
  float cm = 1.f;
  if (p.clip > 0.f) {
    double gn = sqrt(tot);
    if (!(gn < (double)p.clip)) cm = (float)((double)p.clip / gn);
  }
  for (int l = 0; l <= p.nlayers; ++l) mult[l] *= cm;
}


// The same multipliers for up to three leaves (la, lb, lc: wave-uniform), computed by EVERY wave on its own with no LDS
// and no barrier: lane = (layer of the current chunk of 8, slot), one load per lane, 8-lane segmented sums by shuffles.
// For the kernels that apply the update on the way in (FusedP::lazy): there the staged form cost two barriers and a
// serial thread at the head of every workgroup.  No sensory leaf (those launches are excluded when it is un-frozen).
// Sums run as trees here and serially in vode_multipliers: the doubles may differ in the last bit.
// first0: the caller's own early load of chunk 0 (vode_multipliers_first_load), so that this wave's wait for it does not
// also wait for every row load issued in between (vector memory returns in order)
__device__ inline double vode_multipliers_first_load(const UpdateP& p) {
  const int lane = threadIdx.x & 63, l = lane >> 3;
  return (l < p.nlayers) ? p.sumsq[l * 8 + (lane & 7)] : 0.0;
}
__device__ inline void vode_multipliers_wave(const UpdateP& p, int la, int lb, int lc, float& ca, float& cb, float& cc, double first0) {
  const int lane = threadIdx.x & 63;
  double tot = 0.0;
  float ma = 1.f, mb = 1.f, mc = 1.f;
  for (int l0 = 0; l0 < p.nlayers; l0 += 8) {
    const int l = l0 + (lane >> 3);
    double v = first0;
    if (l0 > 0) v = (l < p.nlayers) ? p.sumsq[l * 8 + (lane & 7)] : 0.0;
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);   // n2 of layer l in its 8 lanes
    float m = 1.f;
    if (p.use_grad_norm) {
      double nrm = sqrt(v);
      m = (float)((double)p.grad_norm_target / (nrm + 1e-8));
      v = v * (double)m * (double)m;
    }
    double cs = v;
    cs += __shfl_xor(cs, 8, 64); cs += __shfl_xor(cs, 16, 64); cs += __shfl_xor(cs, 32, 64);   // over the chunk's layers
    tot += cs;
    if ((la >> 3) == (l0 >> 3)) ma = __shfl(m, 8 * (la & 7), 64);
    if ((lb >> 3) == (l0 >> 3)) mb = __shfl(m, 8 * (lb & 7), 64);
    if ((lc >> 3) == (l0 >> 3)) mc = __shfl(m, 8 * (lc & 7), 64);
  }
  float cm = 1.f;
  if (p.clip > 0.f) {
    double gn = sqrt(tot);
    if (!(gn < (double)p.clip)) cm = (float)((double)p.clip / gn);
  }
  ca = ma * cm; cb = mb * cm; cc = mc * cm;
}
__device__ inline void vode_multipliers_wave(const UpdateP& p, int la, int lb, int lc, float& ca, float& cb, float& cc) {
  vode_multipliers_wave(p, la, lb, lc, ca, cb, cc, vode_multipliers_first_load(p));
}
