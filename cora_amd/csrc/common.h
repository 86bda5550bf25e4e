// common.h -- shared device helpers and kernel parameter blocks (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CORA_LOG2E 1.4426950408889634f

// ---- tanh-GELU (jax.nn.gelu approximate=True, pcx/nn/_layer.py:439) -------------------------
// 0.5 a (1 + tanh(u)) = a * sigmoid(2u),  u = c0 (a + c1 a^3).  Written on v_exp_f32 / v_rcp_f32
// (1 ulp each) with no IEEE division: 7 VALU ops forward, 12 for the derivative.
//   s = sigmoid(2u) = 1 - 1/(1 + e),  e = exp(2u)       (e = inf -> s = 1, e = 0 -> s = 0)
//   gelu'(a) = s + a s (1 - s) * 2 c0 (1 + 3 c1 a^2)
#define CORA_GELU_K0 2.3022081981f    /* 2 * log2(e) * sqrt(2/pi) */
#define CORA_GELU_K1 0.1029432396f    /* K0 * 0.044715 */
__device__ __forceinline__ float cora_gelu(float a) {
  float a2 = a * a;
  float e = __builtin_amdgcn_exp2f(a * __builtin_fmaf(CORA_GELU_K1, a2, CORA_GELU_K0));
  float r = __builtin_amdgcn_rcpf(1.0f + e);
  return __builtin_fmaf(-a, r, a);
}
__device__ __forceinline__ float cora_gelu_grad(float a) {
  float a2 = a * a;
  float e = __builtin_amdgcn_exp2f(a * __builtin_fmaf(CORA_GELU_K1, a2, CORA_GELU_K0));
  float r = __builtin_amdgcn_rcpf(1.0f + e);      // 1 - s
  float s_ = 1.0f - r;
  float du = __builtin_fmaf(0.2140644488f, a2, 1.5957691216f);   // 2 c0 (1 + 3 c1 a^2)
  return __builtin_fmaf(a * s_ * r, du, s_);
}

// both at once (the inference kernel keeps gelu'(a) for its backward instead of a: one exp2 / rcp per value)
__device__ __forceinline__ void cora_gelu_both(float a, float& g, float& gp) {
  float a2 = a * a;
  float e = __builtin_amdgcn_exp2f(a * __builtin_fmaf(CORA_GELU_K1, a2, CORA_GELU_K0));
  float r = __builtin_amdgcn_rcpf(1.0f + e);      // 1 - s
  float s_ = 1.0f - r;
  float du = __builtin_fmaf(0.2140644488f, a2, 1.5957691216f);
  g = __builtin_fmaf(-a, r, a);
  gp = __builtin_fmaf(a * s_ * r, du, s_);
}

// two values per call on the packed fp32 pipe (v_pk_mul_f32 / v_pk_fma_f32); exp2 / rcp stay scalar
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cora_gelu_both2(f32x2 a, f32x2& g, f32x2& gp) {
  const f32x2 a2 = a * a;
  const f32x2 u = a * (a2 * CORA_GELU_K1 + CORA_GELU_K0);
  f32x2 r;
  r.x = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u.x));
  r.y = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u.y));
  const f32x2 s_ = 1.0f - r;
  const f32x2 du = a2 * 0.2140644488f + 1.5957691216f;
  g = a - a * r;
  gp = (a * s_ * r) * du + s_;
}

// ---- wave (64 lanes) and block reductions ------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over a 256-thread block; result valid in thread 0.  `sh` = 4 doubles of LDS.
__device__ __forceinline__ double block_sum_256(double v, double* sh) {
  v = wave_sum_d(v);
  int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) r = sh[0] + sh[1] + sh[2] + sh[3];
  return r;
}

// Kernel attributes (dynamic LDS above 48 KB) are per device: true the first time `flags` is asked on
// the current device, so a process that drives several GPUs sets them on each.
inline bool first_use_on_device(bool (&flags)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  if (flags[dev]) return false;
  flags[dev] = true;
  return true;
}

// ---- GEMM parameter block ----------------------------------------------------------------------
enum GemmOut {
  OUT_STORE = 0,     // C = v
  OUT_ERR = 1,       // e = tgt - v ; C = e ; red += 1/2 e^2 + l1|tgt| + l2 tgt^2
  OUT_GELUGRAD = 2,  // C = v * gelu'(aux)
  OUT_GRAD = 3,      // g = alpha2 * (own + l1 sign(hreg) + 2 l2 hreg - v) ; C = g ; red += g^2
  OUT_ATOMIC = 4     // atomicAdd(C, v)
};

struct GemmP {
  const float* A;   // operand rows x K   (K-inner)  or K x rows (K-outer, "TN")
  const float* Bw;  // operand cols x K   (K-inner, "NT": weight [out,in]) or K x cols (K-outer)
  float* C;
  int R, O, K;           // rows, cols, reduction length (per group)
  int lda, ldb, ldc;
  long long sA, sB, sC;  // per-group element strides
  int groups, splitk;
  int a_gelu;            // apply tanh-GELU to A while staging (MLP2 reads the pre-activation)
  int b_gelu;            // same for the B operand (weight-gradient of linear2)
  int bf16;              // opt-in: 1 = operands rounded to bf16 in registers, bf16 MFMA, fp32 accumulate;
                         // 2 = two-term bf16 split of both operands, three bf16 MFMAs (near-fp32 products);
                         // 3 = three-term split, six MFMAs (fp32-grade products; bf16-staged NT kernel only --
                         //     every other operand form runs the fp32 kernel)
  float alpha;           // v = alpha * acc (+ bias + pos + res1 + res2)
  const float* bias; long long sBias;
  const float* pos; int posN;
  const float* res1; const float* res2;   // same shape / strides as C
  int mode;
  const float* tgt;      // OUT_ERR target (h of the predicted vode); OUT_GRAD: own error (may be null)
  const float* hreg;     // OUT_GRAD: h for the regulariser terms (may be null)
  const float* aux;      // OUT_GELUGRAD: pre-activation a   (ld = ldc)
  float alpha2, l1, l2;
  double* red; int red_stride;   // red[g * red_stride] accumulates energy / sum of squares
  // bf16 modes: the B operand (a weight) already converted to bf16 -- hi, second term (split forms) and third
  // term (three-term form) -- indexed like Bw (ctx.hip keeps the copies fresh); null -> B is converted from
  // Bw while staging
  const unsigned short* Bh; const unsigned short* Bl; const unsigned short* B3;
  // bf16 mode, backward-data scratch kept as bf16 (ctx.hip: block_backward): a16 -- A points at bf16 elements (same lda / sA,
  // in elements; plain form with pre-converted B only: the kernel would round A to bf16 anyway, so the products are
  // bit-identical); c16 -- OUT_GELUGRAD stores C as bf16 (the next GEMM's a16 operand)
  int a16, c16;
};

void launch_gemm(const GemmP& p, bool a_kouter, bool b_kouter, hipStream_t st);
// bf16 mode, both operands K-inner (gemm_bf16.hip): operands rounded to bf16 while staging into LDS
bool gemm_bf16_nt_supported(const GemmP& p);
void launch_gemm_bf16_nt(const GemmP& p, hipStream_t st);
#define GEMM_MULTI_MAX 6
struct GemmMulti { int n; int start[GEMM_MULTI_MAX + 1]; GemmP p[GEMM_MULTI_MAX]; };
// n independent K-outer x K-outer (TN) problems in one launch; start[] is filled by the launcher
void launch_gemm_multi_tn(const GemmMulti& m, hipStream_t st);

// ---- attention ------------------------------------------------------------------------------------
// qkv: [rows, 3D] (q | k | v, head h = columns h*dh..), O: [rows, D], lse2: [rows, H] (base-2 LSE)
int launch_attn_fwd(const float* qkv, float* O, float* lse2, int nimg, int N, int H, int D,
                    hipStream_t st);
// dqkv: [rows, 3D]; delta: [rows, H] scratch
int launch_attn_bwd(const float* qkv, const float* O, const float* lse2, const float* dO,
                    float* delta, float* dqkv, int nimg, int N, int H, int D, hipStream_t st);
// single-pass backward (N % 64 == 0, N > 64, head dim 16 / 32 / 64): delta [rows, H] is an INPUT (rowsum(dO * O), left by
// rows_mid); dq leaves as attn_bwd_fused_parts(N) partial copies [rows, D] at dqa + k * pstride (their sum is dq: one per
// block of 128 own keys, plain stores, deterministic); dqkv receives dk | dv
bool attn_bwd_fused_supported(int N, int H, int D);
int attn_bwd_fused_parts(int N);
int launch_attn_bwd_fused(const float* qkv, const float* lse2, const float* dO, const float* delta, float* dqa, long long pstride,
                          float* dqkv, int nimg, int N, int H, int D, hipStream_t st);
// bf16 mode (attn_bf16.hip): the same backward on bf16 MFMA; -1 (nothing launched) where it does not apply
bool attn_bwd_bf16_supported(int N, int H, int D);
bool attn_fwd_bf16_supported(int N, int H, int D);
int launch_attn_fwd_bf16(const float* qkv, float* O, float* lse2, int nimg, int N, int H, int D, hipStream_t st);
int launch_attn_bwd_bf16(const float* qkv, const float* O, const float* lse2, const float* dO,
                         float* delta, float* dqkv, int nimg, int N, int H, int D, hipStream_t st,
                         int out16 = 0 /* dqkv written as bf16 (same element offsets): the next GEMM's GemmP::a16 operand */);

// ---- row-wise / element-wise ---------------------------------------------------------------------
int launch_ln_fwd(const float* x, const float* gamma, const float* beta, long long sParam,
                  float* xn, float* stats, int rows_per_group, int groups, int D, hipStream_t st);
// mode 2: out = r + dy + LNbwd(dxn)   (raw cotangent, no assembly; r / dy may be null in mode 1)
// mode 0 (LN2): out = r + LNbwd(dxn)
// mode 1 (LN1): dxt = r + dy + LNbwd(dxn);  out = gscale[g]*(own + l1 sign(x) + 2 l2 x - dxt);
//               red[(g) * red_stride] += out^2
struct LnBwdP {
  const float* dxn; const float* x; const float* stats; const float* gamma; long long sParam;
  const float* r; const float* dy; const float* own;
  float* out; int rows_per_group, groups, D, mode;
  const float* gscale; float l1, l2; double* red; int red_stride;
};
int launch_ln_bwd(const LnBwdP& p, hipStream_t st);

void launch_patchify(const float* img, float* xp, int B, int F, int C, int Himg, int Wimg, int p,
                     int to_image, hipStream_t st);
// vode update: scale/clip from sumsq, m = g*mult + mu*m, h -= eta*m
struct UpdateP {
  float* h; float* m; const float* g; long long n0, nl; int nlayers;  // layer 0 has n0 floats, others nl
  const double* sumsq;  // [nlayers] sums of squares of the (scaled) grads of this step
  float clip, mu; const float* eta; int t;
  int use_grad_norm; float grad_norm_target;
  // un-frozen sensory vode (inpainting loop, src/decoder_transformer.py:904-906): one more leaf in
  // the global norm.  Its gradient is g_s = sens_s * e_img, so ||g_s||^2 = 2 sens_s^2 * (1/2 sum e_img^2),
  // and the sensory energy is already reduced into sens_e[0..sens_n).  Null: sensory vode frozen.
  const double* sens_e; int sens_n; float sens_s;
  float* h_out; float* m_out;   // null: in place; else the updated vodes / momentum go here (launch_update only)
  float* g_out;                 // not null: the gradients are copied here on the way through
};
void launch_update(const UpdateP& p, hipStream_t st);
// sensory state (patch layout, n floats): m_s = c g_s + mu m_s ; h_s -= eta m_s, same multipliers as launch_update
void launch_sens_update(const UpdateP& p, float* hs, float* ms, const float* e_img, long long n, hipStream_t st);
// soft clamp of the known pixels, :982-994: h_s = mask == 0 ? alpha x + (1 - alpha) h_s : h_s
void launch_sens_clamp(float* hs, const float* xt, const float* mask, float alpha, long long n, hipStream_t st);
// Fused path, pass 1 of the vode step: g_v = s_v (e_v + l1 sign(h_v) + 2 l2 h_v - bp_v) for vodes 1..L+1,
// its sum of squares (8 slots per vode) and the vode energies 1/2 e^2 + l1 |h| + l2 h^2 (8 slots per vode).
// bp and g may be the same array.  launch_update then applies clip + momentum SGD.
struct VodeGradP {
  const float* bp; const float* e; const float* h;   // [nlay, nl] each: J^T e, e and h of vodes 1..
  float* g;                                          // [nlay, nl] out
  const float* gscale; long long nl; int nlay;
  float l1, l2;
  double* sumsq;   // slots of vode 1 (vode 0's are filled by the edge path)
  double* esum;    // slots of vode 1
};
void launch_vode_grad(const VodeGradP& p, hipStream_t st);
// column sums: out[g*sOut + c] += alpha * sum_rows a[row, c] * (b ? b'[row, c] : 1), where
// b' = b, or (b - mean[row]) * rstd[row] when stats ([rows,2]) is given; out2 (optional): the plain column
// sums of a from the same pass.  cols % 4 == 0.
void launch_colsum(const float* a, const float* b, const float* stats, float* out, long long sOut,
                   int rows_per_group, int groups, int cols, float alpha, hipStream_t st, float* out2 = nullptr);
void launch_sumsq(const float* a, long long n, float scale, double* out, hipStream_t st);
struct AdamP {
  float* w; const float* g; float* mu; float* nu; long long n;
  const double* sumsq; float gscale, clip, lr, b1, b2, eps, wd; float bc1, bc2;
  const float* hyp;   // if set: {lr, bc1, bc2} read from device memory (the captured learning graph)
};
void launch_adamw(const AdamP& p, hipStream_t st);
// red[0] += sum (clip01(a)-clip01(b))^2
void launch_mse(const float* a, const float* b, long long n, double* out, hipStream_t st);
void launch_zero(void* ptr, size_t bytes, hipStream_t st);
// bf16 copies of n floats: hi = bf16(x), lo = bf16(x - hi), t3 = bf16(x - hi - lo) (lo, t3 may be null)
void launch_cvt_bf16(const float* src, unsigned short* hi, unsigned short* lo, long long n, hipStream_t st,
                     unsigned short* t3 = nullptr);

// ---- fused per-(block, image) kernel (N = 64, D = 64, H = 1) -----------------------------------------
struct FusedP {
  const float* H1; float* E2; float* BP1; const float* W; const float* WT;
  long long oLn1g, oLn1b, oLn2g, oLn2b, oWqkv, oWo, oW1, ob1, oW2, ob2;
  long long tWqkv, tWo, tW1, tW2;
  double* energy;   // STORE pass: per-vode energies accumulated here directly (null in the inference step)
  double* esens;    // inference step: 8 slots for the sensory energy (the vode-step kernel derives the rest from e)
  int B; float l1, l2;
  int H;            // heads: 1, 2 or 4 (head dim 64, 32, 16)
  // edge layers (vode 0 / 1 and vode L+1 / sensory), one extra workgroup per image
  int L; float s0;
  const float* H0; const float* XP; const float* pos;
  float* E1; float* EIMG; float* G0; double* sumsq0;
  long long oWp, obp, oWout, obout, tWp, tWout;
  // MODE_STORE: tensors the weight-gradient GEMMs read ([L*B*N, cols], same layout as the general path)
  float *S_xn1, *S_att, *S_xn2, *S_a, *S_da, *S_dy, *S_dqkv;
  float* S_stash;   // [L*B*N, H D]: P (one 64-key row per head) parked between forward and backward
  float* dW;        // MODE_STORE: bias / LayerNorm-parameter gradients land here (null otherwise)
  float* colpart;   // MODE_STORE: per-wave column sums of the block workgroups (fused_colpart_floats)
  // chain kernel: z [B*N, P] and the clipped squared reconstruction error
  float* Z; double* mse;
  unsigned long long* dbg;   // diagnostic builds only (-DCORA_STAMPS)
  // lazy != 0 (inference step t + 1 of the D = 64 path): the vode update of step t is applied as the rows are loaded --
  // h' = h - eta (mult g + mu m) from the OLD arenas (H1 / H0, Mo1 / M0o, Go1 / G0o; up = the parameters and the
  // sums-of-squares row of step t), and the workgroup that reads a vode as its INPUT writes h', m' to the NEW arenas
  // (Hn1 / H0n, Mn1 / M0n); BP1 / G0 point into the new gradient arena.  Saves the update kernel's launch and hides its
  // HBM traffic under the MFMA-bound block kernel.
  int lazy;
  const float* Mo1; const float* Go1; float* Hn1; float* Mn1;
  const float* M0o; const float* G0o; float* H0n; float* M0n;
  UpdateP up;
};
int launch_fused_block(const FusedP& p, int L, int store, hipStream_t st);
// CORA_PAIR=1: the experimental 8-wave form of the inference kernel (fused_block.hip, block_body_pair); off by default
bool fused_pair_wanted(int L, int B);
int launch_chain(const FusedP& p, int store_h, hipStream_t st);
// wide variant (fused_wide.hip): N = 64, D = 96 / 128, inference step only; uses H1, E2, BP1, W, WT, S_stash, S_da
bool fused_wide_supported(int N, int P, int D, int H, int M);
int launch_fused_wide(const FusedP& p, int L, int D, hipStream_t st);
// two workgroups per CU (fused_wide2.hip): D = 128 (4 / 2 / 1 heads), D = 96 (2 / 1 heads); also uses S_dqkv as a dk | dv stash
bool fused_wide2_supported(int N, int P, int D, int H, int M);
int launch_fused_wide2(const FusedP& p, int L, int D, hipStream_t st);
// status-"forward" walk of one image per workgroup through all L blocks (D = 128, heads of 32); in place when
// stride == 0 and xin == xout, else block g's output goes to xout + g * stride
int launch_chain_wide2(const FusedP& p, int L, int D, const float* xin, float* xout, long long stride, hipStream_t st);
// the same for every shape fused_wide_supported() takes (one workgroup per CU form)
int launch_chain_wide(const FusedP& p, int L, int D, const float* xin, float* xout, long long stride, hipStream_t st);
// ---- row-tile fused kernels (fused_rows.hip): D = 64, M = 256, any sequence length ---------------------------
// One workgroup per 64-row tile of one block; attention (attn.hip) runs between rows_qkv / rows_mid / rows_out.
struct RowsP {
  const float* X;        // vodes 1..L+1, [L+1][rows][64]: block g reads X + g nl and predicts X + (g + 1) nl
  float* E2;             // errors of vodes 2..L+1: e_{g+2} -> E2 + g nl
  const float* E1;       // errors of vodes 1..L (own error of the vode block g reads): E1 + g nl
  float* G1;             // gradients of vodes 1..L: g_{g+1} -> G1 + g nl
  const float* W; const float* WT;
  long long oLn1g, oLn1b, oLn2g, oLn2b, oWqkv, oWo, oW1, ob1, oW2, ob2, tWqkv, tWo, tW1, tW2;
  float* qkv; const float* att; float* dO; float* rdy; float* dqkv;   // [L rows, 3D | D | D | D | 3D]
  // edge != 0 (inference launches, P = 48): rows / 64 extra workgroups do the top edge (in rows_qkv: e_1, energy of vode 1
  // at energy[-1], g_0 and its sum of squares at sumsq[-8 ..]) and the bottom edge (in rows_out: e_img, sensory energy at
  // esens, g_{L+1} and its sum of squares at sumsq[8 L ..]) instead of four per-op GEMM launches
  int edge; int N; float s0;
  const float* H0; const float* XP; const float* pos; float* EIMG; float* G0; double* esens;
  long long oWp, obp, oWout, obout, tWp, tWout;
  float* delta; int H;   // not null: rows_mid also leaves delta[row, head] = sum over the head's columns of dO * att
  const float* dqa; int ndq; long long dq_stride;   // not null: rows_out sums dq from ndq partial copies [L rows, D], dq_stride apart
  long long rows;        // B * N token rows per block (multiple of 64)
  int L;
  double* energy;        // energy[g] += energy of vode g + 2
  double* sumsq;         // sumsq[g * 8 + slot] += |g_{g+1}|^2
  const float* gscale; float l1, l2;
  // store pass (learning phase): operands of the weight-gradient contractions + per-wave column sums
  float *S_xn1, *S_xn2, *S_a, *S_da, *S_dy, *colpart;
};
bool fused_rows_supported(long long rows, int D, int M);
size_t fused_rows_colpart_floats(int L, long long rows);
// store: 0 inference step, 1 learning-phase store pass, 2 one block of the status-"forward" walk (rows_qkv and rows_mid
// only: L = 1, X = the block's input rows, E2 = where its output rows go, weight offsets advanced to the block)
int launch_rows_qkv(const RowsP& p, int store, hipStream_t st);
int launch_rows_mid(const RowsP& p, int store, hipStream_t st);
int launch_rows_out(const RowsP& p, int store, hipStream_t st);
void launch_rows_colpart_reduce(const RowsP& p, float* dW, hipStream_t st);   // after the three store launches
void launch_colpart_reduce(const FusedP& p, int L, hipStream_t st);   // after a MODE_STORE launch
size_t fused_colpart_floats(int L, int B);
void launch_transpose(const float* src, float* dst, int rows, int cols, int groups, hipStream_t st);
struct TransposeSet { int n; const float* src[6]; float* dst[6]; int rows[6], cols[6], groups[6], zstart[6]; };
void launch_transpose_set(const TransposeSet& t, hipStream_t st);   // all matrices of the set in one launch
