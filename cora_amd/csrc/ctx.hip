// ctx.hip -- context, memory layout, step orchestration, hipGraph capture and the C ABI
// (include/cora_hip.h) of libcora_hip.so.  Host code only; kernels live in gemm/attn/rowwise.hip.
//
// Step order follows train_on_batch, /root/reference/src/decoder_transformer.py:392-499 (mode 2).
#include "../../include/cora_hip.h"
#include "common.h"

#include <rccl/rccl.h>
#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_create_err;

enum Kind { K_GEMM_NT = 0, K_GEMM_NN, K_GEMM_TN, K_ATTN_FWD, K_ATTN_BWD, K_LN_FWD, K_LN_BWD, K_UPDATE, K_OTHER, K_FUSED, K_GRADFIN, K_ROWS, K_COUNT };
const char* kKindNames[K_COUNT] = {"gemm_nt", "gemm_nn", "gemm_tn", "attn_fwd", "attn_bwd",
                                   "ln_fwd", "ln_bwd", "vode_update", "other", "pc_block_fused", "vode_grad", "pc_rows_fused"};

struct ProfRec { int kind; hipEvent_t e0, e1; double flops, bytes; int n; };   // n: launches between the two events (burst)

// RCCL is bound at run time (dlopen on the first cora_comm_* call): the one-GPU path never loads it.
struct RcclApi {
  void* so = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
RcclApi g_rccl;
const char* rccl_load() {   // nullptr on success, else what failed
  if (g_rccl.so) return nullptr;
  void* so = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!so) so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!so) return "librccl.so not found (dlopen)";
  RcclApi a; a.so = so;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(so, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(so, "ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(so, "ncclCommDestroy");
  a.AllReduce = (decltype(a.AllReduce))dlsym(so, "ncclAllReduce");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(so, "ncclGetErrorString");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString) return "librccl.so lacks a required symbol";
  g_rccl = a;
  return nullptr;
}

}  // namespace

// ---- allocation-free entry point (cora_forward_static): its scratch lives in the code object ---------
// 16 MiB per GPU, held from module load until the process ends.  In HOST code the array's name is only
// a stand-in for the symbol, not a device address: the per-device address comes from
// hipGetSymbolAddress (static_scratch_ptr) and only that pointer is ever handed to kernels.
#define CORA_STATIC_SCRATCH_FLOATS (4u * 1024u * 1024u)
static __device__ __align__(256) float scratch[CORA_STATIC_SCRATCH_FLOATS];
static float* g_scratch_ptr[64] = {};      // per device, filled once with that device current (under g_scratch_mu)
static std::mutex g_scratch_mu;
// Device address of this GPU's copy of `scratch` (each GPU holds its own copy at its own address).
// Must run with `device` current and outside stream capture: cora_create does it for its device, and
// cora_forward_static does it on its first call on a device.
static int static_scratch_ptr(int device, float** out, std::string* err) {
  if (device < 0 || device >= 64) { *err = "device index out of range"; return -1; }
  std::lock_guard<std::mutex> lock(g_scratch_mu);   // two host threads may make the first call on a device together
  if (!g_scratch_ptr[device]) {
    float* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
      *err = std::string("hipGetSymbolAddress(scratch) failed: ") + hipGetErrorString(e);
      return -2;
    }
    g_scratch_ptr[device] = ptr;
  }
  *out = g_scratch_ptr[device];
  return 0;
}


struct cora_ctx {
  cora_config cfg;
  int device = 0;
  hipStream_t st = nullptr;
  std::string err;

  // dims
  int N, P, D, H, M, L, B;
  long long BN;       // B*N rows per layer
  long long n0, nl;   // floats in vode 0 / vodes 1..L+1
  long long nV;       // n0 + (L+1)*nl
  int Tmax;

  // weight arena + optimizer
  size_t nW = 0;
  float *W = nullptr, *dW = nullptr, *amu = nullptr, *anu = nullptr;
  bool dW_owned = true;
  int64_t adam_count = 0;
  size_t oLn1g, oLn1b, oLn2g, oLn2b, oWqkv, oWo, oW1, ob1, oW2, ob2, oWp, obp, oWout, obout;
  // use_vode_state_layernorm: gamma / beta of the L+1 vode LayerNorms ([L+1, D] each), the normalised
  // vodes x' the blocks read, their row statistics, and the cotangents dE/dx' of vodes 1..L+1
  bool vln = false;
  size_t oVg = 0, oVb = 0;
  float *hln = nullptr, *stv = nullptr, *dxr = nullptr;
  float* pos = nullptr;
  float* WT = nullptr;      // transposed copies for the fused kernel's backward GEMMs
  size_t tWqkv = 0, tWo = 0, tW1 = 0, tW2 = 0, tWp = 0, tWout = 0, nWT = 0;
  // bf16 mode: bf16 copies of the weights (hi, lo: forward GEMMs use the split form) and of their transposes
  // (hi only: backward-data GEMMs use plain bf16), refreshed with WT after every weight update
  unsigned short *Wh = nullptr, *Wl = nullptr, *WTh = nullptr;          // bf16 copies of W (hi, second term) and WT (hi)
  unsigned short *W3 = nullptr, *WTl = nullptr, *WT3 = nullptr;          // dtype_mode 2: third term of W; second, third of WT
  bool fused = false;     // D = 64 fused kernel (inference, STORE and chain modes)
  bool fusedw = false;    // D = 96 / 128 fused kernel (inference step only; learning phase and chain stay per-op)
  bool fusedw2 = false;   // ... its two-workgroups-per-CU form (D = 128, heads of 32)
  bool frows_bwd1 = false;   // ... with the single-pass attention backward (N a multiple of 64, > 64)
  bool frows = false;     // D = 64 at any sequence length: row-tile fused kernels around attn.hip (fused_rows.hip)
  // bf16 mode, per-op inference step: the two widest backward-data scratch tensors live as bf16 between their producer
  // and the one GEMM that reads them (da: gelu' epilogue -> W1^T GEMM; dqkv: attention backward -> Wqkv^T GEMM).  The
  // consumer rounds its A operand to bf16 anyway, so results are bit-identical; the learning pass keeps fp32.
  bool da16 = false, dqkv16 = false;
  bool slotted() const { return fused || fusedw; }   // inference-step energies kept as 8 partial sums per vode (esum)

  // vodes
  float *Hs = nullptr, *Ms = nullptr, *Gs = nullptr, *Es = nullptr;  // Es: e_1..e_{L+1} then e_img
  // D = 64 fused path, deferred vode update (FusedP::lazy): inside a T-step loop the update of step t is applied by the
  // inference launch of step t + 1 as it loads its rows, out of place -- so vodes, momentum and gradients ping-pong
  // between two sets of arenas.  `cur` is the set the vodes live in (always 0 between API calls: the last step of a loop
  // updates explicitly, into set 0), `pend` the step whose update is waiting, `lazy_T` the length of the running loop
  // (0: every update is launched as its own kernel, as on every other path).
  float *Hs2 = nullptr, *Ms2 = nullptr, *Gs2 = nullptr;
  bool lazy_ok = false;
  int lazy_T = 0, pend = -1, cur = 0;
  float* Hc() const { return cur ? Hs2 : Hs; }
  float* Mc() const { return cur ? Ms2 : Ms; }
  float* Gc() const { return cur ? Gs2 : Gs; }
  float* xp = nullptr;     // target, patch layout [BN, P]
  float* ximg = nullptr;   // image staging [B, F*C*H*W]
  long long nImg;

  // block scratch for Gmax groups
  int Gmax;
  float *xn1, *st1, *qkv, *att, *lse, *y, *xn2, *st2, *a;
  float *da, *dzn, *dy, *dO, *delta, *dqkv, *dxn;
  float *ua, *ub, *z, *stash = nullptr;
  float* dqparts = nullptr;  // row-tile path, single-pass attention backward: ceil(N / 128) partial copies of dq, [L B N, D] each
  unsigned long long* dbg = nullptr;
  float* gscale = nullptr;   // [L] layer scale of the vode each block reads
  float* eta_dev = nullptr;  // [Tmax]
  // inpainting loop (cora_unmask): ground truth and mask in patch layout, momentum of the un-frozen
  // sensory vode (allocated on first use); sens_live marks the sensory vode as un-frozen
  float *xt = nullptr, *mk = nullptr, *msens = nullptr;
  float* colpart = nullptr;    // fused STORE pass: per-wave column sums (bias / LayerNorm gradients)
  bool sens_live = false;
  float sens_scale = 1.f;
  double* energies = nullptr;  // [(Tmax+2), L+3]
  double* sumsq = nullptr;     // [(Tmax+2), L+2, 8 slots]
  double* esum = nullptr;      // [(Tmax+2), L+3, 8 slots]  energy slots of the fused inference step
  double* scal = nullptr;      // [0]=w-grad sumsq, [1]=mse sum
  float scale_l[64];

  std::map<int, hipGraphExec_t> graphs;
  bool learn_warm = false;
  bool infer_pending = false;   // cora_infer(out = NULL) launched, not yet synchronised (cora_wgrad ends it)
  hipGraphExec_t learn_graph = nullptr;
  // data-parallel path: the two halves around the all-reduce, captured separately
  hipGraphExec_t wgrad_graph = nullptr, apply_graph = nullptr;
  // data-parallel path with RCCL bound inside the library (cora_comm_init / cora_step_dist): the
  // collectives sit INSIDE the captured graphs
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  bool dist_warm = false;
  bool dist_graph_broken = false;   // a collective could not be captured (or cora_dist_set_eager): cora_step_dist stays on eager launches
  int dist_last_mode = 0;           // how the last cora_step_dist ran: 1 eager, 2 graph replay (cora_dist_mode)
  std::map<int, hipGraphExec_t> dist_graphs;   // key 2 T + exact_clip: T-step loop (+ per-step sum-of-squares all-reduce)
  hipGraphExec_t dist_learn_graph = nullptr;   // wgrad, all-reduce of dE/dW, clip + AdamW, reconstruction
  float apply_graph_scale = 0.f;   // cora_step: weight gradients + AdamW + reconstruction, captured once
  float* hyp_dev = nullptr;               // {eta_w, 1 - b1^count, 1 - b2^count} for the captured AdamW
  double* host_out = nullptr;             // pinned: scalars read back at the end of cora_step
  float* host_in = nullptr;               // pinned staging of eta_h[Tmax] + AdamW hyper-parameters (no sync to upload)
  hipEvent_t ev2 = nullptr;
  bool use_graph = true;
  bool warmed = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int last_T = 0;

  // profiling
  bool prof = false;
  int prof_burst = 1;   // cora_profile_step: the block kernel is timed as a burst of identical back-to-back launches
  std::vector<ProfRec> recs;

  std::vector<void*> allocs;
};

namespace {

#define HCHECK(call)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      char buf_[512];                                                                     \
      snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
               __FILE__, __LINE__);                                                       \
      c->err = buf_;                                                                      \
      return -2;                                                                          \
    }                                                                                     \
  } while (0)

#define FAIL(code, msg) \
  do { c->err = (msg); return (code); } while (0)

struct KScope {
  cora_ctx* c; bool on;
  KScope(cora_ctx* c_, int kind, double flops, double bytes, int n = 1) : c(c_), on(c_->prof) {
    if (!on) return;
    ProfRec r; r.kind = kind; r.flops = flops; r.bytes = bytes; r.n = n;
    hipEventCreate(&r.e0); hipEventCreate(&r.e1);
    hipEventRecord(r.e0, c->st);
    c->recs.push_back(r);
  }
  ~KScope() { if (on) hipEventRecord(c->recs.back().e1, c->st); }
};

int dalloc(cora_ctx* c, float** p, long long n) {
  void* v = nullptr;
  if (n < 1) n = 1;
  HCHECK(hipMalloc(&v, (size_t)n * sizeof(float)));
  HCHECK(hipMemsetAsync(v, 0, (size_t)n * sizeof(float), c->st));
  c->allocs.push_back(v);
  *p = (float*)v;
  return 0;
}

// A cora_infer(out = NULL) may still be in flight and still reads the pinned staging buffers and the
// vodes: every entry point that touches vodes, eta_dev, xp or the energy log ends it first.
int flush_lazy_update(cora_ctx* c);
void fill_update(cora_ctx* c, UpdateP& u, int t);
int settle_pending(cora_ctx* c) {
  if (c->pend >= 0) {   // a split-API loop abandoned between cora_infer_update and the next cora_infer_grad
    HCHECK(hipSetDevice(c->device));
    if (int rc_ = flush_lazy_update(c)) return rc_;
  }
  c->lazy_T = 0;
  if (!c->infer_pending) return 0;
  HCHECK(hipSetDevice(c->device));
  HCHECK(hipStreamSynchronize(c->st));
  c->infer_pending = false;
  return 0;
}

float* vode_ptr(cora_ctx* c, float* base, int l) { return l == 0 ? base : base + c->n0 + (long long)(l - 1) * c->nl; }
float* err_ptr(cora_ctx* c, int l) { return c->Es + (long long)(l - 1) * c->nl; }  // l = 1..L+1 ; L+2 = e_img

void drop_learn_graphs(cora_ctx* c) {
  if (c->learn_graph || c->wgrad_graph || c->apply_graph || c->dist_learn_graph) hipStreamSynchronize(c->st);
  if (c->dist_learn_graph) { hipGraphExecDestroy(c->dist_learn_graph); c->dist_learn_graph = nullptr; }
  if (c->learn_graph) { hipGraphExecDestroy(c->learn_graph); c->learn_graph = nullptr; }
  if (c->wgrad_graph) { hipGraphExecDestroy(c->wgrad_graph); c->wgrad_graph = nullptr; }
  if (c->apply_graph) { hipGraphExecDestroy(c->apply_graph); c->apply_graph = nullptr; }
}

// layer scales g_l *= base^{(L+1) - l} (src/decoder_transformer.py:321-341; the un-frozen sensory vode has
// index L+2) from the current cfg.use_scaling / scale_base; the captured inference graphs bake them
int apply_grad_options(cora_ctx* c) {
  float gs[64];
  for (int l = 0; l <= c->L + 1; ++l)
    c->scale_l[l] = c->cfg.use_scaling ? (float)pow((double)c->cfg.scale_base, (double)(c->L + 1 - l)) : 1.f;
  c->sens_scale = c->cfg.use_scaling ? (float)pow((double)c->cfg.scale_base, -1.0) : 1.f;
  for (int g = 0; g < 64; ++g) gs[g] = (g <= c->L) ? c->scale_l[g + 1] : 1.f;
  if (hipMemcpyAsync(c->gscale, gs, sizeof gs, hipMemcpyHostToDevice, c->st) != hipSuccess ||
      hipStreamSynchronize(c->st) != hipSuccess) return -2;
  for (auto& kv : c->graphs) hipGraphExecDestroy(kv.second);
  c->graphs.clear();
  for (auto& kv : c->dist_graphs) hipGraphExecDestroy(kv.second);
  c->dist_graphs.clear();
  return 0;
}

GemmP gemm_base(cora_ctx* c, int kind = 0 /* 0 forward, 1 backward-data, 2 weight-gradient */) {
  GemmP p;
  memset(&p, 0, sizeof p);
  p.groups = 1; p.splitk = 1; p.alpha = 1.f; p.alpha2 = 1.f; p.mode = OUT_STORE;
  p.l1 = 0.f; p.l2 = 0.f; p.red_stride = 1;
  // bf16 mode: backward-data and weight-gradient GEMMs take plain bf16 operands (their error is
  // relative to the result); the forward predictions feed e = h - u, which cancels most of u, so
  // they use the split form (measured: plain bf16 there alone costs 2.5e-2 on the 6-block vodes)
  // dtype_mode 2 ("bf16x3"): three-term split everywhere, fp32-grade products
  p.bf16 = (c->cfg.dtype_mode == 1) ? (kind == 0 ? 2 : 1) : (c->cfg.dtype_mode == 2 ? 3 : 0);
  return p;
}

void run_gemm(cora_ctx* c, const GemmP& p_in, bool ako, bool bko) {
  GemmP p = p_in;
  static const bool no_pre = getenv("CORA_BF16_NO_PRE") != nullptr;   // diagnostic: convert weights while staging
  if (c->Wh && p.bf16 && !ako && !bko && !no_pre) {
    // bf16 mode, weight operand: hand the kernel the copy that is already bf16 (same element offsets)
    if (p.Bw >= c->W && p.Bw < c->W + c->nW) {
      const size_t o = p.Bw - c->W;
      p.Bh = c->Wh + o; p.Bl = c->Wl + o; p.B3 = c->W3 ? c->W3 + o : nullptr;
    } else if (c->WT && p.Bw >= c->WT && p.Bw < c->WT + c->nWT && (p.bf16 == 1 || (p.bf16 == 3 && c->WT3))) {
      const size_t o = p.Bw - c->WT;
      p.Bh = c->WTh + o; p.Bl = c->WTl ? c->WTl + o : nullptr; p.B3 = c->WT3 ? c->WT3 + o : nullptr;
    }
  }
  int kind = ako ? K_GEMM_TN : (bko ? K_GEMM_NN : K_GEMM_NT);
  double fl = 2.0 * p.R * (double)p.O * p.K * p.groups;
  double by = p.groups * ((p.a16 ? 2.0 : 4.0) * p.R * p.K + 4.0 * p.O * p.K + (p.c16 ? 2.0 : 4.0) * p.R * p.O);
  KScope s(c, kind, fl, by);
  launch_gemm(p, ako, bko, c->st);
}

// Weight operand of a backward-data GEMM (contraction over the weight's OUTPUT index): K-outer as stored.
// In bf16 mode the GEMMs read the transposed copy instead -- K-inner, the one operand form the
// bf16-staged kernel (gemm_bf16.hip) takes.  Sets p.Bw / p.ldb (p.K and p.O already set); returns b_kouter.
bool bwd_weight(cora_ctx* c, GemmP& p, size_t w_off, size_t wt_off) {
  if (c->cfg.dtype_mode != 0 && c->WT) { p.Bw = c->WT + wt_off; p.ldb = p.K; return false; }
  p.Bw = c->W + w_off; p.ldb = p.O;
  return true;
}

// patch projection: u_1 = h_0 Wp^T + bp + pos   (src/decoder_transformer.py:222-228)
void patch_proj(cora_ctx* c, const float* h0, float* out, int mode, const float* tgt, double* red) {
  GemmP p = gemm_base(c);
  p.A = h0; p.Bw = c->W + c->oWp; p.C = out;
  p.R = (int)c->BN; p.O = c->D; p.K = c->P; p.lda = c->P; p.ldb = c->P; p.ldc = c->D;
  p.bias = c->W + c->obp; p.pos = c->pos; p.posN = c->N;
  p.mode = mode; p.tgt = tgt; p.red = red; p.l1 = c->cfg.l1; p.l2 = c->cfg.l2;
  run_gemm(c, p, false, false);
}

// output projection: z = h_{L+1} Wout^T + bout (patch layout; unpatchify is a permutation, :244-250)
void out_proj(cora_ctx* c, const float* hin, float* out, int mode, const float* tgt, double* red) {
  GemmP p = gemm_base(c);
  p.A = hin; p.Bw = c->W + c->oWout; p.C = out;
  p.R = (int)c->BN; p.O = c->P; p.K = c->D; p.lda = c->D; p.ldb = c->D; p.ldc = c->P;
  p.bias = c->W + c->obout;
  p.mode = mode; p.tgt = tgt; p.red = red;  // sensory vode: plain se_energy, no regulariser
  run_gemm(c, p, false, false);
}

// Forward of G transformer blocks (blocks b0..b0+G-1), block g reading xin + g*BN*D:
//   u = 2x + MHA(LN1 x) + MLP(LN2(x + MHA(LN1 x)))   (pcx/nn/_layer.py:428-447 + decoder :238)
// mode OUT_ERR : C = tgt - u written to `out` (errors), energies accumulated at red[g]
// mode OUT_STORE: C = u
int block_forward(cora_ctx* c, int G, int b0, const float* xin, float* out, int mode, const float* tgt, double* red) {
  const int D = c->D, M = c->M, H = c->H;
  const int R = (int)c->BN;
  const long long rows = (long long)G * R;
  {
    KScope s(c, K_LN_FWD, 0, 8.0 * rows * D);
    if (launch_ln_fwd(xin, c->W + c->oLn1g + (size_t)b0 * D, c->W + c->oLn1b + (size_t)b0 * D, D, c->xn1, c->st1, R, G, D, c->st))
      FAIL(-3, "layer norm: hidden size not supported");
  }
  GemmP p = gemm_base(c);
  p.groups = G;
  // QKV
  p.A = c->xn1; p.Bw = c->W + c->oWqkv + (size_t)b0 * 3 * D * D; p.C = c->qkv;
  p.R = R; p.O = 3 * D; p.K = D; p.lda = D; p.ldb = D; p.ldc = 3 * D;
  p.sA = (long long)R * D; p.sB = 3LL * D * D; p.sC = (long long)R * 3 * D;
  run_gemm(c, p, false, false);
  {
    double fl = 4.0 * G * c->B * (double)c->N * c->N * D;
    KScope s(c, K_ATTN_FWD, fl, 4.0 * rows * 4 * D);
    if (c->cfg.dtype_mode == 1 && attn_fwd_bf16_supported(c->N, H, D)) {
      if (launch_attn_fwd_bf16(c->qkv, c->att, c->lse, G * c->B, c->N, H, D, c->st)) FAIL(-3, "attention (bf16): launch set-up failed");
    } else if (launch_attn_fwd(c->qkv, c->att, c->lse, G * c->B, c->N, H, D, c->st)) FAIL(-3, "attention: head dim not supported (need 16,24,32,48,64,96,128)");
  }
  // y = x + att Wo^T
  p = gemm_base(c); p.groups = G;
  p.A = c->att; p.Bw = c->W + c->oWo + (size_t)b0 * D * D; p.C = c->y;
  p.R = R; p.O = D; p.K = D; p.lda = D; p.ldb = D; p.ldc = D;
  p.sA = (long long)R * D; p.sB = (long long)D * D; p.sC = (long long)R * D;
  p.res1 = xin;
  run_gemm(c, p, false, false);
  {
    KScope s(c, K_LN_FWD, 0, 8.0 * rows * D);
    launch_ln_fwd(c->y, c->W + c->oLn2g + (size_t)b0 * D, c->W + c->oLn2b + (size_t)b0 * D, D, c->xn2, c->st2, R, G, D, c->st);
  }
  // a = LN2(y) W1^T + b1
  p = gemm_base(c); p.groups = G;
  p.A = c->xn2; p.Bw = c->W + c->oW1 + (size_t)b0 * M * D; p.C = c->a;
  p.R = R; p.O = M; p.K = D; p.lda = D; p.ldb = D; p.ldc = M;
  p.sA = (long long)R * D; p.sB = (long long)M * D; p.sC = (long long)R * M;
  p.bias = c->W + c->ob1 + (size_t)b0 * M; p.sBias = M;
  run_gemm(c, p, false, false);
  // u = x + y + gelu(a) W2^T + b2
  p = gemm_base(c); p.groups = G;
  p.A = c->a; p.a_gelu = 1; p.Bw = c->W + c->oW2 + (size_t)b0 * D * M; p.C = out;
  p.R = R; p.O = D; p.K = M; p.lda = M; p.ldb = M; p.ldc = D;
  p.sA = (long long)R * M; p.sB = (long long)D * M; p.sC = (long long)R * D;
  p.bias = c->W + c->ob2 + (size_t)b0 * D; p.sBias = D;
  p.res1 = xin; p.res2 = c->y;
  p.mode = mode; p.tgt = tgt; p.red = red; p.red_stride = 1; p.l1 = c->cfg.l1; p.l2 = c->cfg.l2;
  run_gemm(c, p, false, false);
  return 0;
}

// Hand-derived VJP of G blocks with cotangent r (the next vode's error):
//   dy = r + LN2^T(W1^T(gelu'(a) * (W2^T r)));  dx = r + dy + LN1^T(MHA^T(dy))
// and, when gout != null, the vode gradient g_l = scale_l (e_l + reg - dx) with its sum of squares.
// raw: gout receives the cotangent dx itself (no gradient assembly) -- the vode LayerNorm stage finishes it.
int block_backward(cora_ctx* c, int G, int b0, const float* xin, const float* r, const float* own, float* gout, double* red,
                   bool raw = false, bool half = false /* inference step: da / dqkv as bf16 where the ctx allows (da16, dqkv16) */) {
  const bool da16 = half && c->da16, dq16 = half && c->dqkv16;
  const int D = c->D, M = c->M, H = c->H;
  const int R = (int)c->BN;
  const long long rows = (long long)G * R;
  GemmP p = gemm_base(c, 1); p.groups = G;
  // da = (r W2) * gelu'(a)
  p.A = r; p.C = c->da;
  p.R = R; p.O = M; p.K = D; p.lda = D; p.ldc = M;
  p.sA = (long long)R * D; p.sB = (long long)D * M; p.sC = (long long)R * M;
  p.mode = OUT_GELUGRAD; p.aux = c->a; p.c16 = da16;
  run_gemm(c, p, false, bwd_weight(c, p, c->oW2 + (size_t)b0 * D * M, c->tW2 + (size_t)b0 * D * M));
  // dzn = da W1
  p = gemm_base(c, 1); p.groups = G;
  p.A = c->da; p.C = c->dzn; p.a16 = da16;
  p.R = R; p.O = D; p.K = M; p.lda = M; p.ldc = D;
  p.sA = (long long)R * M; p.sB = (long long)M * D; p.sC = (long long)R * D;
  run_gemm(c, p, false, bwd_weight(c, p, c->oW1 + (size_t)b0 * M * D, c->tW1 + (size_t)b0 * M * D));
  {
    LnBwdP q; memset(&q, 0, sizeof q);
    q.dxn = c->dzn; q.x = c->y; q.stats = c->st2; q.gamma = c->W + c->oLn2g + (size_t)b0 * D; q.sParam = D;
    q.r = r; q.out = c->dy; q.rows_per_group = R; q.groups = G; q.D = D; q.mode = 0;
    KScope s(c, K_LN_BWD, 0, 16.0 * rows * D);
    if (launch_ln_bwd(q, c->st)) FAIL(-3, "layer norm backward: hidden size not supported (D > 1024)");
  }
  // dO = dy Wo
  p = gemm_base(c, 1); p.groups = G;
  p.A = c->dy; p.C = c->dO;
  p.R = R; p.O = D; p.K = D; p.lda = D; p.ldc = D;
  p.sA = (long long)R * D; p.sB = (long long)D * D; p.sC = (long long)R * D;
  run_gemm(c, p, false, bwd_weight(c, p, c->oWo + (size_t)b0 * D * D, c->tWo + (size_t)b0 * D * D));
  {
    double fl = 14.0 * G * c->B * (double)c->N * c->N * D;   // 7 products (S and dP are recomputed per side)
    KScope s(c, K_ATTN_BWD, fl, 4.0 * rows * 9 * D);
    if (c->cfg.dtype_mode == 1 && attn_bwd_bf16_supported(c->N, H, D)) {
      if (launch_attn_bwd_bf16(c->qkv, c->att, c->lse, c->dO, c->delta, c->dqkv, G * c->B, c->N, H, D, c->st, dq16)) FAIL(-3, "attention backward (bf16): launch set-up failed");
    } else {
      if (dq16) FAIL(-3, "attention backward: bf16 scratch without the bf16 kernels");
      if (launch_attn_bwd(c->qkv, c->att, c->lse, c->dO, c->delta, c->dqkv, G * c->B, c->N, H, D, c->st)) FAIL(-3, "attention backward: head dim not supported");
    }
  }
  // dxn = dqkv Wqkv
  p = gemm_base(c, 1); p.groups = G;
  p.A = c->dqkv; p.C = c->dxn; p.a16 = dq16;
  p.R = R; p.O = D; p.K = 3 * D; p.lda = 3 * D; p.ldc = D;
  p.sA = (long long)R * 3 * D; p.sB = 3LL * D * D; p.sC = (long long)R * D;
  run_gemm(c, p, false, bwd_weight(c, p, c->oWqkv + (size_t)b0 * 3 * D * D, c->tWqkv + (size_t)b0 * 3 * D * D));
  if (gout) {
    LnBwdP q; memset(&q, 0, sizeof q);
    q.dxn = c->dxn; q.x = xin; q.stats = c->st1; q.gamma = c->W + c->oLn1g + (size_t)b0 * D; q.sParam = D;
    q.r = r; q.dy = c->dy; q.own = own; q.out = gout; q.rows_per_group = R; q.groups = G; q.D = D; q.mode = raw ? 2 : 1;
    q.gscale = c->gscale + b0; q.l1 = c->cfg.l1; q.l2 = c->cfg.l2; q.red = red; q.red_stride = 1;
    KScope s(c, K_LN_BWD, 0, 28.0 * rows * D);
    if (launch_ln_bwd(q, c->st)) FAIL(-3, "layer norm backward: shape not supported");
  }
  return 0;
}

// transposed weight copies (W^T) read by the fused kernel's backward GEMMs
void refresh_wt(cora_ctx* c) {
  if (!c->WT) return;
  const int D = c->D, M = c->M, L = c->L;
  KScope s(c, K_OTHER, 0, 8.0 * c->nWT);
  TransposeSet t; memset(&t, 0, sizeof t);
  auto add = [&](size_t so, size_t to, int rows, int cols, int groups) {
    t.src[t.n] = c->W + so; t.dst[t.n] = c->WT + to; t.rows[t.n] = rows; t.cols[t.n] = cols; t.groups[t.n] = groups; ++t.n;
  };
  add(c->oWqkv, c->tWqkv, 3 * D, D, L); add(c->oWo, c->tWo, D, D, L); add(c->oW1, c->tW1, M, D, L);
  add(c->oW2, c->tW2, D, M, L); add(c->oWp, c->tWp, D, c->P, 1); add(c->oWout, c->tWout, c->P, D, 1);
  launch_transpose_set(t, c->st);
  if (c->Wh) {
    launch_cvt_bf16(c->W, c->Wh, c->Wl, (long long)c->nW, c->st, c->W3);
    launch_cvt_bf16(c->WT, c->WTh, c->WTl, (long long)c->nWT, c->st, c->WT3);
  }
}

// flops of one block forward + VJP per image (2 FLOP per MAC): 48 N D^2 + 12 N^2 D
double block_step_flops(cora_ctx* c) {
  return (double)c->L * c->B * (48.0 * c->N * c->D * (double)c->D + 12.0 * c->N * (double)c->N * c->D);
}

// energy log row `slot` (L+3 doubles, index = vode) to the host.  Inference steps of the fused path
// keep 8 partial sums per vode (no same-address atomics in the hot kernels); they are folded here.
int energy_row(cora_ctx* c, int slot, double* en) {
  const int L = c->L;
  if (c->slotted() && slot < c->Tmax) {
    std::vector<double> s8((size_t)(L + 3) * 8);
    if (hipMemcpy(s8.data(), c->esum + (long long)slot * (L + 3) * 8, sizeof(double) * s8.size(), hipMemcpyDeviceToHost) != hipSuccess) return -2;
    for (int v = 0; v < L + 3; ++v) {
      double e = 0.0;
      for (int k = 0; k < 8; ++k) e += s8[(size_t)v * 8 + k];
      en[v] = e;
    }
    return 0;
  }
  return hipMemcpy(en, c->energies + (long long)slot * (L + 3), sizeof(double) * (L + 3), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}

// parameter block of the fused kernels (pc_block_fused / pc_chain)
void fill_fused(cora_ctx* c, FusedP& f, double* en /* energy row of this pass */, double* ss /* sumsq row */) {
  const int L = c->L;
  memset(&f, 0, sizeof f);
  f.H1 = vode_ptr(c, c->Hc(), 1); f.E2 = err_ptr(c, 2); f.BP1 = vode_ptr(c, c->Gc(), 1); f.W = c->W; f.WT = c->WT;
  f.oLn1g = c->oLn1g; f.oLn1b = c->oLn1b; f.oLn2g = c->oLn2g; f.oLn2b = c->oLn2b; f.oWqkv = c->oWqkv; f.oWo = c->oWo;
  f.oW1 = c->oW1; f.ob1 = c->ob1; f.oW2 = c->oW2; f.ob2 = c->ob2;
  f.tWqkv = c->tWqkv; f.tWo = c->tWo; f.tW1 = c->tW1; f.tW2 = c->tW2;
  f.energy = en ? en + 2 : nullptr; f.esens = nullptr; f.B = c->B; f.l1 = c->cfg.l1; f.l2 = c->cfg.l2;
  f.L = L; f.s0 = c->scale_l[0]; f.H0 = vode_ptr(c, c->Hc(), 0); f.XP = c->xp; f.pos = c->pos;
  f.E1 = err_ptr(c, 1); f.EIMG = err_ptr(c, L + 2); f.G0 = vode_ptr(c, c->Gc(), 0); f.sumsq0 = ss;
  f.oWp = c->oWp; f.obp = c->obp; f.oWout = c->oWout; f.obout = c->obout; f.tWp = c->tWp; f.tWout = c->tWout;
  f.S_xn1 = c->xn1; f.S_att = c->att; f.S_xn2 = c->xn2; f.S_a = c->a; f.S_da = c->da;
  f.S_dy = c->dy; f.S_dqkv = c->dqkv;
  f.S_stash = c->stash; f.H = c->H;
  f.Z = c->z; f.mse = nullptr;
  f.dbg = c->dbg;
}

// parameter block of the row-tile fused kernels (fused_rows.hip)
void fill_rows(cora_ctx* c, RowsP& r, double* en2 /* energy of vode 2 */, double* ss1 /* sumsq slots of vode 1 */) {
  memset(&r, 0, sizeof r);
  r.X = vode_ptr(c, c->Hs, 1); r.E2 = err_ptr(c, 2); r.E1 = err_ptr(c, 1); r.G1 = vode_ptr(c, c->Gs, 1);
  r.W = c->W; r.WT = c->WT;
  r.oLn1g = c->oLn1g; r.oLn1b = c->oLn1b; r.oLn2g = c->oLn2g; r.oLn2b = c->oLn2b; r.oWqkv = c->oWqkv; r.oWo = c->oWo;
  r.oW1 = c->oW1; r.ob1 = c->ob1; r.oW2 = c->oW2; r.ob2 = c->ob2;
  r.tWqkv = c->tWqkv; r.tWo = c->tWo; r.tW1 = c->tW1; r.tW2 = c->tW2;
  r.qkv = c->qkv; r.att = c->att; r.dO = c->dO; r.rdy = c->dzn; r.dqkv = c->dqkv;
  r.rows = c->BN; r.L = c->L; r.energy = en2; r.sumsq = ss1; r.gscale = c->gscale; r.l1 = c->cfg.l1; r.l2 = c->cfg.l2;
  r.S_xn1 = c->xn1; r.S_xn2 = c->xn2; r.S_a = c->a; r.S_da = c->da; r.S_dy = c->dy; r.colpart = c->colpart;
  r.H = c->H;
  if (c->frows_bwd1) {   // single-pass attention backward (attn.hip): dq as per-key-block partial copies
    r.delta = c->delta; r.dqa = c->dqparts; r.ndq = attn_bwd_fused_parts(c->N); r.dq_stride = (long long)c->L * c->nl;
  }
  // edge layers as extra workgroups of the inference launches (the caller sets r.edge)
  r.N = c->N; r.s0 = c->scale_l[0]; r.H0 = vode_ptr(c, c->Hs, 0); r.XP = c->xp; r.pos = c->pos;
  r.EIMG = err_ptr(c, c->L + 2); r.G0 = vode_ptr(c, c->Gs, 0);
  r.oWp = c->oWp; r.obp = c->obp; r.oWout = c->oWout; r.obout = c->obout; r.tWp = c->tWp; r.tWout = c->tWout;
}
// the inference step of the row-tile path folds the two edge layers into its launches (fused_rows.hip: P = 48 only)
bool rows_edge_fused(const cora_ctx* c) {
  static const bool off = getenv("CORA_ROWS_NO_EDGE") != nullptr;
  return c->frows && c->P == 48 && !off;
}
// Row-tile path, forward half of all L blocks: q|k|v, attention, then e_{2..L+1}, their energies, dO and r + dy.
// store: also leave the operands of the weight-gradient contractions (learning phase).
int block_forward_rows(cora_ctx* c, double* en2, int store, double* ss1 = nullptr) {
  const int L = c->L, D = c->D;
  const double rows = (double)L * c->BN;
  RowsP r; fill_rows(c, r, en2, ss1);
  r.edge = (ss1 != nullptr && !store && rows_edge_fused(c)) ? 1 : 0;   // top edge: e_1, energy of vode 1, g_0
  { KScope s(c, K_ROWS, 2.0 * rows * D * 3 * D, 4.0 * rows * 4 * D);
    if (launch_rows_qkv(r, store, c->st)) FAIL(-3, "rows_qkv: launch set-up failed"); }
  { KScope s(c, K_ATTN_FWD, 4.0 * L * c->B * (double)c->N * c->N * D, 4.0 * rows * 4 * D);
    if (c->cfg.dtype_mode == 1 && attn_fwd_bf16_supported(c->N, c->H, D)) {   // bf16 mode: bf16-MFMA attention between the fp32 row kernels
      if (launch_attn_fwd_bf16(c->qkv, c->att, c->lse, L * c->B, c->N, c->H, D, c->st)) FAIL(-3, "attention (bf16): launch set-up failed");
    } else if (launch_attn_fwd(c->qkv, c->att, c->lse, L * c->B, c->N, c->H, D, c->st)) FAIL(-3, "attention: head dim not supported"); }
  { KScope s(c, K_ROWS, rows * (4.0 * D * D + 8.0 * D * c->M), 4.0 * rows * 6 * D);
    if (launch_rows_mid(r, store, c->st)) FAIL(-3, "rows_mid: launch set-up failed"); }
  return 0;
}
// ... and the backward half: attention backward, then the vode gradients g_1..g_L with their sums of squares
// (store: the LayerNorm-1 column sums instead)
int block_backward_rows(cora_ctx* c, double* ss1, int store, double* esens = nullptr) {
  const int L = c->L, D = c->D;
  const double rows = (double)L * c->BN;
  RowsP r; fill_rows(c, r, nullptr, ss1);
  r.edge = (esens != nullptr && !store && rows_edge_fused(c)) ? 1 : 0;  // bottom edge: e_img, sensory energy, g_{L+1}
  r.esens = esens;
  if (c->frows_bwd1) {
    KScope s(c, K_ATTN_BWD, 10.0 * L * c->B * (double)c->N * c->N * D, 4.0 * rows * (7 + attn_bwd_fused_parts(c->N)) * D);   // 5 products
    if (launch_attn_bwd_fused(c->qkv, c->lse, c->dO, c->delta, c->dqparts, (long long)L * c->nl, c->dqkv, L * c->B, c->N, c->H, D, c->st)) FAIL(-3, "attention backward (single pass): launch set-up failed");
  } else {
    KScope s(c, K_ATTN_BWD, 14.0 * L * c->B * (double)c->N * c->N * D, 4.0 * rows * 9 * D);
    if (c->cfg.dtype_mode == 1 && attn_bwd_bf16_supported(c->N, c->H, D)) {
      if (launch_attn_bwd_bf16(c->qkv, c->att, c->lse, c->dO, c->delta, c->dqkv, L * c->B, c->N, c->H, D, c->st)) FAIL(-3, "attention backward (bf16): launch set-up failed");
    } else if (launch_attn_bwd(c->qkv, c->att, c->lse, c->dO, c->delta, c->dqkv, L * c->B, c->N, c->H, D, c->st)) FAIL(-3, "attention backward: head dim not supported");
  }
  { KScope s(c, K_ROWS, 2.0 * rows * D * 3 * D, 4.0 * rows * 7 * D);
    if (launch_rows_out(r, store, c->st)) FAIL(-3, "rows_out: launch set-up failed"); }
  return 0;
}

// energy + errors at the current vodes into slot `slot` of the energy log (status None forward)
// vodes 1..L+1 through their LayerNorms (use_vode_state_layernorm, :232-241): hln = LN_v(h_v)
void vode_ln_forward(cora_ctx* c) {
  const int L = c->L, D = c->D;
  KScope s(c, K_LN_FWD, 0, 8.0 * (L + 1) * c->nl);
  launch_ln_fwd(vode_ptr(c, c->Hs, 1), c->W + c->oVg, c->W + c->oVb, D, c->hln, c->stv, (int)c->BN, L + 1, D, c->st);
}
// what block g / the output projection read: the vode itself, or its normalised copy
const float* vode_in(cora_ctx* c, int l) { return c->vln ? c->hln + (long long)(l - 1) * c->nl : vode_ptr(c, c->Hs, l); }

// ss != null (the inference step of the row-tile path with P = 48): the edge layers ride in the rows launches -- e_1, the
// energy of vode 1 and g_0 here; e_img, the sensory energy and g_{L+1} in block_backward_rows, which must follow
int errors_pass(cora_ctx* c, int slot, int store = 0, double* ss = nullptr) {
  const int L = c->L;
  double* en = c->energies + (long long)slot * (L + 3);
  const bool edge = ss != nullptr && !store && rows_edge_fused(c);
  if (c->vln) vode_ln_forward(c);
  if (!edge) patch_proj(c, vode_ptr(c, c->Hs, 0), err_ptr(c, 1), OUT_ERR, vode_ptr(c, c->Hs, 1), en + 1);
  if (L > 0) {
    int rc = c->frows ? block_forward_rows(c, en + 2, store, edge ? ss + 8 : nullptr)
                      : block_forward(c, L, 0, vode_in(c, 1), err_ptr(c, 2), OUT_ERR, vode_ptr(c, c->Hs, 2), en + 2);
    if (rc) return rc;
  }
  if (!edge) out_proj(c, vode_in(c, L + 1), err_ptr(c, L + 2), OUT_ERR, c->xp, en + L + 2);
  return 0;
}

// A deferred vode update is waiting (cora_ctx::pend): point the launch at the old arenas for its reads and at the other
// set for what it writes (FusedP::lazy), and flip.
bool lazy_args(cora_ctx* c, FusedP& f) {
  if (c->pend < 0) return false;
  float *Ho = c->Hc(), *Mo = c->Mc(), *Go = c->Gc();
  c->cur ^= 1;
  float *Hn = c->Hc(), *Mn = c->Mc(), *Gn = c->Gc();
  f.lazy = 1;
  f.H0 = Ho; f.H1 = Ho + c->n0; f.M0o = Mo; f.Mo1 = Mo + c->n0; f.G0o = Go; f.Go1 = Go + c->n0;
  f.H0n = Hn; f.Hn1 = Hn + c->n0; f.M0n = Mn; f.Mn1 = Mn + c->n0; f.G0 = Gn; f.BP1 = Gn + c->n0;
  fill_update(c, f.up, c->pend);
  c->pend = -1;
  return true;
}

// first half of inference step t: errors, energies, dE/dh for every un-frozen vode with layer
// scaling, and the per-vode sums of squares of this shard (row t of c->sumsq)
int infer_grad_step(cora_ctx* c, int t) {
  const int L = c->L;
  int rc = 0;
  double* ss = c->sumsq + (long long)t * (L + 2) * 8;
  if (c->fused) {
    // fused kernel: e, J^T e, g_0 and the sensory energy ; vode_grad: g (in place over J^T e), its sums
    // of squares and the other energies (8 slots each: no same-address atomics in the hot kernels)
    double* es = c->esum + (long long)t * (L + 3) * 8;
    FusedP f;
    fill_fused(c, f, nullptr, ss);
    f.esens = es + (long long)(L + 2) * 8;
    const bool lazy = lazy_args(c, f);   // the update of step `pend` rides in this launch
    {
      double fl = block_step_flops(c) + 8.0 * c->B * (double)c->N * c->P * c->D;
      // profiling: a burst of identical launches under one event pair (the launch is idempotent: it reads the vodes and
      // -- with a deferred update -- writes the OTHER arena set), so the figure is the back-to-back duration a graph
      // replay sees, without the event packets and the idle gap of a single bracketed eager launch
      const int nb = c->prof ? c->prof_burst : 1;
      KScope s(c, K_FUSED, fl, 4.0 * 4 * L * c->nl + (lazy ? 20.0 * c->nV : 0.0), nb);
      for (int i = 0; i < nb; ++i)
        if (launch_fused_block(f, L, 0, c->st)) FAIL(-3, "fused block kernel: launch setup failed");
    }
    VodeGradP v; memset(&v, 0, sizeof v);
    v.bp = vode_ptr(c, c->Gc(), 1); v.e = err_ptr(c, 1); v.h = vode_ptr(c, c->Hc(), 1); v.g = vode_ptr(c, c->Gc(), 1);
    v.gscale = c->gscale; v.nl = c->nl; v.nlay = L + 1; v.l1 = c->cfg.l1; v.l2 = c->cfg.l2;
    v.sumsq = ss + 8; v.esum = es + 8;
    KScope s2(c, K_GRADFIN, 0, 12.0 * (L + 1) * c->nl);
    launch_vode_grad(v, c->st);
  } else if (c->fusedw) {
    // wide fused kernel (D = 96 / 128): one launch for all L blocks (e_2..e_{L+1} and J^T e for vodes 1..L) and,
    // in B extra workgroups, the two edge layers (e_1, g_0, e_img, J^T e for vode L+1); then vode_grad as on
    // the D = 64 path
    double* es = c->esum + (long long)t * (L + 3) * 8;
    FusedP f;
    fill_fused(c, f, nullptr, ss);
    f.esens = es + (long long)(L + 2) * 8;
    const bool lazy = lazy_args(c, f);   // wide2 only (lazy_ok): the update of step `pend` rides in this launch
    {
      const int nb = c->prof ? c->prof_burst : 1;   // burst under one event pair, as on the D = 64 path
      KScope s(c, K_FUSED, block_step_flops(c) + 8.0 * c->B * (double)c->N * c->P * c->D, 4.0 * 4 * L * c->nl + (lazy ? 20.0 * c->nV : 0.0), nb);
      for (int i = 0; i < nb; ++i)
        if (c->fusedw2 ? launch_fused_wide2(f, L, c->D, c->st) : launch_fused_wide(f, L, c->D, c->st)) FAIL(-3, "wide fused block kernel: launch setup failed");
    }
    VodeGradP v; memset(&v, 0, sizeof v);
    v.bp = vode_ptr(c, c->Gc(), 1); v.e = err_ptr(c, 1); v.h = vode_ptr(c, c->Hc(), 1); v.g = vode_ptr(c, c->Gc(), 1);
    v.gscale = c->gscale; v.nl = c->nl; v.nlay = L + 1; v.l1 = c->cfg.l1; v.l2 = c->cfg.l2;
    v.sumsq = ss + 8; v.esum = es + 8;
    KScope s2(c, K_GRADFIN, 0, 12.0 * (L + 1) * c->nl);
    launch_vode_grad(v, c->st);
  } else {
    int rc0 = errors_pass(c, t, 0, ss);
    if (rc0) return rc0;
  }
  if (rows_edge_fused(c)) {
    // row-tile path, P = 48: attention backward, then g_1..g_L and -- in the same launch -- the bottom edge
    if ((rc = block_backward_rows(c, ss + 8, 0, c->energies + (long long)t * (L + 3) + (L + 2)))) return rc;
  } else if (!c->fused && !c->fusedw) {
  // vode 0: g_0 = -e_1 Wp   (no own energy: energy_fn=None, src/decoder_transformer.py:143-153)
  GemmP p = gemm_base(c, 1);
  p.A = err_ptr(c, 1); p.C = vode_ptr(c, c->Gs, 0);
  p.R = (int)c->BN; p.O = c->P; p.K = c->D; p.lda = c->D; p.ldc = c->P;
  p.mode = OUT_GRAD; p.alpha2 = c->scale_l[0]; p.red = ss + 0;
  run_gemm(c, p, false, bwd_weight(c, p, c->oWp, c->tWp));
  // vode L+1: g = e_{L+1} + reg - e_img Wout
  p = gemm_base(c, 1);
  p.A = err_ptr(c, L + 2); p.C = vode_ptr(c, c->Gs, L + 1);
  p.R = (int)c->BN; p.O = c->D; p.K = c->P; p.lda = c->P; p.ldc = c->D;
  const bool bko_out = bwd_weight(c, p, c->oWout, c->tWout);
  if (c->vln) {   // cotangent w.r.t. the NORMALISED vode L+1; LN_L^T and the assembly follow below
    p.C = c->dxr + (long long)L * c->nl; p.mode = OUT_STORE;
  } else {
    p.mode = OUT_GRAD; p.alpha2 = c->scale_l[L + 1]; p.tgt = err_ptr(c, L + 1); p.hreg = vode_ptr(c, c->Hs, L + 1);
    p.l1 = c->cfg.l1; p.l2 = c->cfg.l2; p.red = ss + 8 * (L + 1);
  }
  run_gemm(c, p, false, bko_out);
  if (L > 0) {
    if (c->frows) rc = block_backward_rows(c, ss + 8, 0);
    else if (c->vln) rc = block_backward(c, L, 0, vode_in(c, 1), err_ptr(c, 2), nullptr, c->dxr, nullptr, true, true);
    else rc = block_backward(c, L, 0, vode_ptr(c, c->Hs, 1), err_ptr(c, 2), err_ptr(c, 1), vode_ptr(c, c->Gs, 1), ss + 8, false, true);
    if (rc) return rc;
  }
  if (c->vln) {   // g_v = s_v (e_v + reg - LN_{v-1}^T dx'_v) for v = 1..L+1 in one grouped pass
    LnBwdP q; memset(&q, 0, sizeof q);
    q.dxn = c->dxr; q.x = vode_ptr(c, c->Hs, 1); q.stats = c->stv; q.gamma = c->W + c->oVg; q.sParam = c->D;
    q.own = err_ptr(c, 1); q.out = vode_ptr(c, c->Gs, 1); q.rows_per_group = (int)c->BN; q.groups = L + 1; q.D = c->D; q.mode = 1;
    q.gscale = c->gscale; q.l1 = c->cfg.l1; q.l2 = c->cfg.l2; q.red = ss + 8; q.red_stride = 1;
    KScope s(c, K_LN_BWD, 0, 20.0 * (L + 1) * c->nl);
    if (launch_ln_bwd(q, c->st)) FAIL(-3, "vode layer norm backward: shape not supported");
  }
  }
  return 0;
}

// second half: clip by the global norm (row t of c->sumsq -- summed over ranks first in the exact
// data-parallel mode), momentum SGD on every vode
void fill_update(cora_ctx* c, UpdateP& u, int t) {
  const int L = c->L;
  memset(&u, 0, sizeof u);
  u.h = c->Hc(); u.m = c->Mc(); u.g = c->Gc(); u.n0 = c->n0; u.nl = c->nl; u.nlayers = L + 2;
  u.sumsq = c->sumsq + (long long)t * (L + 2) * 8; u.clip = c->cfg.clip_h; u.mu = c->cfg.momentum; u.eta = c->eta_dev; u.t = t;
  u.use_grad_norm = c->cfg.use_grad_norm; u.grad_norm_target = c->cfg.grad_norm_target;
}
int infer_update_step(cora_ctx* c, int t) {
  const int L = c->L;
  if (c->lazy_ok && t + 1 < c->lazy_T && !c->sens_live) {   // D = 64 fused path and wide2: the next inference launch applies it
    c->pend = t;
    return 0;
  }
  UpdateP u;
  fill_update(c, u, t);
  // last step of a loop with deferred updates: back into set 0, the gradient included (cora_get_vode_grad reads it there)
  if (c->cur) { u.h_out = c->Hs; u.m_out = c->Ms; u.g_out = c->Gs; }
  if (c->sens_live) {   // un-frozen sensory vode: one more leaf in the clip norm, then its own update
    if (c->slotted()) { u.sens_e = c->esum + ((long long)t * (L + 3) + (L + 2)) * 8; u.sens_n = 8; }
    else { u.sens_e = c->energies + (long long)t * (L + 3) + (L + 2); u.sens_n = 1; }
    u.sens_s = c->sens_scale;
  }
  {
    KScope s(c, K_UPDATE, 0, 20.0 * c->nV);
    launch_update(u, c->st);
  }
  c->cur = 0;
  if (c->sens_live) {
    KScope s(c, K_OTHER, 0, 20.0 * c->n0);
    launch_sens_update(u, c->xp, c->msens, err_ptr(c, L + 2), c->n0, c->st);
  }
  return 0;
}
int flush_lazy_update(cora_ctx* c) {
  if (c->pend < 0) return 0;
  const int t = c->pend, keep = c->lazy_T;
  c->pend = -1; c->lazy_T = 0;
  int rc = infer_update_step(c, t);
  c->lazy_T = keep;
  return rc;
}

int infer_step(cora_ctx* c, int t) {
  int rc = infer_grad_step(c, t);
  return rc ? rc : infer_update_step(c, t);
}

void zero_infer_logs(cora_ctx* c, int T) {
  const int L = c->L;
  launch_zero(c->energies, ((sizeof(double) * (size_t)T * (L + 3)) + 15) / 16 * 16, c->st);
  launch_zero(c->sumsq, sizeof(double) * (size_t)T * (L + 2) * 8, c->st);
  if (c->slotted()) launch_zero(c->esum, sizeof(double) * (size_t)T * (L + 3) * 8, c->st);
  launch_zero(c->Ms, sizeof(float) * (size_t)c->nV, c->st);   // optim_h.init, :414
}

int enqueue_infer(cora_ctx* c, int T) {
  // Zero-fill KERNELS, not hipMemsetAsync: memset nodes inside the captured graph were observed
  // (ROCm 7.2, gfx950) to leave the 100 MB momentum buffer stale on some replays, which blew the
  // vodes up at step 1 of a later batch; tests/test_gpu_parity.py::test_graph_replay_matches_eager
  // guards this.
  zero_infer_logs(c, T);
  c->lazy_T = T;   // steps 0 .. T-2 may hand their vode update to the next step's launch (D = 64 fused path)
  int rc = 0;
  for (int t = 0; t < T && !rc; ++t) rc = infer_step(c, t);
  c->lazy_T = 0;
  return rc;
}

// status "forward" chain (rule 'h -> u', :167-168,175-176): u_1 from h_0, each block reads the
// previous u.  If into_vodes, u_l is written into h_l (INIT rule 'h, u <- u').  Leaves z in c->z.
int chain_forward(cora_ctx* c, bool into_vodes, double* mse = nullptr) {
  const int L = c->L;
  if (c->fused) {
    FusedP f;
    fill_fused(c, f, nullptr, nullptr);
    f.mse = mse;
    double fl = (double)c->B * (4.0 * c->N * c->P * c->D + L * (24.0 * c->N * c->D * (double)c->D + 4.0 * c->N * (double)c->N * c->D));
    KScope s(c, K_OTHER, fl, 0);
    if (launch_chain(f, into_vodes ? 1 : 0, c->st)) FAIL(-3, "chain kernel launch failed");
    return 0;
  }
  // with vode LayerNorms the vode keeps u (INIT: h := u) and hands LN(u) on, :232-241
  auto hand_on = [&](const float* u, int i) -> const float* {
    if (!c->vln) return u;
    float* dst = c->hln + (long long)i * c->nl;
    KScope s(c, K_LN_FWD, 0, 8.0 * c->nl);
    launch_ln_fwd(u, c->W + c->oVg + (size_t)i * c->D, c->W + c->oVb + (size_t)i * c->D, c->D, dst,
                  c->stv + 2LL * i * c->BN, (int)c->BN, 1, c->D, c->st);
    return dst;
  };
  float* cur = into_vodes ? vode_ptr(c, c->Hs, 1) : c->ua;
  patch_proj(c, vode_ptr(c, c->Hs, 0), cur, OUT_STORE, nullptr, nullptr);
  static const bool no_chain = getenv("CORA_NO_WIDE_CHAIN") != nullptr;   // diagnostic: per-op chain on the wide path
  if (c->fusedw && L > 0 && !no_chain) {
    // one workgroup walks one image through all L blocks (fused_wide2.hip: wide2_chain_body, fused_wide.hip:
    // wide_chain_body); INIT keeps every u as the next vode, the reconstruction walks in place
    FusedP f;
    fill_fused(c, f, nullptr, nullptr);
    double fl = (double)c->B * L * (24.0 * c->N * c->D * (double)c->D + 4.0 * c->N * (double)c->N * c->D);
    {
      KScope s(c, K_OTHER, fl, 0);
      float* xo = into_vodes ? vode_ptr(c, c->Hs, 2) : cur;
      const long long stride = into_vodes ? c->nl : 0;
      if (c->fusedw2 ? launch_chain_wide2(f, L, c->D, cur, xo, stride, c->st) : launch_chain_wide(f, L, c->D, cur, xo, stride, c->st))
        FAIL(-3, "wide chain kernel launch failed");
    }
    out_proj(c, into_vodes ? vode_ptr(c, c->Hs, L + 1) : cur, c->z, OUT_STORE, nullptr, nullptr);
    return 0;
  }
  static const bool no_rows_chain = getenv("CORA_NO_ROWS_CHAIN") != nullptr;   // diagnostic / cross-check: per-op walk on the row-tile path
  if (c->frows && L > 0 && !no_rows_chain) {
    // row-tile path: each block of the walk is rows_qkv -> attention -> rows_mid (chain mode) instead of seven per-op launches
    const long long D = c->D, M = c->M;
    for (int i = 0; i < L; ++i) {
      float* nxt = into_vodes ? vode_ptr(c, c->Hs, i + 2) : (cur == c->ua ? c->ub : c->ua);
      RowsP r; fill_rows(c, r, nullptr, nullptr);
      r.L = 1; r.edge = 0; r.X = cur; r.E2 = nxt; r.delta = nullptr; r.dqa = nullptr;
      r.oLn1g += i * D; r.oLn1b += i * D; r.oLn2g += i * D; r.oLn2b += i * D;
      r.oWqkv += i * 3 * D * D; r.oWo += i * D * D; r.oW1 += i * M * D; r.ob1 += i * M; r.oW2 += i * D * M; r.ob2 += i * D;
      { KScope s(c, K_ROWS, 2.0 * c->BN * D * 3 * D, 4.0 * c->BN * 4 * D);
        if (launch_rows_qkv(r, 2, c->st)) FAIL(-3, "rows_qkv (chain): launch set-up failed"); }
      { KScope s(c, K_ATTN_FWD, 4.0 * c->B * (double)c->N * c->N * D, 4.0 * c->BN * 4 * D);
        if (c->cfg.dtype_mode == 1 && attn_fwd_bf16_supported(c->N, c->H, c->D)) {
          if (launch_attn_fwd_bf16(c->qkv, c->att, c->lse, c->B, c->N, c->H, c->D, c->st)) FAIL(-3, "attention (bf16): launch set-up failed");
        } else if (launch_attn_fwd(c->qkv, c->att, c->lse, c->B, c->N, c->H, c->D, c->st)) FAIL(-3, "attention: head dim not supported"); }
      { KScope s(c, K_ROWS, c->BN * (2.0 * D * D + 4.0 * D * M), 4.0 * c->BN * 3 * D);
        if (launch_rows_mid(r, 2, c->st)) FAIL(-3, "rows_mid (chain): launch set-up failed"); }
      cur = nxt;
    }
    out_proj(c, cur, c->z, OUT_STORE, nullptr, nullptr);
    return 0;
  }
  for (int i = 0; i < L; ++i) {
    float* nxt = into_vodes ? vode_ptr(c, c->Hs, i + 2) : (cur == c->ua ? c->ub : c->ua);
    int rc = block_forward(c, 1, i, hand_on(cur, i), nxt, OUT_STORE, nullptr, nullptr);
    if (rc) return rc;
    cur = nxt;
  }
  out_proj(c, hand_on(cur, L), c->z, OUT_STORE, nullptr, nullptr);
  return 0;
}

int upload_x(cora_ctx* c, const float* x, int on_device) {
  if (!x) FAIL(-1, "x is NULL");
  // host x is staged through ximg; device x is patchified where it lies (every caller of upload_x ends its
  // entry point with a stream synchronisation or documents that x must stay valid until the step is settled)
  const float* src = x;
  if (!on_device) {
    HCHECK(hipMemcpyAsync(c->ximg, x, sizeof(float) * (size_t)c->nImg, hipMemcpyHostToDevice, c->st));
    src = c->ximg;
  }
  KScope s(c, K_OTHER, 0, 8.0 * c->nImg);
  launch_patchify(src, c->xp, c->B, c->cfg.F, c->cfg.C, c->cfg.Himg, c->cfg.Wimg, c->cfg.patch, 0, c->st);
  return 0;
}

int pick_splitk(long long tiles, int K) {
  long long s = 1024 / (tiles > 0 ? tiles : 1);
  long long kmax = K / 64;
  if (s > kmax) s = kmax;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return (int)s;
}

// C[g] (+)= -A^T B  with A [rows, R], B [rows, O]  (weight gradient, summed over the rows)
GemmP wgrad_problem(cora_ctx* c, int G, const float* A, int R, long long sA, const float* Bm, int O, long long sB,
                    float* C, long long sC, int b_gelu) {
  GemmP p = gemm_base(c, 2);
  p.groups = G;
  p.A = A; p.Bw = Bm; p.C = C;
  p.R = R; p.O = O; p.K = (int)c->BN; p.lda = R; p.ldb = O; p.ldc = O;
  p.sA = sA; p.sB = sB; p.sC = sC;
  p.alpha = -1.f; p.mode = OUT_ATOMIC; p.b_gelu = b_gelu;
  return p;
}
void wgrad_gemm(cora_ctx* c, int G, const float* A, int R, long long sA, const float* Bm, int O, long long sB,
                float* C, long long sC, int b_gelu) {
  GemmP p = wgrad_problem(c, G, A, R, sA, Bm, O, sB, C, sC, b_gelu);
  long long tiles = (long long)((R + 127) / 128) * ((O + 63) / 64) * G;
  p.splitk = pick_splitk(tiles, p.K);
  run_gemm(c, p, true, true);
}

}  // namespace

// =================================================================================================
extern "C" {

const char* cora_last_error(const cora_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

// configuration checks shared by cora_create and cora_forward_static: nullptr = fine
static const char* check_config(const cora_config* cfg) {
  if (cfg->dtype_mode < 0 || cfg->dtype_mode > 2) return "dtype_mode: 0 (fp32 MFMA, parity path), 1 (bf16 MFMA operands, fp32 accumulate and state) or 2 (three-term bf16 split in the GEMMs: fp32-grade products on the bf16 pipe)";
  if (cfg->N < 1 || cfg->P < 4 || cfg->D < 4 || cfg->H < 1 || cfg->L < 0 || cfg->B < 1 || cfg->M < 4) return "non-positive dimension";
  if (cfg->D % cfg->H) return "hidden_size must be divisible by num_heads";
  if (cfg->D % 4 || cfg->P % 4 || cfg->M % 4) return "D, P and M must be multiples of 4 (16-byte rows)";
  if (cfg->L + 3 > 64) return "too many blocks (L <= 61)";
  {   // the GEMM epilogue addresses one group's output with 32-bit element offsets
    long long widest = cfg->M > 3 * cfg->D ? cfg->M : 3 * cfg->D;
    if (cfg->P > widest) widest = cfg->P;
    if ((long long)cfg->B * cfg->N * widest >= (1LL << 31)) return "B*N*max(M, 3D, P) must stay below 2^31 elements";
  }
  if (cfg->P != cfg->patch * cfg->patch * cfg->C) return "P != patch*patch*C";
  if (cfg->F < 1 || cfg->patch < 1 || cfg->N != cfg->F * (cfg->Himg / cfg->patch) * (cfg->Wimg / cfg->patch)) return "N != F*(H/p)*(W/p)";
  if (cfg->L > 0) {
    int dh = cfg->D / cfg->H;
    if (!(dh == 16 || dh == 24 || dh == 32 || dh == 48 || dh == 64 || dh == 96 || dh == 128)) return "head dim must be one of 16,24,32,48,64,96,128";
    if (cfg->D > 1024) return "hidden_size > 1024 not supported by the LayerNorm kernels";
  }
  return nullptr;
}

// dims and the weight-arena layout (per-type stacks, then the two projections, then the vode LayerNorms)
static void layout_ctx(cora_ctx* c, const cora_config* cfg, int device) {
  c->cfg = *cfg; c->device = device;
  c->N = cfg->N; c->P = cfg->P; c->D = cfg->D; c->H = cfg->H; c->M = cfg->M; c->L = cfg->L; c->B = cfg->B;
  c->BN = (long long)c->B * c->N;
  c->n0 = c->BN * c->P; c->nl = c->BN * c->D; c->nV = c->n0 + (long long)(c->L + 1) * c->nl;
  c->Tmax = cfg->max_T > 0 ? cfg->max_T : 64;
  c->nImg = (long long)c->B * cfg->F * cfg->C * cfg->Himg * cfg->Wimg;
  const size_t D = c->D, P = c->P, M = c->M, L = c->L;
  size_t o = 0;
  c->oLn1g = o; o += L * D; c->oLn1b = o; o += L * D; c->oLn2g = o; o += L * D; c->oLn2b = o; o += L * D;
  c->oWqkv = o; o += L * 3 * D * D; c->oWo = o; o += L * D * D;
  c->oW1 = o; o += L * M * D; c->ob1 = o; o += L * M; c->oW2 = o; o += L * D * M; c->ob2 = o; o += L * D;
  c->oWp = o; o += D * P; c->obp = o; o += D; c->oWout = o; o += P * D; c->obout = o; o += P;
  c->vln = cfg->use_vode_ln != 0;
  if (c->vln) { c->oVg = o; o += (size_t)(L + 1) * D; c->oVb = o; o += (size_t)(L + 1) * D; }
  c->nW = o;
}

int cora_create(const cora_config* cfg, int device, cora_ctx** out) {
  if (!cfg || !out) { g_create_err = "null argument"; return -1; }
  *out = nullptr;
  auto bad = [&](const char* m) { g_create_err = m; return -1; };
  if (const char* e = check_config(cfg)) return bad(e);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return bad("no HIP device visible");
  if (device < 0 || device >= ndev) return bad("device index out of range");
  if (hipSetDevice(device) != hipSuccess) return bad("hipSetDevice failed");

  cora_ctx* c = new cora_ctx();
  layout_ctx(c, cfg, device);
  c->use_graph = getenv("CORA_NO_GRAPH") == nullptr;
  auto fail = [&](int rc) { g_create_err = c->err; cora_destroy(c); return rc; };
  if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess) { c->err = "hipStreamCreate failed"; return fail(-2); }
  hipEventCreate(&c->ev0); hipEventCreate(&c->ev1); hipEventCreate(&c->ev2);
  const size_t D = c->D, P = c->P, M = c->M, L = c->L;

  int rc = 0;
#define DA(ptr, n) if ((rc = dalloc(c, &(ptr), (long long)(n)))) return fail(rc)
  DA(c->W, c->nW); DA(c->dW, c->nW); DA(c->amu, c->nW); DA(c->anu, c->nW);
  DA(c->pos, (long long)c->N * D);
  c->fused = (c->N == 64 && c->D == 64 && (c->H == 1 || c->H == 2 || c->H == 4) && c->M == 256 && c->P == 48 && c->L > 0 && cfg->dtype_mode == 0 && !c->vln && getenv("CORA_NO_FUSED") == nullptr);
  c->fusedw = (!c->fused && c->L > 0 && cfg->dtype_mode == 0 && !c->vln && fused_wide_supported(c->N, c->P, c->D, c->H, c->M) &&
               getenv("CORA_NO_FUSED") == nullptr);
  c->fusedw2 = c->fusedw && fused_wide2_supported(c->N, c->P, c->D, c->H, c->M) && getenv("CORA_WIDE1") == nullptr;
  // bf16 mode (dtype_mode 1) takes the row kernels too where its bf16-MFMA attention applies (sequences longer than one
  // chunk): the row-local GEMMs then run on the fp32 MFMA kernels -- more accurate than the mode asks for, and at
  // D = 64 faster than the per-op bf16 GEMMs they replace -- with the bf16 attention kernels between them
  const bool rows_dtype = cfg->dtype_mode == 0 ||
                          (cfg->dtype_mode == 1 && attn_fwd_bf16_supported(c->N, c->H, c->D) && attn_bwd_bf16_supported(c->N, c->H, c->D));
  c->frows = (!c->fused && !c->fusedw && c->L > 0 && rows_dtype && !c->vln && fused_rows_supported(c->BN, c->D, c->M) &&
              getenv("CORA_NO_FUSED") == nullptr && getenv("CORA_NO_ROWS") == nullptr);
  c->frows_bwd1 = c->frows && cfg->dtype_mode == 0 && (c->H == 1 || c->H == 2 || c->H == 4) &&
                  attn_bwd_fused_supported(c->N, c->H, c->D) && getenv("CORA_ATTN_BWD2") == nullptr;
  if (c->fused || c->fusedw || c->frows || cfg->dtype_mode != 0) {   // transposed weight copies: fused kernels' backward GEMMs; bf16 mode's backward-data GEMMs
    size_t t = 0;
    c->tWqkv = t; t += L * 3 * D * D; c->tWo = t; t += L * D * D; c->tW1 = t; t += L * M * D; c->tW2 = t; t += L * D * M;
    c->tWp = t; t += D * P; c->tWout = t; t += P * D;
    c->nWT = t;
    DA(c->WT, c->nWT);
    if (cfg->dtype_mode != 0) {
      float* tmpb = nullptr;
      DA(tmpb, (long long)(c->nW + 1) / 2 + 4); c->Wh = (unsigned short*)tmpb;
      DA(tmpb, (long long)(c->nW + 1) / 2 + 4); c->Wl = (unsigned short*)tmpb;
      DA(tmpb, (long long)(c->nWT + 1) / 2 + 4); c->WTh = (unsigned short*)tmpb;
      if (cfg->dtype_mode == 2) {
        DA(tmpb, (long long)(c->nW + 1) / 2 + 4); c->W3 = (unsigned short*)tmpb;
        DA(tmpb, (long long)(c->nWT + 1) / 2 + 4); c->WTl = (unsigned short*)tmpb;
        DA(tmpb, (long long)(c->nWT + 1) / 2 + 4); c->WT3 = (unsigned short*)tmpb;
      }
    }
    if (cfg->dtype_mode == 1 && c->WTh && L > 0 && !getenv("CORA_NO_BWD16") && !getenv("CORA_BF16_NO_PRE") && !getenv("CORA_BF16_INREG")) {
      c->da16 = (M % 8 == 0);
      c->dqkv16 = ((3 * D) % 8 == 0) && attn_bwd_bf16_supported(c->N, c->H, c->D);
    }
    if (c->fused) DA(c->colpart, (long long)fused_colpart_floats(L, c->B));
    if (c->frows) DA(c->colpart, (long long)fused_rows_colpart_floats((int)L, c->BN));
    if (c->frows_bwd1) DA(c->dqparts, (long long)attn_bwd_fused_parts(c->N) * (long long)L * c->nl);
  }
  DA(c->Hs, c->nV); DA(c->Ms, c->nV); DA(c->Gs, c->nV);
  // second arena set of the deferred vode update (see cora_ctx::Hs2); the experimental 8-wave kernel has no such form
  if ((c->fused || c->fusedw2) && !getenv("CORA_NO_LAZY") && !fused_pair_wanted((int)L, c->B)) {
    DA(c->Hs2, c->nV); DA(c->Ms2, c->nV); DA(c->Gs2, c->nV);
    c->lazy_ok = true;
  }
  DA(c->Es, (long long)(L + 1) * c->nl + c->n0);
  if (c->vln) { DA(c->hln, (long long)(L + 1) * c->nl); DA(c->stv, 2LL * (L + 1) * c->BN); DA(c->dxr, (long long)(L + 1) * c->nl); }
  DA(c->xp, c->n0); DA(c->ximg, c->nImg);
  c->Gmax = c->L > 0 ? c->L : 1;
  const long long rows = (long long)c->Gmax * c->BN;
  DA(c->xn1, rows * D); DA(c->st1, rows * 2); DA(c->qkv, rows * 3 * D); DA(c->att, rows * D); DA(c->lse, rows * c->H);
  DA(c->y, rows * D); DA(c->xn2, rows * D); DA(c->st2, rows * 2); DA(c->a, rows * M);
  DA(c->da, rows * M); DA(c->dzn, rows * D); DA(c->dy, rows * D); DA(c->dO, rows * D); DA(c->delta, rows * c->H);
  DA(c->dqkv, rows * 3 * D); DA(c->dxn, rows * D);
  DA(c->ua, c->nl); DA(c->ub, c->nl); DA(c->z, c->n0);
  if (c->fused) DA(c->stash, rows * c->H * D);
  if (c->fusedw) DA(c->stash, rows * (c->H * 64 + 3 * D));   // per token row: one 64-key row of P per head, then q (| k | v: two-workgroup form)
  DA(c->gscale, 64); DA(c->eta_dev, c->Tmax + 1);
  float* tmp = nullptr;
  DA(tmp, 2LL * (c->Tmax + 2) * (L + 3)); c->energies = (double*)tmp;
  DA(tmp, 16LL * (c->Tmax + 2) * (L + 2)); c->sumsq = (double*)tmp;
  DA(tmp, 16LL * (c->Tmax + 2) * (L + 3)); c->esum = (double*)tmp;
  DA(tmp, 8); c->scal = (double*)tmp;
  DA(c->hyp_dev, 4 + 4LL * c->Tmax);   // [0..3] single step; then one {eta_w, bc1, bc2, -} per inference step (iPC mode)
  if (hipHostMalloc((void**)&c->host_out, sizeof(double) * ((size_t)(L + 3) * 9 + 8), hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&c->host_in, sizeof(float) * ((size_t)c->Tmax + 8), hipHostMallocDefault) != hipSuccess)
    return fail(-2);
#undef DA
  if (apply_grad_options(c)) { c->err = "initial upload failed"; return fail(-2); }
  { float* sp = nullptr; if (int rc2 = static_scratch_ptr(device, &sp, &c->err)) return fail(rc2); }
  // Launch every kernel of the inference step once outside stream capture (code-object load,
  // dynamic-LDS attributes).  All state is zero and eta = 0, so nothing observable changes.
  rc = enqueue_infer(c, 1);
  if (rc) return fail(rc);
  if (hipStreamSynchronize(c->st) != hipSuccess || hipGetLastError() != hipSuccess) { c->err = "warm-up step failed"; return fail(-2); }
  c->warmed = true;
  *out = c;
  return 0;
}

int cora_destroy(cora_ctx* c) {
  if (!c) return 0;
  hipSetDevice(c->device);
  if (c->st) hipStreamSynchronize(c->st);
  for (auto& kv : c->graphs) hipGraphExecDestroy(kv.second);
  for (auto& kv : c->dist_graphs) hipGraphExecDestroy(kv.second);
  drop_learn_graphs(c);
  if (c->comm && g_rccl.CommDestroy) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
  for (void* p : c->allocs) hipFree(p);
  if (c->host_out) hipHostFree(c->host_out);
  if (c->host_in) hipHostFree(c->host_in);
  if (c->ev0) hipEventDestroy(c->ev0);
  if (c->ev1) hipEventDestroy(c->ev1);
  if (c->ev2) hipEventDestroy(c->ev2);
  if (c->st) hipStreamDestroy(c->st);
  delete c;
  return 0;
}

// Diagnostic (stamps build): allocate / fetch the per-workgroup phase stamps of the fused kernel.
extern "C" int cora_dbg_stamps(cora_ctx* c, unsigned long long* host, size_t n) {
  if (!c) return -1;
  size_t need = (size_t)(c->L + 1) * c->B * 16;
  if (!c->dbg) {
    void* v = nullptr;
    HCHECK(hipMalloc(&v, need * 8));
    HCHECK(hipMemset(v, 0, need * 8));
    c->allocs.push_back(v);
    c->dbg = (unsigned long long*)v;
    for (auto& kv : c->graphs) hipGraphExecDestroy(kv.second);
    c->graphs.clear();
    return 0;
  }
  if (host && n == need) {
    HCHECK(hipStreamSynchronize(c->st));
    HCHECK(hipMemcpy(host, c->dbg, need * 8, hipMemcpyDeviceToHost));
  }
  return 0;
}

int cora_kernel_path(const cora_ctx* c) { return c ? (c->fused ? 1 : (c->fusedw ? 2 : (c->frows ? 3 : 0))) : -1; }

int cora_sync(cora_ctx* c) {
  if (!c) return -1;
  HCHECK(hipStreamSynchronize(c->st));
  return 0;
}

int cora_weights_size(const cora_ctx* c, size_t* n) {
  if (!c || !n) return -1;
  *n = c->nW;
  return 0;
}

int cora_weight_offset(const cora_ctx* cc, const char* key, size_t* offset, size_t* count) {
  cora_ctx* c = const_cast<cora_ctx*>(cc);
  if (!c || !key || !offset || !count) return -1;
  const size_t D = c->D, P = c->P, M = c->M;
  std::string k(key);
  auto set = [&](size_t o, size_t n) { *offset = o; *count = n; return 0; };
  if (k == ".patch_projection.nn.weight") return set(c->oWp, D * P);
  if (k == ".patch_projection.nn.bias") return set(c->obp, D);
  if (k == ".output_projection.nn.weight") return set(c->oWout, P * D);
  if (k == ".output_projection.nn.bias") return set(c->obout, P);
  if (c->vln) {
    int vi = -1; char leaf[16] = {0};
    if (sscanf(k.c_str(), ".vode_output_layernorms[%d].nn.%15s", &vi, leaf) == 2 && vi >= 0 && vi <= c->L) {
      if (!strcmp(leaf, "weight")) return set(c->oVg + (size_t)vi * D, D);
      if (!strcmp(leaf, "bias")) return set(c->oVb + (size_t)vi * D, D);
    }
  }
  const char* pre = ".transformer_blocks[";
  if (k.compare(0, strlen(pre), pre) == 0) {
    size_t close = k.find(']');
    if (close == std::string::npos) FAIL(-4, "malformed key: " + k);
    char* endp = nullptr;
    long i = strtol(k.c_str() + strlen(pre), &endp, 10);
    if (endp != k.c_str() + close || i < 0 || i >= c->L) FAIL(-4, "block index out of range: " + k);
    std::string leaf = k.substr(close + 1);
    if (leaf == ".layer_norm1.nn.weight") return set(c->oLn1g + i * D, D);
    if (leaf == ".layer_norm1.nn.bias") return set(c->oLn1b + i * D, D);
    if (leaf == ".layer_norm2.nn.weight") return set(c->oLn2g + i * D, D);
    if (leaf == ".layer_norm2.nn.bias") return set(c->oLn2b + i * D, D);
    if (leaf == ".attention.nn.query_proj.weight") return set(c->oWqkv + i * 3 * D * D, D * D);
    if (leaf == ".attention.nn.key_proj.weight") return set(c->oWqkv + i * 3 * D * D + D * D, D * D);
    if (leaf == ".attention.nn.value_proj.weight") return set(c->oWqkv + i * 3 * D * D + 2 * D * D, D * D);
    if (leaf == ".attention.nn.output_proj.weight") return set(c->oWo + i * D * D, D * D);
    if (leaf == ".linear1.nn.weight") return set(c->oW1 + i * M * D, M * D);
    if (leaf == ".linear1.nn.bias") return set(c->ob1 + i * M, M);
    if (leaf == ".linear2.nn.weight") return set(c->oW2 + i * D * M, D * M);
    if (leaf == ".linear2.nn.bias") return set(c->ob2 + i * D, D);
  }
  FAIL(-4, "unknown parameter key: " + k);
}

static int copy_in(cora_ctx* c, float* dst, const float* host, size_t n, size_t expect) {
  if (!host) FAIL(-1, "null host pointer");
  if (n != expect) FAIL(-1, "size mismatch");
  HCHECK(hipSetDevice(c->device));
  if (int rc_ = settle_pending(c)) return rc_;
  HCHECK(hipMemcpyAsync(dst, host, n * sizeof(float), hipMemcpyHostToDevice, c->st));
  HCHECK(hipStreamSynchronize(c->st));
  return 0;
}
static int copy_out(cora_ctx* c, float* host, const float* src, size_t n, size_t expect) {
  if (!host) FAIL(-1, "null host pointer");
  if (n != expect) FAIL(-1, "size mismatch");
  HCHECK(hipSetDevice(c->device));
  if (int rc_ = settle_pending(c)) return rc_;
  HCHECK(hipMemcpyAsync(host, src, n * sizeof(float), hipMemcpyDeviceToHost, c->st));
  HCHECK(hipStreamSynchronize(c->st));
  return 0;
}

int cora_set_weights(cora_ctx* c, const float* host, size_t n) {
  if (!c) return -1;
  int rc = copy_in(c, c->W, host, n, c->nW);
  if (rc) return rc;
  refresh_wt(c);
  HCHECK(hipStreamSynchronize(c->st));
  return 0;
}
int cora_get_weights(cora_ctx* c, float* host, size_t n) { if (!c) return -1; return copy_out(c, host, c->W, n, c->nW); }
int cora_set_pos(cora_ctx* c, const float* host, size_t n) { if (!c) return -1; return copy_in(c, c->pos, host, n, (size_t)c->N * c->D); }

int cora_vode_size(const cora_ctx* c, int l, size_t* n) {
  if (!c || !n || l < 0 || l > c->L + 1) return -1;
  *n = (size_t)(l == 0 ? c->n0 : c->nl);
  return 0;
}
int cora_set_vode(cora_ctx* c, int l, const float* host, size_t n) {
  if (!c) return -1;
  if (l < 0 || l > c->L + 1) FAIL(-1, "vode index out of range (the sensory vode is set through x)");
  return copy_in(c, vode_ptr(c, c->Hs, l), host, n, (size_t)(l == 0 ? c->n0 : c->nl));
}
int cora_get_vode(cora_ctx* c, int l, float* host, size_t n) {
  if (!c) return -1;
  if (l < 0 || l > c->L + 1) FAIL(-1, "vode index out of range");
  return copy_out(c, host, vode_ptr(c, c->Hs, l), n, (size_t)(l == 0 ? c->n0 : c->nl));
}
int cora_get_vode_grad(cora_ctx* c, int l, float* host, size_t n) {
  if (!c) return -1;
  if (l < 0 || l > c->L + 1) FAIL(-1, "vode index out of range");
  size_t cnt = (size_t)(l == 0 ? c->n0 : c->nl);
  int rc = copy_out(c, host, vode_ptr(c, c->Gs, l), n, cnt);
  if (rc) return rc;
  if (c->cfg.use_grad_norm && c->last_T > 0) {   // normalize_vode_h_gradients, :353-389
    double s8[8], ss = 0.0;
    HCHECK(hipMemcpy(s8, c->sumsq + ((long long)(c->last_T - 1) * (c->L + 2) + l) * 8, sizeof s8, hipMemcpyDeviceToHost));
    for (double v : s8) ss += v;
    float m = (float)((double)c->cfg.grad_norm_target / (sqrt(ss) + 1e-8));
    for (size_t i = 0; i < cnt; ++i) host[i] *= m;
  }
  return 0;
}

int cora_init_vodes(cora_ctx* c, const float* h0) {
  if (!c) return -1;
  HCHECK(hipSetDevice(c->device));
  if (int rc_ = settle_pending(c)) return rc_;
  if (h0) HCHECK(hipMemcpyAsync(c->Hs, h0, sizeof(float) * (size_t)c->n0, hipMemcpyHostToDevice, c->st));
  else HCHECK(hipMemsetAsync(c->Hs, 0, sizeof(float) * (size_t)c->n0, c->st));
  int rc = chain_forward(c, true);
  if (rc) return rc;
  HCHECK(hipStreamSynchronize(c->st));
  return 0;
}

int cora_set_adam_state(cora_ctx* c, const float* mu, const float* nu, int64_t count) {
  if (!c) return -1;
  int rc = copy_in(c, c->amu, mu, c->nW, c->nW);
  if (rc) return rc;
  rc = copy_in(c, c->anu, nu, c->nW, c->nW);
  if (rc) return rc;
  c->adam_count = count;
  return 0;
}
int cora_get_adam_state(cora_ctx* c, float* mu, float* nu, int64_t* count) {
  if (!c) return -1;
  int rc = copy_out(c, mu, c->amu, c->nW, c->nW);
  if (rc) return rc;
  rc = copy_out(c, nu, c->anu, c->nW, c->nW);
  if (rc) return rc;
  if (count) *count = c->adam_count;
  return 0;
}

int cora_set_optim(cora_ctx* c, float clip_h, float momentum, float clip_w, float b1, float b2, float eps, float wd) {
  if (!c) return -1;
  cora_config& f = c->cfg;
  bool graph_dirty = (f.clip_h != clip_h) || (f.momentum != momentum);
  bool learn_dirty = (f.clip_w != clip_w) || (f.adam_b1 != b1) || (f.adam_b2 != b2) || (f.adam_eps != eps) || (f.adam_wd != wd);
  f.clip_h = clip_h; f.momentum = momentum; f.clip_w = clip_w;
  f.adam_b1 = b1; f.adam_b2 = b2; f.adam_eps = eps; f.adam_wd = wd;
  if (graph_dirty) {
    HCHECK(hipSetDevice(c->device));
    HCHECK(hipStreamSynchronize(c->st));
    for (auto& kv : c->graphs) hipGraphExecDestroy(kv.second);
    c->graphs.clear();
    for (auto& kv : c->dist_graphs) hipGraphExecDestroy(kv.second);
    c->dist_graphs.clear();
  }
  if (learn_dirty) {
    HCHECK(hipSetDevice(c->device));
    drop_learn_graphs(c);
  }
  return 0;
}

int cora_set_grad_options(cora_ctx* c, int use_scaling, float scale_base, int use_grad_norm, float grad_norm_target) {
  if (!c) return -1;
  cora_config& f = c->cfg;
  if (f.use_scaling == use_scaling && f.scale_base == scale_base && f.use_grad_norm == use_grad_norm &&
      f.grad_norm_target == grad_norm_target) return 0;
  HCHECK(hipSetDevice(c->device));
  HCHECK(hipStreamSynchronize(c->st));
  f.use_scaling = use_scaling; f.scale_base = scale_base; f.use_grad_norm = use_grad_norm; f.grad_norm_target = grad_norm_target;
  if (apply_grad_options(c)) FAIL(-2, "layer-scale upload failed");
  return 0;
}

int infer_graph(cora_ctx* c, int T, hipGraphExec_t* out);

int cora_infer(cora_ctx* c, const float* x, int x_on_device, int T, const float* eta_h, cora_step_out* out) {
  if (!c) return -1;
  if (T < 1 || T > c->Tmax) FAIL(-1, "T out of range (1..max_T)");
  if (!eta_h) FAIL(-1, "eta_h is NULL");
  HCHECK(hipSetDevice(c->device));
  // out == NULL (graph mode): nothing is read back here, so the call returns with the loop still in
  // flight and cora_wgrad, which must follow, ends it -- one host synchronisation less per batch on
  // the data-parallel path.  eta_h then goes through the ctx's pinned staging buffer.
  const bool deferred = !out && c->use_graph && !c->prof;
  if (int rc_ = settle_pending(c)) return rc_;
  int rc = upload_x(c, x, x_on_device);
  if (rc) return rc;
  if (deferred) {
    memcpy(c->host_in, eta_h, sizeof(float) * T);
    HCHECK(hipMemcpyAsync(c->eta_dev, c->host_in, sizeof(float) * T, hipMemcpyHostToDevice, c->st));
  } else {
    HCHECK(hipMemcpyAsync(c->eta_dev, eta_h, sizeof(float) * T, hipMemcpyHostToDevice, c->st));
    HCHECK(hipStreamSynchronize(c->st));   // eta_h / x are caller-owned: copied before we return
  }
  c->last_T = T;
  if (!c->use_graph || c->prof) {
    HCHECK(hipEventRecord(c->ev0, c->st));
    rc = enqueue_infer(c, T);
    if (rc) return rc;
    HCHECK(hipEventRecord(c->ev1, c->st));
  } else {
    hipGraphExec_t gi = nullptr;
    if ((rc = infer_graph(c, T, &gi))) return rc;
    HCHECK(hipEventRecord(c->ev0, c->st));
    HCHECK(hipGraphLaunch(gi, c->st));
    HCHECK(hipEventRecord(c->ev1, c->st));
  }
  if (deferred) { c->infer_pending = true; return 0; }
  HCHECK(hipStreamSynchronize(c->st));
  if (out) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, c->ev0, c->ev1);
    out->infer_ms = ms;
    std::vector<double> en((size_t)c->L + 3);
    if (energy_row(c, T - 1, en.data())) FAIL(-2, "energy read-back failed");
    double tot = 0.0;
    for (double v : en) tot += v;
    out->final_h_energy = tot;
  }
  return 0;
}

// weight gradients at the final vodes (:482-483) into dW; no synchronisation, capturable
int enqueue_wgrad(cora_ctx* c) {
  const int L = c->L, D = c->D, M = c->M, P = c->P;
  const int R = (int)c->BN;
  const int slot = c->Tmax;
  launch_zero(c->energies + (long long)slot * (L + 3), (sizeof(double) * (L + 3) + 15) / 16 * 16, c->st);
  launch_zero(c->dW, (sizeof(float) * c->nW + 15) / 16 * 16, c->st);
  int rc = 0;
  if (c->fused) {
    // one fused launch evaluates E at the final vodes (:482-483) and leaves every tensor the
    // weight-gradient contractions need; its vode-gradient outputs go to scratch (final_h_grad stays)
    FusedP f;
    fill_fused(c, f, c->energies + (long long)slot * (L + 3), c->sumsq + (long long)(c->Tmax + 1) * (L + 2) * 8);
    f.BP1 = c->qkv; f.G0 = c->ua; f.dW = c->dW; f.colpart = c->colpart;
    KScope s(c, K_FUSED, block_step_flops(c), 0);
    if (launch_fused_block(f, L, 1, c->st)) FAIL(-3, "fused block kernel: launch setup failed");
    launch_colpart_reduce(f, L, c->st);
  } else {
    rc = errors_pass(c, slot, c->frows ? 1 : 0);   // learning_step_fn evaluates E at the final vodes, :482-483
    if (rc) return rc;
    if (c->frows) {   // backward half with the LayerNorm-1 column sums, then every bias / LayerNorm gradient of the blocks
      if ((rc = block_backward_rows(c, nullptr, 1))) return rc;
      RowsP r; fill_rows(c, r, nullptr, nullptr);
      KScope s(c, K_OTHER, 0, 4.0 * fused_rows_colpart_floats(L, c->BN));
      launch_rows_colpart_reduce(r, c->dW, c->st);
    }
  }
  // dE/dtheta = -(du/dtheta)^T e
  if (c->fused || c->frows) {
    // the six contractions (K = B*N rows each, 56 output tiles in all) go out as ONE launch, split
    // over K so that the launch has ~1k workgroups; launched one by one each filled a fraction of
    // the chip (64..768 workgroups) and they took 256 us together
    const float* r = err_ptr(c, 2);
    const long long sD = (long long)R * D, sM = (long long)R * M, s3 = (long long)R * 3 * D;
    GemmMulti m; memset(&m, 0, sizeof m);
    m.n = 6;
    m.p[0] = wgrad_problem(c, L, r, D, sD, c->a, M, sM, c->dW + c->oW2, (long long)D * M, 0);        // dW2 = -r^T gelu(a); the STORE pass left gelu(a) in c->a
    m.p[1] = wgrad_problem(c, L, c->da, M, sM, c->xn2, D, sD, c->dW + c->oW1, (long long)M * D, 0);   // dW1 = -da^T LN2(y)
    m.p[2] = wgrad_problem(c, L, c->dqkv, 3 * D, s3, c->xn1, D, sD, c->dW + c->oWqkv, 3LL * D * D, 0);
    m.p[3] = wgrad_problem(c, L, c->dy, D, sD, c->att, D, sD, c->dW + c->oWo, (long long)D * D, 0);   // dWo = -dy^T att
    m.p[4] = wgrad_problem(c, 1, err_ptr(c, 1), D, 0, vode_ptr(c, c->Hs, 0), P, 0, c->dW + c->oWp, 0, 0);
    m.p[5] = wgrad_problem(c, 1, err_ptr(c, L + 2), P, 0, vode_ptr(c, c->Hs, L + 1), D, 0, c->dW + c->oWout, 0, 0);
    long long tiles = 0;
    double fl = 0.0;
    for (int k = 0; k < m.n; ++k) {
      tiles += (long long)((m.p[k].R + 127) / 128) * ((m.p[k].O + 63) / 64) * m.p[k].groups;
      fl += 2.0 * m.p[k].R * (double)m.p[k].O * m.p[k].K * m.p[k].groups;
    }
    const int sk = pick_splitk(tiles, (int)c->BN);
    for (int k = 0; k < m.n; ++k) m.p[k].splitk = sk;
    {
      KScope s(c, K_GEMM_TN, fl, 0);
      launch_gemm_multi_tn(m, c->st);
    }
    if (c->frows) {   // the edge layers of this path are per-op GEMMs: their bias gradients are plain column sums
      KScope s(c, K_OTHER, 0, 4.0 * R * (D + P));
      launch_colsum(err_ptr(c, 1), nullptr, nullptr, c->dW + c->obp, 0, R, 1, D, -1.f, c->st);
      launch_colsum(err_ptr(c, L + 2), nullptr, nullptr, c->dW + c->obout, 0, R, 1, P, -1.f, c->st);
    }
    return 0;
  }
  wgrad_gemm(c, 1, err_ptr(c, 1), D, 0, vode_ptr(c, c->Hs, 0), P, 0, c->dW + c->oWp, 0, 0);
  if (!c->fused) { KScope s(c, K_OTHER, 0, 4.0 * R * D); launch_colsum(err_ptr(c, 1), nullptr, nullptr, c->dW + c->obp, 0, R, 1, D, -1.f, c->st); }
  wgrad_gemm(c, 1, err_ptr(c, L + 2), P, 0, vode_in(c, L + 1), D, 0, c->dW + c->oWout, 0, 0);
  if (!c->fused) { KScope s(c, K_OTHER, 0, 4.0 * R * P); launch_colsum(err_ptr(c, L + 2), nullptr, nullptr, c->dW + c->obout, 0, R, 1, P, -1.f, c->st); }
  if (L > 0) {
    const float* r = err_ptr(c, 2);
    const float* xin = vode_in(c, 1);
    if (!c->fused) {
      // with vode LayerNorms the cotangents dE/dx' are needed too (their gamma / beta gradients)
      rc = block_backward(c, L, 0, xin, r, nullptr, c->vln ? c->dxr : nullptr, nullptr, c->vln);
      if (rc) return rc;
    }
    const long long sD = (long long)R * D, sM = (long long)R * M, s3 = (long long)R * 3 * D;
    wgrad_gemm(c, L, r, D, sD, c->a, M, sM, c->dW + c->oW2, (long long)D * M, 1);       // dW2 = -r^T gelu(a)
    wgrad_gemm(c, L, c->da, M, sM, c->xn2, D, sD, c->dW + c->oW1, (long long)M * D, 0);  // dW1 = -da^T LN2(y)
    wgrad_gemm(c, L, c->dy, D, sD, c->att, D, sD, c->dW + c->oWo, (long long)D * D, 0);  // dWo = -dy^T att
    wgrad_gemm(c, L, c->dqkv, 3 * D, s3, c->xn1, D, sD, c->dW + c->oWqkv, 3LL * D * D, 0);
    if (!c->fused) {   // the fused STORE kernel leaves these column sums itself
      KScope s(c, K_OTHER, 0, 4.0 * L * R * (6.0 * D + M));
      launch_colsum(r, nullptr, nullptr, c->dW + c->ob2, D, R, L, D, -1.f, c->st);
      launch_colsum(c->da, nullptr, nullptr, c->dW + c->ob1, M, R, L, M, -1.f, c->st);
      launch_colsum(c->dzn, c->y, c->st2, c->dW + c->oLn2g, D, R, L, D, -1.f, c->st, c->dW + c->oLn2b);   // d gamma and d beta in one pass
      launch_colsum(c->dxn, xin, c->st1, c->dW + c->oLn1g, D, R, L, D, -1.f, c->st, c->dW + c->oLn1b);
    }
  }
  if (c->vln) {   // vode LayerNorm parameters: d gamma_v = -sum dx'_v * xhat_v, d beta_v = -sum dx'_v
    GemmP p = gemm_base(c, 1);
    p.A = err_ptr(c, L + 2); p.C = c->dxr + (long long)L * c->nl;
    p.R = R; p.O = D; p.K = P; p.lda = P; p.ldc = D; p.mode = OUT_STORE;
    run_gemm(c, p, false, bwd_weight(c, p, c->oWout, c->tWout));  // dx'_{L+1} = e_img Wout
    KScope s(c, K_OTHER, 0, 12.0 * (L + 1) * R * D);
    launch_colsum(c->dxr, vode_ptr(c, c->Hs, 1), c->stv, c->dW + c->oVg, D, R, L + 1, D, -1.f, c->st, c->dW + c->oVb);
  }
  return 0;
}

// clip + AdamW on dW (scaled by scale_by), transposes for the fused kernel, reconstruction with the
// updated weights and its clipped MSE (:484-494).  eta_w and the bias corrections come from
// c->hyp_dev so the sequence can sit in a captured graph.  No synchronisation.
int enqueue_adamw(cora_ctx* c, float scale_by, const float* hyp) {
  launch_zero(c->scal, 32, c->st);
  { KScope s(c, K_OTHER, 0, 4.0 * c->nW); launch_sumsq(c->dW, (long long)c->nW, scale_by, c->scal, c->st); }
  AdamP a; memset(&a, 0, sizeof a);
  a.w = c->W; a.g = c->dW; a.mu = c->amu; a.nu = c->anu; a.n = (long long)c->nW; a.sumsq = c->scal;
  a.gscale = scale_by; a.clip = c->cfg.clip_w; a.b1 = c->cfg.adam_b1; a.b2 = c->cfg.adam_b2;
  a.eps = c->cfg.adam_eps; a.wd = c->cfg.adam_wd; a.hyp = hyp;
  { KScope s(c, K_OTHER, 0, 28.0 * c->nW); launch_adamw(a, c->st); }
  refresh_wt(c);
  return 0;
}
// reconstruction under status "forward" with the current weights and its clipped MSE, :488-494
int enqueue_recon(cora_ctx* c) {
  int rc = chain_forward(c, false, c->scal + 1);
  if (rc) return rc;
  if (!c->fused) { KScope s(c, K_OTHER, 0, 8.0 * c->n0); launch_mse(c->z, c->xp, c->n0, c->scal + 1, c->st); }
  return 0;
}
int enqueue_apply(cora_ctx* c, float scale_by) {
  int rc = enqueue_adamw(c, scale_by, c->hyp_dev);
  return rc ? rc : enqueue_recon(c);
}

// host side of one AdamW step: stage {eta_w, 1 - b1^count, 1 - b2^count} for count = adam_count + 1
// (the caller advances adam_count once the launches that consume them are enqueued)
int stage_adam_hyp(cora_ctx* c, float eta_w) {
  const double cnt = (double)(c->adam_count + 1);
  float h[4] = {eta_w, (float)(1.0 - pow((double)c->cfg.adam_b1, cnt)),
                (float)(1.0 - pow((double)c->cfg.adam_b2, cnt)), 0.f};
  HCHECK(hipMemcpyAsync(c->hyp_dev, h, sizeof h, hipMemcpyHostToDevice, c->st));
  HCHECK(hipStreamSynchronize(c->st));   // h is a stack buffer
  return 0;
}

// Step scalars to the pinned host buffer: [energy row of step T-1 (8 slots per vode, or plain)] [w-energy row]
// [mse sum].  (A gather kernel writing through the device view of the pinned buffer instead of the three
// copies was measured: no difference, 835-842 k either way at cfg 2.)
static int enqueue_result_readback(cora_ctx* c, int T) {
  const int L = c->L;
  double* ho = c->host_out;
  const size_t nslot = (size_t)(L + 3) * 8;
  const double* erow = c->slotted() ? c->esum + (long long)(T - 1) * (L + 3) * 8 : c->energies + (long long)(T - 1) * (L + 3);
  const size_t nrow = c->slotted() ? nslot : (size_t)(L + 3);
  HCHECK(hipMemcpyAsync(ho, erow, sizeof(double) * nrow, hipMemcpyDeviceToHost, c->st));
  HCHECK(hipMemcpyAsync(ho + nslot, c->energies + (long long)c->Tmax * (L + 3), sizeof(double) * (L + 3), hipMemcpyDeviceToHost, c->st));
  HCHECK(hipMemcpyAsync(ho + nslot + (L + 3), c->scal + 1, sizeof(double), hipMemcpyDeviceToHost, c->st));
  return 0;
}

int cora_wgrad(cora_ctx* c, cora_step_out* out) {
  if (!c) return -1;
  HCHECK(hipSetDevice(c->device));
  const int L = c->L, slot = c->Tmax;
  int rc = 0;
  const bool graphed = c->use_graph && !c->prof && c->learn_warm;
  if (graphed && !c->wgrad_graph) {
    hipGraph_t graph = nullptr;
    HCHECK(hipStreamBeginCapture(c->st, hipStreamCaptureModeThreadLocal));
    rc = enqueue_wgrad(c);
    hipError_t e = hipStreamEndCapture(c->st, &graph);
    if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) FAIL(-2, std::string("hipStreamEndCapture failed: ") + hipGetErrorString(e));
    HCHECK(hipGraphInstantiate(&c->wgrad_graph, graph, nullptr, nullptr, 0));
    hipGraphDestroy(graph);
  }
  if (c->infer_pending) {
    // ends a cora_infer(out = NULL): both phases behind ONE synchronisation, scalars through pinned memory
    c->infer_pending = false;
    if (graphed) HCHECK(hipGraphLaunch(c->wgrad_graph, c->st));
    else if ((rc = enqueue_wgrad(c))) return rc;
    HCHECK(hipEventRecord(c->ev2, c->st));
    const int T = c->last_T;
    double* ho = c->host_out;
    const size_t nslot = (size_t)(L + 3) * 8;
    if (c->slotted()) HCHECK(hipMemcpyAsync(ho, c->esum + (long long)(T - 1) * (L + 3) * 8, sizeof(double) * nslot, hipMemcpyDeviceToHost, c->st));
    else HCHECK(hipMemcpyAsync(ho, c->energies + (long long)(T - 1) * (L + 3), sizeof(double) * (L + 3), hipMemcpyDeviceToHost, c->st));
    HCHECK(hipMemcpyAsync(ho + nslot, c->energies + (long long)slot * (L + 3), sizeof(double) * (L + 3), hipMemcpyDeviceToHost, c->st));
    HCHECK(hipStreamSynchronize(c->st));
    if (out) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, c->ev0, c->ev1); out->infer_ms = ms;
      hipEventElapsedTime(&ms, c->ev1, c->ev2); out->learn_ms = ms;
      double eh = 0.0, ew = 0.0;
      for (size_t k = 0; k < (c->slotted() ? nslot : (size_t)(L + 3)); ++k) eh += ho[k];
      for (int k = 0; k < L + 3; ++k) ew += ho[nslot + k];
      out->final_h_energy = eh; out->final_w_energy = ew;
    }
    return 0;
  }
  HCHECK(hipEventRecord(c->ev0, c->st));
  if (graphed) HCHECK(hipGraphLaunch(c->wgrad_graph, c->st));
  else if ((rc = enqueue_wgrad(c))) return rc;
  HCHECK(hipEventRecord(c->ev1, c->st));
  HCHECK(hipStreamSynchronize(c->st));
  if (out) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, c->ev0, c->ev1);
    out->learn_ms = ms;
    std::vector<double> en((size_t)L + 3);
    HCHECK(hipMemcpy(en.data(), c->energies + (long long)slot * (L + 3), sizeof(double) * en.size(), hipMemcpyDeviceToHost));
    double tot = 0.0;
    for (double v : en) tot += v;
    out->final_w_energy = tot;
  }
  return 0;
}

int cora_apply_wgrad(cora_ctx* c, float eta_w, float scale_by, cora_step_out* out) {
  if (!c) return -1;
  HCHECK(hipSetDevice(c->device));
  int rc = stage_adam_hyp(c, eta_w);
  if (rc) return rc;
  const bool graphed = c->use_graph && !c->prof && c->learn_warm;
  if (graphed && (!c->apply_graph || c->apply_graph_scale != scale_by)) {
    if (c->apply_graph) { hipGraphExecDestroy(c->apply_graph); c->apply_graph = nullptr; }
    hipGraph_t graph = nullptr;
    HCHECK(hipStreamBeginCapture(c->st, hipStreamCaptureModeThreadLocal));
    rc = enqueue_apply(c, scale_by);
    hipError_t e = hipStreamEndCapture(c->st, &graph);
    if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) FAIL(-2, std::string("hipStreamEndCapture failed: ") + hipGetErrorString(e));
    HCHECK(hipGraphInstantiate(&c->apply_graph, graph, nullptr, nullptr, 0));
    hipGraphDestroy(graph);
    c->apply_graph_scale = scale_by;
  }
  HCHECK(hipEventRecord(c->ev0, c->st));
  if (graphed) HCHECK(hipGraphLaunch(c->apply_graph, c->st));
  else if ((rc = enqueue_apply(c, scale_by))) return rc;
  c->adam_count += 1;
  HCHECK(hipEventRecord(c->ev1, c->st));
  HCHECK(hipStreamSynchronize(c->st));
  c->learn_warm = true;
  if (out) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, c->ev0, c->ev1);
    out->learn_ms += ms;
    double s = 0.0;
    HCHECK(hipMemcpy(&s, c->scal + 1, sizeof(double), hipMemcpyDeviceToHost));
    out->train_mse = s / (double)c->n0;
  }
  return 0;
}

// instantiate (once per T) the captured inference loop
int infer_graph(cora_ctx* c, int T, hipGraphExec_t* out) {
  auto it = c->graphs.find(T);
  if (it == c->graphs.end()) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    HCHECK(hipStreamBeginCapture(c->st, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_infer(c, T);
    hipError_t e = hipStreamEndCapture(c->st, &graph);
    if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) FAIL(-2, std::string("hipStreamEndCapture failed: ") + hipGetErrorString(e));
    HCHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipGraphDestroy(graph);
    c->graphs[T] = exec;
    it = c->graphs.find(T);
  }
  *out = it->second;
  return 0;
}

int cora_step(cora_ctx* c, const float* x, int x_on_device, int T, const float* eta_h, float eta_w, cora_step_out* out) {
  if (!c) return -1;
  if (!c->use_graph || c->prof || !c->learn_warm) {   // eager: the three public phases back to back
    // (cora_apply_wgrad sets learn_warm once every learning-phase kernel has run outside stream capture)
    cora_step_out tmp; memset(&tmp, 0, sizeof tmp);
    int rc = cora_infer(c, x, x_on_device, T, eta_h, &tmp);
    if (rc) return rc;
    rc = cora_wgrad(c, &tmp);
    if (rc) return rc;
    rc = cora_apply_wgrad(c, eta_w, 1.0f / (float)c->B, &tmp);
    if (rc) return rc;
    if (out) *out = tmp;
    return 0;
  }
  // Two graph launches and ONE host synchronisation per batch: the inference loop (per-T graph) and
  // the learning phase (weight gradients, clip + AdamW, transposes, reconstruction).
  if (T < 1 || T > c->Tmax) FAIL(-1, "T out of range (1..max_T)");
  if (!eta_h) FAIL(-1, "eta_h is NULL");
  HCHECK(hipSetDevice(c->device));
  if (int rc_ = settle_pending(c)) return rc_;   // an abandoned cora_infer(out = NULL) still reads the pinned staging
  const int L = c->L;
  int rc = upload_x(c, x, x_on_device);   // host x: staged synchronously by the runtime; device x: valid until we return
  if (rc) return rc;
  // eta_h and the AdamW scalars go through pinned memory the ctx owns: asynchronous, and the call
  // ends with a stream synchronisation before that memory is written again
  const double cnt = (double)(c->adam_count + 1);
  memcpy(c->host_in, eta_h, sizeof(float) * T);
  float* hy = c->host_in + c->Tmax;
  hy[0] = eta_w;
  hy[1] = (float)(1.0 - pow((double)c->cfg.adam_b1, cnt));
  hy[2] = (float)(1.0 - pow((double)c->cfg.adam_b2, cnt));
  hy[3] = 0.f;
  HCHECK(hipMemcpyAsync(c->eta_dev, c->host_in, sizeof(float) * T, hipMemcpyHostToDevice, c->st));
  HCHECK(hipMemcpyAsync(c->hyp_dev, hy, sizeof(float) * 4, hipMemcpyHostToDevice, c->st));
  c->last_T = T;
  hipGraphExec_t gi = nullptr;
  if ((rc = infer_graph(c, T, &gi))) return rc;
  if (!c->learn_graph) {
    hipGraph_t graph = nullptr;
    HCHECK(hipStreamBeginCapture(c->st, hipStreamCaptureModeThreadLocal));
    rc = enqueue_wgrad(c);
    if (!rc) rc = enqueue_apply(c, 1.0f / (float)c->B);
    hipError_t e = hipStreamEndCapture(c->st, &graph);
    if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) FAIL(-2, std::string("hipStreamEndCapture failed: ") + hipGetErrorString(e));
    HCHECK(hipGraphInstantiate(&c->learn_graph, graph, nullptr, nullptr, 0));
    hipGraphDestroy(graph);
  }
  HCHECK(hipEventRecord(c->ev0, c->st));
  HCHECK(hipGraphLaunch(gi, c->st));
  HCHECK(hipEventRecord(c->ev1, c->st));
  HCHECK(hipGraphLaunch(c->learn_graph, c->st));
  c->adam_count += 1;
  HCHECK(hipEventRecord(c->ev2, c->st));
  // scalars back through pinned memory: [energy row of step T-1 (slots or plain)] [w-energy row] [mse sum]
  double* ho = c->host_out;
  const size_t nslot = (size_t)(L + 3) * 8;
  if (int rcr = enqueue_result_readback(c, T)) return rcr;
  HCHECK(hipStreamSynchronize(c->st));
  if (out) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, c->ev0, c->ev1); out->infer_ms = ms;
    hipEventElapsedTime(&ms, c->ev1, c->ev2); out->learn_ms = ms;
    double eh = 0.0, ew = 0.0;
    for (size_t k = 0; k < (c->slotted() ? nslot : (size_t)(L + 3)); ++k) eh += ho[k];
    for (int k = 0; k < L + 3; ++k) ew += ho[nslot + k];
    out->final_h_energy = eh; out->final_w_energy = ew;
    out->train_mse = ho[nslot + (L + 3)] / (double)c->n0;
  }
  return 0;
}

// ---- exact data-parallel clip: the inference loop driven step by step from the host ---------------
// (see include/cora_hip.h: cora_infer_begin / _grad / _sumsq / _update / _end)
int cora_infer_begin(cora_ctx* c, const float* x, int x_on_device, int T, const float* eta_h) {
  if (!c) return -1;
  if (T < 1 || T > c->Tmax) FAIL(-1, "T out of range (1..max_T)");
  if (!eta_h) FAIL(-1, "eta_h is NULL");
  HCHECK(hipSetDevice(c->device));
  if (int rc_ = settle_pending(c)) return rc_;
  int rc = upload_x(c, x, x_on_device);
  if (rc) return rc;
  HCHECK(hipMemcpyAsync(c->eta_dev, eta_h, sizeof(float) * T, hipMemcpyHostToDevice, c->st));
  HCHECK(hipStreamSynchronize(c->st));
  c->last_T = T;
  HCHECK(hipEventRecord(c->ev0, c->st));
  zero_infer_logs(c, T);
  c->lazy_T = T;   // until cora_infer_end (or any other entry point: settle_pending)
  return 0;
}
int cora_infer_grad(cora_ctx* c, int t) {
  if (!c) return -1;
  if (t < 0 || t >= c->last_T) FAIL(-1, "step index out of range");
  HCHECK(hipSetDevice(c->device));
  return infer_grad_step(c, t);
}
int cora_infer_sumsq(cora_ctx* c, int t, double* host, size_t n, int set) {
  if (!c) return -1;
  if (t < 0 || t >= c->last_T) FAIL(-1, "step index out of range");
  if (!host || n != (size_t)(c->L + 2) * 8) FAIL(-1, "sumsq row has (L+2)*8 doubles");
  HCHECK(hipSetDevice(c->device));
  double* row = c->sumsq + (long long)t * (c->L + 2) * 8;
  if (set) HCHECK(hipMemcpyAsync(row, host, sizeof(double) * n, hipMemcpyHostToDevice, c->st));
  else HCHECK(hipMemcpyAsync(host, row, sizeof(double) * n, hipMemcpyDeviceToHost, c->st));
  HCHECK(hipStreamSynchronize(c->st));
  return 0;
}
int cora_infer_update(cora_ctx* c, int t) {
  if (!c) return -1;
  if (t < 0 || t >= c->last_T) FAIL(-1, "step index out of range");
  HCHECK(hipSetDevice(c->device));
  return infer_update_step(c, t);
}
int cora_infer_end(cora_ctx* c, cora_step_out* out) {
  if (!c) return -1;
  HCHECK(hipSetDevice(c->device));
  if (int rc_ = flush_lazy_update(c)) return rc_;   // a loop cut short of its last step
  c->lazy_T = 0;
  HCHECK(hipEventRecord(c->ev1, c->st));
  HCHECK(hipStreamSynchronize(c->st));
  if (out && c->last_T > 0) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, c->ev0, c->ev1);
    out->infer_ms = ms;
    std::vector<double> en((size_t)c->L + 3);
    if (energy_row(c, c->last_T - 1, en.data())) FAIL(-2, "energy read-back failed");
    double tot = 0.0;
    for (double v : en) tot += v;
    out->final_h_energy = tot;
  }
  return 0;
}

// MODE 1 of train_on_batch (update_weights_every_inference_step, :418-439): every iteration is one
// vode step followed by one weight step at the updated vodes; see include/cora_hip.h
// dist: the data-parallel form (cora_step_ipc_dist) -- exact != 0: the sums-of-squares row of every vode step summed over
// the ranks between its two halves; dE/dW summed over the ranks before every weight step, scale 1 / (B * world)
static int step_ipc_impl(cora_ctx* c, const float* x, int x_on_device, int T, const float* eta_h, const float* eta_w, cora_step_out* out,
                         bool dist, bool exact);
int cora_step_ipc(cora_ctx* c, const float* x, int x_on_device, int T, const float* eta_h, const float* eta_w, cora_step_out* out) {
  return step_ipc_impl(c, x, x_on_device, T, eta_h, eta_w, out, false, false);
}
#define NCHECK(call)                                                                                   \
  do {                                                                                                 \
    ncclResult_t r_ = (call);                                                                          \
    if (r_ != ncclSuccess) {                                                                           \
      c->err = std::string(#call " failed: ") + g_rccl.GetErrorString(r_);                             \
      return -5;                                                                                       \
    }                                                                                                  \
  } while (0)
static int step_ipc_impl(cora_ctx* c, const float* x, int x_on_device, int T, const float* eta_h, const float* eta_w, cora_step_out* out,
                         bool dist, bool exact) {
  if (!c) return -1;
  if (dist && !c->comm) FAIL(-1, "cora_step_ipc_dist: call cora_comm_init first");
  if (T < 1 || T > c->Tmax) FAIL(-1, "T out of range (1..max_T)");
  if (!eta_h || !eta_w) FAIL(-1, "eta_h / eta_w is NULL");
  HCHECK(hipSetDevice(c->device));
  if (int rc_ = settle_pending(c)) return rc_;
  const int L = c->L;
  int rc = upload_x(c, x, x_on_device);
  if (rc) return rc;
  std::vector<float> hy((size_t)4 * T);
  for (int t = 0; t < T; ++t) {
    const double cnt = (double)(c->adam_count + t + 1);
    hy[4 * t] = eta_w[t];
    hy[4 * t + 1] = (float)(1.0 - pow((double)c->cfg.adam_b1, cnt));
    hy[4 * t + 2] = (float)(1.0 - pow((double)c->cfg.adam_b2, cnt));
    hy[4 * t + 3] = 0.f;
  }
  HCHECK(hipMemcpyAsync(c->eta_dev, eta_h, sizeof(float) * T, hipMemcpyHostToDevice, c->st));
  HCHECK(hipMemcpyAsync(c->hyp_dev + 4, hy.data(), sizeof(float) * hy.size(), hipMemcpyHostToDevice, c->st));
  HCHECK(hipStreamSynchronize(c->st));
  c->last_T = T;
  HCHECK(hipEventRecord(c->ev0, c->st));
  zero_infer_logs(c, T);
  const float scale_by = 1.0f / ((float)c->B * (dist ? (float)c->comm_world : 1.0f));
  const size_t nrow = (size_t)(L + 2) * 8;
  for (int t = 0; t < T && !rc; ++t) {
    rc = infer_grad_step(c, t);                                                // :421-424
    if (!rc && dist && exact) {
      double* row = c->sumsq + (long long)t * nrow;
      NCHECK(g_rccl.AllReduce(row, row, nrow, ncclDouble, ncclSum, c->comm, c->st));
    }
    if (!rc) rc = infer_update_step(c, t);
    if (!rc) rc = enqueue_wgrad(c);                                            // :426-427
    if (!rc && dist) NCHECK(g_rccl.AllReduce(c->dW, c->dW, c->nW, ncclFloat, ncclSum, c->comm, c->st));
    if (!rc) rc = enqueue_adamw(c, scale_by, c->hyp_dev + 4 + 4 * t);          // :428
    if (!rc) c->adam_count += 1;
  }
  if (rc) return rc;
  HCHECK(hipEventRecord(c->ev1, c->st));
  if ((rc = enqueue_recon(c))) return rc;
  HCHECK(hipEventRecord(c->ev2, c->st));
  HCHECK(hipStreamSynchronize(c->st));
  c->learn_warm = true;
  if (out) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, c->ev0, c->ev1); out->infer_ms = ms;
    hipEventElapsedTime(&ms, c->ev1, c->ev2); out->learn_ms = ms;
    std::vector<double> en((size_t)L + 3);
    if (energy_row(c, T - 1, en.data())) FAIL(-2, "energy read-back failed");
    double tot = 0.0;
    for (double v : en) tot += v;
    out->final_h_energy = tot;
    HCHECK(hipMemcpy(en.data(), c->energies + (long long)c->Tmax * (L + 3), sizeof(double) * en.size(), hipMemcpyDeviceToHost));
    tot = 0.0;
    for (double v : en) tot += v;
    out->final_w_energy = tot;
    double sse = 0.0;
    HCHECK(hipMemcpy(&sse, c->scal + 1, sizeof(double), hipMemcpyDeviceToHost));
    out->train_mse = sse / (double)c->n0;
  }
  return 0;
}

// ---- data parallelism with RCCL bound inside the library ---------------------------------------------
// (see include/cora_hip.h: cora_comm_unique_id / cora_comm_init / cora_step_dist)
int cora_comm_unique_id(void* id, size_t nbytes) {
  if (!id || nbytes != sizeof(ncclUniqueId)) { g_create_err = "unique id buffer must be 128 bytes"; return -1; }
  if (const char* e = rccl_load()) { g_create_err = e; return -5; }
  ncclUniqueId u;
  ncclResult_t r = g_rccl.GetUniqueId(&u);
  if (r != ncclSuccess) { g_create_err = std::string("ncclGetUniqueId failed: ") + g_rccl.GetErrorString(r); return -5; }
  memcpy(id, &u, sizeof u);
  return 0;
}

int cora_comm_init(cora_ctx* c, const void* id, size_t nbytes, int rank, int world) {
  if (!c) return -1;
  if (!id || nbytes != sizeof(ncclUniqueId)) FAIL(-1, "unique id buffer must be 128 bytes");
  if (world < 1 || rank < 0 || rank >= world) FAIL(-1, "rank / world out of range");
  if (c->comm) FAIL(-1, "communicator already initialised");
  if (const char* e = rccl_load()) FAIL(-5, e);
  HCHECK(hipSetDevice(c->device));
  if (int rc_ = settle_pending(c)) return rc_;
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  NCHECK(g_rccl.CommInitRank(&c->comm, world, u, rank));
  c->comm_rank = rank; c->comm_world = world;
  c->dist_warm = false;
  return 0;
}

int cora_comm_info(const cora_ctx* c, int* rank, int* world) {
  if (!c) return -1;
  if (rank) *rank = c->comm ? c->comm_rank : -1;
  if (world) *world = c->comm ? c->comm_world : 0;
  return 0;
}

// T inference steps; with exact_clip the per-vode sums of squares of step t (row t of c->sumsq, (L+2)*8
// doubles) are summed over the ranks between the gradient half and the update half, so
// optax.clip_by_global_norm (and normalize_vode_h_gradients) see the norm of the WHOLE batch
static int enqueue_infer_dist(cora_ctx* c, int T, bool exact) {
  zero_infer_logs(c, T);
  const size_t nrow = (size_t)(c->L + 2) * 8;
  c->lazy_T = T;   // a deferred update reads row t after its all-reduce: same stream, launched later
  for (int t = 0; t < T; ++t) {
    int rc = infer_grad_step(c, t);
    if (rc) { c->lazy_T = 0; return rc; }
    if (exact) {
      double* row = c->sumsq + (long long)t * nrow;
      ncclResult_t r_ = g_rccl.AllReduce(row, row, nrow, ncclDouble, ncclSum, c->comm, c->st);
      if (r_ != ncclSuccess) { c->lazy_T = 0; c->err = std::string("ncclAllReduce (sums of squares) failed: ") + g_rccl.GetErrorString(r_); return -5; }
    }
    if ((rc = infer_update_step(c, t))) { c->lazy_T = 0; return rc; }
  }
  c->lazy_T = 0;
  return 0;
}
// dE/dW of this shard, SUM over ranks, then the identical clip + AdamW step (scale 1 / (B_local * world),
// /root/reference/src/decoder_transformer.py:484) and the reconstruction on every rank
static int enqueue_learn_dist(cora_ctx* c) {
  int rc = enqueue_wgrad(c);
  if (rc) return rc;
  NCHECK(g_rccl.AllReduce(c->dW, c->dW, c->nW, ncclFloat, ncclSum, c->comm, c->st));
  return enqueue_apply(c, 1.0f / ((float)c->B * (float)c->comm_world));
}

// Replay graphs on this call?  Not before the first (eager) call has run every kernel and collective outside
// stream capture, not under the profiler, not after a refused capture or cora_dist_set_eager.
static bool dist_graphs_wanted(cora_ctx* c) {
  static const bool no_dist_graph = getenv("CORA_DIST_NO_GRAPH") != nullptr;   // operator switch: eager collectives
  return c->use_graph && !c->prof && c->dist_warm && !c->dist_graph_broken && !no_dist_graph;
}
// Build the two graphs of the distributed step if they are missing.  Capturing only records: if RCCL (or HIP)
// refuses a collective under stream capture on this system, nothing has run, and the step falls back -- for
// good -- to the eager launch sequence of the first call, which is known to work by then.  The eager sequence
// issues the same collectives in the same order, so ranks that disagree about graph vs eager still match.
static bool dist_build_graphs(cora_ctx* c, int T, bool exact) {
  auto capture = [&](auto&& enqueue, hipGraphExec_t* exec) -> bool {
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(c->st, hipStreamCaptureModeRelaxed) != hipSuccess) return false;
    const int rce = enqueue();
    const hipError_t e = hipStreamEndCapture(c->st, &graph);
    bool ok = (rce == 0 && e == hipSuccess && graph != nullptr);
    if (ok) ok = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph) hipGraphDestroy(graph);
    if (!ok) (void)hipGetLastError();
    return ok;
  };
  const int key = 2 * T + (exact ? 1 : 0);
  bool ok = true;
  if (c->dist_graphs.find(key) == c->dist_graphs.end()) {
    hipGraphExec_t exec = nullptr;
    if (capture([&]() { return enqueue_infer_dist(c, T, exact); }, &exec)) c->dist_graphs[key] = exec;
    else ok = false;
  }
  if (ok && !c->dist_learn_graph && !capture([&]() { return enqueue_learn_dist(c); }, &c->dist_learn_graph)) {
    c->dist_learn_graph = nullptr;
    ok = false;
  }
  if (!ok) c->dist_graph_broken = true;
  return ok;
}

// Capture (not launch) the graphs cora_step_dist(T, exact_clip) would replay: 0 = ready, 1 = this system refuses
// a collective under capture (the ctx stays on eager launches), < 0 error.  Lets the host agree on graph vs eager
// across ranks BEFORE the first replay (cora_amd/dist.py: DataParallel.agree_on_graphs).
int cora_dist_prepare(cora_ctx* c, int T, int exact_clip) {
  if (!c) return -1;
  if (!c->comm) FAIL(-1, "cora_dist_prepare: call cora_comm_init first");
  if (T < 1 || T > c->Tmax) FAIL(-1, "T out of range (1..max_T)");
  if (!c->dist_warm) FAIL(-1, "cora_dist_prepare: run one cora_step_dist first (RCCL sets its channels up on first use)");
  HCHECK(hipSetDevice(c->device));
  if (int rc_ = settle_pending(c)) return rc_;
  if (!dist_graphs_wanted(c)) return 1;
  return dist_build_graphs(c, T, exact_clip != 0) ? 0 : 1;
}
int cora_dist_set_eager(cora_ctx* c, int eager) {
  if (!c) return -1;
  c->dist_graph_broken = eager != 0;
  return 0;
}
// How the LAST cora_step_dist ran: 0 = none yet / no communicator, 1 = eager launches (first call, CORA_NO_GRAPH,
// refused capture or cora_dist_set_eager), 2 = two graph replays with the collectives as graph nodes.
int cora_dist_mode(const cora_ctx* c) { return c ? c->dist_last_mode : -1; }

int cora_step_dist(cora_ctx* c, const float* x, int x_on_device, int T, const float* eta_h, float eta_w,
                   int exact_clip, cora_step_out* out) {
  if (!c) return -1;
  if (!c->comm) FAIL(-1, "cora_step_dist: call cora_comm_init first");
  if (T < 1 || T > c->Tmax) FAIL(-1, "T out of range (1..max_T)");
  if (!eta_h) FAIL(-1, "eta_h is NULL");
  HCHECK(hipSetDevice(c->device));
  if (int rc_ = settle_pending(c)) return rc_;
  const int L = c->L;
  const bool exact = exact_clip != 0;
  int rc = upload_x(c, x, x_on_device);
  if (rc) return rc;
  const double cnt = (double)(c->adam_count + 1);
  memcpy(c->host_in, eta_h, sizeof(float) * T);
  float* hy = c->host_in + c->Tmax;
  hy[0] = eta_w;
  hy[1] = (float)(1.0 - pow((double)c->cfg.adam_b1, cnt));
  hy[2] = (float)(1.0 - pow((double)c->cfg.adam_b2, cnt));
  hy[3] = 0.f;
  HCHECK(hipMemcpyAsync(c->eta_dev, c->host_in, sizeof(float) * T, hipMemcpyHostToDevice, c->st));
  HCHECK(hipMemcpyAsync(c->hyp_dev, hy, sizeof(float) * 4, hipMemcpyHostToDevice, c->st));
  c->last_T = T;
  bool graphed = dist_graphs_wanted(c);
  hipGraphExec_t infer_exec = nullptr;
  if (graphed) {
    if (dist_build_graphs(c, T, exact)) infer_exec = c->dist_graphs[2 * T + (exact ? 1 : 0)];
    else graphed = false;
  }
  if (!graphed) {
    // first call (and CORA_NO_GRAPH, and after a refused capture): eager launches -- every kernel and
    // every collective runs outside stream capture (RCCL sets up its channels and buffers on first use)
    HCHECK(hipEventRecord(c->ev0, c->st));
    if ((rc = enqueue_infer_dist(c, T, exact))) return rc;
    HCHECK(hipEventRecord(c->ev1, c->st));
    if ((rc = enqueue_learn_dist(c))) return rc;
  } else {
    HCHECK(hipEventRecord(c->ev0, c->st));
    HCHECK(hipGraphLaunch(infer_exec, c->st));
    HCHECK(hipEventRecord(c->ev1, c->st));
    HCHECK(hipGraphLaunch(c->dist_learn_graph, c->st));
  }
  c->dist_last_mode = graphed ? 2 : 1;
  c->adam_count += 1;
  HCHECK(hipEventRecord(c->ev2, c->st));
  double* ho = c->host_out;
  const size_t nslot = (size_t)(L + 3) * 8;
  if (int rcr = enqueue_result_readback(c, T)) return rcr;
  HCHECK(hipStreamSynchronize(c->st));
  c->dist_warm = true;
  c->learn_warm = true;
  if (out) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, c->ev0, c->ev1); out->infer_ms = ms;
    hipEventElapsedTime(&ms, c->ev1, c->ev2); out->learn_ms = ms;
    double eh = 0.0, ew = 0.0;
    for (size_t k = 0; k < (c->slotted() ? nslot : (size_t)(L + 3)); ++k) eh += ho[k];
    for (int k = 0; k < L + 3; ++k) ew += ho[nslot + k];
    out->final_h_energy = eh; out->final_w_energy = ew;
    out->train_mse = ho[nslot + (L + 3)] / (double)c->n0;
  }
  return 0;
}

int cora_step_ipc_dist(cora_ctx* c, const float* x, int x_on_device, int T, const float* eta_h, const float* eta_w,
                       int exact_clip, cora_step_out* out) {
  return step_ipc_impl(c, x, x_on_device, T, eta_h, eta_w, out, true, exact_clip != 0);
}

int cora_wgrad_buffer(cora_ctx* c, void** dev, size_t* n) {
  if (!c || !dev || !n) return -1;
  *dev = c->dW; *n = c->nW;
  return 0;
}
int cora_bind_wgrad_buffer(cora_ctx* c, void* dev, size_t n) {
  if (!c) return -1;
  if (!dev || n != c->nW) FAIL(-1, "bind_wgrad_buffer: size mismatch");
  if (((uintptr_t)dev) % 16) FAIL(-1, "bind_wgrad_buffer: pointer must be 16-byte aligned");
  drop_learn_graphs(c);
  c->dW = (float*)dev;   // the old buffer stays in c->allocs and is freed with the ctx
  c->dW_owned = false;
  return 0;
}
int cora_get_wgrad(cora_ctx* c, float* host, size_t n) { if (!c) return -1; return copy_out(c, host, c->dW, n, c->nW); }

int cora_get_energies(cora_ctx* c, int T, double* total, double* per_layer) {
  if (!c) return -1;
  if (T < 1 || T > c->Tmax) FAIL(-1, "T out of range");
  HCHECK(hipSetDevice(c->device));
  if (int rc_ = settle_pending(c)) return rc_;
  const int W = c->L + 3;
  std::vector<double> en((size_t)T * W);
  HCHECK(hipStreamSynchronize(c->st));
  for (int t = 0; t < T; ++t)
    if (energy_row(c, t, en.data() + (size_t)t * W)) FAIL(-2, "energy read-back failed");
  for (int t = 0; t < T; ++t) {
    double s = 0.0;
    for (int l = 0; l < W; ++l) s += en[(size_t)t * W + l];
    if (total) total[t] = s;
  }
  if (per_layer) memcpy(per_layer, en.data(), sizeof(double) * en.size());
  return 0;
}

static int download_image(cora_ctx* c, const float* zpatch, float* host, size_t n) {
  if (n != (size_t)c->nImg) FAIL(-1, "image size mismatch");
  launch_patchify(c->ximg, const_cast<float*>(zpatch), c->B, c->cfg.F, c->cfg.C, c->cfg.Himg, c->cfg.Wimg, c->cfg.patch, 1, c->st);
  HCHECK(hipMemcpyAsync(host, c->ximg, sizeof(float) * n, hipMemcpyDeviceToHost, c->st));
  HCHECK(hipStreamSynchronize(c->st));
  return 0;
}

int cora_forward(cora_ctx* c, float* xhat, size_t n) {
  if (!c) return -1;
  if (!xhat) FAIL(-1, "null output");
  HCHECK(hipSetDevice(c->device));
  if (int rc_ = settle_pending(c)) return rc_;
  int rc = chain_forward(c, false);
  if (rc) return rc;
  return download_image(c, c->z, xhat, n);
}

size_t cora_forward_static_floats(const cora_config* cfg) {
  if (!cfg) return 0;
  const size_t R = (size_t)cfg->B * cfg->N, D = cfg->D, M = cfg->M, P = cfg->P, H = cfg->H, L = cfg->L;
  auto up = [](size_t n) { return (n + 63) / 64 * 64; };     // every region starts on a 256-byte boundary
  size_t n = up(R * D) * 6 /* xn1 att y xn2 ua ub */ + up(R * 3 * D) + up(R * M) + up(R * P) + 2 * up(R * 2) + up(R * H);
  if (cfg->use_vode_ln) n += up((L + 1) * R * D) + up(2 * (L + 1) * R);
  return n;
}

int cora_forward_static(const cora_config* cfg, int device, const float* W_dev, size_t n_weights, const float* pos_dev,
                        const float* h0_dev, float* xhat_dev, void* hip_stream) {
  auto bad = [&](int rc, const std::string& m) { g_create_err = "cora_forward_static: " + m; return rc; };
  if (!cfg || !W_dev || !pos_dev || !h0_dev || !xhat_dev) return bad(-1, "null argument");
  if (const char* e = check_config(cfg)) return bad(-1, e);
  const size_t need = cora_forward_static_floats(cfg);
  if (need > (size_t)CORA_STATIC_SCRATCH_FLOATS) {
    char buf[256];
    snprintf(buf, sizeof buf, "needs %zu floats of scratch, the static buffer holds %u (16 MiB): lower B (rows = B*N = %lld)",
             need, (unsigned)CORA_STATIC_SCRATCH_FLOATS, (long long)cfg->B * cfg->N);
    return bad(-6, buf);
  }
  if (hipSetDevice(device) != hipSuccess) return bad(-2, "hipSetDevice failed");
  float* base = nullptr;
  std::string err;
  if (int rc = static_scratch_ptr(device, &base, &err)) return bad(rc, err);
  // A host-side view of the problem: the same launch code as cora_forward, every pointer either the
  // caller's or carved from this GPU's scratch.  Nothing is allocated, copied or synchronised.
  cora_ctx v;
  layout_ctx(&v, cfg, device);
  if (n_weights != v.nW) return bad(-1, "weight arena size mismatch (see cora_weights_size)");
  v.st = (hipStream_t)hip_stream;
  v.W = const_cast<float*>(W_dev); v.pos = const_cast<float*>(pos_dev); v.Hs = const_cast<float*>(h0_dev);
  v.fused = false; v.prof = false; v.Gmax = 1;
  const size_t R = (size_t)v.BN, D = v.D;
  size_t o = 0;
  auto take = [&](size_t n) { float* q = base + o; o += (n + 63) / 64 * 64; return q; };
  v.xn1 = take(R * D); v.att = take(R * D); v.y = take(R * D); v.xn2 = take(R * D); v.ua = take(R * D); v.ub = take(R * D);
  v.qkv = take(R * 3 * D); v.a = take(R * v.M); v.z = take(R * v.P); v.st1 = take(R * 2); v.st2 = take(R * 2); v.lse = take(R * v.H);
  if (v.vln) { v.hln = take((size_t)(v.L + 1) * R * D); v.stv = take(2 * (size_t)(v.L + 1) * R); }
  int rc = chain_forward(&v, false);
  if (rc) return bad(rc, v.err);
  launch_patchify(xhat_dev, v.z, v.B, cfg->F, cfg->C, cfg->Himg, cfg->Wimg, cfg->patch, 1, v.st);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return bad(-2, std::string("kernel launch failed: ") + hipGetErrorString(e));
  return 0;
}

int cora_energy(cora_ctx* c, const float* x_host, double* E, double* per_layer, float* xhat, size_t n) {
  if (!c) return -1;
  HCHECK(hipSetDevice(c->device));
  if (int rc_ = settle_pending(c)) return rc_;
  const int L = c->L, slot = c->Tmax + 1;
  if (x_host) { int rc = upload_x(c, x_host, 0); if (rc) return rc; }
  HCHECK(hipMemsetAsync(c->energies + (long long)slot * (L + 3), 0, sizeof(double) * (L + 3), c->st));
  int rc = errors_pass(c, slot);
  if (rc) return rc;
  std::vector<double> en((size_t)L + 3);
  HCHECK(hipStreamSynchronize(c->st));
  HCHECK(hipMemcpy(en.data(), c->energies + (long long)slot * (L + 3), sizeof(double) * en.size(), hipMemcpyDeviceToHost));
  double tot = 0.0;
  for (double v : en) tot += v;
  if (E) *E = tot;
  if (per_layer) memcpy(per_layer, en.data(), sizeof(double) * en.size());
  if (xhat) {
    out_proj(c, vode_in(c, L + 1), c->z, OUT_STORE, nullptr, nullptr);   // hln is current: errors_pass just ran
    return download_image(c, c->z, xhat, n);
  }
  return 0;
}

// `unmask_on_batch_enhanced` (src/decoder_transformer.py:833-1046): see include/cora_hip.h.
// eta_w != NULL: `update_weights_during_unmasking` (:967-979) -- after every vode step, dE/dW at the
// updated vodes and one clip + AdamW step with scale `scale_by`, before the clamp.
static int unmask_impl(cora_ctx* c, const float* x_true, const float* x_init, const float* mask, int use_corruption,
                       int status_init, int T, const float* eta_h, float alpha, float* recons, float* xhat,
                       float* sensory, double* grad_norms, const float* eta_w, float scale_by) {
  if (!c) return -1;
  if (T < 0 || T > c->Tmax) FAIL(-1, "T out of range (0..max_T)");
  if (!x_true || !x_init) FAIL(-1, "x_true / x_init is NULL");
  if (T > 0 && !eta_h) FAIL(-1, "eta_h is NULL");
  if (use_corruption && !mask) FAIL(-1, "use_corruption needs a mask");
  HCHECK(hipSetDevice(c->device));
  if (int rc_ = settle_pending(c)) return rc_;
  const int L = c->L;
  int rc = 0;
  if (use_corruption && !c->msens) {
    if ((rc = dalloc(c, &c->xt, c->n0)) || (rc = dalloc(c, &c->mk, c->n0)) || (rc = dalloc(c, &c->msens, c->n0))) return rc;
  }
  if (use_corruption) {   // ground truth and mask go to patch layout once (patchify is a permutation)
    HCHECK(hipMemcpyAsync(c->ximg, x_true, sizeof(float) * (size_t)c->nImg, hipMemcpyHostToDevice, c->st));
    launch_patchify(c->ximg, c->xt, c->B, c->cfg.F, c->cfg.C, c->cfg.Himg, c->cfg.Wimg, c->cfg.patch, 0, c->st);
    HCHECK(hipMemcpyAsync(c->ximg, mask, sizeof(float) * (size_t)c->nImg, hipMemcpyHostToDevice, c->st));
    launch_patchify(c->ximg, c->mk, c->B, c->cfg.F, c->cfg.C, c->cfg.Himg, c->cfg.Wimg, c->cfg.patch, 0, c->st);
    launch_zero(c->msens, sizeof(float) * (size_t)c->n0, c->st);
  }
  if ((rc = upload_x(c, x_init, 0))) return rc;             // forward(x_init): the sensory vode takes x_init, :900-901
  if (status_init && (rc = chain_forward(c, true))) return rc;   // STATUS.INIT: h_l := u_l chained from h_0
  if (T > 0) HCHECK(hipMemcpyAsync(c->eta_dev, eta_h, sizeof(float) * T, hipMemcpyHostToDevice, c->st));
  std::vector<float> hy;
  if (eta_w && T > 0) {
    hy.resize((size_t)4 * T);
    for (int t = 0; t < T; ++t) {
      const double cnt = (double)(c->adam_count + t + 1);
      hy[4 * t] = eta_w[t];
      hy[4 * t + 1] = (float)(1.0 - pow((double)c->cfg.adam_b1, cnt));
      hy[4 * t + 2] = (float)(1.0 - pow((double)c->cfg.adam_b2, cnt));
      hy[4 * t + 3] = 0.f;
    }
    HCHECK(hipMemcpyAsync(c->hyp_dev + 4, hy.data(), sizeof(float) * hy.size(), hipMemcpyHostToDevice, c->st));
  }
  HCHECK(hipStreamSynchronize(c->st));
  launch_zero(c->energies, ((sizeof(double) * (size_t)(T + 1) * (L + 3)) + 15) / 16 * 16, c->st);
  launch_zero(c->sumsq, sizeof(double) * (size_t)(T + 1) * (L + 2) * 8, c->st);
  launch_zero(c->esum, sizeof(double) * (size_t)(T + 1) * (L + 3) * 8, c->st);
  launch_zero(c->Ms, sizeof(float) * (size_t)c->nV, c->st);   // optim_h.init, :841
  c->sens_live = use_corruption != 0;                        // :904-911
  c->last_T = T;
  for (int t = 0; t < T && !rc; ++t) {
    if (recons) {                                            // reconstruction BEFORE the step, :916-919
      if ((rc = chain_forward(c, false))) break;
      if ((rc = download_image(c, c->z, recons + (size_t)t * c->nImg, (size_t)c->nImg))) break;
    }
    if ((rc = infer_step(c, t))) break;                      // :922-965
    if (eta_w) {                                             // :967-979
      if ((rc = enqueue_wgrad(c))) break;
      if ((rc = enqueue_adamw(c, scale_by, c->hyp_dev + 4 + 4 * t))) break;
      c->adam_count += 1;
    }
    if (use_corruption) launch_sens_clamp(c->xp, c->xt, c->mk, alpha, c->n0, c->st);   // :982-994
  }
  c->sens_live = false;                                      // re-frozen, :1037
  if (rc) return rc;
  if ((rc = chain_forward(c, false))) return rc;             // :1002-1004
  if (xhat && (rc = download_image(c, c->z, xhat, (size_t)c->nImg))) return rc;
  if (sensory && (rc = download_image(c, c->xp, sensory, (size_t)c->nImg))) return rc;
  HCHECK(hipStreamSynchronize(c->st));
  if (grad_norms) {   // [T, L+3]: L2 norm of the (scaled) gradient of every vode, the quantity logged at :936-958
    std::vector<double> s8((size_t)(L + 2) * 8), en((size_t)L + 3);
    for (int t = 0; t < T; ++t) {
      HCHECK(hipMemcpy(s8.data(), c->sumsq + (long long)t * (L + 2) * 8, sizeof(double) * s8.size(), hipMemcpyDeviceToHost));
      for (int l = 0; l < L + 2; ++l) {
        double n2 = 0.0;
        for (int k = 0; k < 8; ++k) n2 += s8[(size_t)l * 8 + k];
        grad_norms[(size_t)t * (L + 3) + l] = sqrt(n2);
      }
      double gs = 0.0;
      if (use_corruption) {
        if (energy_row(c, t, en.data())) FAIL(-2, "energy read-back failed");
        gs = sqrt(2.0 * en[L + 2]) * (double)c->sens_scale;
      }
      grad_norms[(size_t)t * (L + 3) + L + 2] = gs;
    }
  }
  return 0;
}

int cora_unmask(cora_ctx* c, const float* x_true, const float* x_init, const float* mask, int use_corruption,
                int status_init, int T, const float* eta_h, float alpha, float* recons, float* xhat,
                float* sensory, double* grad_norms) {
  return unmask_impl(c, x_true, x_init, mask, use_corruption, status_init, T, eta_h, alpha, recons, xhat, sensory,
                     grad_norms, nullptr, 0.f);
}
int cora_unmask_learn(cora_ctx* c, const float* x_true, const float* x_init, const float* mask, int use_corruption,
                      int status_init, int T, const float* eta_h, float alpha, const float* eta_w, float scale_by,
                      float* recons, float* xhat, float* sensory, double* grad_norms) {
  if (!c) return -1;
  if (!eta_w) FAIL(-1, "eta_w is NULL");
  return unmask_impl(c, x_true, x_init, mask, use_corruption, status_init, T, eta_h, alpha, recons, xhat, sensory,
                     grad_norms, eta_w, scale_by);
}

int cora_profile_step(cora_ctx* c, int cap, char* names, int* launches, double* ms, double* flops, double* bytes, int* n_out) {
  if (!c) return -1;
  if (cap < K_COUNT) FAIL(-1, "profile: capacity too small");
  HCHECK(hipSetDevice(c->device));
  if (int rc_ = settle_pending(c)) return rc_;
  // snapshot the vodes + momentum so profiling does not disturb the trajectory
  float *hsave = nullptr, *msave = nullptr;
  HCHECK(hipMalloc((void**)&hsave, sizeof(float) * (size_t)c->nV));
  HCHECK(hipMalloc((void**)&msave, sizeof(float) * (size_t)c->nV));
  HCHECK(hipMemcpyAsync(hsave, c->Hs, sizeof(float) * (size_t)c->nV, hipMemcpyDeviceToDevice, c->st));
  HCHECK(hipMemcpyAsync(msave, c->Ms, sizeof(float) * (size_t)c->nV, hipMemcpyDeviceToDevice, c->st));
  float eta0 = 0.f;
  HCHECK(hipStreamSynchronize(c->st));
  HCHECK(hipMemcpy(c->eta_dev, &eta0, sizeof(float), hipMemcpyHostToDevice));
  HCHECK(hipMemsetAsync(c->energies, 0, sizeof(double) * (c->L + 3), c->st));
  HCHECK(hipMemsetAsync(c->sumsq, 0, sizeof(double) * (c->L + 2) * 8, c->st));
  HCHECK(hipMemsetAsync(c->esum, 0, sizeof(double) * (c->L + 3) * 8, c->st));
  const int tp = 0;
  c->lazy_T = 2;       // profile the steady state of a T-step loop: where the path defers the vode update into the next
                       // step's launch (D = 64 fused), every profiled step carries the previous one's
  // warm: the set-up above (allocations, copies, synchronisations) leaves the GPU idle for tens of milliseconds and its
  // clocks take tens of milliseconds of load to come back (kernel trace, scripts/prof_trace.sh: the same launch runs
  // 208 us in the first batch after an idle period and 188 us from the third on) -- run steps for >= 60 ms first, so the
  // launches below are timed at the clocks the timed region of a benchmark runs at
  {
    const auto w0 = std::chrono::steady_clock::now();
    int rcw = 0;
    do {
      for (int i = 0; i < 32 && !rcw; ++i) rcw = infer_step(c, tp);
      HCHECK(hipStreamSynchronize(c->st));
    } while (!rcw && std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < 0.06);
    if (rcw) { c->lazy_T = 0; hipFree(hsave); hipFree(msave); return rcw; }
  }
  c->prof = true; c->recs.clear();
  c->prof_burst = 8;
  const int reps = 5;
  int rc = 0;
  for (int i = 0; i < reps && !rc; ++i) rc = infer_step(c, tp);
  c->prof = false; c->prof_burst = 1;
  if (!rc) rc = flush_lazy_update(c);
  c->lazy_T = 0;
  HCHECK(hipStreamSynchronize(c->st));
  for (int k = 0; k < K_COUNT; ++k) { launches[k] = 0; ms[k] = 0; flops[k] = 0; bytes[k] = 0; snprintf(names + 32 * k, 32, "%s", kKindNames[k]); }
  for (auto& r : c->recs) {
    float t = 0.f;
    hipEventElapsedTime(&t, r.e0, r.e1);
    t /= (float)r.n;   // a burst counts as ONE launch of its average duration: per-step sums stay per-step
    launches[r.kind] += 1; ms[r.kind] += t; flops[r.kind] += r.flops; bytes[r.kind] += r.bytes;
    hipEventDestroy(r.e0); hipEventDestroy(r.e1);
  }
  c->recs.clear();
  if (n_out) *n_out = K_COUNT;
  HCHECK(hipMemcpyAsync(c->Hs, hsave, sizeof(float) * (size_t)c->nV, hipMemcpyDeviceToDevice, c->st));
  HCHECK(hipMemcpyAsync(c->Ms, msave, sizeof(float) * (size_t)c->nV, hipMemcpyDeviceToDevice, c->st));
  HCHECK(hipStreamSynchronize(c->st));
  hipFree(hsave); hipFree(msave);
  return rc;
}

}  // extern "C"
