/* cora_hip.h -- C ABI of libcora_hip.so: the MI355X (gfx950) implementation of Cora's PC-ViT
 * predictive-coding step.
 *
 * The reference (/root/reference, pure Python on JAX) has no FFI / plugin interface for this path:
 * the step is reached by ordinary Python calls.  This ABI is therefore the
 * boundary a replacement exports and the host package `cora_amd` binds with ctypes; each entry
 * point cites the reference code it replaces.  All functions return 0 on success and a negative
 * code on error; cora_last_error() gives the message.  No exceptions cross the ABI, there is no
 * global state except the per-thread last-error string for failed cora_create calls.
 *
 * Ownership / threading: a cora_ctx owns all of its device memory, its HIP stream and its
 * hipGraphs.  Host buffers are caller-owned and copied synchronously.  One ctx per GPU; a ctx is
 * not thread-safe; distinct ctxs are independent.
 *
 * Layouts (all float32, row-major, batch-major):
 *   vode l = 0        : [B, N, P]      (h_0, the top latent in patch space)
 *   vode l = 1..L+1   : [B, N, D]
 *   images            : [B, C, H, W]  or  [B, F, C, H, W]   (sensory vode, frozen = the input x)
 *   weights           : one flat arena; cora_weight_offset() maps a pcx key string
 *                       (".transformer_blocks[3].linear1.nn.weight", ...) to (offset, count).
 */
#ifndef CORA_HIP_H
#define CORA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cora_ctx cora_ctx;

/* Static description of one model + step recipe.  Replaces TransformerConfig's derived dims
 * (src/decoder_transformer.py:100-125) and the optimizer recipe of
 * examples/debug_transformer_wandb.py:1116-1167. */
typedef struct cora_config {
  int32_t N;          /* num_patches (tokens)                                    */
  int32_t P;          /* patch_dim = p*p*C                                       */
  int32_t D;          /* hidden_size                                             */
  int32_t H;          /* num_heads (D % H == 0)                                  */
  int32_t M;          /* mlp_hidden_dim = int(D*mlp_ratio)                       */
  int32_t L;          /* num_blocks                                              */
  int32_t B;          /* batch size held by THIS ctx (per-GPU shard)             */
  int32_t C;          /* image channels                                          */
  int32_t patch;      /* patch_size p                                            */
  int32_t Himg;       /* image height                                            */
  int32_t Wimg;       /* image width                                             */
  int32_t F;          /* frames (1 for images)                                   */
  int32_t dtype_mode; /* 0 = fp32 MFMA (parity path); 1 = opt-in bf16 MFMA: GEMM and (sequences longer
                       * than 64 tokens) attention operands rounded to bf16 on their way into LDS,
                       * two-term split for the forward predictions, fp32 accumulate, all state fp32
                       * (per-op path only; bar 1e-2); 2 = opt-in "bf16x3": every K-inner GEMM takes
                       * both operands as three bf16 terms and keeps the six products above 2^-24
                       * (fp32-grade products from the bf16 pipe; per-op path, attention fp32; held
                       * to the fp32 bars).                                                     */
  int32_t use_scaling;   /* use_inference_lr_scaling (:449-453)                  */
  float   scale_base;    /* inference_lr_scale_base (:308-350)                   */
  float   l1;            /* intermediate_l1_coeff (pcx/predictive_coding/_energy.py:61-68) */
  float   l2;            /* intermediate_l2_coeff                                */
  float   clip_h;        /* h_grad_clip_norm, <=0 disables (optax.clip_by_global_norm) */
  float   clip_w;        /* w_grad_clip_norm, <=0 disables                       */
  float   momentum;      /* hidden_momentum (optax.sgd trace decay)              */
  float   adam_b1, adam_b2, adam_eps, adam_wd;   /* optax.adamw                  */
  int32_t use_grad_norm;    /* use_vode_grad_norm (:353-389)                     */
  float   grad_norm_target; /* vode_grad_norm_target                             */
  int32_t max_T;         /* largest T any call will use (sizes the energy log)   */
  int32_t use_vode_ln;   /* use_vode_state_layernorm (:212-218, 232-241): what vodes 1..L+1 hand on goes
                          * through its own LayerNorm (keys .vode_output_layernorms[i].nn.weight/bias).
                          * Per-op kernels only. */
} cora_config;

/* Results of one step; replaces the 5-tuple returned by train_on_batch
 * (src/decoder_transformer.py:499).  Gradient tensors stay on the device and are fetched with
 * cora_get_vode_grads / cora_get_wgrad. */
typedef struct cora_step_out {
  double final_h_energy;   /* energy at the start of the last inference step (h_energies[-1]) */
  double final_w_energy;   /* energy at the final vodes (learning_step_fn value)               */
  double train_mse;        /* mean((clip(xhat)-clip(x))^2) under status "forward", :488-494     */
  float  infer_ms;         /* device time of the T-step graph (HIP events on the ctx stream)    */
  float  learn_ms;         /* device time of wgrad + optimizer + reconstruction                 */
} cora_step_out;

/* --- lifetime ----------------------------------------------------------------------------- */
/* Replaces TransformerDecoder.__init__ (src/decoder_transformer.py:129-218): allocates vodes,
 * weights, optimizer state and scratch on `device`. */
int cora_create(const cora_config* cfg, int device, cora_ctx** out);
int cora_destroy(cora_ctx* ctx);
const char* cora_last_error(const cora_ctx* ctx /* may be NULL */);

/* --- weights: the LayerParam pytree (pcx/utils/_serialisation.py:25-105) -------------------- */
int cora_weights_size(const cora_ctx* ctx, size_t* n_floats);
int cora_weight_offset(const cora_ctx* ctx, const char* pcx_key, size_t* offset, size_t* count);
int cora_set_weights(cora_ctx* ctx, const float* host, size_t n_floats);
int cora_get_weights(cora_ctx* ctx, float* host, size_t n_floats);
/* positional_embedding [N, D] (src/decoder_transformer.py:202-209), a constant of the model */
int cora_set_pos(cora_ctx* ctx, const float* host, size_t n_floats);

/* --- vodes: VodeParam h (pcx/predictive_coding/_vode.py:198) ------------------------------ */
int cora_vode_size(const cora_ctx* ctx, int l, size_t* n_floats);
int cora_set_vode(cora_ctx* ctx, int l, const float* host, size_t n_floats);
int cora_get_vode(cora_ctx* ctx, int l, float* host, size_t n_floats);
/* Scaled (and, if enabled, normalised) dE/dh of the last inference step: final_h_grad, :476-478 */
int cora_get_vode_grad(cora_ctx* ctx, int l, float* host, size_t n_floats);
/* INIT-status forward (examples/debug_transformer_wandb.py:1204-1211, default rule 'h, u <- u'
 * pcx/predictive_coding/_vode.py:201): h_0 := h0_host (or zeros if NULL), h_l := u_l chained. */
int cora_init_vodes(cora_ctx* ctx, const float* h0_host);

/* --- optimizer state of optim_w (optax.adamw: mu, nu, count) -- pcx/utils/_optim.py:47,128 -- */
int cora_set_adam_state(cora_ctx* ctx, const float* mu, const float* nu, int64_t count);
int cora_get_adam_state(cora_ctx* ctx, float* mu, float* nu, int64_t* count);

/* Replace the optimizer recipe baked in at cora_create (the reference passes optim_h / optim_w to
 * every train_on_batch call, src/decoder_transformer.py:393).  Cached hipGraphs are dropped when
 * a value changes.  AdamW moments and count are kept. */
int cora_set_optim(cora_ctx* ctx, float clip_h, float momentum, float clip_w, float adam_b1,
                   float adam_b2, float adam_eps, float adam_wd);

/* --- the step ------------------------------------------------------------------------------- */
/* Whole train_on_batch (src/decoder_transformer.py:392-499, mode 2): set the sensory vode to x,
 * zero the vode optimizer state, T x (energy, dE/dh, scale, clip, momentum-SGD) captured in one
 * hipGraph, then dE/dW at the final vodes, clip + AdamW (scale_by = 1/B), then the "forward"
 * reconstruction and its clipped MSE.  x: [B,C,H,W] floats on the host (x_on_device=0) or on
 * this ctx's device (1).  eta_h[T]: vode learning rate per step; eta_w: weight learning rate. */
int cora_step(cora_ctx* ctx, const float* x, int x_on_device, int T, const float* eta_h,
              float eta_w, cora_step_out* out);

/* The same step split at the data-parallel exchange point: cora_infer + cora_wgrad leave the
 * LOCAL sum of dE/dW in the wgrad buffer; the host all-reduces that buffer (RCCL) and then calls
 * cora_apply_wgrad with scale_by = 1/B_total.  cora_infer with out == NULL returns while the loop is
 * still in flight (x, if a device pointer, must stay valid); the cora_wgrad that follows ends both
 * with one synchronisation and reports final_h_energy / infer_ms as well. */
int cora_infer(cora_ctx* ctx, const float* x, int x_on_device, int T, const float* eta_h,
               cora_step_out* out);
int cora_wgrad(cora_ctx* ctx, cora_step_out* out);
int cora_apply_wgrad(cora_ctx* ctx, float eta_w, float scale_by, cora_step_out* out);
/* Gradient conditioning of the vode step, changeable after cora_create: exponential layer scaling
 * (:308-350) and per-leaf normalisation (:353-389).  The reference applies both in MODE 2 of
 * train_on_batch (:449-460) and in the linear-probe extractor (examples/linear_probe.py:223-232),
 * only the scaling in unmask_on_batch_enhanced (:925-933) and neither in MODE 1 (:421-424), so a host
 * that mixes them switches here.  Drops the captured inference graphs when something changes. */
int cora_set_grad_options(cora_ctx* ctx, int use_scaling, float scale_base, int use_grad_norm, float grad_norm_target);

/* MODE 1 of train_on_batch, `update_weights_every_inference_step` (iPC,
 * /root/reference/src/decoder_transformer.py:418-439): T iterations of { one vode step on the RAW
 * gradients -- the reference applies neither layer scaling nor gradient normalisation in this mode,
 * so call cora_set_grad_options(ctx, 0, .., 0, ..) first --, dE/dW at the updated vodes,
 * clip + AdamW with scale 1/B }, then the "forward" reconstruction.  eta_h[T], eta_w[T] = the two
 * schedules (the AdamW count advances by T).  out: energies / MSE as in cora_step, the gradients of the
 * LAST iteration stay readable (cora_get_vode_grad / cora_get_wgrad).  Eager launches. */
int cora_step_ipc(cora_ctx* ctx, const float* x, int x_on_device, int T, const float* eta_h,
                  const float* eta_w, cora_step_out* out);

/* The inference loop of cora_infer driven step by step from the host, for the EXACT data-parallel
 * vode-gradient clip: optax.clip_by_global_norm (/root/reference/examples/debug_transformer_wandb.py:1137-1140)
 * takes the norm over the whole batch, so with the batch sharded over ranks the per-vode sums of
 * squares of step t must be summed over ranks between the gradient half and the update half:
 *   cora_infer_begin(x, T, eta_h);
 *   for t < T: cora_infer_grad(t); cora_infer_sumsq(t, row, n, 0); <all-reduce SUM row>;
 *              cora_infer_sumsq(t, row, n, 1); cora_infer_update(t);
 *   cora_infer_end(out)         -- out->final_h_energy is this shard's part
 * row = (L+2)*8 doubles (8 partial sums per vode).  Eager launches, one host round trip per step:
 * the captured single-graph cora_infer with the per-shard norm stays the fast path.
 * On the D = 64 and D = 128 / heads-of-32 fused paths cora_infer_update(t) for t < T - 1 launches nothing: the
 * step is applied by cora_infer_grad(t + 1) as it loads its rows (so row t must be final when that is called).
 * cora_infer_end -- or any other entry point, if the loop is cut short -- brings the vodes up to date. */
int cora_infer_begin(cora_ctx* ctx, const float* x, int x_on_device, int T, const float* eta_h);
int cora_infer_grad(cora_ctx* ctx, int t);
int cora_infer_sumsq(cora_ctx* ctx, int t, double* host_row, size_t n_doubles, int set);
int cora_infer_update(cora_ctx* ctx, int t);
int cora_infer_end(cora_ctx* ctx, cora_step_out* out);

/* Data parallelism with RCCL bound INSIDE the library, so that both exchanges of the sharded step sit in
 * the captured hipGraphs (no host round trip per inference step).  The reference is single-device: the
 * semantics kept are those of its global batch -- energy summed over the batch
 * (/root/reference/src/decoder_transformer.py:305), optax.clip_by_global_norm over the whole batch on every
 * inference step (/root/reference/examples/debug_transformer_wandb.py:1137-1140), dE/dW scaled by 1/B (:484).
 *   rank 0: cora_comm_unique_id(id, 128); the host hands the 128 bytes to every rank (any transport);
 *   every rank: cora_comm_init(ctx, id, 128, rank, world)       -- ncclCommInitRank, collective
 *   per batch:  cora_step_dist(ctx, x_shard, ..., exact_clip, out)
 * cora_step_dist = cora_step on this rank's B images with (exact_clip != 0) one SUM all-reduce of the
 * (L+2)*8 per-vode sums of squares between the gradient half and the update half of every inference step,
 * and one SUM all-reduce of the flat dE/dW buffer before the identical clip + AdamW step with scale
 * 1 / (B * world) on every rank.  exact_clip = 0 clips by the shard's own norm (no collective in the loop;
 * equal to the reference only when no shard clips and use_grad_norm is off).  The first call runs eagerly
 * (RCCL warm-up), later calls replay two graphs.  out: energies / MSE of THIS shard.
 * librccl.so is dlopen'ed by the first cora_comm_* call; the one-GPU entry points never load it. */
int cora_comm_unique_id(void* id128, size_t nbytes);
int cora_comm_init(cora_ctx* ctx, const void* id128, size_t nbytes, int rank, int world);
int cora_comm_info(const cora_ctx* ctx, int* rank, int* world);   /* rank -1 / world 0: no communicator */
int cora_step_dist(cora_ctx* ctx, const float* x, int x_on_device, int T, const float* eta_h, float eta_w,
                   int exact_clip, cora_step_out* out);
/* MODE 1 (iPC, cora_step_ipc above; /root/reference/src/decoder_transformer.py:418-439) on a sharded batch: every one of
 * the T iterations sums the per-vode sums of squares over the ranks between the two halves of its vode step
 * (exact_clip != 0) and the flat dE/dW buffer before its clip + AdamW step (scale 1 / (B * world)), so every rank takes
 * the weight steps of the global batch.  T + T small collectives per batch, eager launches (as cora_step_ipc). */
int cora_step_ipc_dist(cora_ctx* ctx, const float* x, int x_on_device, int T, const float* eta_h,
                       const float* eta_w, int exact_clip, cora_step_out* out);
/* Graph vs eager collectives, agreed across ranks before the first replay (no counterpart in the reference,
 * which is single-device; its jit compiles one executable per (config, T, B), src/decoder_transformer.py:392).
 * After the first cora_step_dist (which always runs eagerly), cora_dist_prepare CAPTURES -- does not launch --
 * the two graphs the next call with this (T, exact_clip) would replay: 0 = ready, 1 = a collective was refused
 * under stream capture on this system (the ctx then stays on eager launches, which issue the same collectives
 * in the same order), < 0 error.  The host all-reduces MIN of "ready" over its own process group and, unless
 * every rank is ready, calls cora_dist_set_eager(ctx, 1) on all of them, so a run is either graph replay
 * everywhere or eager everywhere (cora_amd/dist.py: DataParallel.agree_on_graphs).  cora_dist_mode reports how
 * the LAST cora_step_dist ran: 0 none yet, 1 eager launches, 2 graph replay -- bench.py prints it, so an eager
 * number cannot pass for a graphed one. */
int cora_dist_prepare(cora_ctx* ctx, int T, int exact_clip);
int cora_dist_set_eager(cora_ctx* ctx, int eager);
int cora_dist_mode(const cora_ctx* ctx);

/* Device pointer + length of the flat dE/dW buffer (same layout as the weight arena). */
int cora_wgrad_buffer(cora_ctx* ctx, void** dev_ptr, size_t* n_floats);
/* Let the caller own the buffer instead (e.g. a torch tensor used for the all-reduce). */
int cora_bind_wgrad_buffer(cora_ctx* ctx, void* dev_ptr, size_t n_floats);
int cora_get_wgrad(cora_ctx* ctx, float* host, size_t n_floats);   /* final_w_grad, :483 */

/* Per-step energies of the last cora_infer: total[T] and per-vode [T, L+3] (index l = vode). */
int cora_get_energies(cora_ctx* ctx, int T, double* total, double* per_layer);

/* `forward(None, model=...)` under status "forward" (:488-489): xhat [B,C,H,W] to the host. */
int cora_forward(cora_ctx* ctx, float* xhat_host, size_t n_floats);
/* cora_forward without a ctx and WITHOUT ANY ALLOCATION -- no counterpart in the reference.  For callers
 * that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  The status-"forward" reconstruction (:488-489) of B images from their top latents:
 *   W_dev   [n_weights]  flat weight arena on `device` (layout of cora_weight_offset / cora_get_weights)
 *   pos_dev [N, D], h0_dev [B, N, P]  ->  xhat_dev [B, C, H, W]   (all device pointers, caller-owned)
 * Every intermediate lives in a 16 MiB static __device__ array of the library (one copy per GPU, held
 * until the process ends); cora_forward_static_floats(cfg) says how much of it a configuration needs and
 * the call fails with a clear error above 4 Mi floats (cfg 2 shape: 56 640 floats per image, B <= 74).  Only kernel launches on
 * `hip_stream` (a hipStream_t, NULL = the null stream): no hipMalloc / hipFree, no copies, no
 * synchronisation.  The FIRST call on a device resolves the array's address (hipGetSymbolAddress) and so
 * must not sit inside a stream capture unless a cora_create on that device came first.  All calls on one
 * GPU share that GPU's array (it keeps the last call's contents; nothing in this path needs it zeroed):
 * calls that overlap on one GPU race.  Host side: thread-safe (the per-device address is resolved under a lock,
 * the error string is per thread).  Errors: cora_last_error(NULL). */
size_t cora_forward_static_floats(const cora_config* cfg);
int cora_forward_static(const cora_config* cfg, int device, const float* W_dev, size_t n_weights,
                        const float* pos_dev, const float* h0_dev, float* xhat_dev, void* hip_stream);

/* `unmask_on_batch_enhanced` (/root/reference/src/decoder_transformer.py:833-1046): the inference-only /
 * inpainting loop.  x_true, x_init, mask: host [B,C,H,W] (mask already broadcast over channels,
 * 1 = masked; ignored unless use_corruption).  The sensory vode starts from x_init (:896-901); with
 * use_corruption it is un-frozen for the T steps (:904-906): gradient h_s - u with layer scale
 * base^-1 (:324-326), part of the global-norm clip and the momentum step (:965), then the known
 * pixels are blended towards x_true with weight alpha (:982-994).  status_init != 0 re-chains
 * h_l := u_l from the current h_0 first (:899-901; h_0 itself is whatever cora_set_vode left).
 * Outputs (host, each may be NULL): recons [T,B,C,H,W] = status-"forward" reconstruction before
 * every step (:916-919); xhat = the final one (:1002-1004); sensory = final sensory state;
 * grad_norms [T, L+3] = L2 norm of every vode's scaled gradient per step (:936-958).
 * eta_h[T] as in cora_infer.  The loss (:1006-1034) is plain host arithmetic on xhat. */
int cora_unmask(cora_ctx* ctx, const float* x_true, const float* x_init, const float* mask,
                int use_corruption, int status_init, int T, const float* eta_h, float alpha,
                float* recons, float* xhat, float* sensory, double* grad_norms);

/* The same loop with `update_weights_during_unmasking` (:967-979): after every vode step, dE/dW at the
 * updated vodes (target = the current sensory state) and one clip + AdamW weight step with learning
 * rate eta_w[t] and gradient scale `scale_by` (the reference passes 1 / x.shape[0], :978), before the
 * clamp.  The AdamW count advances by T. */
int cora_unmask_learn(cora_ctx* ctx, const float* x_true, const float* x_init, const float* mask,
                      int use_corruption, int status_init, int T, const float* eta_h, float alpha,
                      const float* eta_w, float scale_by, float* recons, float* xhat, float* sensory,
                      double* grad_norms);

/* `energy(model=...)` (:302-305) with the sensory vode set to x (host, may be NULL to keep the
 * last one): batch-summed E, per-vode E [L+3] (may be NULL) and the status-None prediction. */
int cora_energy(cora_ctx* ctx, const float* x_host, double* E, double* per_layer,
                float* xhat_host, size_t n_floats);

/* --- measurement ---------------------------------------------------------------------------- */
/* Runs inference steps eagerly (step size 0; vodes and momentum are restored afterwards) with HIP events on the ctx
 * stream and reports, per kernel kind, launches / total ms / algorithmic FLOPs and bytes over 5 steps.  Before it
 * records it keeps the GPU under load for >= 60 ms (clocks take tens of milliseconds to come back after an idle
 * period), and the fused block kernel is timed as a burst of 8 identical back-to-back launches under one event
 * pair, counted as one launch of the average duration -- what a graph replay executes.  names: `cap` x char[32].
 * No counterpart in the reference (bench.py's roofline leg). */
int cora_profile_step(cora_ctx* ctx, int cap, char* names, int* launches, double* ms,
                      double* flops, double* bytes, int* n_out);
int cora_sync(cora_ctx* ctx);
/* Which kernels this ctx's inference step runs: 0 = per-op path (grouped GEMM / attention / LayerNorm
 * launches), 1 = the fused per-(block, image) kernel (D = 64, 64 tokens), 2 = its wide variant (D = 96 / 128,
 * inference step only), 3 = the row-tile fused kernels around the attention kernels (D = 64, any sequence
 * length: the 1024-token video shape); -1 for a NULL ctx.  For reports and tests. */
int cora_kernel_path(const cora_ctx* ctx);
/* Diagnostic builds only (`make -C cora_amd/csrc stamps`, -DCORA_STAMPS): the first call allocates the
 * per-workgroup s_memtime stamp table of the fused kernel, later calls copy it out ((L+1)*B*16 words).
 * In the shipped library the kernels never write it.  No counterpart in the reference. */
int cora_dbg_stamps(cora_ctx* ctx, unsigned long long* host, size_t n_words);

#ifdef __cplusplus
}
#endif
#endif /* CORA_HIP_H */
