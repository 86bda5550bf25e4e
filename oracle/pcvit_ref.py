"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of Cora's PC-ViT step.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.
Nothing under `cora_amd/` imports it; the product path is the HIP library and fails loudly without it.

What it restates (all paths under /root/reference, read as text only -- the reference is pure
Python on JAX/equinox/optax, none of which is installed here, so nothing of it was imported or run):

  model forward (status None / "forward") ..... src/decoder_transformer.py:220-256
  unpatchify .................................. src/decoder_transformer.py:259-294
  vmapped energy + psum over the batch ........ src/decoder_transformer.py:302-305
  layer-wise inference-LR scaling ............. src/decoder_transformer.py:308-350
  vode gradient normalisation (opt-in) ........ src/decoder_transformer.py:353-389
  train_on_batch step order ................... src/decoder_transformer.py:392-499
  TransformerBlock / MHA / LN / Linear ........ pcx/nn/_layer.py:77-87,296-315,359-447
  Vode call semantics (store u, return h) ..... pcx/predictive_coding/_vode.py:204-236,282-354
  se / regularised energies ................... pcx/predictive_coding/_energy.py:27-30,38-68
  optimizer wrapper ........................... pcx/utils/_optim.py:53-169
  optimizer recipe (clip -> sgd / adamw) ...... examples/debug_transformer_wandb.py:1116-1167
  LR schedule ................................. examples/debug_transformer_wandb.py:157-197

Third-party arithmetic that is NOT under /root/reference and is restated from its published
definition (pinned versions from poetry.lock: equinox 0.11.8, jax 0.4.35, optax 0.2.3):
  eqx.nn.Linear            y = W x + b, W:[out,in]
  eqx.nn.LayerNorm         (x-mean)*rsqrt(max(var,0)+eps)*gamma+beta, biased variance, eps=1e-5
  eqx.nn.MultiheadAttention  q,k,v = x W^T (no bias), heads = contiguous column blocks,
                           logits = (q/sqrt(d)) k^T, softmax over keys, concat, output W^T (no bias)
  jax.nn.gelu              approximate=True (tanh form)
  optax.clip_by_global_norm  g if ||g|| < c else g/||g||*c
  optax.sgd(momentum)      trace: m <- g + mu*m ; update = -lr(count)*m  (no nesterov)
  optax.adamw              mu,nu EMA, bias correction by count, /(sqrt(nu_hat)+eps), + wd*p, * -lr

PARITY PINNING.  The reference's own tests pin only: se_energy (tests/pcx/test_vode.py:55-75),
Linear known answer (tests/pcx/test_nn.py:25-48, tests/conftest.py:100-106) and LayerNorm
zero-mean/bias-shift (tests/pcx/test_nn.py:135-165).  Those are re-encoded in
tests/test_oracle.py.  Attention, the block, the decoder, scaling, clipping, momentum, AdamW and
the T-step trajectory have no reference test, golden vector or fixture: for those this oracle is
**parity unpinned** (it is cross-checked only against the independent hand-VJP in pcvit_np.py).

Gradients here come from torch autograd (fp64 by default), mirroring the reference's use of
jax.value_and_grad; oracle/pcvit_np.py re-derives them by hand with NumPy.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

# ----------------------------------------------------------------------------------------------
# dims / hyper-parameters
# ----------------------------------------------------------------------------------------------


@dataclass
class Dims:
    """Derived dimensions, src/decoder_transformer.py:100-125."""
    image_shape: tuple  # (C,H,W) or (F,C,H,W)
    patch_size: int
    hidden_size: int
    num_heads: int
    num_blocks: int
    mlp_ratio: float = 4.0

    def __post_init__(self):
        if len(self.image_shape) == 4:
            self.F, self.C, self.Himg, self.Wimg = self.image_shape
            self.is_video = True
        else:
            self.C, self.Himg, self.Wimg = self.image_shape
            self.F = 1
            self.is_video = False
        p = self.patch_size
        self.hp, self.wp = self.Himg // p, self.Wimg // p
        self.N = self.F * self.hp * self.wp
        self.P = p * p * self.C
        self.D = self.hidden_size
        self.H = self.num_heads
        self.L = self.num_blocks
        self.M = int(self.hidden_size * self.mlp_ratio)


@dataclass
class Hyper:
    """Step hyper-parameters (ModelConfig fields, src/config.py:10-117)."""
    l1: float = 0.0
    l2: float = 0.0
    use_scaling: bool = True
    scale_base: Optional[float] = 1.25
    clip_h: Optional[float] = 2000.0
    clip_w: Optional[float] = 500.0
    momentum: float = 0.4
    wd: float = 2e-4
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-8
    use_grad_norm: bool = False
    grad_norm_target: float = 1.0


# ----------------------------------------------------------------------------------------------
# pcx key strings (pcx/utils/_serialisation.py:47-60 + pcx/core/_module.py:72-84)
# ----------------------------------------------------------------------------------------------

def block_keys(i: int) -> Dict[str, str]:
    b = f".transformer_blocks[{i}]"
    return {
        "ln1_g": f"{b}.layer_norm1.nn.weight", "ln1_b": f"{b}.layer_norm1.nn.bias",
        "ln2_g": f"{b}.layer_norm2.nn.weight", "ln2_b": f"{b}.layer_norm2.nn.bias",
        "wq": f"{b}.attention.nn.query_proj.weight", "wk": f"{b}.attention.nn.key_proj.weight",
        "wv": f"{b}.attention.nn.value_proj.weight", "wo": f"{b}.attention.nn.output_proj.weight",
        "w1": f"{b}.linear1.nn.weight", "b1": f"{b}.linear1.nn.bias",
        "w2": f"{b}.linear2.nn.weight", "b2": f"{b}.linear2.nn.bias",
    }


K_WP, K_BP = ".patch_projection.nn.weight", ".patch_projection.nn.bias"
K_WOUT, K_BOUT = ".output_projection.nn.weight", ".output_projection.nn.bias"


def vln_keys(i: int):
    """Vode-state LayerNorm i (after vode i+1), src/decoder_transformer.py:212-218."""
    return f".vode_output_layernorms[{i}].nn.weight", f".vode_output_layernorms[{i}].nn.bias"


# ----------------------------------------------------------------------------------------------
# elementary ops
# ----------------------------------------------------------------------------------------------

def se_energy(h, u):
    """pcx/predictive_coding/_energy.py:27-30 (element-wise)."""
    e = h - u
    return 0.5 * (e * e)


def regularized_plus_se_energy(h, u, l1, l2):
    """pcx/predictive_coding/_energy.py:38-68 (element-wise)."""
    return 0.5 * (h - u) ** 2 + l1 * h.abs() + l2 * h * h


def linear(x, W, b=None):
    """eqx.nn.Linear vmapped over leading axes: y = x W^T + b (W:[out,in])."""
    y = x @ W.transpose(-1, -2)
    return y if b is None else y + b


def layer_norm(x, g, b, eps=1e-5):
    """eqx.nn.LayerNorm per token; pcx/nn/_layer.py:296-315 (eps 1e-5)."""
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    inv = torch.rsqrt(torch.clamp(var, min=0.0) + eps)
    return (x - mean) * inv * g + b


def gelu_tanh(x):
    """jax.nn.gelu(approximate=True), the default used at pcx/nn/_layer.py:439."""
    c = math.sqrt(2.0 / math.pi)
    return 0.5 * x * (1.0 + torch.tanh(c * (x + 0.044715 * x ** 3)))


def mha(xn, Wq, Wk, Wv, Wo, H):
    """eqx.nn.MultiheadAttention(num_heads, D) self-attention, no bias, no mask, no dropout."""
    Bn, N, D = xn.shape
    d = D // H
    q = linear(xn, Wq).reshape(Bn, N, H, d).transpose(1, 2)
    k = linear(xn, Wk).reshape(Bn, N, H, d).transpose(1, 2)
    v = linear(xn, Wv).reshape(Bn, N, H, d).transpose(1, 2)
    logits = (q / math.sqrt(d)) @ k.transpose(-1, -2)
    p = torch.softmax(logits, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(Bn, N, D)
    return linear(o, Wo)


def transformer_block(x, W, i, H):
    """pcx/nn/_layer.py:428-447 (dropout is the identity: inference=not enable_dropout=True)."""
    k = block_keys(i)
    xn = layer_norm(x, W[k["ln1_g"]], W[k["ln1_b"]])
    y = x + mha(xn, W[k["wq"]], W[k["wk"]], W[k["wv"]], W[k["wo"]], H)
    z = layer_norm(y, W[k["ln2_g"]], W[k["ln2_b"]])
    z = gelu_tanh(linear(z, W[k["w1"]], W[k["b1"]]))
    z = linear(z, W[k["w2"]], W[k["b2"]])
    return y + z


def unpatchify(z, dims: Dims):
    """src/decoder_transformer.py:259-294: 'n (ph pw c) -> c (hn ph) (wn pw)'.
    Video extension (the reference's video path does not run end-to-end): frame-major
    patches, n = f*(hp*wp) + i*wp + j -> [F,C,H,W]."""
    Bn = z.shape[0]
    p, C = dims.patch_size, dims.C
    z = z.reshape(Bn, dims.F, dims.hp, dims.wp, p, p, C)
    img = z.permute(0, 1, 6, 2, 4, 3, 5).reshape(Bn, dims.F, C, dims.Himg, dims.Wimg)
    return img if dims.is_video else img[:, 0]


def patchify(img, dims: Dims):
    """inverse of unpatchify."""
    Bn = img.shape[0]
    p, C = dims.patch_size, dims.C
    x = img.reshape(Bn, dims.F, C, dims.hp, p, dims.wp, p)
    return x.permute(0, 1, 3, 5, 4, 6, 2).reshape(Bn, dims.N, dims.P)


# ----------------------------------------------------------------------------------------------
# model forward: status None (each vode returns h) and status "forward" (each vode returns u)
# ----------------------------------------------------------------------------------------------

def predictions(h: Sequence[torch.Tensor], W, pos, dims: Dims, chain: bool = False):
    """TransformerDecoder.__call__ (src/decoder_transformer.py:220-256).
    Returns u_1..u_{L+2}; u_{L+2} is the image prediction.  chain=False: status None, block i reads
    h_{i+1} (pcx/predictive_coding/_vode.py:223-236).  chain=True: rule 'h -> u'
    (src/decoder_transformer.py:167-168,175-176): each vode hands its u downstream."""
    vln = vln_keys(0)[0] in W            # use_vode_state_layernorm: the value a vode hands on is normalised, :232-241

    def hand_on(x, i):
        if not vln:
            return x
        kg, kb = vln_keys(i)
        return layer_norm(x, W[kg], W[kb])

    us = []
    u = linear(h[0], W[K_WP], W[K_BP]) + pos
    us.append(u)
    x = hand_on(u if chain else h[1], 0)
    for i in range(dims.L):
        u = x + transformer_block(x, W, i, dims.H)   # double residual, :238
        us.append(u)
        x = hand_on(u if chain else h[i + 2], i + 1)
    z = linear(x, W[K_WOUT], W[K_BOUT])
    us.append(unpatchify(z, dims))
    return us


def layer_energies(h, us, y, l1, l2):
    """Per-vode energies summed over the batch: vode0 has energy_fn=None -> 0
    (pcx/predictive_coding/_vode.py:337,351-352); vodes 1..L+1 regularised+SE; sensory SE."""
    out = [torch.zeros((), dtype=us[0].dtype)]
    for l in range(1, len(us)):
        out.append(regularized_plus_se_energy(h[l], us[l - 1], l1, l2).sum())
    out.append(se_energy(y, us[-1]).sum())
    return out


def lr_schedule(base_lr, warmup_steps, total_steps, min_lr_factor):
    """examples/debug_transformer_wandb.py:157-197 (float64 here; the reference computes in fp32)."""
    min_lr = base_lr * min_lr_factor

    def f(step):
        step = float(step)
        dur = max(float(total_steps - warmup_steps), 1e-8)
        ratio = min(max((step - warmup_steps) / dur, 0.0), 1.0)
        decay = min_lr + (base_lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * ratio))
        if warmup_steps > 0:
            warm = base_lr * min(max(step / float(warmup_steps), 0.0), 1.0)
            return warm if step < warmup_steps else decay
        return decay
    return f


# ----------------------------------------------------------------------------------------------
# the step
# ----------------------------------------------------------------------------------------------

@dataclass
class AdamState:
    mu: Dict[str, np.ndarray] = field(default_factory=dict)
    nu: Dict[str, np.ndarray] = field(default_factory=dict)
    count: int = 0


def _global_norm(ts):
    return torch.sqrt(sum((t * t).sum() for t in ts))


def _clip(ts, c):
    """optax.clip_by_global_norm."""
    if c is None or c <= 0:
        return list(ts)
    n = _global_norm(ts)
    if bool(n < c):
        return list(ts)
    return [t / n * c for t in ts]


def h_gradients(h, W, pos, y, dims, hp: Hyper):
    """value_and_grad(energy) w.r.t. un-frozen VodeParams, src/decoder_transformer.py:396,445."""
    hv = [t.detach().clone().requires_grad_(True) for t in h]
    us = predictions(hv, W, pos, dims)
    Es = layer_energies(hv, us, y, hp.l1, hp.l2)
    E = sum(Es)
    grads = torch.autograd.grad(E, hv, allow_unused=True)
    grads = [g if g is not None else torch.zeros_like(t) for g, t in zip(grads, hv)]
    return E.detach(), [e.detach() for e in Es], grads, us[-1].detach()


def scale_h_grads(grads, dims: Dims, hp: Hyper, normalise: bool = True):
    """apply_exponential_layer_scaling (:308-350) then, where the caller does so, normalize_vode_h_gradients
    (:353-389).  train_on_batch MODE 2 (:449-460) and examples/linear_probe.py:223-232 do both;
    unmask_on_batch_enhanced (:925-933) only scales."""
    out = list(grads)
    if hp.use_scaling:
        base = hp.scale_base if hp.scale_base is not None else 1.0
        base_idx = dims.L + 1
        out = [g * (base ** float(base_idx - l)) for l, g in enumerate(out)]
    if hp.use_grad_norm and normalise:
        out = [g / (torch.sqrt((g * g).sum()) + 1e-8) * hp.grad_norm_target for g in out]
    return out


def train_on_batch(T: int, x: np.ndarray, weights: Dict[str, np.ndarray], h: List[np.ndarray],
                   pos: np.ndarray, dims: Dims, hp: Hyper, eta_h: Sequence[float], eta_w: float,
                   adam: Optional[AdamState] = None, dtype=torch.float64, update_weights=True):
    """src/decoder_transformer.py:392-499, MODE 2 (T inference steps, then one weight update).

    h: L+2 arrays (h_0 [B,N,P], h_1..h_{L+1} [B,N,D]); they persist across batches.
    x: [B,C,H,W] (or [B,F,C,H,W]); the sensory vode's h is set to x (:253-254).
    eta_h[t]: the vode learning rate at inference step t (optimizer state is re-initialised every
    batch, :414, so the schedule count restarts at 0).  Returns a dict of everything the parity
    tests compare."""
    W = {k: torch.as_tensor(np.asarray(v), dtype=dtype) for k, v in weights.items()}
    hs = [torch.as_tensor(np.asarray(t), dtype=dtype) for t in h]
    pos_t = torch.as_tensor(np.asarray(pos), dtype=dtype)
    y = torch.as_tensor(np.asarray(x), dtype=dtype)
    B = y.shape[0]
    mom = [torch.zeros_like(t) for t in hs]           # optim_h.init, :414
    energies, per_layer = [], []
    g_scaled = None
    for t in range(T):
        E, Es, grads, _ = h_gradients(hs, W, pos_t, y, dims, hp)
        energies.append(float(E))
        per_layer.append([float(e) for e in Es])
        g_scaled = scale_h_grads(grads, dims, hp)
        g_clip = _clip(g_scaled, hp.clip_h)
        mom = [g + hp.momentum * m for g, m in zip(g_clip, mom)]     # optax.trace
        hs = [hh - eta_h[t] * m for hh, m in zip(hs, mom)]
    # learning step at the final h, :482-484
    Wg = {k: v.detach().clone().requires_grad_(True) for k, v in W.items()}
    us = predictions(hs, Wg, pos_t, dims)
    Es_w = layer_energies(hs, us, y, hp.l1, hp.l2)
    Ew = sum(Es_w)
    keys = list(Wg.keys())
    wg = torch.autograd.grad(Ew, [Wg[k] for k in keys])
    w_grad = {k: g.detach() for k, g in zip(keys, wg)}     # final_w_grad (unscaled, as returned :499)
    out = {
        "energies": np.array(energies), "layer_energies": np.array(per_layer),
        "final_h_grad": [g.detach().numpy() for g in g_scaled] if g_scaled is not None else None,
        "final_w_energy": float(Ew.detach()), "w_layer_energies": np.array([float(e.detach()) for e in Es_w]),
        "w_grad": {k: v.numpy() for k, v in w_grad.items()},
        "h": [t.detach().numpy() for t in hs],
    }
    newW = W
    if update_weights:
        adam = adam if adam is not None else AdamState()
        g_list = [w_grad[k] * (1.0 / B) for k in keys]                # scale_by=1/B, :484
        g_list = _clip(g_list, hp.clip_w)
        adam.count += 1
        newW = {}
        for k, g in zip(keys, g_list):
            mu = torch.as_tensor(adam.mu.get(k, np.zeros(g.shape)), dtype=dtype)
            nu = torch.as_tensor(adam.nu.get(k, np.zeros(g.shape)), dtype=dtype)
            mu = hp.b1 * mu + (1 - hp.b1) * g
            nu = hp.b2 * nu + (1 - hp.b2) * g * g
            mu_hat = mu / (1 - hp.b1 ** adam.count)
            nu_hat = nu / (1 - hp.b2 ** adam.count)
            upd = mu_hat / (torch.sqrt(nu_hat) + hp.eps) + hp.wd * W[k]
            newW[k] = W[k] - eta_w * upd
            adam.mu[k], adam.nu[k] = mu.numpy(), nu.numpy()
        out["adam"] = adam
    out["weights"] = {k: v.detach().numpy() for k, v in newW.items()}
    # reconstruction under status "forward" with the UPDATED weights, :488-494
    with torch.no_grad():
        xhat = predictions(hs, newW, pos_t, dims, chain=True)[-1]
        mse = ((xhat.clamp(0.0, 1.0) - y.clamp(0.0, 1.0)) ** 2).mean()
    out["xhat"] = xhat.numpy()
    out["train_mse"] = float(mse)
    return out


def init_vodes(h0: np.ndarray, weights, pos, dims: Dims, dtype=torch.float64):
    """INIT pass (examples/debug_transformer_wandb.py:1204-1211; default rule 'h, u <- u',
    pcx/predictive_coding/_vode.py:201): h_0 is given (0.01*N(0,1) or zeros, :145-152), h_l := u_l
    chained.  Returns h_0..h_{L+1}."""
    W = {k: torch.as_tensor(np.asarray(v), dtype=dtype) for k, v in weights.items()}
    pos_t = torch.as_tensor(np.asarray(pos), dtype=dtype)
    h0t = torch.as_tensor(np.asarray(h0), dtype=dtype)
    with torch.no_grad():
        us = predictions([h0t], W, pos_t, dims, chain=True)
    return [h0t.numpy()] + [u.numpy() for u in us[:-1]]


def energy_and_prediction(h, weights, pos, x, dims: Dims, hp: Hyper, dtype=torch.float64):
    """`energy(model=...)` (src/decoder_transformer.py:302-305): (batch-summed E, image prediction)."""
    W = {k: torch.as_tensor(np.asarray(v), dtype=dtype) for k, v in weights.items()}
    hs = [torch.as_tensor(np.asarray(t), dtype=dtype) for t in h]
    pos_t = torch.as_tensor(np.asarray(pos), dtype=dtype)
    y = torch.as_tensor(np.asarray(x), dtype=dtype)
    with torch.no_grad():
        us = predictions(hs, W, pos_t, dims)
        Es = layer_energies(hs, us, y, hp.l1, hp.l2)
    return float(sum(Es)), np.array([float(e) for e in Es]), us[-1].numpy()


def unmask_on_batch(T: int, x_true: np.ndarray, x_init: np.ndarray, mask: Optional[np.ndarray],
                    weights, h: List[np.ndarray], pos, dims: Dims, hp: Hyper, eta_h: Sequence[float],
                    alpha: float = 1.0, use_corruption: bool = True, status_init: bool = False,
                    dtype=torch.float64, normalise: bool = False, eta_w: Optional[Sequence[float]] = None,
                    adam: Optional[AdamState] = None, scale_by: Optional[float] = None):
    """`unmask_on_batch_enhanced` (src/decoder_transformer.py:833-1046), the inference-only /
    inpainting loop.  x_init is what the sensory vode starts from (the corrupted image, :896,900-901),
    mask is [B,1,H,W] with 1 = masked (:882-894).  With use_corruption the sensory vode is un-frozen
    (:904-906): it gets the gradient dE/dh_s = h_s - u_{L+2}, the layer scale base^{(L+1)-(L+2)}
    (:324-326 with vode index L+2), takes part in the global-norm clip and the momentum step (:965),
    and after each step the known pixels are blended back towards the ground truth (:982-994).
    status_init: the INIT pass of :899-901 re-chains h_l := u_l from h_0 (h_0 itself is kept: the
    reference re-draws it from its own RNG, which is not reproduced).  Returns the per-step
    status-"forward" reconstructions (:916-919), the final one (:1002-1004), the loss (:1006-1032)
    and the final states.
    normalise: False = unmask_on_batch_enhanced (layer scaling only, :925-933); True = the linear-probe
    extractor (examples/linear_probe.py:223-232), which also normalises each leaf.
    eta_w: `update_weights_during_unmasking` (:967-979): after every vode step, dE/dW at the updated
    vodes and one clip + AdamW step with scale_by (= 1/x.shape[0], :978)."""
    W = {k: torch.as_tensor(np.asarray(v), dtype=dtype) for k, v in weights.items()}
    pos_t = torch.as_tensor(np.asarray(pos), dtype=dtype)
    hs = [torch.as_tensor(np.asarray(t), dtype=dtype) for t in h]
    xt = torch.as_tensor(np.asarray(x_true), dtype=dtype)
    if eta_w is not None:
        adam = adam if adam is not None else AdamState()
        scale_by = scale_by if scale_by is not None else 1.0 / xt.shape[0]
    hsens = torch.as_tensor(np.asarray(x_init), dtype=dtype).clone()
    if status_init:
        with torch.no_grad():
            us = predictions([hs[0]], W, pos_t, dims, chain=True)
        hs = [hs[0]] + list(us[:-1])
    mk = None
    if use_corruption and mask is not None:
        mk = torch.as_tensor(np.asarray(mask), dtype=dtype).expand_as(xt)
    state = hs + ([hsens] if use_corruption else [])
    mom = [torch.zeros_like(t) for t in state]
    recons, energies, norms = [], [], []
    for t in range(T):
        with torch.no_grad():
            recons.append(predictions(hs, W, pos_t, dims, chain=True)[-1].numpy())
        hv = [s.detach().clone().requires_grad_(True) for s in state]
        y = hv[-1] if use_corruption else hsens
        us = predictions(hv[:dims.L + 2], W, pos_t, dims)
        E = sum(layer_energies(hv[:dims.L + 2], us, y, hp.l1, hp.l2))
        grads = torch.autograd.grad(E, hv, allow_unused=True)
        grads = [g if g is not None else torch.zeros_like(s) for g, s in zip(grads, hv)]
        energies.append(float(E.detach()))
        g_scaled = scale_h_grads(grads, dims, hp, normalise)        # vode index L+2 -> base^-1
        norms.append([float(torch.sqrt((g * g).sum())) for g in g_scaled])
        g_clip = _clip(g_scaled, hp.clip_h)
        mom = [g + hp.momentum * m for g, m in zip(g_clip, mom)]
        state = [s - eta_h[t] * m for s, m in zip(state, mom)]
        hs = state[:dims.L + 2]
        if eta_w is not None:                              # :967-979, before the clamp
            ysens = state[-1] if use_corruption else hsens
            Wg = {k: v.detach().clone().requires_grad_(True) for k, v in W.items()}
            us_w = predictions(hs, Wg, pos_t, dims)
            Ew = sum(layer_energies(hs, us_w, ysens, hp.l1, hp.l2))
            keys = list(Wg.keys())
            wg = torch.autograd.grad(Ew, [Wg[k] for k in keys])
            g_list = _clip([g.detach() * scale_by for g in wg], hp.clip_w)
            adam.count += 1
            newW = {}
            for k, g in zip(keys, g_list):
                mu = torch.as_tensor(adam.mu.get(k, np.zeros(g.shape)), dtype=dtype)
                nu = torch.as_tensor(adam.nu.get(k, np.zeros(g.shape)), dtype=dtype)
                mu = hp.b1 * mu + (1 - hp.b1) * g
                nu = hp.b2 * nu + (1 - hp.b2) * g * g
                upd = (mu / (1 - hp.b1 ** adam.count)) / (torch.sqrt(nu / (1 - hp.b2 ** adam.count)) + hp.eps) + hp.wd * W[k]
                newW[k] = W[k] - eta_w[t] * upd
                adam.mu[k], adam.nu[k] = mu.numpy(), nu.numpy()
            W = newW
        if use_corruption:
            cur = state[-1]
            target = torch.where(mk == 0, xt, cur)
            state[-1] = alpha * target + (1.0 - alpha) * cur
            hsens = state[-1]
    with torch.no_grad():
        xhat = predictions(hs, W, pos_t, dims, chain=True)[-1]
    if use_corruption:
        nm = mk.sum()
        nu = float(xhat.numel()) - nm
        d2 = (xhat.clamp(0.0, 1.0) - xt) ** 2
        loss = ((d2 * mk).sum() / torch.clamp(nm, min=1e-8) + (d2 * (1 - mk)).sum() / torch.clamp(nu, min=1e-8)) / 2
    else:
        loss = ((xhat.clamp(0.0, 1.0).reshape(-1) - xt.reshape(-1)) ** 2).mean()
    return {"recons": recons, "xhat": xhat.numpy(), "loss": float(loss), "energies": np.array(energies),
            "grad_norms": np.array(norms), "h": [t.numpy() for t in hs], "sensory": hsens.numpy(),
            "weights": {k: v.detach().numpy() for k, v in W.items()}}


def train_on_batch_ipc(T: int, x: np.ndarray, weights: Dict[str, np.ndarray], h: List[np.ndarray], pos: np.ndarray,
                       dims: Dims, hp: Hyper, eta_h: Sequence[float], eta_w: Sequence[float],
                       adam: Optional[AdamState] = None, dtype=torch.float64):
    """src/decoder_transformer.py:392-499, MODE 1 (`update_weights_every_inference_step`, :418-439):
    every iteration is one vode step on the RAW gradients (no layer scaling, no normalisation: the
    scan body at :421-424 hands h_grad straight to optim_h) followed by dE/dW at the updated vodes
    and one clip + AdamW weight step with scale 1/B (:426-428).  Returns what the parity test compares."""
    W = {k: torch.as_tensor(np.asarray(v), dtype=dtype) for k, v in weights.items()}
    hs = [torch.as_tensor(np.asarray(t), dtype=dtype) for t in h]
    pos_t = torch.as_tensor(np.asarray(pos), dtype=dtype)
    y = torch.as_tensor(np.asarray(x), dtype=dtype)
    B = y.shape[0]
    adam = adam if adam is not None else AdamState()
    mom = [torch.zeros_like(t) for t in hs]
    keys = list(W.keys())
    energies, w_energies = [], []
    grads = w_grad = None
    w_grad_minabs = None          # per entry, the smallest |dE/dW| over the T steps (the parity test's noise filter)
    for t in range(T):
        E, _, grads, _ = h_gradients(hs, W, pos_t, y, dims, hp)
        energies.append(float(E))
        g_clip = _clip(list(grads), hp.clip_h)
        mom = [g + hp.momentum * m for g, m in zip(g_clip, mom)]
        hs = [hh - eta_h[t] * m for hh, m in zip(hs, mom)]
        Wg = {k: v.detach().clone().requires_grad_(True) for k, v in W.items()}
        us = predictions(hs, Wg, pos_t, dims)
        Ew = sum(layer_energies(hs, us, y, hp.l1, hp.l2))
        w_energies.append(float(Ew.detach()))
        wg = torch.autograd.grad(Ew, [Wg[k] for k in keys])
        w_grad = {k: g.detach() for k, g in zip(keys, wg)}
        w_grad_minabs = ({k: v.abs() for k, v in w_grad.items()} if w_grad_minabs is None else
                         {k: torch.minimum(w_grad_minabs[k], v.abs()) for k, v in w_grad.items()})
        g_list = _clip([w_grad[k] * (1.0 / B) for k in keys], hp.clip_w)
        adam.count += 1
        newW = {}
        for k, g in zip(keys, g_list):
            mu = torch.as_tensor(adam.mu.get(k, np.zeros(g.shape)), dtype=dtype)
            nu = torch.as_tensor(adam.nu.get(k, np.zeros(g.shape)), dtype=dtype)
            mu = hp.b1 * mu + (1 - hp.b1) * g
            nu = hp.b2 * nu + (1 - hp.b2) * g * g
            upd = (mu / (1 - hp.b1 ** adam.count)) / (torch.sqrt(nu / (1 - hp.b2 ** adam.count)) + hp.eps) + hp.wd * W[k]
            newW[k] = W[k] - eta_w[t] * upd
            adam.mu[k], adam.nu[k] = mu.numpy(), nu.numpy()
        W = newW
    with torch.no_grad():
        xhat = predictions(hs, W, pos_t, dims, chain=True)[-1]
        mse = ((xhat.clamp(0.0, 1.0) - y.clamp(0.0, 1.0)) ** 2).mean()
    return {"energies": np.array(energies), "w_energies": np.array(w_energies),
            "final_h_grad": [g.detach().numpy() for g in grads], "w_grad": {k: v.numpy() for k, v in w_grad.items()},
            "w_grad_minabs": {k: v.numpy() for k, v in w_grad_minabs.items()},
            "h": [t.detach().numpy() for t in hs], "weights": {k: v.detach().numpy() for k, v in W.items()},
            "train_mse": float(mse), "adam": adam}
