"""Inference-only consumers of the T-step loop: the inpainting / reconstruction
loop `unmask_on_batch_enhanced` (/root/reference/src/decoder_transformer.py:833-1046),
`calculate_psnr` (:815-830) and the linear-probe feature extractor
(/root/reference/examples/linear_probe.py:195-288).  All arithmetic of the loop runs in
libcora_hip.so (`cora_unmask`); masks, the loss and pooling are plain host NumPy."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .model import TransformerDecoder, _check_x, _fptr
from .optim import Optim, lr_at


def calculate_psnr(img1, img2, data_range: float = 1.0, epsilon: float = 1e-8) -> float:
    """Peak signal-to-noise ratio with the reference's epsilon floor on the MSE (:815-830)."""
    mse = float(np.mean((np.asarray(img1, dtype=np.float64) - np.asarray(img2, dtype=np.float64)) ** 2))
    return float(10.0 * np.log10((data_range ** 2) / max(mse, epsilon)))


def make_mask(model: TransformerDecoder, x_batch: np.ndarray, corrupt_ratio: float,
              rng: Optional[np.random.Generator] = None) -> Tuple[np.ndarray, np.ndarray]:
    """The reference's two corruptions (:868-896): returns (mask [B,1,H,W] with 1 = masked, x_init).
    Lower-half: the bottom floor(H * ratio) rows become grey 0.5.  Random: Bernoulli(ratio) pixels
    become 0.1 * N(0,1) noise (NumPy RNG here: JAX's threefry stream is not reproduced)."""
    B, Cc, H, W = x_batch.shape
    if model.config.use_lower_half_mask:
        ratio = min(max(float(corrupt_ratio), 0.0), 1.0)
        rows = int(np.floor(H * ratio))
        mask = np.zeros((B, 1, H, W), dtype=np.float32)
        if rows > 0:
            mask[:, :, H - rows:, :] = 1.0
        x_c = np.where(mask == 1, np.float32(0.5), x_batch)
    else:
        rng = rng if rng is not None else np.random.default_rng(0)
        mask = (rng.random((B, 1, H, W)) < corrupt_ratio).astype(np.float32)
        noise = (rng.standard_normal(x_batch.shape) * 0.1).astype(np.float32)
        x_c = np.where(mask == 1, noise, x_batch)
    return mask, np.ascontiguousarray(x_c, dtype=np.float32)


def _run_unmask(model: TransformerDecoder, optim_h: Optim, x_true, x_init, mask, use_corruption: bool,
                T: int, want_recons: bool, want_norms: bool, normalise: bool = False,
                optim_w: Optional[Optim] = None, w_scale_by: float = 1.0):
    lib = _lib.load()
    B = x_true.shape[0]
    if model.config.use_status_init_in_unmasking:
        model.init_vodes(B)           # STATUS.INIT forward (:899-901): fresh h_0, h_l := u_l chained
    elif model._ctx is None:
        raise RuntimeError("vodes are not initialised: run an INIT forward first")
    ctx = model._ensure_ctx(B)
    if T > model.max_T:
        raise ValueError(f"T={T} exceeds the model's max_T={model.max_T}")
    rh = optim_h._recipe()
    # unmask_on_batch_enhanced only scales (:925-933); the linear-probe extractor also normalises
    model._grad_options(True, normalise)
    if optim_w is not None:
        _, rw = model._apply_recipes(optim_w, optim_h)
    else:
        model._apply_recipes_h(optim_h)
    optim_h.clear()
    optim_h.init()                                                            # :840-841
    eta = np.array([lr_at(rh.base.learning_rate, t) for t in range(max(T, 1))], dtype=np.float32)
    shape = (B,) + tuple(model.config.image_shape)
    recons = np.empty((T,) + shape, dtype=np.float32) if (want_recons and T > 0) else None
    xhat = np.empty(shape, dtype=np.float32)
    sens = np.empty(shape, dtype=np.float32)
    norms = np.zeros((T, model.L + 3), dtype=np.float64) if (want_norms and T > 0) else None
    mk = None
    if use_corruption:
        mk = np.ascontiguousarray(np.broadcast_to(mask.reshape((B, 1) + mask.shape[-2:]) if mask.ndim == 4 else mask, shape),
                                  dtype=np.float32)
    head = (ctx, _fptr(x_true), _fptr(x_init), _fptr(mk) if mk is not None else None, int(use_corruption),
            0, int(T), _fptr(eta), C.c_float(float(model.config.inference_clamp_alpha)))
    tail = (_fptr(recons) if recons is not None else None, _fptr(xhat), _fptr(sens),
            _fptr(norms) if norms is not None else None)
    if optim_w is not None and T > 0:     # update_weights_during_unmasking, :967-979
        eta_w = np.array([lr_at(rw.base.learning_rate, optim_w.count + t) for t in range(T)], dtype=np.float32)
        _lib.check(lib.cora_unmask_learn(*head, _fptr(eta_w), C.c_float(w_scale_by), *tail), ctx)
        optim_w.count += T
    else:
        _lib.check(lib.cora_unmask(*head, *tail), ctx)
    optim_h.clear()                                                           # :999
    model._x_host = sens.copy()      # the sensory vode keeps its final (blended) value, re-frozen :1037
    return recons, xhat, norms


def unmask_on_batch_enhanced(use_corruption: bool, corrupt_ratio: float, target_T_values: List[int], x, *,
                             model: TransformerDecoder, optim_h: Optim, optim_w: Optional[Optim] = None,
                             log_inference_grads: bool = False, rng: Optional[np.random.Generator] = None):
    """Drop-in for the reference function of the same name (:833-1046).  Returns
    (loss, all_reconstructions, mask, inference_grad_norms_log): the masked/unmasked-balanced MSE
    (:1006-1034), the max(target_T_values) per-step status-"forward" reconstructions (:916-919),
    the mask [B,1,H,W] (zeros_like(x) without corruption, :1033-1034) and, if asked, the per-step
    L2 norms of the scaled vode gradients (:936-958).  With `config.update_weights_during_unmasking`
    and an `optim_w`, every vode step is followed by a weight step scaled by 1 / x.shape[0] (:967-979)."""
    model.eval()
    learn = bool(model.config.update_weights_during_unmasking) and optim_w is not None
    x = _check_x(model, x)
    expected_bs = model._B if model._ctx is not None else x.shape[0]
    x_batch = np.repeat(x, expected_bs, axis=0) if x.shape[0] != expected_bs else x   # :857-860
    if x_batch.ndim != 4:
        raise ValueError("unmask_on_batch_enhanced expects [B,C,H,W] images (as the reference does, :862)")
    max_T = max(target_T_values) if target_T_values else 0
    mask, x_init = None, x_batch
    if use_corruption:
        mask, x_init = make_mask(model, x_batch, corrupt_ratio, rng)
    recons, xhat, norms = _run_unmask(model, optim_h, x_batch, x_init, mask, bool(use_corruption), max_T,
                                      True, log_inference_grads, optim_w=optim_w if learn else None,
                                      w_scale_by=1.0 / x.shape[0])
    all_recons = [recons[t] for t in range(max_T)] if recons is not None else []
    xc = np.clip(xhat.astype(np.float64), 0.0, 1.0)
    if use_corruption:
        mb = np.broadcast_to(mask, xhat.shape).astype(np.float64)
        n_masked = mb.sum()
        n_unmasked = float(xhat.size) - n_masked
        d2 = (xc - x_batch) ** 2
        loss = 0.5 * ((d2 * mb).sum() / max(n_masked, 1e-8) + (d2 * (1 - mb)).sum() / max(n_unmasked, 1e-8))
    else:
        loss = float(((xc.reshape(-1) - x_batch.reshape(-1)) ** 2).mean())
        mask = np.zeros_like(x_batch)
    log: Dict[int, Dict[str, float]] = {}
    if log_inference_grads and norms is not None:
        for t in range(max_T):
            log[t] = {f"vode_{i}_grad_norm": float(norms[t, i]) for i in range(model.L + 3)}
    return float(loss), all_recons, mask, log


def extract_features_from_batch(x_batch, model: TransformerDecoder, optim_h: Optim, inference_steps_for_extraction: int,
                                target_vode_indices_list: Sequence[int], use_gap_flag: bool) -> np.ndarray:
    """Linear-probe features (/root/reference/examples/linear_probe.py:195-288): set the sensory vode
    to the batch (INIT-chained if `use_status_init_in_unmasking`), run T inference steps with the
    sensory vode frozen, then concatenate the chosen vodes' h on the last axis and, if asked,
    average over the token axis."""
    x = _check_x(model, x_batch)
    _run_unmask(model, optim_h, x, x, None, False, int(inference_steps_for_extraction), False, False,
                normalise=True)                              # examples/linear_probe.py:223-232: scaling AND normalisation
    n_vodes = model.L + 3
    feats = []
    for idx in target_vode_indices_list:
        actual = idx + n_vodes if idx < 0 else idx
        if not 0 <= actual < n_vodes:
            raise ValueError(f"Invalid Vode index {idx} (resolved to {actual}). Model has {n_vodes} Vodes.")
        feats.append(model.vodes[actual].h.get())
    if not feats:
        raise ValueError("No features were extracted. target_vode_indices_list might be empty.")
    final = np.concatenate(feats, axis=-1) if len(feats) > 1 else feats[0]
    if use_gap_flag and final.ndim == 3:
        final = final.mean(axis=1)
    return final
