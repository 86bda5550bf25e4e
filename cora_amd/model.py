"""Host-side mirror of the reference's PC-ViT API for the predictive-coding step.

  TransformerDecoder ....... /root/reference/src/decoder_transformer.py:128-256
  forward / energy ......... /root/reference/src/decoder_transformer.py:297-305
  train_on_batch ........... /root/reference/src/decoder_transformer.py:392-499
  step() context manager ... /root/reference/pcx/utils/_misc.py:17-71
  STATUS ................... /root/reference/pcx/predictive_coding/_vode.py:34

The objects here hold no arithmetic: vodes, weights and optimizer state live on the GPU inside a
`cora_ctx` (include/cora_hip.h) and every compute call goes through the ctypes C-ABI.  There is no
CPU fallback -- without libcora_hip.so the constructor raises.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import Dict, List, Optional

import numpy as np

from . import _lib
from .config import TransformerConfig
from .optim import Optim, adamw, lr_at, sgd
from .params import (BLOCK_LEAVES, K_BOUT, K_BP, K_WOUT, K_WP, block_key, init_params, param_shapes, vln_key,
                     vode_key)
from .posenc import create_positional_encoding

STATUS_FORWARD = "forward"


class STATUS:
    NONE = None
    INIT = "init"


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class _Param:
    """`.get()` / `.set()` / `.shape` like a pcx Param."""
    frozen = False

    def get(self):
        raise NotImplementedError

    def set(self, value):
        raise NotImplementedError

    @property
    def value(self):
        return self.get()

    @property
    def shape(self):
        return self.get().shape


class LayerParam(_Param):
    def __init__(self, model: "TransformerDecoder", key: str, shape):
        self._m, self.key, self._shape = model, key, tuple(shape)

    @property
    def shape(self):
        return self._shape

    def get(self) -> np.ndarray:
        return self._m._get_weight(self.key).reshape(self._shape)

    def set(self, value) -> None:
        v = np.ascontiguousarray(value, dtype=np.float32)
        if v.shape != self._shape:
            raise ValueError(f"{self.key}: expected shape {self._shape}, got {v.shape}")
        self._m._set_weight(self.key, v)


class VodeParam(_Param):
    def __init__(self, model: "TransformerDecoder", l: int):
        self._m, self.l = model, l
        self.frozen = False

    def get(self) -> Optional[np.ndarray]:
        return self._m._get_vode(self.l)

    def set(self, value) -> None:
        self._m._set_vode(self.l, value)


class Vode:
    def __init__(self, model, l):
        self.h = VodeParam(model, l)
        self.status = None


class _NN:
    def __init__(self, **params):
        self.__dict__.update(params)


class _Layer:
    def __init__(self, nn: _NN):
        self.nn = nn


class _Attention(_Layer):
    pass


class TransformerBlock:
    def __init__(self, model, i, D, M):
        P = lambda leaf, shp: LayerParam(model, block_key(i, leaf), shp)
        self.layer_norm1 = _Layer(_NN(weight=P("layer_norm1.nn.weight", (D,)), bias=P("layer_norm1.nn.bias", (D,))))
        self.layer_norm2 = _Layer(_NN(weight=P("layer_norm2.nn.weight", (D,)), bias=P("layer_norm2.nn.bias", (D,))))
        proj = lambda name: _NN(weight=P(f"attention.nn.{name}.weight", (D, D)), bias=None)
        self.attention = _Attention(_NN(query_proj=proj("query_proj"), key_proj=proj("key_proj"),
                                        value_proj=proj("value_proj"), output_proj=proj("output_proj")))
        self.linear1 = _Layer(_NN(weight=P("linear1.nn.weight", (M, D)), bias=P("linear1.nn.bias", (M,))))
        self.linear2 = _Layer(_NN(weight=P("linear2.nn.weight", (D, M)), bias=P("linear2.nn.bias", (D,))))


class GradView:
    """Lazy view of a gradient pytree left on the device (final_h_grad / final_w_grad)."""

    def __init__(self, fetch, keys):
        self._fetch, self._keys = fetch, list(keys)

    def keys(self):
        return list(self._keys)

    def __getitem__(self, key):
        return self._fetch(key)

    def items(self):
        return [(k, self._fetch(k)) for k in self._keys]


class TransformerDecoder:
    """Same constructor and attribute layout as the reference model; state lives on the GPU."""

    def __init__(self, config: TransformerConfig, *, seed: int = 42, device: int = 0, max_T: int = 64,
                 compute_dtype: str = "fp32"):
        _lib.load()                      # fail loudly if the HIP library is missing
        if compute_dtype not in ("fp32", "bf16", "bf16x3"):
            raise ValueError("compute_dtype must be 'fp32' (parity path), 'bf16' (opt-in: bf16 MFMA operands) or "
                             "'bf16x3' (opt-in: three-term bf16 split in the GEMMs, fp32-grade products)")
        self._dtype_mode = {"fp32": 0, "bf16": 1, "bf16x3": 2}[compute_dtype]
        self.config = config
        self.device = device
        self.max_T = max_T
        self.status = None
        self.training = True
        c = config
        if len(c.image_shape) == 4:
            self._F, self._C, self._Himg, self._Wimg = c.image_shape
        else:
            self._F = 1
            self._C, self._Himg, self._Wimg = c.image_shape
        self.N, self.P, self.D, self.M, self.L, self.H = (c.num_patches, c.patch_dim, c.hidden_size,
                                                          c.mlp_hidden_dim, c.num_blocks, c.num_heads)
        vln = bool(c.use_vode_state_layernorm)
        self._shapes = dict(param_shapes(self.P, self.D, self.M, self.L, vln))
        self._host_w: Dict[str, np.ndarray] = init_params(self.P, self.D, self.M, self.L, seed, vln)
        self._rng = np.random.default_rng(seed + 1)
        self.positional_embedding = create_positional_encoding(
            image_shape=c.image_shape, hidden_size=c.hidden_size, patch_size=c.patch_size,
            is_video=c.is_video, num_frames=c.num_frames if c.is_video else None, theta=c.theta)
        self.vodes: List[Vode] = [Vode(self, l) for l in range(self.L + 3)]
        self.vodes[-1].h.frozen = True
        self.transformer_blocks = [TransformerBlock(self, i, self.D, self.M) for i in range(self.L)]
        LP = lambda k: LayerParam(self, k, self._shapes[k])
        self.patch_projection = _Layer(_NN(weight=LP(K_WP), bias=LP(K_BP)))
        self.output_projection = _Layer(_NN(weight=LP(K_WOUT), bias=LP(K_BOUT)))
        # LayerNorms on what vodes 1..L+1 hand on (reference :212-218); per-op kernels only
        self.vode_output_layernorms: list = (
            [_Layer(_NN(weight=LP(vln_key(i, "weight")), bias=LP(vln_key(i, "bias")))) for i in range(self.L + 1)]
            if vln else [])
        self._ctx = None
        self._B = None
        self._offsets: Dict[str, tuple] = {}
        self._optim_sig = None
        self._x_host: Optional[np.ndarray] = None
        self._dp = None

    # -- mode switches (pcx Module.train/eval) -----------------------------------------------------
    def train(self):
        self.training = True
        return self

    def eval(self):
        self.training = False
        return self

    # -- context ------------------------------------------------------------------------------------
    def _make_cfg(self, B: int) -> _lib.CoraConfig:
        c = self.config
        return _lib.CoraConfig(
            N=self.N, P=self.P, D=self.D, H=self.H, M=self.M, L=self.L, B=B, C=self._C,
            patch=c.patch_size, Himg=self._Himg, Wimg=self._Wimg, F=self._F, dtype_mode=self._dtype_mode,
            use_scaling=int(bool(c.use_inference_lr_scaling)),
            scale_base=float(c.inference_lr_scale_base if c.inference_lr_scale_base is not None else 1.0),
            l1=float(c.intermediate_l1_coeff), l2=float(c.intermediate_l2_coeff),
            clip_h=0.0, clip_w=0.0, momentum=0.0, adam_b1=0.9, adam_b2=0.999, adam_eps=1e-8,
            adam_wd=0.0, use_grad_norm=int(bool(c.use_vode_grad_norm)),
            grad_norm_target=float(c.vode_grad_norm_target), max_T=self.max_T,
            use_vode_ln=int(bool(c.use_vode_state_layernorm)))

    def _ensure_ctx(self, B: int):
        if self._ctx is not None:
            if B != self._B:
                raise ValueError(f"batch size changed from {self._B} to {B}: vodes are persistent [B,...] "
                                 "state (SURVEY 3.3); use drop_last or build a second model")
            return self._ctx
        lib = _lib.load()
        cfg = self._make_cfg(B)
        ctx = C.c_void_p()
        _lib.check(lib.cora_create(C.byref(cfg), self.device, C.byref(ctx)), None)
        self._ctx, self._B = ctx, B
        n = C.c_size_t()
        _lib.check(lib.cora_weights_size(ctx, C.byref(n)), ctx)
        self._nW = n.value
        flat = np.zeros(self._nW, dtype=np.float32)
        for key, arr in self._host_w.items():
            off, cnt = self._offset(key)
            flat[off:off + cnt] = arr.reshape(-1)
        _lib.check(lib.cora_set_weights(ctx, _fptr(flat), flat.size), ctx)
        pos = np.ascontiguousarray(self.positional_embedding, dtype=np.float32)
        _lib.check(lib.cora_set_pos(ctx, _fptr(pos), pos.size), ctx)
        self._host_w = {}
        return ctx

    def close(self):
        if self._ctx is not None:
            _lib.load().cora_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _offset(self, key: str):
        if key not in self._offsets:
            off, cnt = C.c_size_t(), C.c_size_t()
            _lib.check(_lib.load().cora_weight_offset(self._ctx, key.encode(), C.byref(off), C.byref(cnt)), self._ctx)
            self._offsets[key] = (off.value, cnt.value)
        return self._offsets[key]

    # -- weights ------------------------------------------------------------------------------------
    def param_keys(self) -> List[str]:
        return list(self._shapes.keys())

    def _all_weights(self) -> np.ndarray:
        flat = np.empty(self._nW, dtype=np.float32)
        _lib.check(_lib.load().cora_get_weights(self._ctx, _fptr(flat), flat.size), self._ctx)
        return flat

    def _get_weight(self, key: str) -> np.ndarray:
        if self._ctx is None:
            return self._host_w[key].copy()
        off, cnt = self._offset(key)
        return self._all_weights()[off:off + cnt].copy()

    def _set_weight(self, key: str, v: np.ndarray) -> None:
        if self._ctx is None:
            self._host_w[key] = v.copy()
            return
        flat = self._all_weights()
        off, cnt = self._offset(key)
        flat[off:off + cnt] = v.reshape(-1)
        _lib.check(_lib.load().cora_set_weights(self._ctx, _fptr(flat), flat.size), self._ctx)

    def get_weights(self) -> Dict[str, np.ndarray]:
        if self._ctx is None:
            return {k: v.copy() for k, v in self._host_w.items()}
        flat = self._all_weights()
        out = {}
        for k, shp in self._shapes.items():
            off, cnt = self._offset(k)
            out[k] = flat[off:off + cnt].reshape(shp).copy()
        return out

    def set_weights(self, weights: Dict[str, np.ndarray]) -> None:
        for k, v in weights.items():
            if k not in self._shapes:
                raise KeyError(f"unknown parameter '{k}'")
            if tuple(np.shape(v)) != self._shapes[k]:
                raise ValueError(f"{k}: expected shape {self._shapes[k]}, got {np.shape(v)}")
        if self._ctx is None:
            for k, v in weights.items():
                self._host_w[k] = np.ascontiguousarray(v, dtype=np.float32)
            return
        flat = self._all_weights()
        for k, v in weights.items():
            off, cnt = self._offset(k)
            flat[off:off + cnt] = np.asarray(v, dtype=np.float32).reshape(-1)
        _lib.check(_lib.load().cora_set_weights(self._ctx, _fptr(flat), flat.size), self._ctx)

    # -- vodes --------------------------------------------------------------------------------------
    def _vode_shape(self, l: int):
        if l == 0:
            return (self._B, self.N, self.P)
        if l <= self.L + 1:
            return (self._B, self.N, self.D)
        return (self._B,) + tuple(self.config.image_shape)

    def _get_vode(self, l: int) -> Optional[np.ndarray]:
        if self._ctx is None:
            return None
        if l == self.L + 2:
            return None if self._x_host is None else self._x_host.copy()
        out = np.empty(self._vode_shape(l), dtype=np.float32)
        _lib.check(_lib.load().cora_get_vode(self._ctx, l, _fptr(out), out.size), self._ctx)
        return out

    def _set_vode(self, l: int, value) -> None:
        v = np.ascontiguousarray(value, dtype=np.float32)
        self._ensure_ctx(v.shape[0])
        if l == self.L + 2:
            self._x_host = v.reshape(self._vode_shape(l)).copy()
            return
        if v.shape != self._vode_shape(l):
            raise ValueError(f"vode {l}: expected shape {self._vode_shape(l)}, got {v.shape}")
        _lib.check(_lib.load().cora_set_vode(self._ctx, l, _fptr(v), v.size), self._ctx)

    def init_vodes(self, batch_size: int, h0: Optional[np.ndarray] = None) -> None:
        """INIT pass: h_0 = 0.01*N(0,1) if use_noise else zeros (reference :145-152), h_l := u_l."""
        ctx = self._ensure_ctx(batch_size)
        if h0 is None and self.config.use_noise:
            h0 = (0.01 * self._rng.standard_normal((batch_size, self.N, self.P))).astype(np.float32)
        ptr = None
        if h0 is not None:
            h0 = np.ascontiguousarray(h0, dtype=np.float32)
            if h0.shape != (batch_size, self.N, self.P):
                raise ValueError("h0 shape mismatch")
            ptr = _fptr(h0)
        _lib.check(_lib.load().cora_init_vodes(ctx, ptr), ctx)

    # -- optimizer recipe -> device -------------------------------------------------------------------
    def _apply_recipes(self, optim_w: Optim, optim_h: Optim):
        rh, rw = optim_h._recipe(), optim_w._recipe()
        if not isinstance(rh.base, sgd):
            raise NotImplementedError("vode optimizer must be sgd(lr, momentum) (AdamW-for-h: SURVEY 8(f) rank 4)")
        if not isinstance(rw.base, adamw):
            raise NotImplementedError("weight optimizer must be adamw")
        sig = (float(rh.clip or 0.0), float(rh.base.momentum or 0.0), float(rw.clip or 0.0),
               float(rw.base.b1), float(rw.base.b2), float(rw.base.eps), float(rw.base.weight_decay))
        if sig != self._optim_sig:
            _lib.check(_lib.load().cora_set_optim(self._ctx, *[C.c_float(v) for v in sig]), self._ctx)
            self._optim_sig = sig
        return rh, rw

    def _grad_options(self, scaling: bool, normalise: Optional[bool] = None):
        """Which gradient conditioning the next loop applies, out of what the config enables.
        MODE 2 of train_on_batch (reference :449-460) and the linear-probe extractor
        (examples/linear_probe.py:223-232): layer scaling and per-leaf normalisation.
        unmask_on_batch_enhanced (:925-933): layer scaling only.  MODE 1 (:421-424): neither."""
        c = self.config
        normalise = scaling if normalise is None else normalise
        _lib.check(_lib.load().cora_set_grad_options(
            self._ctx, int(bool(c.use_inference_lr_scaling) and scaling),
            C.c_float(float(c.inference_lr_scale_base if c.inference_lr_scale_base is not None else 1.0)),
            int(bool(c.use_vode_grad_norm) and normalise), C.c_float(float(c.vode_grad_norm_target))), self._ctx)

    def _apply_recipes_h(self, optim_h: Optim):
        """Vode-optimizer half only (inference-only loops); the weight half keeps its last values."""
        rh = optim_h._recipe()
        if not isinstance(rh.base, sgd):
            raise NotImplementedError("vode optimizer must be sgd(lr, momentum) (AdamW-for-h: SURVEY 8(f) rank 4)")
        prev = self._optim_sig or (0.0, 0.0, 0.0, 0.9, 0.999, 1e-8, 0.0)
        sig = (float(rh.clip or 0.0), float(rh.base.momentum or 0.0)) + tuple(prev[2:])
        if sig != self._optim_sig:
            _lib.check(_lib.load().cora_set_optim(self._ctx, *[C.c_float(v) for v in sig]), self._ctx)
            self._optim_sig = sig
        return rh

    def kernel_path(self) -> str:
        """"fused" / "fused-wide" (one workgroup per (block, image)), "fused-rows" (row-tile kernels around the attention
        kernels) or "per-op" -- which kernels the step runs."""
        return {1: "fused", 2: "fused-wide", 3: "fused-rows"}.get(_lib.load().cora_kernel_path(self._ctx), "per-op")

    # -- adam state (optimizer state is not part of the reference's checkpoints, but is exposed) ----
    def _reset_adam_state(self):
        """optim_w.init(model): AdamW moments and count back to zero (no-op before the ctx exists: a
        fresh ctx starts from zeros)."""
        if self._ctx is None:
            return
        z = np.zeros(self._nW, dtype=np.float32)
        _lib.check(_lib.load().cora_set_adam_state(self._ctx, _fptr(z), _fptr(z), 0), self._ctx)

    def get_adam_state(self):
        mu = np.empty(self._nW, dtype=np.float32)
        nu = np.empty(self._nW, dtype=np.float32)
        cnt = C.c_int64()
        _lib.check(_lib.load().cora_get_adam_state(self._ctx, _fptr(mu), _fptr(nu), C.byref(cnt)), self._ctx)
        return mu, nu, cnt.value


# ---------------------------------------------------------------------------------------------------
@contextlib.contextmanager
def step(model: TransformerDecoder, status=None, clear_params=None):
    """pcx.utils.step: set the status on the model and its vodes for the duration of the block."""
    prev = model.status
    model.status = status
    for v in model.vodes:
        v.status = status
    try:
        yield
    finally:
        model.status = prev if status is not None else None
        for v in model.vodes:
            v.status = model.status


def _check_x(model: TransformerDecoder, x) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    if tuple(x.shape[1:]) != tuple(model.config.image_shape):
        raise ValueError(f"x has shape {x.shape}, expected [B, {tuple(model.config.image_shape)}]")
    return x


def forward(x, *, model: TransformerDecoder):
    """`forward(x, model=model)`: behaviour follows the model status set by `step(...)`.
    INIT: initialise every vode by a chained forward and set the sensory vode to x.
    None: set the sensory vode to x, return the status-None prediction (block i reads h_{i+1}).
    "forward": chained reconstruction from h_0 (rule 'h -> u')."""
    lib = _lib.load()
    if x is not None:
        x = _check_x(model, x)
    if model.status == STATUS.INIT:
        if x is None:
            raise ValueError("INIT forward needs a batch to size the vodes")
        model.init_vodes(x.shape[0])
        model._x_host = x.copy()
    if model._ctx is None:
        raise RuntimeError("vodes are not initialised: run an INIT forward first")
    n = int(np.prod(model._vode_shape(model.L + 2)))
    out = np.empty(model._vode_shape(model.L + 2), dtype=np.float32)
    if model.status == STATUS_FORWARD:
        _lib.check(lib.cora_forward(model._ctx, _fptr(out), n), model._ctx)
        return out
    if x is not None:
        model._x_host = x.copy()
    if model._x_host is None:
        raise RuntimeError("the sensory vode has no value yet")
    E = C.c_double()
    _lib.check(lib.cora_energy(model._ctx, _fptr(model._x_host), C.byref(E), None, _fptr(out), n), model._ctx)
    return out


def energy(*, model: TransformerDecoder):
    """`energy(model=model)` -> (batch-summed energy, status-None prediction)."""
    lib = _lib.load()
    if model._ctx is None or model._x_host is None:
        raise RuntimeError("model has no vodes / sensory value yet")
    n = int(np.prod(model._vode_shape(model.L + 2)))
    out = np.empty(model._vode_shape(model.L + 2), dtype=np.float32)
    E = C.c_double()
    _lib.check(lib.cora_energy(model._ctx, _fptr(model._x_host), C.byref(E), None, _fptr(out), n), model._ctx)
    return E.value, out


def layer_energies(model: TransformerDecoder) -> np.ndarray:
    """Per-vode energies [L+3] at the current vodes (what log_vode_stats reads in the reference)."""
    per = np.zeros(model.L + 3, dtype=np.float64)
    E = C.c_double()
    _lib.check(_lib.load().cora_energy(model._ctx, _fptr(model._x_host), C.byref(E), _fptr(per), None, 0), model._ctx)
    return per


def train_on_batch(T: int, x, *, model: TransformerDecoder, optim_w: Optim, optim_h: Optim, epoch=None, step=None):
    """Drop-in for the reference `train_on_batch`: returns
    (final_h_energy, final_w_energy, final_h_grad, final_w_grad, train_mse)."""
    lib = _lib.load()
    model.train()
    x = _check_x(model, x)
    B = x.shape[0]
    if model._ctx is None and not model.config.use_status_init_in_training:
        raise RuntimeError("vodes are not initialised: run `with step(model, STATUS.INIT): forward(x, model=model)` first")
    if model.config.use_status_init_in_training:
        model.init_vodes(B)
    ctx = model._ensure_ctx(B)
    rh, rw = model._apply_recipes(optim_w, optim_h)
    optim_h.init()                                   # count restarts every batch (reference :414)
    eta_h = np.array([lr_at(rh.base.learning_rate, t) for t in range(T)], dtype=np.float32)
    eta_w = lr_at(rw.base.learning_rate, optim_w.count)
    out = _lib.CoraStepOut()
    model._x_host = x.copy()
    model._grad_options(not model.config.update_weights_every_inference_step)
    if model.config.update_weights_every_inference_step:      # MODE 1 (iPC), reference :418-439
        if model._dp is not None and not model._dp.in_library:
            raise NotImplementedError("iPC mode under data parallelism needs DataParallel(backend='rccl'): the dE/dW "
                                      "all-reduce of every inference step runs inside the library (cora_step_ipc_dist)")
        eta_ws = np.array([lr_at(rw.base.learning_rate, optim_w.count + t) for t in range(T)], dtype=np.float32)
        if model._dp is not None:
            _lib.check(lib.cora_step_ipc_dist(ctx, _fptr(x), 0, T, _fptr(eta_h), _fptr(eta_ws),
                                              int(model._dp.exact_clip), C.byref(out)), ctx)
        else:
            _lib.check(lib.cora_step_ipc(ctx, _fptr(x), 0, T, _fptr(eta_h), _fptr(eta_ws), C.byref(out)), ctx)
        optim_w.count += T - 1                               # the common `count += 1` below makes it T
    elif model._dp is None:
        _lib.check(lib.cora_step(ctx, _fptr(x), 0, T, _fptr(eta_h), C.c_float(eta_w), C.byref(out)), ctx)
    elif model._dp.in_library:                                # RCCL inside the captured graphs
        _lib.check(lib.cora_step_dist(ctx, _fptr(x), 0, T, _fptr(eta_h), C.c_float(eta_w),
                                      int(model._dp.exact_clip), C.byref(out)), ctx)
        model._dp.agree_on_graphs(model, T)                   # once per T, after the first (eager) call
    else:
        if model._dp.exact_clip:
            model._dp.infer_exact(model, ctx, x, T, eta_h, out)
        else:
            _lib.check(lib.cora_infer(ctx, _fptr(x), 0, T, _fptr(eta_h), None), ctx)   # ended by cora_wgrad
        _lib.check(lib.cora_wgrad(ctx, C.byref(out)), ctx)
        world = model._dp.allreduce()
        _lib.check(lib.cora_apply_wgrad(ctx, C.c_float(eta_w), C.c_float(1.0 / (B * world)), C.byref(out)), ctx)
    optim_w.count += 1
    optim_h.clear()                                  # reference :496
    model.last_step = out

    def fetch_h(l):
        g = np.empty(model._vode_shape(l), dtype=np.float32)
        _lib.check(lib.cora_get_vode_grad(ctx, l, _fptr(g), g.size), ctx)
        return g

    def fetch_w(key):
        flat = np.empty(model._nW, dtype=np.float32)
        _lib.check(lib.cora_get_wgrad(ctx, _fptr(flat), flat.size), ctx)
        off, cnt = model._offset(key)
        return flat[off:off + cnt].reshape(model._shapes[key]).copy()

    h_grad = GradView(fetch_h, range(model.L + 2))
    w_grad = GradView(fetch_w, model.param_keys())
    return out.final_h_energy, out.final_w_energy, h_grad, w_grad, out.train_mse


def step_energies(model: TransformerDecoder, T: int):
    """Per-step total energies [T] and per-vode energies [T, L+3] of the last train_on_batch."""
    tot = np.zeros(T, dtype=np.float64)
    per = np.zeros((T, model.L + 3), dtype=np.float64)
    _lib.check(_lib.load().cora_get_energies(model._ctx, T, _fptr(tot), _fptr(per)), model._ctx)
    return tot, per
