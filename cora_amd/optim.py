"""Optimizer recipe objects for the PC-ViT step.

The reference builds optax transformations and wraps them in pcx `Optim`
(/root/reference/examples/debug_transformer_wandb.py:1116-1167,
/root/reference/pcx/utils/_optim.py:32-169).  The HIP step implements exactly the recipes used
there -- `chain(clip_by_global_norm(c), sgd(lr, momentum))` for the vodes and
`chain(clip_by_global_norm(c), adamw(lr, weight_decay=...))` for the weights -- so these classes
only DESCRIBE a recipe (same constructor names and argument meaning as optax 0.2.3); the
arithmetic runs in the fused vode-update / AdamW kernels.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Optional, Sequence, Union

Schedule = Union[float, Callable[[int], float]]


@dataclass
class clip_by_global_norm:
    max_norm: float


@dataclass
class sgd:
    learning_rate: Schedule
    momentum: Optional[float] = None
    nesterov: bool = False

    def __post_init__(self):
        if self.nesterov:
            raise NotImplementedError("nesterov momentum is not used by the reference recipe")


@dataclass
class adamw:
    learning_rate: Schedule
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-8
    weight_decay: float = 1e-4


class chain:
    def __init__(self, *transforms):
        self.transforms = tuple(transforms)
        clips = [t for t in transforms if isinstance(t, clip_by_global_norm)]
        bases = [t for t in transforms if isinstance(t, (sgd, adamw))]
        if len(clips) > 1 or len(bases) != 1 or len(clips) + len(bases) != len(transforms):
            raise ValueError("supported recipes: chain([clip_by_global_norm], sgd | adamw)")
        if clips and transforms[0] is not clips[0]:
            raise ValueError("clip_by_global_norm must precede the optimizer (reference order)")
        self.clip = clips[0].max_norm if clips else None
        self.base = bases[0]


def _as_chain(t) -> chain:
    return t if isinstance(t, chain) else chain(t)


def lr_at(lr: Schedule, count: int) -> float:
    return float(lr(count)) if callable(lr) else float(lr)


def create_learning_rate_schedule(base_lr, warmup_steps, total_steps, min_lr_factor):
    """Linear warm-up then cosine decay to base_lr*min_lr_factor; same formula as
    /root/reference/examples/debug_transformer_wandb.py:157-197."""
    min_lr = base_lr * min_lr_factor

    def lr_schedule(step):
        step = float(step)
        duration = max(float(total_steps - warmup_steps), 1e-8)
        ratio = min(max((step - warmup_steps) / duration, 0.0), 1.0)
        decay_lr = min_lr + (base_lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * ratio))
        if warmup_steps > 0:
            warm_lr = base_lr * min(max(step / float(warmup_steps), 0.0), 1.0)
            return warm_lr if step < warmup_steps else decay_lr
        return decay_lr

    return lr_schedule


class Optim:
    """Shape of pcx.utils.Optim (/root/reference/pcx/utils/_optim.py:32-169): built from a
    zero-argument constructor of the transformation; `init`/`clear` bracket the state's lifetime.

    The optimizer STATE lives on the device inside the model's context (vode momentum is re-zeroed
    by every train_on_batch call, :414; AdamW moments persist across calls, _optim.py:47,128).
    This object keeps the recipe and the schedule counter."""

    def __init__(self, optax_opt: Callable[[], object], parameters=None):
        self.optax_opt_fn = optax_opt
        self.recipe: Optional[chain] = None
        self.count = 0
        if parameters is not None:
            self.init(parameters)

    def init(self, parameters=None) -> None:
        """(Re-)initialise the optimizer state (pcx/utils/_optim.py:150-163): the schedule count restarts
        and, for an AdamW recipe bound to a model, the moments and the bias-correction count on the
        device are zeroed with it -- in the reference mu, nu and count are one optax state."""
        self.recipe = _as_chain(self.optax_opt_fn())
        self.count = 0
        if parameters is not None and isinstance(self.recipe.base, adamw) and hasattr(parameters, "_reset_adam_state"):
            parameters._reset_adam_state()

    def clear(self) -> None:
        self.recipe = None
        self.count = 0

    def _recipe(self) -> chain:
        return self.recipe if self.recipe is not None else _as_chain(self.optax_opt_fn())

    def step(self, *a, **k):
        raise NotImplementedError(
            "Optim.step is fused into the HIP step (train_on_batch); it cannot be called on its own")


def make_optimizers(cfg, model=None, total_train_steps: int = 1):
    """The reference driver's optimizer construction for a ModelConfig
    (/root/reference/examples/debug_transformer_wandb.py:1091-1167): returns (optim_w, optim_h)."""
    if cfg.use_lr_schedule_w:
        w_lr = create_learning_rate_schedule(cfg.peak_lr_weights, cfg.warmup_steps, total_train_steps,
                                             cfg.lr_schedule_min_lr_factor)
    else:
        w_lr = cfg.peak_lr_weights
    if cfg.use_lr_schedule_h:
        h_lr = create_learning_rate_schedule(cfg.peak_lr_hidden, cfg.warmup_steps, total_train_steps,
                                             cfg.lr_schedule_min_lr_factor)
    else:
        h_lr = cfg.peak_lr_hidden
    if cfg.use_adamw_for_hidden_optimizer:
        raise NotImplementedError("AdamW for the vodes is a later scope row (SURVEY 8(f) rank 4)")
    h_steps, w_steps = [], []
    if cfg.h_grad_clip_norm is not None and cfg.h_grad_clip_norm > 0:
        h_steps.append(clip_by_global_norm(cfg.h_grad_clip_norm))
    h_steps.append(sgd(h_lr, momentum=cfg.hidden_momentum))
    if cfg.w_grad_clip_norm is not None and cfg.w_grad_clip_norm > 0:
        w_steps.append(clip_by_global_norm(cfg.w_grad_clip_norm))
    w_steps.append(adamw(w_lr, weight_decay=cfg.weight_decay))
    optim_h = Optim(lambda: chain(*h_steps))
    optim_w = Optim(lambda: chain(*w_steps), parameters=model)
    return optim_w, optim_h
