"""`save_params` / `load_params` in the pcx .npz layout.

Same file format as /root/reference/pcx/utils/_serialisation.py:25-105: one
`np.savez_compressed` archive keyed by `jtu.keystr(path)` of each selected parameter
(".patch_projection.nn.weight", ".transformer_blocks[0].attention.nn.query_proj.weight",
".vodes[0].h", ...), so checkpoints written by the reference driver
(examples/debug_transformer_wandb.py:1278) load here and vice versa.
`filter` accepts the same spellings: a class or union of classes out of LayerParam / VodeParam, or a
predicate on a parameter object (the reference's `Callable[[Any], bool]`, :28,41-45), which -- as in the
reference, :41-53 -- is asked for EVERY parameter (each LayerParam proxy, each vode's h), so a predicate
may select inside a family (some blocks only, only vodes whose h is set, ...).
"""
from __future__ import annotations

from types import UnionType
from typing import Dict

import numpy as np

from .model import LayerParam, TransformerDecoder, VodeParam
from .params import vode_key


def _wants(filter, cls, sample=None) -> bool:
    if isinstance(filter, UnionType):
        return cls in filter.__args__
    if isinstance(filter, type):
        return issubclass(cls, filter)
    if isinstance(filter, (tuple, list)):
        return cls in filter
    if callable(filter):
        return bool(filter(sample))
    raise TypeError("filter must be LayerParam, VodeParam, a union of them or a predicate")


def _layer_params(model):
    """key -> LayerParam proxy for every weight of the model, in checkpoint order."""
    return {k: LayerParam(model, k, model._shapes[k]) for k in model.param_keys()}


def save_params(model: TransformerDecoder, path: str, filter=LayerParam) -> None:
    data: Dict[str, np.ndarray] = {}
    for l in range(model.L + 3):
        if _wants(filter, VodeParam, model.vodes[l].h):
            v = model.vodes[l].h.get()
            if v is not None:
                data[vode_key(l)] = v
    chosen = [k for k, prox in _layer_params(model).items() if _wants(filter, LayerParam, prox)]
    if chosen:
        weights = model.get_weights()             # one device read for the whole arena
        data.update({k: weights[k] for k in chosen})
    np.savez_compressed(path, **data)


def load_params(model: TransformerDecoder, path: str, filter=LayerParam) -> None:
    path = path if path.endswith(".npz") else f"{path}.npz"
    with np.load(path) as loaded:          # allow_pickle stays False
        weights = {}
        for key, prox in _layer_params(model).items():
            if not _wants(filter, LayerParam, prox):
                continue
            if key not in loaded:
                raise KeyError(f"Parameter '{key}' not found in the file '{path}'.")
            weights[key] = loaded[key]
        if weights:
            model.set_weights(weights)
        for l in range(model.L + 3):
            if not _wants(filter, VodeParam, model.vodes[l].h):
                continue
            key = vode_key(l)
            if key not in loaded:
                raise KeyError(f"Parameter '{key}' not found in the file '{path}'.")
            model.vodes[l].h.set(loaded[key])
