"""Parameter inventory of the PC-ViT: pcx key strings, shapes and equinox-style initialisation.

Key strings are what `pcx.utils.save_params` writes (jtu.keystr of the path to each LayerParam /
VodeParam: /root/reference/pcx/utils/_serialisation.py:47-60, module flattening
/root/reference/pcx/core/_module.py:72-84).  Layer inventory follows TransformerDecoder.__init__
(/root/reference/src/decoder_transformer.py:179-199) and pxnn.TransformerBlock.__init__
(/root/reference/pcx/nn/_layer.py:410-426); attention projections carry no bias
(/root/reference/pcx/nn/_layer.py:373-376).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

K_WP, K_BP = ".patch_projection.nn.weight", ".patch_projection.nn.bias"
K_WOUT, K_BOUT = ".output_projection.nn.weight", ".output_projection.nn.bias"


def block_key(i: int, leaf: str) -> str:
    return f".transformer_blocks[{i}].{leaf}"


BLOCK_LEAVES = (
    "layer_norm1.nn.weight", "layer_norm1.nn.bias", "layer_norm2.nn.weight", "layer_norm2.nn.bias",
    "attention.nn.query_proj.weight", "attention.nn.key_proj.weight",
    "attention.nn.value_proj.weight", "attention.nn.output_proj.weight",
    "linear1.nn.weight", "linear1.nn.bias", "linear2.nn.weight", "linear2.nn.bias",
)


def vode_key(l: int) -> str:
    return f".vodes[{l}].h"


def vln_key(i: int, leaf: str) -> str:
    """LayerNorm applied to what vode i+1 hands on (use_vode_state_layernorm,
    /root/reference/src/decoder_transformer.py:212-218); leaf = "weight" | "bias"."""
    return f".vode_output_layernorms[{i}].nn.{leaf}"


def param_shapes(P: int, D: int, M: int, L: int, vode_ln: bool = False) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) of every LayerParam, in the model's attribute order."""
    out: List[Tuple[str, Tuple[int, ...]]] = []
    for i in range(L):
        shp = {
            "layer_norm1.nn.weight": (D,), "layer_norm1.nn.bias": (D,),
            "layer_norm2.nn.weight": (D,), "layer_norm2.nn.bias": (D,),
            "attention.nn.query_proj.weight": (D, D), "attention.nn.key_proj.weight": (D, D),
            "attention.nn.value_proj.weight": (D, D), "attention.nn.output_proj.weight": (D, D),
            "linear1.nn.weight": (M, D), "linear1.nn.bias": (M,),
            "linear2.nn.weight": (D, M), "linear2.nn.bias": (D,),
        }
        out += [(block_key(i, leaf), shp[leaf]) for leaf in BLOCK_LEAVES]
    out += [(K_WP, (D, P)), (K_BP, (D,)), (K_WOUT, (P, D)), (K_BOUT, (P,))]
    if vode_ln:
        for i in range(L + 1):
            out += [(vln_key(i, "weight"), (D,)), (vln_key(i, "bias"), (D,))]
    return out


def init_params(P: int, D: int, M: int, L: int, seed: int = 42, vode_ln: bool = False) -> Dict[str, np.ndarray]:
    """equinox 0.11.8 defaults: Linear weight and bias ~ U(-1/sqrt(in), 1/sqrt(in)); LayerNorm
    gamma=1, beta=0.  (The reference draws from the JAX threefry stream of pcx's RKG; that bit
    stream is not reproduced -- weights are inputs of the step)"""
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    shapes = dict(param_shapes(P, D, M, L, vode_ln))
    for key, shp in shapes.items():
        if "layer_norm" in key or "layernorms" in key:
            out[key] = (np.ones if key.endswith("weight") else np.zeros)(shp, dtype=np.float32)
    for key, shp in shapes.items():
        if key.endswith(".weight") and "layer_norm" not in key and "layernorms" not in key:
            lim = 1.0 / np.sqrt(shp[1])
            out[key] = rng.uniform(-lim, lim, size=shp).astype(np.float32)
            bkey = key[:-len("weight")] + "bias"
            if bkey in shapes:
                out[bkey] = rng.uniform(-lim, lim, size=shapes[bkey]).astype(np.float32)
    return {k: out[k] for k in shapes}
