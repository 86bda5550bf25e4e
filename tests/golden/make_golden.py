"""Generates tests/golden/*.npz: seeded inputs + fp64 oracle outputs of the PC-ViT step.

The reference (JAX/equinox/optax) cannot be imported in the build container, so these vectors come
from the CPU restatement oracle/pcvit_ref.py (torch autograd, fp64), cross-checked against the
independent hand-VJP oracle/pcvit_np.py.  They pin the oracle against accidental change and give
the GPU tests a reference that does not depend on re-running the oracle.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pcvit_ref as R  # noqa: E402
from tests.util import make_case  # noqa: E402

GOLDEN = {
    # name: (image_shape, patch, D, H, L, B, T, l1, l2, clip_h, clip_w)
    "g_0block": ((3, 8, 8), 4, 16, 1, 0, 4, 2, 0.0, 0.0, 2000.0, 500.0),
    "g_1block": ((3, 8, 8), 4, 16, 1, 1, 4, 3, 0.0, 0.0, 2000.0, 500.0),
    "g_2block_h2_reg_clip": ((3, 16, 16), 4, 32, 2, 2, 4, 3, 1e-3, 1e-3, 5.0, 0.5),
    "g_video": ((2, 3, 8, 8), 4, 32, 2, 1, 2, 2, 0.0, 0.0, 2000.0, 500.0),
}


def generate(name):
    shape, p, D, H, L, B, T, l1, l2, ch, cw = GOLDEN[name]
    case = make_case(shape, p, D, H, L, B, seed=100 + len(name), l1=l1, l2=l2, clip_h=ch, clip_w=cw)
    eta = [0.095] * T
    out = R.train_on_batch(T, case["x"], case["W"], case["h"], case["pos"], case["dims"], case["hp"], eta, 1e-3)
    data = {"x": case["x"], "pos": case["pos"], "meta": np.array([p, D, H, L, B, T], dtype=np.int64),
            "image_shape": np.array(shape, dtype=np.int64),
            "hyper": np.array([l1, l2, ch, cw, case["hp"].momentum, case["hp"].scale_base]),
            "energies": out["energies"], "layer_energies": out["layer_energies"],
            "final_w_energy": np.array(out["final_w_energy"]), "train_mse": np.array(out["train_mse"])}
    for k, v in case["W"].items():
        data["W" + k] = v
        data["gW" + k] = out["w_grad"][k].astype(np.float32)
    for l, t in enumerate(case["h"]):
        data[f"h_in{l}"] = t
        data[f"h_out{l}"] = out["h"][l].astype(np.float32)
        data[f"gh{l}"] = out["final_h_grad"][l].astype(np.float32)
    return data


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for name in GOLDEN:
        np.savez_compressed(os.path.join(here, name + ".npz"), **generate(name))
        print("wrote", name)
