"""GPU parity: the HIP path (through the C-ABI) against the fp64 oracle on identical seeded inputs.

Tolerance (BASELINE.json north_star): fp32 path, rel <= 1e-4 on final vodes, per-step energies and
weight gradients.  `relerr` is max|a-b| / max|b| per tensor."""
import zlib

import numpy as np
import pytest

from oracle import pcvit_ref as R
from tests.util import build_model, make_case, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-4

CASES = {
    # name: (image_shape, patch, D, H, L, B, T)
    "0block": ((3, 16, 16), 4, 32, 1, 0, 4, 3),
    "tiny_1block_h1": ((3, 16, 16), 4, 32, 1, 1, 4, 3),
    "tiny_2block_h2": ((3, 16, 16), 4, 32, 2, 2, 4, 3),
    "cfg2_shape_B4": ((3, 32, 32), 4, 64, 1, 6, 4, 5),
    "cfg3_shape_B2": ((3, 32, 32), 4, 128, 4, 12, 2, 3),
    "d96_h2": ((3, 32, 32), 4, 96, 2, 2, 2, 3),
    "cfg4_shape_B1": ((3, 224, 224), 16, 384, 8, 2, 4, 2),
    "cfg5_video_small": ((4, 3, 32, 32), 4, 64, 1, 2, 1, 2),
}


def run_case(name, l1=0.0, l2=0.0, clip_h=2000.0, clip_w=500.0):
    import cora_amd as ca
    shape, p, D, H, L, B, T = CASES[name]
    case = make_case(shape, p, D, H, L, B, seed=zlib.crc32(name.encode()) % 1000, l1=l1, l2=l2, clip_h=clip_h, clip_w=clip_w)
    model, optim_w, optim_h = build_model(case)
    eta = [0.095] * T
    ref = R.train_on_batch(T, case["x"], case["W"], case["h"], case["pos"], case["dims"], case["hp"], eta, 1e-3)
    Eh, Ew, gh, gw, mse = ca.train_on_batch(T, case["x"], model=model, optim_w=optim_w, optim_h=optim_h)
    tot, per = ca.step_energies(model, T)
    errs = {}
    errs["energies"] = relerr(tot, ref["energies"])
    errs["layer_energies"] = relerr(per, ref["layer_energies"])
    errs["final_h_energy"] = abs(Eh - ref["energies"][-1]) / abs(ref["energies"][-1])
    errs["final_w_energy"] = abs(Ew - ref["final_w_energy"]) / abs(ref["final_w_energy"])
    for l in range(L + 2):
        errs[f"h{l}"] = relerr(model.vodes[l].h.get(), ref["h"][l])
        errs[f"gh{l}"] = relerr(gh[l], ref["final_h_grad"][l])
    gnorm = max(np.abs(v).max() for v in ref["w_grad"].values())
    for k, v in ref["w_grad"].items():
        errs["gw" + k] = float(np.abs(gw[k].astype(np.float64) - v).max() / gnorm)
    neww = model.get_weights()
    for k, v in ref["weights"].items():
        errs["w" + k] = relerr(neww[k], v)
    errs["train_mse"] = abs(mse - ref["train_mse"]) / abs(ref["train_mse"])
    model.close()
    bad = {k: v for k, v in errs.items() if not (v <= TOL)}
    assert not bad, f"{name}: parity failures (rel > {TOL}): {bad}"
    return errs


@pytest.mark.parametrize("name", list(CASES))
def test_step_parity(name):
    run_case(name)


def test_regularisers_and_clipping():
    # l1/l2 vode regularisers on, and clip thresholds low enough that both clips trigger
    run_case("tiny_2block_h2", l1=1e-3, l2=1e-3, clip_h=5.0, clip_w=0.5)


def test_two_batches_persistent_state_and_adam():
    """vodes persist across batches, AdamW moments and count persist (reference _optim.py:47,128)."""
    import cora_amd as ca
    shape, p, D, H, L, B, T = CASES["tiny_2block_h2"]
    case = make_case(shape, p, D, H, L, B, seed=5)
    model, optim_w, optim_h = build_model(case)
    eta = [0.095] * T
    rng = np.random.default_rng(9)
    x2 = rng.uniform(0, 1, case["x"].shape).astype(np.float32)
    r1 = R.train_on_batch(T, case["x"], case["W"], case["h"], case["pos"], case["dims"], case["hp"], eta, 1e-3)
    r2 = R.train_on_batch(T, x2, r1["weights"], r1["h"], case["pos"], case["dims"], case["hp"], eta, 1e-3, adam=r1["adam"])
    ca.train_on_batch(T, case["x"], model=model, optim_w=optim_w, optim_h=optim_h)
    Eh, Ew, gh, gw, mse = ca.train_on_batch(T, x2, model=model, optim_w=optim_w, optim_h=optim_h)
    assert abs(Ew - r2["final_w_energy"]) / abs(r2["final_w_energy"]) < TOL
    neww = model.get_weights()
    for k, v in r2["weights"].items():
        assert relerr(neww[k], v) < TOL, k
    for l in range(L + 2):
        assert relerr(model.vodes[l].h.get(), r2["h"][l]) < TOL
    assert abs(mse - r2["train_mse"]) / r2["train_mse"] < TOL
    model.close()


def test_init_forward_energy_api():
    """INIT forward (h_l := u_l chained), status-None energy() and status-"forward" reconstruction."""
    import cora_amd as ca
    shape, p, D, H, L, B, T = CASES["tiny_2block_h2"]
    case = make_case(shape, p, D, H, L, B, seed=11)
    model, _, _ = build_model(case)
    model.init_vodes(B, h0=case["h0"])
    href = R.init_vodes(case["h0"], case["W"], case["pos"], case["dims"])
    for l in range(L + 2):
        assert relerr(model.vodes[l].h.get(), href[l]) < TOL
    for l, t in enumerate(case["h"]):
        model.vodes[l].h.set(t)
    with ca.step(model, None):
        ca.forward(case["x"], model=model)
    E, xhat = ca.energy(model=model)
    Eref, per_ref, xhat_ref = R.energy_and_prediction(case["h"], case["W"], case["pos"], case["x"], case["dims"], case["hp"])
    assert abs(E - Eref) / Eref < TOL
    assert relerr(xhat, xhat_ref) < TOL
    assert relerr(ca.layer_energies(model), per_ref) < TOL
    with ca.step(model, ca.STATUS_FORWARD):
        rec = ca.forward(None, model=model)
    import torch
    Wt = {k: torch.as_tensor(v, dtype=torch.float64) for k, v in case["W"].items()}
    ht = [torch.as_tensor(t, dtype=torch.float64) for t in case["h"]]
    rec_ref = R.predictions(ht, Wt, torch.as_tensor(case["pos"], dtype=torch.float64), case["dims"], chain=True)[-1].numpy()
    assert relerr(rec, rec_ref) < TOL
    model.close()
