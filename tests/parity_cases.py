"""Oracle parity at the sizes that are benchmarked (not a test module: helpers shared by
tests/test_gpu_fullsize_oracle.py and scripts/parity_table.py).

Every case builds the inputs the way bench.py does (weights `init_params(seed 42)`, x ~ U[0,1),
h_0 = 0.01 N(0,1), vodes 1..L+1 INIT-chained, the named config's own optimizer recipe incl. its
learning-rate schedules), runs ONE whole `train_on_batch` on the HIP path and on the oracle
(`oracle/pcvit_ref.py`, fp64, plus an fp32 run of the same oracle = the noise floor of the math in
fp32), and returns one row per compared tensor:

    {"err": max|gpu - fp64| / scale, "floor": max|fp32 oracle - fp64| / scale, "bar": the bar applied}

scale = max|fp64| of that tensor (weight gradients: the largest entry over ALL weight gradients).
bar = the PLAIN 1e-4 (BASELINE.json north_star) for everything north_star names -- final vodes, per-step
energies, weight gradients -- and for final_w_energy / train_mse.  Only final_h_grad (gh*, the last
step's dE/dh that train_on_batch returns for logging) is held to max(1e-4, 3 x floor): from
forward-initialised vodes its upper layers are pure cancellation noise in ANY fp32 implementation (the
fp32 oracle is up to 12x max|fp64| away from the fp64 oracle on gh1 of the 12-block case), the row says so
("bar" > 1e-4).
"""
from __future__ import annotations

import dataclasses

import numpy as np
import torch

from oracle import pcvit_ref as R

TOL = 1e-4

# name -> (config name, overrides, B, T).  Reference sources of the configs: src/config.py:121-137
# (debug_tiny), :442-517 (6block), sweeps/sweep_12block_comprehensive.yaml:8-47 (12block, canonical
# D=128 H=4), src/decoder_transformer.py:1070-1078 (ImageNet 224/16 preset), :100-125 with a 4-tuple
# image_shape (16-frame video, N = 1024 tokens).
FULL_CASES = {
    "cfg2_6block_B200_T20": ("6block", {}, 200, 20),
    "cfg1_debug_tiny_B200_T20": ("debug_tiny", {}, 200, 20),
    "cfg3_12block_D128_H4_B32_T20": ("12block", {}, 32, 20),
    "cfg3_12block_D96_H4_B16_T20": ("12block", {"hidden_size": 96}, 16, 20),
    "cfg5_video16_N1024_L2_B1_T20": ("video16", {"num_blocks": 2}, 1, 20),
    "cfg5_video16_N1024_L6_B1_T20": ("video16", {}, 1, 20),          # config 5 at its own depth
    "cfg4_imagenet224_L6_B2_T10": ("imagenet224", {}, 2, 10),
}


# Cases where fp32 itself cannot hold the plain 1e-4 on every tensor: the SAME oracle run in fp32 is further than
# 1e-4 from its fp64 run (config 5 at its own depth, B = 1: 1024 tokens x 6 blocks x 20 chained steps with the clip
# firing on a single image's norm -- fp32 oracle vs fp64 oracle: h1 8.0e-4, train_mse 2.1e-4; the HIP path sits AT
# that floor, 7.2e-4 / 2.1e-4).  No fp32 implementation -- the reference's JAX path included -- can be held to 1e-4
# against fp64 there, so for these cases EVERY tensor gets the rule final_h_grad gets everywhere:
# bar = max(1e-4, 3 x fp32-oracle floor).  The rows say which tensors it relaxed ("bar" > 1e-4).
NOISE_FLOOR_CASES = {"cfg5_video16_N1024_L6_B1_T20"}

# the same cases in the opt-in bf16 mode (bar 1e-2): the shapes its bf16 GEMM + bf16 attention kernels serve,
# with all T chained steps (the 64-token configs run their fused fp32 kernels only in fp32 mode)
BF16_FULL_CASES = ["cfg4_imagenet224_L6_B2_T10", "cfg5_video16_N1024_L2_B1_T20", "cfg5_video16_N1024_L6_B1_T20",
                   "cfg3_12block_D128_H4_B32_T20"]
BF16_TOL = 1e-2
# dtype_mode 2 (three-term bf16 split in the per-op GEMMs, fp32-grade products) at full size, held to the fp32 bars
BF16X3_FULL_CASES = ["cfg2_6block_B200_T20", "cfg3_12block_D128_H4_B32_T20", "cfg4_imagenet224_L6_B2_T10"]


def build_full_case(name: str, compute_dtype: str = "fp32"):
    import cora_amd as ca
    from cora_amd.optim import lr_at
    from cora_amd.params import init_params
    from cora_amd.posenc import create_positional_encoding
    cfg_name, over, B, T = FULL_CASES[name]
    mc = ca.MODEL_CONFIGS[cfg_name]
    if over:
        mc = dataclasses.replace(mc, **over)
    tc = ca.transformer_config_from(mc)
    dims = R.Dims(tuple(tc.image_shape), tc.patch_size, tc.hidden_size, tc.num_heads, tc.num_blocks)
    W = init_params(dims.P, dims.D, dims.M, dims.L, 42)
    pos = create_positional_encoding(tuple(tc.image_shape), dims.D, tc.patch_size, theta=tc.theta)
    rng = np.random.default_rng(1234)
    x = rng.uniform(0, 1, (B,) + tuple(tc.image_shape)).astype(np.float32)
    h0 = (0.01 * rng.standard_normal((B, dims.N, dims.P))).astype(np.float32)
    h = [np.asarray(t, dtype=np.float32) for t in R.init_vodes(h0, W, pos, dims)]
    hp = R.Hyper(l1=mc.intermediate_l1_coeff, l2=mc.intermediate_l2_coeff, use_scaling=mc.use_inference_lr_scaling,
                 scale_base=mc.inference_lr_scale_base, clip_h=mc.h_grad_clip_norm, clip_w=mc.w_grad_clip_norm,
                 momentum=mc.hidden_momentum, wd=mc.weight_decay)
    model = ca.TransformerDecoder(tc, seed=42, max_T=max(T, 64), compute_dtype=compute_dtype)
    model.positional_embedding = pos
    model.set_weights(W)
    for l, t in enumerate(h):
        model.vodes[l].h.set(t)
    # the driver's recipe for this config (examples/debug_transformer_wandb.py:1091-1167): 250 batches x epochs steps
    optim_w, optim_h = ca.make_optimizers(mc, model, total_train_steps=250 * mc.epochs)
    eta_h = [lr_at(optim_h._recipe().base.learning_rate, t) for t in range(T)]
    eta_w = lr_at(optim_w._recipe().base.learning_rate, 0)
    return dict(mc=mc, tc=tc, dims=dims, W=W, pos=pos, x=x, h=h, hp=hp, model=model, optim_w=optim_w, optim_h=optim_h,
                eta_h=eta_h, eta_w=eta_w, B=B, T=T)


def run_full_case(name: str, compute_dtype: str = "fp32"):
    """-> (rows, info): rows[tensor] = {"err", "floor", "bar"}; info = sizes and timings.
    compute_dtype "bf16": the opt-in bf16 mode, bar 1e-2 on what north_star names (final vodes, per-step
    energies, weight gradients) and on final_w_energy / train_mse; the gh* rows (final_h_grad, returned for
    logging: cancellation noise of e = h - u, the fp32 oracle itself is 2e-2 off on gh0 of the ImageNet shape)
    are recorded with "bar": None and not held, as in tests/test_gpu_parity.py::test_bf16_opt_in."""
    import time
    import cora_amd as ca
    c = build_full_case(name, compute_dtype)
    tol = BF16_TOL if compute_dtype == "bf16" else TOL
    T, L = c["T"], c["dims"].L
    args = (T, c["x"], c["W"], c["h"], c["pos"], c["dims"], c["hp"], c["eta_h"], c["eta_w"])
    t0 = time.perf_counter()
    ref = R.train_on_batch(*args)
    t1 = time.perf_counter()
    r32 = R.train_on_batch(*args, dtype=torch.float32)
    t2 = time.perf_counter()
    model = c["model"]
    kernel_path = model.kernel_path()
    Eh, Ew, gh, gw, mse = ca.train_on_batch(T, c["x"], model=model, optim_w=c["optim_w"], optim_h=c["optim_h"])
    tot, per = ca.step_energies(model, T)
    rows = {}

    def put(key, got, want, f32, scale=None):
        want = np.asarray(want, dtype=np.float64)
        sc = (np.abs(want).max() + 1e-30) if scale is None else scale
        err = float(np.abs(np.asarray(got, dtype=np.float64) - want).max() / sc)
        floor = float(np.abs(np.asarray(f32, dtype=np.float64) - want).max() / sc)
        bar = max(tol, 3.0 * floor) if (key.startswith("gh") or (name in NOISE_FLOOR_CASES and compute_dtype != "bf16")) else tol
        if compute_dtype == "bf16" and key.startswith("gh"):
            bar = None
        rows[key] = {"err": err, "floor": floor, "bar": bar}

    put("energies[T]", tot, ref["energies"], r32["energies"])
    put("layer_energies[T,L+3]", per, ref["layer_energies"], r32["layer_energies"])
    put("final_w_energy", Ew, ref["final_w_energy"], r32["final_w_energy"])
    put("train_mse", mse, ref["train_mse"], r32["train_mse"])
    for l in range(L + 2):
        put(f"h{l}", model.vodes[l].h.get(), ref["h"][l], r32["h"][l])
    for l in range(L + 2):
        put(f"gh{l}", gh[l], ref["final_h_grad"][l], r32["final_h_grad"][l])
    gnorm = max(np.abs(v).max() for v in ref["w_grad"].values())
    for k, v in ref["w_grad"].items():
        put("gw" + k, gw[k], v, r32["w_grad"][k], scale=gnorm)
    info = {"config": FULL_CASES[name][0], "overrides": FULL_CASES[name][1], "B": c["B"], "T": T,
            "N": c["dims"].N, "D": c["dims"].D, "H": c["dims"].H, "L": L,
            "oracle_fp64_s": round(t1 - t0, 2), "oracle_fp32_s": round(t2 - t1, 2),
            "kernel_path": kernel_path, "compute_dtype": compute_dtype,
            "energy_first_last": [float(ref["energies"][0]), float(ref["energies"][-1])]}
    model.close()
    return rows, info


def summarise(rows):
    """worst err, worst err/bar, and the tensors whose bar is not the plain 1e-4"""
    held = {k: v for k, v in rows.items() if v["bar"] is not None}
    worst = max(held, key=lambda k: held[k]["err"] / held[k]["bar"])
    relaxed = {k: v for k, v in held.items() if v["bar"] > TOL}
    return worst, relaxed
