"""GPU parity: the HIP path (through the C-ABI) against the fp64 oracle on identical seeded inputs.

Tolerance (BASELINE.json north_star): fp32 path, rel <= 1e-4 on final vodes, per-step energies and
weight gradients.  `relerr` is max|a-b| / max|b| per tensor.

Bars.  What north_star names -- final vodes, per-step energies (total and per vode), the weight
gradients, plus the two scalars of the 5-tuple (final_w_energy, train_mse) -- is held to the PLAIN
1e-4, no relaxation (measured: <= 3e-6 on every case incl. B=200 / T=20 / 12 blocks,
profiles/r2_parity_table.json).  Only `final_h_grad` (gh*: the last step's dE/dh, returned by
train_on_batch for logging) gets a noise-floor rule: the double residual u_{l+1} = 2 h_l + ... makes
forward-initialised vodes grow like 2^l, so e_l = h_l - u_l cancels ~l bits in ANY fp32
implementation (the reference's JAX path included) and at 12 blocks the fp32 CPU restatement itself
is 1e-3 .. 10x away from the fp64 one on dE/dh of the upper vodes.  gh* is held to
max(1e-4, 3 x floor), floor = error of the oracle run in fp32 against the oracle run in fp64.

AdamW's first steps divide by |g|+eps, which amplifies fp32 noise in near-zero gradient entries
without bound, so the updated weights are checked against the oracle's AdamW applied to the
gradient the GPU actually produced (kernel check, 1e-6) -- the gradient itself is held to the bar
above."""
import zlib

import numpy as np
import pytest
import torch

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
    "d64_h2_fused": ((3, 32, 32), 4, 64, 2, 3, 3, 4),      # fused kernel, head dim 32
    "d64_h4_fused_12block": ((3, 32, 32), 4, 64, 4, 12, 2, 3),   # fused kernel, head dim 16 (a 12-block sweep point)
    "cfg3_shape_B2": ((3, 32, 32), 4, 128, 4, 12, 2, 3),
    # pc_block_wide2_kernel at a depth where no fp32 floor relaxes a bar: its VJP held to the plain 1e-4 on every gh
    "d128_h4_shallow": ((3, 32, 32), 4, 128, 4, 2, 2, 3),
    # heads of 128 / 64 on the same kernel: 32-column slices, scores accumulated across a head's slices (sliced form)
    "d128_h1_shallow": ((3, 32, 32), 4, 128, 1, 2, 2, 3),
    "d128_h2_shallow": ((3, 32, 32), 4, 128, 2, 2, 3, 3),
    "d128_h2_12block": ((3, 32, 32), 4, 128, 2, 12, 2, 3),
    # N = 64 and D = 96 / 128 but a patch dim other than 48: the wide fused kernels' edge layers are written for
    # P = 48, so these must take the per-op path (asserted in test_wide_kernels_need_patch_dim_48)
    "d128_h4_P16": ((1, 32, 32), 4, 128, 4, 2, 2, 3),
    "d96_h2_P192": ((3, 64, 64), 8, 96, 2, 2, 2, 3),
    "n36_h2_ragged_chunk": ((3, 24, 24), 4, 32, 2, 2, 2, 3),   # N = 36: masked rows in the single-chunk attention kernels
    # odd row counts (B * N = 147 / 243): a LayerNorm-backward wave's four rows straddle two vodes, attention tiles
    # with one live row (49 = 3 x 16 + 1; 81 = 64 + 16 + 1)
    "n49_h2_odd_rows": ((1, 28, 28), 4, 64, 2, 2, 3, 3),
    "n81_h3_odd_rows": ((3, 36, 36), 4, 96, 3, 3, 3, 2),
    "d96_h2": ((3, 32, 32), 4, 96, 2, 2, 2, 3),               # wide2: heads of 48 as three 16-column slices
    "d96_h1_shallow": ((3, 32, 32), 4, 96, 1, 2, 3, 3),        # wide2: one head of 96 as three 32-column slices
    "d96_h2_12block": ((3, 32, 32), 4, 96, 2, 12, 2, 3),
    "d96_h4_headdim24": ((3, 32, 32), 4, 96, 4, 2, 2, 3),      # wide2: heads of 24 = one and a half column groups, lane-masked slices
    "d96_h4_12block": ((3, 32, 32), 4, 96, 4, 12, 3, 3),
    "cfg4_shape_B1": ((3, 224, 224), 16, 384, 8, 2, 4, 2),
    "d1024_h16_widest": ((3, 64, 64), 8, 1024, 16, 1, 2, 2),   # the widest hidden size the LayerNorm kernels take (16 float4 per lane)
    "cfg5_video_small": ((4, 3, 32, 32), 4, 64, 1, 2, 1, 2),
    # the row-tile fused kernels (fused_rows.hip: D = 64, B*N a multiple of 64, any sequence length) around attn.hip
    "rows_n256_h2_L3": ((4, 3, 32, 32), 4, 64, 2, 3, 1, 3),
    "rows_n64_P16_h4": ((1, 32, 32), 4, 64, 4, 2, 3, 3),     # 64 tokens but P != 48: not the per-(block, image) kernel
    "rows_n512_h1_B2": ((8, 3, 32, 32), 4, 64, 1, 2, 2, 3),
    "rows_n256_h4": ((4, 3, 32, 32), 4, 64, 4, 2, 1, 3),        # single-pass attention backward at head dim 16
    "rows_n192_h1": ((3, 3, 32, 32), 4, 64, 1, 2, 1, 3),        # N = 192: the second 128-key block is half empty
    "n144_h4_headdim16": ((3, 48, 48), 4, 64, 4, 2, 2, 3),     # N = 144: two chunks + one 16-row tile, head dim 16
    "n144_h2_headdim32": ((3, 48, 48), 4, 64, 2, 1, 2, 2),
    "n144_h1_headdim96": ((3, 48, 48), 4, 96, 1, 1, 2, 2),     # chunked attention at the wide head dims
    "n144_h1_headdim128": ((3, 48, 48), 4, 128, 1, 1, 2, 2),
}


def run_case(name, l1=0.0, l2=0.0, clip_h=2000.0, clip_w=500.0, return_floor=False, compute_dtype="fp32", expect_path=None, **kw):
    import cora_amd as ca
    shape, p, D, H, L, B, T = CASES[name]
    case = make_case(shape, p, D, H, L, B, seed=zlib.crc32(name.encode()) % 1000, l1=l1, l2=l2, clip_h=clip_h, clip_w=clip_w, **kw)
    model, optim_w, optim_h = build_model(case, compute_dtype=compute_dtype)
    if expect_path is not None:
        assert model.kernel_path() == expect_path, (name, model.kernel_path())
    eta = [0.095] * T
    args = (T, case["x"], case["W"], case["h"], case["pos"], case["dims"], case["hp"], eta, 1e-3)
    ref = R.train_on_batch(*args)
    r32 = R.train_on_batch(*args, dtype=torch.float32)        # fp32 noise floor of the same math
    Eh, Ew, gh, gw, mse = ca.train_on_batch(T, case["x"], model=model, optim_w=optim_w, optim_h=optim_h)
    tot, per = ca.step_energies(model, T)
    errs, floor = {}, {}

    def put(key, got, want, f32, scale=None):
        want = np.asarray(want, dtype=np.float64)
        sc = (np.abs(want).max() + 1e-30) if scale is None else scale
        errs[key] = float(np.abs(np.asarray(got, dtype=np.float64) - want).max() / sc)
        floor[key] = float(np.abs(np.asarray(f32, dtype=np.float64) - want).max() / sc)

    put("energies", tot, ref["energies"], r32["energies"])
    put("layer_energies", per, ref["layer_energies"], r32["layer_energies"])
    put("final_h_energy", Eh, ref["energies"][-1], r32["energies"][-1])
    put("final_w_energy", Ew, ref["final_w_energy"], r32["final_w_energy"])
    for l in range(L + 2):
        put(f"h{l}", model.vodes[l].h.get(), ref["h"][l], r32["h"][l])
        put(f"gh{l}", gh[l], ref["final_h_grad"][l], r32["final_h_grad"][l])
    gnorm = max(np.abs(v).max() for v in ref["w_grad"].values())
    gw_gpu = {k: gw[k] for k in ref["w_grad"]}
    for k, v in ref["w_grad"].items():
        put("gw" + k, gw_gpu[k], v, r32["w_grad"][k], scale=gnorm)
    put("train_mse", mse, ref["train_mse"], r32["train_mse"])
    # AdamW kernel: oracle update applied to the GPU's own gradient
    neww = model.get_weights()
    hp = case["hp"]
    g = {k: torch.as_tensor(v, dtype=torch.float64) / B for k, v in gw_gpu.items()}
    g = dict(zip(g.keys(), R._clip(list(g.values()), hp.clip_w)))
    for k, gk in g.items():
        mu, nu = (1 - hp.b1) * gk, (1 - hp.b2) * gk * gk
        upd = (mu / (1 - hp.b1)) / (torch.sqrt(nu / (1 - hp.b2)) + hp.eps) + hp.wd * torch.as_tensor(case["W"][k], dtype=torch.float64)
        want = case["W"][k].astype(np.float64) - 1e-3 * upd.numpy()
        errs["adamw" + k] = relerr(neww[k], want)
        floor["adamw" + k] = 0.0
    model.close()
    bad = {k: (v, floor[k]) for k, v in errs.items() if not v <= bar_for(k, floor[k])}
    assert not bad, f"{name}: parity failures (err, fp32 floor): {bad}"
    return (errs, floor) if return_floor else errs


def bar_for(key, floor):
    """1e-5 for the AdamW kernel check, max(1e-4, 3 x fp32 floor) for final_h_grad, plain 1e-4 for the rest."""
    if key.startswith("adamw"):
        return 1e-5
    if key.startswith("gh"):
        return max(TOL, 3.0 * floor)
    return TOL


@pytest.mark.parametrize("name", list(CASES))
def test_step_parity(name):
    run_case(name)


@pytest.mark.parametrize("name,path", [("d128_h4_P16", "per-op"), ("d96_h2_P192", "per-op"), ("d128_h4_shallow", "fused-wide"),
                                       ("d128_h1_shallow", "fused-wide"), ("cfg2_shape_B4", "fused"),
                                       ("cfg5_video_small", "fused-rows"), ("rows_n64_P16_h4", "fused-rows"),
                                       ("n144_h4_headdim16", "per-op")])
def test_wide_kernels_need_patch_dim_48(name, path):
    """cora_kernel_path: the wide fused kernels (D = 96 / 128) take P = 48 only -- their edge layers hard-code the
    4x4x3 patch (fused_wide_common.h: edge_path_wide) -- so a 64-token config with another patch dim (a 1-channel
    32x32 image: P = 16; 64x64 with 8x8 patches: P = 192) runs the per-op path, and passes the same bars there."""
    run_case(name, expect_path=path)


@pytest.mark.parametrize("name", ["d128_h1_shallow", "d128_h2_shallow", "d128_h4_shallow", "d96_h2", "d96_h1_shallow", "d96_h4_headdim24"])
def test_one_workgroup_wide_kernel_at_d128(name, monkeypatch):
    """CORA_WIDE1=1: pc_block_wide_kernel (one workgroup per CU, K / V of all heads in LDS) instead of
    pc_block_wide2_kernel -- round 2's kernel for these shapes, kept reachable as the second implementation."""
    monkeypatch.setenv("CORA_WIDE1", "1")
    run_case(name, expect_path="fused-wide")


@pytest.mark.parametrize("name", ["cfg2_shape_B4", "d128_h2_shallow", "d96_h4_headdim24", "rows_n256_h2_L3", "tiny_2block_h2"])
def test_profile_step_leaves_the_trajectory_alone(name):
    """cora_profile_step (bench.py's roofline leg) warms the clocks with inference steps at step size 0 and times the
    block kernel as a burst of identical launches: vodes, momentum and the next batches must come out as if it had never
    run -- on every kernel path (D = 64 fused with the deferred update, wide2, row-tile, per-op)."""
    import ctypes as C
    import cora_amd as ca
    from cora_amd import _lib
    shape, p, D, H, L, B, T = CASES[name]
    case = make_case(shape, p, D, H, L, B, seed=77, clip_h=3.0, clip_w=0.5)
    res = []
    for prof in (False, True):
        model, ow, oh = build_model(case)
        outs = [ca.train_on_batch(T, case["x"], model=model, optim_w=ow, optim_h=oh)]
        if prof:
            lib, ctx, cap = _lib.load(), model._ctx, 16
            names = C.create_string_buffer(32 * cap)
            launches, nout = (C.c_int * cap)(), C.c_int()
            ms, fl, by = (C.c_double * cap)(), (C.c_double * cap)(), (C.c_double * cap)()
            _lib.check(lib.cora_profile_step(ctx, cap, names, launches, ms, fl, by, C.byref(nout)), ctx)
            assert nout.value > 0 and sum(launches[k] for k in range(nout.value)) > 0
            assert all(np.isfinite(ms[k]) and ms[k] >= 0 for k in range(nout.value))
        outs.append(ca.train_on_batch(T, case["x"], model=model, optim_w=ow, optim_h=oh))
        tot, per = ca.step_energies(model, T)
        res.append(dict(outs=outs, tot=tot, per=per, h=[model.vodes[l].h.get() for l in range(L + 2)], w=model.get_weights()))
        model.close()
    a, b = res
    assert relerr(a["tot"], b["tot"]) <= 1e-6 and relerr(a["per"], b["per"]) <= 1e-6
    for oa, ob in zip(a["outs"], b["outs"]):
        assert oa[0] == pytest.approx(ob[0], rel=1e-6) and oa[1] == pytest.approx(ob[1], rel=1e-6) and oa[4] == pytest.approx(ob[4], rel=1e-5)
    for l in range(L + 2):
        assert relerr(a["h"][l], b["h"][l]) <= 2e-6, l
    # two AdamW steps: split-K float atomics make dE/dW differ run to run at the 1e-7 level and AdamW's division by
    # sqrt(nu) + eps amplifies that on near-zero entries -- a leak of profile state would be O(1)
    for k in a["w"]:
        assert relerr(a["w"][k], b["w"][k]) <= 5e-5, k


def test_regularisers_and_clipping():
    # l1/l2 vode regularisers on, and clip thresholds low enough that both clips trigger
    run_case("tiny_2block_h2", l1=1e-3, l2=1e-3, clip_h=5.0, clip_w=0.5)


@pytest.mark.parametrize("name", ["rows_n256_h2_L3", "rows_n64_P16_h4"])
def test_rows_path_regularisers_clip_and_grad_norm(name):
    """The row-tile fused kernels with everything their epilogues fold in switched on: l1 / l2 vode regularisers
    (energy in rows_mid, gradient in rows_out), both clips firing, and normalize_vode_h_gradients on top."""
    run_case(name, l1=1e-3, l2=1e-3, clip_h=5.0, clip_w=0.5, expect_path="fused-rows")
    run_case(name, use_grad_norm=True, grad_norm_target=1.5, clip_h=2.0, expect_path="fused-rows")


@pytest.mark.parametrize("name", ["rows_n512_h1_B2", "rows_n256_h2_L3"])
def test_rows_path_with_two_kernel_attention_backward(name, monkeypatch):
    """CORA_ATTN_BWD2=1: the row-tile kernels around the two-kernel attention backward (attn_bwd_q / attn_bwd_kv,
    S and dP recomputed per side) instead of the single-pass one (dQ summed with float atomics)."""
    monkeypatch.setenv("CORA_ATTN_BWD2", "1")
    run_case(name, expect_path="fused-rows")


def test_rows_path_with_per_op_edge_layers(monkeypatch):
    """CORA_ROWS_NO_EDGE=1: the two edge layers as per-op GEMM launches instead of extra workgroups of rows_qkv / rows_out."""
    monkeypatch.setenv("CORA_ROWS_NO_EDGE", "1")
    run_case("rows_n512_h1_B2", expect_path="fused-rows")


def test_rows_path_matches_per_op_path(monkeypatch):
    """Two independent implementations of the same step -- the row-tile fused kernels and the per-op GEMM /
    LayerNorm launches (CORA_NO_ROWS=1) -- on the same inputs: vodes, energies, dE/dW after a whole step."""
    import cora_amd as ca
    shape, p, D, H, L, B, T = CASES["rows_n512_h1_B2"]
    case = make_case(shape, p, D, H, L, B, seed=31)
    res = []
    for rows in (True, False):
        if rows:
            monkeypatch.delenv("CORA_NO_ROWS", raising=False)
        else:
            monkeypatch.setenv("CORA_NO_ROWS", "1")
        model, ow, oh = build_model(case)
        assert model.kernel_path() == ("fused-rows" if rows else "per-op")
        Eh, Ew, gh, gw, mse = ca.train_on_batch(T, case["x"], model=model, optim_w=ow, optim_h=oh)
        tot, per = ca.step_energies(model, T)
        res.append(dict(Ew=Ew, mse=mse, tot=tot, per=per, h=[model.vodes[l].h.get() for l in range(L + 2)],
                        gw={k: gw[k] for k in model.param_keys()}))
        model.close()
    a, b = res
    assert relerr(a["tot"], b["tot"]) <= 1e-5 and relerr(a["per"], b["per"]) <= 1e-5
    assert abs(a["Ew"] - b["Ew"]) <= 1e-5 * abs(b["Ew"]) and abs(a["mse"] - b["mse"]) <= 1e-5 * abs(b["mse"])
    for l in range(L + 2):
        assert relerr(a["h"][l], b["h"][l]) <= 1e-5, l
    gmax = max(np.abs(v).max() for v in b["gw"].values())
    for k in b["gw"]:
        assert np.abs(a["gw"][k] - b["gw"][k]).max() <= 1e-4 * gmax, k


FUSED64_CASES = ["cfg2_shape_B4", "d64_h2_fused", "d64_h4_fused_12block"]


@pytest.mark.parametrize("pair", ["1", "0"])
@pytest.mark.parametrize("name", FUSED64_CASES)
def test_fused_kernel_both_wave_forms_match_oracle(name, pair, monkeypatch):
    """The D = 64 per-(block, image) inference kernel exists in two forms: 4 waves (the shipped one) and an opt-in
    8-wave form splitting every GEMM between wave pairs (`block_body_pair`, CORA_PAIR=1; one, two and four heads
    take different exchange points; measured and found not to pay).  Both against the oracle."""
    monkeypatch.setenv("CORA_PAIR", pair)
    run_case(name, expect_path="fused")


@pytest.mark.parametrize("H", [1, 2, 4])
def test_fused_pair_kernel_matches_four_wave_kernel_at_B25(H, monkeypatch):
    """B = 25 images x 6 blocks -- the shard an 8-GPU run of the reference's batch of 200 gives each rank, and the
    launch size the 8-wave form was built for: a whole step (T = 8, clip firing) on both kernels must agree on
    vodes, energies and weight gradients to fp32 rounding (only the summation order inside GEMM rows differs)."""
    import cora_amd as ca
    case = make_case((3, 32, 32), 4, 64, H, 6, 25, seed=77 + H, clip_h=3.0, clip_w=0.5)
    T, res = 8, []
    for pair in ("1", "0"):
        monkeypatch.setenv("CORA_PAIR", pair)
        model, ow, oh = build_model(case)
        assert model.kernel_path() == "fused"
        outs = [ca.train_on_batch(T, case["x"], model=model, optim_w=ow, optim_h=oh) for _ in range(2)]   # eager, then graph replay
        tot, per = ca.step_energies(model, T)
        res.append(dict(out=outs[-1], tot=tot, per=per, h=[model.vodes[l].h.get() for l in range(8)],
                        gw={k: outs[-1][3][k] for k in model.param_keys()}))
        model.close()
    a, b = res
    assert relerr(a["tot"], b["tot"]) <= 1e-5 and relerr(a["per"], b["per"]) <= 1e-5
    for l in range(8):
        assert relerr(a["h"][l], b["h"][l]) <= 2e-5, l
    gmax = max(np.abs(v).max() for v in b["gw"].values())
    for k in b["gw"]:
        assert np.abs(a["gw"][k] - b["gw"][k]).max() <= 1e-4 * gmax, k


@pytest.mark.parametrize("T", [1, 2, 3, 6])
@pytest.mark.parametrize("variant", ["h1", "h2_gradnorm", "h4_reg", "wide2", "wide2_gradnorm_reg", "wide2_h2", "wide2_h1_reg", "wide2_d96_h2", "wide2_d96_h1_gradnorm", "wide2_d96_h4_reg"])
def test_deferred_vode_update_matches_update_kernel(variant, T, monkeypatch):
    """D = 64 fused path and the D = 128 / heads-of-32 kernel (wide2): inside a T-step loop the momentum-SGD update of step t is applied by the inference launch
    of step t + 1 as it loads its rows (FusedP::lazy, vodes / momentum / gradients ping-ponging between two arena
    sets; the last step updates explicitly, back into set 0).  CORA_NO_LAZY=1 launches update_kernel after every step
    instead.  Same arithmetic, so vodes, final gradients, energies and dE/dW must agree to fp32 rounding -- with the
    clip firing, with per-leaf gradient normalisation, with regularisers, for odd and even numbers of deferred
    updates (T - 1), and across two batches (the second replays the captured graph)."""
    import cora_amd as ca
    D, H = {"h1": (64, 1), "h2_gradnorm": (64, 2), "h4_reg": (64, 4), "wide2": (128, 4), "wide2_gradnorm_reg": (128, 4),
            "wide2_h2": (128, 2), "wide2_h1_reg": (128, 1), "wide2_d96_h2": (96, 2), "wide2_d96_h1_gradnorm": (96, 1), "wide2_d96_h4_reg": (96, 4)}[variant]
    kw = dict(clip_h=3.0, clip_w=0.5)
    if "gradnorm" in variant:
        kw.update(use_grad_norm=True, grad_norm_target=2.0)
    if "reg" in variant:
        kw.update(l1=1e-3, l2=1e-3)
    case = make_case((3, 32, 32), 4, D, H, 3, 5, seed=900 + T, **kw)
    res = []
    for lazy in (True, False):
        if lazy:
            monkeypatch.delenv("CORA_NO_LAZY", raising=False)
        else:
            monkeypatch.setenv("CORA_NO_LAZY", "1")
        model, ow, oh = build_model(case)
        assert model.kernel_path() == ("fused" if D == 64 else "fused-wide")
        outs = [ca.train_on_batch(T, case["x"], model=model, optim_w=ow, optim_h=oh) for _ in range(3)]
        tot, per = ca.step_energies(model, T)
        res.append(dict(out=outs, tot=tot, per=per, h=[model.vodes[l].h.get() for l in range(5)],
                        gh=[np.asarray(outs[-1][2][l]) for l in range(5)], gw={k: outs[-1][3][k] for k in model.param_keys()}))
        model.close()
    a, b = res
    assert relerr(a["tot"], b["tot"]) <= 1e-6 and relerr(a["per"], b["per"]) <= 1e-6
    for oa, ob in zip(a["out"], b["out"]):
        assert oa[0] == pytest.approx(ob[0], rel=1e-6) and oa[1] == pytest.approx(ob[1], rel=1e-6) and oa[4] == pytest.approx(ob[4], rel=1e-5)
    for l in range(5):
        assert relerr(a["h"][l], b["h"][l]) <= 2e-6, l
        assert relerr(a["gh"][l], b["gh"][l]) <= 2e-5, l
    gmax = max(np.abs(v).max() for v in b["gw"].values())
    for k in b["gw"]:
        assert np.abs(a["gw"][k] - b["gw"][k]).max() <= 2e-5 * gmax, k


def test_split_loop_cut_short_still_applies_the_deferred_update(monkeypatch):
    """cora_infer_begin / _grad / _update driven from the host and abandoned after two steps of four: the update of
    step 1 is still waiting for a launch that never comes (D = 64 path), so the next entry point that looks at the
    vodes (here cora_get_vode) has to apply it.  Against the same calls with CORA_NO_LAZY=1."""
    import ctypes as C
    import cora_amd as ca
    from cora_amd import _lib
    case = make_case((3, 32, 32), 4, 64, 2, 3, 4, seed=5, clip_h=3.0, clip_w=0.5)
    x = np.ascontiguousarray(case["x"], dtype=np.float32)
    eta = np.full(4, 0.095, dtype=np.float32)
    fp = lambda a: a.ctypes.data_as(C.c_void_p)
    res = []
    for lazy in (True, False):
        if lazy:
            monkeypatch.delenv("CORA_NO_LAZY", raising=False)
        else:
            monkeypatch.setenv("CORA_NO_LAZY", "1")
        model, ow, oh = build_model(case)
        ca.train_on_batch(3, case["x"], model=model, optim_w=ow, optim_h=oh)     # vodes initialised, one learning step
        lib, ctx = _lib.load(), model._ctx
        _lib.check(lib.cora_infer_begin(ctx, fp(x), 0, 4, fp(eta)), ctx)
        for t in range(2):
            _lib.check(lib.cora_infer_grad(ctx, t), ctx)
            _lib.check(lib.cora_infer_update(ctx, t), ctx)
        res.append([model.vodes[l].h.get() for l in range(5)])
        ca.train_on_batch(3, case["x"], model=model, optim_w=ow, optim_h=oh)     # and the ctx is usable afterwards
        res[-1] += [model.vodes[l].h.get() for l in range(5)]
        model.close()
    for a, b in zip(*res):
        assert relerr(a, b) <= 2e-6


BF16_TOL = 1e-2


BF16_CASES = ["tiny_2block_h2", "cfg2_shape_B4", "cfg4_shape_B1", "cfg3_shape_B2", "cfg5_video_small", "n36_h2_ragged_chunk",
              "n144_h4_headdim16", "n144_h2_headdim32", "n144_h1_headdim96", "n144_h1_headdim128"]


def run_bf16_case(name):
    """One train_on_batch step in bf16 mode against the fp64 oracle: {tensor: max|a-b| / max|b|}."""
    import cora_amd as ca
    shape, p, D, H, L, B, T = CASES[name]
    case = make_case(shape, p, D, H, L, B, seed=zlib.crc32(name.encode()) % 1000)
    model, optim_w, optim_h = build_model(case, compute_dtype="bf16")
    eta = [0.095] * T
    ref = R.train_on_batch(T, case["x"], case["W"], case["h"], case["pos"], case["dims"], case["hp"], eta, 1e-3)
    Eh, Ew, gh, gw, mse = ca.train_on_batch(T, case["x"], model=model, optim_w=optim_w, optim_h=optim_h)
    tot, per = ca.step_energies(model, T)
    errs = {"energies": relerr(tot, ref["energies"]), "final_w_energy": relerr(Ew, ref["final_w_energy"]),
            "train_mse": relerr(mse, ref["train_mse"])}
    for l in range(L + 2):
        errs[f"h{l}"] = relerr(model.vodes[l].h.get(), ref["h"][l])
    gnorm = max(np.abs(v).max() for v in ref["w_grad"].values())
    for k, v in ref["w_grad"].items():
        errs["gw" + k] = float(np.abs(gw[k].astype(np.float64) - v).max() / gnorm)
    model.close()
    return errs


@pytest.mark.parametrize("name", BF16_CASES)
def test_bf16_opt_in(name):
    """Opt-in bf16 mode (BASELINE.json north_star: "bf16 as an opt-in mode", bar rel <= 1e-2 on final
    vodes, per-step energies and weight gradients; configs[3] names bf16 MFMA for the 224x224 case):
    GEMM operands are rounded to bf16 on their way into v_mfma_f32_32x32x16_bf16 (two-term split for the
    forward predictions), attention longer than one chunk runs on v_mfma_f32_16x16x16_bf16 (split operands
    in the forward, plain bf16 in the backward), accumulation and every stored tensor stay fp32.  Gradients are compared per
    tensor against the largest entry of that tensor family, as in the fp32 test."""
    errs = run_bf16_case(name)
    print(name, "bf16 max rel err", max(errs.values()), max(errs, key=errs.get))
    bad = {k: v for k, v in errs.items() if not v <= BF16_TOL}
    assert not bad, f"{name}: bf16 parity failures: {bad}"


BF16X3_CASES = ["tiny_2block_h2", "cfg2_shape_B4", "cfg3_shape_B2", "cfg4_shape_B1", "cfg5_video_small", "n36_h2_ragged_chunk",
                "d96_h4_headdim24", "n144_h1_headdim128"]


@pytest.mark.parametrize("name", BF16X3_CASES)
def test_bf16x3_mode_holds_the_fp32_bar(name):
    """compute_dtype="bf16x3" (dtype_mode 2): every K-inner GEMM of the per-op path takes both operands as three
    bf16 terms (hi + mid + lo = the fp32 value to 2^-24) and keeps the six products above 2^-24 -- fp32-grade
    products from the bf16 matrix pipe; attention, accumulation and state stay fp32.  Held to the SAME bars as
    the fp32 path (plain 1e-4; 1e-5 on the AdamW check), not to the bf16 mode's 1e-2."""
    run_case(name, compute_dtype="bf16x3")


@pytest.mark.parametrize("rows", ["64", "128", "256"])
@pytest.mark.parametrize("name", ["cfg4_shape_B1", "cfg5_video_small", "n144_h4_headdim16", "n144_h1_headdim128"])
def test_bf16_attention_workgroup_shapes(name, rows, monkeypatch):
    """The bf16-MFMA attention kernels (attn_bf16.hip) pick 64 / 128 own rows per workgroup from the launch
    size (256 = two own tiles per wave is kept for experiments), so the small test shapes would only ever
    run the 64-row form: force each through CORA_ATTN_BF16_ROWS (read at every launch).  Same bar."""
    monkeypatch.setenv("CORA_ATTN_BF16_ROWS", rows)
    errs = run_bf16_case(name)
    bad = {k: v for k, v in errs.items() if not v <= BF16_TOL}
    assert not bad, f"{name} rows={rows}: bf16 parity failures: {bad}"


@pytest.mark.parametrize("name", ["cfg4_shape_B1", "cfg5_video_small", "n144_h4_headdim16", "n144_h1_headdim128"])
def test_attention_eight_wave_workgroups(name, monkeypatch):
    """fp32 attention with 8 waves (128 own rows) per workgroup -- chosen by launch size, so forced here."""
    monkeypatch.setenv("CORA_ATTN_WAVES", "8")
    run_case(name)


@pytest.mark.parametrize("name", ["tiny_2block_h2", "cfg2_shape_B4"])
def test_vode_grad_norm_option(name):
    # normalize_vode_h_gradients (src/decoder_transformer.py:353-389): per-leaf g / (||g|| + 1e-8) * target,
    # on the general path and on the fused path; the clip then sees the normalised leaves
    run_case(name, use_grad_norm=True, grad_norm_target=1.5, clip_h=2.0)


@pytest.mark.parametrize("name", ["tiny_2block_h2", "cfg2_shape_B4", "rows_n256_h2_L3"])
def test_ipc_mode_weight_update_every_step(name):
    """MODE 1 of train_on_batch (`update_weights_every_inference_step`, reference :418-439) against the
    oracle: vodes and energies to the fp32 bar.  AdamW's first steps turn every gradient entry into
    roughly +-eta_w regardless of its size, so weights after T such steps are compared on the
    entries whose fp64 gradient is not noise-sized at ANY of the T steps (|g_t| > 1e-3 max|g|: an entry that is
    noise at step 1 alone already takes +-eta_w with the sign of its rounding error -- random shapes of
    scripts/fuzz_oracle.py showed 0.03 ... 0.4 of the total step on such entries, the fp32 oracle itself 0.1), and the LAST iteration's vode
    gradient -- evaluated with weights that went through T-1 such steps -- is held to 1e-3."""
    import cora_amd as ca
    shape, p, D, H, L, B, T = CASES[name]
    case = make_case(shape, p, D, H, L, B, seed=77, clip_h=3.0)
    model, optim_w, optim_h = build_model(case, ipc=True)
    eta, eta_w = [0.095] * T, [1e-3] * T
    args = (T, case["x"], case["W"], case["h"], case["pos"], case["dims"], case["hp"], eta, eta_w)
    ref = R.train_on_batch_ipc(*args)
    r32 = R.train_on_batch_ipc(*args, dtype=torch.float32)
    Eh, Ew, gh, gw, mse = ca.train_on_batch(T, case["x"], model=model, optim_w=optim_w, optim_h=optim_h)
    assert optim_w.count == T
    tot, _ = ca.step_energies(model, T)
    errs, floor = {}, {}

    def put(key, got, want, f32):
        want = np.asarray(want, dtype=np.float64)
        sc = np.abs(want).max() + 1e-30
        errs[key] = float(np.abs(np.asarray(got, dtype=np.float64) - want).max() / sc)
        floor[key] = float(np.abs(np.asarray(f32, dtype=np.float64) - want).max() / sc)

    put("energies", tot, ref["energies"], r32["energies"])
    put("final_w_energy", Ew, ref["w_energies"][-1], r32["w_energies"][-1])
    put("train_mse", mse, ref["train_mse"], r32["train_mse"])
    for l in range(L + 2):
        put(f"h{l}", model.vodes[l].h.get(), ref["h"][l], r32["h"][l])
        put(f"gh{l}", gh[l], ref["final_h_grad"][l], r32["final_h_grad"][l])
    neww = model.get_weights()
    gmax = max(np.abs(v).max() for v in ref["w_grad"].values())
    for k, v in ref["weights"].items():
        sel = ref["w_grad_minabs"][k] > 1e-3 * gmax
        if sel.any():
            d = np.abs(neww[k].astype(np.float64) - v)[sel].max()
            d32 = np.abs(r32["weights"][k].astype(np.float64) - v)[sel].max()
            errs["w" + k] = float(d / (T * 1e-3)); floor["w" + k] = float(d32 / (T * 1e-3))   # in units of the total step size
    model.close()
    bad = {k: (v, floor[k]) for k, v in errs.items()
           if not v <= (max(2e-2, 3.0 * floor[k]) if k.startswith("w.") else
                        max(1e-3 if k.startswith("gh") else TOL, 3.0 * floor[k]))}
    assert not bad, f"{name}: iPC parity failures (err, fp32 floor): {bad}"


def test_two_batches_persistent_state_and_adam():
    """vodes persist across batches, AdamW moments and count persist (reference _optim.py:47,128).

    The second batch is checked in the form run_case uses for the first: the oracle restarts from the
    GPU's own state after batch 1 (weights, vodes, AdamW moments, count = 1), so its vodes, energies and
    dE/dW for batch 2 are held to the plain 1e-4 bar, and the weights after batch 2 are held to 1e-5
    against the oracle's clip + AdamW (count = 2, the GPU's batch-1 moments) applied to the GPU's own
    batch-2 gradient -- AdamW divides by sqrt(nu), so comparing two independently rounded trajectories of
    weights would only bound each move by lr."""
    import cora_amd as ca
    shape, p, D, H, L, B, T = CASES["tiny_2block_h2"]
    case = make_case(shape, p, D, H, L, B, seed=5)
    model, optim_w, optim_h = build_model(case)
    hp = case["hp"]
    eta = [0.095] * T
    rng = np.random.default_rng(9)
    x2 = rng.uniform(0, 1, case["x"].shape).astype(np.float32)
    ca.train_on_batch(T, case["x"], model=model, optim_w=optim_w, optim_h=optim_h)
    W1 = model.get_weights()
    h1 = [model.vodes[l].h.get() for l in range(L + 2)]
    mu1, nu1, cnt1 = model.get_adam_state()
    assert cnt1 == 1

    def per_key(flat):
        out = {}
        for k, shp in model._shapes.items():
            off, cnt = model._offset(k)
            out[k] = flat[off:off + cnt].reshape(shp).astype(np.float64)
        return out
    adam1 = R.AdamState(mu=per_key(mu1), nu=per_key(nu1), count=1)
    r2 = R.train_on_batch(T, x2, W1, h1, case["pos"], case["dims"], hp, eta, 1e-3, adam=adam1)
    Eh, Ew, gh, gw, mse = ca.train_on_batch(T, x2, model=model, optim_w=optim_w, optim_h=optim_h)
    assert abs(Ew - r2["final_w_energy"]) / abs(r2["final_w_energy"]) < TOL
    assert abs(mse - r2["train_mse"]) / r2["train_mse"] < TOL
    for l in range(L + 2):
        assert relerr(model.vodes[l].h.get(), r2["h"][l]) < TOL, l
    gnorm = max(np.abs(v).max() for v in r2["w_grad"].values())
    for k, v in r2["w_grad"].items():
        assert np.abs(gw[k].astype(np.float64) - v).max() / gnorm < TOL, k
    # AdamW, second step: oracle update from the GPU's moments applied to the GPU's own gradient
    mu2, nu2, cnt2 = model.get_adam_state()
    assert cnt2 == 2
    mu2, nu2 = per_key(mu2), per_key(nu2)
    neww = model.get_weights()
    g = {k: torch.as_tensor(gw[k], dtype=torch.float64) / B for k in r2["w_grad"]}
    g = dict(zip(g.keys(), R._clip(list(g.values()), hp.clip_w)))
    m1, n1 = per_key(mu1), per_key(nu1)
    for k, gk in g.items():
        mu = hp.b1 * torch.as_tensor(m1[k]) + (1 - hp.b1) * gk
        nu = hp.b2 * torch.as_tensor(n1[k]) + (1 - hp.b2) * gk * gk
        upd = (mu / (1 - hp.b1 ** 2)) / (torch.sqrt(nu / (1 - hp.b2 ** 2)) + hp.eps) + hp.wd * torch.as_tensor(W1[k], dtype=torch.float64)
        want = W1[k].astype(np.float64) - 1e-3 * upd.numpy()
        assert relerr(neww[k], want) <= 1e-5, ("adamw", k)
        # the moments themselves: nu squares the clipped fp32 gradient (twice its rounding) -> the plain bar
        assert relerr(mu2[k], mu.numpy()) <= 1e-5 and relerr(nu2[k], nu.numpy()) <= TOL, ("moments", k)
    model.close()


@pytest.mark.parametrize("name", ["0block", "tiny_2block_h2", "cfg2_shape_B4", "cfg5_video_small"])
def test_vode_state_layernorm(name):
    """use_vode_state_layernorm (src/decoder_transformer.py:212-218, 232-241): what vodes 1..L+1 hand on goes
    through a LayerNorm of its own; L+1 extra (gamma, beta) pairs in the parameter tree and in dE/dW.  The
    cfg2 shape takes the per-op path here (the fused kernel is built for the flag's default, off)."""
    errs = run_case(name, vode_ln=True, l1=1e-3, l2=1e-3, clip_h=5.0)
    assert any("vode_output_layernorms" in k for k in errs)


@pytest.mark.parametrize("name,vode_ln", [("tiny_2block_h2", False), ("tiny_2block_h2", True),
                                          ("cfg2_shape_B4", False), ("cfg3_shape_B2", False), ("d96_h2", False)])
def test_init_forward_energy_api(name, vode_ln):
    """INIT forward (h_l := u_l chained), status-None energy() and status-"forward" reconstruction -- on the
    per-op chain, the D=64 chain kernel (cfg 2 shape), the wide chain kernel (cfg 3 shape: one workgroup walks
    one image through 12 blocks) and the per-op chain under the one-workgroup wide kernel (D=96)."""
    import cora_amd as ca
    shape, p, D, H, L, B, T = CASES[name]
    case = make_case(shape, p, D, H, L, B, seed=11, vode_ln=vode_ln)
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


@pytest.mark.parametrize("per_op_walk", [False, True])
@pytest.mark.parametrize("name", ["rows_n256_h2_L3", "rows_n64_P16_h4", "rows_n512_h1_B2", "cfg5_video_small"])
def test_rows_path_init_and_reconstruction_walk(name, per_op_walk, monkeypatch):
    """Row-tile path: the status-"forward" walk (INIT pass and reconstruction) runs rows_qkv -> attention -> rows_mid in
    chain mode, one block per launch trio; CORA_NO_ROWS_CHAIN=1 keeps the seven per-op launches per block.  Both against
    the oracle: INIT vodes, the reconstruction, and the two against each other through train_mse of a whole step."""
    import cora_amd as ca
    import torch
    if per_op_walk:
        monkeypatch.setenv("CORA_NO_ROWS_CHAIN", "1")
    shape, p, D, H, L, B, T = CASES[name]
    case = make_case(shape, p, D, H, L, B, seed=13)
    model, ow, oh = build_model(case)
    assert model.kernel_path() == "fused-rows"
    model.init_vodes(B, h0=case["h0"])
    href = R.init_vodes(case["h0"], case["W"], case["pos"], case["dims"])
    for l in range(L + 2):
        assert relerr(model.vodes[l].h.get(), href[l]) < TOL, l
    for l, t in enumerate(case["h"]):
        model.vodes[l].h.set(t)
    with ca.step(model, ca.STATUS_FORWARD):
        rec = ca.forward(None, model=model)
    Wt = {k: torch.as_tensor(v, dtype=torch.float64) for k, v in case["W"].items()}
    ht = [torch.as_tensor(t, dtype=torch.float64) for t in case["h"]]
    rec_ref = R.predictions(ht, Wt, torch.as_tensor(case["pos"], dtype=torch.float64), case["dims"], chain=True)[-1].numpy()
    assert relerr(rec, rec_ref) < TOL
    ref = R.train_on_batch(T, case["x"], case["W"], case["h"], case["pos"], case["dims"], case["hp"], [0.095] * T, 1e-3)
    out = ca.train_on_batch(T, case["x"], model=model, optim_w=ow, optim_h=oh)
    assert out[4] == pytest.approx(ref["train_mse"], rel=TOL)
    model.close()


def test_rows_path_walk_in_bf16_mode():
    """bf16 mode on the row-tile path: the walk uses the bf16-MFMA attention between the fp32 row kernels (bar 1e-2)."""
    import cora_amd as ca
    shape, p, D, H, L, B, T = CASES["rows_n256_h2_L3"]
    case = make_case(shape, p, D, H, L, B, seed=13)
    model, ow, oh = build_model(case, compute_dtype="bf16")
    assert model.kernel_path() == "fused-rows"
    model.init_vodes(B, h0=case["h0"])
    href = R.init_vodes(case["h0"], case["W"], case["pos"], case["dims"])
    for l in range(L + 2):
        assert relerr(model.vodes[l].h.get(), href[l]) < 1e-2, l
    model.close()


@pytest.mark.parametrize("name", ["g_0block", "g_1block", "g_2block_h2_reg_clip", "g_video"])
def test_golden_vectors(name):
    """HIP path against the committed fixtures (tests/golden/*.npz, made by make_golden.py)."""
    import os
    import cora_amd as ca
    from tests.golden.make_golden import GOLDEN
    shape, p, D, H, L, B, T, l1, l2, ch, cw = GOLDEN[name]
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    dims = R.Dims(tuple(shape), p, D, H, L)
    hp = R.Hyper(l1=l1, l2=l2, clip_h=ch, clip_w=cw)
    case = dict(dims=dims, hp=hp, pos=z["pos"], x=z["x"],
                W={k[1:]: z[k] for k in z.files if k.startswith("W.")},
                h=[z[f"h_in{l}"] for l in range(L + 2)])
    model, optim_w, optim_h = build_model(case)
    Eh, Ew, gh, gw, mse = ca.train_on_batch(T, case["x"], model=model, optim_w=optim_w, optim_h=optim_h)
    tot, per = ca.step_energies(model, T)
    assert relerr(tot, z["energies"]) < TOL
    assert relerr(per, z["layer_energies"]) < TOL
    assert abs(Ew - float(z["final_w_energy"])) / float(z["final_w_energy"]) < TOL
    assert abs(mse - float(z["train_mse"])) / float(z["train_mse"]) < TOL
    for l in range(L + 2):
        assert relerr(model.vodes[l].h.get(), z[f"h_out{l}"]) < TOL, l
        assert relerr(gh[l], z[f"gh{l}"]) < TOL, l
    gnorm = max(np.abs(z["gW" + k]).max() for k in case["W"])
    for k in case["W"]:
        assert np.abs(gw[k] - z["gW" + k]).max() / gnorm < TOL, k
    model.close()


def test_graph_replay_matches_eager(monkeypatch):
    """Six consecutive batches at the full cfg-2 size: the hipGraph replay path and the eager launch
    path (CORA_NO_GRAPH=1) must walk the same trajectory.  Guards the replay-only failure seen with
    memset nodes inside the captured graph (stale momentum on the 5th replay)."""
    import cora_amd as ca
    mc = ca.MODEL_CONFIGS["6block"]
    rng = np.random.default_rng(1234)
    x = rng.uniform(0, 1, (200, 3, 32, 32)).astype(np.float32)
    h0 = (0.01 * rng.standard_normal((200, 64, 48))).astype(np.float32)
    traces = []
    for eager in (False, True):
        if eager:
            monkeypatch.setenv("CORA_NO_GRAPH", "1")
        else:
            monkeypatch.delenv("CORA_NO_GRAPH", raising=False)
        model = ca.TransformerDecoder(ca.transformer_config_from(mc), seed=42)
        model.init_vodes(200, h0=h0)
        optim_w, optim_h = ca.make_optimizers(mc, model, total_train_steps=10)
        tr = []
        for k in range(7):
            Eh, Ew, _, _, mse = ca.train_on_batch(20, x, model=model, optim_w=optim_w, optim_h=optim_h)
            tr.append((Eh, Ew, mse))
            if k == 1:
                torch.cuda.synchronize()      # the failing pattern had a device sync after 2 replays
        traces.append(np.array(tr))
        model.close()
    assert np.all(np.isfinite(traces[0])), traces[0]
    np.testing.assert_allclose(traces[0], traces[1], rtol=2e-3)


@pytest.mark.parametrize("name", ["cfg4_shape_B1", "n144_h2_headdim32", "n144_h1_headdim128"])
def test_bf16_mode_backward_scratch_as_bf16_changes_no_bit(name, monkeypatch):
    """bf16 mode, per-op inference step: da (gelu' epilogue -> W1^T GEMM) and dqkv (attention backward -> Wqkv^T GEMM) live
    as bf16 between their producer and the one GEMM that reads them.  That GEMM rounds its A operand to bf16 anyway, so
    vodes and their gradients must come out bit-identical to the fp32-scratch form (CORA_NO_BWD16=1); weight gradients
    (learning pass: fp32 scratch in both) and energies agree to accumulation-order noise (split-K / slot atomics)."""
    import cora_amd as ca
    shape, p, D, H, L, B, T = CASES[name]

    def run():
        case = make_case(shape, p, D, H, L, B, seed=5)
        model, optim_w, optim_h = build_model(case, compute_dtype="bf16")
        assert model.kernel_path() == "per-op"
        Eh, Ew, gh, gw, mse = ca.train_on_batch(T, case["x"], model=model, optim_w=optim_w, optim_h=optim_h)
        tot, _ = ca.step_energies(model, T)
        hs = [model.vodes[l].h.get() for l in range(L + 2)]
        ghs = [np.asarray(gh[l]) for l in range(L + 2)]          # lazy device views: fetch before the ctx goes
        gws = {k: np.asarray(gw[k]) for k in gw.keys()}
        model.close()
        return hs, ghs, gws, np.asarray(tot), mse

    a = run()
    monkeypatch.setenv("CORA_NO_BWD16", "1")
    b = run()
    for l in range(L + 2):
        assert np.array_equal(a[0][l], b[0][l]), f"vode {l}"
        assert np.array_equal(a[1][l], b[1][l]), f"final_h_grad {l}"
    gmax = max(np.abs(v).max() for v in b[2].values())
    for k in b[2]:
        assert np.abs(a[2][k] - b[2][k]).max() <= 1e-6 * gmax, k
    assert relerr(a[3], b[3]) <= 1e-9 and abs(a[4] - b[4]) <= 1e-6 * abs(b[4])     # MSE: after the (split-K) weight update
