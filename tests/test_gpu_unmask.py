"""GPU parity of the inference-only loops: `unmask_on_batch_enhanced`
(/root/reference/src/decoder_transformer.py:833-1046) and the linear-probe feature extractor
(/root/reference/examples/linear_probe.py:195-288), HIP path through the C-ABI against the fp64
oracle on identical seeded inputs.  Bar: rel <= 1e-4 (or 3x the fp32 floor of the oracle itself,
see tests/test_gpu_parity.py)."""
import numpy as np
import pytest
import torch

from oracle import pcvit_ref as R
from tests.util import build_model, make_case

pytestmark = pytest.mark.gpu
TOL = 1e-4

SHAPES = {
    # name: (image_shape, patch, D, H, L, B, T)
    "fused_shape": ((3, 32, 32), 4, 64, 1, 3, 3, 4),     # N=64, D=64, H=1: the fused kernel path
    "fused_h4": ((3, 32, 32), 4, 64, 4, 2, 2, 3),        # fused kernel, 4 heads of 16
    "general_h2": ((3, 16, 16), 4, 32, 2, 2, 4, 4),      # general (per-op) path
    "rows_n256_h2": ((3, 64, 64), 4, 64, 2, 2, 1, 3),    # row-tile fused kernels (edge layers inside their launches)
}


def _cmp(errs, floor, key, got, want, f32):
    want = np.asarray(want, dtype=np.float64)
    sc = np.abs(want).max() + 1e-30
    errs[key] = float(np.abs(np.asarray(got, dtype=np.float64) - want).max() / sc)
    floor[key] = float(np.abs(np.asarray(f32, dtype=np.float64) - want).max() / sc)


def _check(name, errs, floor):
    # plain 1e-4, except the per-step gradient NORMS (a logging quantity: dE/dh is cancellation noise on the
    # upper vodes in any fp32 implementation, see tests/test_gpu_parity.py), held to max(1e-4, 3 x fp32 floor)
    bad = {k: (v, floor[k]) for k, v in errs.items() if not v <= (max(TOL, 3.0 * floor[k]) if "gnorm" in k else TOL)}
    assert not bad, f"{name}: parity failures (err, fp32 floor): {bad}"


@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("lower_half,alpha,clip_h", [(True, 1.0, 2000.0), (False, 0.6, 3.0)])
def test_unmask_with_corruption(name, lower_half, alpha, clip_h):
    _unmask_with_corruption(name, lower_half, alpha, clip_h, False)


@pytest.mark.parametrize("name", list(SHAPES))
def test_unmask_vode_state_layernorm(name):
    """The same loop with use_vode_state_layernorm (:232-241) -- per-op path for both shapes."""
    _unmask_with_corruption(name, False, 0.6, 3.0, True)


def _unmask_with_corruption(name, lower_half, alpha, clip_h, vode_ln):
    import cora_amd as ca
    shape, p, D, H, L, B, T = SHAPES[name]
    case = make_case(shape, p, D, H, L, B, seed=21, clip_h=clip_h, vode_ln=vode_ln)
    model, optim_w, optim_h = build_model(case)
    model.config.use_lower_half_mask = lower_half
    model.config.inference_clamp_alpha = alpha
    x = case["x"]
    mask, x_init = ca.make_mask(model, x, 0.5, np.random.default_rng(3))
    args = (T, x, x_init, mask, case["W"], case["h"], case["pos"], case["dims"], case["hp"], [0.095] * T)
    ref = R.unmask_on_batch(*args, alpha=alpha, use_corruption=True)
    r32 = R.unmask_on_batch(*args, alpha=alpha, use_corruption=True, dtype=torch.float32)
    loss, recons, mask_out, log = ca.unmask_on_batch_enhanced(
        True, 0.5, [2, T], x, model=model, optim_h=optim_h, log_inference_grads=True, rng=np.random.default_rng(3))
    assert np.array_equal(mask_out, mask) and len(recons) == T
    errs, floor = {}, {}
    for t in range(T):
        _cmp(errs, floor, f"recon{t}", recons[t], ref["recons"][t], r32["recons"][t])
        got = [log[t][f"vode_{i}_grad_norm"] for i in range(L + 3)]
        _cmp(errs, floor, f"gnorm{t}", got, ref["grad_norms"][t], r32["grad_norms"][t])
    with ca.step(model, ca.STATUS_FORWARD):
        xhat = ca.forward(None, model=model)
    _cmp(errs, floor, "xhat", xhat, ref["xhat"], r32["xhat"])
    _cmp(errs, floor, "sensory", model.vodes[L + 2].h.get(), ref["sensory"], r32["sensory"])
    _cmp(errs, floor, "loss", loss, ref["loss"], r32["loss"])
    for l in range(L + 2):
        _cmp(errs, floor, f"h{l}", model.vodes[l].h.get(), ref["h"][l], r32["h"][l])
    if alpha == 1.0:   # hard clamp: the known pixels are exactly the ground truth after every step
        s = model.vodes[L + 2].h.get()
        mb = np.broadcast_to(mask, s.shape)
        assert np.array_equal(s[mb == 0], x[mb == 0])
    model.close()
    _check(name, errs, floor)


@pytest.mark.parametrize("name", list(SHAPES))
def test_unmask_without_corruption_and_features(name):
    import cora_amd as ca
    shape, p, D, H, L, B, T = SHAPES[name]
    case = make_case(shape, p, D, H, L, B, seed=22)
    model, optim_w, optim_h = build_model(case)
    x = case["x"]
    args = (T, x, x, None, case["W"], case["h"], case["pos"], case["dims"], case["hp"], [0.095] * T)
    ref = R.unmask_on_batch(*args, use_corruption=False)
    r32 = R.unmask_on_batch(*args, use_corruption=False, dtype=torch.float32)
    loss, recons, mask_out, log = ca.unmask_on_batch_enhanced(False, 0.0, [T], x, model=model, optim_h=optim_h)
    assert mask_out.shape == x.shape and not mask_out.any() and log == {}
    errs, floor = {}, {}
    for t in range(T):
        _cmp(errs, floor, f"recon{t}", recons[t], ref["recons"][t], r32["recons"][t])
    _cmp(errs, floor, "loss", loss, ref["loss"], r32["loss"])
    for l in range(L + 2):
        _cmp(errs, floor, f"h{l}", model.vodes[l].h.get(), ref["h"][l], r32["h"][l])
    assert np.array_equal(model.vodes[L + 2].h.get(), x)        # frozen sensory vode
    # linear-probe features: same loop from the same start, then GAP over tokens of vodes 1 and L+1
    for l, t in enumerate(case["h"]):
        model.vodes[l].h.set(t)
    f = ca.extract_features_from_batch(x, model, optim_h, T, [1, -2], True)
    want = np.concatenate([ref["h"][1], ref["h"][L + 1]], axis=-1).mean(axis=1)
    w32 = np.concatenate([r32["h"][1], r32["h"][L + 1]], axis=-1).mean(axis=1)
    assert f.shape == (B, 2 * D)
    _cmp(errs, floor, "features", f, want, w32)
    model.close()
    _check(name, errs, floor)


def test_unmask_status_init_flag():
    """status_init of the C entry point: h_l := u_l re-chained from the current h_0 before the loop."""
    import ctypes as C
    import cora_amd as ca
    from cora_amd import _lib
    from cora_amd.model import _fptr
    shape, p, D, H, L, B, T = SHAPES["fused_shape"]
    case = make_case(shape, p, D, H, L, B, seed=23)
    model, optim_w, optim_h = build_model(case)
    model._apply_recipes_h(optim_h)
    x = case["x"]
    eta = np.full(T, 0.095, dtype=np.float32)
    ref = R.unmask_on_batch(T, x, x, None, case["W"], case["h"], case["pos"], case["dims"], case["hp"], list(eta),
                            use_corruption=False, status_init=True)
    xhat = np.empty_like(x)
    _lib.check(_lib.load().cora_unmask(model._ctx, _fptr(x), _fptr(x), None, 0, 1, T, _fptr(eta), C.c_float(1.0),
                                       None, _fptr(xhat), None, None), model._ctx)
    sc = np.abs(ref["xhat"]).max()
    assert np.abs(xhat - ref["xhat"]).max() / sc <= 2e-4
    model.close()


def test_train_epoch_and_pretext_metrics():
    """`train` (:601-690) is the mean of per-batch train_on_batch scalars; `eval_pretext_metrics`
    (:502-598) recomputes its metrics from the last per-step reconstruction."""
    import cora_amd as ca
    shape, p, D, H, L, B, T = SHAPES["fused_shape"]
    case = make_case(shape, p, D, H, L, B, seed=31)
    rng = np.random.default_rng(4)
    batches = [(rng.uniform(0, 1, case["x"].shape).astype(np.float32), None) for _ in range(2)]
    model, optim_w, optim_h = build_model(case)
    avg_w, avg_mse, gh, gw = ca.train(batches, T, model=model, optim_w=optim_w, optim_h=optim_h, epoch=0)
    model2, ow2, oh2 = build_model(case)
    outs = [ca.train_on_batch(T, xb, model=model2, optim_w=ow2, optim_h=oh2) for xb, _ in batches]
    assert avg_w == pytest.approx(np.mean([o[1] for o in outs]), rel=1e-6)
    assert avg_mse == pytest.approx(np.mean([o[4] for o in outs]), rel=1e-6)
    # two contexts: float atomics in the reductions make runs agree to rounding, not bit for bit
    assert np.abs(gh[1] - outs[-1][2][1]).max() <= 1e-4 * np.abs(gh[1]).max()
    model.config.use_lower_half_mask = True
    m = ca.eval_pretext_metrics(batches, [T], True, 0.5, model=model, optim_h=optim_h)
    assert set(m) == {"full_mse", "masked_mse", "full_l1", "masked_l1", "full_psnr", "masked_psnr"}
    assert m["full_mse"] > 0 and m["masked_mse"] > 0 and m["full_psnr"] == pytest.approx(
        np.mean([10 * np.log10(1.0 / max(v, 1e-8)) for v in [m["full_mse"]]]), abs=1.0)
    model.close(); model2.close()


@pytest.mark.parametrize("name", ["fused_shape", "general_h2"])
def test_grad_norm_option_in_inference_loops(name):
    """`use_vode_grad_norm` in the two inference-only loops: unmask_on_batch_enhanced applies only the
    layer scaling (:925-933, no normalize_vode_h_gradients call), the linear-probe extractor scales AND
    normalises each leaf (/root/reference/examples/linear_probe.py:223-232).  With the option on the two
    walk different trajectories from the same start; each must match its own oracle run."""
    import cora_amd as ca
    shape, p, D, H, L, B, T = SHAPES[name]
    case = make_case(shape, p, D, H, L, B, seed=24, clip_h=2.0, use_grad_norm=True, grad_norm_target=1.5)
    model, optim_w, optim_h = build_model(case)
    x = case["x"]
    args = (T, x, x, None, case["W"], case["h"], case["pos"], case["dims"], case["hp"], [0.095] * T)
    errs, floor = {}, {}
    ref = R.unmask_on_batch(*args, use_corruption=False, normalise=False)
    r32 = R.unmask_on_batch(*args, use_corruption=False, normalise=False, dtype=torch.float32)
    loss, recons, _, log = ca.unmask_on_batch_enhanced(False, 0.0, [T], x, model=model, optim_h=optim_h,
                                                       log_inference_grads=True)
    _cmp(errs, floor, "unmask.loss", loss, ref["loss"], r32["loss"])
    for l in range(L + 2):
        _cmp(errs, floor, f"unmask.h{l}", model.vodes[l].h.get(), ref["h"][l], r32["h"][l])
    got = [log[T - 1][f"vode_{i}_grad_norm"] for i in range(L + 2)]      # norms of the SCALED, un-normalised leaves
    _cmp(errs, floor, "unmask.gnorm", got, ref["grad_norms"][T - 1][:L + 2], r32["grad_norms"][T - 1][:L + 2])
    h_unmask = model.vodes[L + 1].h.get()
    for l, t in enumerate(case["h"]):
        model.vodes[l].h.set(t)
    f = ca.extract_features_from_batch(x, model, optim_h, T, [1, -2], False)
    refn = R.unmask_on_batch(*args, use_corruption=False, normalise=True)
    r32n = R.unmask_on_batch(*args, use_corruption=False, normalise=True, dtype=torch.float32)
    want = np.concatenate([refn["h"][1], refn["h"][L + 1]], axis=-1)
    w32 = np.concatenate([r32n["h"][1], r32n["h"][L + 1]], axis=-1)
    _cmp(errs, floor, "features", f, want, w32)
    # the two loops really differ here, so a wrong flag cannot pass both comparisons
    assert np.abs(refn["h"][L + 1] - ref["h"][L + 1]).max() > 1e-3 * np.abs(ref["h"][L + 1]).max()
    assert np.abs(model.vodes[L + 1].h.get() - h_unmask).max() > 1e-3 * np.abs(h_unmask).max()
    model.close()
    _check(name, errs, floor)


@pytest.mark.parametrize("name", ["fused_shape", "general_h2"])
def test_unmask_update_weights_during_unmasking(name):
    """`update_weights_during_unmasking` (:967-979): every vode step of the loop is followed by dE/dW at
    the updated vodes and one clip + AdamW step scaled by 1 / x.shape[0], before the clamp.  Vodes,
    reconstructions and loss to the fp32 bar; weights compared where the fp64 gradient is not
    noise-sized, in units of the total AdamW move (see test_ipc_mode... in test_gpu_parity.py)."""
    import cora_amd as ca
    shape, p, D, H, L, B, T = SHAPES[name]
    case = make_case(shape, p, D, H, L, B, seed=25, clip_h=3.0)
    model, optim_w, optim_h = build_model(case)
    model.config.update_weights_during_unmasking = True
    model.config.use_lower_half_mask = True
    model.config.inference_clamp_alpha = 0.7
    x = case["x"]
    mask, x_init = ca.make_mask(model, x, 0.5)
    args = (T, x, x_init, mask, case["W"], case["h"], case["pos"], case["dims"], case["hp"], [0.095] * T)
    kw = dict(alpha=0.7, use_corruption=True, eta_w=[1e-3] * T)
    ref = R.unmask_on_batch(*args, **kw)
    r32 = R.unmask_on_batch(*args, dtype=torch.float32, **kw)
    loss, recons, _, _ = ca.unmask_on_batch_enhanced(True, 0.5, [T], x, model=model, optim_h=optim_h, optim_w=optim_w)
    assert optim_w.count == T and model.get_adam_state()[2] == T
    errs, floor = {}, {}
    for t in range(T):
        _cmp(errs, floor, f"recon{t}", recons[t], ref["recons"][t], r32["recons"][t])
    _cmp(errs, floor, "loss", loss, ref["loss"], r32["loss"])
    for l in range(L + 2):
        _cmp(errs, floor, f"h{l}", model.vodes[l].h.get(), ref["h"][l], r32["h"][l])
    _cmp(errs, floor, "sensory", model.vodes[L + 2].h.get(), ref["sensory"], r32["sensory"])
    neww = model.get_weights()
    moved = 0
    for k, v in ref["weights"].items():
        step = np.abs(v - case["W"][k].astype(np.float64))
        sel = step > 0.5 * T * 1e-3            # entries AdamW moved at (nearly) the full rate every step: gradient not noise-sized
        if sel.any():
            moved += int(sel.sum())
            d = np.abs(neww[k].astype(np.float64) - v)[sel].max() / (T * 1e-3)
            d32 = np.abs(r32["weights"][k].astype(np.float64) - v)[sel].max() / (T * 1e-3)
            assert d <= max(2e-2, 3.0 * d32), (k, d, d32)
        assert np.abs(neww[k] - case["W"][k]).max() > 0, k       # every tensor did move
    assert moved > 100
    model.close()
    _check(name, errs, floor)
