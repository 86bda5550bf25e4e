"""The oracle's restatements of the third-party arithmetic on the path (equinox 0.11.8 layers, optax 0.2.3
optimizers -- neither is importable here) cross-checked against torch's own, independently
written implementations of the same published definitions.  This does not pin parity with the reference
(nothing short of running JAX could); it removes the risk that oracle/pcvit_ref.py and its hand-VJP twin
share one misreading of a formula.  CPU only.

  eqx.nn.MultiheadAttention (pcx/nn/_layer.py:359-402)      <-> F.scaled_dot_product_attention, nn.MultiheadAttention
  eqx.nn.LayerNorm (pcx/nn/_layer.py:296-315)               <-> F.layer_norm
  pxnn.TransformerBlock (pcx/nn/_layer.py:428-447)          <-> nn.LayerNorm / nn.MultiheadAttention / nn.Linear / nn.GELU("tanh")
  optax.chain(clip_by_global_norm, adamw) (examples/debug_transformer_wandb.py:1153-1159) <-> torch.optim.AdamW
  optax.chain(clip_by_global_norm, sgd(momentum)) (:1134-1140)                            <-> torch.optim.SGD(momentum)
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pcvit_ref as R
from tests.util import make_case

torch.set_default_dtype(torch.float32)
DT = torch.float64


def _rand(rng, *shape):
    return torch.as_tensor(rng.standard_normal(shape), dtype=DT)


@pytest.mark.parametrize("N,D,H", [(16, 32, 1), (36, 32, 2), (64, 64, 4), (20, 96, 4)])
def test_mha_matches_torch_sdpa_and_module(N, D, H):
    rng = np.random.default_rng(N + D + H)
    x = _rand(rng, 3, N, D)
    Wq, Wk, Wv, Wo = (_rand(rng, D, D) / math.sqrt(D) for _ in range(4))
    got = R.mha(x, Wq, Wk, Wv, Wo, H)
    # (a) torch's fused attention on explicitly projected heads: softmax(q k^T / sqrt(d)) v
    d = D // H
    q, k, v = ((x @ W.T).reshape(3, N, H, d).transpose(1, 2) for W in (Wq, Wk, Wv))
    o = F.scaled_dot_product_attention(q, k, v)            # default scale 1/sqrt(d), no mask, no dropout
    want = o.transpose(1, 2).reshape(3, N, D) @ Wo.T
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-10, atol=1e-12)
    # (b) torch's module with the same projections (head h = columns h*d .. (h+1)*d of the projected q/k/v)
    m = torch.nn.MultiheadAttention(D, H, bias=False, batch_first=True, dtype=DT)
    with torch.no_grad():
        m.in_proj_weight.copy_(torch.cat([Wq, Wk, Wv], 0))
        m.out_proj.weight.copy_(Wo)
        want2, _ = m(x, x, x, need_weights=False)
    np.testing.assert_allclose(got.numpy(), want2.numpy(), rtol=1e-9, atol=1e-11)


def test_layer_norm_matches_torch():
    rng = np.random.default_rng(1)
    x = _rand(rng, 5, 7, 48) * 3 + 1
    g, b = _rand(rng, 48), _rand(rng, 48)
    want = F.layer_norm(x, (48,), g, b, eps=1e-5)           # biased variance, eps inside the sqrt
    np.testing.assert_allclose(R.layer_norm(x, g, b).numpy(), want.numpy(), rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("D,H", [(32, 2), (64, 1)])
def test_transformer_block_matches_torch_modules(D, H):
    """y = x + MHA(LN1 x); out = y + W2 gelu_tanh(W1 LN2 y + b1) + b2  (pcx/nn/_layer.py:428-447), and the
    decoder's extra residual u = x + block(x) (src/decoder_transformer.py:238), built from torch.nn modules."""
    case = make_case((3, 16, 16), 4, D, H, 1, 2, seed=2)
    W = {k: torch.as_tensor(v, dtype=DT) for k, v in case["W"].items()}
    k = R.block_keys(0)
    x = torch.as_tensor(case["h"][1], dtype=DT)
    got = R.transformer_block(x, W, 0, H)
    ln1, ln2 = torch.nn.LayerNorm(D, eps=1e-5, dtype=DT), torch.nn.LayerNorm(D, eps=1e-5, dtype=DT)
    att = torch.nn.MultiheadAttention(D, H, bias=False, batch_first=True, dtype=DT)
    l1, l2 = torch.nn.Linear(D, 4 * D, dtype=DT), torch.nn.Linear(4 * D, D, dtype=DT)
    act = torch.nn.GELU(approximate="tanh")
    with torch.no_grad():
        ln1.weight.copy_(W[k["ln1_g"]]); ln1.bias.copy_(W[k["ln1_b"]])
        ln2.weight.copy_(W[k["ln2_g"]]); ln2.bias.copy_(W[k["ln2_b"]])
        att.in_proj_weight.copy_(torch.cat([W[k["wq"]], W[k["wk"]], W[k["wv"]]], 0)); att.out_proj.weight.copy_(W[k["wo"]])
        l1.weight.copy_(W[k["w1"]]); l1.bias.copy_(W[k["b1"]]); l2.weight.copy_(W[k["w2"]]); l2.bias.copy_(W[k["b2"]])
        xn = ln1(x)
        y = x + att(xn, xn, xn, need_weights=False)[0]
        want = y + l2(act(l1(ln2(y))))
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-9, atol=1e-11)
    us = R.predictions([torch.as_tensor(t, dtype=DT) for t in case["h"]], W, torch.as_tensor(case["pos"], dtype=DT), case["dims"])
    np.testing.assert_allclose(us[1].numpy(), (x + want).numpy(), rtol=1e-9, atol=1e-11)     # double residual


def test_clip_matches_torch_up_to_its_epsilon():
    """optax.clip_by_global_norm scales by max_norm / norm (no epsilon) when norm >= max_norm; torch's
    clip_grad_norm_ uses max_norm / (norm + 1e-6).  The oracle keeps optax's form; the two agree to 1e-6 / norm."""
    rng = np.random.default_rng(3)
    gs = [_rand(rng, 4, 5), _rand(rng, 7)]
    ps = [torch.nn.Parameter(torch.zeros_like(g)) for g in gs]
    for p, g in zip(ps, gs):
        p.grad = g.clone()
    n = float(R._global_norm(gs))
    total = torch.nn.utils.clip_grad_norm_(ps, max_norm=0.5 * n)
    assert float(total) == pytest.approx(n, rel=1e-12)
    for p, o in zip(ps, R._clip(gs, 0.5 * n)):
        np.testing.assert_allclose(p.grad.numpy(), o.numpy(), rtol=2e-6 / n + 1e-12)
    for p, g in zip(ps, gs):
        p.grad = g.clone()
    torch.nn.utils.clip_grad_norm_(ps, max_norm=2.0 * n)               # below the threshold: untouched
    for p, g, o in zip(ps, gs, R._clip(gs, 2.0 * n)):
        assert torch.equal(p.grad, g) and torch.equal(o, g)


def test_clip_plus_adamw_matches_torch_optim_over_two_steps():
    """The oracle's weight step (clip_by_global_norm(c_w) -> adamw(lr, b1, b2, eps, wd); scale_by = 1/B,
    src/decoder_transformer.py:484) against torch.optim.AdamW fed the same clipped gradient, for two
    consecutive batches (bias corrections at count 1 and 2, decoupled weight decay multiplied by lr)."""
    case = make_case((3, 16, 16), 4, 32, 2, 2, 3, seed=9, clip_w=0.05)     # low threshold: the clip fires
    hp, B = case["hp"], case["x"].shape[0]
    W64 = {k: v.astype(np.float64) for k, v in case["W"].items()}
    params = {k: torch.nn.Parameter(torch.as_tensor(v, dtype=DT)) for k, v in W64.items()}
    opt = torch.optim.AdamW(list(params.values()), lr=1e-3, betas=(hp.b1, hp.b2), eps=hp.eps, weight_decay=hp.wd)
    adam, W, h = None, W64, [t.astype(np.float64) for t in case["h"]]
    rng = np.random.default_rng(10)
    for step in range(2):
        x = rng.uniform(0, 1, case["x"].shape).astype(np.float32)
        out = R.train_on_batch(2, x, W, h, case["pos"], case["dims"], hp, [0.095] * 2, 1e-3, adam=adam)
        g = [torch.as_tensor(out["w_grad"][k], dtype=DT) / B for k in params]
        assert float(R._global_norm(g)) > hp.clip_w
        for p, gk in zip(params.values(), R._clip(g, hp.clip_w)):
            p.grad = gk.clone()
        opt.step()
        for k, p in params.items():
            np.testing.assert_allclose(out["weights"][k], p.detach().numpy(), rtol=1e-9, atol=1e-12, err_msg=f"step {step} {k}")
        adam, W, h = out["adam"], out["weights"], out["h"]
    assert adam.count == 2


def test_momentum_sgd_trajectory_matches_torch_optim():
    """The oracle's vode step (scale -> clip_by_global_norm(c_h) -> optax.sgd(momentum): m = g + mu m, h -= eta m;
    state zeroed at the start of every batch, :414) against torch.optim.SGD(momentum=mu, dampening=0) driven
    with the same scaled + clipped gradients, T = 4 steps with a per-step learning rate."""
    case = make_case((3, 16, 16), 4, 32, 2, 2, 2, seed=12, clip_h=3.0, momentum=0.4)
    hp, dims = case["hp"], case["dims"]
    T, eta = 4, [0.095, 0.08, 0.07, 0.05]
    out = R.train_on_batch(T, case["x"], case["W"], case["h"], case["pos"], dims, hp, eta, 1e-3, update_weights=False)
    W = {k: torch.as_tensor(v, dtype=DT) for k, v in case["W"].items()}
    pos, y = torch.as_tensor(case["pos"], dtype=DT), torch.as_tensor(case["x"], dtype=DT)
    hs = [torch.nn.Parameter(torch.as_tensor(t, dtype=DT)) for t in case["h"]]
    opt = torch.optim.SGD(hs, lr=eta[0], momentum=hp.momentum, dampening=0.0, nesterov=False)
    fired = 0
    for t in range(T):
        _, _, grads, _ = R.h_gradients([p.detach() for p in hs], W, pos, y, dims, hp)
        g = R.scale_h_grads(grads, dims, hp)
        fired += int(float(R._global_norm(g)) >= hp.clip_h)
        for p, gk in zip(hs, R._clip(g, hp.clip_h)):
            p.grad = gk.clone()
        for grp in opt.param_groups:
            grp["lr"] = eta[t]
        opt.step()
    assert fired > 0
    for l, p in enumerate(hs):
        np.testing.assert_allclose(out["h"][l], p.detach().numpy(), rtol=1e-10, atol=1e-12, err_msg=f"vode {l}")
