"""Oracle checks that run on CPU: the three known-answer facts the reference's own tests pin,
autograd-vs-hand-VJP agreement of the two restatements, and the committed golden vectors."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import pcvit_np as NP
from oracle import pcvit_ref as R
from tests.golden.make_golden import GOLDEN, generate
from tests.util import make_case

HERE = os.path.dirname(os.path.abspath(__file__))


def test_se_energy_known_answer():
    # /root/reference/tests/pcx/test_vode.py:55-75: h=[1,2,3], u=[1.5,2.5,3.5] -> 0.5*sum(0.25)=0.375
    h = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64)
    u = torch.tensor([1.5, 2.5, 3.5], dtype=torch.float64)
    assert float(R.se_energy(h, u).sum()) == pytest.approx(0.375)
    assert float(R.regularized_plus_se_energy(h, u, 0.0, 0.0).sum()) == pytest.approx(0.375)


def test_integration_energy_is_half_sum_sq():
    # /root/reference/tests/pcx/test_integration.py:130-131: energy == 0.5*sum((h-u)^2)
    rng = np.random.default_rng(0)
    h, u = torch.tensor(rng.standard_normal((4, 5))), torch.tensor(rng.standard_normal((4, 5)))
    assert float(R.se_energy(h, u).sum()) == pytest.approx(0.5 * float(((h - u) ** 2).sum()))


def test_linear_known_answer():
    # /root/reference/tests/pcx/test_nn.py:25-48 with the fixture of tests/conftest.py:100-106
    W = torch.tensor([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]], dtype=torch.float64)
    b = torch.tensor([0.1, 0.2], dtype=torch.float64)
    x = torch.tensor([0.5, 1.0, 1.5], dtype=torch.float64)
    y = R.linear(x, W, b).numpy()
    np.testing.assert_allclose(y, [0.5 + 2.0 + 4.5 + 0.1, 2.0 + 5.0 + 9.0 + 0.2], rtol=1e-12)


def test_layernorm_reference_properties():
    # /root/reference/tests/pcx/test_nn.py:135-165: mean(out) ~ 0 with (1,0); mean(out) ~ 1 with (2,1)
    x = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0], dtype=torch.float64)
    out = R.layer_norm(x, torch.ones(5, dtype=torch.float64), torch.zeros(5, dtype=torch.float64))
    assert abs(float(out.mean())) < 1e-5
    out = R.layer_norm(x, 2 * torch.ones(5, dtype=torch.float64), torch.ones(5, dtype=torch.float64))
    assert abs(float(out.mean()) - 1.0) < 1e-5
    xn, xh, rstd = NP.ln_fwd(x.numpy(), 2 * np.ones(5), np.ones(5))
    np.testing.assert_allclose(xn, out.numpy(), rtol=1e-12)


def test_gelu_is_tanh_form():
    a = torch.linspace(-4, 4, 33, dtype=torch.float64)
    want = torch.nn.functional.gelu(a, approximate="tanh")
    np.testing.assert_allclose(R.gelu_tanh(a).numpy(), want.numpy(), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(NP.gelu(a.numpy()), want.numpy(), rtol=1e-12, atol=1e-15)


def test_patchify_roundtrip_and_channel_innermost():
    dims = R.Dims((3, 8, 8), 4, 16, 1, 0)
    img = torch.arange(2 * 3 * 8 * 8, dtype=torch.float64).reshape(2, 3, 8, 8)
    xp = R.patchify(img, dims)
    assert xp.shape == (2, 4, 48)
    np.testing.assert_array_equal(R.unpatchify(xp, dims).numpy(), img.numpy())
    # /root/reference/src/decoder_transformer.py:277-292: pixel(c, i*p+a, j*p+b) = z[i*wp+j, (a*p+b)*C + c]
    b, c, i, j, a, bb = 1, 2, 1, 0, 3, 2
    assert float(xp[b, i * 2 + j, (a * 4 + bb) * 3 + c]) == float(img[b, c, i * 4 + a, j * 4 + bb])
    np.testing.assert_array_equal(NP.patchify(img.numpy(), dims), xp.numpy())


@pytest.mark.parametrize("shape,p,D,H,L,B,l1,l2,clip", [
    ((3, 16, 16), 4, 32, 2, 2, 3, 0.0, 0.0, 2000.0),
    ((3, 16, 16), 4, 32, 1, 1, 2, 1e-3, 1e-3, 5.0),
    ((2, 3, 8, 8), 4, 24, 2, 1, 2, 0.0, 0.0, 2000.0),
    ((3, 8, 8), 4, 16, 1, 0, 2, 0.0, 0.0, 2000.0),
])
def test_autograd_matches_hand_vjp(shape, p, D, H, L, B, l1, l2, clip):
    case = make_case(shape, p, D, H, L, B, seed=3, l1=l1, l2=l2, clip_h=clip)
    W = {k: v.astype(np.float64) for k, v in case["W"].items()}
    h = [t.astype(np.float64) for t in case["h"]]
    T = 3
    eta = [0.095] * T
    out = R.train_on_batch(T, case["x"], W, h, case["pos"], case["dims"], case["hp"], eta, 1e-3)
    xp = NP.patchify(case["x"].astype(np.float64), case["dims"])
    h2, en, pl, g = NP.infer(T, xp, W, h, case["pos"].astype(np.float64), case["dims"], case["hp"], eta)
    np.testing.assert_allclose(en, out["energies"], rtol=1e-12)
    np.testing.assert_allclose(pl, out["layer_energies"], rtol=1e-11, atol=1e-12)
    for a, b in zip(h2, out["h"]):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-11)
    for a, b in zip(g, out["final_h_grad"]):
        np.testing.assert_allclose(a, b, rtol=1e-8, atol=1e-10)
    E, Es, gg, wg, z = NP.step_grads(h2, W, case["pos"].astype(np.float64), xp, case["dims"], case["hp"], want_wgrad=True)
    assert E == pytest.approx(out["final_w_energy"], rel=1e-12)
    assert set(wg) == set(out["w_grad"])
    for k in wg:
        np.testing.assert_allclose(wg[k], out["w_grad"][k], rtol=1e-8, atol=1e-10)


def test_vode_layernorm_weight_gradient_by_finite_differences():
    """use_vode_state_layernorm (:212-218, 232-241): the oracle's dE/dW of the extra LayerNorms against a
    central difference of its own energy at the final vodes (no hand VJP exists for this option)."""
    case = make_case((3, 16, 16), 4, 32, 2, 2, 2, seed=4, vode_ln=True)
    W = {k: v.astype(np.float64) for k, v in case["W"].items()}
    h = [t.astype(np.float64) for t in case["h"]]
    out = R.train_on_batch(2, case["x"], W, h, case["pos"], case["dims"], case["hp"], [0.095] * 2, 1e-3)
    assert sum("vode_output_layernorms" in k for k in out["w_grad"]) == 6
    E0 = out["final_w_energy"]
    for k in (".vode_output_layernorms[0].nn.weight", ".vode_output_layernorms[1].nn.bias", ".vode_output_layernorms[2].nn.weight"):
        for j in (0, 17):
            Wp = {a: b.copy() for a, b in W.items()}; Wm = {a: b.copy() for a, b in W.items()}
            Wp[k][j] += 1e-6; Wm[k][j] -= 1e-6
            Ep = R.energy_and_prediction(out["h"], Wp, case["pos"], case["x"], case["dims"], case["hp"])[0]
            Em = R.energy_and_prediction(out["h"], Wm, case["pos"], case["x"], case["dims"], case["hp"])[0]
            assert (Ep - Em) / 2e-6 == pytest.approx(out["w_grad"][k][j], rel=1e-5, abs=1e-7)
    assert E0 == pytest.approx(R.energy_and_prediction(out["h"], W, case["pos"], case["x"], case["dims"], case["hp"])[0], rel=1e-12)


def test_layer_scaling_exponents():
    # /root/reference/src/decoder_transformer.py:321-341: factor base^{(L+1) - l}
    dims = R.Dims((3, 8, 8), 4, 16, 1, 2)
    hp = R.Hyper(scale_base=2.0)
    g = [torch.ones(1, dtype=torch.float64) for _ in range(4)]
    out = R.scale_h_grads(g, dims, hp)
    assert [float(t) for t in out] == [8.0, 4.0, 2.0, 1.0]


def test_clip_by_global_norm_semantics():
    g = [torch.tensor([3.0]), torch.tensor([4.0])]
    assert [float(t) for t in R._clip(g, 10.0)] == [3.0, 4.0]          # norm 5 < 10: untouched
    out = R._clip(g, 2.5)
    assert float(out[0]) == pytest.approx(1.5) and float(out[1]) == pytest.approx(2.0)


def test_lr_schedule_shape():
    f = R.lr_schedule(0.1, 10, 110, 0.5)
    assert f(0) == 0.0 and f(5) == pytest.approx(0.05) and f(10) == pytest.approx(0.1)
    assert f(110) == pytest.approx(0.05) and f(60) == pytest.approx(0.075)
    g = R.lr_schedule(0.1, 0, 100, 0.5)
    assert g(0) == pytest.approx(0.1)


@pytest.mark.parametrize("name", list(GOLDEN))
def test_oracle_reproduces_golden(name):
    path = os.path.join(HERE, "golden", name + ".npz")
    assert os.path.exists(path), "run tests/golden/make_golden.py"
    want = np.load(path)
    got = generate(name)
    assert set(got) == set(want.files)
    for k in want.files:
        np.testing.assert_allclose(got[k], want[k], rtol=1e-6, atol=1e-9, err_msg=k)


def test_golden_files_all_listed():
    files = {os.path.basename(f)[:-4] for f in glob.glob(os.path.join(HERE, "golden", "*.npz"))}
    assert files == set(GOLDEN)


def test_unmask_oracle_consistency():
    """Inference-only loop restatement (src/decoder_transformer.py:833-1046): without corruption it
    is the T-step loop of train_on_batch with the sensory vode frozen; with a hard clamp (alpha = 1)
    the known pixels equal the ground truth after every step and the un-frozen sensory vode adds
    one gradient leaf scaled by base^-1."""
    from tests.util import make_case
    case = make_case((3, 16, 16), 4, 32, 2, 2, 2, seed=1)
    x = case["x"]
    T = 3
    common = (case["W"], case["h"], case["pos"], case["dims"], case["hp"], [0.095] * T)
    r0 = R.unmask_on_batch(T, x, x, None, *common, use_corruption=False)
    tr = R.train_on_batch(T, x, case["W"], case["h"], case["pos"], case["dims"], case["hp"], [0.095] * T, 1e-3)
    for a, b in zip(r0["h"], tr["h"]):
        assert np.array_equal(a, b)
    assert np.array_equal(r0["sensory"], x.astype(np.float64))
    m = np.zeros((2, 1, 16, 16), np.float32)
    m[:, :, 8:] = 1
    xi = np.where(m == 1, 0.5, x).astype(np.float32)
    r1 = R.unmask_on_batch(T, x, xi, m, *common, alpha=1.0)
    mb = np.broadcast_to(m, x.shape)
    assert np.array_equal(r1["sensory"][mb == 0], x.astype(np.float64)[mb == 0])
    assert r1["grad_norms"].shape == (T, case["dims"].L + 3) and (r1["grad_norms"][:, -1] > 0).all()
    assert len(r1["recons"]) == T and r1["loss"] > 0
    # first-step sensory gradient norm = ||x_init - u|| / base  (layer scale base^{(L+1)-(L+2)})
    us = R.predictions([torch.as_tensor(t, dtype=torch.float64) for t in case["h"]],
                       {k: torch.as_tensor(v, dtype=torch.float64) for k, v in case["W"].items()},
                       torch.as_tensor(case["pos"], dtype=torch.float64), case["dims"])
    want = float(torch.sqrt(((torch.as_tensor(xi, dtype=torch.float64) - us[-1]) ** 2).sum())) / case["hp"].scale_base
    assert abs(r1["grad_norms"][0, -1] - want) <= 1e-9 * want
