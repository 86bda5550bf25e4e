"""npz checkpoint round trip on the GPU: `pxu.save_params(model, path,
filter=LayerParam | VodeParam)` as the reference driver calls it
(/root/reference/examples/debug_transformer_wandb.py:1278,1289) -> fresh model -> `load_params`
(/root/reference/pcx/utils/_serialisation.py:63-105) must reproduce weights and vodes bit for bit, and
the next train_on_batch step must be the one the original model takes."""
import numpy as np
import pytest

from tests.test_host import _reference_key_set
from tests.util import build_model, make_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape_name", ["fused", "general_vln"])
def test_npz_round_trip_weights_vodes_and_next_step(tmp_path, shape_name):
    import cora_amd as ca
    if shape_name == "fused":
        case = make_case((3, 32, 32), 4, 64, 1, 2, 3, seed=61)
    else:
        case = make_case((3, 16, 16), 4, 32, 2, 2, 4, seed=62, vode_ln=True)
    T, L = 3, case["dims"].L
    model, ow, oh = build_model(case)
    ca.train_on_batch(T, case["x"], model=model, optim_w=ow, optim_h=oh)      # a trained state, sensory vode set
    path = str(tmp_path / "model_epoch1")
    ca.save_params(model, path, filter=ca.LayerParam | ca.VodeParam)
    with np.load(path + ".npz") as z:
        assert set(z.files) == _reference_key_set(L, vode_ln=(shape_name == "general_vln"))
        assert z[".vodes[0].h"].shape == (case["x"].shape[0], case["dims"].N, case["dims"].P)
        np.testing.assert_array_equal(z[f".vodes[{L + 2}].h"], case["x"])     # the sensory vode holds the last batch
    w0 = model.get_weights()
    h0 = [model.vodes[l].h.get() for l in range(L + 3)]

    # fresh model of the same config, different seed; vodes sized by an INIT forward as in the reference
    fresh = ca.TransformerDecoder(model.config, seed=1234)
    fresh.positional_embedding = case["pos"]
    fresh.init_vodes(case["x"].shape[0])
    ca.load_params(fresh, path, filter=ca.LayerParam | ca.VodeParam)
    w1 = fresh.get_weights()
    for k, v in w0.items():
        np.testing.assert_array_equal(w1[k], v, err_msg=k)
    for l in range(L + 3):
        np.testing.assert_array_equal(fresh.vodes[l].h.get(), h0[l], err_msg=f"vode {l}")

    # the next step from the restored state == the next step of the original.  The optimizer state is not
    # part of the checkpoint, so both sides start from fresh optimizers.
    rng = np.random.default_rng(5)
    x2 = rng.uniform(0, 1, case["x"].shape).astype(np.float32)
    _, ow_a, oh_a = build_optims(ca, case, model)
    _, ow_b, oh_b = build_optims(ca, case, fresh)
    a = ca.train_on_batch(T, x2, model=model, optim_w=ow_a, optim_h=oh_a)
    b = ca.train_on_batch(T, x2, model=fresh, optim_w=ow_b, optim_h=oh_b)
    # float atomics in the energy / split-K reductions: equal to rounding, not bit for bit
    assert a[0] == pytest.approx(b[0], rel=1e-6) and a[1] == pytest.approx(b[1], rel=1e-6) and a[4] == pytest.approx(b[4], rel=1e-6)
    for l in range(L + 2):
        x, y = model.vodes[l].h.get(), fresh.vodes[l].h.get()
        assert np.abs(x - y).max() <= 1e-6 * np.abs(x).max(), l
    wa, wb = model.get_weights(), fresh.get_weights()
    for k in wa:
        assert np.abs(wa[k] - wb[k]).max() <= 1e-5 * max(np.abs(wa[k]).max(), 1e-3), k
    model.close(); fresh.close()


def build_optims(ca, case, model):
    hp = case["hp"]
    oh = ca.Optim(lambda: ca.chain(ca.clip_by_global_norm(hp.clip_h), ca.sgd(0.095, momentum=hp.momentum)))
    ow = ca.Optim(lambda: ca.chain(ca.clip_by_global_norm(hp.clip_w), ca.adamw(1e-3, weight_decay=hp.wd)), model)
    return model, ow, oh


def test_optim_w_init_resets_device_adam_state():
    """pxu.Optim.init re-creates the whole optax state (/root/reference/pcx/utils/_optim.py:150-163): for
    the AdamW weight optimizer that is mu, nu AND the bias-correction count."""
    import cora_amd as ca
    case = make_case((3, 16, 16), 4, 32, 2, 1, 2, seed=63)
    model, ow, oh = build_model(case)
    ca.train_on_batch(2, case["x"], model=model, optim_w=ow, optim_h=oh)
    mu, nu, cnt = model.get_adam_state()
    assert cnt == 1 and ow.count == 1 and np.abs(mu).max() > 0
    ow.init(model)
    mu, nu, cnt = model.get_adam_state()
    assert cnt == 0 and ow.count == 0 and not mu.any() and not nu.any()
    model.close()
