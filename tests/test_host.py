"""Host logic that needs no GPU: configs, positional encodings, optimizer recipes, parameter
inventory / npz key names, ABI symbol table."""
import os
import re

import numpy as np
import pytest

import cora_amd as ca
from cora_amd import _lib
from cora_amd.params import init_params, param_shapes
from cora_amd.posenc import create_positional_encoding, get_sinusoidal_1d

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_transformer_config_derived_dims():
    c = ca.TransformerConfig(image_shape=(3, 32, 32), hidden_size=64, num_heads=1, num_blocks=6, patch_size=4)
    assert (c.num_patches, c.patch_dim, c.mlp_hidden_dim, c.is_video) == (64, 48, 256, False)
    assert c.axes_dim == [16, 16]
    v = ca.TransformerConfig(image_shape=(16, 3, 32, 32), hidden_size=64, num_heads=1, num_blocks=6, patch_size=4)
    assert (v.num_patches, v.patch_dim, v.is_video, v.num_frames) == (1024, 48, True, 16)
    i = ca.TransformerConfig(image_shape=(3, 224, 224), hidden_size=384, num_heads=8, num_blocks=6, patch_size=16)
    assert (i.num_patches, i.patch_dim, i.mlp_hidden_dim) == (196, 768, 1536)


def test_named_configs_match_reference_values():
    m = ca.MODEL_CONFIGS
    assert set(["debug_tiny", "0block", "1block", "2block", "5block", "6block"]) <= set(m)
    t = m["debug_tiny"]   # /root/reference/src/config.py:121-137
    assert (t.hidden_size, t.num_heads, t.num_blocks, t.peak_lr_hidden, t.inference_lr_scale_base,
            t.h_grad_clip_norm, t.hidden_momentum, t.theta, t.use_lr_schedule_h) == (64, 1, 5, 0.07, 1.1, 1000.0, 0.1, 100, True)
    s = m["6block"]       # /root/reference/src/config.py:442-517
    assert (s.hidden_size, s.num_heads, s.num_blocks, s.peak_lr_hidden, s.hidden_momentum, s.theta,
            s.inference_lr_scale_base, s.h_grad_clip_norm, s.w_grad_clip_norm, s.peak_lr_weights,
            s.weight_decay, s.use_lr_schedule_h, s.use_lr_schedule_w, s.batch_size, s.inference_steps) == \
        (64, 1, 6, 0.095, 0.4, 10_000, 1.25, 2000.0, 500.0, 0.001, 2e-4, False, False, 200, 20)
    f = m["5block"]       # /root/reference/src/config.py:366-441
    assert (f.h_grad_clip_norm, f.hidden_momentum, f.hidden_lr_inference, f.theta) == (5000.0, 0.3, 0.1, 100)
    tw = m["12block"]     # sweeps/sweep_12block_comprehensive.yaml
    assert (tw.num_blocks, tw.hidden_size, tw.num_heads, tw.batch_size) == (12, 128, 4, 200)
    for name in ("0block", "1block", "2block"):
        assert m[name].num_blocks == int(name[0]) and m[name].use_lr_schedule_h is True


def test_create_config_signature_and_errors():
    c = ca.create_config(hidden_size=64, num_blocks=6, num_heads=1, use_inference_lr_scaling=True,
                         inference_lr_scale_base=1.25, use_status_init_in_training=False)
    assert c.image_shape == (3, 32, 32) and c.num_patches == 64 and c.use_inference_lr_scaling
    with pytest.raises(ValueError):
        ca.create_config(dataset="imagenet")
    tc = ca.transformer_config_from(ca.MODEL_CONFIGS["6block"])
    assert (tc.hidden_size, tc.num_blocks, tc.theta, tc.use_status_init_in_training) == (64, 6, 10_000, False)


def test_positional_encoding_2d():
    pe = create_positional_encoding((3, 32, 32), 64, 4, theta=10000)
    assert pe.shape == (64, 64) and pe.dtype == np.float32
    # token n = i*8 + j: first half encodes the row i, second half the column j (src/utils.py:64-75)
    np.testing.assert_array_equal(pe[0, :32], pe[7, :32])
    np.testing.assert_array_equal(pe[3, 32:], pe[8 * 5 + 3, 32:])
    # position 0 -> sin 0 = 0 on even, cos 0 = 1 on odd indices
    np.testing.assert_allclose(pe[0, 0::2], 0.0, atol=1e-7)
    np.testing.assert_allclose(pe[0, 1::2], 1.0, atol=1e-7)
    one = get_sinusoidal_1d(np.arange(4), 8, theta=100.0)
    k = np.arange(0, 8, 2)
    np.testing.assert_allclose(one[:, 0::2], np.sin(np.arange(4)[:, None] * np.exp(-k * np.log(100.0) / 8)), rtol=1e-5, atol=1e-6)


def test_positional_encoding_3d_pads():
    pe = create_positional_encoding((16, 3, 32, 32), 64, 4, theta=10000)
    assert pe.shape == (1024, 64)
    # 64 // 3 = 21 -> 20 per axis, 60 used, 4 zero-padded (src/utils.py:130-141, :21)
    np.testing.assert_array_equal(pe[:, 60:], 0.0)


def test_optimizer_recipes():
    oh = ca.Optim(lambda: ca.chain(ca.clip_by_global_norm(2000.0), ca.sgd(0.095, momentum=0.4)))
    r = oh._recipe()
    assert r.clip == 2000.0 and r.base.momentum == 0.4
    with pytest.raises(ValueError):
        ca.chain(ca.sgd(0.1), ca.clip_by_global_norm(1.0))
    with pytest.raises(NotImplementedError):
        oh.step(None, None)
    ow, oh2 = ca.make_optimizers(ca.MODEL_CONFIGS["debug_tiny"], total_train_steps=100)
    assert ow._recipe().base.weight_decay == 2e-4 and callable(oh2._recipe().base.learning_rate)
    sched = ca.create_learning_rate_schedule(0.1, 10, 110, 0.5)
    assert sched(0) == 0.0 and sched(10) == pytest.approx(0.1) and sched(110) == pytest.approx(0.05)


def test_param_inventory_and_pcx_keys():
    shapes = dict(param_shapes(48, 64, 256, 2))
    assert shapes[".transformer_blocks[1].attention.nn.query_proj.weight"] == (64, 64)
    assert shapes[".transformer_blocks[0].linear1.nn.weight"] == (256, 64)
    assert shapes[".patch_projection.nn.weight"] == (64, 48) and shapes[".output_projection.nn.bias"] == (48,)
    assert not any("proj.bias" in k for k in shapes)        # attention has no biases
    W = init_params(48, 64, 256, 2, seed=1)
    assert set(W) == set(shapes)
    assert np.all(W[".transformer_blocks[0].layer_norm1.nn.weight"] == 1.0)
    lim = 1 / np.sqrt(64)
    assert np.abs(W[".transformer_blocks[0].linear1.nn.weight"]).max() <= lim
    n = sum(int(np.prod(s)) for s in shapes.values())
    assert n == 2 * (4 * 64 + 4 * 64 * 64 + 2 * 256 * 64 + 256 + 64) + 64 * 48 + 64 + 48 * 64 + 48


def test_abi_library_loads_and_exports_every_declared_symbol():
    """The C-ABI library must load on a CPU-only host and export what include/cora_hip.h declares."""
    hdr = open(os.path.join(ROOT, "include", "cora_hip.h")).read()
    declared = set(re.findall(r"\b(cora_[a-z_]+)\s*\(", hdr))
    declared -= {"cora_get_vode_grads"}   # mentioned in a comment only
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libcora_hip.so does not export {name}"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))


def test_create_fails_loudly_without_gpu_or_on_bad_config():
    import ctypes as C
    import torch
    lib = _lib.load()
    cfg = _lib.CoraConfig(N=4, P=48, D=16, H=1, M=64, L=1, B=2, C=3, patch=4, Himg=8, Wimg=8, F=1, dtype_mode=7)
    ctx = C.c_void_p()
    assert lib.cora_create(C.byref(cfg), 0, C.byref(ctx)) != 0
    assert b"dtype_mode" in lib.cora_last_error(None)
    cfg.dtype_mode = 0
    cfg.D = 18    # not a multiple of 4
    assert lib.cora_create(C.byref(cfg), 0, C.byref(ctx)) != 0
    if not torch.cuda.is_available():
        cfg.D = 16
        assert lib.cora_create(C.byref(cfg), 0, C.byref(ctx)) != 0
        assert b"HIP device" in lib.cora_last_error(None)
        with pytest.raises(_lib.CoraError):
            m = ca.TransformerDecoder(ca.create_config(hidden_size=16, num_heads=1, num_blocks=1))
            m.init_vodes(2)


def test_vode_layernorm_inventory():
    """use_vode_state_layernorm adds L+1 LayerNorms to the parameter tree (:212-218)."""
    m = ca.TransformerDecoder(ca.create_config(hidden_size=16, num_heads=1, num_blocks=2, use_vode_state_layernorm=True))
    assert len(m.vode_output_layernorms) == 3
    keys = m.param_keys()
    assert ".vode_output_layernorms[2].nn.weight" in keys and ".vode_output_layernorms[0].nn.bias" in keys
    m0 = ca.TransformerDecoder(ca.create_config(hidden_size=16, num_heads=1, num_blocks=2))
    assert len(m0.vode_output_layernorms) == 0


def test_make_mask_and_psnr():
    """Host halves of unmask_on_batch_enhanced: the two corruptions (:868-896) and calculate_psnr (:815-830)."""
    import cora_amd as ca
    cfg = ca.TransformerConfig(image_shape=(3, 8, 8), hidden_size=16, num_heads=1, num_blocks=1, patch_size=4,
                               update_weights_every_inference_step=False)
    model = ca.TransformerDecoder(cfg)
    x = np.random.default_rng(0).uniform(0, 1, (2, 3, 8, 8)).astype(np.float32)
    model.config.use_lower_half_mask = True
    mask, xc = ca.make_mask(model, x, 0.3)
    assert mask.shape == (2, 1, 8, 8) and mask[:, :, 6:].all() and not mask[:, :, :6].any()   # floor(8*0.3) = 2 rows
    assert (xc[:, :, 6:] == 0.5).all() and np.array_equal(xc[:, :, :6], x[:, :, :6])
    mask0, xc0 = ca.make_mask(model, x, 0.0)
    assert not mask0.any() and np.array_equal(xc0, x)
    model.config.use_lower_half_mask = False
    mask, xc = ca.make_mask(model, x, 0.5, np.random.default_rng(1))
    mb = np.broadcast_to(mask, x.shape)
    assert set(np.unique(mask)) <= {0.0, 1.0} and 0.2 < mask.mean() < 0.8
    assert np.array_equal(xc[mb == 0], x[mb == 0]) and np.abs(xc[mb == 1]).max() < 1.0
    assert ca.calculate_psnr(x, x) == pytest.approx(80.0)            # epsilon floor: 10 log10(1 / 1e-8)
    assert ca.calculate_psnr(np.zeros(4), np.full(4, 0.1)) == pytest.approx(20.0)


def _reference_key_set(L, vode_ln=False, vodes=True):
    """The archive keys `pcx.utils.save_params(model, path, filter=LayerParam | VodeParam)` writes for the
    reference's TransformerDecoder: `jtu.keystr` of the path to every selected Param
    (/root/reference/pcx/utils/_serialisation.py:47-60).  Written out by hand from the reference's attribute
    names -- TransformerDecoder.__init__ (/root/reference/src/decoder_transformer.py:140-218: `vodes`,
    `transformer_blocks`, `patch_projection`, `output_projection`, `vode_output_layernorms`),
    pxnn.TransformerBlock (/root/reference/pcx/nn/_layer.py:419-426: `layer_norm1/2`, `attention`,
    `linear1/2`), Layer.nn (:53-58) and the equinox field names (`weight`, `bias`, `query_proj`, ...) --
    so that it does not share code with cora_amd.params."""
    keys = set()
    for i in range(L):
        b = ".transformer_blocks[%d]" % i
        for ln in ("layer_norm1", "layer_norm2"):
            keys |= {b + "." + ln + ".nn.weight", b + "." + ln + ".nn.bias"}
        for pr in ("query_proj", "key_proj", "value_proj", "output_proj"):
            keys.add(b + ".attention.nn." + pr + ".weight")           # use_*_bias=False: no bias leaves
        for lin in ("linear1", "linear2"):
            keys |= {b + "." + lin + ".nn.weight", b + "." + lin + ".nn.bias"}
    for lin in ("patch_projection", "output_projection"):
        keys |= {"." + lin + ".nn.weight", "." + lin + ".nn.bias"}
    if vode_ln:
        for i in range(L + 1):
            keys |= {".vode_output_layernorms[%d].nn.weight" % i, ".vode_output_layernorms[%d].nn.bias" % i}
    if vodes:
        keys |= {".vodes[%d].h" % l for l in range(L + 3)}
    return keys


@pytest.mark.parametrize("vode_ln", [False, True])
def test_npz_layer_param_keys_and_round_trip(tmp_path, vode_ln):
    """save_params / load_params (pcx/utils/_serialisation.py:25-105) on the LayerParam tree, no GPU needed:
    the key set is the reference's, an archive holding exactly those keys loads into a fresh model, a
    missing key raises the reference's KeyError, class / union / predicate filters all work."""
    L = 2
    cfg = ca.create_config(hidden_size=16, num_heads=2, num_blocks=L, use_vode_state_layernorm=vode_ln)
    m = ca.TransformerDecoder(cfg, seed=3)
    path = str(tmp_path / "ckpt")
    ca.save_params(m, path, filter=ca.LayerParam)                # np.savez_compressed appends .npz, as in the reference
    want = _reference_key_set(L, vode_ln, vodes=False)
    with np.load(path + ".npz") as z:
        assert set(z.files) == want
        assert z[".transformer_blocks[1].linear1.nn.weight"].shape == (64, 16)
        assert z[".transformer_blocks[0].attention.nn.value_proj.weight"].dtype == np.float32
    # an archive with exactly the reference's keys (written here with plain NumPy, not by save_params) loads
    rng = np.random.default_rng(0)
    shapes = dict(param_shapes(cfg.patch_dim, 16, 64, L, vode_ln))
    blob = {k: rng.standard_normal(shapes[k]).astype(np.float32) for k in want}
    np.savez_compressed(str(tmp_path / "ref_style"), **blob)
    m2 = ca.TransformerDecoder(cfg, seed=99)
    ca.load_params(m2, str(tmp_path / "ref_style"))              # '.npz' added like the reference (:87)
    got = m2.get_weights()
    for k in want:
        np.testing.assert_array_equal(got[k], blob[k])
    np.testing.assert_array_equal(m2.transformer_blocks[1].attention.nn.key_proj.weight.get(),
                                  blob[".transformer_blocks[1].attention.nn.key_proj.weight"])
    # predicate filter (the reference accepts Callable[[Any], bool], :28)
    ca.save_params(m2, str(tmp_path / "pred"), filter=lambda p: isinstance(p, ca.LayerParam))
    with np.load(str(tmp_path / "pred.npz")) as z:
        assert set(z.files) == want
    # the predicate is asked per parameter (reference :41-53), so it can select INSIDE a family: block 1 only
    only_b1 = lambda p: isinstance(p, ca.LayerParam) and p.key.startswith(".transformer_blocks[1]")
    ca.save_params(m2, str(tmp_path / "b1"), filter=only_b1)
    with np.load(str(tmp_path / "b1.npz")) as z:
        assert set(z.files) == {k for k in want if k.startswith(".transformer_blocks[1]")}
    m3 = ca.TransformerDecoder(cfg, seed=5)
    before = m3.get_weights()
    ca.load_params(m3, str(tmp_path / "b1"), filter=only_b1)       # loads block 1, leaves everything else alone
    after = m3.get_weights()
    for k in want:
        np.testing.assert_array_equal(after[k], blob[k] if k.startswith(".transformer_blocks[1]") else before[k])
    del blob[".output_projection.nn.bias"]
    np.savez_compressed(str(tmp_path / "short"), **blob)
    with pytest.raises(KeyError, match="not found in the file"):
        ca.load_params(ca.TransformerDecoder(cfg), str(tmp_path / "short.npz"))
    with pytest.raises(TypeError):
        ca.save_params(m, path, filter=3)


def test_optim_init_resets_schedule_count():
    ow = ca.Optim(lambda: ca.chain(ca.clip_by_global_norm(1.0), ca.adamw(1e-3)))
    ow.count = 7
    ow.init()
    assert ow.count == 0 and ow._recipe().base.weight_decay == 1e-4


def test_round2_entry_points_reject_bad_arguments_without_a_gpu():
    """Error behaviour of the entry points added in round 2 (no compute, no GPU): null contexts and malformed
    buffers come back as negative status codes with a message, never as a crash."""
    import ctypes as C
    lib = _lib.load()
    out = _lib.CoraStepOut()
    assert lib.cora_step_dist(None, None, 0, 20, None, C.c_float(1e-3), 1, C.byref(out)) == -1
    assert lib.cora_comm_init(None, None, 128, 0, 1) == -1
    assert lib.cora_comm_info(None, None, None) == -1
    assert lib.cora_kernel_path(None) == -1
    small = C.create_string_buffer(16)
    assert lib.cora_comm_unique_id(small, 16) == -1           # the RCCL id is 128 bytes
    assert b"128" in lib.cora_last_error(None)
    # allocation-free forward: nulls and oversize problems are refused before anything touches a device
    cfg = _lib.CoraConfig(N=64, P=48, D=64, H=1, M=256, L=6, B=200, C=3, patch=4, Himg=32, Wimg=32, F=1, dtype_mode=0,
                          max_T=64)
    assert lib.cora_forward_static(C.byref(cfg), 0, None, 0, None, None, None, None) == -1
    assert lib.cora_forward_static_floats(C.byref(cfg)) > 4 * 1024 * 1024      # B=200 does not fit the 16 MiB array
    dummy = C.create_string_buffer(64)
    rc = lib.cora_forward_static(C.byref(cfg), 0, dummy, 1, dummy, dummy, dummy, None)
    assert rc != 0 and (b"scratch" in lib.cora_last_error(None) or b"weight" in lib.cora_last_error(None))
