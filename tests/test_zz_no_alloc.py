"""The allocation-free entry point `cora_forward_static` (include/cora_hip.h) against the existing
`cora_forward`, through the C-ABI: two calls in a row on one GPU, the second with a smaller input, so
that stale contents of the shared static scratch would show.  Bar: the test shapes take the per-op
kernels in both entry points (same launches on the same inputs) -> bit-exact; on the cfg-2 shape
`cora_forward` runs the fused chain kernel instead -> rel <= 1e-5 (fp32, ③'s bar is 1e-4)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.util import build_model, make_case

pytestmark = pytest.mark.gpu


def _static_forward(model, cfg_B, W_flat, pos, h0, stream=None):
    from cora_amd import _lib
    lib = _lib.load()
    cfg = model._make_cfg(cfg_B)
    dev = torch.device("cuda", 0)
    Wd = torch.from_numpy(W_flat).to(dev)
    pd = torch.from_numpy(np.ascontiguousarray(pos, dtype=np.float32)).to(dev)
    hd = torch.from_numpy(np.ascontiguousarray(h0, dtype=np.float32)).to(dev)
    out = torch.full((cfg_B,) + tuple(model.config.image_shape), float("nan"), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    rc = lib.cora_forward_static(C.byref(cfg), 0, C.c_void_p(Wd.data_ptr()), Wd.numel(), C.c_void_p(pd.data_ptr()),
                                 C.c_void_p(hd.data_ptr()), C.c_void_p(out.data_ptr()), None)
    assert rc == 0, lib.cora_last_error(None)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("shape_name,exact", [("per_op_h2", True), ("per_op_vln", True), ("cfg2_shape", False)])
def test_forward_static_matches_cora_forward_two_calls(shape_name, exact):
    import cora_amd as ca
    shapes = {"per_op_h2": ((3, 16, 16), 4, 32, 2, 2, False), "per_op_vln": ((3, 24, 24), 4, 32, 2, 1, True),
              "cfg2_shape": ((3, 32, 32), 4, 64, 1, 3, False)}
    shape, p, D, H, L, vln = shapes[shape_name]
    results = []
    for B in (8, 3 if shape_name != "per_op_vln" else 1):         # second call: a smaller input
        case = make_case(shape, p, D, H, L, B, seed=70 + B, vode_ln=vln)
        model, _, _ = build_model(case)
        with ca.step(model, ca.STATUS_FORWARD):
            ref = ca.forward(None, model=model)                     # the existing entry point: cora_forward
        got = _static_forward(model, B, model._all_weights(), case["pos"], case["h"][0])
        model.close()
        assert np.isfinite(got).all()
        if exact:
            np.testing.assert_array_equal(got, ref)
        else:
            assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
        results.append(got)
    assert results[0].shape[0] > results[1].shape[0]


def test_forward_static_rejects_oversize_and_bad_args():
    import cora_amd as ca
    from cora_amd import _lib
    lib = _lib.load()
    case = make_case((3, 32, 32), 4, 64, 1, 2, 4, seed=1)
    model, _, _ = build_model(case)
    cfg = model._make_cfg(200)                                     # B = 200 at the cfg-2 shape: 11.4 Mi floats > 4 Mi
    need = lib.cora_forward_static_floats(C.byref(cfg))
    assert need > 4 * 1024 * 1024
    dummy = torch.zeros(16, device="cuda")
    pt = C.c_void_p(dummy.data_ptr())
    rc = lib.cora_forward_static(C.byref(cfg), 0, pt, model._nW, pt, pt, pt, None)
    assert rc != 0 and b"scratch" in lib.cora_last_error(None)
    # the documented limit at the cfg-2 shape: 56 640 floats per image -> B = 74 fits the 4 Mi floats, B = 75 does not
    c74, c75 = model._make_cfg(74), model._make_cfg(75)
    assert lib.cora_forward_static_floats(C.byref(c74)) == 74 * 56640 <= 4 * 1024 * 1024
    assert lib.cora_forward_static_floats(C.byref(c75)) > 4 * 1024 * 1024
    rc = lib.cora_forward_static(C.byref(c75), 0, pt, model._nW, pt, pt, pt, None)
    assert rc != 0 and b"scratch" in lib.cora_last_error(None)
    cfg = model._make_cfg(4)
    assert lib.cora_forward_static_floats(C.byref(cfg)) < 4 * 1024 * 1024
    rc = lib.cora_forward_static(C.byref(cfg), 0, pt, model._nW + 1, pt, pt, pt, None)
    assert rc != 0 and b"weight arena size" in lib.cora_last_error(None)
    rc = lib.cora_forward_static(C.byref(cfg), 0, None, model._nW, pt, pt, pt, None)
    assert rc != 0
    model.close()
