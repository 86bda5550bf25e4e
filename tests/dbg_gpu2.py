import sys, zlib
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import pcvit_ref as R
from tests.util import build_model, make_case, relerr
from tests.test_gpu_parity import CASES
import cora_amd as ca
name = sys.argv[1]
shape, p, D, H, L, B, _ = CASES[name]
for T in (1, 2, 3):
    case = make_case(shape, p, D, H, L, B, seed=zlib.crc32(name.encode()) % 1000)
    model, ow, oh = build_model(case)
    eta = [0.095] * T
    ref = R.train_on_batch(T, case["x"], case["W"], case["h"], case["pos"], case["dims"], case["hp"], eta, 1e-3)
    r32 = R.train_on_batch(T, case["x"], case["W"], case["h"], case["pos"], case["dims"], case["hp"], eta, 1e-3, dtype=torch.float32)
    Eh, Ew, gh, gw, mse = ca.train_on_batch(T, case["x"], model=model, optim_w=ow, optim_h=oh)
    print("T", T)
    print(" gpu gh", ["%.1e" % relerr(gh[l], ref["final_h_grad"][l]) for l in range(L + 2)])
    print(" f32 gh", ["%.1e" % relerr(r32["final_h_grad"][l], ref["final_h_grad"][l]) for l in range(L + 2)])
    print(" gpu h ", ["%.1e" % relerr(model.vodes[l].h.get(), ref["h"][l]) for l in range(L + 2)])
    print(" f32 h ", ["%.1e" % relerr(r32["h"][l], ref["h"][l]) for l in range(L + 2)])
    gn = max(np.abs(v).max() for v in ref["w_grad"].values())
    print(" gpu gw max", max(np.abs(gw[k] - v).max() / gn for k, v in ref["w_grad"].items()),
          " f32 gw max", max(np.abs(r32["w_grad"][k] - v).max() / gn for k, v in ref["w_grad"].items()))
    print(" |g| max per layer", ["%.1e" % np.abs(g).max() for g in ref["final_h_grad"]])
    model.close()
