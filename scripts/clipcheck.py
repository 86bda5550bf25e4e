"""Diagnostic: global vode-gradient norm at the last inference step vs the clip threshold."""
import sys
import numpy as np
sys.path.insert(0, ".")
import cora_amd as ca
B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mc = ca.MODEL_CONFIGS["6block"]
model = ca.TransformerDecoder(ca.transformer_config_from(mc), seed=42)
rng = np.random.default_rng(1234)
x = rng.uniform(0, 1, (B, 3, 32, 32)).astype(np.float32)
model.init_vodes(B, h0=(0.01 * rng.standard_normal((B, 64, 48))).astype(np.float32))
ow, oh = ca.make_optimizers(mc, model, 10)
for T in (1, 5, 20):
    model.init_vodes(B, h0=(0.01 * rng.standard_normal((B, 64, 48))).astype(np.float32))
    he, we, hg, wg, mse = ca.train_on_batch(T, x, model=model, optim_w=ow, optim_h=oh)
    n2 = [float((hg[l].astype(np.float64) ** 2).sum()) for l in range(model.L + 2)]
    print("T", T, "global grad norm", sum(n2) ** 0.5, "per layer", [round(v ** 0.5, 2) for v in n2], "clip", mc.h_grad_clip_norm, "E", he)
