"""Diagnostic: per-phase s_memtime breakdown of pc_block_fused (stamps build).  Usage:
   CORA_HIP_LIB=cora_amd/libcora_hip_stamps.so CORA_NO_GRAPH=1 python scripts/stamps.py B"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, ".")
import cora_amd as ca
from cora_amd import _lib
B = int(sys.argv[1])
mc = ca.MODEL_CONFIGS["6block"]
model = ca.TransformerDecoder(ca.transformer_config_from(mc), seed=42)
rng = np.random.default_rng(0)
x = rng.uniform(0, 1, (B, 3, 32, 32)).astype(np.float32)
model.init_vodes(B, h0=(0.01 * rng.standard_normal((B, 64, 48))).astype(np.float32))
ow, oh = ca.make_optimizers(mc, model, 10)
lib = _lib.load()
lib.cora_dbg_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.cora_dbg_stamps(model._ctx, None, 0)
ca.train_on_batch(3, x, model=model, optim_w=ow, optim_h=oh)
L = 6
n = (L + 1) * B * 16
st = np.zeros(n, dtype=np.uint64)
# the wgrad pass (MODE_STORE) does not stamp, so these are from the last inference step
lib.cora_dbg_stamps(model._ctx, st.ctypes.data_as(C.c_void_p), n)
st = st.reshape(-1, 16)[: L * B, :14].astype(np.int64)
d = np.diff(st, axis=1)
names = ["LN1+x load", "QKV(3 tiles)", "attn fwd", "Wo", "LN2", "MLP fwd(8 tiles)", "err+energy", "MLP bwd(8)",
         "LN2 bwd", "dO tile", "passA", "passB", "dxn(3 tiles)", ]
tot = (st[:, 13] - st[:, 0])
print("B", B, "WGs", len(tot), "mean total cycles(100MHz ticks?)", tot.mean(), "min", tot.min(), "max", tot.max())
for k, nm in enumerate(names):
    print(f"  {nm:20s} mean {d[:, k].mean():9.0f}  ({100 * d[:, k].mean() / tot.mean():5.1f} %)")
span = st[:, 13].max() - st[:, 0].min()
print("kernel span", span)
