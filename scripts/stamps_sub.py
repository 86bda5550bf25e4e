"""Diagnostic: zoom stamps (-DCORA_STAMPS=2) over the three QKV tiles."""
import ctypes as C, sys
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
lib.cora_dbg_stamps(model._ctx, st.ctypes.data_as(C.c_void_p), n)
st = st.reshape(-1, 16)[: L * B, :14].astype(np.int64)
d = np.diff(st, axis=1)
names = ["(fwd..)", "publish W2T", "prefetch+am load issue", "gemm+gelugrad", "publish W1T", "prefetch+gemm (async)"] * 2
d = d[:, 1:]
for k in range(12):
    print(f"  {names[k]:24s} mean {d[:, k].mean():9.0f}  min {d[:, k].min():7d} max {d[:, k].max():7d}")
