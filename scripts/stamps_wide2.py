"""Diagnostic (-DCORA_STAMPS=2): s_memtime inside stages 2 and 3 of pc_block_wide."""
import ctypes as C, sys, dataclasses
import numpy as np
sys.path.insert(0, ".")
import cora_amd as ca
from cora_amd import _lib
B = int(sys.argv[1])
mc = ca.MODEL_CONFIGS["12block"]
L = mc.num_blocks
model = ca.TransformerDecoder(ca.transformer_config_from(mc), seed=42)
rng = np.random.default_rng(0)
x = rng.uniform(0, 1, (B, 3, 32, 32)).astype(np.float32)
model.init_vodes(B, h0=(0.01 * rng.standard_normal((B, 64, 48))).astype(np.float32))
ow, oh = ca.make_optimizers(mc, model, 10)
lib = _lib.load()
lib.cora_dbg_stamps(model._ctx, None, 0)
ca.train_on_batch(3, x, model=model, optim_w=ow, optim_h=oh)
n = (L + 1) * B * 16
st = np.zeros(n, dtype=np.uint64)
lib.cora_dbg_stamps(model._ctx, st.ctypes.data_as(C.c_void_p), n)
st = st.reshape(-1, 16)[: L * B, :10].astype(np.int64)
d = np.diff(st, axis=1)
names = ["s2 barrier wait", "s2 group 0 (+first reads)", "s2 groups 1-3", "s2 groups 4-7", "s2 -> s3 (unpack, zero)", "s3 barrier wait", "s3 group 0", "s3 groups 1-3", "s3 groups 4-7"]
for k, nm in enumerate(names):
    print(f"  {nm:28s} mean {d[:, k].mean():8.0f} min {d[:, k].min():8.0f} max {d[:, k].max():8.0f}")
