"""Diagnostic: per-phase s_memtime breakdown of pc_block_wide (stamps build).  Usage:
   CORA_HIP_LIB=cora_amd/libcora_hip_stamps.so CORA_NO_GRAPH=1 python scripts/stamps_wide.py B [D H]"""
import ctypes as C, sys, dataclasses
import numpy as np
sys.path.insert(0, ".")
import cora_amd as ca
from cora_amd import _lib
B = int(sys.argv[1])
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4
mc = dataclasses.replace(ca.MODEL_CONFIGS["12block"], hidden_size=D, num_heads=H)
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
st = st.reshape(-1, 16)[: L * B, :14].astype(np.int64)
if (st[:, 10] == 0).all():          # the two-workgroup kernel has no stamp between the attention-backward passes
    st[:, 10] = st[:, 9]
d = np.diff(st, axis=1)
names = ["LN1", "QKV (6 tiles)", "attn fwd", "Wo (2) + LN2", "MLP fwd (16)", "error", "MLP bwd (16)", "LN2 bwd",
         "dO (2) + stores", "pass A", "pass B", "dxn (6)", "LN1 bwd + store"]
mf = [0, 6 * 128, 0, 2 * 128, 16 * 128, 0, 16 * 128, 0, 2 * 128, 0, 0, 6 * 128, 0]
tot = (st[:, 13] - st[:, 0])
print("B", B, "D", D, "H", H, "WGs", len(tot), "mean total cycles", tot.mean(), "min", tot.min(), "max", tot.max())
for k, nm in enumerate(names):
    print(f"  {nm:20s} mean {d[:, k].mean():9.0f}  ({100 * d[:, k].mean() / tot.mean():5.1f} %)   weight-tile MFMA cycles {mf[k] * 32 * (D // 128 if D == 128 else 1)}")
print("kernel span", st[:, 13].max() - st[:, 0].min())
