"""ctypes binding of libcora_hip.so (include/cora_hip.h).  There is NO fallback: if the HIP library
is missing or fails to load, importing anything that computes raises."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CORA_HIP_LIB") or os.path.join(_HERE, "libcora_hip.so")   # env: diagnostic builds


class CoraConfig(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("P", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("M", C.c_int32),
        ("L", C.c_int32), ("B", C.c_int32), ("C", C.c_int32), ("patch", C.c_int32),
        ("Himg", C.c_int32), ("Wimg", C.c_int32), ("F", C.c_int32), ("dtype_mode", C.c_int32),
        ("use_scaling", C.c_int32), ("scale_base", C.c_float), ("l1", C.c_float), ("l2", C.c_float),
        ("clip_h", C.c_float), ("clip_w", C.c_float), ("momentum", C.c_float),
        ("adam_b1", C.c_float), ("adam_b2", C.c_float), ("adam_eps", C.c_float),
        ("adam_wd", C.c_float), ("use_grad_norm", C.c_int32), ("grad_norm_target", C.c_float),
        ("max_T", C.c_int32), ("use_vode_ln", C.c_int32),
    ]


class CoraStepOut(C.Structure):
    _fields_ = [("final_h_energy", C.c_double), ("final_w_energy", C.c_double),
                ("train_mse", C.c_double), ("infer_ms", C.c_float), ("learn_ms", C.c_float)]


# name -> (restype, argtypes); every symbol include/cora_hip.h declares
_P = C.c_void_p
_FP = C.POINTER(C.c_float)
_DP = C.POINTER(C.c_double)
_SZ = C.c_size_t
SIGNATURES = {
    "cora_create": (C.c_int, [C.POINTER(CoraConfig), C.c_int, C.POINTER(_P)]),
    "cora_destroy": (C.c_int, [_P]),
    "cora_last_error": (C.c_char_p, [_P]),
    "cora_weights_size": (C.c_int, [_P, C.POINTER(_SZ)]),
    "cora_weight_offset": (C.c_int, [_P, C.c_char_p, C.POINTER(_SZ), C.POINTER(_SZ)]),
    "cora_set_weights": (C.c_int, [_P, _P, _SZ]),
    "cora_get_weights": (C.c_int, [_P, _P, _SZ]),
    "cora_set_pos": (C.c_int, [_P, _P, _SZ]),
    "cora_vode_size": (C.c_int, [_P, C.c_int, C.POINTER(_SZ)]),
    "cora_set_vode": (C.c_int, [_P, C.c_int, _P, _SZ]),
    "cora_get_vode": (C.c_int, [_P, C.c_int, _P, _SZ]),
    "cora_get_vode_grad": (C.c_int, [_P, C.c_int, _P, _SZ]),
    "cora_init_vodes": (C.c_int, [_P, _P]),
    "cora_set_adam_state": (C.c_int, [_P, _P, _P, C.c_int64]),
    "cora_get_adam_state": (C.c_int, [_P, _P, _P, C.POINTER(C.c_int64)]),
    "cora_set_optim": (C.c_int, [_P] + [C.c_float] * 7),
    "cora_step": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_float, C.POINTER(CoraStepOut)]),
    "cora_infer": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.POINTER(CoraStepOut)]),
    "cora_set_grad_options": (C.c_int, [_P, C.c_int, C.c_float, C.c_int, C.c_float]),
    "cora_step_ipc": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, C.POINTER(CoraStepOut)]),
    "cora_infer_begin": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "cora_infer_grad": (C.c_int, [_P, C.c_int]),
    "cora_infer_sumsq": (C.c_int, [_P, C.c_int, _P, _SZ, C.c_int]),
    "cora_infer_update": (C.c_int, [_P, C.c_int]),
    "cora_infer_end": (C.c_int, [_P, C.POINTER(CoraStepOut)]),
    "cora_wgrad": (C.c_int, [_P, C.POINTER(CoraStepOut)]),
    "cora_apply_wgrad": (C.c_int, [_P, C.c_float, C.c_float, C.POINTER(CoraStepOut)]),
    "cora_comm_unique_id": (C.c_int, [_P, _SZ]),
    "cora_comm_init": (C.c_int, [_P, _P, _SZ, C.c_int, C.c_int]),
    "cora_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "cora_step_dist": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_float, C.c_int, C.POINTER(CoraStepOut)]),
    "cora_step_ipc_dist": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, C.c_int, C.POINTER(CoraStepOut)]),
    "cora_dist_prepare": (C.c_int, [_P, C.c_int, C.c_int]),
    "cora_dist_set_eager": (C.c_int, [_P, C.c_int]),
    "cora_dist_mode": (C.c_int, [_P]),
    "cora_wgrad_buffer": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_SZ)]),
    "cora_bind_wgrad_buffer": (C.c_int, [_P, _P, _SZ]),
    "cora_get_wgrad": (C.c_int, [_P, _P, _SZ]),
    "cora_get_energies": (C.c_int, [_P, C.c_int, _P, _P]),
    "cora_forward": (C.c_int, [_P, _P, _SZ]),
    "cora_forward_static_floats": (_SZ, [C.POINTER(CoraConfig)]),
    "cora_forward_static": (C.c_int, [C.POINTER(CoraConfig), C.c_int, _P, _SZ, _P, _P, _P, _P]),
    "cora_energy": (C.c_int, [_P, _P, _DP, _P, _P, _SZ]),
    "cora_unmask": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_float, _P, _P, _P, _P]),
    "cora_unmask_learn": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_float, _P, C.c_float, _P, _P, _P, _P]),
    "cora_dbg_stamps": (C.c_int, [_P, _P, _SZ]),
    "cora_profile_step": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, C.POINTER(C.c_int)]),
    "cora_sync": (C.c_int, [_P]),
    "cora_kernel_path": (C.c_int, [_P]),
}

_lib: Optional[C.CDLL] = None


class CoraError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library; raise loudly if it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CoraError(
            f"{LIB_PATH} not found: build it with `make -C cora_amd/csrc` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`).  cora_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, ctx=None):
    if rc != 0:
        msg = load().cora_last_error(ctx)
        raise CoraError(f"libcora_hip error {rc}: {msg.decode() if msg else '?'}")
