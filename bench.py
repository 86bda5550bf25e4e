#!/usr/bin/env python3
"""bench.py -- PC inference images x T-steps / second for the PC-ViT step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one whole train_on_batch-equivalent call on one synthetic batch: T inference steps
(hipGraph) + dE/dW + (RCCL all-reduce when N > 1) + clip/AdamW + "forward" reconstruction and MSE.
Inputs are resident in HBM when the timed region starts (x is a device tensor).  Weak scaling:
every rank holds B images (BASELINE.json configs[1]: 6-block PC-ViT, 64 tokens, fp32, T=20, B=200).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0


def flops_per_image_step(N, P, D, L):
    """S = 8 N P D + L (48 N D^2 + 12 N^2 D)."""
    return 8.0 * N * P * D + L * (48.0 * N * D * D + 12.0 * N * N * D)


def cpu_baseline(mc, tc, B, T, seconds_budget=20.0):
    """The oracle (fp32 torch-CPU restatement, oracle/pcvit_ref.py) timed on this host's cores on a
    bounded sample of the SAME workload: whole steps (T inference steps + weight update + recon) at
    the same B, repeated until ~seconds_budget elapsed.  Reported, never the thing measured."""
    import torch
    from oracle import pcvit_ref as R
    from cora_amd.params import init_params
    from cora_amd.posenc import create_positional_encoding
    dims = R.Dims(tuple(tc.image_shape), tc.patch_size, tc.hidden_size, tc.num_heads, tc.num_blocks)
    W = init_params(dims.P, dims.D, dims.M, dims.L, 42)
    pos = create_positional_encoding(tuple(tc.image_shape), dims.D, tc.patch_size, theta=tc.theta)
    rng = np.random.default_rng(1234)
    x = rng.uniform(0, 1, (B,) + tuple(tc.image_shape)).astype(np.float32)
    h0 = (0.01 * rng.standard_normal((B, dims.N, dims.P))).astype(np.float32)
    h = R.init_vodes(h0, W, pos, dims, dtype=torch.float32)
    hp = R.Hyper(l1=mc.intermediate_l1_coeff, l2=mc.intermediate_l2_coeff,
                 use_scaling=mc.use_inference_lr_scaling, scale_base=mc.inference_lr_scale_base,
                 clip_h=mc.h_grad_clip_norm, clip_w=mc.w_grad_clip_norm, momentum=mc.hidden_momentum,
                 wd=mc.weight_decay)
    eta = [mc.peak_lr_hidden] * T
    R.train_on_batch(1, x, W, h, pos, dims, hp, eta[:1], mc.peak_lr_weights, dtype=torch.float32)  # warm
    n, t0 = 0, time.perf_counter()
    while True:
        R.train_on_batch(T, x, W, h, pos, dims, hp, eta, mc.peak_lr_weights, dtype=torch.float32)
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 8:
            break
    return {"value": B * T * n / el, "unit": "images*T-steps/s", "cores": int(torch.get_num_threads()),
            "kind": "port", "sample": f"{n} whole steps (T={T} + weight update + recon) at B={B}, "
            f"fp32 torch-CPU oracle, {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="6block")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: the config's batch_size)")
    ap.add_argument("--T", type=int, default=None)
    ap.add_argument("--hidden-size", type=int, default=None, help="override the config (12-block sweep points)")
    ap.add_argument("--num-heads", type=int, default=None)
    ap.add_argument("--num-blocks", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                    help="fp32 = parity path (the headline number); bf16 = opt-in bf16-MFMA mode (per-op path)")
    args = ap.parse_args()

    import torch
    import cora_amd as ca
    from cora_amd import _lib

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    torch.cuda.set_device(local_rank)
    dist = None
    force_dp = os.environ.get("CORA_FORCE_DP") == "1"     # exercise the N>1 code path on one GPU
    if world > 1 or force_dp:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if force_dp and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    mc = ca.MODEL_CONFIGS[args.config]
    over = {k: v for k, v in (("hidden_size", args.hidden_size), ("num_heads", args.num_heads),
                              ("num_blocks", args.num_blocks)) if v is not None}
    if over:
        import dataclasses
        mc = dataclasses.replace(mc, **over)
    tc = ca.transformer_config_from(mc)
    B = args.batch or mc.batch_size
    T = args.T or mc.inference_steps
    lib = _lib.load()

    model = ca.TransformerDecoder(tc, seed=42, device=local_rank, max_T=max(T, 64), compute_dtype=args.dtype)
    rng = np.random.default_rng(1234 + rank)
    x_host = rng.uniform(0, 1, (B,) + tuple(tc.image_shape)).astype(np.float32)
    h0 = (0.01 * rng.standard_normal((B, model.N, model.P))).astype(np.float32)
    model.init_vodes(B, h0=h0)
    optim_w, optim_h = ca.make_optimizers(mc, model, total_train_steps=args.steps + args.warmup)
    model._apply_recipes(optim_w, optim_h)
    ctx = model._ctx
    dp = None
    if world > 1 or force_dp:
        from cora_amd.dist import DataParallel
        dp = DataParallel(model, B)
    x_dev = torch.from_numpy(x_host).cuda()
    torch.cuda.synchronize()
    eta_h = np.full(T, mc.peak_lr_hidden, dtype=np.float32)
    eta_w = C.c_float(mc.peak_lr_weights)
    out = _lib.CoraStepOut()
    xptr = C.c_void_p(x_dev.data_ptr())
    eptr = eta_h.ctypes.data_as(C.c_void_p)

    def one_step():
        if dp is None:
            _lib.check(lib.cora_step(ctx, xptr, 1, T, eptr, eta_w, C.byref(out)), ctx)
        else:
            _lib.check(lib.cora_infer(ctx, xptr, 1, T, eptr, None), ctx)      # in flight; cora_wgrad ends it
            _lib.check(lib.cora_wgrad(ctx, C.byref(out)), ctx)
            w = dp.allreduce()
            _lib.check(lib.cora_apply_wgrad(ctx, eta_w, C.c_float(1.0 / (B * w)), C.byref(out)), ctx)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    infer_ms = 0.0
    trace = []
    for _ in range(args.steps):
        one_step()
        infer_ms += out.infer_ms
        trace.append((out.final_h_energy, out.final_w_energy, out.train_mse))
        if os.environ.get("CORA_TRACE") == "2":
            tot = np.zeros(T)
            lib.cora_get_energies(ctx, T, tot.ctypes.data_as(C.c_void_p), None)
            print("per-step E:", ["%.3g" % v for v in tot], file=sys.stderr)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if os.environ.get("CORA_TRACE") == "1":
        print("trace:", trace, file=sys.stderr)
    timing_build = "CORA_HIP_LIB" in os.environ      # diagnostic libraries may compute garbage on purpose
    if not timing_build and (not np.isfinite(out.final_h_energy) or not np.isfinite(out.train_mse)):
        raise SystemExit(f"non-finite energy / mse in the timed region: {trace}")

    result = None
    if rank == 0:
        value = world * B * T * args.steps / elapsed
        S = flops_per_image_step(model.N, model.P, model.D, model.L)
        # per-kernel device times of ONE inference step, HIP events on the ctx stream
        cap = 16
        names = C.create_string_buffer(32 * cap)
        launches = (C.c_int * cap)()
        ms = (C.c_double * cap)()
        fl = (C.c_double * cap)()
        by = (C.c_double * cap)()
        nout = C.c_int()
        _lib.check(lib.cora_profile_step(ctx, cap, names, launches, ms, fl, by, C.byref(nout)), ctx)
        kinds = {}
        for k in range(nout.value):
            nm = names.raw[32 * k:32 * (k + 1)].split(b"\0")[0].decode()
            if launches[k]:
                kinds[nm] = {"launches": launches[k], "ms": ms[k], "flops": fl[k], "bytes": by[k]}
        dom = max(kinds, key=lambda n: kinds[n]["ms"])
        d = kinds[dom]
        mfma_bound = d["flops"] > 0
        if mfma_bound:
            achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": None,
                    "avg_launch_us": 1e3 * d["ms"] / d["launches"],
                    "flops_per_launch": d["flops"] / d["launches"]}
        else:
            achieved = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": achieved / PEAK_HBM_GBS, "traffic": None,
                    "avg_launch_us": 1e3 * d["ms"] / d["launches"]}
        # HBM bytes per launch of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        # (scripts/pmc_traffic.sh), committed under profiles/.  FETCH_SIZE reads 1/2 of wide coalesced
        # reads on gfx950 (MI355X_MICROARCH.md, HBM) -> doubled; both counters are in KB.
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
            key = next((k for k in tj if dom == "pc_block_fused" and k.startswith("void pc_block_fused_kernel<0")), None)
            if key and args.config == "6block" and B == 200 and not over:
                roof["traffic"] = 1024.0 * (2.0 * tj[key]["FETCH_SIZE_per_launch_KB"] + tj[key]["WRITE_SIZE_per_launch_KB"])
                roof["traffic_source"] = "profiles/r1_pmc_traffic.json (rocprofv3 PMC, offline)"
        except Exception:
            pass
        reps = 5  # cora_profile_step repeats the step 5 times
        step_ms_kernels = sum(v["ms"] for v in kinds.values()) / reps
        result = {
            "metric": "PC inference images x T-steps/sec (node)", "value": value, "unit": "images*T-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.dtype == "fp32" else "bf16 operands / f32 accumulate", "data": "synthetic",
            "config": {"workload": f"{args.config} PC-ViT train_on_batch: N={model.N} tokens, P={model.P}, "
                                   f"D={model.D}, H={model.H}, L={model.L}, B={B}/GPU, T={T} inference steps + "
                                   "weight update + recon", "global_batch": world * B, "T": T,
                       "parallelism": f"dp{world}"},
            "infer_only_images_steps_per_s": world * B * T * args.steps / (infer_ms * 1e-3),
            "algorithmic_flops_per_image_step": S,
            "whole_step_mfma_frac": (value * S / 1e12) / (PEAK_FP32_MATRIX_TFLOPS * world),
            "roofline": roof,
            "kernel_ms_per_inference_step": {k: v["ms"] / reps for k, v in kinds.items()},
            # bandwidth-bound kernels of the step (energy + gradient finish, clip + momentum update):
            # algorithmic bytes / measured launch time, against the HBM peak
            "hbm_kernels": {k: {"achieved_GBs": v["bytes"] / (v["ms"] * 1e-3) / 1e9,
                                "frac_of_hbm_peak": v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                "bytes_per_launch": v["bytes"] / v["launches"],
                                "avg_launch_us": 1e3 * v["ms"] / v["launches"]}
                            for k, v in kinds.items() if v["flops"] == 0 and v["bytes"] > 0},
            "inference_step_kernel_ms_total": step_ms_kernels,
        }
    if dist is not None:
        dist.barrier()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(mc, tc, B, T)
        else:
            result["cpu_baseline"] = None
        if timing_build:    # not a product number: the library was swapped for a diagnostic build
            result["diagnostic_library"] = os.environ["CORA_HIP_LIB"]
        print(json.dumps(result))
    model.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
