#!/usr/bin/env python3
"""bench.py -- PC inference images x T-steps / second for the PC-ViT step on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one whole train_on_batch-equivalent call on one synthetic batch: T inference steps
(hipGraph) + dE/dW + (RCCL all-reduce when N > 1) + clip/AdamW + "forward" reconstruction and MSE.
Inputs are resident in HBM when the timed region starts (x is a device tensor).

Workload of the headline line: BASELINE.json configs[1], the 6-block PC-ViT (64 tokens, fp32, T=20) at
B=200 images per GPU (weak scaling, the default) or, with --global-batch G, G images sharded over the
ranks (strong scaling: what north_star / configs[2] describe).  The 12-block sweep point named in the
same metric (D=128, H=4, L=12) is measured in the same run and reported under "extra"."cfg3".
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense
PEAK_BF16_MATRIX_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16 MFMA dense (the 5 PF headline figure is 2:1 sparse)
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="6block")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: the config's batch_size)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="strong scaling: this many images in total, sharded over the ranks (north_star: the batch is sharded)")
    ap.add_argument("--T", type=int, default=None)
    ap.add_argument("--hidden-size", type=int, default=None, help="override the config (12-block sweep points)")
    ap.add_argument("--num-heads", type=int, default=None)
    ap.add_argument("--num-blocks", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the 12-block (cfg 3) measurement")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "bf16x3"],
                    help="fp32 = parity path (the headline number); bf16 = opt-in bf16-MFMA mode; bf16x3 = opt-in "
                         "three-term bf16 split in the per-op GEMMs (fp32-grade products; priced against the fp32 peak)")
    ap.add_argument("--clip-mode", default="exact", choices=["exact", "local"],
                    help="N > 1: 'exact' all-reduces the vode-gradient sums of squares every inference step "
                         "(the reference's whole-batch clip_by_global_norm), 'local' clips per shard")
    ap.add_argument("--dp-backend", default="rccl", choices=["rccl", "torch"],
                    help="N > 1: 'rccl' = collectives inside the captured graphs (cora_step_dist), "
                         "'torch' = step split at the exchange point, torch.distributed all-reduce")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="rank processes only rendezvous (gloo) and report: exercises the launcher without a GPU")
    return ap.parse_args(argv)


# ---- launcher: `python bench.py --gpus N` starts N fresh rank processes BEFORE anything touches the GPU ----
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int) -> int:
    """One child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, as
    torch.distributed.run would set them).  The parent never initialises HIP; children are started as
    new processes (no fork of a GPU process, no exec after GPU init).  Rank 0's stdout is ours."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CORA_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + 3000
    try:
        while procs and any(p.poll() is None for p in procs):
            time.sleep(0.2)
            bad = [p for p in procs if p.poll() not in (None, 0)]
            if bad or time.time() > deadline:        # one rank died (or we ran out of time): stop the others by PID
                rc = bad[0].returncode if bad else 124
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                break
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
            if rc == 0 and p.returncode:
                rc = p.returncode
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def dry_run_rank(args, rank, world):
    """Launcher rehearsal (CPU, gloo): rendezvous, barrier, MAX over ranks, one JSON line from rank 0."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    B = (args.global_batch // world) if args.global_batch else (args.batch or 200)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "world_observed": dist.get_world_size(),
                          "max_over_ranks": float(t.item()), "scaling": "strong" if args.global_batch else "weak",
                          "per_gpu_batch": B, "global_batch": B * world}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


# ---- one measured workload ------------------------------------------------------------------------------
def measure(args, cfg_name, over, B, T, steps, warmup, rank, local_rank, world, dist, profile=True, dtype=None):
    dtype = dtype or args.dtype
    import dataclasses
    import torch
    import cora_amd as ca
    from cora_amd import _lib
    mc = ca.MODEL_CONFIGS[cfg_name]
    if over:
        mc = dataclasses.replace(mc, **over)
    tc = ca.transformer_config_from(mc)
    T = T or mc.inference_steps
    lib = _lib.load()
    rng = np.random.default_rng(1234 + rank)
    x_host = rng.uniform(0, 1, (B,) + tuple(tc.image_shape)).astype(np.float32)
    force_dp = os.environ.get("CORA_FORCE_DP") == "1"     # exercise the N>1 code path on one GPU

    def build(backend):
        """Model + data-parallel binding.  backend: None (one GPU), "rccl" (collectives inside the library's
        graphs) or "torch" (step split at the exchange point, per-shard clip)."""
        model = ca.TransformerDecoder(tc, seed=42, device=local_rank, max_T=max(T, 64), compute_dtype=dtype)
        h0 = (0.01 * np.random.default_rng(4321 + rank).standard_normal((B, model.N, model.P))).astype(np.float32)
        model.init_vodes(B, h0=h0)
        optim_w, optim_h = ca.make_optimizers(mc, model, total_train_steps=steps + warmup)
        model._apply_recipes(optim_w, optim_h)
        dp, mode = None, "single"
        if backend is not None:
            from cora_amd.dist import DataParallel
            exact = args.clip_mode == "exact"
            if backend == "rccl":
                try:
                    dp = DataParallel(model, B, exact_clip=exact, backend="rccl")
                except Exception as e:       # RCCL could not be bound inside the library: say so, use the split step
                    print(f"[bench] in-library RCCL unavailable ({e}); falling back to the torch.distributed path", file=sys.stderr)
                    backend = "torch"
            if backend == "torch":
                exact = False                # the split step's exact mode is a host round trip per inference step
                dp = DataParallel(model, B, exact_clip=False, backend="torch")
            mode = f"{backend}:{'exact' if exact else 'local'}-clip"
        return model, dp, mode

    model, dp, mode = build(args.dp_backend if (world > 1 or force_dp) else None)
    ctx = model._ctx
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
        elif dp.in_library:
            _lib.check(lib.cora_step_dist(ctx, xptr, 1, T, eptr, eta_w, int(dp.exact_clip), C.byref(out)), ctx)
        else:
            _lib.check(lib.cora_infer(ctx, xptr, 1, T, eptr, None), ctx)      # in flight; cora_wgrad ends it
            _lib.check(lib.cora_wgrad(ctx, C.byref(out)), ctx)
            w = dp.allreduce()
            _lib.check(lib.cora_apply_wgrad(ctx, eta_w, C.c_float(1.0 / (B * w)), C.byref(out)), ctx)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up; if the in-library RCCL step fails on any rank (e.g. a collective the runtime will not run under a
    # graph), every rank rebuilds on the torch.distributed path instead of losing the run
    failed = 0
    try:
        for _ in range(warmup):
            one_step()
    except _lib.CoraError as e:
        if dp is None or not dp.in_library:
            raise
        failed = 1
        print(f"[bench] rank {rank}: in-library RCCL step failed ({e})", file=sys.stderr)
    if dist is not None and dp is not None and dp.in_library:
        flag = torch.tensor([failed], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        failed = int(flag.item())
    if failed:
        print("[bench] falling back to the torch.distributed data-parallel path (per-shard clip)", file=sys.stderr)
        model.close()
        model, dp, mode = build("torch")
        ctx = model._ctx
        for _ in range(warmup):
            one_step()
    barrier()
    t0 = time.perf_counter()
    infer_ms = 0.0
    trace = []
    for _ in range(steps):
        one_step()
        infer_ms += out.infer_ms
        trace.append((out.final_h_energy, out.final_w_energy, out.train_mse))
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

    res = None
    if rank == 0:
        bf16 = dtype == "bf16"
        peak = PEAK_BF16_MATRIX_TFLOPS if bf16 else PEAK_FP32_MATRIX_TFLOPS
        value = world * B * T * steps / elapsed
        S = flops_per_image_step(model.N, model.P, model.D, model.L)
        res = {"value": value, "ms_per_step": 1e3 * elapsed / steps, "per_gpu_batch": B, "T": T,
               "workload": f"{cfg_name} PC-ViT train_on_batch: N={model.N} tokens, P={model.P}, D={model.D}, H={model.H}, "
                           f"L={model.L}, B={B}/GPU, T={T} inference steps + weight update + recon",
               "infer_only_images_steps_per_s": world * B * T * steps / (infer_ms * 1e-3),
               "algorithmic_flops_per_image_step": S,
               "whole_step_mfma_frac": (value * S / 1e12) / (peak * world), "mfma_peak_tflops": peak,
               "dp_mode": mode, "mc": mc, "tc": tc}
        if profile:
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
            if d["flops"] > 0:
                achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
                roof = {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": peak,
                        "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                        "avg_launch_us": 1e3 * d["ms"] / d["launches"],
                        "flops_per_launch": d["flops"] / d["launches"]}
            else:
                achieved = d["bytes"] / (d["ms"] * 1e-3) / 1e9
                roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": achieved / PEAK_HBM_GBS, "traffic": None,
                        "avg_launch_us": 1e3 * d["ms"] / d["launches"]}
            reps = 5  # cora_profile_step repeats the step 5 times
            res["roofline"] = roof
            res["kernel_ms_per_inference_step"] = {k: v["ms"] / reps for k, v in kinds.items()}
            # bandwidth-bound kernels of the step: algorithmic bytes / measured launch time
            res["hbm_kernels"] = {k: {"achieved_GBs": v["bytes"] / (v["ms"] * 1e-3) / 1e9,
                                      "frac_of_hbm_peak": v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                      "bytes_per_launch": v["bytes"] / v["launches"],
                                      "avg_launch_us": 1e3 * v["ms"] / v["launches"]}
                                  for k, v in kinds.items() if v["flops"] == 0 and v["bytes"] > 0}
            res["inference_step_kernel_ms_total"] = sum(v["ms"] for v in kinds.values()) / reps
    model.close()
    del x_dev
    return res


def offline_traffic(roof, cfg_name, B, over):
    """HBM bytes per launch of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (scripts/pmc_traffic.sh), committed under profiles/.  FETCH_SIZE reads 1/2 of wide coalesced reads on
    gfx950 (MI355X_MICROARCH.md, HBM) -> doubled; both counters are in KB.  Offline: not measured in this run."""
    if over or B != 200 or roof.get("kernel") != "pc_block_fused":
        return
    # (file, kernel-name prefix) per benchmarked configuration
    table = {"6block": (("r2_pmc_traffic.json", "r1_pmc_traffic.json"), "void pc_block_fused_kernel<0"),
             "12block": (("r2_pmc_traffic_12block.json",), "void pc_block_wide2_kernel")}
    if cfg_name not in table:
        return
    files, prefix = table[cfg_name]
    for fn in files:
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", fn)))
        except Exception:
            continue
        key = next((k for k in tj if k.startswith(prefix)), None)
        if key and "WRITE_SIZE_per_launch_KB" in tj[key] and "FETCH_SIZE_per_launch_KB" in tj[key]:
            roof["traffic"] = 1024.0 * (2.0 * tj[key]["FETCH_SIZE_per_launch_KB"] + tj[key]["WRITE_SIZE_per_launch_KB"])
            roof["traffic_source"] = f"profiles/{fn} (rocprofv3 PMC, offline)"
            return


def main():
    args = parse_args()
    in_child = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not in_child:
        # plain `python bench.py --gpus N`: become the launcher (nothing has touched the GPU yet)
        raise SystemExit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run_launch:
        return dry_run_rank(args, rank, world)

    # stdout carries exactly ONE line, the JSON result: libraries that print from C (RCCL's version banner at
    # communicator creation goes to stdout) are sent to stderr for the whole run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import cora_amd as ca
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    torch.cuda.set_device(local_rank)
    dist = None
    force_dp = os.environ.get("CORA_FORCE_DP") == "1"
    if world > 1 or force_dp:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if force_dp and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    over = {k: v for k, v in (("hidden_size", args.hidden_size), ("num_heads", args.num_heads),
                              ("num_blocks", args.num_blocks)) if v is not None}
    mc0 = ca.MODEL_CONFIGS[args.config]
    strong = args.global_batch is not None
    if strong:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} is not divisible by {world} ranks")
        B = args.global_batch // world
    else:
        B = args.batch or mc0.batch_size
    if (B * 64) % 4:
        raise SystemExit("per-GPU batch too ragged")

    main_res = measure(args, args.config, over, B, args.T, args.steps, args.warmup, rank, local_rank, world, dist)
    extra = None
    if not args.no_extra and not over and args.config == "6block":
        # the other half of BASELINE.json's metric ("6/12-block"): the canonical 12-block sweep point
        esteps, ewarm = max(3, args.steps // 3), max(2, min(args.warmup, 3))
        extra = measure(args, "12block", {}, B, args.T, esteps, ewarm, rank, local_rank, world, dist)
        if extra is not None:
            extra["steps"], extra["warmup"] = esteps, ewarm

    extra4 = None
    if extra is not None and world == 1 and args.dtype == "fp32":
        # BASELINE config 4's shape (ImageNet 224/16 preset, its own batch size) on the per-op path: fp32 against the
        # opt-in bf16 mode, three steps each -- the "bf16 mode >= 2 x fp32" figure, measured by the driver's own run
        mc4 = __import__("cora_amd").MODEL_CONFIGS["imagenet224"]
        extra4 = {}
        for dt in ("fp32", "bf16"):
            m4 = measure(args, "imagenet224", {}, mc4.batch_size, args.T, 3, 1, rank, local_rank, world, dist, dtype=dt)
            extra4[dt] = {k: m4[k] for k in ("workload", "value", "ms_per_step", "whole_step_mfma_frac", "kernel_ms_per_inference_step")}
            extra4[dt].update(steps=3, warmup=1, batch=mc4.batch_size)
        extra4["bf16_over_fp32"] = extra4["bf16"]["value"] / extra4["fp32"]["value"]

    if rank == 0:
        r = main_res
        mc, tc = r.pop("mc"), r.pop("tc")
        roof = r.pop("roofline")
        offline_traffic(roof, args.config, B, over)
        result = {
            "metric": "PC inference images x T-steps/sec (node)", "value": r["value"], "unit": "images*T-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16 operands / f32 accumulate",
                                         "bf16x3": "f32 as three bf16 terms per GEMM operand / f32 accumulate"}[args.dtype], "data": "synthetic",
            "config": {"workload": r["workload"], "global_batch": world * B, "T": r["T"],
                       "parallelism": f"dp{world}" + ("" if world == 1 and r["dp_mode"] == "single" else f" ({r['dp_mode']})"),
                       "rccl_world_size": (dist.get_world_size() if dist is not None else 1)},
            "infer_only_images_steps_per_s": r["infer_only_images_steps_per_s"],
            "algorithmic_flops_per_image_step": r["algorithmic_flops_per_image_step"],
            "whole_step_mfma_frac": r["whole_step_mfma_frac"],
            "roofline": roof,
            "kernel_ms_per_inference_step": r["kernel_ms_per_inference_step"],
            "hbm_kernels": r["hbm_kernels"],
            "inference_step_kernel_ms_total": r["inference_step_kernel_ms_total"],
        }
        if extra is not None:
            extra.pop("mc"); extra.pop("tc")
            offline_traffic(extra["roofline"], "12block", B, {})
            result["extra"] = {"cfg3": {k: extra[k] for k in (
                "workload", "value", "ms_per_step", "steps", "warmup", "whole_step_mfma_frac", "roofline",
                "infer_only_images_steps_per_s", "algorithmic_flops_per_image_step", "kernel_ms_per_inference_step",
                "dp_mode")}}
        if extra4 is not None:
            result["extra"]["cfg4"] = extra4
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(mc, tc, B, r["T"])
        else:
            result["cpu_baseline"] = None
        if "CORA_HIP_LIB" in os.environ:    # not a product number: the library was swapped for a diagnostic build
            result["diagnostic_library"] = os.environ["CORA_HIP_LIB"]
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
