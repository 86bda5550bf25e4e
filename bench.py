#!/usr/bin/env python3
"""bench.py -- PC inference images x T-steps / second for the PC-ViT step on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one whole train_on_batch-equivalent call on one synthetic batch: T inference steps
(hipGraph) + dE/dW + (RCCL all-reduce when N > 1) + clip/AdamW + "forward" reconstruction and MSE.
Inputs are resident in HBM when the timed region starts (x is a device tensor).

Workload of the headline line: BASELINE.json configs[1], the 6-block PC-ViT (64 tokens, fp32, T=20) at
B=200 images per GPU (weak scaling, the default) or, with --global-batch G, G images sharded over the
ranks (strong scaling: what north_star / configs[2] describe).  With N > 1 BOTH scalings are measured in
the same run and reported under "scalings" (the top-level fields are the one the flags select).

Timing: W untimed warm-up calls, then K calls bracketed by barrier + device synchronisation;
every call is also timed on its own (perf_counter around the synchronising call, MAX over ranks per call).
`value` = B_total * T / MEDIAN call time; `ms_per_step` is the mean over the bracket (the contract's K-step
region), `ms_per_step_median/min/max` sit next to it.

"extra" (N = 1 unless noted): cfg3 = the canonical 12-block sweep point (also at N > 1); cfg3_grid = all nine
(D, H) points of sweeps/sweep_12block_comprehensive.yaml; cfg4 / cfg5 = the ImageNet-preset and 16-frame video
shapes in fp32 and in the opt-in bf16 mode; small_batch = cfg 2 / cfg 3 at B = 25 / 50 / 100 (what one GPU of
a 8 / 4 / 2-GPU strong-scaling run of the reference's batch of 200 would see).
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
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense
PEAK_BF16_MATRIX_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16 MFMA dense (the 5 PF headline figure is 2:1 sparse)
PEAK_HBM_GBS = 8000.0
REFERENCE_BATCH = 200             # /root/reference/src/config.py:474, sweeps/sweep_12block_comprehensive.yaml:15-16

DTYPE_NAMES = {"fp32": "f32", "bf16": "bf16 operands / f32 accumulate",
               "bf16x3": "f32 as three bf16 terms per GEMM operand / f32 accumulate"}


def flops_per_image_step(N, P, D, L):
    """S = 8 N P D + L (48 N D^2 + 12 N^2 D)."""
    return 8.0 * N * P * D + L * (48.0 * N * D * D + 12.0 * N * N * D)


# ---- watchdog: a rank that makes no progress exits non-zero instead of waiting for a peer forever ----------
_progress = [time.time()]


def tick():
    _progress[0] = time.time()


def start_watchdog():
    """A rank whose peer died (or fell out of a collective) would otherwise sit in hipStreamSynchronize until the
    launcher's overall deadline.  Every call of the measured loops ticks; no tick for CORA_BENCH_STALL_S seconds
    (default 240) ends this process with exit code 75, and the launcher (ours or torch.distributed.run) then stops
    the other ranks."""
    limit = float(os.environ.get("CORA_BENCH_STALL_S", "240"))

    def watch():
        while True:
            time.sleep(5.0)
            if time.time() - _progress[0] > limit:
                sys.stderr.write(f"[bench] rank {os.environ.get('RANK', '0')}: no progress for {limit:.0f} s -- "
                                 "a peer rank is gone or a collective is stuck; exiting 75\n")
                sys.stderr.flush()
                os._exit(75)
    threading.Thread(target=watch, daemon=True).start()


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
    tick()
    n, t0 = 0, time.perf_counter()
    while True:
        R.train_on_batch(T, x, W, h, pos, dims, hp, eta, mc.peak_lr_weights, dtype=torch.float32)
        tick()
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
    ap.add_argument("--steps", type=int, default=50, help="timed calls (SURVEY 8(d): >= 50)")
    ap.add_argument("--warmup", type=int, default=10, help="untimed warm-up calls (SURVEY 8(d): >= 10)")
    ap.add_argument("--config", default="6block")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: the config's batch_size)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="strong scaling: this many images in total, sharded over the ranks (north_star: the batch is sharded)")
    ap.add_argument("--T", type=int, default=None)
    ap.add_argument("--hidden-size", type=int, default=None, help="override the config (12-block sweep points)")
    ap.add_argument("--num-heads", type=int, default=None)
    ap.add_argument("--num-blocks", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="headline workload only (no cfg3 / grid / cfg4 / cfg5 / small-batch legs)")
    ap.add_argument("--sweep-grid", dest="sweep_grid", action="store_true", default=None,
                    help="measure all nine (D, H) points of the 12-block sweep grid (default: on at N = 1 for the 6block line)")
    ap.add_argument("--no-sweep-grid", dest="sweep_grid", action="store_false")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "bf16x3"],
                    help="fp32 = parity path (the headline number); bf16 = opt-in bf16-MFMA mode; bf16x3 = opt-in "
                         "three-term bf16 split in the per-op GEMMs (fp32-grade products; priced against the fp32 peak)")
    ap.add_argument("--clip-mode", default="exact", choices=["exact", "local"],
                    help="N > 1: 'exact' all-reduces the vode-gradient sums of squares every inference step "
                         "(the reference's whole-batch clip_by_global_norm), 'local' clips per shard")
    ap.add_argument("--dp-backend", default="rccl", choices=["rccl", "torch"],
                    help="N > 1: 'rccl' = collectives inside the captured graphs (cora_step_dist), "
                         "'torch' = step split at the exchange point, torch.distributed all-reduce")
    ap.add_argument("--dist-graphs", action="store_true",
                    help="N > 1 with --dp-backend rccl: replay the step as two captured graphs with the collectives as graph "
                         "nodes (agreed across ranks first).  Default at N > 1: eager launches of the same sequence -- measured "
                         "equal on one GPU (DESIGN.md section 5) and the mode every RCCL deployment exercises")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="rank processes only rendezvous (gloo) and report: exercises the launcher and the shape of "
                         "the N > 1 line without a GPU")
    return ap.parse_args(argv)


# ---- launcher: `python bench.py --gpus N` starts N fresh rank processes BEFORE anything touches the GPU ----
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int) -> int:
    """One child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, as
    torch.distributed.run would set them).  The parent never initialises HIP; children are started as
    new processes (no fork of a GPU process, no exec after GPU init).  Rank 0's stdout is ours.  A rank that
    stalls ends itself (start_watchdog), and one dead rank makes the launcher stop the others."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CORA_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + 1500
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


# ---- the JSON line (pure functions: the CPU tests build an N > 1 line from synthetic measurements) ---------
def scaling_entry(res, world, scaling):
    """One measured scaling of the data-parallel step: what `scalings.weak` / `scalings.strong` hold."""
    return {"scaling": scaling, "value": res["value"], "unit": "images*T-steps/s", "ms_per_step": res["ms_per_step"],
            "ms_per_step_median": res["ms_per_step_median"], "ms_per_step_min": res["ms_per_step_min"],
            "ms_per_step_max": res["ms_per_step_max"], "steps": res["steps"], "warmup": res["warmup"],
            "config": {"workload": res["workload"], "global_batch": world * res["per_gpu_batch"],
                       "per_gpu_batch": res["per_gpu_batch"], "T": res["T"],
                       "parallelism": parallelism_name(world, res["dp_mode"])},
            "whole_step_mfma_frac": res["whole_step_mfma_frac"]}


def parallelism_name(world, dp_mode):
    return f"dp{world}" + ("" if world == 1 and dp_mode == "single" else f" ({dp_mode})")


def build_line(args, world, main, strong, scalings, rccl_world):
    """The contract's JSON object from the headline measurement `main` (+ optional sections added by the caller)."""
    line = {
        "metric": "PC inference images x T-steps/sec (node)", "value": main["value"], "unit": "images*T-steps/s",
        "n_gpus": world, "steps": main["steps"], "warmup": main["warmup"],
        "ms_per_step": main["ms_per_step"], "ms_per_step_median": main["ms_per_step_median"],
        "ms_per_step_min": main["ms_per_step_min"], "ms_per_step_max": main["ms_per_step_max"],
        "timing": "value = global_batch * T / MEDIAN per-call time (perf_counter around each synchronising call, MAX over "
                  "ranks per call); ms_per_step = mean over the barrier-bracketed K-call region",
        "value_from_mean": main["value_from_mean"],
        "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": DTYPE_NAMES[args.dtype], "data": "synthetic",
        "config": {"workload": main["workload"], "global_batch": world * main["per_gpu_batch"], "T": main["T"],
                   "parallelism": parallelism_name(world, main["dp_mode"]), "rccl_world_size": rccl_world},
    }
    if scalings:
        line["scalings"] = scalings
    return line


def dry_run_rank(args, rank, world):
    """Launcher rehearsal (CPU, gloo): rendezvous, barrier, MAX over ranks, and rank 0 prints a line with the
    SHAPE of the real N > 1 line (both scalings, built by the same build_line / scaling_entry) from synthetic
    timings -- marked "dry_run", no throughput claim."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    strong = args.global_batch is not None
    B = (args.global_batch // world) if strong else (args.batch or REFERENCE_BATCH)

    def fake(b):
        ms = float(t.item())
        return {"value": world * b * 20 / (ms * 1e-3), "value_from_mean": world * b * 20 / (ms * 1e-3), "ms_per_step": ms,
                "ms_per_step_median": ms, "ms_per_step_min": ms, "ms_per_step_max": ms, "steps": args.steps, "warmup": args.warmup,
                "workload": "dry run (no GPU work)", "per_gpu_batch": b, "T": 20, "dp_mode": "dry-run", "whole_step_mfma_frac": None}
    if rank == 0:
        main = fake(B)
        scalings = None
        if world > 1:
            scalings = {"weak": scaling_entry(fake(args.batch or REFERENCE_BATCH), world, "weak")}
            gb = args.global_batch or REFERENCE_BATCH
            if gb % world == 0:
                scalings["strong"] = scaling_entry(fake(gb // world), world, "strong")
        line = build_line(args, world, main, strong, scalings, dist.get_world_size())
        line.update({"dry_run": True, "world_observed": dist.get_world_size(), "max_over_ranks": float(t.item()),
                     "per_gpu_batch": B, "global_batch": B * world})
        print(json.dumps(line), flush=True)
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
    tick()

    def build(backend):
        """Model + data-parallel binding.  backend: None (one GPU), "rccl" (collectives inside the library's
        graphs) or "torch" (step split at the exchange point, per-shard clip).  Whether RCCL could be bound
        inside the library is agreed over the torch group BEFORE any step runs (a rank that fell back alone
        would leave its peers inside a collective)."""
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
                bound = 1
                try:
                    dp = DataParallel(model, B, exact_clip=exact, backend="rccl")
                except Exception as e:       # RCCL could not be bound inside the library on this rank
                    print(f"[bench] rank {rank}: in-library RCCL unavailable ({e})", file=sys.stderr)
                    bound = 0
                if dist is not None and dist.is_initialized() and world > 1:
                    flag = torch.tensor([bound], dtype=torch.int32, device="cuda")
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    bound = int(flag.item())
                if not bound:
                    print("[bench] falling back to the torch.distributed data-parallel path on every rank", file=sys.stderr)
                    model.close()
                    return build("torch")
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
        tick()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        tick()

    # Warm-up.  In-library RCCL: the first call runs eagerly, then the ranks agree (over the torch group, before any
    # replay) on graph replay vs eager collectives.  A step that fails here is an error, not a reason to fall back:
    # the peers may already be inside a collective, and the watchdog / launcher end the run.
    # Graph replay of the collectives is the default on one GPU (CORA_FORCE_DP, tested there) and opt-in (--dist-graphs)
    # with more than one rank: it has never run on > 1 GPU and measures equal to eager launches on one.
    dist_graphs = dp is not None and dp.in_library and (world == 1 or args.dist_graphs)
    if dp is not None and dp.in_library and not dist_graphs:
        _lib.check(lib.cora_dist_set_eager(ctx, 1), ctx)
    for k in range(max(warmup, 1)):
        one_step()
        if k == 0 and dist_graphs:
            dp.agree_on_graphs(model, T)
    barrier()
    per_call = np.zeros(steps, dtype=np.float64)
    infer_ms = 0.0
    trace = []
    t0 = time.perf_counter()
    for k in range(steps):
        c0 = time.perf_counter()
        one_step()                       # ends with a stream synchronisation: the call time is the device time + host
        per_call[k] = time.perf_counter() - c0
        infer_ms += out.infer_ms
        trace.append((out.final_h_energy, out.final_w_energy, out.train_mse))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor(np.concatenate([[elapsed], per_call]), dtype=torch.float64,
                         device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t = t.cpu().numpy()
        elapsed, per_call = float(t[0]), t[1:]
    if os.environ.get("CORA_TRACE") == "1":
        print("trace:", trace, file=sys.stderr)
    timing_build = "CORA_HIP_LIB" in os.environ      # diagnostic libraries may compute garbage on purpose
    if not timing_build and (not np.isfinite(out.final_h_energy) or not np.isfinite(out.train_mse)):
        raise SystemExit(f"non-finite energy / mse in the timed region: {trace}")
    if dp is not None and dp.in_library:
        mode += ":" + dp.step_mode(model)            # "graph" (collectives replayed as graph nodes) or "eager"

    res = None
    if rank == 0:
        bf16 = dtype == "bf16"
        peak = PEAK_BF16_MATRIX_TFLOPS if bf16 else PEAK_FP32_MATRIX_TFLOPS
        med = float(np.median(per_call))
        value = world * B * T / med
        S = flops_per_image_step(model.N, model.P, model.D, model.L)
        res = {"value": value, "value_from_mean": world * B * T * steps / elapsed,
               "ms_per_step": 1e3 * elapsed / steps, "ms_per_step_median": 1e3 * med,
               "ms_per_step_min": 1e3 * float(per_call.min()), "ms_per_step_max": 1e3 * float(per_call.max()),
               "steps": steps, "warmup": max(warmup, 1), "per_gpu_batch": B, "T": T, "kernel_path": model.kernel_path(),
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
            if dom == "pc_block_fused" and not os.environ.get("CORA_NO_LAZY") and not os.environ.get("CORA_PAIR"):
                # in a T-step loop this launch also applies the previous step's momentum-SGD update to
                # the rows it loads (what used to be the separate, HBM-bound update launch), at unchanged MFMA work
                roof["note"] = "launch carries the previous step's vode update (no separate update launch)"
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
    tick()
    return res


def offline_traffic(roof, cfg_name, B, over):
    """HBM bytes per launch of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (scripts/pmc_traffic.sh), committed under profiles/.  FETCH_SIZE reads 1/2 of wide coalesced reads on
    gfx950 (MI355X_MICROARCH.md, HBM) -> doubled; both counters are in KB.  Offline: not measured in this run."""
    if over or B != 200 or roof.get("kernel") != "pc_block_fused":
        return
    # (file, kernel-name prefix) per benchmarked configuration
    table = {"6block": (("r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json"), ("void pc_block_fused_kernel<3", "void pc_block_fused_kernel<0")),
             "12block": (("r3_pmc_traffic_12block.json", "r2_pmc_traffic_12block.json",), ("void pc_block_wide2_kernel",))}
    if cfg_name not in table:
        return
    files, prefixes = table[cfg_name]
    for fn in files:
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", fn)))
        except Exception:
            continue
        key = next((k for pre in prefixes for k in tj if k.startswith(pre)), None)
        if key and "WRITE_SIZE_per_launch_KB" in tj[key] and "FETCH_SIZE_per_launch_KB" in tj[key]:
            roof["traffic"] = 1024.0 * (2.0 * tj[key]["FETCH_SIZE_per_launch_KB"] + tj[key]["WRITE_SIZE_per_launch_KB"])
            roof["traffic_source"] = f"profiles/{fn} (rocprofv3 PMC, offline)"
            return


BRIEF = ("workload", "value", "ms_per_step", "ms_per_step_median", "steps", "warmup", "whole_step_mfma_frac", "kernel_path")


def brief(res, *more):
    return {k: res[k] for k in BRIEF + more if k in res}


def main():
    args = parse_args()
    in_child = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not in_child:
        # plain `python bench.py --gpus N`: become the launcher (nothing has touched the GPU yet)
        raise SystemExit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # rehearsal of the N > 1 flow on a one-GPU box: every rank on device 0, gloo for the process group and the
    # torch data-parallel backend (RCCL refuses two ranks on one device).  Exercises sharding, both scalings, the
    # extras and the line; its throughput means nothing and the line says so.
    one_gpu = os.environ.get("CORA_BENCH_ONE_GPU") == "1" and world > 1
    if one_gpu:
        local_rank = 0
        args.dp_backend = "torch"
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
    tick()
    start_watchdog()
    dist = None
    force_dp = os.environ.get("CORA_FORCE_DP") == "1"
    if world > 1 or force_dp:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if force_dp and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        tick()

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
    ctx_args = (rank, local_rank, world, dist)
    headline = args.config == "6block" and not over
    extras_on = not args.no_extra and headline

    main_res = measure(args, args.config, over, B, args.T, args.steps, args.warmup, *ctx_args)

    # N > 1: the other scaling of the same workload in the same run (weak = the config's batch on every GPU,
    # strong = the reference's batch of 200 sharded over the ranks).  Decided from the flags alone, so every rank
    # takes the same path (measure() returns a result on rank 0 only).
    other_name, other_B = None, None
    if world > 1 and not args.no_extra:
        if strong:
            other_name, other_B = "weak", (args.batch or mc0.batch_size)
        else:
            gb = REFERENCE_BATCH if args.batch is None else args.batch
            if gb % world == 0:
                other_name, other_B = "strong", gb // world
    other_res = None
    if other_B:
        other_res = measure(args, args.config, over, other_B, args.T, max(5, args.steps // 2), max(3, args.warmup // 2),
                            *ctx_args, profile=False)

    extra = extra_other = None
    if extras_on:
        # the other half of BASELINE.json's metric ("6/12-block"): the canonical 12-block sweep point
        esteps, ewarm = max(5, args.steps // 5), max(2, min(args.warmup, 3))
        extra = measure(args, "12block", {}, B, args.T, esteps, ewarm, *ctx_args)
        if other_B:
            extra_other = measure(args, "12block", {}, other_B, args.T, 5, 2, *ctx_args, profile=False)

    single = world == 1 and args.dtype == "fp32" and extras_on
    extra4 = extra5 = grid = small = None
    if single:
        # BASELINE configs 4 and 5 (ImageNet 224/16 preset; 16-frame video, N = 1024) at their own batch sizes:
        # fp32 against the opt-in bf16 mode, a few steps each, measured by the driver's own run
        extra4, extra5 = {}, {}
        for name, dst in (("imagenet224", extra4), ("video16", extra5)):
            mcx = ca.MODEL_CONFIGS[name]
            for dt in ("fp32", "bf16"):
                m = measure(args, name, {}, mcx.batch_size, args.T, 4, 2, *ctx_args, dtype=dt)
                dst[dt] = brief(m, "kernel_ms_per_inference_step", "roofline", "infer_only_images_steps_per_s")
                dst[dt]["batch"] = mcx.batch_size
            dst["bf16_over_fp32"] = dst["bf16"]["value"] / dst["fp32"]["value"]
        # strong-scaling cost made visible on ONE GPU: the per-GPU batch of an 8 / 4 / 2-GPU run of the reference's
        # batch of 200 (north_star shards THE batch); 8 x the B=25 figure is the ceiling of that 8-GPU run
        small = {}
        for name, full in (("6block", main_res), ("12block", extra)):
            rows = {}
            for b in (25, 50, 100):
                m = measure(args, name, {}, b, args.T, 5, 3, *ctx_args, profile=False)
                rows[f"B{b}"] = {"value": m["value"], "ms_per_step_median": m["ms_per_step_median"],
                                 "per_image_rate_vs_B200": m["value"] / full["value"]}
            rows["strong_scaling_ceiling_8gpu"] = 8.0 * rows["B25"]["value"]
            rows["strong_scaling_ceiling_4gpu"] = 4.0 * rows["B50"]["value"]
            rows["strong_scaling_ceiling_2gpu"] = 2.0 * rows["B100"]["value"]
            small["cfg2" if name == "6block" else "cfg3"] = rows
    if (args.sweep_grid if args.sweep_grid is not None else single) and world == 1:
        # BASELINE config 3 is the sweep GRID (sweeps/sweep_12block_comprehensive.yaml:8-47), not only its canonical point
        grid = {}
        for D in (64, 96, 128):
            for H in (1, 2, 4):
                m = measure(args, "12block", {"hidden_size": D, "num_heads": H}, REFERENCE_BATCH, args.T, 3, 2, *ctx_args)
                grid[f"D{D}_H{H}"] = {"value": m["value"], "ms_per_step_median": m["ms_per_step_median"],
                                      "whole_step_mfma_frac": m["whole_step_mfma_frac"], "kernel_path": m["kernel_path"],
                                      "dominant_kernel": m["roofline"]["kernel"], "dominant_frac": m["roofline"]["frac"]}

    if rank == 0:
        r = main_res
        mc, tc = r.pop("mc"), r.pop("tc")
        roof = r.pop("roofline")
        offline_traffic(roof, args.config, B, over)
        scalings = None
        if world > 1:
            scalings = {"strong" if strong else "weak": scaling_entry(r, world, "strong" if strong else "weak")}
            if other_res is not None:
                scalings[other_name] = scaling_entry(other_res, world, other_name)
        result = build_line(args, world, r, strong, scalings, dist.get_world_size() if dist is not None else 1)
        result.update({
            "infer_only_images_steps_per_s": r["infer_only_images_steps_per_s"],
            "algorithmic_flops_per_image_step": r["algorithmic_flops_per_image_step"],
            "whole_step_mfma_frac": r["whole_step_mfma_frac"],
            "roofline": roof,
            "kernel_ms_per_inference_step": r["kernel_ms_per_inference_step"],
            "hbm_kernels": r["hbm_kernels"],
            "inference_step_kernel_ms_total": r["inference_step_kernel_ms_total"],
        })
        if extra is not None:
            extra.pop("mc"); extra.pop("tc")
            offline_traffic(extra["roofline"], "12block", B, {})
            result["extra"] = {"cfg3": brief(extra, "roofline", "infer_only_images_steps_per_s", "algorithmic_flops_per_image_step",
                                             "kernel_ms_per_inference_step", "dp_mode", "ms_per_step_min", "ms_per_step_max")}
            if extra_other is not None:
                result["extra"]["cfg3"]["scalings"] = {
                    "strong" if strong else "weak": scaling_entry(extra, world, "strong" if strong else "weak"),
                    other_name: scaling_entry(extra_other, world, other_name)}
        for key, val in (("cfg4", extra4), ("cfg5", extra5), ("cfg3_grid", grid), ("small_batch", small)):
            if val is not None:
                result.setdefault("extra", {})[key] = val
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(mc, tc, B, r["T"])
        else:
            result["cpu_baseline"] = None
        if one_gpu:
            result["rehearsal"] = f"{world} ranks sharing ONE GPU over gloo (CORA_BENCH_ONE_GPU=1): flow check, not a throughput"
        if "CORA_HIP_LIB" in os.environ:    # not a product number: the library was swapped for a diagnostic build
            result["diagnostic_library"] = os.environ["CORA_HIP_LIB"]
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    if dist is not None:
        tick()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
