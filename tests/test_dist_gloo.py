"""The N>1 path on CPU: world_size-2 gloo.  Batch sharding + one SUM all-reduce of the flat dE/dW
buffer must reproduce the unsharded gradient (the energy is a plain sum over images,
/root/reference/src/decoder_transformer.py:305), and every rank must end with identical weights."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cora_amd.dist import allreduce_sum_, shard_slice
from oracle import pcvit_np as NP
from tests.util import make_case


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _flat(wg, keys):
    return np.concatenate([wg[k].reshape(-1) for k in keys])


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = make_case((3, 16, 16), 4, 32, 2, 2, 4, seed=21)
    W = {k: v.astype(np.float64) for k, v in case["W"].items()}
    keys = sorted(W)
    sl = shard_slice(4, rank, world)
    h = [t[sl].astype(np.float64) for t in case["h"]]
    xp = NP.patchify(case["x"][sl].astype(np.float64), case["dims"])
    _, _, _, wg, _ = NP.step_grads(h, W, case["pos"].astype(np.float64), xp, case["dims"], case["hp"], want_wgrad=True)
    buf = torch.from_numpy(_flat(wg, keys).copy())
    w = allreduce_sum_(buf)
    q.put((rank, w, buf.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_wgrad_allreduce_matches_unsharded():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    case = make_case((3, 16, 16), 4, 32, 2, 2, 4, seed=21)
    W = {k: v.astype(np.float64) for k, v in case["W"].items()}
    keys = sorted(W)
    h = [t.astype(np.float64) for t in case["h"]]
    xp = NP.patchify(case["x"].astype(np.float64), case["dims"])
    _, _, _, wg, _ = NP.step_grads(h, W, case["pos"].astype(np.float64), xp, case["dims"], case["hp"], want_wgrad=True)
    full = _flat(wg, keys)
    for rank, w, buf in res:
        assert w == world
        np.testing.assert_allclose(buf, full, rtol=1e-10, atol=1e-12)
    np.testing.assert_array_equal(res[0][2], res[1][2])   # replicas identical -> identical AdamW


def test_shard_slice():
    assert shard_slice(200, 3, 8) == slice(75, 100)
    with pytest.raises(ValueError):
        shard_slice(10, 0, 4)


def test_allreduce_without_process_group_is_identity():
    t = torch.ones(4)
    assert allreduce_sum_(t) == 1 and torch.equal(t, torch.ones(4))


def test_bench_self_launch_two_ranks_cpu():
    """`python bench.py --gpus 2` started as a plain command must start its own two rank processes
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set per child, parent never touches the GPU), rendezvous
    and print ONE JSON line from rank 0.  --dry-run-launch swaps the GPU workload for a gloo barrier +
    MAX-reduce so the launcher path runs on a CPU-only host."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-launch",
                        "--global-batch", "200"], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["world_observed"] == 2 and r["max_over_ranks"] == 2.0
    assert r["scaling"] == "strong" and r["per_gpu_batch"] == 100 and r["global_batch"] == 200
    # the N > 1 line carries BOTH scalings (built by the same build_line / scaling_entry as the measured line): the
    # config's batch on every GPU (weak) and the reference's batch of 200 sharded over the ranks (strong), each with
    # its own global batch, and a parallelism string that names the data-parallel mode
    assert set(r["scalings"]) == {"weak", "strong"}
    wk, st = r["scalings"]["weak"], r["scalings"]["strong"]
    assert wk["scaling"] == "weak" and wk["config"]["global_batch"] == 400 and wk["config"]["per_gpu_batch"] == 200
    assert st["scaling"] == "strong" and st["config"]["global_batch"] == 200 and st["config"]["per_gpu_batch"] == 100
    assert r["config"]["global_batch"] == 200 and r["config"]["parallelism"].startswith("dp2 (")
    for e in (wk, st):
        assert {"value", "ms_per_step", "ms_per_step_median", "ms_per_step_min", "ms_per_step_max"} <= set(e)
        assert e["config"]["parallelism"].startswith("dp2 (")
    assert {"ms_per_step_median", "ms_per_step_min", "ms_per_step_max", "value_from_mean", "timing"} <= set(r)
    # a rank that fails takes the launcher down with a non-zero exit (no hang)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-launch",
                        "--global-batch", "7"], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0      # dry run does not validate divisibility: both ranks report and exit cleanly


def test_bench_line_names_graph_or_eager_collectives():
    """bench.build_line / scaling_entry (pure functions): the parallelism string of an N > 1 line carries the
    data-parallel mode bench.measure composed -- backend, clip mode and whether cora_step_dist REPLAYED graphs or
    launched its collectives eagerly (cora_dist_mode) -- so an eager number cannot pass for a graphed one."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    args = bench.parse_args(["--gpus", "8"])
    assert args.steps >= 50 and args.warmup >= 10          # SURVEY 8(d) protocol defaults

    def res(b, mode):
        return {"value": 1.0, "value_from_mean": 1.0, "ms_per_step": 1.0, "ms_per_step_median": 1.0, "ms_per_step_min": 1.0,
                "ms_per_step_max": 1.0, "steps": 50, "warmup": 10, "workload": "w", "per_gpu_batch": b, "T": 20,
                "dp_mode": mode, "whole_step_mfma_frac": 0.5}
    sc = {"weak": bench.scaling_entry(res(200, "rccl:exact-clip:graph"), 8, "weak"),
          "strong": bench.scaling_entry(res(25, "rccl:exact-clip:eager"), 8, "strong")}
    line = bench.build_line(args, 8, res(200, "rccl:exact-clip:graph"), False, sc, 8)
    assert line["scaling"] == "weak" and line["config"]["global_batch"] == 1600
    assert line["config"]["parallelism"] == "dp8 (rccl:exact-clip:graph)"
    assert line["scalings"]["strong"]["config"]["parallelism"] == "dp8 (rccl:exact-clip:eager)"
    assert line["scalings"]["strong"]["config"]["global_batch"] == 200
    assert bench.parallelism_name(1, "single") == "dp1"
