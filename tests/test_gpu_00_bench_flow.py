"""`bench.py --gpus 2` end to end on the one-GPU box (rehearsal mode, see bench.py: CORA_BENCH_ONE_GPU).

This file sorts first among the GPU tests on purpose: the launcher is started while the pytest process has not
initialised HIP yet.  Starting child programs from a process that has (subprocess.run after the other GPU tests) faulted
intermittently on the GPU box; the launcher additionally goes through a fresh "spawn" interpreter, as the two-rank tests
of tests/test_gpu_dp_exact.py do."""
import os

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _bench_launcher(root, q):
    # a fresh interpreter (multiprocessing "spawn") that has not touched the GPU: it may start the launcher as a child.
    # subprocess from the pytest process itself -- which has initialised HIP by now -- faults on the GPU box.
    import subprocess
    import sys
    env = dict(os.environ, CORA_BENCH_ONE_GPU="1", CORA_BENCH_STALL_S="120")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--T", "4"], env=env, capture_output=True, text=True, timeout=280, cwd=root)
    q.put((out.returncode, out.stdout[-20000:], out.stderr[-2000:]))


def test_bench_two_rank_flow_on_one_gpu():
    """`bench.py --gpus 2` end to end with real measurements: the launcher, two ranks (here both on device 0 over
    gloo, CORA_BENCH_ONE_GPU=1 -- RCCL refuses two ranks on one device), sharded data, the barrier-bracketed timed
    region with MAX over ranks, both scalings in ONE line on rank 0's stdout.  The numbers are not throughputs (two
    ranks share a GPU) and the line says so; what is pinned is that the N > 1 flow runs and what its line holds."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    proc = ctx.Process(target=_bench_launcher, args=(root, q))
    proc.start()
    rc, stdout, stderr = q.get(timeout=300)
    proc.join(timeout=60)
    assert rc == 0, stderr
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "rehearsal" in line
    assert line["config"]["global_batch"] == 400 and line["config"]["parallelism"].startswith("dp2")
    assert set(line["scalings"]) == {"weak", "strong"}
    assert line["scalings"]["strong"]["config"]["global_batch"] == 200 and line["scalings"]["strong"]["config"]["per_gpu_batch"] == 100
    assert np.isfinite(line["value"]) and line["value"] > 0 and line["cpu_baseline"] is None
    assert set(line["extra"]["cfg3"]["scalings"]) == {"weak", "strong"}
