"""Data-parallel parity on the GPU: 2 ranks x B/2 images must reproduce 1 rank x B images.

The reference is single-device; its semantics for a global batch are: energy summed over the batch
(/root/reference/src/decoder_transformer.py:305), vode gradients clipped by the norm over the WHOLE
batch every inference step (examples/debug_transformer_wandb.py:1137-1140), dE/dW scaled by 1/B
(:484).  With `DataParallel(exact_clip=True)` the sharded run must land on the same vodes and the
same updated weights as the unsharded one, clip firing on every step.  Two processes share the one
GPU of the test box and exchange through gloo (RCCL refuses two ranks on one device)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests.util import build_model, make_case

pytestmark = pytest.mark.gpu

SHAPES = {
    "fused_shape": ((3, 32, 32), 4, 64, 1, 2, 4, 3),
    "general_h2": ((3, 16, 16), 4, 32, 2, 2, 4, 3),
    "rows_n256_h2": ((4, 3, 32, 32), 4, 64, 2, 2, 2, 3),     # row-tile fused kernels: 2 ranks x 1 image of 256 tokens
}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _case(name):
    gn = name.endswith("+gradnorm")         # use_vode_grad_norm: every leaf normalised by its WHOLE-batch norm
    shape, p, D, H, L, B, T = SHAPES[name.split("+")[0]]
    return make_case(shape, p, D, H, L, B, seed=41, clip_h=3.0, clip_w=0.5, use_grad_norm=gn, grad_norm_target=2.0), T


def _worker(rank, world, port, name, exact, q):
    import torch.distributed as dist
    import cora_amd as ca
    from cora_amd.dist import DataParallel, shard_slice
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case, T = _case(name)
    B = case["x"].shape[0]
    sl = shard_slice(B, rank, world)
    shard = dict(case)
    shard["x"] = case["x"][sl]
    shard["h"] = [t[sl] for t in case["h"]]
    model, optim_w, optim_h = build_model(shard)
    DataParallel(model, B // world, exact_clip=exact)
    Eh, Ew, gh, gw, mse = ca.train_on_batch(T, shard["x"], model=model, optim_w=optim_w, optim_h=optim_h)
    hs = [model.vodes[l].h.get() for l in range(model.L + 2)]
    q.put((rank, hs, model.get_weights(), float(Eh)))
    dist.barrier()
    model.close()
    dist.destroy_process_group()


def _run_sharded(name, exact):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, exact, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("name", list(SHAPES))
def test_two_ranks_exact_clip_match_one_rank(name):
    import cora_amd as ca
    case, T = _case(name)
    model, optim_w, optim_h = build_model(case)
    Eh, Ew, gh, gw, mse = ca.train_on_batch(T, case["x"], model=model, optim_w=optim_w, optim_h=optim_h)
    L = model.L
    h_ref = [model.vodes[l].h.get() for l in range(L + 2)]
    w_ref = model.get_weights()
    model.close()

    res = _run_sharded(name, exact=True)
    for l in range(L + 2):
        got = np.concatenate([res[0][1][l], res[1][1][l]], axis=0)
        err = np.abs(got - h_ref[l]).max() / (np.abs(h_ref[l]).max() + 1e-30)
        assert err <= 2e-5, f"vode {l}: sharded vs unsharded rel err {err}"
    assert abs(res[0][3] + res[1][3] - Eh) <= 1e-5 * abs(Eh)          # energies add up over shards
    for k, v in w_ref.items():
        for r in res:      # replicas identical to the unsharded update (AdamW amplifies rounding of tiny gradients)
            assert np.abs(r[2][k] - v).max() <= 2e-4 * (np.abs(v).max() + 1e-30) + 1e-6, k
        assert np.array_equal(res[0][2][k], res[1][2][k]), f"replicas diverged on {k}"

    # the per-shard ('local') clip is a different trajectory when the clip fires: make sure the test can tell
    loc = _run_sharded(name, exact=False)
    got = np.concatenate([loc[0][1][L + 1], loc[1][1][L + 1]], axis=0)
    assert np.abs(got - h_ref[L + 1]).max() / np.abs(h_ref[L + 1]).max() > 1e-4


@pytest.mark.parametrize("name", list(SHAPES))
def test_split_step_single_rank_matches_fused_step(name):
    """World size 1, no process group: cora_infer(out=NULL) + cora_wgrad + identity all-reduce +
    cora_apply_wgrad (the data-parallel call sequence, inference loop left in flight until the weight
    gradients are enqueued) against the one-call cora_step, two batches each so the graphed path runs."""
    import cora_amd as ca
    from cora_amd.dist import DataParallel
    case, T = _case(name)
    res = []
    for split in (False, True):
        model, ow, oh = build_model(case)
        if split:
            DataParallel(model, case["x"].shape[0], backend="torch")
        outs = [ca.train_on_batch(T, case["x"], model=model, optim_w=ow, optim_h=oh) for _ in range(3)]
        res.append((outs[-1], [model.vodes[l].h.get() for l in range(len(case["h"]))], model.get_weights()))
        model.close()
    (a, ha, wa), (b, hb, wb) = res
    assert a[0] == pytest.approx(b[0], rel=1e-5) and a[1] == pytest.approx(b[1], rel=1e-5) and a[4] == pytest.approx(b[4], rel=1e-5)
    for x, y in zip(ha, hb):
        assert np.abs(x - y).max() <= 1e-4 * np.abs(x).max()
    for k in wa:
        assert np.abs(wa[k] - wb[k]).max() <= 1e-4 * max(np.abs(wa[k]).max(), 1e-3), k


@pytest.mark.parametrize("name", ["fused_shape+gradnorm", "general_h2+gradnorm"])
def test_two_ranks_grad_norm_match_one_rank(name):
    """`use_vode_grad_norm` (/root/reference/src/decoder_transformer.py:353-389) divides every leaf by its
    whole-batch norm on every step, so DataParallel must run the batch-global sums even when the caller
    asked for the local mode (it switches itself, with a warning): 2 ranks x B/2 == 1 rank x B."""
    import cora_amd as ca
    case, T = _case(name)
    model, optim_w, optim_h = build_model(case)
    ca.train_on_batch(T, case["x"], model=model, optim_w=optim_w, optim_h=optim_h)
    L = model.L
    h_ref = [model.vodes[l].h.get() for l in range(L + 2)]
    model.close()
    res = _run_sharded(name, exact=False)
    for l in range(L + 2):
        got = np.concatenate([res[0][1][l], res[1][1][l]], axis=0)
        err = np.abs(got - h_ref[l]).max() / (np.abs(h_ref[l]).max() + 1e-30)
        assert err <= 2e-5, f"vode {l}: sharded vs unsharded rel err {err}"


@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("exact", [True, False])
def test_in_library_rccl_world1_matches_single_gpu_step(name, exact):
    """`cora_step_dist` with a world-size-1 RCCL communicator (ncclCommInitRank with one rank; the
    all-reduces of the sums-of-squares rows and of dE/dW are captured inside the two graphs) must walk
    the trajectory of the one-GPU `cora_step`: four batches each, so the eager first call AND the graph
    replays run.  The 2-rank numerics are covered by the gloo tests above (RCCL refuses two ranks on one
    device); this one covers the captured-collective launch path itself."""
    import ctypes as C
    import cora_amd as ca
    from cora_amd import _lib
    from cora_amd.dist import DataParallel
    case, T = _case(name)
    res = []
    for dist_path in (False, True):
        model, ow, oh = build_model(case)
        if dist_path:
            dp = DataParallel(model, case["x"].shape[0], backend="rccl", exact_clip=exact)
            r, w = C.c_int(), C.c_int()
            _lib.check(_lib.load().cora_comm_info(model._ctx, C.byref(r), C.byref(w)), model._ctx)
            assert dp.in_library and (r.value, w.value) == (0, 1)
        outs, modes = [], []
        for _ in range(4):
            outs.append(ca.train_on_batch(T, case["x"], model=model, optim_w=ow, optim_h=oh))
            if dist_path:
                modes.append(dp.step_mode(model))
        if dist_path:
            # cora_dist_mode: the first call launches eagerly (RCCL warm-up), then -- the ranks having agreed through
            # cora_dist_prepare / DataParallel.agree_on_graphs -- every later call replays the captured graphs
            assert modes == ["eager", "graph", "graph", "graph"], modes
            assert dp.agree_on_graphs(model, T) == "graph"
            # an operator (or a rank that could not capture) can pin every rank to eager collectives
            _lib.check(_lib.load().cora_dist_set_eager(model._ctx, 1), model._ctx)
            outs.append(ca.train_on_batch(T, case["x"], model=model, optim_w=ow, optim_h=oh))
            assert dp.step_mode(model) == "eager"
            _lib.check(_lib.load().cora_dist_set_eager(model._ctx, 0), model._ctx)
        else:
            outs.append(ca.train_on_batch(T, case["x"], model=model, optim_w=ow, optim_h=oh))
        tot, per = ca.step_energies(model, T)
        res.append((outs, [model.vodes[l].h.get() for l in range(len(case["h"]))], model.get_weights(), tot,
                    model.get_adam_state()[2]))
        model.close()
    (a, ha, wa, ea, ca_), (b, hb, wb, eb, cb_) = res
    assert ca_ == cb_ == 5
    # same kernels on the same inputs; the only run-to-run freedom is the order of float atomics in the
    # split-K weight gradients and the slotted energy sums
    for oa, ob in zip(a, b):
        assert oa[0] == pytest.approx(ob[0], rel=1e-6) and oa[1] == pytest.approx(ob[1], rel=1e-6) and oa[4] == pytest.approx(ob[4], rel=1e-6)
    np.testing.assert_allclose(ea, eb, rtol=1e-6)
    for x, y in zip(ha, hb):
        assert np.abs(x - y).max() <= 2e-5 * np.abs(x).max()
    for k in wa:
        assert np.abs(wa[k] - wb[k]).max() <= 1e-4 * max(np.abs(wa[k]).max(), 1e-3), k


@pytest.mark.parametrize("name", ["fused_shape", "general_h2", "rows_n256_h2"])
@pytest.mark.parametrize("exact", [True, False])
def test_ipc_mode_in_library_rccl_world1_matches_single_gpu(name, exact):
    """MODE 1 (`update_weights_every_inference_step`, /root/reference/src/decoder_transformer.py:418-439) on the
    data-parallel entry point `cora_step_ipc_dist`: with a world-size-1 communicator the per-step all-reduces of the
    sums-of-squares row and of dE/dW are identities, so two batches must reproduce `cora_step_ipc` -- vodes, energies,
    the T AdamW steps per batch.  (More than one rank: same collectives as cora_step_dist, unverified on hardware.)"""
    import ctypes as C
    import cora_amd as ca
    from cora_amd.dist import DataParallel
    case, T = _case(name)
    res = []
    for dist_path in (False, True):
        model, ow, oh = build_model(case, ipc=True)
        if dist_path:
            dp = DataParallel(model, case["x"].shape[0], backend="rccl", exact_clip=exact)
            assert dp.in_library
        outs = [ca.train_on_batch(T, case["x"], model=model, optim_w=ow, optim_h=oh) for _ in range(2)]
        tot, per = ca.step_energies(model, T)
        res.append((outs, [model.vodes[l].h.get() for l in range(len(case["h"]))], model.get_weights(), tot,
                    model.get_adam_state()[2]))
        model.close()
    (a, ha, wa, ea, ca_), (b, hb, wb, eb, cb_) = res
    assert ca_ == cb_ == 2 * T
    for oa, ob in zip(a, b):
        assert oa[0] == pytest.approx(ob[0], rel=1e-6) and oa[1] == pytest.approx(ob[1], rel=1e-6) and oa[4] == pytest.approx(ob[4], rel=1e-6)
    np.testing.assert_allclose(ea, eb, rtol=1e-6)
    for x, y in zip(ha, hb):
        assert np.abs(x - y).max() <= 2e-5 * np.abs(x).max()
    for k in wa:
        assert np.abs(wa[k] - wb[k]).max() <= 1e-4 * max(np.abs(wa[k]).max(), 1e-3), k


def test_ipc_mode_needs_the_in_library_backend():
    """The split (torch.distributed) transport has no per-inference-step weight exchange: it must say so."""
    import cora_amd as ca
    from cora_amd.dist import DataParallel
    case, T = _case("general_h2")
    model, ow, oh = build_model(case, ipc=True)
    DataParallel(model, case["x"].shape[0], backend="torch")
    with pytest.raises(NotImplementedError, match="backend='rccl'"):
        ca.train_on_batch(T, case["x"], model=model, optim_w=ow, optim_h=oh)
    model.close()
