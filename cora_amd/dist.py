"""Batch-sharded data parallelism for the PC step: one process per GPU, RCCL over xGMI.

The reference is single-device; its step sums the energy over the whole batch
(`jax.lax.psum` over the vmap axis, /root/reference/src/decoder_transformer.py:305) and scales
dE/dW by 1/B (:484).  Sharding the batch over k ranks therefore needs exactly one exchange per
batch: the SUM over ranks of the flat dE/dW buffer, after which every rank applies the identical
clip + AdamW update with scale_by = 1/(B_local * k).  Inference (T vode steps) needs no collective.

optax.clip_by_global_norm on the vode gradients couples the whole batch every inference step.
Two modes: 'local' (fast: the whole T-step loop stays one captured graph and each shard clips by
its own norm -- identical to the reference only when no shard clips AND `use_vode_grad_norm` is
off) and `exact_clip=True` (the per-vode sums of squares are all-reduced between the gradient half
and the update half of every inference step -- the reference's trajectory for the global batch, at
the price of one host round trip per step).  `normalize_vode_h_gradients`
(/root/reference/src/decoder_transformer.py:353-389) divides every leaf by its WHOLE-batch norm on
every step, so with `use_vode_grad_norm` a sharded run is only meaningful in the exact mode:
DataParallel switches to it (with a warning) when the world has more than one rank.
"""
from __future__ import annotations

from typing import Optional


def allreduce_sum_(tensor, group=None) -> int:
    """In-place SUM all-reduce of a torch tensor over the process group; returns the world size.
    Backend-agnostic on purpose: RCCL ("nccl") on the GPUs, gloo in the CPU tests."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return 1
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    return dist.get_world_size(group)


def shard_slice(batch: int, rank: int, world: int) -> slice:
    """Images [rank*B/k, (rank+1)*B/k) live on rank `rank` (vodes are persistent per slot)."""
    if batch % world:
        raise ValueError(f"global batch {batch} is not divisible by world size {world}")
    per = batch // world
    return slice(rank * per, (rank + 1) * per)


class DataParallel:
    """Binds a model to the data-parallel step.  Two transports:

    backend="rccl"  : RCCL bound inside libcora_hip.so (`cora_comm_init` / `cora_step_dist`): both
                      exchanges -- the per-step sums of squares of the exact clip and the dE/dW sum --
                      sit inside the captured hipGraphs.  torch.distributed (any backend) is only used
                      once, to hand rank 0's 128-byte RCCL id to the other ranks.
    backend="torch" : the step split at the exchange point (`cora_infer` / `cora_wgrad` / all-reduce by
                      torch.distributed / `cora_apply_wgrad`); with exact_clip one host round trip per
                      inference step.  Works with gloo (CPU tests, several ranks on one GPU).
    backend="auto"  : "torch" under a gloo group, "rccl" otherwise."""

    def __init__(self, model, batch_size: int, group=None, exact_clip: bool = False, backend: str = "auto"):
        import torch
        import ctypes as C
        from . import _lib
        import warnings
        import torch.distributed as dist
        self.group = group
        self.exact_clip = bool(exact_clip)
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        if backend == "auto":
            backend = "torch" if (dist.is_initialized() and dist.get_backend(group) == "gloo") else "rccl"
        if backend not in ("rccl", "torch"):
            raise ValueError("backend must be 'auto', 'rccl' or 'torch'")
        self.backend, self.world, self.rank = backend, world, rank
        self.in_library = backend == "rccl"
        if world > 1 and bool(model.config.use_vode_grad_norm) and not self.exact_clip:
            warnings.warn("use_vode_grad_norm normalises each vode gradient by its whole-batch norm on every "
                          "step: data-parallel shards need the batch-global sums -> exact_clip=True")
            self.exact_clip = True
        ctx = model._ensure_ctx(batch_size)
        model._dp = self
        if self.in_library:
            lib = _lib.load()
            ident = [None]
            # every rank asks for an id (rank 0's is the one used): that call needs librccl loaded, so a rank on
            # which it cannot be bound finds out HERE, and the ranks agree over the torch group before any of them
            # enters ncclCommInitRank -- where the others would wait for the missing rank until a timeout
            raw = C.create_string_buffer(128)
            rc = lib.cora_comm_unique_id(raw, 128)
            ok = 1 if rc == 0 else 0
            if world > 1:
                on_gpu = dist.get_backend(group) != "gloo"
                flag = torch.tensor([ok], dtype=torch.int32, device=torch.device("cuda", model.device) if on_gpu else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                ok = int(flag.item())
            if not ok:
                model._dp = None
                if rc != 0:
                    _lib.check(rc, None)
                raise RuntimeError("in-library RCCL could not be bound on every rank")
            if rank == 0:
                ident[0] = raw.raw
            if world > 1:
                dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            _lib.check(lib.cora_comm_init(ctx, ident[0], 128, rank, world), ctx)
            self.buf = None
            self._agreed_T = set()
            return
        dev = torch.device("cuda", model.device)
        self.buf = torch.zeros(model._nW, dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        _lib.check(_lib.load().cora_bind_wgrad_buffer(ctx, C.c_void_p(self.buf.data_ptr()), self.buf.numel()), ctx)

    def agree_on_graphs(self, model, T: int) -> str:
        """In-library RCCL step: after the first (eager) cora_step_dist, every rank captures -- without
        launching -- the graphs the next call would replay (cora_dist_prepare) and the ranks agree over the
        torch group: graph replay only if EVERY rank could capture, eager collectives everywhere otherwise.
        The decision is taken before any replay, over a transport that is independent of the captured
        collectives, so no rank can sit in a graphed all-reduce waiting for one that fell back.
        Returns "graph" or "eager"; a no-op after the first call for a given T."""
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _lib
        if not self.in_library:
            return "n/a"
        lib, ctx = _lib.load(), model._ctx
        if T in self._agreed_T:
            return "graph" if lib.cora_dist_mode(ctx) != 1 else "eager"
        rc = lib.cora_dist_prepare(ctx, T, int(self.exact_clip))
        if rc < 0:
            _lib.check(rc, ctx)
        ready = 1 if rc == 0 else 0
        if self.world > 1 and dist.is_initialized():
            on_gpu = dist.get_backend(self.group) != "gloo"
            flag = torch.tensor([ready], dtype=torch.int32, device=torch.device("cuda", model.device) if on_gpu else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            ready = int(flag.item())
        if not ready:
            _lib.check(lib.cora_dist_set_eager(ctx, 1), ctx)
        self._agreed_T.add(T)
        return "graph" if ready else "eager"

    def step_mode(self, model) -> str:
        """How the last in-library step ran: "graph" (collectives replayed as graph nodes) or "eager"."""
        from . import _lib
        if not self.in_library:
            return "split"
        return {1: "eager", 2: "graph"}.get(_lib.load().cora_dist_mode(model._ctx), "none")

    def allreduce(self) -> int:
        # cora_wgrad synchronised its stream before returning, so the buffer is complete here.
        import torch
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_backend(self.group) == "gloo":
            host = self.buf.cpu()            # gloo (CPU tests, several ranks on one GPU): stage through the host
            world = allreduce_sum_(host, self.group)
            self.buf.copy_(host)
        else:
            world = allreduce_sum_(self.buf, self.group)
        torch.cuda.synchronize(self.buf.device)
        return world

    def infer_exact(self, model, ctx, x, T: int, eta_h, out) -> None:
        """The T-step loop with the batch-global clip norm (cora_infer_begin/.../end)."""
        import ctypes as C
        import numpy as np
        import torch
        import torch.distributed as dist
        from . import _lib
        lib = _lib.load()
        fp = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(lib.cora_infer_begin(ctx, fp(x), 0, T, fp(eta_h)), ctx)
        row = np.zeros((model.L + 2) * 8, dtype=np.float64)
        gpu_coll = dist.is_initialized() and dist.get_backend(self.group) != "gloo"
        for t in range(T):
            _lib.check(lib.cora_infer_grad(ctx, t), ctx)
            _lib.check(lib.cora_infer_sumsq(ctx, t, fp(row), row.size, 0), ctx)
            tt = torch.from_numpy(row)
            if gpu_coll:
                tt = tt.to(self.buf.device)
            allreduce_sum_(tt, self.group)
            row[:] = tt.cpu().numpy()
            _lib.check(lib.cora_infer_sumsq(ctx, t, fp(row), row.size, 1), ctx)
            _lib.check(lib.cora_infer_update(ctx, t), ctx)
        _lib.check(lib.cora_infer_end(ctx, C.byref(out)), ctx)
