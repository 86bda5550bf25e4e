"""Batch-sharded data parallelism for the PC step: one process per GPU, RCCL over xGMI.

The reference is single-device; its step sums the energy over the whole batch
(`jax.lax.psum` over the vmap axis, /root/reference/src/decoder_transformer.py:305) and scales
dE/dW by 1/B (:484).  Sharding the batch over k ranks therefore needs exactly one exchange per
batch: the SUM over ranks of the flat dE/dW buffer, after which every rank applies the identical
clip + AdamW update with scale_by = 1/(B_local * k).  Inference (T vode steps) needs no collective.

Deviation, documented in DESIGN.md: optax.clip_by_global_norm on the vode gradients couples the
whole batch every inference step; here the norm is taken per shard ('local' mode), which is
identical whenever no shard clips.
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
    """Owns the torch view of the flat dE/dW buffer and runs the one collective of the step."""

    def __init__(self, model, batch_size: int, group=None):
        import torch
        import ctypes as C
        from . import _lib
        self.group = group
        ctx = model._ensure_ctx(batch_size)
        dev = torch.device("cuda", model.device)
        self.buf = torch.zeros(model._nW, dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        _lib.check(_lib.load().cora_bind_wgrad_buffer(ctx, C.c_void_p(self.buf.data_ptr()), self.buf.numel()), ctx)
        model._dp = self

    def allreduce(self) -> int:
        # cora_wgrad synchronised its stream before returning, so the buffer is complete here.
        import torch
        world = allreduce_sum_(self.buf, self.group)
        torch.cuda.synchronize(self.buf.device)
        return world
