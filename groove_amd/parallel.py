"""Voice sharding across the GPUs of one node (no reference counterpart).

One process per GPU.  Voices are independent up to the main-mixer sum, so the project's
voices are cut into contiguous index ranges; every rank renders its range into a private bus
and the buses are summed ONCE per render (never per block: the message is frames*8 B and
latency-bound on xGMI).  On GPUs the sum is `groove_bus_reduce` (RCCL ncclReduce on the ctx
stream); `reduce_bus_host` is the same step over torch.distributed for host buffers (gloo in
the CPU test tier).
"""
import numpy as np


def voice_range(total_voices, rank, world):
    """Contiguous shard [lo, hi) of rank; shards differ by at most one voice and cover [0, V)."""
    lo = total_voices * rank // world
    hi = total_voices * (rank + 1) // world
    return lo, hi


def exchange_unique_id(dist, ctx, rank):
    """Rank 0 creates the RCCL unique id; the launcher's process group broadcasts the 128 bytes."""
    box = [ctx.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def reduce_bus_host(dist, bus, root=0):
    """Sum-reduce a host bus [frames][2] float32 onto `root` (returns the reduced array on root,
    the local array elsewhere)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(bus, dtype=np.float32).copy())
    dist.reduce(t, dst=root, op=dist.ReduceOp.SUM)
    return t.numpy()
