"""Multi-GPU plumbing: reads shard embarrassingly (each read's result depends
only on the read, the immutable index and AlignOpts; reference
src/aligner.rs:123), the index is replicated per GPU, and the only exchange is
one all-reduce of the counter vector (RCCL when the backend is "nccl")."""
import numpy as np

from .capi import COUNTER_NAMES, N_COUNTERS


def shard_bounds(n_items, rank, world):
    """Contiguous shard [begin, end) of rank: concatenating the shards in rank
    order restores the input order (the reference writes in input order,
    src/aligner.rs:54-115)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    return (n_items * rank) // world, (n_items * (rank + 1)) // world


def shard_reads(bases, offsets, rank, world):
    """The slice of a packed read set (bases, offsets[n+1]) owned by rank."""
    n = len(offsets) - 1
    b, e = shard_bounds(n, rank, world)
    off = offsets[b : e + 1].astype("<u8")
    return bases[int(off[0]) : int(off[-1])], (off - off[0]).astype("<u8")


def allreduce_counters(counters, dist=None, device=None):
    """Sum the THM_N_COUNTERS vector over all ranks (one collective)."""
    import torch

    c = np.ascontiguousarray(counters, dtype=np.uint64)
    if c.shape != (N_COUNTERS,):
        raise ValueError("expected %d counters" % N_COUNTERS)
    t = torch.from_numpy(c.astype(np.int64))
    if device is not None:
        t = t.to(device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(np.uint64)


def counters_dict(counters):
    return dict(zip(COUNTER_NAMES, [int(v) for v in counters]))
