"""Data-parallel helpers: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests).

The reference has no distributed code at all; the scheme here is
new: every rank holds a shard of the data, draws its own minibatch, runs the
replicated O(M^3) part (Kmm, Cholesky, global KL) locally, and the ranks
exchange ONE contiguous flat gradient per Adam step.  Global variational noise
must agree across ranks (stream id 0 everywhere); per-datapoint noise and
minibatch indices are rank-distinct.
"""
from __future__ import annotations

import numpy as np


def world():
    """(rank, world_size) of the initialised default group, (0, 1) otherwise."""
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def rng_stream_ids(rank):
    """Stream ids of the three device RNG streams of a rank."""
    return {"global": 0, "local": 1 + 2 * int(rank), "index": 2 + 2 * int(rank)}


def shard_rows(n_rows, rank, world_size):
    """[begin, end) of the contiguous block of rows owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_rows), int(world_size))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def allreduce_gradient(flat, segments=None):
    """Sum the flat gradient buffer over ranks, in place (one collective per
    contiguous segment; the whole buffer when `segments` is None).  The caller
    folds the 1/world_size of a mean into the Adam kernel's `gscale`."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return flat
    if segments is None:
        dist.all_reduce(flat)
    else:
        for o, s in segments:
            dist.all_reduce(flat[o:o + s])
    return flat


def gradient_scale(world_size, dp_reduce):
    """Factor applied to the summed gradient: 'mean' when every rank's objective
    already estimates the full ELBO from its own minibatch (the usual (N/n)*ll - KL
    form), 'sum' when the objective is a plain sum over minibatch rows."""
    if dp_reduce not in ("mean", "sum"):
        raise ValueError("dp_reduce must be 'mean' or 'sum'")
    return 1.0 / world_size if (world_size > 1 and dp_reduce == "mean") else 1.0
