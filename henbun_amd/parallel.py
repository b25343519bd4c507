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


class Communicator:
    """This rank's RCCL communicator behind the C ABI (hb_comm_init / hb_allreduce_sum).

    The 128-byte rendezvous token is created on rank 0 and handed round with torch.distributed's object broadcast
    (any backend); the collective itself never goes through torch, so it can be issued on the plan's own HIP
    stream and captured into the step graph.  `Communicator.create()` returns None when RCCL cannot serve this
    process group (not resolvable, or several ranks sharing one device in the one-GPU rehearsal): the caller then
    keeps torch.distributed's all_reduce, eagerly."""

    _cached = {}

    def __init__(self, handle, rank, world_size):
        self.handle, self.rank, self.world_size = handle, rank, world_size

    @classmethod
    def create(cls, device):
        import ctypes
        import os

        import torch.distributed as dist

        from . import _lib

        rank, ws = world()
        key = (rank, ws, str(device))
        if key in cls._cached:
            return cls._cached[key]
        lib = _lib.lib()
        ok = bool(lib.raw("hb_comm_available")()) and not os.environ.get("HENBUN_ONE_DEVICE")
        if dist.is_available() and dist.is_initialized() and ws > 1:
            # every rank must take the same decision
            votes = [None] * ws
            dist.all_gather_object(votes, ok)
            ok = all(votes)
        comm = None
        if ok:
            buf = ctypes.create_string_buffer(128)
            if rank == 0:
                lib.call("hb_comm_unique_id", buf)
            token = [buf.raw]
            if dist.is_available() and dist.is_initialized() and ws > 1:
                dist.broadcast_object_list(token, src=0)
            handle = ctypes.c_void_p(None)
            multi = dist.is_available() and dist.is_initialized() and ws > 1

            def agree(flag):
                if not multi:
                    return bool(flag)
                votes = [None] * ws
                dist.all_gather_object(votes, bool(flag))
                return all(votes)

            # 1. communicator set-up, voted on BEFORE any collective is issued: the test all-reduce below is attempted
            # only when every rank holds a communicator.  (ncclCommInitRank is itself a rendezvous: a rank whose call
            # THROWS while its peers are still inside theirs leaves them waiting for RCCL's own timeout -- that case
            # cannot be turned into a fallback from here; every other failure is.)
            try:
                lib.call("hb_comm_init", ctypes.create_string_buffer(token[0], 128), rank, ws, ctypes.byref(handle))
                made = True
            except Exception:
                made = False
            good = agree(made)
            if good:
                comm = cls(handle, rank, ws)
                if ws > 1:
                    # 2. first use, checked: (rank + 1) summed over the ranks must be R (R + 1) / 2 everywhere -- a
                    # communicator that cannot do this is dropped on EVERY rank and the step keeps torch.distributed's
                    # all_reduce (the "torch-eager" exchange), instead of training on garbage
                    try:
                        import torch

                        from . import hip_ops

                        x = torch.full((256,), float(rank + 1), dtype=torch.float32, device=device)
                        torch.cuda.synchronize(device)
                        hip_ops.allreduce_sum(x, handle)
                        torch.cuda.synchronize(device)
                        fine = bool((x == float(ws * (ws + 1) // 2)).all().item())
                    except Exception:
                        fine = False
                    good = agree(fine)
            if not good:
                comm = None
                if made and handle:
                    # release the native communicator instead of leaking it behind the cached None
                    try:
                        lib.call("hb_comm_destroy", handle)
                    except Exception:
                        pass
        cls._cached[key] = comm
        return comm

    def graph_safe(self, session):
        """True when an RCCL all-reduce captured into a hipGraph replays correctly for this communicator: a
        1024-element buffer of (rank + 1) is reduced by a captured graph, replayed twice, and compared with the
        closed form R(R+1)/2 on every rank; the ranks then agree on the verdict.  Checked once."""
        if getattr(self, "_graph_safe", None) is not None:
            return self._graph_safe
        import torch

        from . import hip_ops

        ok = True
        try:
            want = float(self.world_size * (self.world_size + 1) // 2)
            x = torch.empty(1024, dtype=torch.float32, device=session.device)
            torch.cuda.synchronize()
            with torch.cuda.stream(session.stream):
                x.fill_(float(self.rank + 1))
                hip_ops.allreduce_sum(x, self.handle)          # eager warm-up (connection set-up happens here)
                session.stream.synchronize()
                ok = ok and bool((x == want).all().item())
                g = hip_ops.CapturedGraph()
                g.begin()
                try:
                    hip_ops.allreduce_sum(x, self.handle)
                finally:
                    g.end()
                for _ in range(2):
                    x.fill_(float(self.rank + 1))
                    g.launch()
                    session.stream.synchronize()
                    ok = ok and bool((x == want).all().item())
        except Exception:
            ok = False
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and self.world_size > 1:
            votes = [None] * self.world_size
            dist.all_gather_object(votes, bool(ok))
            ok = all(votes)
        self._graph_safe = bool(ok)
        return self._graph_safe

    def allreduce_sum(self, flat):
        """In place, asynchronous on the current henbun_amd stream (capturable)."""
        from . import hip_ops

        hip_ops.allreduce_sum(flat, self.handle)


def gradient_scale(world_size, dp_reduce):
    """Factor applied to the summed gradient: 'mean' when every rank's objective
    already estimates the full ELBO from its own minibatch (the usual (N/n)*ll - KL
    form), 'sum' when the objective is a plain sum over minibatch rows."""
    if dp_reduce not in ("mean", "sum"):
        raise ValueError("dp_reduce must be 'mean' or 'sum'")
    return 1.0 / world_size if (world_size > 1 and dp_reduce == "mean") else 1.0
