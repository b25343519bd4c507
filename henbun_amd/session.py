"""Device-side state of one Model: flat parameter buffer, resident datasets,
RNG streams, minibatch index feeder, data-parallel group.

Plays the part of `tf.Session()` in the reference (Henbun/model.py:57): the
thing that owns parameter storage and executes compiled objectives.  There is
no CPU path: constructing the device state without the HIP backend or without
a GPU raises.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np

from . import graph as G
from ._settings import np_float_type, settings


class Indexer:
    """Train/test split + with-replacement minibatch indices (reference
    Henbun/model.py:126-153): shuffle once, hold out floor(test_frac*N)."""

    def __init__(self):
        self.data_size = None
        self.test_frac = 0.1

    def setUp(self, data_size):
        self.data_size = data_size
        self.test_size = int(np.floor(self.data_size * self.test_frac))
        self.train_size = data_size - self.test_size
        index = np.arange(self.data_size)
        np.random.shuffle(index)
        self._train_index = index[: self.train_size]
        self._test_index = index[self.train_size:]
        self._dev = {}

    def train_index(self, minibatch_size):
        return self._train_index[np.random.randint(0, self.train_size, minibatch_size)]

    def test_index(self, minibatch_size):
        return self._test_index[np.random.randint(0, self.test_size, minibatch_size)]


class Session:
    def __init__(self, model, dtype=None, seed=None, device=None):
        self.model = model
        self.np_dtype = np_float_type(dtype)
        self.seed = int(settings.runtime.seed if seed is None else seed)
        self._device_arg = device
        self._ready = False
        self.trace_id = 0
        self.trace_minibatch = None
        self.layout_version = 0
        self._layout: List = []          # [(Variable, offset, size)]
        self._offsets: Dict[int, tuple] = {}
        self.theta = None
        self._data_bufs: Dict[int, object] = {}
        self.rank, self.world_size = 0, 1
        self.injected_indices = None
        self.param_version = 0           # bumped by every write to the flat parameter buffer (host writes, optimiser steps)

    # ------------------------------------------------------------------ device bring-up
    def _ensure_device(self):
        if self._ready:
            return
        from . import _lib

        _lib.lib()  # raises ImportError when libhenbun_hip.so is missing: no fallback
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("henbun_amd computes only through its HIP backend: no GPU is visible "
                               "(there is no CPU fallback)")
        from . import hip_ops

        self.torch, self.H = torch, hip_ops
        if self._device_arg is not None:
            self.device = torch.device(self._device_arg)
        else:
            # one process per GPU: LOCAL_RANK names the card (HENBUN_ONE_DEVICE=1: rehearsal of the multi-rank path
            # with every rank on the current device of a one-GPU box)
            one = os.environ.get("HENBUN_ONE_DEVICE")
            self.device = torch.device("cuda", torch.cuda.current_device() if one else
                                       int(os.environ.get("LOCAL_RANK", torch.cuda.current_device())))
        torch.cuda.set_device(self.device)
        self.torch_dtype = torch.float64 if self.np_dtype == np.float64 else torch.float32
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.rank, self.world_size = torch.distributed.get_rank(), torch.distributed.get_world_size()
        # RNG streams: 'global' identical on every rank (global variational noise must agree),
        # 'local' / 'index' rank-distinct (per-datapoint noise, minibatch indices).
        from . import parallel

        ids = parallel.rng_stream_ids(self.rank)
        self.rngs = {k: hip_ops.Rng(self.seed, stream_id=v, device=self.device) for k, v in ids.items()}
        self.stream = torch.cuda.Stream(device=self.device)
        self._ready = True

    def reseed(self, seed):
        self.seed = int(seed)
        if self._ready:
            from . import parallel

            for k, v in parallel.rng_stream_ids(self.rank).items():
                self.rngs[k].reseed(self.seed, v)

    # ------------------------------------------------------------------ flat parameter store
    def invalidate(self):
        """The tree changed (a child was attached): re-derive the layout lazily."""
        self._layout_dirty = True

    def _global_variables(self):
        seen, out = set(), []
        for v in self.model.get_variables():
            if v.is_parameter and id(v) not in seen:
                seen.add(id(v))
                out.append(v)
        return out

    def ensure_layout(self):
        self._ensure_device()
        vs = self._global_variables()
        if [id(v) for v in vs] == [id(v) for v, _, _ in self._layout] and self.theta is not None:
            return
        old = {id(v): self.read_raw(v) for v, _, _ in self._layout} if self.theta is not None else {}
        off, layout = 0, []
        for v in vs:
            size = int(np.prod(v._full_shape)) if v._full_shape else 1
            layout.append((v, off, size))
            off += size
        host = np.zeros(max(off, 1), dtype=self.np_dtype)
        for v, o, s in layout:
            if id(v) in old and not v._assigned:
                host[o:o + s] = old[id(v)].reshape(-1)
            else:
                host[o:o + s] = np.asarray(v._host_raw, dtype=self.np_dtype).reshape(-1)
                v._assigned = False
        self.theta = self.torch.as_tensor(host).to(self.device)
        if self.world_size > 1:
            # data parallel: every rank must start from rank 0's parameters
            self.torch.distributed.broadcast(self.theta, 0)
        self._layout = layout
        self._offsets = {id(v): (o, s) for v, o, s in layout}
        self.layout_version += 1
        self.param_version += 1

    def initialize(self):
        """Upload pending (assigned) values; afterwards nothing is pending."""
        self.ensure_layout()
        dirty = False
        for v, o, s in self._layout:
            if v._assigned:
                self.write_raw(v, v._host_raw)
                v.finalize()
                dirty = True
        if dirty and self.world_size > 1:
            self.torch.distributed.broadcast(self.theta, 0)

    def param_view(self, v):
        o, s = self._offsets[id(v)]
        return self.theta[o:o + s].view(tuple(v._full_shape))

    def read_raw(self, v):
        o, s = self._offsets[id(v)]
        self.torch.cuda.synchronize()
        return self.theta[o:o + s].detach().cpu().numpy().astype(np.float64).reshape(v._full_shape)

    def write_raw(self, v, raw):
        o, s = self._offsets[id(v)]
        arr = np.array(np.broadcast_to(np.asarray(raw, dtype=self.np_dtype), v._full_shape))
        self.torch.cuda.synchronize()
        self.theta[o:o + s].copy_(self.torch.as_tensor(arr.reshape(-1)).to(self.device))
        self.torch.cuda.synchronize()
        self.param_version += 1

    def read_value(self, v):
        self.ensure_layout()
        if v._assigned:
            self.write_raw(v, v._host_raw)
            v.finalize()
        return np.asarray(v.transform.forward(self.read_raw(v)))

    # ------------------------------------------------------------------ data
    def data_buffer(self, var):
        self._ensure_device()
        b = self._data_bufs.get(id(var))
        if b is None:
            arr = np.ascontiguousarray(np.asarray(var.data, dtype=self.np_dtype))
            b = self.torch.as_tensor(arr).to(self.device)
            self._data_bufs[id(var)] = b
        return b

    def data_changed(self, var):
        b = self._data_bufs.get(id(var))
        if b is not None:
            arr = np.ascontiguousarray(np.asarray(var.data, dtype=self.np_dtype))
            b.copy_(self.torch.as_tensor(arr).to(self.device))

    # ------------------------------------------------------------------ plans
    def make_plan(self, outputs, binds=None, minibatch=None, training=True):
        """Lower `outputs` to a launch plan.  minibatch = None (all rows, in order)
        or the number of rows drawn per run from the train (or test) split."""
        self._ensure_device()
        self.ensure_layout()
        sess = self
        feeder = {"idx": None, "emitted": False}

        def resolver_with_plan(plan):
            def resolve(t):
                kind = t.node.op[5:]
                var = t.node.attrs["var"]
                if kind == "param":
                    if id(var) not in sess._offsets:
                        raise RuntimeError("parameter %s is not part of model %s" % (var.long_name, sess.model.name))
                    return sess.param_view(var)
                if kind == "data":
                    return sess.data_buffer(var)
                if kind == "minibatch":
                    full = sess.data_buffer(var)
                    n = t.shape[0]
                    if minibatch is None:
                        if n != full.shape[0]:
                            raise ValueError("traced with minibatch size %d but run with all %d rows" % (n, full.shape[0]))
                        return full
                    out = sess.torch.empty(t.shape, dtype=sess.torch_dtype, device=sess.device)
                    index = sess.model._index
                    if index.data_size is None or index.data_size != full.shape[0]:
                        sess.model._setup_index()
                    if feeder["idx"] is None:
                        feeder["idx"] = sess.torch.zeros(n, dtype=sess.torch.int64, device=sess.device)
                        feeder["err"] = sess.torch.zeros(1, dtype=sess.torch.int32, device=sess.device)
                        key = "train" if training else "test"
                        perm = index._dev.get(key)
                        if perm is None:
                            src = index._train_index if training else index._test_index
                            perm = sess.torch.as_tensor(np.ascontiguousarray(src, dtype=np.int64)).to(sess.device)
                            index._dev[key] = perm
                        feeder["perm"] = perm
                        hi = index.train_size if training else index.test_size
                        idx, rng = feeder["idx"], sess.rngs["index"]
                        plan.index_buffer = idx
                        plan.index_range = hi
                        if str(settings.runtime.index_source) == "device":
                            feeder["draw"] = (rng, hi)

                            def draw():
                                # once the multi-array gather exists it draws the indices itself (same values, same
                                # RNG state: one launch instead of two)
                                feeder["drawn"] = False
                                if not plan.indices_injected and not (feeder.get("mg") and n <= rng.nlanes):
                                    rng.randint(n, 0, hi, out=idx)
                                    feeder["drawn"] = True
                            plan.steps.append(draw)
                    idx, perm, err = feeder["idx"], feeder["perm"], feeder["err"]
                    H = sess.H
                    # every minibatch array takes the same rows: ONE gather launch for all of them, emitted with the
                    # first array (the list is complete by the time the plan runs)
                    if "arrays" not in feeder:
                        feeder["arrays"] = []
                        feeder["leaves"] = []
                        # the fused draw + gather launch may ride on a later host launch (side jobs, graph.Plan)
                        feeder["cell"] = plan.side_candidate(feeder["leaves"])

                        def gather():
                            mg = feeder.get("mg")
                            if mg is None:
                                srcs = [a for a, _ in feeder["arrays"]]
                                outs = [o for _, o in feeder["arrays"]]
                                if len(srcs) <= 8 and len({tuple(a.shape[:1]) for a in srcs}) == 1:
                                    mg = feeder["mg"] = H.MultiGather(srcs, outs, idx, perm, err)
                                else:
                                    mg = feeder["mg"] = False
                            fused = (mg and feeder.get("draw") and not feeder.get("drawn") and not plan.indices_injected
                                     and n <= feeder["draw"][0].nlanes)
                            if fused:
                                mg.launch_draw(feeder["draw"][0], 0, feeder["draw"][1], use_perm=not plan.indices_raw,
                                               defer=feeder["cell"]["defer"])
                            elif mg:
                                mg.launch(use_perm=not plan.indices_raw)
                            else:
                                for a, o in feeder["arrays"]:
                                    H.gather_rows(a, idx, None if plan.indices_raw else perm, out=o, err=err)

                        plan.steps.append(gather)
                    feeder["arrays"].append((full, out))
                    feeder["leaves"].append(t)
                    plan.gather_err = err
                    return out
                raise RuntimeError("unknown leaf kind " + kind)

            return resolve

        plan = _SessionPlan.__new__(_SessionPlan)
        plan.index_buffer = None
        plan.index_range = None
        plan.indices_injected = False
        plan.indices_raw = False
        plan.gather_err = None
        plan.host_indices = str(settings.runtime.index_source) != "device"
        plan.session = self
        plan.training = training
        self.torch.cuda.synchronize()  # uploads above ran on the default stream
        G.Plan.__init__(plan, outputs, self.torch_dtype, self.device, resolver_with_plan(plan), self.rngs, binds=binds,
                        stream=self.stream)
        self.torch.cuda.synchronize()
        return plan


class _SessionPlan(G.Plan):
    """Plan + minibatch index handling."""

    def set_indices(self, idx, raw=True):
        """Use these row indices for the next runs (raw=True: indices into the
        full data array; raw=False: positions inside the train/test split)."""
        if self.index_buffer is None:
            raise ValueError("this plan draws no minibatch")
        idx = np.asarray(idx, dtype=np.int64).reshape(-1)
        if idx.shape[0] != self.index_buffer.shape[0]:
            raise ValueError("expected %d indices, got %d" % (self.index_buffer.shape[0], idx.shape[0]))
        self.torch.cuda.synchronize()
        self.index_buffer.copy_(self.torch.as_tensor(idx).to(self.device))
        self.torch.cuda.synchronize()
        self.indices_injected, self.indices_raw = True, raw
        self._graph = self._graphs.get(self._state_key())

    def clear_indices(self):
        """Back to drawing fresh minibatch indices on every run (reference model.py:232-267 draws per call):
        injection lasts for the call that asked for it."""
        if self.indices_injected:
            self.indices_injected, self.indices_raw = False, False
            self._graph = self._graphs.get(self._state_key())

    def _state_key(self):
        return (G.Plan._state_key(self), self.indices_injected, self.indices_raw)

    def check(self):
        """Cholesky status (G.Plan.check) and, when the row indices came from the caller, their range: a row index
        outside the data set makes the gather write zeros and raise its flag (device-drawn indices cannot be)."""
        G.Plan.check(self)
        err = getattr(self, "gather_err", None)
        if err is not None and (self.indices_injected or self.host_indices):
            if int(err.cpu().item()) != 0:
                err.zero_()
                raise IndexError("minibatch indices outside the data set (rows were zero-filled)")

    def run(self):
        if self.host_indices and self.index_buffer is not None and not self.indices_injected:
            index = self.session.model._index
            n = self.index_buffer.shape[0]
            pos = np.random.randint(0, self.index_range, n)
            with self._on_stream():
                self.index_buffer.copy_(self.torch.as_tensor(pos).to(self.device), non_blocking=False)
        G.Plan.run(self)
