"""Model / AutoOptimize / Optimizer: the driver of the ELBO step.

Same surface as reference Henbun/model.py:13-269:
    class M(hb.model.Model):
        def setUp(self): ...parameters...
        @hb.model.AutoOptimize()
        def ELBO(self): ...
    m.ELBO().compile(optimizer=hb.tf.train.AdamOptimizer(1e-3))
    m.ELBO().run(minibatch_size)       -> float
    m.ELBO().optimize(maxiter, minibatch_size)

`compile` traces the method in tf_mode into a henbun_amd graph, differentiates
it (graph-level autodiff stands in for TF's), and lowers objective + gradients +
the fused flat-buffer Adam update into one hipGraph per minibatch size;
`optimize` replays it (one host call per step).  Under torch.distributed the
flat gradient is all-reduced (RCCL) between backward and Adam.
"""
from __future__ import annotations

from functools import wraps

import numpy as np

from . import graph as G
from ._settings import settings
from .param import Data, MinibatchData, Parameterized, Variable, graph_key
from . import parallel
from .session import Indexer, Session


class AdamOptimizer:
    """tf.train.AdamOptimizer hyper-parameters + slots (TF-1 update rule, see
    include/henbun_hip.h hb_adam_step).  One instance may serve several compiled
    objectives; its slots (m, v, step count) are kept per parameter layout."""

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-08, name="Adam"):
        self.learning_rate, self.beta1, self.beta2, self.epsilon = learning_rate, beta1, beta2, epsilon
        self._slots = {}

    def slots(self, session):
        key = (id(session), session.layout_version)
        s = self._slots.get(key)
        if s is None:
            torch = session.torch
            n = session.theta.numel()
            s = {
                "m": torch.zeros(n, dtype=session.torch_dtype, device=session.device),
                "v": torch.zeros(n, dtype=session.torch_dtype, device=session.device),
                "t": torch.zeros(1, dtype=torch.int64, device=session.device),
                # sticky failure record of hb_adam_step: {step number of the first blocked update, its status word}
                "fail": torch.zeros(2, dtype=torch.int64, device=session.device),
            }
            self._slots[key] = s
        return s


class Model(Parameterized):
    """Root of a parameter tree; owns the device session (reference model.py:13-123)."""

    def __init__(self, name="model", dtype=None, seed=None, **kw):
        Parameterized.__init__(self)
        self._name = name
        self._session = Session(self, dtype=dtype, seed=seed)
        self._index = Indexer()
        self.setUp(**kw)

    @property
    def name(self):
        return self._name

    def setUp(self):
        pass

    def _begin_tf_mode(self):
        self._session.trace_id += 1
        Parameterized._begin_tf_mode(self)

    def initialize(self):
        """Make pending assignments effective on the device (reference model.py:76-82)."""
        self._session.initialize()
        self.finalize()

    def run(self, tensor, feed_dict=None):
        """Evaluate a graph tensor (built in tf_mode) with the current parameters
        (reference model.py:84-96).  MinibatchData is fed whole."""
        self.initialize()
        plan = self._session.make_plan([G.as_tensor(tensor)])
        plan.run()
        plan.check()
        return plan.value(plan.outputs[0])

    def validate(self):
        """reference model.py:98-117."""
        for p in self.get_variables(graph_key.LOCAL):
            if p._tensor is None:
                raise ValueError("local variable " + p.long_name + " is not fed.")
        self._setup_index()

    def _setup_index(self):
        mb = [d for d in self.get_variables(graph_key.DATA) if isinstance(d, MinibatchData)]
        if len(mb) > 1:
            for d in mb:
                if d.data_size != mb[0].data_size:
                    raise ValueError("Minibatch data" + d.long_name + " is not the same size.")
        if len(mb) > 0:
            data_size = mb[0].data_size
            if self._index.data_size is None or self._index.data_size != data_size:
                self._index.setUp(data_size)

    def test_feed_dict(self, minibatch_size=None):
        return self.get_feed_dict(self._index.test_index(minibatch_size))


class AutoOptimize:
    """Decorator: `m.method()` returns the cached Optimizer of that method
    (reference model.py:155-188)."""

    def __call__(self, method):
        @wraps(method)
        def runnable(instance):
            name = "_" + method.__name__ + "_AF_optimizer"
            d = object.__getattribute__(instance, "__dict__")
            if name not in d:
                object.__setattr__(instance, name, Optimizer(instance, method))
            return d[name]

        return runnable


_DEFAULT_ADAM = AdamOptimizer()
STATE_FORMAT_VERSION = 2   # Optimizer.save_state: 2 = layout_names / layout_spans (no pickle), adam_fail, format_version


class Optimizer:
    def __init__(self, model_instance, likelihood_method):
        self.model = model_instance
        self.likelihood_method = likelihood_method
        self.method_op = None
        self.optimize_op = None
        self._optimizer = None
        self._collection = graph_key.VARIABLES
        self._plans = {}
        self.dp_reduce = "mean"
        self.last_plan = None

    # ------------------------------------------------------------------ tracing
    def _trace(self, minibatch):
        sess = self.model._session
        sess.trace_minibatch = minibatch
        try:
            with self.model.tf_mode():
                obj = self.likelihood_method(self.model)
        finally:
            sess.trace_minibatch = None
        if not isinstance(obj, G.Tensor):
            obj = G.constant(np.asarray(obj, dtype=np.float64))
        if obj.size != 1:
            raise ValueError("the objective must be a scalar, got shape %s" % (obj.shape,))
        return G.reshape(obj, [])

    def compile(self, optimizer=_DEFAULT_ADAM, collection=graph_key.VARIABLES, global_step=None, dp_reduce="mean"):
        """Trace + validate; device plans are built per minibatch size on first use
        (reference model.py:206-230).  `dp_reduce`: how per-rank gradients combine
        under data parallelism ('mean': each rank's objective already estimates the
        full ELBO from its own minibatch; 'sum': the objective is a plain sum over rows)."""
        print("compiling...")
        self._optimizer = optimizer
        self._collection = collection
        self.dp_reduce = dp_reduce
        self._plans = {}
        self.model.initialize()
        mbs = [d for d in self.model.get_variables(graph_key.DATA) if isinstance(d, MinibatchData)]
        probe = None
        if mbs:
            self.model._setup_index()
            probe = min(2, self.model._index.train_size)
        self.model._session.probing = True
        try:
            self.method_op = self._trace(probe)
        finally:
            self.model._session.probing = False
        self.model.validate()
        self.optimize_op = True
        print("finished.")

    # ------------------------------------------------------------------ plans
    def _settings_key(self):
        n = settings.numerics
        return (n.jitter_level, n.clip_by_value, n.clip_value_min, n.clip_value_max, str(getattr(n, "kl_form", "mc")), str(getattr(n, "contraction", "native")),
                str(settings.runtime.index_source), bool(settings.runtime.fuse_elementwise),
                bool(getattr(settings.runtime, "force_dp", False)), str(getattr(settings.runtime, "dp_exchange", "auto")),
                str(getattr(settings.runtime, "ewise", "jit")), bool(getattr(settings.runtime, "side_jobs", True)),
                bool(getattr(settings.runtime, "serial_chains", True)))

    @staticmethod
    def _dp_active(sess):
        """Data-parallel step: more than one rank, or `settings.runtime.force_dp` (runs the exchange step with a
        one-rank communicator: how the multi-rank code path is exercised on a one-GPU box)."""
        return sess.world_size > 1 or bool(getattr(settings.runtime, "force_dp", False))

    def dp_objective(self):
        """Mean over ranks of the objective of the last optimisation step (it rides behind the gradient in the
        all-reduce); None outside data-parallel runs."""
        plan = self.last_plan
        if plan is None or getattr(plan, "dp_mode", "none") == "none":
            return None
        plan.stream.synchronize() if plan.stream is not None else plan.torch.cuda.synchronize()
        sess = self.model._session
        return float(plan.gflat[sess.theta.numel()].item()) / max(sess.world_size, 1)

    def _get_plan(self, kind, minibatch, training=True):
        sess = self.model._session
        self.model.initialize()
        key = (kind, minibatch, training, sess.layout_version, self._settings_key())
        plan = self._plans.get(key)
        if plan is not None:
            return plan
        if minibatch is not None:
            self.model._setup_index()
        obj = self._trace(minibatch)
        if kind == "run":
            plan = sess.make_plan([obj], minibatch=minibatch, training=training)
            plan.objective = obj
        else:
            leaves = []
            for t in self.model.get_tf_variables(self._collection):
                if t is not None and t.node.op == "leaf:param" and t not in leaves:
                    leaves.append(t)
            # the flat buffer receives d objective / d theta; the optimiser minimises -objective (reference
            # model.py:206-220) by reading it with a negative gscale: negation is exact, so the update is
            # bit-identical to differentiating -objective, without a sign flip over every gradient
            grads = [g if g is None or not g.node.op.startswith("leaf:") else G.unary("COPY", g)
                     for g in G.gradients(obj, leaves)]
            torch = sess.torch
            # [gradient of every global leaf | objective value | failure flag]: the two tail words travel with the
            # gradient through the data-parallel all-reduce (hb_dp_pack)
            P = sess.theta.numel()
            gflat = torch.zeros(P + 2, dtype=sess.torch_dtype, device=sess.device)
            binds, segs = [], []
            for t, g in zip(leaves, grads):
                o, s = sess._offsets[id(t.node.attrs["var"])]
                segs.append((o, s))
                if g is not None:
                    binds.append((g, gflat[o:o + s]))
            outs = [obj] + [g for g in grads if g is not None]
            plan = sess.make_plan(outs, binds=binds, minibatch=minibatch, training=True)
            plan.objective = obj
            plan.gflat = gflat
            # merge optimised leaves into contiguous segments of the flat buffer
            segs.sort()
            merged = []
            for o, s in segs:
                if merged and merged[-1][0] + merged[-1][1] == o:
                    merged[-1] = (merged[-1][0], merged[-1][1] + s)
                else:
                    merged.append((o, s))
            plan.segments = merged
            opt = self._optimizer
            slots = opt.slots(sess)
            H = sess.H
            theta = sess.theta
            gscale = -parallel.gradient_scale(sess.world_size, self.dp_reduce)

            info = plan.info_words()
            plan.fail = slots["fail"]

            def adam():
                # one fused launch per contiguous segment; they share the step counter, which the last one advances.
                # A factorisation that failed in this step turns the update into a no-op (hb_adam_step): the
                # parameters stay at the last good step, as when tf.cholesky raises before apply_gradients.
                for i, (o, s) in enumerate(merged):
                    H.adam_step(theta[o:o + s], gflat[o:o + s], slots["m"][o:o + s], slots["v"][o:o + s], slots["t"],
                                lr=opt.learning_rate, b1=opt.beta1, b2=opt.beta2, eps=opt.epsilon, gscale=gscale,
                                tick=(i == len(merged) - 1), info=info, dpflag=plan.dpflag, fail=slots["fail"])

            plan.dpflag = None
            plan.adam = adam
            plan.chain_kind[id(adam)] = "full"     # hb_adam_step records itself into a serial chain (small parameter sets)
            plan.eager_tail = []
            plan.dp_mode = "none"
            if not self._dp_active(sess):
                plan.steps.append(adam)
                plan.side_effect_steps.add(adam)
            else:
                # data parallel: [forward + backward] -> pack -> ONE all-reduce of the whole flat buffer -> Adam.
                # With RCCL behind the C ABI the exchange is issued on the plan's stream and is part of the
                # captured graph; otherwise (no RCCL for this group) it runs eagerly through torch.distributed.
                tail, objbuf = gflat[P:P + 2], plan.buf(obj)
                plan.dpflag = gflat[P + 1:P + 2]

                def pack():
                    H.dp_pack(tail, objbuf, info)

                comm = parallel.Communicator.create(sess.device)
                exch = lambda: comm.allreduce_sum(gflat)
                # Where the exchange runs (settings.runtime.dp_exchange):
                #   eager  after the captured forward+backward graph, on the plan's stream: pack, ONE RCCL all-reduce
                #          issued through the C ABI, Adam -- three extra host calls per step, no collective inside a graph;
                #   graph  the same three steps recorded into the step graph (one host call per step), after
                #          Communicator.graph_safe has replayed a captured all-reduce against its closed form;
                #   auto   (default) graph for a one-rank communicator (settings.runtime.force_dp: how the tail is
                #          exercised on a one-GPU box), eager for world_size > 1: a captured multi-rank RCCL
                #          collective has not yet been run on hardware by this project (no multi-GPU box in the
                #          build loop), and an all-reduce that hangs inside a graph takes the whole job with it.
                mode = str(getattr(settings.runtime, "dp_exchange", "auto"))
                if mode not in ("auto", "graph", "eager"):
                    raise ValueError("settings.runtime.dp_exchange must be auto, graph or eager")
                want_graph = mode == "graph" or (mode == "auto" and sess.world_size == 1)
                if comm is not None and want_graph and comm.graph_safe(sess):
                    plan.steps += [pack, exch, adam]
                    # none of the three runs in the capture warm-up pass: a re-capture on one rank only (per-call
                    # indices, injected noise) must not issue a collective the other ranks do not take part in
                    plan.side_effect_steps.update([pack, exch, adam])
                    plan.dp_mode = "rccl-in-graph"
                elif comm is not None:
                    plan.eager_tail = [pack, exch, adam]
                    plan.dp_mode = "rccl-eager"
                else:
                    plan.eager_tail = [pack, lambda: parallel.allreduce_gradient(gflat), adam]
                    plan.dp_mode = "torch-eager"
                plan.dp_comm = comm
        plan.sess = sess
        if (kind != "run" and not plan.eager_tail and plan.param_only_steps
                and bool(getattr(settings.runtime, "trailing_transforms", True))):
            # Trailing transforms: the elementwise programs that only read parameters (softplus of the raw hyper-parameters
            # and the like: a ~4.4 us launch of its own at the head of every step) run at the END of the step instead, right
            # behind the Adam update and inside its serial chain -- they produce the values of the NEXT replay.  The
            # session's parameter version says when that is not enough (first replay, a host-side assignment, another
            # plan's update, restore): _run_steps then runs them once, eagerly, before launching.
            for st in plan.param_only_steps:
                plan.steps.remove(st)
                plan.steps.append(st)
            plan.prologue = list(plan.param_only_steps)
            with plan._on_stream():
                for st in plan.prologue:
                    st()
            plan.trail_version = sess.param_version
        if settings.runtime.graph_capture:
            plan.capture()
        self._plans[key] = plan
        return plan

    # ------------------------------------------------------------------ reference API
    def feed_dict(self, minibatch_size=None, training=True):
        """Host-side view of what a step would be fed (reference model.py:232-243)."""
        if minibatch_size is None:
            return self.model.get_feed_dict(None)
        if training:
            return self.model.get_feed_dict(self.model._index.train_index(minibatch_size))
        return self.model.get_feed_dict(self.model._index.test_index(minibatch_size))

    def _ensure_compiled(self):
        if self._optimizer is None:
            self.compile()

    def run(self, minibatch_size=None, training=True, indices=None):
        """Objective value at the current parameters (reference model.py:245-253).
        `indices` optionally fixes the minibatch rows (indices into the data array)."""
        self._ensure_compiled()
        plan = self._get_plan("run", minibatch_size, training)
        self._apply_indices(plan, indices)
        plan.run()
        plan.check()
        self.last_plan = plan
        return float(plan.value(plan.objective))

    @staticmethod
    def _apply_indices(plan, indices):
        """`indices` fixes the rows of THIS call only; without it every call draws afresh
        (reference model.py:232-267)."""
        if indices is not None:
            plan.set_indices(indices)
        elif plan.index_buffer is not None:
            plan.clear_indices()

    def optimize(self, maxiter=1, minibatch_size=None, indices=None):
        """`maxiter` Adam steps (reference model.py:255-269)."""
        self._ensure_compiled()
        plan = self._get_plan("opt", minibatch_size)
        self._apply_indices(plan, indices)
        self._run_steps(plan, int(maxiter))
        self.last_plan = plan
        self._check_step_failure(plan)
        plan.check()

    @staticmethod
    def _run_steps(plan, k):
        """`k` asynchronous optimisation steps of a compiled 'opt' plan (one graph launch each; plus the eager
        exchange + update when the data-parallel tail could not be captured)."""
        sess = getattr(plan, "sess", None)
        if plan.prologue and plan.trail_version != sess.param_version:
            # the values the trailing transforms left behind are not those of the current parameters
            with plan._on_stream():
                for st in plan.prologue:
                    st()
        if sess is not None and getattr(plan, "adam", None) is not None:
            sess.param_version += k
            plan.trail_version = sess.param_version
        if not plan.eager_tail:
            for _ in range(k):
                plan.run()
        else:
            for _ in range(k):
                plan.run()
                with plan._on_stream():
                    for s in plan.eager_tail:
                        s()

    def _check_step_failure(self, plan):
        """Raise CholeskyError if an update was blocked by a failed factorisation.  The parameters, Adam slots and
        step count are those of the last good step; the record is cleared so that the caller may continue.
        NOT rewound: the steps of the same `optimize` call that were already queued behind the failing one still
        ran their draws (their updates were blocked too: `fail` is sticky on the device), so the minibatch-index and
        noise streams have advanced by the remaining iterations of that call -- unlike the reference, where
        tf.cholesky raises inside session.run and nothing further is drawn (model.py:265-266).  A caller that needs
        the streams at the failing step restores them from a save_state checkpoint."""
        plan.stream.synchronize() if plan.stream is not None else plan.torch.cuda.synchronize()
        step, what = (int(x) for x in plan.fail.cpu().numpy())
        if step != 0:
            plan.fail.zero_()
            plan.torch.cuda.synchronize()
            why = ("a factorisation on another rank failed" if what < 0 else
                   "leading minor %d is not positive definite" % what)
            raise G.CholeskyError("optimize: Adam step %d was not applied: %s; the parameters were left at the "
                                  "last good step (the RNG streams have advanced past it)" % (step, why))

    # ------------------------------------------------------------------ exact resume
    def save_state(self, path):
        """Everything a bit-exact resume of `optimize` needs, in one .npz: the raw parameters, the Adam slots and
        step count, the three device RNG streams of this rank and the Indexer's train/test split.  (The reference's
        `save` keeps the parameters only, param.py:562-603; with data parallelism every rank saves its own file --
        the parameters and the `global` stream are identical across ranks, `local` / `index` are not.)"""
        self._ensure_compiled()
        self.model.initialize()
        sess = self.model._session
        sess.torch.cuda.synchronize()
        slots = self._optimizer.slots(sess)
        out = {
            "format_version": np.array([STATE_FORMAT_VERSION], dtype=np.int64),
            "theta": sess.theta.cpu().numpy(),
            "adam_m": slots["m"].cpu().numpy(), "adam_v": slots["v"].cpu().numpy(), "adam_t": slots["t"].cpu().numpy(),
            "adam_fail": slots["fail"].cpu().numpy(),     # sticky record of a blocked update (hb_adam_step)
        }
        layout = self._layout_rows()
        out["layout_names"] = np.array([r[0] for r in layout], dtype=np.str_)
        out["layout_spans"] = np.array([[r[1], r[2]] for r in layout], dtype=np.int64).reshape(-1, 2)
        for k, r in sess.rngs.items():
            out["rng_" + k] = r.state.cpu().numpy()
        idx = self.model._index
        if idx.data_size is not None:
            out["index_train"], out["index_test"] = np.asarray(idx._train_index), np.asarray(idx._test_index)
        p = path if path.endswith(".npz") else path + ".npz"
        np.savez(p, **out)
        return p

    def _layout_rows(self):
        sess = self.model._session
        return sorted((v.long_name, int(o), int(s)) for v in self.model.get_variables()
                      for o, s in [sess._offsets.get(id(v), (-1, -1))] if o >= 0)

    def restore_state(self, path):
        """Inverse of save_state.  All device buffers are overwritten IN PLACE, so plans captured before the call
        keep replaying on the restored state."""
        self._ensure_compiled()
        self.model.initialize()
        sess = self.model._session
        torch = sess.torch
        p = path if path.endswith(".npz") else path + ".npz"
        with np.load(p, allow_pickle=False) as f:
            if "format_version" not in f.files:
                if "layout" in f.files or "layout_names" in f.files:
                    raise ValueError("checkpoint %s was written by an earlier revision of save_state (no format_version; "
                                     "the pickled 'layout' array was replaced by layout_names / layout_spans): re-save it "
                                     "with this revision" % p)
                raise ValueError("%s is not a save_state checkpoint" % p)
            ver = int(f["format_version"][0])
            if ver != STATE_FORMAT_VERSION:
                raise ValueError("checkpoint %s has format_version %d, this revision reads %d" % (p, ver, STATE_FORMAT_VERSION))
            saved = [(str(a), int(b), int(c)) for a, (b, c) in zip(f["layout_names"].tolist(), f["layout_spans"].tolist())]
            if saved != self._layout_rows():
                raise ValueError("checkpoint %s was written for a different parameter layout" % p)
            slots = self._optimizer.slots(sess)
            torch.cuda.synchronize()
            sess.theta.copy_(torch.as_tensor(f["theta"]).to(sess.theta.dtype))
            sess.param_version += 1
            slots["m"].copy_(torch.as_tensor(f["adam_m"]).to(slots["m"].dtype))
            slots["v"].copy_(torch.as_tensor(f["adam_v"]).to(slots["v"].dtype))
            slots["t"].copy_(torch.as_tensor(f["adam_t"]))
            slots["fail"].copy_(torch.as_tensor(f["adam_fail"]))
            for k, r in sess.rngs.items():
                r.state.copy_(torch.as_tensor(f["rng_" + k]))
            if "index_train" in f.files:
                idx = self.model._index
                idx._train_index, idx._test_index = f["index_train"], f["index_test"]
                idx.data_size = len(idx._train_index) + len(idx._test_index)
                idx.train_size, idx.test_size = len(idx._train_index), len(idx._test_index)
                for key, arr in (("train", idx._train_index), ("test", idx._test_index)):
                    dev = idx._dev.get(key)
                    if dev is not None:
                        dev.copy_(torch.as_tensor(np.ascontiguousarray(arr, dtype=np.int64)))
            torch.cuda.synchronize()

    def gradients(self, minibatch_size=None, indices=None):
        """{long_name: d objective / d raw parameter} at the current parameters
        (not in the reference; used by the parity tests).  Runs forward+backward
        without the Adam update."""
        self._ensure_compiled()
        sess = self.model._session
        self.model.initialize()
        obj = self._trace(minibatch_size)
        leaves, names = [], []
        for v in self.model.get_variables(self._collection):
            if v.is_parameter and v._leaf not in leaves:
                leaves.append(v._leaf)
                names.append(v.long_name)
        grads = G.gradients(obj, leaves)
        outs = [obj] + [g for g in grads if g is not None]
        plan = sess.make_plan(outs, minibatch=minibatch_size)
        self.last_plan = plan
        self._apply_indices(plan, indices)
        plan.run()
        plan.check()
        res = {}
        for nme, t, g in zip(names, leaves, grads):
            res[nme] = np.zeros(t.shape) if g is None else plan.value(g).astype(np.float64)
        return float(plan.value(obj)), res
