"""Host logic that needs no GPU: the C-ABI library loads and exports every
declared symbol, the parameter tree / tf_mode / feed semantics mirror the
reference, settings push/pop, Indexer, data-parallel helpers (gloo, 2 ranks)."""
import os
import json
import re
import subprocess
import sys

import numpy as np
import pytest

import henbun_amd as hb
from henbun_amd import graph as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tf = hb.tf


# ---------------------------------------------------------------- C ABI
def test_library_exports_every_declared_symbol():
    from henbun_amd import _lib

    lib = _lib.lib()  # loads without a GPU; raises if the .so is missing
    assert lib.raw("hb_version")() == 2      # HB_ABI_VERSION (include/henbun_hip.h: history of the bumps)
    header = open(os.path.join(ROOT, "include", "henbun_hip.h")).read()
    declared = set(re.findall(r"\b(hb_[A-Za-z0-9_]+)\s*\(", header))
    bound = set(_lib.declared_symbols())
    assert declared == bound, (declared - bound, bound - declared)
    import ctypes

    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(dll, name), name


def test_bad_arguments_are_reported_not_crashed():
    """Argument validation happens before any launch, so it is testable without a GPU."""
    from henbun_amd import _lib

    lib = _lib.lib()
    rc = lib.raw("hb_reduce_f32")(0, None, None, -1, 1, 1, None, 0, None)
    assert rc < 0 and "negative" in lib.last_error()
    rc = lib.raw("hb_gram_fwd_f64")(7, None, 0, None, 0, None, 0, 1, None, 1, 1, 1, 1, 0.0, None)
    assert rc < 0 and "kind" in lib.last_error()
    with pytest.raises(_lib.HipBackendError):
        lib.call("hb_rng_randint", None, 0, None, 1, 0, 0, None)


def test_elementwise_program_source_generates_and_compiles_for_gfx950():
    """hb_ewise_jit_build in its dry-run form (no handle requested: no device needed): the generated kernel text for a
    program with broadcast inputs, a 4-input gradient op and sum-reduced outputs compiles with hiprtc for gfx950."""
    import ctypes
    from ctypes import c_double, c_int, c_long, c_void_p

    from henbun_amd import _lib, hip_ops as H

    L = _lib.lib()
    if not L.raw("hb_ewise_jit_available")():
        pytest.skip("hiprtc cannot be loaded in this process")
    E = H.EW
    code = [[E["SOFTPLUS"], 2, 0, 0, 0], [E["AFFINE"], 3, 2, 0, 0], [E["MUL"], 4, 3, 1, 0], [E["GAUSS_LOGPDF_GRAD"], 5, 0, 1, 4]]
    params = [[0, 0], [2.0, 1e-6], [0, 0], [3.0, 0]]
    cd = (c_int * 20)(*[v for ins in code for v in ins])
    pr = (c_double * 8)(*[float(v) for p in params for v in p])
    ins = (c_void_p * 2)(0x1000, 0x2000)
    outs = (c_void_p * 3)(0x3000, 0x4000, 0x5000)
    istr = (c_long * 4)(0, 0, 1, 0)
    ostr = (c_long * 6)(1, 0, 0, 0, 0, 0)
    oregs = (c_int * 3)(4, 5 + H.EW_PROG_SUM, 7 + H.EW_PROG_SUM)
    shape = (c_long * 2)(8192, 1)
    for suf in ("_f32", "_f64"):
        n, red = c_long(0), c_int(0)
        src = ctypes.create_string_buffer(8192)
        L.call("hb_ewise_jit_build" + suf, 4, cd, pr, 2, ins, istr, 3, outs, oregs, ostr, 2, shape, None, ctypes.byref(n),
               ctypes.byref(red), src, 8192)
        text = src.value.decode()
        assert n.value == 8192 and red.value == 1
        assert "typedef %s T;" % ("float" if suf == "_f32" else "double") in text
        assert "ew_apply<T>(%d, r0, r1, r4, r3" % E["GAUSS_LOGPDF_GRAD"] in text and "block_sum(racc1, smem)" in text
    # a bad program is reported, not compiled
    oregs_bad = (c_int * 3)(39, 5, 7)
    with pytest.raises(_lib.HipBackendError):
        L.call("hb_ewise_jit_build_f32", 4, cd, pr, 2, ins, istr, 3, outs, oregs_bad, ostr, 2, shape, None, ctypes.byref(n),
               ctypes.byref(red), None, 0)


def test_no_gpu_means_loud_failure_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")

    class M(hb.model.Model):
        def setUp(self):
            self.p = hb.param.Variable([2])

        @hb.model.AutoOptimize()
        def f(self):
            return -tf.reduce_sum(tf.square(self.p))

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        M().f().run()


# ---------------------------------------------------------------- parameter tree (reference testing/test_param.py)
def test_naming_and_tree():
    m = hb.model.Model()
    m.p = hb.param.Variable([2, 3])
    m.sub = hb.param.Parameterized()
    m.sub.q = hb.param.Variable([1])
    assert m.p.name == "p" and m.p.long_name == "model.p"
    assert m.sub.q.long_name == "model.sub.q"
    assert m.sub.q.highest_parent is m
    assert [v.long_name for v in m.get_variables()] == ["model.p", "model.sub.q"]
    orphan = hb.param.Variable([1])
    assert orphan.name == "unnamed"
    m.lst = hb.param.ParamList([hb.param.Variable([1]), hb.param.Variable([2])])
    assert m.lst[1].name == "item1" and m.lst[1].long_name == "model.lst.item1"
    with pytest.raises(TypeError):
        m.lst[0] = "x"


def test_truncated_normal_bounds_and_assign():
    # reference testing/test_param.py:286-296: initial values within two standard deviations
    np.random.seed(0)
    v = hb.param.Variable([1000], mean=1.0, stddev=0.5)
    assert np.all(np.abs(v._host_raw - 1.0) <= 1.0 + 1e-12)
    m = hb.model.Model()
    m.a = hb.param.Variable([3], transform=hb.transforms.positive)
    m.a = np.array([0.5, 1.0, 2.0])
    assert m.a._assigned
    assert np.allclose(hb.transforms.positive.forward(m.a._host_raw), [0.5, 1.0, 2.0])
    m.a = 3.0  # scalar assignment broadcasts (reference param.py:397-402)
    assert np.allclose(hb.transforms.positive.forward(m.a._host_raw), 3.0)


def test_tf_mode_returns_tensors_and_feed_order():
    # reference testing/test_param.py:117-149, test_variationals.py:205-222
    m = hb.model.Model()
    m.v = hb.variationals.Normal([3], collections=hb.param.graph_key.LOCAL)
    m.w = hb.variationals.Normal([2], q_shape="fullrank", n_layers=[4], collections=hb.param.graph_key.LOCAL)
    m.g = hb.param.Variable([2])
    assert m.v.feed_size == 6 and m.w.feed_size == 2 + 4
    assert isinstance(m.g, hb.param.Variable)
    x = G.leaf("data", (7, 6), var=None)
    xw = G.leaf("data", (4, 7, 6), var=None)
    with m.tf_mode():
        assert isinstance(m.g, G.Tensor)
        m.v = x
        m.w = xw
        tv, tw = m.v, m.w
        assert isinstance(m.get_variables()[0], hb.param.Variable)  # works in tf_mode too
    assert tv.shape == (7, 3) and tw.shape == (4, 7, 2)
    # q_mu takes the first `size` columns, q_sqrt the rest
    mu_t = m.v.q_mu._tensor
    assert mu_t.node.inputs[0].node.attrs["offset"] == 0 if mu_t.node.op == "reshape" else True
    sq = m.v.q_sqrt._tensor
    src = sq if sq.node.op == "strided" else sq.node.inputs[0]
    assert src.node.attrs["offset"] == 3 and src.shape == (7, 3)
    assert m.w.q_sqrt._tensor.shape == (4, 7, 2, 2)
    # an unfed LOCAL variational is an error (reference model.py:103-105)
    m2 = hb.model.Model()
    m2.z = hb.variationals.Normal([2], collections=hb.param.graph_key.LOCAL)
    with pytest.raises(ValueError, match="not fed"):
        with m2.tf_mode():
            m2.z


def test_kl_tree_and_collections():
    m = hb.model.Model()
    assert isinstance(m.KL(), np.ndarray) and m.KL() == 0  # no variational: numpy zero (param.py:557-558)
    m.a = hb.variationals.Normal([3])
    m.b = hb.variationals.Gaussian([2], collections=["other"])
    with m.tf_mode():
        kl_all = m.KL()
        kl_other = m.KL("other")
    assert isinstance(kl_all, G.Tensor) and kl_all.size == 1
    assert isinstance(kl_other, G.Tensor)
    ops = {n.op for n in G.topo_order([kl_other])}
    assert "diag_sample_kl" in ops
    assert len([n for n in G.topo_order([kl_other]) if n.op == "diag_sample_kl"]) == 1
    assert len([n for n in G.topo_order([kl_all]) if n.op == "diag_sample_kl"]) == 2


def test_data_and_minibatch_data():
    # reference testing/test_data.py
    d = hb.param.Data(np.ones((3, 2)))
    assert np.all(d.value == 1)
    with pytest.raises(ValueError):
        d.assign(np.ones((4, 2)))
    with pytest.raises(NotImplementedError):
        hb.param.Data(np.array(["a"]))
    mb = hb.param.MinibatchData(np.arange(20).reshape(10, 2))
    assert mb.data_size == 10 and mb.shape == [2]
    fd = mb.get_feed_dict(np.array([1, 3]))
    assert np.array_equal(list(fd.values())[0], [[2, 3], [6, 7]])
    assert mb.get_feed_dict(None) == {}

    class M(hb.model.Model):
        def setUp(self):
            self.a = hb.param.MinibatchData(np.zeros((10, 1)))
            self.b = hb.param.MinibatchData(np.zeros((11, 1)))

    with pytest.raises(ValueError, match="not the same size"):
        M().validate()


def test_settings_push_pop_and_clip():
    # reference testing/test_tf_wraps.py:10-42
    assert hb.settings.numerics.jitter_level == 1e-5 and hb.settings.numerics.clip_by_value is False
    x = G.leaf("data", (3,), var=None)
    assert hb.tf_wraps.clip(x) is x
    cfg = hb.settings.get_settings()
    cfg.numerics.clip_by_value = True
    cfg.numerics.jitter_level = 3e-4
    with hb.settings.temp_settings(cfg):
        assert hb.settings.numerics.jitter_level == 3e-4
        c = hb.tf_wraps.clip(x)
        assert c.node.attrs["f"] == "CLIP" and c.node.attrs["p"][:2] == (-50.0, 50.0)
    assert hb.settings.numerics.jitter_level == 1e-5


def test_indexer_and_autooptimize_cache():
    np.random.seed(0)
    idx = hb.session.Indexer()
    idx.setUp(100)
    assert idx.train_size == 90 and idx.test_size == 10
    assert set(idx.train_index(500)).isdisjoint(set(idx.test_index(50)))

    class M(hb.model.Model):
        def setUp(self, k=1):
            self.k = k
            self.p = hb.param.Variable([2])

        @hb.model.AutoOptimize()
        def f(self):
            return -tf.reduce_sum(tf.square(self.p))

    m = M(k=3)  # setUp(**kw) (reference testing/test_model.py:137-146)
    assert m.k == 3
    assert m.f() is m.f()
    assert hasattr(m, "_f_AF_optimizer")


def test_transforms_roundtrip_and_log_jacobian():
    # reference testing/test_transforms.py:39-75 (numerical Jacobian instead of TF's)
    import graph_oracle as GO

    x = np.random.RandomState(0).randn(6)
    for t in (hb.transforms.Identity(), hb.transforms.Exp(), hb.transforms.Log1pe(), hb.transforms.Logistic(7.3, 19.4)):
        y = t.forward(x)
        assert np.allclose(t.backward(y), x, atol=1e-4)
        leaf = G.leaf("data", x.shape, var=None)
        fw, lj = t.tf_forward(leaf), t.tf_log_jacobian(leaf)
        vals = GO.evaluate([fw, lj], {leaf: x})
        assert np.allclose(vals[fw].numpy(), y, atol=1e-10)
        h = 1e-6
        num = np.sum(np.log((t.forward(x + h) - t.forward(x - h)) / (2 * h)))
        assert np.isclose(vals[lj].numpy().sum(), num, atol=1e-5)


# ---------------------------------------------------------------- data parallel (gloo, world_size 2)
_DP_SCRIPT = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from henbun_amd import parallel
dist.init_process_group("gloo")
rank, world = parallel.world()
assert world == 2
ids = parallel.rng_stream_ids(rank)
gathered = [None, None]
dist.all_gather_object(gathered, ids)
assert gathered[0]["global"] == gathered[1]["global"] == 0
assert gathered[0]["local"] != gathered[1]["local"] and gathered[0]["index"] != gathered[1]["index"]
assert len({v for g in gathered for k, v in g.items() if k != "global"}) == 4
spans = [parallel.shard_rows(1001, r, world) for r in range(world)]
assert spans[0][0] == 0 and spans[-1][1] == 1001 and spans[0][1] == spans[1][0]
# flat-gradient exchange: segments summed in place, mean folded into the Adam scale
g = torch.arange(10, dtype=torch.float64) * (rank + 1)
parallel.allreduce_gradient(g, [(0, 4), (6, 4)])
exp = torch.arange(10, dtype=torch.float64) * 3
exp[4:6] = torch.arange(4, 6, dtype=torch.float64) * (rank + 1)   # outside the segments: untouched
assert torch.equal(g, exp), (rank, g)
assert parallel.gradient_scale(world, "mean") == 0.5 and parallel.gradient_scale(world, "sum") == 1.0
# a mean-reduced step equals the single-process step on the pooled gradient
theta = torch.ones(10, dtype=torch.float64)
theta -= 0.1 * parallel.gradient_scale(world, "mean") * g
ref = torch.ones(10, dtype=torch.float64) - 0.1 * 0.5 * exp
assert torch.allclose(theta, ref)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_data_parallel_helpers_gloo_two_ranks(tmp_path):
    script = tmp_path / "dp.py"
    script.write_text(_DP_SCRIPT % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def _bench_json(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bench_gpus_2_launches_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher around it starts the two ranks itself (torch.distributed.run on
    127.0.0.1, children of a parent that never touches a GPU) and rank 0's ONE JSON line says n_gpus = 2.  --dry-run
    stops each rank before device bring-up (gloo group, shard arithmetic, one all-reduce): the N > 1 plumbing on CPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3",
                        "--warmup", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _bench_json(r.stdout)
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["plumbing_ok"] is True
    assert out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 2 * out["config"]["per_gpu_batch"]
    assert out["config"]["shard_rows_rank0"] == 500_000


def test_bench_refuses_a_world_size_that_is_not_what_was_asked_for():
    """--gpus 2 inside a 1-rank launch (WORLD_SIZE=1) must fail loudly instead of printing an n_gpus = 1 line."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_side_job_list_can_be_discarded_without_a_gpu():
    """hb_side_discard / hb_side_pending are host-side list operations (no launch): exported and callable on CPU."""
    from henbun_amd import _lib

    lib = _lib.lib()
    assert lib.raw("hb_side_pending")() == 0
    assert lib.raw("hb_side_discard")() == 0


def test_serial_chain_source_generates_and_compiles_for_gfx950():
    """A serial chain (csrc/chain.cuh) recorded WITHOUT a GPU -- the chain-aware entry points push a job instead of
    launching while hb_chain_begin is in effect -- turns into one kernel whose source hiprtc compiles for gfx950: the
    prelude (ew_math / ew_apply / rng_core / chain_bodies) is self-contained and the generated calls match the bodies."""
    import ctypes

    from henbun_amd import _lib

    lib = _lib.lib()
    if not lib.raw("hb_ewise_jit_available")():
        pytest.skip("hiprtc is not loadable in this process")
    fake = lambda k: ctypes.c_void_p(0x10000 * (k + 1))      # never dereferenced: nothing is launched
    assert lib.raw("hb_chain_discard")() == 0
    lib.call("hb_chain_begin")
    try:
        ws = fake(20)
        lib.call("hb_gauss_ll_f32", fake(0), fake(1), None, fake(2), 2048, fake(3), fake(4), fake(5), fake(6), ws, 64, None)
        lib.call("hb_adam_step_f32", fake(7), fake(8), fake(9), fake(10), 1500, 1e-3, 0.9, 0.999, 1e-8, -1.0, fake(11), 1,
                 fake(12), 1, None, fake(13), None)
        buf = ctypes.create_string_buffer(1 << 16)
        lib.call("hb_chain_source", buf, 1 << 16)
        src = buf.value.decode()
        assert "hb_gauss_ll_single_body<T>" in src and "hb_adam_body<T>" in src and src.count("__syncthreads();") == 1
        assert "__launch_bounds__(1024)" in src and "0x" not in src      # no addresses in the text: it is the cache key
        lib.call("hb_chain_compile_dry")
    finally:
        lib.raw("hb_chain_discard")()
    assert lib.raw("hb_chain_discard")() == 0


def _gate_graph(E=4, n=4096):
    from henbun_amd import graph as G

    G.reset_interning()
    f_all = G.leaf("var", (2 * E, 1, n), name="f_all")
    kr, kv = G.leaf("var", (1,), name="kr"), G.leaf("var", (1,), name="kv")
    y = G.leaf("var", (1, n), name="y")
    f_e = f_all[:E, 0, :]
    g_e = f_all[E:, 0, :] * G.unary("SQRT", kr)
    w = G.unary("EXP", g_e - G.reduce_max(g_e, 0, keep_dims=True))
    w = w / G.reduce_sum(w, 0, keep_dims=True)
    f = G.reduce_sum(w * f_e, 0, keep_dims=True) * kv
    loss = G.reshape(G.reduce_sum(G.square(f - y)), [])
    return G, loss, G.gradients(loss, [f_all, kr, kv])


def test_softmax_gate_becomes_two_column_clusters_and_an_in_place_concatenation():
    """The softmax gate of the expert mixture (reference notebooks/Expert_GPR.ipynb:139-147) and its VJP: the planner
    groups the elementwise ops, the tf.reduce_max / reduce_sum over the expert axis and the row-block slices of the
    batched draw into two column programs (graph.cluster_columns); the gradient of the draw is the CONCATENATION of the
    experts' and the gates' halves (leading-axis _scatter_partition), not two zero-filled scatters added up."""
    G, loss, grads = _gate_graph()
    order = G.topo_order([loss] + grads)
    cm = G.cluster_columns(order, outputs=[loss] + grads)
    clusters = list({id(c): c for c in cm.values()}.values())
    assert len(clusters) == 2 and all((c.R, c.n) == (4, 4096) for c in clusters)
    fwd, bwd = sorted(clusters, key=lambda c: order.index(c.nodes[0]))
    assert sum(m.op == "strided" for m in fwd.nodes) == 2 and sum(m.op == "reduce" for m in fwd.nodes) == 3
    assert sum(m.op == "reduce" for m in bwd.nodes) >= 2
    # no row reduction, slice, scatter or add of [2E, n] arrays is left outside
    left = [m for m in order if m.id not in cm]
    assert not [m for m in left if m.op in ("strided", "scatter_strided")]
    assert not [m for m in left if m.op == "reduce" and m.attrs["K2"] > 1]
    assert grads[0].node.op == "concat" and grads[0].node.attrs["axis"] == 0
    # regular elementwise clustering leaves the column members alone
    reg = G.cluster_elementwise(order, skip=cm)
    assert not set(reg) & set(cm)
    # short arrays stay with the plain clusters
    G2, loss2, grads2 = _gate_graph(n=64)
    assert not G2.cluster_columns(G2.topo_order([loss2] + grads2), outputs=[loss2] + grads2)


def test_column_program_source_compiles_for_gfx950_without_a_device():
    import ctypes
    from ctypes import c_double, c_int, c_long, c_void_p

    from henbun_amd import _lib
    from henbun_amd import hip_ops as H

    lib = _lib.lib()
    if not lib.raw("hb_ewise_jit_available")():
        pytest.skip("hiprtc not available")
    E = H.EW
    code = [[H.COLPROG_MAX, 1, 0, -1, -1], [E["SUB"], 2, 0, 1, -1], [E["EXP"], 3, 2, -1, -1], [H.COLPROG_SUM, 4, 3, -1, -1],
            [E["DIV"], 5, 3, 4, -1]]
    codeA = (c_int * 25)(*[v for r in code for v in r])
    par = (c_double * 10)(*[0.0] * 10)
    src = ctypes.create_string_buffer(8192)
    for suf in ("_f32", "_f64"):
        lib.call("hb_ewise_colprog_build" + suf, 5, codeA, par, 1, (c_void_p * 1)(None), (c_long * 2)(1000, 1), 2,
                 (c_void_p * 2)(None, None), (c_int * 2)(5, 4), (c_long * 4)(1000, 1, 0, 1), 4, 1000, None, src, 8192)
        text = src.value.decode()
        assert "hb_jit_kernel" in text and "r4[0] = acc" in text and "for (int q = 0; q < 4; ++q)" in text
    # a register read before it is written is refused
    bad = (c_int * 5)(E["EXP"], 3, 2, -1, -1)
    with pytest.raises(_lib.HipBackendError):
        lib.call("hb_ewise_colprog_build_f32", 1, bad, par, 1, (c_void_p * 1)(None), (c_long * 2)(1000, 1), 1,
                 (c_void_p * 1)(None), (c_int * 1)(3), (c_long * 2)(1000, 1), 4, 1000, None, src, 8192)


def test_host_code_of_the_abi_is_clean_under_asan_and_ubsan():
    """tools/host_sanitize/run.sh: a HOST-ONLY, ASan + UBSan instrumented build of csrc/*.hip and a C++ driver that walks
    the ABI's host code -- program validators, the hiprtc source generators (elementwise, column programs, serial chains:
    dry-run compiles), the side-job / chain recorders, ~20 argument-check failures -- without a GPU.  Every call either
    succeeds on the host or returns an error with a message; no sanitizer report."""
    import shutil
    import subprocess

    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run(["bash", os.path.join(root, "tools", "host_sanitize", "run.sh")], capture_output=True, text=True, timeout=600)
    out = p.stdout + p.stderr
    assert p.returncode == 0 and "all checks passed" in out, out[-3000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-3000:]
