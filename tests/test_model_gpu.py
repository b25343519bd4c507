"""GPU end-to-end parity: the compiled ELBO path (trace -> autodiff -> HIP plan
-> fused Adam) against the CPU oracle, through the reference-style model API.

North-star bar: ELBO and gradients within 1e-5 relative on fp64 inputs at
injected noise; tolerances are written at each assertion."""
import os

import numpy as np
import pytest
import torch

import henbun_amd as hb
import henbun_oracle as O

from henbun_amd.models import SVGP, Amortised, DenseGPR, svgp_data
from parity import observe, tile_err

pytestmark = pytest.mark.gpu
tf = hb.tf


def raw(v):
    return O.T(v._host_raw) if v._host_raw is not None else None


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def make_svgp(N, M, n, q_shape, dtype, seed=0, residual="diagonal"):
    np.random.seed(seed)
    rng = np.random.RandomState(seed)
    X, Y, Z = svgp_data(N, M, seed)
    eps = rng.randn(N)
    m = SVGP(X=X, Y=Y, Z=Z, q_shape=q_shape, residual=residual, eps=eps, dtype=dtype)
    m.gp.kern.lengthscales = np.ones(1) * 0.9
    m.k_var = np.ones(1) * 1.3
    m.var = np.ones(1) * 0.4
    if q_shape == "fullrank":
        m.u.q_sqrt = 0.3 * np.eye(M) + 0.02 * rng.randn(M, M)
    u = rng.randn(M)
    m.u.inject_noise(u)
    idx = rng.randint(0, N, n)
    return m, (X, Y, Z, eps, u, idx)


def oracle_svgp(m, data, jitter, q_shape, residual="diagonal"):
    X, Y, Z, eps, u, idx = data
    M = Z.shape[0]
    sess = m._session
    params = {
        "z": O.T(sess.read_raw(m.gp.z)), "ell_raw": O.T(sess.read_raw(m.gp.kern.lengthscales)),
        "q_mu": O.T(sess.read_raw(m.u.q_mu)).reshape(1, M), "q_sqrt": O.T(sess.read_raw(m.u.q_sqrt)),
        "k_var_raw": O.T(sess.read_raw(m.k_var)), "var_raw": O.T(sess.read_raw(m.var)),
    }
    fn = lambda p: O.svgp_elbo(p, O.T(X[idx]), O.T(Y[idx]), float(X.shape[0]), O.T(u), O.T(eps[idx]), jitter=jitter,
                               q_shape=q_shape, residual=residual)
    return fn, params


NAMES = [("model.gp.z", "z"), ("model.gp.kern.lengthscales", "ell_raw"), ("model.u.q_mu", "q_mu"),
         ("model.u.q_sqrt", "q_sqrt"), ("model.k_var", "k_var_raw"), ("model.var", "var_raw")]


@pytest.mark.parametrize("q_shape,N,M,n", [("diagonal", 1000, 64, 1000),    # BASELINE cfg 1 sizes
                                           ("diagonal", 3000, 200, 700),
                                           ("fullrank", 2000, 96, 512),
                                           ("diagonal", 5000, 512, 2048),    # cfg-2 inducing count
                                           ("diagonal", 20000, 512, 8192)])  # cfg-2 inducing count AND minibatch
def test_svgp_fp64_elbo_and_gradient_parity(q_shape, N, M, n):
    cfg = hb.settings.get_settings()
    cfg.numerics.jitter_level = 1e-5
    with hb.settings.temp_settings(cfg):
        m, data = make_svgp(N, M, n, q_shape, "float64")
        opt = m.ELBO()
        opt.compile()
        val, grads = opt.gradients(minibatch_size=n, indices=data[5])
        assert np.isclose(opt.run(minibatch_size=n, indices=data[5]), val, rtol=1e-12)
        fn, params = oracle_svgp(m, data, 1e-5, q_shape)
        ref_val, ref = O.grads_of(fn, params)
    assert abs(val - ref_val.item()) <= 1e-5 * abs(ref_val.item()), (val, ref_val.item())
    for mine, theirs in NAMES:
        e = rel_err(grads[mine], ref[theirs].numpy())
        assert e <= 1e-5, "%s: relative gradient error %.3e > 1e-5" % (mine, e)


def test_svgp_fp32_tracks_fp64():
    """fp32 (the benchmark dtype) against the fp64 oracle: reported, loosely bounded (cond(Kmm) ~ 1e5)."""
    cfg = hb.settings.get_settings()
    cfg.numerics.jitter_level = 1e-5
    with hb.settings.temp_settings(cfg):
        m, data = make_svgp(4000, 256, 1024, "diagonal", "float32")
        opt = m.ELBO()
        opt.compile()
        val, grads = opt.gradients(minibatch_size=1024, indices=data[5])
        fn, params = oracle_svgp(m, data, 1e-5, "diagonal")
        ref_val, ref = O.grads_of(fn, params)
    # observed on MI355X (round 3): ELBO 6.3e-6; worst 32-entry tile of z 4.9e-3, lengthscales 8.0e-4, q_mu 6.9e-4,
    # q_sqrt 6.4e-4, k_var 8.6e-6, var 8.8e-6 (jitter 1e-5: cond(Kmm) ~ 1e5, fp32 unit roundoff 6e-8)
    observe("svgp_fp32_tracks_fp64/ELBO", abs(val - ref_val.item()) / abs(ref_val.item()), 5e-5)
    bound = {"model.gp.z": 4e-2, "model.gp.kern.lengthscales": 6e-3, "model.u.q_mu": 6e-3, "model.u.q_sqrt": 6e-3}
    for mine, theirs in NAMES:
        # worst 32-entry tile of every leaf gradient (tests/parity.py), not max-norm over the whole leaf
        observe("svgp_fp32_tracks_fp64/" + mine, tile_err(grads[mine], ref[theirs].numpy()), bound.get(mine, 8e-5))


@pytest.mark.parametrize("capture,fuse", [(True, True), (False, True), (True, False)])
def test_svgp_adam_trajectory_matches_oracle(capture, fuse):
    """10 Adam steps at fixed noise/minibatch == the oracle's TF-formula Adam on autograd gradients
    (hipGraph replay or eager launches; elementwise clusters fused into one launch or not)."""
    cfg = hb.settings.get_settings()
    cfg.numerics.jitter_level = 1e-5
    cfg.runtime.graph_capture = capture
    cfg.runtime.fuse_elementwise = fuse
    with hb.settings.temp_settings(cfg):
        m, data = make_svgp(1500, 48, 400, "diagonal", "float64", seed=3)
        opt = m.ELBO()
        opt.compile(optimizer=tf.train.AdamOptimizer(0.01))
        fn, params = oracle_svgp(m, data, 1e-5, "diagonal")
        names = list(params)
        leaves = [params[k].clone() for k in names]
        adam = O.AdamTF(leaves, lr=0.01)
        for _ in range(10):
            _, g = O.grads_of(fn, dict(zip(names, leaves)))
            adam.step([-g[k] for k in names])  # minimise -ELBO
        opt.optimize(maxiter=10, minibatch_size=400, indices=data[5])
    sess = m._session
    got = {"z": sess.read_raw(m.gp.z), "ell_raw": sess.read_raw(m.gp.kern.lengthscales),
           "q_mu": sess.read_raw(m.u.q_mu), "q_sqrt": sess.read_raw(m.u.q_sqrt),
           "k_var_raw": sess.read_raw(m.k_var), "var_raw": sess.read_raw(m.var)}
    for k, ref in zip(names, leaves):
        assert rel_err(got[k], ref.numpy()) <= 1e-6, k


def test_trailing_transforms_follow_host_side_parameter_writes():
    """The start-of-step transforms run at the END of a step for the next replay (model.py: trailing transforms).  A
    host-side assignment, a restore, or another plan's update between two replays makes those values stale: the
    session's parameter version makes the optimiser recompute them first.  Same call sequence with the mechanism on and
    off => bit-identical parameters; and the plan really has the transform step behind the update."""
    res = {}
    for trailing in (True, False):
        cfg = hb.settings.get_settings()
        cfg.numerics.jitter_level = 1e-4
        cfg.runtime.trailing_transforms = trailing
        with hb.settings.temp_settings(cfg):
            m, data = make_svgp(3000, 64, 512, "diagonal", "float32", seed=11)
            opt = m.ELBO()
            opt.compile(optimizer=tf.train.AdamOptimizer(0.01))
            opt.optimize(maxiter=3, minibatch_size=512, indices=data[5])
            m.k_var = np.ones(1) * 0.7                     # deferred host assignment between two replays
            opt.optimize(maxiter=2, minibatch_size=512, indices=data[5])
            v = opt.run(minibatch_size=512, indices=data[5])      # another plan reads the parameters in between
            m.gp.kern.lengthscales = np.ones(1) * 1.1
            opt.optimize(maxiter=2, minibatch_size=512, indices=data[5])
            plan = opt.last_plan
            assert bool(plan.prologue) == trailing
            res[trailing] = (m._session.theta.clone(), v)
    assert torch.equal(res[True][0], res[False][0]) and res[True][1] == res[False][1]


def test_early_start_forward_through_the_model_api():
    """settings.runtime.early_forward (opt-in): the compiled step launches the forward contraction inside the persistent
    Cholesky's grid -- together with the side jobs of that launch (minibatch gather, sample of q(u)), whose outputs the
    forward reads after their release.  Same minibatches, same noise streams: the training trajectory stays within fp32
    rounding of the two-launch form, the plan really carries the fused step, and replays of the captured graph are
    repeatable."""
    res = {}
    for early in (True, False):
        cfg = hb.settings.get_settings()
        cfg.numerics.jitter_level = 1e-4
        cfg.runtime.early_forward = early
        with hb.settings.temp_settings(cfg):
            m, data = make_svgp(6000, 128, 2048, "diagonal", "float32", seed=5)
            opt = m.ELBO()
            opt.compile(optimizer=tf.train.AdamOptimizer(0.01))
            opt.optimize(maxiter=8, minibatch_size=2048)
            plan = opt.last_plan
            labels = [plan.step_labels.get(id(s), "other") for s in plan.steps]
            assert ("cholesky+sgp" in " ".join(labels)) == early, labels
            res[early] = m._session.theta.clone()
    a, b = res[True].double(), res[False].double()
    # eight Adam steps of 0.01 each: a parameter moves by up to 0.08; the two forms differ by fp32 rounding in A
    assert bool(((a - b).abs() <= 5e-4 + 1e-4 * b.abs()).all())    # observed: 2e-5 of the largest inducing point, 4.6e-5 in one hyper-parameter


def _two_rank_sessions(N, M, n, lr, seed=3):
    """Two models in ONE process set up as rank 0 / rank 1 of a world of two (no second process, no process group): shard
    of the data (parallel.shard_rows), rank-distinct `local` / `index` streams (parallel.rng_stream_ids), the global data
    count in the objective, dp_reduce='mean', the data-parallel tail built by the optimiser (pack -> exchange -> Adam,
    eager form).  The exchange itself -- what the all-reduce returns -- is done by the test: the two flat buffers are added."""
    from henbun_amd import parallel

    rng = np.random.RandomState(seed)
    X, Y, Z = svgp_data(N, M, seed)
    eps, u = rng.randn(N), rng.randn(M)
    ranks = []
    for r in range(2):
        b, e = parallel.shard_rows(N, r, 2)
        np.random.seed(seed)                       # every rank starts from the same parameters
        m = SVGP(X=X[b:e], Y=Y[b:e], Z=Z, eps=eps[b:e], dtype="float64")
        m.N = N                                     # the objective estimates the FULL-data ELBO on every rank
        m.gp.kern.lengthscales = np.ones(1) * 0.9
        m.k_var = np.ones(1) * 1.3
        m.var = np.ones(1) * 0.4
        m.u.inject_noise(u)                         # the global noise agrees across ranks
        m.initialize()
        sess = m._session
        sess.rank, sess.world_size = r, 2           # (after the layout exists: nothing is broadcast)
        sess.reseed(sess.seed)
        assert sess.rngs["global"].stream_id == 0 and sess.rngs["local"].stream_id == parallel.rng_stream_ids(r)["local"]
        idx = rng.randint(0, e - b, n)
        opt = m.ELBO()
        opt.compile(optimizer=tf.train.AdamOptimizer(lr), dp_reduce="mean")
        plan = opt._get_plan("opt", n)
        opt._apply_indices(plan, idx)
        assert plan.dp_mode in ("rccl-eager", "torch-eager") and len(plan.eager_tail) == 3
        ranks.append(dict(m=m, sess=sess, opt=opt, plan=plan, idx=idx, lo=b))
    return ranks, (X, Y, Z, eps, u)


def _two_rank_step(ranks, break_rank=None):
    """forward + backward + pack on both ranks, the test-side sum of the two flat buffers, Adam on both ranks."""
    for rk in ranks:
        rk["plan"].run()
    if break_rank is not None:       # a factorisation failed on one rank: its status word goes non-zero before the pack
        with ranks[break_rank]["plan"]._on_stream():
            ranks[break_rank]["plan"].info_words()[:1].fill_(7)
    for rk in ranks:
        with rk["plan"]._on_stream():
            rk["plan"].eager_tail[0]()                       # hb_dp_pack
    torch.cuda.synchronize()
    parts = [rk["plan"].gflat.clone() for rk in ranks]
    total = parts[0] + parts[1]                              # what ONE all-reduce(sum) of the flat buffer returns
    for rk in ranks:
        rk["plan"].gflat.copy_(total)
    torch.cuda.synchronize()
    for rk in ranks:
        with rk["plan"]._on_stream():
            rk["plan"].eager_tail[2]()                       # hb_adam_step with gscale = -1/2
    torch.cuda.synchronize()
    return parts, total


def test_two_rank_data_parallel_step_equals_the_oracle_on_the_pooled_minibatch():
    """the N > 1 step is the right step.  Two ranks' packed gradients, summed (the all-reduce) and
    scaled by dp_reduce='mean' inside hb_adam_step, equal the oracle's gradient of the ELBO on the POOLED 2 n minibatch,
    and the updated parameters equal the oracle's TF-formula Adam step on it (fp64, injected noise, <= 1e-5); both ranks
    end bit-identical.  Then the failure case: rank 1's factorisation fails => neither rank updates, both latch it."""
    N, M, n, lr = 3000, 48, 400, 0.01
    cfg = hb.settings.get_settings()
    cfg.numerics.jitter_level = 1e-5
    cfg.runtime.force_dp = True
    cfg.runtime.dp_exchange = "eager"
    with hb.settings.temp_settings(cfg):
        ranks, (X, Y, Z, eps, u) = _two_rank_sessions(N, M, n, lr)
        s0 = ranks[0]["sess"]
        P = s0.theta.numel()
        theta_before = s0.theta.clone()
        assert torch.equal(theta_before, ranks[1]["sess"].theta)
        # the oracle on the pooled minibatch (rows of rank 0's draw, then rank 1's)
        rows = np.concatenate([rk["lo"] + rk["idx"] for rk in ranks])
        m0 = ranks[0]["m"]
        params = {"z": O.T(s0.read_raw(m0.gp.z)), "ell_raw": O.T(s0.read_raw(m0.gp.kern.lengthscales)),
                  "q_mu": O.T(s0.read_raw(m0.u.q_mu)).reshape(1, M), "q_sqrt": O.T(s0.read_raw(m0.u.q_sqrt)),
                  "k_var_raw": O.T(s0.read_raw(m0.k_var)), "var_raw": O.T(s0.read_raw(m0.var))}
        fn = lambda p: O.svgp_elbo(p, O.T(X[rows]), O.T(Y[rows]), float(N), O.T(u), O.T(eps[rows]), jitter=1e-5)
        ref_val, ref_g = O.grads_of(fn, params)
        parts, total = _two_rank_step(ranks)
        # (1) the mean of the two ranks' objectives is the pooled ELBO; no failure flag
        assert abs(total[P].item() / 2 - ref_val.item()) <= 1e-9 * abs(ref_val.item())
        assert total[P + 1].item() == 0
        # (2) summed gradient / 2 == the oracle's gradient on the pooled minibatch, leaf by leaf
        off = {id(v): (o, sz) for v, o, sz in s0._layout}
        leaf = {"z": m0.gp.z, "ell_raw": m0.gp.kern.lengthscales, "q_mu": m0.u.q_mu, "q_sqrt": m0.u.q_sqrt,
                "k_var_raw": m0.k_var, "var_raw": m0.var}
        for k, v in leaf.items():
            o, sz = off[id(v)]
            assert rel_err(total[o:o + sz].cpu().numpy() / 2, ref_g[k].numpy()) <= 1e-5, k
        # ... and the two ranks' own gradients differ (different shards, different rows): the sum is not a formality
        assert not torch.equal(parts[0][:P], parts[1][:P])
        # (3) the update: both ranks bit-identical, equal to the oracle's TF-formula Adam step on the pooled gradient
        assert torch.equal(s0.theta, ranks[1]["sess"].theta) and not torch.equal(s0.theta, theta_before)
        names = list(params)
        leaves = [params[k].clone() for k in names]
        O.AdamTF(leaves, lr=lr).step([-ref_g[k] for k in names])
        for k, refp in zip(names, leaves):
            assert rel_err(s0.read_raw(leaf[k]), refp.numpy()) <= 1e-6, k
        # (4) failure containment across ranks: rank 1's status word is non-zero => the flag slot of the SUM is non-zero on
        # both ranks, neither updates, both latch the failure (hb_adam_step), and optimize() would raise on both
        theta_good = s0.theta.clone()
        _, total = _two_rank_step(ranks, break_rank=1)
        assert total[P + 1].item() != 0
        for rk in ranks:
            assert torch.equal(rk["sess"].theta, theta_good)
            step, what = (int(x) for x in rk["plan"].fail.cpu().numpy())
            assert step != 0
            with pytest.raises(hb.CholeskyError):
                rk["opt"]._check_step_failure(rk["plan"])


def test_amortised_fp64_parity_and_training():
    np.random.seed(5)
    rng = np.random.RandomState(0)
    N, Din, H, L, n = 3000, 12, 32, 4, 512
    Z0 = rng.randn(N, L)
    Y = np.tanh(Z0 @ rng.randn(L, Din) / np.sqrt(L)) + 0.1 * rng.randn(N, Din)
    m = Amortised(Y=Y, L=L, H=H, dtype="float64")
    u = rng.randn(n, L)
    m.z.inject_noise(u)
    idx = rng.randint(0, N, n)
    opt = m.ELBO()
    opt.compile(dp_reduce="sum")
    val, grads = opt.gradients(minibatch_size=n, indices=idx)
    sess = m._session
    params = {
        "enc_w0": O.T(sess.read_raw(m.enc.matbias0.w)), "enc_b0": O.T(sess.read_raw(m.enc.matbias0.b)),
        "enc_w1": O.T(sess.read_raw(m.enc.matbias1.w)), "enc_b1": O.T(sess.read_raw(m.enc.matbias1.b)),
        "dec_w0": O.T(sess.read_raw(m.dec.matbias0.w)), "dec_b0": O.T(sess.read_raw(m.dec.matbias0.b)),
        "var_raw": O.T(sess.read_raw(m.var)),
    }
    ref_val, ref = O.grads_of(lambda p: O.amortised_elbo(p, O.T(Y[idx]), O.T(u)), params)
    assert abs(val - ref_val.item()) <= 1e-8 * abs(ref_val.item())
    names = {"model.enc.matbias0.w": "enc_w0", "model.enc.matbias0.b": "enc_b0", "model.enc.matbias1.w": "enc_w1",
             "model.enc.matbias1.b": "enc_b1", "model.dec.matbias0.w": "dec_w0", "model.dec.matbias0.b": "dec_b0",
             "model.var": "var_raw"}
    for k, r in names.items():
        assert rel_err(grads[k], ref[r].numpy()) <= 1e-7, k
    # training with in-kernel noise and device-drawn minibatches improves the held-out objective
    m.z.inject_noise(None)
    before = np.mean([m.ELBO().run(minibatch_size=256, training=False) for _ in range(5)])
    opt2 = m.ELBO()
    opt2.compile(optimizer=tf.train.AdamOptimizer(0.01), dp_reduce="sum")
    opt2.optimize(maxiter=300, minibatch_size=256)
    after = np.mean([m.ELBO().run(minibatch_size=256, training=False) for _ in range(5)])
    assert after > before


def test_dense_gpr_parity():
    np.random.seed(2)
    rng = np.random.RandomState(0)
    n = 40  # notebooks/GaussianProcess.ipynb:75-76 size
    X = np.sort(rng.uniform(0, 6, (n, 1)), axis=0)
    Y = np.sin(X) + 0.3 * rng.randn(n, 1)
    cfg = hb.settings.get_settings()
    cfg.numerics.jitter_level = 1e-4
    with hb.settings.temp_settings(cfg):
        m = DenseGPR(X=X, Y=Y, dtype="float64")
        m.q.q_sqrt = 0.5 * np.eye(n) + 0.02 * rng.randn(n, n)
        u = rng.randn(n)
        m.q.inject_noise(u)
        opt = m.ELBO()
        opt.compile()
        val, grads = opt.gradients()
    sess = m._session
    ps = [v for v in m.get_variables() if v.is_parameter]
    leaves = {v.long_name: O.T(sess.read_raw(v)).clone().requires_grad_(True) for v in ps}
    ell = O.log1pe_forward(leaves["model.kern.lengthscales"])
    L = O.kern_cholesky(O.T(X), ell, 1e-4)
    S = leaves["model.q.q_sqrt"]
    xs = O.sample_fullrank(leaves["model.q.q_mu"], S, O.T(u))
    kl = O.kl_normal(S, O.T(u), xs, "fullrank")
    f = (L @ xs.reshape(n, 1)) * torch.sqrt(O.log1pe_forward(leaves["model.k_var"]))
    elbo = torch.sum(O.gaussian(O.T(Y), f, O.log1pe_forward(leaves["model.var"]))) - kl
    ref = torch.autograd.grad(elbo, [leaves[v.long_name] for v in ps])
    assert abs(val - elbo.item()) <= 1e-8 * abs(elbo.item())
    for v, r in zip(ps, ref):
        assert rel_err(grads[v.long_name], r.numpy()) <= 1e-5, v.long_name


def test_square_model_converges_and_collections():
    """reference testing/test_model.py:8-29,61-74: Adam drives -sum(p^2) to 0 (lr 0.01, 1500 its, atol 1e-4)."""

    class SquareModel(hb.model.Model):
        def setUp(self):
            self.p = hb.param.Variable([2, 3])
            self.q = hb.param.Variable([2, 3], collections=["other"])

        @hb.model.AutoOptimize()
        def likelihood(self):
            return -tf.reduce_sum(tf.square(self.p)) - tf.reduce_sum(tf.square(self.q))

    np.random.seed(0)
    m = SquareModel()
    q0 = None
    m.likelihood().compile(optimizer=tf.train.AdamOptimizer(0.01))
    m.initialize()
    q0 = m.q.value.copy()
    m.likelihood().optimize(maxiter=1500)
    assert np.allclose(m.p.value, 0.0, atol=1e-4)
    assert np.allclose(m.q.value, q0)  # other collection untouched
    assert np.isclose(m.likelihood().run(), -np.sum(q0 ** 2), rtol=1e-4)
    m.likelihood().compile(optimizer=tf.train.AdamOptimizer(0.01), collection="other")
    m.likelihood().optimize(maxiter=1500)
    assert np.allclose(m.q.value, 0.0, atol=1e-4)


def test_assign_value_save_restore(tmp_path):
    """reference testing/test_model.py:53-59,76-105."""

    class M(hb.model.Model):
        def setUp(self):
            self.a = hb.param.Variable([3], transform=hb.transforms.positive)
            self.sub = hb.nn.NeuralNet([2, 3, 1])

    np.random.seed(0)
    m = M()
    m.a = np.array([0.5, 1.0, 2.0])
    m.initialize()
    assert np.allclose(m.a.value, [0.5, 1.0, 2.0], atol=1e-5)
    w0 = m.sub.matbias0.w.value.copy()
    path = str(tmp_path / "ckpt")
    m.save(path)
    sub_path = str(tmp_path / "sub")
    m.sub.save(sub_path)
    m.a = np.array([3.0, 3.0, 3.0])
    m.sub.matbias0.w = np.zeros((2, 3))
    m.initialize()
    assert np.allclose(m.a.value, 3.0, atol=1e-5)
    m.sub.restore(sub_path)
    assert np.allclose(m.sub.matbias0.w.value, w0, atol=1e-6)
    assert np.allclose(m.a.value, 3.0, atol=1e-5)  # sub-tree restore leaves the rest alone
    m.a = np.array([9.0, 9.0, 9.0])  # pending assignment must not override the restore
    m.restore(path)
    m.initialize()
    assert np.allclose(m.a.value, [0.5, 1.0, 2.0], atol=1e-5)


def test_cholesky_failure_is_reported():
    class Bad(hb.model.Model):
        def setUp(self):
            self.A = hb.param.Data(np.array([[1.0, 2.0], [2.0, 1.0]]))

        @hb.model.AutoOptimize()
        def obj(self):
            return tf.reduce_sum(tf.cholesky(self.A))

    with pytest.raises(hb.CholeskyError):
        Bad().obj().run()


@pytest.mark.parametrize("dp", [False, True])
def test_cholesky_failure_inside_optimize_leaves_parameters_at_the_last_good_step(dp):
    """(dp: the same through the data-parallel step, where the failure flag rides behind the gradient in the
    all-reduce and hb_adam_step reads it back as `dpflag`.)  tf.cholesky raises inside session.run before apply_gradients (reference model.py:265-266): a failing step
    must not touch theta / Adam slots / step count, also when it happens in the middle of a captured replay loop."""
    class Drift(hb.model.Model):
        def setUp(self):
            self.a = hb.param.Variable([1])
            self.B = hb.param.Data(np.array([[0.0, 1.0], [1.0, 0.0]]))

        @hb.model.AutoOptimize()
        def obj(self):
            # K = I + a*B is positive definite while |a| < 1; the objective pushes a upwards by lr per step
            K = tf.eye(2) + self.a * self.B
            return tf.reduce_sum(tf.cholesky(K)) * 1e-9 + tf.reduce_sum(self.a) * 100.0

    cfg = hb.settings.get_settings()
    cfg.runtime.force_dp = dp
    with hb.settings.temp_settings(cfg):
        _drift_body(Drift, dp)


def _drift_body(Drift, dp):
    m = Drift(dtype="float64")
    m.a = np.array([0.9])
    opt = m.obj()
    opt.compile(optimizer=tf.train.AdamOptimizer(0.03))
    with pytest.raises(hb.CholeskyError) as e:
        opt.optimize(maxiter=12)        # a: 0.90 -> 0.93 -> 0.96 -> 0.99 -> 1.02 (fails in step 5)
    assert opt.last_plan.dp_mode == ("rccl-in-graph" if dp else "none")
    assert "last good step" in str(e.value)
    a = float(m.a.value[0])
    assert np.isfinite(a) and abs(a - 1.02) < 1e-6, a      # the last applied update is step 4's
    slots = opt._optimizer.slots(m._session)
    assert int(slots["t"].item()) == 4 and slots["fail"].tolist() == [0, 0]
    assert bool(torch.isfinite(slots["m"]).all()) and bool(torch.isfinite(slots["v"]).all())
    m.a = np.array([0.5])               # the caller may repair and continue
    opt.optimize(maxiter=2)
    assert abs(float(m.a.value[0]) - 0.56) < 2e-3


def test_data_parallel_step_on_a_one_rank_rccl_group_is_bit_identical_to_the_single_process_step():
    """The multi-rank step -- pack -> ONE RCCL all-reduce of the flat gradient (hb_allreduce_sum, on the plan's
    stream, inside the captured graph) -> Adam with the all-reduced failure flag -- forced with a one-rank `nccl`
    process group on the one GPU (settings.runtime.force_dp).  A one-rank sum is the identity and the mean's
    1/R is 1, so ten steps must leave exactly the bits of the plain single-process path."""
    import random

    import torch.distributed as dist

    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % random.randint(20000, 40000), rank=0,
                                world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        out = {}
        for dp in (False, "graph", "eager"):
            cfg = hb.settings.get_settings()
            cfg.runtime.force_dp = bool(dp)
            cfg.runtime.dp_exchange = dp or "auto"
            with hb.settings.temp_settings(cfg):
                np.random.seed(11)
                X, Y, Z = svgp_data(5000, 64, 1)
                m = SVGP(X=X, Y=Y, Z=Z, dtype="float32", seed=5)
                opt = m.ELBO()
                opt.compile(optimizer=tf.train.AdamOptimizer(0.01))
                opt.optimize(maxiter=10, minibatch_size=512)
                plan = opt.last_plan
                assert plan.is_captured
                # "eager" is what world_size > 1 runs by default (settings.runtime.dp_exchange = auto): the exchange and
                # Adam follow the captured forward+backward graph as three calls on the plan's stream
                assert plan.dp_mode == {False: "none", "graph": "rccl-in-graph", "eager": "rccl-eager"}[dp], plan.dp_mode
                torch.cuda.synchronize()
                out[dp] = (m._session.theta.clone(), plan.gflat.clone(), opt.dp_objective())
        P = out["graph"][0].numel()
        for dp in ("graph", "eager"):
            assert torch.equal(out[False][0], out[dp][0]), "parameters after 10 steps differ between the DP (%s) and plain step" % dp
            assert torch.equal(out[False][1][:P], out[dp][1][:P])
            assert out[False][2] is None and np.isfinite(out[dp][2])
            assert float(out[dp][1][P + 1].item()) == 0.0          # no factorisation failed
    finally:
        if own_group:
            dist.destroy_process_group()


def test_injected_indices_last_for_one_call_and_the_plan_is_recaptured():
    """`indices=` fixes the rows of that call only (the reference draws per call, model.py:232-267); the plan
    goes back to its drawing hipGraph afterwards instead of staying on the eager path."""
    np.random.seed(0)
    X, Y, Z = svgp_data(3000, 16, 0)
    m = SVGP(X=X, Y=Y, Z=Z, dtype="float64")
    opt = m.ELBO()
    opt.compile()
    idx = np.random.RandomState(1).randint(0, 3000, 256)
    assert np.isfinite(opt.run(256, indices=idx))
    plan = opt.last_plan
    assert plan.is_captured and plan.indices_injected
    assert np.array_equal(plan.index_buffer.cpu().numpy(), idx)
    g_inj = plan._graph
    seen = []
    for _ in range(3):
        opt.run(256)
        assert opt.last_plan is plan and plan.is_captured and not plan.indices_injected
        seen.append(plan.index_buffer.cpu().numpy().copy())
    assert not np.array_equal(seen[0], idx) and not np.array_equal(seen[0], seen[1]) and not np.array_equal(seen[1], seen[2])
    assert np.isfinite(opt.run(256, indices=idx))     # and back: the injected graph is cached, not re-captured
    assert plan._graph is g_inj and np.array_equal(plan.index_buffer.cpu().numpy(), idx)
    opt.optimize(3, 256, indices=idx)
    p = opt.last_plan
    assert p.is_captured and p.indices_injected
    opt.optimize(3, 256)
    assert p.is_captured and not p.indices_injected
    assert not np.array_equal(p.index_buffer.cpu().numpy(), idx)


def test_rng_noise_statistics_and_minibatch_indices():
    np.random.seed(0)
    X, Y, Z = svgp_data(2000, 32, 0)
    m = SVGP(X=X, Y=Y, Z=Z, dtype="float64")
    opt = m.ELBO()
    opt.compile()
    vals = [opt.run(minibatch_size=256) for _ in range(30)]
    assert np.std(vals) > 0  # fresh noise + fresh minibatch every run
    # the index plan honours the 10 % hold-out (reference model.py:132,140-149)
    assert m._index.train_size == 1800 and m._index.test_size == 200
    plan = opt.last_plan
    idx = plan.index_buffer.cpu().numpy()
    assert idx.min() >= 0 and idx.max() < 1800


def test_exact_resume_save_state(tmp_path):
    """save_state / restore_state: parameters + Adam slots + RNG streams + index split give a
    bit-identical continuation, in graph-replay mode, without rebuilding any plan."""
    from henbun_amd.models import SVGP, svgp_data

    np.random.seed(11)
    X, Y, Z = svgp_data(3000, 32, 0, domain=16.0)
    m = SVGP(X=X, Y=Y, Z=Z, dtype="float32")
    opt = m.ELBO()
    opt.compile(optimizer=tf.train.AdamOptimizer(1e-2))
    opt.optimize(maxiter=7, minibatch_size=256)
    opt.run(minibatch_size=256)  # build the evaluation plan now: building a plan draws from the RNG streams once
    path = opt.save_state(str(tmp_path / "ck"))
    opt.optimize(maxiter=9, minibatch_size=256)
    theta_a = m._session.theta.clone()
    e_a = opt.run(minibatch_size=256)
    opt.restore_state(path)
    opt.optimize(maxiter=9, minibatch_size=256)
    assert torch.equal(m._session.theta, theta_a)
    assert opt.run(minibatch_size=256) == e_a
    # a different layout is refused
    m2 = SVGP(X=X, Y=Y, Z=Z[:16], dtype="float32")
    o2 = m2.ELBO()
    o2.compile()
    with pytest.raises(ValueError):
        o2.restore_state(path)


@pytest.mark.parametrize("jitter", [1e-5, 1e-4])
def test_cfg2_full_size_properties_fp32(jitter):
    """BASELINE cfg 2 at full per-step size (M = 512, n = 8192, fp32 -- the benchmark dtype), at the jitter bench.py
    runs with (the reference's default 1e-5, henbunrc:11) and at 1e-4, through properties that do not need a reference
    run: bit-determinism of the whole forward+backward, closeness to the fp64 oracle, and the defining identities of the
    fused factor/inverse/contraction kernels (L L^T = K + jitter I, W L = I, L A = K(z,x), v = 1 - colsum(A^2)).
    Beside the kernels' distance from the fp64 oracle the test records the distance of the ORACLE ITSELF EVALUATED IN
    FLOAT32 (torch-CPU float32, reference op order): "as good as fp32 allows" as a measured statement."""
    cfg = hb.settings.get_settings()
    cfg.numerics.jitter_level = jitter
    with hb.settings.temp_settings(cfg):
        m, data = make_svgp(20000, 512, 8192, "diagonal", "float32")
        opt = m.ELBO()
        opt.compile()
        v1, g1 = opt.gradients(minibatch_size=8192, indices=data[5])
        v2, g2 = opt.gradients(minibatch_size=8192, indices=data[5])
        assert v1 == v2 and all(np.array_equal(g1[k], g2[k]) for k in g1), "same inputs must give the same bits"
        fn, params = oracle_svgp(m, data, jitter, "diagonal")
        ref_val, ref = O.grads_of(fn, params)
    tag = "cfg2_fullsize_fp32[j%g]/" % jitter
    # observed on MI355X (round 4, profiles/r04_observed_errors.txt), jitter 1e-4 / 1e-5 (cond(Kmm) ten times larger):
    # ELBO 5.9e-6 / 1.8e-5; worst 32-entry tile of z 1.3e-3 / 5.2e-3, lengthscales 2.4e-4 / 3.3e-4, q_mu 1.8e-4 / 5.3e-4,
    # q_sqrt 1.6e-4 / 5.8e-4, k_var 1.4e-7 .. 6.6e-6 / 2.3e-5, var 9.5e-6 / 2.9e-5
    observe(tag + "ELBO", abs(v1 - ref_val.item()) / abs(ref_val.item()), 5e-5)
    if jitter >= 1e-4:
        bound = {"model.gp.z": 1e-2, "model.gp.kern.lengthscales": 1.5e-3, "model.u.q_mu": 1.8e-3, "model.u.q_sqrt": 1.6e-3,
                 "model.k_var": 2e-5, "model.var": 1e-4}
    else:
        bound = {"model.gp.z": 4e-2, "model.gp.kern.lengthscales": 3e-3, "model.u.q_mu": 5e-3, "model.u.q_sqrt": 5e-3,
                 "model.k_var": 2e-4, "model.var": 3e-4}
    mine_err = {}
    for mine, theirs in NAMES:
        mine_err[mine] = observe(tag + mine, tile_err(g1[mine], ref[theirs].numpy()), bound[mine])
    # ---- the oracle in float32 at the same point.  (1) the reference's own distance form |a|^2 + |b|^2 - 2 a.b: with
    # inputs out to 256 lengthscales its float32 rounding (~4e-3 absolute in K) leaves Kmm + jitter I indefinite -- the
    # reference graph cannot be evaluated in float32 at this configuration at all; (2) the difference form (what the HIP
    # kernels evaluate), reference op order otherwise: its distance from the float64 oracle is the yardstick.
    X, Y, Z, eps, u, idx = data
    f32 = lambda a: O.T(a, dtype=torch.float32)
    p32 = {k: v.to(torch.float32) for k, v in params.items()}

    def fn32(K):
        return lambda p: O.svgp_elbo(p, f32(X[idx]), f32(Y[idx]), float(X.shape[0]), f32(u), f32(eps[idx]), jitter=jitter,
                                     q_shape="diagonal", residual="diagonal", K=K)

    try:
        v_ref32, _ = O.grads_of(fn32(O.rbf_K), p32)
        ref_form = "evaluates: ELBO %.3e from fp64" % (abs(v_ref32.item() - ref_val.item()) / abs(ref_val.item()))
    except Exception as e:      # torch.linalg.cholesky: not positive definite
        ref_form = "fails (%s)" % type(e).__name__
    v32, g32 = O.grads_of(fn32(O.rbf_K_difference), p32)
    print("float32 oracle at cfg 2, jitter %g: reference distance form %s; difference form: ELBO %.3e from fp64 (HIP %.3e)"
          % (jitter, ref_form, abs(v32.item() - ref_val.item()) / abs(ref_val.item()), abs(v1 - ref_val.item()) / abs(ref_val.item())))
    for mine, theirs in NAMES:
        e32 = tile_err(g32[theirs].numpy(), ref[theirs].numpy())
        print("   %-28s float32 oracle %.3e   HIP fp32 %.3e" % (mine, e32, mine_err[mine]))
        # (recorded beside the kernels' values in profiles/r04_observed_errors.txt; observed 1e-4 / 1e-5: z 3.3e-4 / 2.9e-3,
        # lengthscales 5.3e-5 / 5.2e-4, q_mu 2.6e-5 / 1.3e-4, q_sqrt 3.2e-5 / 1.5e-4, k_var 4.7e-7 / 1.4e-5, var 9.7e-7 / 1.9e-5)
        observe(tag + "oracle_f32/" + mine, e32, 10 * {"model.gp.z": 2.9e-3, "model.gp.kern.lengthscales": 5.2e-4, "model.u.q_mu": 1.3e-4,
                                                        "model.u.q_sqrt": 1.5e-4, "model.k_var": 1.4e-5, "model.var": 1.9e-5}[mine])
        # the kernels stay within a small multiple of what a float32 evaluation of the reference op order gives
        # (floor: a leaf the float32 oracle happens to hit almost exactly)
        # observed ratio HIP / float32 oracle: 0.14 .. 9.5 over the six leaves and both jitters
        observe(tag + "vs_oracle_f32/" + mine, mine_err[mine] / max(e32, 1e-6), 30.0)
    # kernel identities at the same size
    H = m._session.H
    rng = np.random.RandomState(0)
    M, n = 512, 8192
    z = torch.as_tensor(np.linspace(0, 256, M)[:, None], dtype=torch.float32).cuda()
    x = torch.as_tensor(rng.uniform(0, 256, (n, 1)), dtype=torch.float32).cuda()
    ell = torch.ones(1, dtype=torch.float32, device="cuda")
    K = H.gram_fwd(z, z, ell, diag_add=1e-3)
    L, W, info = H.cholesky_inverse(K)
    assert info.item() == 0
    Ld, Wd, Kd = L.double().cpu().numpy(), W.double().cpu().numpy(), K.double().cpu().numpy()
    assert np.abs(Ld @ Ld.T - Kd).max() < 5e-5
    assert np.abs(Wd @ Ld - np.eye(M)).max() < 5e-3
    A = H.sgp_A(x, z, ell, W)
    Kzx = H.gram_fwd(z, x, ell).double().cpu().numpy()
    Ad = A.double().cpu().numpy()
    assert np.abs(Ld @ Ad - Kzx).max() < 2e-3
    u = torch.as_tensor(rng.randn(1, M), dtype=torch.float32).cuda()
    eps = torch.as_tensor(rng.randn(n), dtype=torch.float32).cuda()
    f, A2, v, _ = H.sgp_fwd(x, z, ell, W, u, eps_in=eps)
    assert torch.equal(A2, A)
    vd = 1.0 - (Ad ** 2).sum(0)
    assert np.abs(v.double().cpu().numpy() - vd).max() < 1e-4
    fd = u.double().cpu().numpy() @ Ad + np.sqrt(np.abs(vd)) * eps.double().cpu().numpy()
    assert np.abs(f.double().cpu().numpy() - fd).max() < 2e-3


def test_bf16x3_contraction_mode_tracks_the_fp32_step():
    """settings.numerics.contraction = bf16x3: the forward M^2 n contraction on three-term bf16 operands gives the
    fp32 step's ELBO and gradients to fp32 accuracy (both measured against the fp64 oracle), and trains."""
    res = {}
    for mode in ("native", "bf16x3"):
        cfg = hb.settings.get_settings()
        cfg.numerics.jitter_level = 1e-4
        cfg.numerics.contraction = mode
        with hb.settings.temp_settings(cfg):
            m, data = make_svgp(20000, 512, 4096, "diagonal", "float32")
            opt = m.ELBO()
            opt.compile()
            res[mode] = opt.gradients(minibatch_size=4096, indices=data[5])
            if mode == "bf16x3":
                fn, params = oracle_svgp(m, data, 1e-4, "diagonal")
                ref_val, ref = O.grads_of(fn, params)
                m.u.inject_noise(None)
                m.eps = None
                o2 = m.ELBO()
                o2.compile(optimizer=tf.train.AdamOptimizer(1e-3))
                o2.optimize(maxiter=5, minibatch_size=4096)
                assert np.isfinite(o2.run(minibatch_size=4096))
    (v0, g0), (v1, g1) = res["native"], res["bf16x3"]
    e0, e1 = abs(v0 - ref_val.item()), abs(v1 - ref_val.item())
    assert e1 <= 2.0 * e0 + 1e-4 * abs(ref_val.item()), (e0, e1)
    for mine, theirs in NAMES:
        r0, r1 = rel_err(g0[mine], ref[theirs].numpy()), rel_err(g1[mine], ref[theirs].numpy())
        assert r1 <= 2.0 * r0 + 1e-3, (mine, r0, r1)


def test_injected_indices_out_of_range_raise():
    """A caller-supplied minibatch index outside the data set is an error, not silently zero-filled rows
    (the reference would fail inside tf.gather / numpy indexing, param.py:733-739)."""
    np.random.seed(0)
    X, Y, Z = svgp_data(300, 16, seed=0)
    m = SVGP(X=X, Y=Y, Z=Z, dtype="float64", seed=0)
    opt = m.ELBO()
    opt.compile(optimizer=tf.train.AdamOptimizer(1e-3))
    good = np.arange(64)
    assert np.isfinite(opt.run(minibatch_size=64, indices=good))
    bad = good.copy()
    bad[3] = 300
    with pytest.raises(IndexError):
        opt.run(minibatch_size=64, indices=bad)
    assert np.isfinite(opt.run(minibatch_size=64, indices=good))  # the flag is cleared by the failed call
