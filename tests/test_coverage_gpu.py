"""GPU parity for the remaining hot-path-adjacent surface (SURVEY.md 8a rows a13, a15, a18-a21, a23,
a25, a26; 8f rows 2-4): densities, generic SparseGP branches, CsymRBF, the expert mixture of the
reference notebook, Gaussian / variational-weight networks, prediction-style Model.run."""
import numpy as np
import pytest
import torch

import henbun_amd as hb
import henbun_oracle as O
from henbun_amd import graph as G

from henbun_amd.models import ExpertGPR, svgp_data

pytestmark = pytest.mark.gpu
tf = hb.tf


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_densities_on_device(golden):
    # reference testing/test_densities.py:11-75 (atol 1e-5), test_tf_wraps.py:45-59
    g = golden
    m = hb.model.Model(dtype="float64")
    lp0 = hb.densities.gaussian(g["d_a"], 0.0, 2.0)
    lp1 = hb.densities.student_t(g["d_b"], 0.0, 2.0, 3.0)
    mix = hb.densities.bimixture(g["d_frac"], lp0, lp1)
    st_t = hb.densities.student_t(g["s_x"], g["s_mu"], g["s_scale"], g["s_nu"])
    lse = hb.tf_wraps.log_sum_exp(G.constant(g["lse_in"]), 1)
    assert np.allclose(m.run(lp0), g["d_logp0"], atol=1e-10)
    assert np.allclose(m.run(lp1), g["d_logp1"], atol=1e-10)
    assert np.allclose(m.run(mix), g["d_mix"], atol=1e-10)
    assert np.allclose(m.run(st_t), g["s_logp_nuT"], atol=1e-9)
    assert np.allclose(m.run(lse), g["lse_axis1"], atol=1e-10)
    rng = np.random.RandomState(0)
    A = rng.randn(4, 4)
    S = A @ A.T + 4 * np.eye(4)
    x, mu = rng.randn(4), rng.randn(4)
    ref = O.multivariate_normal(O.T(x), O.T(mu), O.T(np.linalg.cholesky(S))).item()
    got = m.run(hb.densities.multivariate_normal(x, mu, np.linalg.cholesky(S)))
    assert np.isclose(float(got), ref, rtol=1e-10)
    # fp32 path of the same ops at the reference's tolerance
    m32 = hb.model.Model(dtype="float32")
    assert np.allclose(m32.run(hb.densities.student_t(g["s_x"], g["s_mu"], g["s_scale"], 3.0)), g["s_logp_nu3"], atol=1e-5)


def test_sparse_gp_generic_branches_and_csym_on_device(golden):
    """3-D x, 'fullrank' residual and the CsymRBF kernel take the composed path
    (reference gp/gp.py:123-143,167-172; kernels.py:113-131) -- same values as the oracle."""
    np.random.seed(3)
    rng = np.random.RandomState(0)
    cfg = hb.settings.get_settings()
    cfg.numerics.jitter_level = 1e-4
    with hb.settings.temp_settings(cfg):
        for kern_cls, K, Kd in ((hb.gp.kernels.UnitRBF, O.rbf_K, None),
                                (hb.gp.kernels.UnitCsymRBF, O.csym_rbf_K, O.csym_rbf_Kdiag)):
            m = hb.model.Model(dtype="float64")
            z = np.linspace(-2.0, 2.0, 20).reshape(-1, 2)
            m.gp = hb.gp.SparseGP(z=z, kern=kern_cls(lengthscales=np.ones(1) * 0.8))
            m.u = hb.variationals.Normal(shape=[4, 10])
            x2, x3 = rng.randn(7, 2), rng.randn(4, 7, 2)
            uval = rng.randn(40)
            m.u.inject_noise(uval)
            e1, e3, ef = rng.randn(7), rng.randn(4, 7), rng.randn(4, 7)
            with m.tf_mode():
                d2 = m.gp.samples(x2, m.u, "diagonal", eps=e1)
                n3 = m.gp.samples(x3, m.u, "neglected")
                d3 = m.gp.samples(x3, m.u, "diagonal", eps=e3)
                f2 = m.gp.samples(x2, m.u, "fullrank", eps=ef)
                kxx = m.gp.kern.K(x3)
                kd = m.gp.kern.Kdiag(G.constant(x2))
            m.initialize()
            s = m._session
            ell = O.log1pe_forward(O.T(s.read_raw(m.gp.kern.lengthscales)))
            us = O.sample_diag(O.T(s.read_raw(m.u.q_mu)), O.T(s.read_raw(m.u.q_sqrt)), O.T(uval)).reshape(4, 10)
            zt = O.T(s.read_raw(m.gp.z))
            kdiag = None if Kd is None else (lambda xx: Kd(xx, ell))
            ref = lambda x, mode, e: O.sparse_samples(O.T(x), us, zt, ell, 1e-4, mode, None if e is None else O.T(e),
                                                      K=K, Kdiag=kdiag).numpy()
            assert np.allclose(m.run(d2), ref(x2, "diagonal", e1), atol=1e-8)
            assert np.allclose(m.run(n3), ref(x3, "neglected", None), atol=1e-8)
            assert np.allclose(m.run(d3), ref(x3, "diagonal", e3), atol=1e-8)
            assert np.allclose(m.run(f2), ref(x2, "fullrank", ef), atol=1e-7)
            assert np.allclose(m.run(kxx), K(O.T(x3), None, ell).numpy(), atol=1e-10)
            if Kd is not None:
                assert np.allclose(m.run(kd), Kd(O.T(x2), ell).numpy(), atol=1e-10)


def test_no_nan_with_small_jitter_fp32():
    # reference testing/test_gp.py:10-29: m = 600 random inducing points, n = 400, fp32, jitter 1e-5
    np.random.seed(0)
    rng = np.random.RandomState(0)
    m = hb.model.Model(dtype="float32")
    m.sparse_gp = hb.gp.SparseGP(z=np.sort(rng.uniform(-3, 3, (600, 1)), axis=0) * 10,
                                 kern=hb.gp.kernels.UnitRBF(lengthscales=np.ones(1)))
    m.u = hb.variationals.Normal(shape=[1, 600])
    x = rng.randn(400, 1) * 10
    with m.tf_mode():
        a = m.run(m.sparse_gp.samples(x, m.u, "neglected"))
        b = m.run(m.sparse_gp.samples(x, m.u, "diagonal"))
    assert a.shape == (1, 400) and not np.any(np.isnan(a)) and not np.any(np.isnan(b))


def test_expert_mixture_parity():
    """Sparse form of notebooks/Expert_GPR.ipynb: three independent GPs, sigmoid gate, x k_var."""
    np.random.seed(1)
    rng = np.random.RandomState(1)
    N, M, n = 1500, 40, 300
    X = np.sort(rng.uniform(0, 20, (N, 1)), axis=0)
    Y = np.where(X < 10, np.sin(3 * X), 0.3 * np.sin(0.5 * X)) + 0.1 * rng.randn(N, 1)
    Z = np.linspace(0, 20, M)[:, None]
    eps = rng.randn(N, 3)
    cfg = hb.settings.get_settings()
    cfg.numerics.jitter_level = 3e-4  # the notebook's setting (Expert_GPR.ipynb:203)
    with hb.settings.temp_settings(cfg):
        m = ExpertGPR(X=X, Y=Y, Z=Z, eps=eps, dtype="float64")
        noises = {g: rng.randn(M) for g in "slr"}
        m.u_s.inject_noise(noises["s"])
        m.u_l.inject_noise(noises["l"])
        m.u_r.inject_noise(noises["r"])
        idx = rng.randint(0, N, n)
        opt = m.ELBO()
        opt.compile()
        val, grads = opt.gradients(minibatch_size=n, indices=idx)
        s = m._session
        params = {"k_var_raw": O.T(s.read_raw(m.k_var)), "k_var_r_raw": O.T(s.read_raw(m.k_var_r)),
                  "var_raw": O.T(s.read_raw(m.var))}
        names = {"model.k_var": "k_var_raw", "model.k_var_r": "k_var_r_raw", "model.var": "var_raw"}
        for g in "slr":
            gp, u = getattr(m, "gp_" + g), getattr(m, "u_" + g)
            params["z_" + g] = O.T(s.read_raw(gp.z))
            params["ell_raw_" + g] = O.T(s.read_raw(gp.kern.lengthscales))
            params["q_mu_" + g] = O.T(s.read_raw(u.q_mu))
            params["q_sqrt_" + g] = O.T(s.read_raw(u.q_sqrt))
            names.update({"model.gp_%s.z" % g: "z_" + g, "model.gp_%s.kern.lengthscales" % g: "ell_raw_" + g,
                          "model.u_%s.q_mu" % g: "q_mu_" + g, "model.u_%s.q_sqrt" % g: "q_sqrt_" + g})
        fn = lambda p: O.expert_elbo(p, O.T(X[idx]), O.T(Y[idx]), float(N), {g: O.T(v) for g, v in noises.items()},
                                     O.T(eps[idx]), jitter=3e-4)
        ref_val, ref = O.grads_of(fn, params)
    assert abs(val - ref_val.item()) <= 1e-5 * abs(ref_val.item())
    for mine, theirs in names.items():
        assert rel_err(grads[mine], ref[theirs].numpy()) <= 1e-5, mine
    # and it trains with in-kernel noise
    for g in "slr":
        getattr(m, "u_" + g).inject_noise(None)
    m.eps = None


def test_gaussian_variational_and_variational_network_weights():
    """variationals.Gaussian (scale * Normal, reference variationals.py:232-291) and NeuralNet with
    Variational weights (nn.py:37,51-54): values match the oracle at injected noise; KL collects all."""
    np.random.seed(0)
    rng = np.random.RandomState(0)

    class M(hb.model.Model):
        def setUp(self):
            self.g = hb.variationals.Gaussian([3, 2], mean=2.0, stddev=0.5)
            self.net = hb.nn.NeuralNet([3, 4, 2], variable_types=hb.variationals.Normal, stddev=0.5)

        @hb.model.AutoOptimize()
        def obj(self):
            x = tf.constant(np.linspace(-1, 1, 15).reshape(5, 3))
            return tf.reduce_sum(tf.square(self.net(x))) + tf.reduce_sum(self.g) - self.KL()

    m = M(dtype="float64")
    noise = {}
    for v in (m.g, m.net.matbias0.w, m.net.matbias0.b, m.net.matbias1.w, m.net.matbias1.b):
        noise[v] = rng.randn(v.size)
        v.inject_noise(noise[v])
    val = m.obj().run()
    s = m._session
    m.initialize()

    def sample(v):
        mu, sq = O.T(s.read_raw(v.q_mu)), O.T(s.read_raw(v.q_sqrt))
        u = O.T(noise[v])
        x = O.sample_diag(mu, sq, u)
        return x, O.kl_normal(sq, u, x, "diagonal")

    xg, kl = sample(m.g)
    scale = O.log1pe_forward(O.T(s.read_raw(m.g.scale)))
    tot = torch.sum(scale * xg.reshape(3, 2))
    ws = []
    for mb in (m.net.matbias0, m.net.matbias1):
        xw, k1 = sample(mb.w)
        xb, k2 = sample(mb.b)
        kl = kl + k1 + k2
        ws.append((xw.reshape(mb.w._shape), xb.reshape(mb.b._shape)))
    x = O.T(np.linspace(-1, 1, 15).reshape(5, 3))
    y = O.neural_net(x, [w for w, _ in ws], [b for _, b in ws])
    ref = (torch.sum(y * y) + tot - kl).item()
    assert np.isclose(val, ref, rtol=1e-10)
    m.obj().compile(optimizer=tf.train.AdamOptimizer(0.01))
    for v in noise:
        v.inject_noise(None)
    m.obj().optimize(maxiter=20)  # runs with in-kernel noise for every variational leaf


def test_prediction_style_evaluation_and_heldout_objective():
    """Model.run(tensor) posterior draws at new inputs and run(training=False)."""
    np.random.seed(0)
    from henbun_amd.models import SVGP

    X, Y, Z = svgp_data(3000, 48, 0)
    m = SVGP(X=X, Y=Y, Z=Z, dtype="float64")
    m.ELBO().compile(optimizer=tf.train.AdamOptimizer(0.01))
    m.ELBO().optimize(maxiter=200, minibatch_size=512)
    xs = np.linspace(0, 24, 50)[:, None]
    with m.tf_mode():
        draws = np.stack([m.run(m.gp.samples(xs, m.u, "neglected") * tf.sqrt(m.k_var)) for _ in range(20)])
    assert draws.shape == (20, 1, 50) and np.all(np.isfinite(draws))
    assert draws.std(0).mean() > 0  # a fresh posterior draw per call
    assert np.mean((draws.mean(0)[0] - np.sin(xs[:, 0])) ** 2) < 0.3  # learned something about sin(x)
    held = [m.ELBO().run(minibatch_size=256, training=False) for _ in range(3)]
    assert np.all(np.isfinite(held))


def test_batched_experts_parity():
    """cfg-5 form: 2E independent sparse GPs as ONE expert-batched SparseGP (batched Gram with per-expert
    lengthscales, batched Cholesky / inverse / M^2 n contraction) == the oracle's loop over single GPs."""
    from henbun_amd.models import ExpertsGPR

    np.random.seed(2)
    rng = np.random.RandomState(2)
    N, M, n, E = 1200, 48, 256, 2
    X = np.sort(rng.uniform(0, 24, (N, 1)), axis=0)
    Y = np.sin(X) * (X < 12) + 0.2 * np.sin(4 * X) * (X >= 12) + 0.1 * rng.randn(N, 1)
    Z = np.linspace(0, 24, M)[:, None]
    ells = [0.5, 2.0, 1.0, 1.5]
    eps = rng.randn(N, 2 * E)
    cfg = hb.settings.get_settings()
    cfg.numerics.jitter_level = 1e-4
    with hb.settings.temp_settings(cfg):
        m = ExpertsGPR(X=X, Y=Y, Z=Z, ells=ells, eps=eps, dtype="float64")
        m.gp.z = np.stack([Z + 0.05 * rng.randn(M, 1) for _ in range(2 * E)])  # experts get their own inducing points
        u = rng.randn(2 * E * M)
        m.u.inject_noise(u)
        idx = rng.randint(0, N, n)
        opt = m.ELBO()
        opt.compile()
        val, grads = opt.gradients(minibatch_size=n, indices=idx)
        s = m._session
        params = {"z": O.T(s.read_raw(m.gp.z)), "ell_raw": O.T(s.read_raw(m.gp.kern.lengthscales)),
                  "q_mu": O.T(s.read_raw(m.u.q_mu)), "q_sqrt": O.T(s.read_raw(m.u.q_sqrt)),
                  "k_var_raw": O.T(s.read_raw(m.k_var)), "k_var_r_raw": O.T(s.read_raw(m.k_var_r)),
                  "var_raw": O.T(s.read_raw(m.var))}
        fn = lambda p: O.experts_elbo(p, O.T(X[idx]), O.T(Y[idx]), float(N), O.T(u), O.T(eps[idx]), jitter=1e-4)
        ref_val, ref = O.grads_of(fn, params)
    assert abs(val - ref_val.item()) <= 1e-5 * abs(ref_val.item()), (val, ref_val.item())
    names = {"model.gp.z": "z", "model.gp.kern.lengthscales": "ell_raw", "model.u.q_mu": "q_mu",
             "model.u.q_sqrt": "q_sqrt", "model.k_var": "k_var_raw", "model.k_var_r": "k_var_r_raw",
             "model.var": "var_raw"}
    for mine, theirs in names.items():
        assert rel_err(grads[mine], ref[theirs].numpy()) <= 1e-5, mine


def test_batched_experts_train_in_graph_mode():
    """Regression: a batch of 8 Cholesky `info` words must survive hipGraph replay (a captured
    32-byte hipMemsetAsync replayed garbage on ROCm 7.2; the reset now happens inside the panel kernel)."""
    from henbun_amd.models import ExpertsGPR

    np.random.seed(3)
    rng = np.random.RandomState(3)
    N, M, n, E = 2000, 64, 512, 4
    X = np.sort(rng.uniform(0, 32, (N, 1)), axis=0)
    Y = np.sin(X) + 0.1 * rng.randn(N, 1)
    Z = np.linspace(0, 32, M)[:, None]
    m = ExpertsGPR(X=X, Y=Y, Z=Z, ells=list(np.linspace(0.6, 1.2, E)) + list(np.linspace(0.8, 1.4, E)), dtype="float32")
    opt = m.ELBO()
    opt.compile(optimizer=tf.train.AdamOptimizer(1e-2))
    e0 = np.mean([opt.run(minibatch_size=n) for _ in range(4)])
    opt.optimize(maxiter=150, minibatch_size=n)
    e1 = np.mean([opt.run(minibatch_size=n) for _ in range(4)])
    assert np.isfinite(e1) and e1 > e0, (e0, e1)


def test_generic_kl_with_normal_prior_and_positive_transform():
    """Variational._KL, the generic Monte-Carlo KL (reference variationals.py:198-209): a Variational that is NOT
    the `Normal` class -- priors.Normal on the TRANSFORMED sample (priors.py:44-52) plus the log-Jacobian of a
    non-identity transform (transforms.py:136-137).  Value and every leaf gradient == the oracle at injected noise,
    diagonal and full-rank."""
    rng = np.random.RandomState(4)
    for q_shape in ("diagonal", "fullrank"):
        class M(hb.model.Model):
            def setUp(self):
                self.v = hb.variationals.Variational([7], q_shape=q_shape, prior=hb.priors.Normal(),
                                                     transform=hb.transforms.Log1pe())

            @hb.model.AutoOptimize()
            def obj(self):
                return tf.reduce_sum(tf.log(self.v)) * 0.7 - self.KL()

        np.random.seed(4)
        m = M(dtype="float64")
        if q_shape == "fullrank":
            m.v.q_sqrt = 0.5 * np.eye(7) + 0.1 * rng.randn(7, 7)
        u = rng.randn(7)
        m.v.inject_noise(u)
        opt = m.obj()
        opt.compile()
        val, grads = opt.gradients()
        s = m._session
        params = {"q_mu": O.T(s.read_raw(m.v.q_mu)), "q_sqrt": O.T(s.read_raw(m.v.q_sqrt))}

        def fn(p):
            x = (O.sample_diag if q_shape == "diagonal" else O.sample_fullrank)(p["q_mu"], p["q_sqrt"], O.T(u))
            kl = O.kl_generic(p["q_sqrt"], O.T(u), x, q_shape, prior_logp=O.prior_normal_logp,
                              transform=O.log1pe_forward, log_jacobian=O.log1pe_log_jacobian)
            return 0.7 * torch.sum(torch.log(O.log1pe_forward(x))) - kl

        ref_val, ref = O.grads_of(fn, params)
        assert abs(val - ref_val.item()) <= 1e-10 * abs(ref_val.item()), (q_shape, val, ref_val.item())
        assert rel_err(grads["model.v.q_mu"], ref["q_mu"].numpy()) <= 1e-9
        assert rel_err(grads["model.v.q_sqrt"], ref["q_sqrt"].numpy()) <= 1e-9
        # with the identity transform the generic form equals Normal._KL (variationals.py:213-230)
        x = O.sample_diag(params["q_mu"], params["q_sqrt"], O.T(u)) if q_shape == "diagonal" else None
        if x is not None:
            assert np.isclose(O.kl_generic(params["q_sqrt"], O.T(u), x, q_shape, prior_logp=O.prior_normal_logp).item(),
                              O.kl_normal(params["q_sqrt"], O.T(u), x, q_shape).item())


def test_offset_gaussian_value_and_gradients():
    """variationals.OffsetGaussian = scale * Normal + offset (reference variationals.py:293-314)."""
    np.random.seed(6)
    rng = np.random.RandomState(6)

    class M(hb.model.Model):
        def setUp(self):
            self.g = hb.variationals.OffsetGaussian([4, 3], mean=1.5, stddev=0.4)

        @hb.model.AutoOptimize()
        def obj(self):
            return tf.reduce_sum(tf.square(self.g - 1.0)) * -0.5 - self.KL()

    m = M(dtype="float64")
    u = rng.randn(12)
    m.g.inject_noise(u)
    opt = m.obj()
    opt.compile()
    val, grads = opt.gradients()
    s = m._session
    params = {"q_mu": O.T(s.read_raw(m.g.q_mu)), "q_sqrt": O.T(s.read_raw(m.g.q_sqrt)),
              "scale_raw": O.T(s.read_raw(m.g.scale)), "offset": O.T(s.read_raw(m.g.offset))}
    assert params["scale_raw"].shape == (1, 1) and params["offset"].shape == (1, 1)   # default shape [1]*rank

    def fn(p):
        x = O.sample_diag(p["q_mu"], p["q_sqrt"], O.T(u))
        g = O.log1pe_forward(p["scale_raw"]) * x.reshape(4, 3) + p["offset"]
        return -0.5 * torch.sum(torch.square(g - 1.0)) - O.kl_normal(p["q_sqrt"], O.T(u), x, "diagonal")

    ref_val, ref = O.grads_of(fn, params)
    assert abs(val - ref_val.item()) <= 1e-10 * abs(ref_val.item())
    for mine, theirs in (("model.g.q_mu", "q_mu"), ("model.g.q_sqrt", "q_sqrt"), ("model.g.scale", "scale_raw"),
                         ("model.g.offset", "offset")):
        assert rel_err(grads[mine], ref[theirs].numpy()) <= 1e-9, mine


@pytest.mark.parametrize("dtype,atol", [("float64", 1e-10), ("float32", 1e-4)])  # reference test_nn.py: atol 1e-4
def test_layered_neural_net_golden(golden, dtype, atol):
    """NeuralNet([3, 2, 4], n_layers=[5]) with layered weights w:[5,in,out], b:[5,1,out] (reference nn.py:10-32,
    testing/test_nn.py:11-29) against the golden chain `n_y`, on the device."""
    g = golden

    class M(hb.model.Model):
        def setUp(self):
            self.net = hb.nn.NeuralNet([3, 2, 4], n_layers=[5])
            self.x = hb.param.Data(g["n_x"])

        @hb.model.AutoOptimize()
        def obj(self):
            return tf.reduce_sum(self.net(self.x))

    m = M(dtype=dtype)
    m.net.matbias0.w, m.net.matbias0.b = g["n_w1"], g["n_b1"]
    m.net.matbias1.w, m.net.matbias1.b = g["n_w2"], g["n_b2"]
    with m.tf_mode():
        y = m.run(m.net(m.x))
    assert y.shape == (5, 6, 4)
    assert np.abs(y - g["n_y"]).max() <= atol
    assert np.isclose(m.obj().run(), g["n_y"].sum(), rtol=1e-4 if dtype == "float32" else 1e-10)


def test_closed_form_kl_mode_matches_the_reference_formula(golden):
    """kl_form='analytic' (north_star: "closed-form diagonal/full-covariance Gaussian KL"): the value equals the
    reference's own `gaussian_KL` (testing/test_variationals.py:326-347, executed into the golden fixture) on the
    reference's parameter draws; gradients equal the oracle's autograd of the same closed form; and the default
    Monte-Carlo estimator (variationals.py:225-230) averages to it (reference test: rtol 0.1 over 100 draws)."""
    g = golden
    for q_shape, sq_key, kl_key in (("diagonal", "v_sq_diag", "v_kl_diag"), ("fullrank", "v_sq_full", "v_kl_full")):
        class M(hb.model.Model):
            def setUp(self):
                self.v = hb.variationals.Normal(10, n_layers=[3], q_shape=q_shape, kl_form="analytic")
                self.w = hb.variationals.Normal(10, n_layers=[3], q_shape=q_shape)      # default: Monte-Carlo

            @hb.model.AutoOptimize()
            def kl(self):
                return object.__getattribute__(self, "v").KL()      # (in tf_mode `self.v` reads as a sample)

            @hb.model.AutoOptimize()
            def kl_mc(self):
                return object.__getattribute__(self, "w").KL()

        m = M(dtype="float64")
        for var in (m.v, m.w):
            var.q_mu = g["v_mu"]
            var.q_sqrt = g[sq_key]
        opt = m.kl()
        opt.compile()
        val, grads = opt.gradients()
        assert np.isclose(val, g[kl_key].item(), rtol=1e-12), (q_shape, val, g[kl_key].item())
        params = {"mu": O.T(g["v_mu"]), "sq": O.T(g[sq_key])}

        def fn(p):
            if q_shape == "diagonal":
                return 0.5 * torch.sum(torch.exp(2 * p["sq"]) - 2 * p["sq"] - 1 + p["mu"] ** 2)
            L = torch.tril(p["sq"])
            ld = torch.log(torch.diagonal(p["sq"], dim1=-2, dim2=-1) ** 2)
            return 0.5 * (torch.sum(L * L) - torch.sum(ld) - ld.numel() + torch.sum(p["mu"] ** 2))

        ref_val, ref = O.grads_of(fn, params)
        assert np.isclose(ref_val.item(), g[kl_key].item(), rtol=1e-12)
        assert rel_err(grads["model.v.q_mu"], ref["mu"].numpy()) <= 1e-10
        assert rel_err(grads["model.v.q_sqrt"], ref["sq"].numpy()) <= 1e-10
        mc = np.mean([m.kl_mc().run() for _ in range(100)])
        assert np.isclose(mc, g[kl_key].item(), rtol=0.1)
    # second parameter set, settings-driven mode (settings.numerics.kl_form), fp32
    cfg = hb.settings.get_settings()
    cfg.numerics.kl_form = "analytic"
    with hb.settings.temp_settings(cfg):
        class M2(hb.model.Model):
            def setUp(self):
                self.d = hb.variationals.Normal([1, 24])
                self.f = hb.variationals.Normal([1, 24], q_shape="fullrank")

            @hb.model.AutoOptimize()
            def kl(self):
                return self.KL()

        m2 = M2(dtype="float32")
        m2.d.q_mu, m2.d.q_sqrt = g["c_mu"][0], g["c_s_diag"][0]
        m2.f.q_mu, m2.f.q_sqrt = g["c_mu"][0], g["c_s_full"][0]
        assert np.isclose(m2.kl().run(), g["c_kl_diag"].item() + g["c_kl_full"].item(), rtol=1e-5)


@pytest.mark.parametrize("kern_name", ["UnitMatern32", "UnitMatern52"])
def test_matern_kernels_and_sparse_gp_through_them(kern_name):
    """Matern-3/2 and 5/2 on `euclid_dist` (reference gp/kernels.py:86-88 has the distance, no Matern class:
    builder extension, parity unpinned by the reference -- oracle formulas are checked against scikit-learn on the
    CPU).  K, Cholesky and a SparseGP ELBO gradient through the generic composition == the oracle."""
    rng = np.random.RandomState(8)
    oK = O.matern32_K if kern_name == "UnitMatern32" else O.matern52_K
    ell = np.array([0.7, 1.3])
    X, X2 = rng.randn(33, 2), rng.randn(20, 2)

    class KM(hb.model.Model):
        def setUp(self):
            self.k = getattr(hb.gp.kernels, kern_name)(ell.copy())

    km = KM(dtype="float64")
    with km.tf_mode():
        K1, K2 = km.run(km.k.K(X)), km.run(km.k.K(X, X2))
        L = km.run(km.k.Cholesky(X))
    assert np.abs(K1 - oK(O.T(X), None, O.T(ell)).numpy()).max() < 1e-12
    assert np.abs(K2 - oK(O.T(X), O.T(X2), O.T(ell)).numpy()).max() < 1e-12
    assert np.abs(L @ L.T - K1 - hb.settings.numerics.jitter_level * np.eye(33)).max() < 1e-10

    N, Mi, n = 600, 24, 128
    Xd, Yd, Z = svgp_data(N, Mi, 3)
    eps = rng.randn(N)

    class SV(hb.model.Model):
        def setUp(self):
            self.N = N
            self.X, self.Y = hb.param.MinibatchData(Xd), hb.param.MinibatchData(Yd)
            self.eps = hb.param.MinibatchData(eps)
            self.gp = hb.gp.SparseGP(kern=getattr(hb.gp.kernels, kern_name)(np.ones(1) * 0.9), z=Z)
            self.u = hb.variationals.Normal(shape=[1, Mi])
            self.var = hb.param.Variable([1], transform=hb.transforms.positive)

        @hb.model.AutoOptimize()
        def ELBO(self):
            f = self.gp.samples(self.X, self.u, q_shape="diagonal", eps=self.eps)
            ll = tf.reduce_sum(hb.densities.gaussian(tf.transpose(self.Y), f, self.var))
            return (self.N / tf.shape(self.X)[0]) * ll - self.KL()

    cfg = hb.settings.get_settings()
    cfg.numerics.jitter_level = 1e-4
    with hb.settings.temp_settings(cfg):
        np.random.seed(8)
        m = SV(dtype="float64")
        u = rng.randn(Mi)
        m.u.inject_noise(u)
        idx = rng.randint(0, N, n)
        opt = m.ELBO()
        opt.compile()
        val, grads = opt.gradients(minibatch_size=n, indices=idx)
        s = m._session
        params = {"z": O.T(s.read_raw(m.gp.z)), "ell_raw": O.T(s.read_raw(m.gp.kern.lengthscales)),
                  "q_mu": O.T(s.read_raw(m.u.q_mu)), "q_sqrt": O.T(s.read_raw(m.u.q_sqrt)),
                  "var_raw": O.T(s.read_raw(m.var))}

        def fn(p):
            x = O.sample_diag(p["q_mu"], p["q_sqrt"], O.T(u))
            f = O.sparse_samples(O.T(Xd[idx]), x.reshape(1, Mi), p["z"], O.log1pe_forward(p["ell_raw"]), 1e-4,
                                 "diagonal", O.T(eps[idx]), K=oK)
            ll = torch.sum(O.gaussian(O.T(Yd[idx]).T, f, O.log1pe_forward(p["var_raw"])))
            return (N / n) * ll - O.kl_normal(p["q_sqrt"], O.T(u), x, "diagonal")

        ref_val, ref = O.grads_of(fn, params)
    assert abs(val - ref_val.item()) <= 1e-8 * abs(ref_val.item())
    for mine, theirs in (("model.gp.z", "z"), ("model.gp.kern.lengthscales", "ell_raw"), ("model.u.q_mu", "q_mu"),
                         ("model.u.q_sqrt", "q_sqrt"), ("model.var", "var_raw")):
        assert rel_err(grads[mine], ref[theirs].numpy()) <= 1e-6, mine


def test_column_sums_survive_a_product_with_a_matutil_epilogue():
    """`X^T G` paired with `reduce_sum(G, 0)` (hb_matmul_colsum) whose square result also has a single matutil
    consumer takes the epilogue form of the GEMM -- the absorbed column sums must then still be computed."""
    rng = np.random.RandomState(3)
    Xh, Gh = rng.randn(256, 64), rng.randn(256, 64)
    for dtype in ("float64", "float32"):
        m = hb.model.Model(dtype=dtype)
        X, Gm = G.constant(Xh), G.constant(Gh)
        low = G.band_part(G.matmul(X, Gm, transpose_a=True), -1, 0)
        cs = G.reduce_sum(Gm, 0)
        w = rng.randn(64)
        val = G.add(G.reduce_sum(low), G.reduce_sum(G.mul(cs, G.constant(w))))
        ref = np.tril(Xh.T @ Gh).sum() + (Gh.sum(0) * w).sum()
        got = float(m.run(val))
        assert abs(got - ref) <= (1e-9 if dtype == "float64" else 2e-4) * max(1.0, abs(ref)), (dtype, got, ref)


def test_vec_to_tri_ops_on_device(golden):
    """hb_vec_to_tri / hb_tri_to_vec (the reference's disabled native op pair, tf_wraps.py:50-71) against the
    reference's own LowerTriangular.forward/backward output (golden lt_*), and each as the other's gradient."""
    g = golden
    m = hb.model.Model(dtype="float64")
    v = G.constant(g["lt_vec"])
    assert np.array_equal(m.run(hb.tf_wraps.vec_to_tri(v)), g["lt_tri"])
    assert np.array_equal(m.run(hb.tf_wraps.tri_to_vec(G.constant(g["lt_tri"]))), g["lt_back"])
    w = np.random.RandomState(0).randn(3, 4, 4)
    gr = G.gradients(G.reduce_sum(G.mul(hb.tf_wraps.vec_to_tri(v), G.constant(w))), [v])[0]
    assert np.allclose(m.run(gr), O.tri_to_vec(O.T(w)).numpy())
    H = m._session.H
    for dt in (torch.float32, torch.float64):          # ragged batch / sizes straight through the C ABI
        for B, N in ((1, 1), (5, 7), (2, 130)):
            a = torch.randn(B, N * (N + 1) // 2, dtype=dt, device="cuda")
            t = H.vec_to_tri(a)
            assert torch.equal(t, torch.as_tensor(O.vec_to_tri(a.cpu().double()).numpy()).to(dt).cuda())
            assert torch.equal(H.tri_to_vec(t), a)


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-9), ("float32", 2e-3)])
def test_tri_packed_fullrank_q_sqrt(dtype, tol):
    """Full-rank q(u) with q_sqrt stored as its packed lower triangle (tri_pack=True): the ELBO
    and every gradient equal the dense form's (and the oracle's), the gradient / parameter / all-reduce payload
    of q_sqrt is M(M+1)/2 instead of M^2, `.value` and assignment still speak dense matrices; also the closed-form
    KL and the generic logdet through the packed storage."""
    from henbun_amd.models import SVGP
    from henbun_amd.param import tri_pack, tri_unpack

    N, M, n = 2000, 96, 512
    rng = np.random.RandomState(9)
    X, Y, Z = svgp_data(N, M, 9)
    eps = rng.randn(N)
    S0 = 0.3 * np.eye(M) + 0.02 * rng.randn(M, M)
    u = rng.randn(M)
    idx = rng.randint(0, N, n)
    cfg = hb.settings.get_settings()
    cfg.numerics.jitter_level = 1e-4
    out = {}
    with hb.settings.temp_settings(cfg):
        for packed in (False, True):
            np.random.seed(9)

            class PSVGP(SVGP):
                def setUp(self, **kw):
                    SVGP.setUp(self, **kw)
                    self.u = hb.variationals.Normal(shape=[1, M], q_shape="fullrank", tri_pack=packed)

            m = PSVGP(X=X, Y=Y, Z=Z, q_shape="fullrank", eps=eps, dtype=dtype)
            m.gp.kern.lengthscales = np.ones(1) * 0.9
            m.u.q_sqrt = S0                      # dense assignment in both storages
            m.u.inject_noise(u)
            opt = m.ELBO()
            opt.compile()
            val, grads = opt.gradients(minibatch_size=n, indices=idx)
            out[packed] = (val, grads, m)
        (v0, g0, m0), (v1, g1, m1) = out[False], out[True]
        assert m1.u.packed and not m0.u.packed
        assert m1.u.q_sqrt._full_shape == [M * (M + 1) // 2]
        assert g1["model.u.q_sqrt"].shape == (M * (M + 1) // 2,)
        assert m1._session.theta.numel() == m0._session.theta.numel() - M * (M - 1) // 2
        assert np.allclose(m1.u.q_sqrt.value, np.tril(S0), atol=1e-6 if dtype == "float32" else 0)
        assert abs(v1 - v0) <= tol * abs(v0)
        for k in g0:
            a = tri_pack(g0[k].reshape(M, M)) if k == "model.u.q_sqrt" else g0[k]
            assert rel_err(g1[k], a) <= tol, k
        if dtype == "float64":
            s = m1._session
            params = {"z": O.T(s.read_raw(m1.gp.z)), "ell_raw": O.T(s.read_raw(m1.gp.kern.lengthscales)),
                      "q_mu": O.T(s.read_raw(m1.u.q_mu)).reshape(1, M), "q_sqrt": O.T(tri_unpack(s.read_raw(m1.u.q_sqrt))),
                      "k_var_raw": O.T(s.read_raw(m1.k_var)), "var_raw": O.T(s.read_raw(m1.var))}
            fn = lambda p: O.svgp_elbo(p, O.T(X[idx]), O.T(Y[idx]), float(N), O.T(u), O.T(eps[idx]), jitter=1e-4,
                                       q_shape="fullrank")
            ref_val, ref = O.grads_of(fn, params)
            assert abs(v1 - ref_val.item()) <= 1e-5 * abs(ref_val.item())
            assert rel_err(g1["model.u.q_sqrt"], tri_pack(ref["q_sqrt"].numpy())) <= 1e-5
            # closed-form KL and generic logdet through the packed storage
            m1.u.kl_form = "analytic"
            kl = float(m1.run(_kl_tensor(m1)))
            assert np.isclose(kl, O.gaussian_kl_analytic(params["q_mu"].numpy(), np.tril(S0)[None], "fullrank"), rtol=1e-10)
        # a few optimisation steps run (Adam on the packed slice) and keep the factor lower-triangular by construction
        m1.u.inject_noise(None)
        m1.u.kl_form = None
        o2 = m1.ELBO()
        o2.compile(optimizer=tf.train.AdamOptimizer(1e-3))
        o2.optimize(maxiter=3, minibatch_size=n)
        assert np.all(np.isfinite(m1.u.q_sqrt.value))


def _kl_tensor(m):
    with m.tf_mode():
        return m.KL()
