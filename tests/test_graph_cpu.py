"""Host logic on CPU: tracing through the Parameterized/tf_mode surface and the
graph-level autodiff, evaluated with the test-only CPU evaluator
(tests/graph_oracle.py) and compared with the oracle's torch-autograd gradients.
No HIP kernel runs here (that is tests/test_*_gpu.py)."""
import numpy as np
import pytest
import torch

import henbun_amd as hb
import henbun_oracle as O
from henbun_amd import graph as G

import graph_oracle as GO
from henbun_amd.models import SVGP, Amortised, DenseGPR

tf = hb.tf


def leaf_values(model, n=None):
    vals = {}
    for v in model.get_variables():
        if v.is_parameter:
            vals[v._leaf] = v._host_raw
        elif isinstance(v, hb.param.MinibatchData):
            for size, leaf in v._leaves.items():
                vals[leaf] = v.data[:size]
        elif isinstance(v, hb.param.Data):
            vals[v._tensor] = v.data
    return vals


def trace(model, method, minibatch=None):
    opt = method()
    return opt._trace(minibatch)


def raw(v):
    return O.T(v._host_raw)


@pytest.mark.parametrize("q_shape", ["diagonal", "fullrank"])
@pytest.mark.parametrize("residual", ["diagonal", "neglected"])
def test_svgp_elbo_and_gradients_match_oracle(q_shape, residual):
    np.random.seed(0)
    rng = np.random.RandomState(0)
    N, M = 50, 12
    X = rng.uniform(0, 6, (N, 1))
    Y = np.sin(X) + 0.3 * rng.randn(N, 1)
    Z = np.linspace(0, 6, M)[:, None]
    eps = rng.randn(N)
    m = SVGP(X=X, Y=Y, Z=Z, q_shape=q_shape, residual=residual, eps=eps)
    m.gp.kern.lengthscales = np.ones(1) * 0.9
    m.k_var = np.ones(1) * 1.3
    m.var = np.ones(1) * 0.4
    if q_shape == "fullrank":
        m.u.q_sqrt = 0.3 * np.eye(M) + 0.05 * rng.randn(M, M)
    u = rng.randn(M)
    m.u.inject_noise(u)
    cfg = hb.settings.get_settings()
    cfg.numerics.jitter_level = 1e-4
    with hb.settings.temp_settings(cfg):
        obj = trace(m, m.ELBO, None)
        leaves = [v._leaf for v in m.get_variables() if v.is_parameter]
        grads = G.gradients(obj, leaves)
    vals = GO.evaluate([obj] + [g for g in grads if g is not None], leaf_values(m))
    params = {
        "z": raw(m.gp.z), "ell_raw": raw(m.gp.kern.lengthscales), "q_mu": raw(m.u.q_mu).reshape(1, M),
        "q_sqrt": raw(m.u.q_sqrt), "k_var_raw": raw(m.k_var), "var_raw": raw(m.var),
    }
    fn = lambda p: O.svgp_elbo(p, O.T(X), O.T(Y), float(N), O.T(u), O.T(eps), jitter=1e-4, q_shape=q_shape,
                               residual=residual)
    val, ref = O.grads_of(fn, params)
    assert np.isclose(vals[obj].item(), val.item(), rtol=1e-10)
    got = {v.long_name: vals[g] for v, g in zip([v for v in m.get_variables() if v.is_parameter], grads)}
    pairs = [("model.gp.z", "z"), ("model.gp.kern.lengthscales", "ell_raw"), ("model.u.q_mu", "q_mu"),
             ("model.u.q_sqrt", "q_sqrt"), ("model.k_var", "k_var_raw"), ("model.var", "var_raw")]
    for mine, theirs in pairs:
        a, b = got[mine].numpy().reshape(-1), ref[theirs].numpy().reshape(-1)
        assert np.allclose(a, b, rtol=1e-7, atol=1e-8 * max(1.0, np.abs(b).max())), (mine, np.abs(a - b).max())


def test_amortised_elbo_and_gradients_match_oracle():
    np.random.seed(1)
    rng = np.random.RandomState(0)
    N, Din, H, L = 24, 6, 5, 3
    Y = rng.randn(N, Din)
    m = Amortised(Y=Y, L=L, H=H)
    u = rng.randn(N, L)
    m.z.inject_noise(u)
    obj = trace(m, m.ELBO, None)
    ps = [v for v in m.get_variables() if v.is_parameter]
    grads = G.gradients(obj, [v._leaf for v in ps])
    vals = GO.evaluate([obj] + grads, leaf_values(m))
    params = {
        "enc_w0": raw(m.enc.matbias0.w), "enc_b0": raw(m.enc.matbias0.b), "enc_w1": raw(m.enc.matbias1.w),
        "enc_b1": raw(m.enc.matbias1.b), "dec_w0": raw(m.dec.matbias0.w), "dec_b0": raw(m.dec.matbias0.b),
        "var_raw": raw(m.var),
    }
    val, ref = O.grads_of(lambda p: O.amortised_elbo(p, O.T(Y), O.T(u)), params)
    assert np.isclose(vals[obj].item(), val.item(), rtol=1e-10)
    names = {"model.enc.matbias0.w": "enc_w0", "model.enc.matbias0.b": "enc_b0", "model.enc.matbias1.w": "enc_w1",
             "model.enc.matbias1.b": "enc_b1", "model.dec.matbias0.w": "dec_w0", "model.dec.matbias0.b": "dec_b0",
             "model.var": "var_raw"}
    for v, g in zip(ps, grads):
        assert np.allclose(vals[g].numpy(), ref[names[v.long_name]].numpy(), rtol=1e-8, atol=1e-10), v.long_name
    # the local KL sums over every minibatch row (reference variationals.py:225-230)
    assert m.z.feed_size == 2 * L


def test_dense_gpr_gradients_through_cholesky():
    np.random.seed(2)
    rng = np.random.RandomState(0)
    n = 15
    X = np.sort(rng.uniform(0, 5, (n, 1)), axis=0)
    Y = np.sin(X) + 0.1 * rng.randn(n, 1)
    m = DenseGPR(X=X, Y=Y)
    m.q.q_sqrt = 0.5 * np.eye(n) + 0.05 * rng.randn(n, n)
    u = rng.randn(n)
    m.q.inject_noise(u)
    cfg = hb.settings.get_settings()
    cfg.numerics.jitter_level = 1e-3
    with hb.settings.temp_settings(cfg):
        obj = trace(m, m.ELBO)
        ps = [v for v in m.get_variables() if v.is_parameter]
        grads = G.gradients(obj, [v._leaf for v in ps])
    vals = GO.evaluate([obj] + grads, leaf_values(m))
    leaves = {v.long_name: raw(v).clone().requires_grad_(True) for v in ps}
    ell = O.log1pe_forward(leaves["model.kern.lengthscales"])
    L = O.kern_cholesky(O.T(X), ell, 1e-3)
    S = leaves["model.q.q_sqrt"]
    xs = O.sample_fullrank(leaves["model.q.q_mu"], S, O.T(u))
    kl = O.kl_normal(S, O.T(u), xs, "fullrank")
    f = (L @ xs.reshape(n, 1)) * torch.sqrt(O.log1pe_forward(leaves["model.k_var"]))
    elbo = torch.sum(O.gaussian(O.T(Y), f, O.log1pe_forward(leaves["model.var"]))) - kl
    ref = torch.autograd.grad(elbo, [leaves[v.long_name] for v in ps])
    assert np.isclose(vals[obj].item(), elbo.item(), rtol=1e-10)
    for v, g, r in zip(ps, grads, ref):
        assert np.allclose(vals[g].numpy(), r.numpy(), rtol=1e-6, atol=1e-8), v.long_name


def _check_vjp(build, shapes, seed=0, positive=False):
    """d sum(w*f(x...)) via graph autodiff vs torch autograd on the evaluator itself."""
    rng = np.random.RandomState(seed)
    arrs = [np.abs(rng.randn(*s)) + 0.5 if positive else rng.randn(*s) for s in shapes]
    leaves = [G.leaf("data", s, var=None) for s in shapes]
    y = build(*leaves)
    w = rng.randn(*y.shape)
    loss = G.reduce_sum(G.mul(y, G.constant(w)))
    grads = G.gradients(loss, leaves)
    vals = GO.evaluate([loss] + [g for g in grads if g is not None], dict(zip(leaves, arrs)))
    ts = [torch.as_tensor(a, dtype=torch.float64).requires_grad_(True) for a in arrs]
    vals2 = GO.evaluate([loss], dict(zip(leaves, ts)))  # evaluator is differentiable torch
    ref = torch.autograd.grad(vals2[loss], ts, allow_unused=True)
    for g, r, s in zip(grads, ref, shapes):
        if r is None:
            assert g is None or np.allclose(vals[g].numpy(), 0)
        else:
            assert g is not None and np.allclose(vals[g].numpy(), r.numpy(), rtol=1e-8, atol=1e-10)


def test_vjp_shape_ops():
    _check_vjp(lambda x: G.transpose(x, [2, 0, 1]), [(2, 3, 4)])
    _check_vjp(lambda x: x[1:, ..., :2], [(3, 2, 4)])
    _check_vjp(lambda x: x[0], [(3, 4)])
    _check_vjp(lambda x, y: G.concat([x, y], 1), [(2, 3), (2, 5)])
    _check_vjp(lambda x, y: G.stack([x, y], -1), [(2, 3), (2, 3)])
    _check_vjp(lambda x: G.tile(x, [2, 3]), [(2, 2)])
    _check_vjp(lambda x: G.broadcast_to(x, [4, 3, 5]), [(3, 1)])
    _check_vjp(lambda x: G.expand_dims(G.squeeze(x, 1), 0), [(3, 1, 2)])
    _check_vjp(lambda x: G.diag_part(x), [(2, 4, 4)])


def test_vjp_slices_that_tile_their_source_pack_into_a_concat():
    # mean / log-std halves of an encoder output (reference variationals.py:70-80): the summed slice gradients
    # are emitted as ONE concat, not zero-filled scatters added together -- and stay correct
    def halves(x):
        return G.unary("EXP", x[:, 3:]) * 2.0 + G.unary("TANH", x[:, :3])

    _check_vjp(halves, [(5, 6)])
    x = G.leaf("data", (5, 6), var=None)
    g = G.gradients(G.reduce_sum(halves(x)), [x])[0]
    assert g.node.op == "concat" and not any(n.op == "scatter_strided" for n in G.topo_order([g]))
    # three parts, given out of order; and a partial cover, which must NOT be packed
    _check_vjp(lambda x: G.reduce_sum(x[:, 4:] * 3.0, 1) + G.reduce_sum(x[:, :1], 1) * G.reduce_sum(G.unary("EXP", x[:, 1:4]), 1), [(4, 6)])
    y = G.leaf("data", (4, 6), var=None)
    g = G.gradients(G.reduce_sum(y[:, :2] * 2.0) + G.reduce_sum(y[:, 3:] * 3.0), [y])[0]
    assert g.node.op != "concat"
    _check_vjp(lambda x: G.reduce_sum(x[:, :2] * 2.0, 1) + G.reduce_sum(x[:, 3:] * 3.0, 1), [(4, 6)])


def test_mlp_backward_fuses_the_activation_gradient_into_the_dx_gemm():
    x, w1, b1, w2 = [G.leaf("data", s, var=None) for s in ((7, 4), (4, 5), (1, 5), (5, 3))]

    def net(x, w1, b1, w2):
        return G.matmul(G.matmul(x, w1, bias=b1, act="sigmoid"), w2)

    for act in ("sigmoid", "tanh", "relu"):
        _check_vjp(lambda x, w1, b1, w2, act=act: G.matmul(G.matmul(x, w1, bias=b1, act=act), w2), [(7, 4), (4, 5), (1, 5), (5, 3)])
    gs = G.gradients(G.reduce_sum(G.unary("SQUARE", net(x, w1, b1, w2))), [w1])
    ops = [n for n in G.topo_order(gs)]
    assert any(n.op == "matmul" and n.attrs.get("actgrad") == "sigmoid" for n in ops)
    assert not any(n.op == "ew" and n.attrs["f"] == "SIGMOID_GRAD" for n in ops)


def test_vjp_reductions_and_broadcast():
    _check_vjp(lambda x: G.reduce_sum(x, [0, 2]), [(3, 4, 5)])
    _check_vjp(lambda x: G.reduce_sum(x, 1, keepdims=True), [(3, 4, 5)])
    _check_vjp(lambda x: G.reduce_mean(x), [(3, 4)])
    _check_vjp(lambda x: G.reduce_max(x, -1), [(3, 4)])
    _check_vjp(lambda x, y: x * y + x / y - y, [(3, 1, 4), (2, 1)], positive=True)
    _check_vjp(lambda x: hb.tf_wraps.log_sum_exp(x, 1), [(3, 4, 2)])
    _check_vjp(lambda x, y: tf.maximum(x, y) + tf.minimum(x, y) * 2.0, [(3, 4), (4,)])


def test_vjp_elementwise():
    for f in ("EXP", "SQUARE", "SIGMOID", "TANH", "SOFTPLUS", "NEG", "ABS", "RELU"):
        _check_vjp(lambda x, f=f: G.unary(f, x), [(3, 4)])
    for f in ("LOG", "SQRT", "RECIP", "RSQRT", "LGAMMA", "LOG1P"):
        _check_vjp(lambda x, f=f: G.unary(f, x), [(3, 4)], positive=True)
    _check_vjp(lambda x: x ** 3.0 + 2.0 / x - (1.5 - x), [(5,)], positive=True)
    _check_vjp(lambda x, mu, v: hb.densities.gaussian(x, mu, v), [(4, 1), (1, 3), (1,)], positive=True)
    _check_vjp(lambda x, mu, s: hb.densities.student_t(x, mu, s, 3.0), [(4, 3), (3,), (1,)], positive=True)
    _check_vjp(lambda x: tf.clip_by_value(x, -0.5, 0.5), [(4, 4)])


def test_vjp_linalg():
    _check_vjp(lambda a, b: G.matmul(a, b), [(3, 4), (4, 5)])
    _check_vjp(lambda a, b: G.matmul(a, b, transpose_a=True, transpose_b=True), [(4, 3), (5, 4)])
    _check_vjp(lambda a, b: G.matmul(a, b), [(2, 3, 4), (4, 5)])
    _check_vjp(lambda a, b: G.matmul(a, b, transpose_b=True), [(3, 4), (2, 5, 4)])
    _check_vjp(lambda a, b, c: G.matmul(a, b, bias=c, act="sigmoid"), [(2, 3, 4), (2, 4, 5), (2, 1, 5)])
    _check_vjp(lambda a, b, c: G.matmul(a, b, bias=c, act="relu"), [(6, 4), (4, 5), (5,)])
    _check_vjp(lambda a: G.band_part(a, -1, 0), [(2, 4, 4)])

    def chol(a):
        spd = G.add_eye(G.matmul(a, a, transpose_b=True), 2.0)
        return G.cholesky(spd)

    _check_vjp(chol, [(4, 4)])
    _check_vjp(chol, [(2, 3, 3)])
    _check_vjp(lambda a: G.trinv(G.cholesky(G.add_eye(G.matmul(a, a, transpose_b=True), 2.0))), [(4, 4)])
    _check_vjp(lambda a, b: G.triangular_solve(G.cholesky(G.add_eye(G.matmul(a, a, transpose_b=True), 2.0)), b),
               [(4, 4), (4, 3)])
    _check_vjp(lambda x, x2, l: G.gram(x, x2, l, "rbf"), [(5, 2), (4, 2), (2,)], positive=True)
    _check_vjp(lambda x, l: G.gram(x, x, l, "csym_rbf"), [(3, 5, 2), (1,)], positive=True)
    _check_vjp(lambda x, x2, l: G.gram(x, x2, l, "sqdist"), [(3, 5, 2), (4, 2), (1,)], positive=True)


def test_sparse_gp_generic_composition_matches_fused_and_oracle():
    """3-D x (batched) and 'fullrank' take the composed path (reference gp/gp.py:123-143,167-172)."""
    np.random.seed(3)
    rng = np.random.RandomState(0)
    m = hb.model.Model()
    z = np.linspace(-2.0, 2.0, 20).reshape(-1, 2)
    m.gp = hb.gp.SparseGP(z=z, kern=hb.gp.kernels.UnitRBF(lengthscales=np.ones(1) * 0.8))
    m.u = hb.variationals.Normal(shape=[4, 10])
    x2 = rng.randn(7, 2)
    x3 = rng.randn(4, 7, 2)
    uval = rng.randn(40)
    m.u.inject_noise(uval)
    e1, e3, ef = rng.randn(7), rng.randn(4, 7), rng.randn(4, 7)
    cfg = hb.settings.get_settings()
    cfg.numerics.jitter_level = 1e-4
    with hb.settings.temp_settings(cfg), m.tf_mode():
        fused = m.gp.samples(x2, m.u, "diagonal", eps=e1)
        negl3 = m.gp.samples(x3, m.u, "neglected")
        diag3 = m.gp.samples(x3, m.u, "diagonal", eps=e3)
        full2 = m.gp.samples(x2, m.u, "fullrank", eps=ef)
        LT = m.gp._effective_LT(G.constant(z))
        cov_f = m.gp._additional_cov(G.constant(x2), m.gp._effective_LT(G.constant(x2)), "fullrank")
        cov_d = m.gp._additional_cov(G.constant(x2), m.gp._effective_LT(G.constant(x2)), "diagonal")
        cholT = G.matrix_transpose(m.gp.kern.Cholesky(G.constant(z)))
    vals = GO.evaluate([fused, negl3, diag3, full2, LT, cov_f, cov_d, cholT], leaf_values(m))
    ell = O.log1pe_forward(raw(m.gp.kern.lengthscales))
    us = O.sample_diag(raw(m.u.q_mu), raw(m.u.q_sqrt), O.T(uval)).reshape(4, 10)
    zt = raw(m.gp.z)
    ref = lambda x, mode, e: O.sparse_samples(O.T(x), us, zt, ell, 1e-4, mode, None if e is None else O.T(e))
    assert fused.shape == (4, 7) and negl3.shape == (4, 7) and full2.shape == (4, 7)  # test_gp.py:133-176
    assert np.allclose(vals[fused].numpy(), ref(x2, "diagonal", e1).numpy(), atol=1e-9)
    assert np.allclose(vals[negl3].numpy(), ref(x3, "neglected", None).numpy(), atol=1e-9)
    assert np.allclose(vals[diag3].numpy(), ref(x3, "diagonal", e3).numpy(), atol=1e-9)
    assert np.allclose(vals[full2].numpy(), ref(x2, "fullrank", ef).numpy(), atol=1e-8)
    # x == z  =>  effective L^T == chol(K)^T (test_gp.py:68-91); diag(full cov) == diagonal cov (test_gp.py:115-131)
    assert np.allclose(vals[LT].numpy(), vals[cholT].numpy(), atol=5e-3)
    assert np.allclose(np.diagonal(vals[cov_f].numpy()), vals[cov_d].numpy(), atol=1e-4)
