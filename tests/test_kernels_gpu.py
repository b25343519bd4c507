"""GPU parity tests: every HIP entry point (through the C ABI) vs the CPU oracle.

fp64 runs must match the oracle to ~1e-10 (north star: 1e-5 relative on fp64
inputs); fp32 runs are checked at fp32-appropriate tolerances.  Gradient
kernels are checked against torch autograd of the oracle's forward.
"""
import os
import numpy as np
import pytest
import torch

import henbun_oracle as O
from parity import observe, prod_err, tile_err

pytestmark = pytest.mark.gpu

DT = {"f32": torch.float32, "f64": torch.float64}
TOL = {"f32": dict(rtol=2e-4, atol=2e-5), "f64": dict(rtol=1e-9, atol=1e-10)}


@pytest.fixture(scope="module")
def H():
    from henbun_amd import hip_ops

    assert torch.cuda.is_available()
    return hip_ops


def dev(a, dt):
    return torch.as_tensor(np.asarray(a), dtype=dt).cuda().contiguous()


def host(t):
    return t.detach().cpu().double().numpy()


def assert_close(got, exp, tol, msg=""):
    got = host(got) if isinstance(got, torch.Tensor) else np.asarray(got)
    exp = host(exp) if isinstance(exp, torch.Tensor) else np.asarray(exp, dtype=np.float64)
    assert got.shape == exp.shape, (got.shape, exp.shape, msg)
    err = np.abs(got - exp)
    bound = tol["atol"] + tol["rtol"] * np.abs(exp)
    assert np.all(err <= bound), "%s max err %.3e (allowed %.3e)" % (msg, err.max(), bound.flat[np.argmax(err - bound)])


# ------------------------------------------------------------------ elementwise
@pytest.mark.parametrize("p", ["f32", "f64"])
def test_ewise_ops(H, p):
    dt, tol = DT[p], TOL[p]
    rng = np.random.RandomState(0)
    a = rng.randn(3, 1, 5)
    b = rng.randn(4, 1)
    pos = np.abs(rng.randn(3, 4, 5)) + 0.1
    A, B, POS = dev(a, dt), dev(b, dt), dev(pos, dt)
    ta, tb, tp = torch.as_tensor(a), torch.as_tensor(b), torch.as_tensor(pos)
    un = {
        "NEG": -ta, "EXP": ta.exp(), "SQUARE": ta * ta, "ABS": ta.abs(), "SIGN": ta.sign(),
        "SIGMOID": torch.sigmoid(ta), "RELU": torch.relu(ta), "SOFTPLUS": torch.nn.functional.softplus(ta),
        "TANH": torch.tanh(ta), "STEP": (ta > 0).double(), "COPY": ta,
    }
    for k, v in un.items():
        assert_close(H.ewise(k, [A]), v, tol, k)
    unp = {"LOG": tp.log(), "SQRT": tp.sqrt(), "RECIP": 1 / tp, "RSQRT": tp.rsqrt(), "LGAMMA": torch.lgamma(tp),
           "LOG1P": torch.log1p(tp), "DIGAMMA": torch.digamma(tp)}
    for k, v in unp.items():
        assert_close(H.ewise(k, [POS]), v, tol, k)
    assert_close(H.ewise("AFFINE", [A], params=[2.5, -1.0]), 2.5 * ta - 1.0, tol)
    assert_close(H.ewise("CLIP", [A], params=[-0.5, 0.7]), ta.clamp(-0.5, 0.7), tol)
    assert_close(H.ewise("POWC", [POS], params=[1.7]), tp ** 1.7, tol)
    bi = {"ADD": ta + tb, "SUB": ta - tb, "MUL": ta * tb, "DIV": ta / tb, "MAX": torch.maximum(ta, tb),
          "MIN": torch.minimum(ta, tb), "GT": (ta > tb).double(), "LE": (ta <= tb).double()}
    for k, v in bi.items():
        assert_close(H.ewise(k, [A, B]), v, tol, k)
    # fused Gaussian log-density and its 3-output gradient (reference densities.py:25-27)
    x, mu, var, g = rng.randn(6, 1), rng.randn(1, 7), np.abs(rng.randn(1)) + 0.3, rng.randn(6, 7)
    tx, tmu, tv = [torch.as_tensor(t, dtype=torch.float64).requires_grad_(True) for t in (x, mu, var)]
    lp = O.gaussian(tx, tmu, tv)
    assert_close(H.ewise("GAUSS_LOGPDF", [dev(x, dt), dev(mu, dt), dev(var, dt)]), lp, tol)
    gx, gmu, gv = H.ewise("GAUSS_LOGPDF_GRAD", [dev(x, dt), dev(mu, dt), dev(var, dt), dev(g, dt)], nout=3)
    ex = torch.autograd.grad(lp, [tx, tmu, tv], torch.as_tensor(g))
    # full-shape per-element grads: reduce on host to compare with the broadcast-summed autograd
    assert_close(host(gx).sum(1, keepdims=True), ex[0], TOL[p] if p == "f64" else dict(rtol=1e-3, atol=1e-4))
    assert_close(host(gmu).sum(0, keepdims=True), ex[1], TOL[p] if p == "f64" else dict(rtol=1e-3, atol=1e-4))
    assert_close(host(gv).sum().reshape(1), ex[2], TOL[p] if p == "f64" else dict(rtol=1e-3, atol=1e-3))


@pytest.mark.parametrize("p", ["f32", "f64"])
@pytest.mark.parametrize("dims", [(1, 7, 1), (1, 100000, 1), (5, 33, 1), (3, 17, 70), (1, 5000, 3), (2, 1, 9), (70, 300, 1)])
def test_reduce(H, p, dims):
    K1, R, K2 = dims
    dt = DT[p]
    x = np.random.RandomState(1).randn(K1, R, K2)
    tol = TOL[p] if p == "f64" else dict(rtol=1e-4, atol=1e-3)
    assert_close(H.reduce_mid(dev(x, dt), K1, R, K2).reshape(K1, K2), x.sum(1), tol)
    assert_close(H.reduce_mid(dev(x, dt), K1, R, K2, op=H.RED_MAX).reshape(K1, K2), x.max(1), tol)


def test_copy_nd_transpose_and_gather(H):
    x = np.random.RandomState(0).randn(3, 4, 5)
    X = dev(x, torch.float64)
    out = torch.empty(5, 3, 4, dtype=torch.float64, device="cuda")
    # out[k,i,j] = x[i,j,k]
    H.copy_nd(X, [1, 20, 5], out, [12, 4, 1], [5, 3, 4])
    assert_close(out, np.transpose(x, (2, 0, 1)), TOL["f64"])
    src = np.random.RandomState(1).randn(50, 3)
    perm = np.random.RandomState(2).permutation(50)
    idx = np.random.RandomState(3).randint(0, 50, 200)
    got = H.gather_rows(dev(src, torch.float32), torch.as_tensor(idx).cuda(), torch.as_tensor(perm).cuda())
    assert np.array_equal(host(got), src.astype(np.float32)[perm[idx]].astype(np.float64))
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    H.gather_rows(dev(src, torch.float32), torch.as_tensor([0, 50]).cuda(), None, err=err)
    assert err.item() == 1
    # empty
    assert H.gather_rows(dev(src, torch.float32), torch.empty(0, dtype=torch.int64, device="cuda")).shape == (0, 3)


def test_matutil(H):
    x = np.random.RandomState(0).randn(2, 5, 5)
    X = dev(x, torch.float64)
    assert_close(H.matutil(X, H.MATUTIL_BAND, lower=-1, upper=0), np.tril(x), TOL["f64"])
    assert_close(H.matutil(X, H.MATUTIL_ADD_EYE, alpha=0.25), x + 0.25 * np.eye(5), TOL["f64"])
    assert_close(H.matutil(X, H.MATUTIL_PHI), np.tril(x, -1) + 0.5 * np.stack([np.diag(np.diag(m)) for m in x]), TOL["f64"])
    assert_close(H.matutil(X, H.MATUTIL_SYM), 0.5 * (x + np.transpose(x, (0, 2, 1))), TOL["f64"])


# ------------------------------------------------------------------ RNG
def test_rng_statistics_and_determinism(H):
    r1 = H.Rng(seed=123, stream_id=0)
    a = host(r1.normal((200001,), torch.float32))
    assert abs(a.mean()) < 0.01 and abs(a.std() - 1) < 0.01
    assert abs(((a ** 3).mean())) < 0.03 and abs((a ** 4).mean() - 3) < 0.08
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 0.01
    assert np.abs(a).max() < 6.7 and (np.abs(a) > 4.0).sum() in range(3, 40)      # tails: 32-bit radius draw, P(|z| > 4) = 6.3e-5
    # same seed/stream -> same draws
    assert np.array_equal(a, host(H.Rng(seed=123, stream_id=0).normal((200001,), torch.float32)))
    # fp64 runs use the double Box-Muller (two generator steps per pair): a different, equally standard-normal sequence
    r2 = H.Rng(seed=123, stream_id=0)
    b = host(r2.normal((200001,), torch.float64))
    assert abs(b.mean()) < 0.01 and abs(b.std() - 1) < 0.01 and abs((b ** 4).mean() - 3) < 0.08
    assert not np.allclose(a[:1000], b[:1000], atol=1e-3)
    # different stream id -> different draws; consecutive draws differ
    r3 = H.Rng(seed=123, stream_id=1)
    c = host(r3.normal((1000,), torch.float64))
    assert not np.allclose(b[:1000], c)
    d = host(r2.normal((1000,), torch.float64))
    assert not np.allclose(b[:1000], d)
    idx = host(H.Rng(seed=5).randint(100000, 10, 20))
    assert idx.min() == 10 and idx.max() == 19
    assert abs(np.bincount(idx.astype(int))[10:].std() / 10000) < 0.02


# ------------------------------------------------------------------ K1/K2
@pytest.mark.parametrize("p", ["f32", "f64"])
@pytest.mark.parametrize("n", [1, 10, 513, 70001])
def test_diag_sample_kl(H, p, n):
    dt, tol = DT[p], TOL[p]
    rng = np.random.RandomState(0)
    mu, s, u = 0.3 * rng.randn(n), -0.5 + 0.3 * rng.randn(n), rng.randn(n)
    xbar, klbar = rng.randn(n), np.array([0.7])
    tm, ts = [torch.as_tensor(t).requires_grad_(True) for t in (mu, s)]
    tx = O.sample_diag(tm, ts, torch.as_tensor(u))
    tkl = O.kl_normal(ts, torch.as_tensor(u), tx, "diagonal")
    x, kl, uo = H.diag_sample_kl_fwd(dev(mu, dt), dev(s, dt), u_in=dev(u, dt))
    assert_close(x, tx, tol)
    kltol = tol if p == "f64" else dict(rtol=1e-4, atol=1e-3 * max(1, n / 1000))
    assert_close(kl, tkl.reshape(1), kltol)
    assert_close(uo, u, tol)
    loss = (tx * torch.as_tensor(xbar)).sum() + 0.7 * tkl
    gm, gs = torch.autograd.grad(loss, [tm, ts])
    mb, sb = H.diag_sample_kl_bwd(dev(s, dt), dev(u, dt), x, dev(xbar, dt), dev(klbar, dt))
    assert_close(mb, gm, tol)
    assert_close(sb, gs, tol if p == "f64" else dict(rtol=1e-3, atol=1e-4))


def test_diag_sample_kl_golden_and_rng(H, golden):
    g = golden
    x, kl, _ = H.diag_sample_kl_fwd(dev(g["v_mu"], torch.float64), dev(g["v_sq_diag"], torch.float64),
                                    u_in=dev(g["v_iid"], torch.float64))
    assert_close(x, g["v_post_diag"], TOL["f64"])  # reference testing/test_variationals.py:85-106
    # in-kernel noise: exported u reproduces x and kl exactly through the oracle; MC mean -> analytic KL
    rng = H.Rng(seed=7)
    mu, sq = dev(g["v_mu"], torch.float64), dev(g["v_sq_diag"], torch.float64)
    acc = 0.0
    for _ in range(300):
        x, kl, u = H.diag_sample_kl_fwd(mu, sq, rng=rng)
        acc += kl.item()
    tx = O.sample_diag(O.T(g["v_mu"]), O.T(g["v_sq_diag"]), O.T(host(u)))
    assert_close(x, tx, TOL["f64"])
    assert np.isclose(kl.item(), O.kl_normal(O.T(g["v_sq_diag"]), O.T(host(u)), tx, "diagonal").item(), rtol=1e-10)
    assert np.isclose(acc / 300, g["v_kl_diag"], rtol=0.1)  # reference test_variationals.py:108-122 (rtol 0.1)


@pytest.mark.parametrize("p", ["f32", "f64"])
@pytest.mark.parametrize("rows,size", [(3, 10), (1, 1), (1, 200), (257, 4)])
def test_fullrank_sample_kl(H, p, rows, size):
    dt, tol = DT[p], TOL[p]
    rng = np.random.RandomState(0)
    mu = 0.3 * rng.randn(rows, size)
    S = 0.2 * rng.randn(rows, size, size) + np.eye(size)
    u = rng.randn(rows, size)
    xbar = rng.randn(rows, size)
    tm, tS = [torch.as_tensor(t).requires_grad_(True) for t in (mu, S)]
    tx = O.sample_fullrank(tm, tS, torch.as_tensor(u))
    tkl = O.kl_normal(tS, torch.as_tensor(u), tx, "fullrank")
    x, kl, _ = H.fullrank_sample_kl_fwd(dev(mu, dt), dev(S, dt), u_in=dev(u, dt))
    ftol = tol if p == "f64" else dict(rtol=1e-3, atol=1e-4)
    assert_close(x, tx, ftol)
    assert_close(kl, tkl.reshape(1), tol if p == "f64" else dict(rtol=1e-3, atol=1e-2))
    loss = (tx * torch.as_tensor(xbar)).sum() + 1.3 * tkl
    gm, gS = torch.autograd.grad(loss, [tm, tS])
    mb, Sb = H.fullrank_sample_kl_bwd(dev(S, dt), dev(u, dt), x, dev(xbar, dt), dev(np.array([1.3]), dt))
    assert_close(mb, gm, ftol)
    assert_close(Sb, gS, ftol)
    assert np.all(np.triu(host(Sb), 1) == 0)  # masked entries get zero gradient (variationals.py:145)


@pytest.mark.parametrize("p", ["f32", "f64"])
@pytest.mark.parametrize("rows,size,packed", [(1, 1024, False), (1, 1024, True), (3, 200, False), (8, 65, True), (1, 1023, False)])
def test_fullrank_sampler_in_one_launch(H, p, rows, size, packed):
    """hb_fullrank_sample_kl_fwd1: noise, sample and KL of a full-rank block (reference variationals.py:138-146,225-230) in ONE
    launch -- every workgroup draws u itself from the unchanged generator states, the last one to arrive folds the KL and
    advances the generator -- against the three-launch form (fill u, rows, fold): the same variates bit for bit, x and kl
    to rounding, the same generator state afterwards (drawn twice in a row), injected noise as well; and against the oracle."""
    dt = DT[p]
    rng = np.random.RandomState(size + rows)
    mu = 0.3 * rng.randn(rows, size)
    S = np.tril(0.05 * rng.randn(rows, size, size)) + np.eye(size)
    Sd = dev(S, dt)
    if packed:
        il = np.tril_indices(size)
        Sd = dev(S[:, il[0], il[1]], dt)
    assert H._lib.lib().raw("hb_fullrank_one_launch_shape")(rows, size) == 1
    res = {}
    for three in (0, 1):
        H.debug_set("fullrank_three_launches", three)
        try:
            g = H.Rng(123)
            a = H.fullrank_sample_kl_fwd(dev(mu, dt), Sd, rng=g, packed=packed)
            b = H.fullrank_sample_kl_fwd(dev(mu, dt), Sd, rng=g, packed=packed)      # second draw: the advanced states
            c = H.fullrank_sample_kl_fwd(dev(mu, dt), Sd, u_in=a[2].clone(), packed=packed)
            torch.cuda.synchronize()
            res[three] = [t.clone() for t in a + b + c]
        finally:
            H.debug_clear()
    one, thr = res[0], res[1]
    for i in (2, 5):                 # u of the first and of the second draw: the same variates from the same states
        assert torch.equal(one[i], thr[i]), i
    for i in (0, 3, 6):              # x (drawn, drawn again, injected): the same sums up to the contraction of a multiply-add
        assert float((one[i] - thr[i]).abs().max()) <= (1e-13 if p == "f64" else 2e-6) * float(thr[i].abs().max()), i
    assert not torch.equal(one[2], one[5])                       # the second draw is a new one
    for i in (1, 4, 7):              # kl: another (fixed) order of the partial sums
        assert abs(float(one[i]) - float(thr[i])) <= (1e-12 if p == "f64" else 2e-6) * max(abs(float(thr[i])), float(rows * size))
    tx = O.sample_fullrank(O.T(mu), O.T(S), O.T(host(one[2])))
    tkl = O.kl_normal(O.T(S), O.T(host(one[2])), tx, "fullrank")
    assert_close(one[0], tx, TOL["f64"] if p == "f64" else dict(rtol=1e-4, atol=1e-4))
    assert abs(float(one[1]) - float(tkl)) <= (1e-10 if p == "f64" else 1e-4) * max(abs(float(tkl)), 1.0)


def test_fullrank_golden(H, golden):
    g = golden
    x, kl, _ = H.fullrank_sample_kl_fwd(dev(g["v_mu"], torch.float64), dev(g["v_sq_full"], torch.float64),
                                        u_in=dev(g["v_iid"], torch.float64))
    assert_close(x, g["v_post_full"], TOL["f64"])


# ------------------------------------------------------------------ K3
@pytest.mark.parametrize("p", ["f32", "f64"])
def test_gram_golden(H, golden, p):
    # reference testing/test_kernels.py:90-182 (atol 1e-4)
    g, dt = golden, DT[p]
    tol = TOL[p]
    X, X2, Xb, X2b = [dev(g[k], dt) for k in ("k_X", "k_X2", "k_Xb", "k_X2b")]
    l1, l2 = dev(g["k_l1"], dt), dev(g["k_l2"], dt)
    assert_close(H.gram_fwd(X, X, l1), g["k_rbf1_XX"], tol)
    assert_close(H.gram_fwd(X, X, l2), g["k_rbf2_XX"], tol)
    assert_close(H.gram_fwd(X, X, l1, kind=H.KERN_CSYM_RBF), g["k_csym_XX"], tol)
    assert_close(H.gram_fwd(X, X2, l1), g["k_rbf1_XX2"], tol)
    assert_close(H.gram_fwd(X, X2, l2), g["k_rbf2_XX2"], tol)
    assert_close(H.gram_fwd(X, X2, l1, kind=H.KERN_CSYM_RBF), g["k_csym_XX2"], tol)
    assert_close(H.gram_fwd(Xb, Xb, l1), g["k_rbf1_b"], tol)
    assert_close(H.gram_fwd(Xb, X2b, l2), g["k_rbf2_b2"], tol)
    assert_close(H.gram_fwd(Xb, X2b, l1, kind=H.KERN_CSYM_RBF), g["k_csym_b2"], tol)


@pytest.mark.parametrize("kind", ["rbf", "csym"])
@pytest.mark.parametrize("ard", [False, True])
def test_gram_bwd(H, kind, ard):
    rng = np.random.RandomState(0)
    B, n, n2, d = 3, 7, 9, 2
    X, X2 = rng.randn(B, n, d), rng.randn(B, n2, d)
    ell = np.exp(0.3 * rng.randn(d if ard else 1))
    Kbar = rng.randn(B, n, n2)
    tX, tX2, tl = [torch.as_tensor(t).requires_grad_(True) for t in (X, X2, ell)]
    K = (O.rbf_K if kind == "rbf" else O.csym_rbf_K)(tX, tX2, tl)
    gX, gX2, gl = torch.autograd.grad((K * torch.as_tensor(Kbar)).sum(), [tX, tX2, tl])
    dt = torch.float64
    k = H.KERN_RBF if kind == "rbf" else H.KERN_CSYM_RBF
    xb, x2b, lb = H.gram_bwd(dev(X, dt), dev(X2, dt), dev(ell, dt), dev(Kbar, dt), kind=k)
    assert_close(xb, gX, TOL["f64"])
    assert_close(x2b, gX2, TOL["f64"])
    assert_close(lb, gl, TOL["f64"])
    # shared (2-D) second operand: gradient summed over the batch
    tZ = torch.as_tensor(X2[0]).requires_grad_(True)
    K = (O.rbf_K if kind == "rbf" else O.csym_rbf_K)(tX, tZ[None].expand(B, -1, -1), tl)
    gZ = torch.autograd.grad((K * torch.as_tensor(Kbar)).sum(), [tZ])[0]
    _, zb, _ = H.gram_bwd(dev(X, dt), dev(X2[0], dt), dev(ell, dt), dev(Kbar, dt), kind=k)
    assert_close(zb, gZ, TOL["f64"])


@pytest.mark.parametrize("kind", ["rbf", "csym"])
@pytest.mark.parametrize("ard", [False, True])
def test_gram_per_batch_lengthscales(H, kind, ard):
    """ell [B, dl]: one independent kernel per batch entry (expert-batched Gram) == a loop of singles."""
    rng = np.random.RandomState(1)
    B, n, n2, d = 4, 11, 6, 2
    X, X2 = rng.randn(B, n, d), rng.randn(n2, d)
    ell = np.exp(0.3 * rng.randn(B, d if ard else 1))
    Kbar = rng.randn(B, n, n2)
    tX, tX2, tl = [torch.as_tensor(t).requires_grad_(True) for t in (X, X2, ell)]
    f = O.rbf_K if kind == "rbf" else O.csym_rbf_K
    K = torch.stack([f(tX[b], tX2, tl[b]) for b in range(B)])
    gX, gX2, gl = torch.autograd.grad((K * torch.as_tensor(Kbar)).sum(), [tX, tX2, tl])
    dt = torch.float64
    k = H.KERN_RBF if kind == "rbf" else H.KERN_CSYM_RBF
    assert_close(H.gram_fwd(dev(X, dt), dev(X2, dt), dev(ell, dt), kind=k), K.detach(), TOL["f64"])
    xb, x2b, lb = H.gram_bwd(dev(X, dt), dev(X2, dt), dev(ell, dt), dev(Kbar, dt), kind=k)
    assert_close(xb, gX, TOL["f64"])
    assert_close(x2b, gX2, TOL["f64"])
    assert_close(lb, gl, TOL["f64"])
    # both operands shared, only the kernels differ (the K(z_e, x) of experts on common inputs)
    K2 = torch.stack([f(tX[0], tX2, tl[b]) for b in range(B)])
    assert_close(H.gram_fwd(dev(X[0], dt), dev(X2, dt), dev(ell, dt), kind=k), K2.detach(), TOL["f64"])


# ------------------------------------------------------------------ matmul
@pytest.mark.parametrize("p", ["f32", "f64"])
@pytest.mark.parametrize("tA,tB", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("shape", [(1, 1, 1), (5, 7, 3), (64, 64, 16), (65, 130, 33), (200, 96, 257)])
def test_matmul_basic(H, p, tA, tB, shape):
    m, n, k = shape
    dt = DT[p]
    rng = np.random.RandomState(0)
    a = rng.randn(*((k, m) if tA else (m, k)))
    b = rng.randn(*((n, k) if tB else (k, n)))
    exp = (a.T if tA else a) @ (b.T if tB else b)
    tol = TOL[p] if p == "f64" else dict(rtol=1e-4, atol=1e-4 * np.sqrt(k))
    assert_close(H.matmul(dev(a, dt), dev(b, dt), transA=tA, transB=tB), exp, tol)


@pytest.mark.parametrize("p", ["f32", "f64"])
def test_matmul_activation_gradient_epilogue(H, p):
    """dx GEMM with the activation derivative as its epilogue (direct, ragged, batched and split-K finish) ==
    the two-pass form: the backward of the reference's MatBias + activation layer (nn.py:31-32,79-84)."""
    dt = DT[p]
    rng = np.random.RandomState(3)
    tol = TOL[p] if p == "f64" else dict(rtol=2e-4, atol=2e-3)
    dact = {"sigmoid": lambda y: y * (1 - y), "tanh": lambda y: 1 - y * y, "relu": lambda y: (y > 0).astype(y.dtype)}
    for act in ("sigmoid", "tanh", "relu"):
        for (m, n, k) in ((300, 70, 33), (256, 256, 32), (7, 5, 3)):
            g, w = rng.randn(m, k), rng.randn(n, k)
            y = rng.rand(m, n) if act != "relu" else np.maximum(rng.randn(m, n), 0)
            got = H.matmul(dev(g, dt), dev(w, dt), transB=True, act=act, actgrad=dev(y, dt))
            assert_close(got, (g @ w.T) * dact[act](y), tol)
    # batched, and the split-K finish path (few tiles, long contraction)
    g, w, y = rng.randn(3, 40, 9), rng.randn(3, 9, 20), rng.rand(3, 40, 20)
    assert_close(H.matmul(dev(g, dt), dev(w, dt), act="sigmoid", actgrad=dev(y, dt), alpha=0.5), 0.5 * (g @ w) * y * (1 - y), tol)
    g, w, y = rng.randn(30, 6000), rng.randn(6000, 50), rng.rand(30, 50)
    assert_close(H.matmul(dev(g, dt), dev(w, dt), act="tanh", actgrad=dev(y, dt)), (g @ w) * (1 - y * y),
                 TOL[p] if p == "f64" else dict(rtol=1e-3, atol=2e-2))
    with pytest.raises(ValueError):
        H.matmul(dev(g, dt), dev(w, dt), act="tanh", actgrad=dev(y, dt), bias=dev(rng.randn(50), dt))


@pytest.mark.parametrize("tB", [False, True])
def test_matmul_row_streaming_form_for_tall_operands(H, tB):
    """fp32 products of a TALL A (a minibatch of rows) with a small op(B) (a MatBias layer's weights; reference
    nn.py:31-32, cfg 4's [32768, 64] x [64, 256] ...) run in the row-streaming form (matmul_rows_kernel: weights resident
    in LDS, A read straight into MFMA fragments): every k-group width (K % 32 / 16 / 8), every accumulator count
    (N <= 32 / 64 / 128 / 256), ragged M and N, bias + activation and the activation-gradient epilogue, against numpy in
    fp64."""
    dt = torch.float32
    rng = np.random.RandomState(11)
    for m, k, n in ((4096, 64, 256), (2500, 256, 32), (3001, 16, 64), (2048, 64, 16), (2304, 32, 250), (2177, 24, 100),
                    (4099, 8, 7), (2048, 40, 128)):
        a = rng.randn(m, k)
        b = rng.randn(*((n, k) if tB else (k, n)))
        full = a @ (b.T if tB else b)
        A_, B_ = dev(a, dt), dev(b, dt)
        tol = dict(rtol=1e-5, atol=2e-5 * np.sqrt(k))    # observed <= 4e-6 sqrt(k)
        got = H.matmul(A_, B_, transB=tB, alpha=0.5)
        assert_close(got, 0.5 * full, tol)
        bias = rng.randn(n)
        assert_close(H.matmul(A_, B_, transB=tB, bias=dev(bias, dt), act="sigmoid", alpha=0.1),
                     1 / (1 + np.exp(-(0.1 * full + bias))), dict(rtol=1e-5, atol=2e-6))
        y = rng.rand(m, n)
        assert_close(H.matmul(A_, B_, transB=tB, act="sigmoid", actgrad=dev(y, dt)), full * y * (1 - y), tol)
        out = torch.full((m, n), float("nan"), dtype=dt, device="cuda")
        H.matmul(A_, B_, transB=tB, out=out)
        assert not torch.isnan(out).any()
    # A = I padded with zero rows and an asymmetric B: the k permutation of the two operand forms and the row / column map
    bb = (np.arange(64 * 96, dtype=np.float64).reshape(64, 96) % 251) - 100
    eye = np.zeros((2048, 64)); eye[:64] = np.eye(64)
    got = host(H.matmul(dev(eye, dt), dev(bb.T.copy() if tB else bb, dt), transB=tB))
    assert np.array_equal(got[:64], bb) and not got[64:].any()


def test_matmul_minibatch_deep_weight_gradient_form(H):
    """fp32 A^T B with a minibatch-deep contraction into a small result (the weight gradient x^T g of a MatBias layer,
    reference nn.py:31-32 under TF autodiff; cfg 4: 32768 rows into 64 x 256, 256 x 32, 16 x 64): slabs of rows folded by
    the split-K finish (sixteen slab loads in flight per round), with and without the column sums of B (the bias gradient)
    riding along: ragged row counts, ragged and sub-tile result shapes, alpha."""
    dt = torch.float32
    rng = np.random.RandomState(21)
    for k, m, n in ((32768, 64, 256), (8192, 256, 32), (4096, 16, 64), (5000, 40, 70), (4100, 7, 5), (6001, 33, 256)):
        a, b = rng.randn(k, m), rng.randn(k, n)
        full = a.T @ b
        tol = dict(rtol=2e-5, atol=4e-6 * k ** 0.5 * 4)     # sums of k products of unit normals: sd sqrt(k); observed <= 1e-6 sqrt(k)
        assert_close(H.matmul(dev(a, dt), dev(b, dt), transA=True, alpha=0.25), 0.25 * full, tol)
        got, cs = H.matmul_colsum(dev(a, dt), dev(b, dt))
        assert_close(got, full, tol)
        assert_close(cs, b.sum(0), tol)
    # an identity block in A picks rows of B exactly (row / column maps of both operand forms, slab boundaries)
    k, m, n = 4096, 64, 96
    a = np.zeros((k, m)); idx = rng.permutation(k)[:m]; a[idx, np.arange(m)] = 1.0
    b = (np.arange(k * n, dtype=np.float64).reshape(k, n) % 1021) - 500
    assert np.array_equal(host(H.matmul(dev(a, dt), dev(b, dt), transA=True)), b[idx])


@pytest.mark.parametrize("tA,tB", [(False, False), (False, True), (True, False), (True, True)])
def test_matmul_in_workgroup_split_k_small_gemms(H, tA, tB):
    """fp32 products with few output tiles (the M^3 GEMMs of the Cholesky VJP: 512^3 at cfg 2, 8 x 512^3 at cfg 5)
    run as one 32x32 tile per workgroup with the contraction split over its four waves and every epilogue applied in
    the workgroup (matmul_wgk_kernel): all transposes, batch, bias + activation, beta, tril / Phi / symmetrise."""
    dt = torch.float32
    rng = np.random.RandomState(7)
    for batch, m, n, k in ((1, 512, 512, 512), (3, 64, 96, 128), (1, 128, 32, 256), (8, 128, 128, 384)):
        a = rng.randn(batch, *((k, m) if tA else (m, k)))
        b = rng.randn(batch, *((n, k) if tB else (k, n)))
        full = np.einsum("bij,bjk->bik", np.transpose(a, (0, 2, 1)) if tA else a, np.transpose(b, (0, 2, 1)) if tB else b)
        A_, B_ = dev(a if batch > 1 else a[0], dt), dev(b if batch > 1 else b[0], dt)
        sq = (lambda x: x if batch > 1 else x[0])
        tol = dict(rtol=1e-4, atol=2e-4 * np.sqrt(k))
        assert_close(H.matmul(A_, B_, transA=tA, transB=tB, alpha=-0.5), sq(-0.5 * full), tol)
        bias = rng.randn(n)
        assert_close(H.matmul(A_, B_, transA=tA, transB=tB, bias=dev(bias, dt), act="tanh", alpha=0.01),
                     sq(np.tanh(0.01 * full + bias)), dict(rtol=1e-4, atol=1e-5))
        c0 = rng.randn(*full.shape)
        out = dev(sq(c0), dt)
        H.matmul(A_, B_, transA=tA, transB=tB, out=out, beta=2.0)
        assert_close(out, sq(full + 2.0 * c0), tol)
        if m == n:
            got = host(H.matmul(A_, B_, transA=tA, transB=tB, tril_out=True))
            assert np.all(np.triu(got, 1) == 0) and np.allclose(got, sq(np.tril(full)), **tol)
            phi = np.tril(full, -1) + 0.5 * np.einsum("bii->bi", full)[:, :, None] * np.eye(m)
            got = host(H.matmul(A_, B_, transA=tA, transB=tB, epilogue=H.MM_PHI_OUT))
            assert np.all(np.triu(got, 1) == 0) and np.allclose(got, sq(phi), **tol)
            got = host(H.matmul(A_, B_, transA=tA, transB=tB, epilogue=H.MM_SYM_OUT))
            assert np.allclose(got, sq(0.5 * (full + np.transpose(full, (0, 2, 1)))), **tol)
            assert np.array_equal(got, np.swapaxes(got, -1, -2)), "symmetrised output must be exactly symmetric"
            # half the lower triangle, mirrored (the symmetric operand of the Cholesky VJP): NaN-prefilled output
            symlow = 0.5 * (np.tril(full) + np.transpose(np.tril(full, -1), (0, 2, 1)))
            out = torch.full(tuple(sq(full).shape), float("nan"), dtype=dt, device="cuda")
            got = host(H.matmul(A_, B_, transA=tA, transB=tB, out=out, epilogue=H.MM_SYMLOW_OUT))
            assert np.allclose(got, sq(symlow), **tol) and np.array_equal(got, np.swapaxes(got, -1, -2))
            got = host(H.matmul(A_, B_, transA=tA, transB=tB, lower_out=True))
            il = np.tril_indices(m)
            assert np.allclose(got[..., il[0], il[1]], sq(full)[..., il[0], il[1]], **tol)
    # A = I with an asymmetric B through this kernel (accumulator row/column map, k mapping of both operand forms)
    bb = np.arange(128 * 128, dtype=np.float64).reshape(128, 128) % 251
    eye = np.eye(128)
    assert np.array_equal(host(H.matmul(dev(eye, dt), dev(bb if not tB else bb.T.copy(), dt), transA=tA, transB=tB)), bb)
    assert np.array_equal(host(H.matmul(dev(bb if not tA else bb.T.copy(), dt), dev(eye, dt), transA=tA, transB=tB)), bb)


def test_matmul_asymmetric_identity(H):
    # A = I with an ASYMMETRIC B catches a swapped accumulator row/col map
    for dt in (torch.float32, torch.float64):
        b = np.arange(64 * 64, dtype=np.float64).reshape(64, 64)
        assert np.array_equal(host(H.matmul(dev(np.eye(64), dt), dev(b, dt))), b)
        assert np.array_equal(host(H.matmul(dev(b, dt), dev(np.eye(64), dt))), b)


@pytest.mark.parametrize("p", ["f32", "f64"])
def test_matmul_batch_bias_act_splitk_lower(H, p):
    dt = DT[p]
    rng = np.random.RandomState(0)
    tol = TOL[p] if p == "f64" else dict(rtol=2e-4, atol=2e-3)
    a, b = rng.randn(5, 6, 3), rng.randn(5, 3, 2)
    bias = rng.randn(5, 1, 2)
    exp = 1 / (1 + np.exp(-(a @ b + bias)))
    assert_close(H.matmul(dev(a, dt), dev(b, dt), bias=dev(bias, dt), act="sigmoid"), exp, tol)
    # 2-D weight broadcast over a batched activation, shared bias, relu / tanh
    w, bs = rng.randn(3, 4), rng.randn(4)
    assert_close(H.matmul(dev(a, dt), dev(w, dt), bias=dev(bs, dt), act="relu"), np.maximum(a @ w + bs, 0), tol)
    assert_close(H.matmul(dev(a, dt), dev(w, dt), bias=dev(bs, dt), act="tanh"), np.tanh(a @ w + bs), tol)
    # split-K path: few tiles, long contraction
    a2, b2 = rng.randn(3, 5000), rng.randn(70, 5000)
    assert_close(H.matmul(dev(a2, dt), dev(b2, dt), transB=True, alpha=-0.5), -0.5 * a2 @ b2.T,
                 TOL[p] if p == "f64" else dict(rtol=1e-3, atol=2e-2))
    # lower-only output: tiles touching the lower triangle are exact
    a3 = rng.randn(200, 40)
    got = host(H.matmul(dev(a3, dt), dev(a3, dt), transB=True, lower_out=True))
    ref = a3 @ a3.T
    il = np.tril_indices(200)
    assert np.allclose(got[il], ref[il], **(TOL[p] if p == "f64" else dict(rtol=1e-3, atol=1e-3)))
    # tril output (direct epilogue and split-K finish): lower triangle exact, strict upper exactly zero
    for kk in (40, 3000):
        a4 = rng.randn(2, 200, kk)
        got = host(H.matmul(dev(a4, dt), dev(a4, dt), transB=True, tril_out=True))
        ref = np.tril(a4 @ np.transpose(a4, (0, 2, 1)))
        assert np.all(np.triu(got, 1) == 0)
        assert np.allclose(got, ref, **(TOL[p] if p == "f64" else dict(rtol=1e-3, atol=2e-2)))
    # Phi / symmetrising epilogues (Cholesky VJP), direct and split-K
    for kk in (24, 2500):
        a5, b5 = rng.randn(2, 130, kk), rng.randn(2, kk, 130)
        full = a5 @ b5
        tol5 = TOL[p] if p == "f64" else dict(rtol=1e-3, atol=2e-2)
        phi = np.tril(full, -1) + 0.5 * np.einsum("bii->bi", full)[:, :, None] * np.eye(130)
        got = host(H.matmul(dev(a5, dt), dev(b5, dt), epilogue=H.MM_PHI_OUT))
        assert np.all(np.triu(got, 1) == 0) and np.allclose(got, phi, **tol5)
        got = host(H.matmul(dev(a5, dt), dev(b5, dt), epilogue=H.MM_SYM_OUT))
        assert np.allclose(got, 0.5 * (full + np.transpose(full, (0, 2, 1))), **tol5)
        symlow = 0.5 * (np.tril(full) + np.transpose(np.tril(full, -1), (0, 2, 1)))
        out = torch.full((2, 130, 130), float("nan"), dtype=dt, device="cuda")
        got = host(H.matmul(dev(a5, dt), dev(b5, dt), out=out, epilogue=H.MM_SYMLOW_OUT))
        assert np.allclose(got, symlow, **tol5) and np.array_equal(got, np.swapaxes(got, -1, -2))
        assert_close(H.matutil(dev(full, dt), 4), symlow, TOL[p])
    # beta accumulate
    c0 = rng.randn(6, 2)
    out = dev(np.broadcast_to(c0, (5, 6, 2)).copy(), dt)
    H.matmul(dev(a, dt), dev(b, dt), out=out, beta=1.0)
    assert_close(out, a @ b + c0, tol)


# ------------------------------------------------------------------ Cholesky / trinv
def _spd(rng, B, M, cond_jitter=1e-3):
    X = np.sort(rng.uniform(0, 0.4 * M, (B, M, 1)), axis=1)
    K = np.exp(-0.5 * (X - np.transpose(X, (0, 2, 1))) ** 2)
    return K + cond_jitter * np.eye(M)


@pytest.mark.parametrize("p", ["f32", "f64"])
@pytest.mark.parametrize("M", [1, 2, 5, 31, 32, 33, 60, 64, 100, 129, 257, 512])
def test_cholesky(H, p, M):
    dt = DT[p]
    rng = np.random.RandomState(M)
    A = _spd(rng, 2, M, 1e-2 if p == "f32" else 1e-5)
    L, info = H.cholesky(dev(A, dt))
    assert info.cpu().tolist() == [0, 0]
    Lh = host(L)
    assert np.all(np.triu(Lh, 1) == 0)
    ref = np.linalg.cholesky(A)
    if p == "f64":
        assert np.allclose(Lh, ref, rtol=1e-8, atol=1e-9)
    else:
        # reference bar: L L^T ~ K, atol 9e-4 (testing/test_kernels.py:196-198)
        # observed on MI355X (round 3) over M = 1 .. 512: L L^T - K 7.1e-8 .. 3.7e-7, L (worst tile) 3.5e-8 .. 5.2e-6
        observe("cholesky_f32[M%d]/LLt-K" % M, np.abs(Lh @ np.transpose(Lh, (0, 2, 1)) - A).max(), 3e-6)
        # against numpy's fp64 factor of the same matrix (cond ~ 1e2 at this nugget): worst 32 x 32 tile
        observe("cholesky_f32[M%d]/L" % M, tile_err(Lh, ref), 5e-5)


@pytest.mark.parametrize("p", ["f32", "f64"])
@pytest.mark.parametrize("M", [1, 7, 32, 33, 64, 65, 100, 200, 256, 512])
def test_cholesky_inverse_fused(H, p, M):
    """hb_cholesky_inverse: L and W = L^-1 from the same launches == separate factor + inverse."""
    dt = DT[p]
    rng = np.random.RandomState(M + 1)
    A = _spd(rng, 3, M, 1e-2 if p == "f32" else 1e-5)
    L, W, info = H.cholesky_inverse(dev(A, dt))
    assert info.cpu().tolist() == [0, 0, 0]
    Lh, Wh = host(L), host(W)
    assert np.all(np.triu(Lh, 1) == 0) and np.all(np.triu(Wh, 1) == 0)
    ref = np.linalg.cholesky(A)
    refW = np.linalg.inv(ref)
    if p == "f64":
        assert np.allclose(Lh, ref, rtol=1e-8, atol=1e-9)
        assert np.allclose(Wh, refW, rtol=1e-7, atol=1e-8 * np.abs(refW).max())
    else:
        # observed on MI355X (round 3) over M = 1 .. 512: L 3.5e-8 .. 4.7e-6, W 7.9e-9 .. 3.3e-5, W L - I 4.3e-8 .. 3.4e-6
        observe("cholesky_inverse_f32[M%d]/L" % M, tile_err(Lh, ref), 4e-5)
        observe("cholesky_inverse_f32[M%d]/W" % M, tile_err(Wh, refW), 3e-4)
        # W L = I to fp32 accuracy (the conditioning of L is ~sqrt of A's)
        err = np.abs(Wh.astype(np.float64) @ Lh.astype(np.float64) - np.eye(M)).max()
        observe("cholesky_inverse_f32[M%d]/WL-I" % M, err, 3e-5)
    L2, _ = H.cholesky(dev(A, dt))
    if p == "f32" and M % 64 == 0:
        # fp32, M % 64 == 0: the fused call is ONE persistent launch (csrc/chol_persist.cuh), the plain factorisation the
        # launch chain -- the same right-looking algorithm in a different operation order: equal to fp32 rounding
        observe("cholesky_inverse_f32[M%d]/L-vs-plain" % M, tile_err(Lh, host(L2)), 4e-5)
        # and the launch-chain form of the fused call (hb_debug_set chol_persist 0) is the plain factorisation bit for bit
        H.debug_set("chol_persist", 0)
        try:
            L3, W3, info3 = H.cholesky_inverse(dev(A, dt))
        finally:
            H.debug_set("chol_persist", 1)
        assert torch.equal(L3, L2) and info3.cpu().tolist() == [0, 0, 0]
        observe("cholesky_inverse_f32[M%d]/W-vs-chain" % M, tile_err(Wh, host(W3)), 3e-4)
    else:
        # the plain factorisation is bit-identical (same kernel, inverse rows are extra width only)
        assert torch.equal(L, L2)


def test_cholesky_inverse_reports_failure(H):
    rng = np.random.RandomState(0)
    A = _spd(rng, 2, 96)
    A[1, 50, 50] = -1.0
    _, _, info = H.cholesky_inverse(dev(A, torch.float64))
    assert info.cpu().tolist() == [0, 51]


def test_cholesky_inplace_and_info(H):
    rng = np.random.RandomState(0)
    A = _spd(rng, 1, 70)[0]
    a = dev(A, torch.float64)
    L, info = H.cholesky(a, out=a)
    assert np.allclose(host(L), np.linalg.cholesky(A), atol=1e-9) and info.item() == 0
    bad = A.copy()
    bad[40, 40] = -1.0  # leading minor 41 is not positive definite
    _, info = H.cholesky(dev(bad, torch.float64))
    assert info.item() == 41
    # golden: reference testing/test_kernels.py:184-198
    _, info = H.cholesky(dev(np.zeros((3, 3)), torch.float32))
    assert info.item() == 1


@pytest.mark.parametrize("p", ["f32", "f64"])
@pytest.mark.parametrize("M", [1, 7, 32, 33, 64, 65, 100, 200, 512])
def test_trinv(H, p, M):
    dt = DT[p]
    rng = np.random.RandomState(M)
    L = np.linalg.cholesky(_spd(rng, 2, M, 1e-2 if p == "f32" else 1e-5))
    W = host(H.trinv(dev(L, dt)))
    assert np.all(np.triu(W, 1) == 0)
    ref = np.linalg.inv(L)
    if p == "f64":
        assert np.allclose(W, ref, rtol=1e-7, atol=1e-8 * np.abs(ref).max())
    else:
        # observed on MI355X (round 3) over M = 1 .. 512: W L - I 5.2e-8 .. 2.3e-6, W (worst tile) 5.2e-8 .. 3.0e-6
        observe("trinv_f32[M%d]/WL-I" % M, np.abs(W @ L - np.eye(M)).max(), 2e-5)
        observe("trinv_f32[M%d]/W" % M, tile_err(W, ref), 2.5e-5)


# ------------------------------------------------------------------ K5/K6 fused sparse GP
def _sgp_case(rng, n, M, d, P, ard):
    if d == 1:
        z = (np.linspace(0, 0.5 * M, M) + 0.05 * rng.randn(M))[:, None]
    else:
        z = rng.uniform(-2, 2, (M, d))
    x = rng.uniform(z.min(), z.max(), (n, d))
    ell = np.exp(0.2 * rng.randn(d if ard else 1)) * (1.0 if d == 1 else 0.6)
    u = rng.randn(P, M)
    eps = rng.randn(n)
    return x, z, ell, u, eps


@pytest.mark.parametrize("p", ["f32", "f64"])
@pytest.mark.parametrize("n,M,d,P,ard,mode", [
    (40, 30, 2, 20, False, "diagonal"),      # reference testing/test_gp.py:59-66 sizes
    (130, 64, 1, 1, False, "diagonal"),
    (257, 60, 1, 3, False, "neglected"),
    (1000, 129, 3, 2, True, "diagonal"),
    (300, 200, 5, 1, True, "diagonal"),
    (2048, 512, 1, 1, False, "diagonal"),
])
def test_sgp_fwd_bwd(H, p, n, M, d, P, ard, mode):
    dt = DT[p]
    rng = np.random.RandomState(0)
    x, z, ell, u, eps = _sgp_case(rng, n, M, d, P, ard)
    jitter = 1e-2 if p == "f32" else 1e-5
    tz, tl, tu = [torch.as_tensor(t).requires_grad_(True) for t in (z, ell, u)]
    tx = torch.as_tensor(x).requires_grad_(True)
    L = O.kern_cholesky(tz.detach(), tl.detach(), jitter).clone().requires_grad_(True)
    # oracle forward with L as an independent leaf (the kernel's contract): A = L^{-1} K(z,x)
    A = torch.linalg.solve_triangular(L, O.rbf_K(tz, tx, tl), upper=False)
    mean = tu @ A
    v = 1.0 - (A * A).sum(0)
    f = mean + torch.sqrt(torch.abs(v)) * torch.as_tensor(eps) if mode == "diagonal" else mean
    fbar = rng.randn(P, n)
    grads = torch.autograd.grad((f * torch.as_tensor(fbar)).sum(), [L, tu, tz, tl, tx])
    m = H.SGP_DIAGONAL if mode == "diagonal" else H.SGP_NEGLECTED
    Ld = dev(L.detach().numpy(), dt)
    W = H.trinv(Ld)
    fo, Ao, vo, eo = H.sgp_fwd(dev(x, dt), dev(z, dt), dev(ell, dt), W, dev(u, dt), eps_in=dev(eps, dt), mode=m)
    tol = TOL[p] if p == "f64" else dict(rtol=5e-3, atol=5e-3)
    if p == "f64":
        tol = dict(rtol=1e-7, atol=1e-8)
    if p == "f32":
        # fp32 forward against the fp64 oracle of the same contract (L fp32-rounded on the way in, jitter 1e-2:
        # cond(L) ~ 1e1..1e2); the fp32 backward kernels are held to the oracle in test_fp32_parity_gpu.py
        tag = "sgp_fwd_f32[n%d,M%d,d%d,P%d]/" % (n, M, d, P)
        # A componentwise against (|W| |K|)_ij (entries small by cancellation carry the rounding of the terms that cancel);
        # observed on MI355X (round 3) over the six shapes: A 1.9e-6 .. 1.0e-4 (the largest at M = 512), v 2.2e-7 .. 5.4e-6, f 1.5e-7 .. 4.4e-5
        Wr = np.linalg.inv(L.detach().numpy())
        Kzx = O.rbf_K(tz, tx, tl).detach().numpy()
        observe(tag + "A", prod_err(host(Ao), A.detach().numpy(), np.abs(Wr), np.abs(Kzx)), 8e-4)
        observe(tag + "v", np.abs(host(vo) - v.detach().numpy()).max(), 5e-5)
        observe(tag + "f", np.abs(host(fo) - f.detach().numpy()).max() / max(1.0, float(f.detach().abs().max())), 4e-4)
        return
    assert_close(Ao, A, tol, "A")
    assert_close(vo, v, tol, "v")
    assert_close(fo, f, tol, "f")
    Lb, ub, zb, lb, xb = H.sgp_bwd(dev(x, dt), dev(z, dt), dev(ell, dt), W, dev(u, dt), dev(eps, dt), Ao, vo,
                                   dev(fbar, dt), mode=m, need_xbar=True)
    gtol = dict(rtol=1e-6, atol=1e-6 * max(1.0, float(grads[0].abs().max())))
    assert_close(Lb, torch.tril(grads[0]), gtol, "Lbar")
    assert_close(ub, grads[1], dict(rtol=1e-7, atol=1e-8), "ubar")
    assert_close(zb, grads[2], dict(rtol=1e-6, atol=1e-6 * max(1.0, float(grads[2].abs().max()))), "zbar")
    assert_close(lb, grads[3], dict(rtol=1e-6, atol=1e-6 * max(1.0, float(grads[3].abs().max()))), "ellbar")
    assert_close(xb.reshape(n, d), grads[4], dict(rtol=1e-6, atol=1e-6 * max(1.0, float(grads[4].abs().max()))), "xbar")


def test_sgp_matches_reference_composition_and_golden(H, golden):
    """Full reference composition chol -> trisolve -> samples (gp/gp.py:99-143) through the HIP kernels."""
    g = golden
    dt = torch.float64
    z, ell, x = g["g_z"], g["g_ell"], g["g_x"]
    K = H.gram_fwd(dev(z, dt), dev(z, dt), dev(ell, dt))
    L, info = H.cholesky(H.matutil(K, H.MATUTIL_ADD_EYE, alpha=1e-5))
    assert info.item() == 0
    W = H.trinv(L)
    u = np.random.RandomState(0).randn(20, 30)
    eps = np.random.RandomState(1).randn(20)
    f, A, v, _ = H.sgp_fwd(dev(x, dt), dev(z, dt), dev(ell, dt), W, dev(u, dt), eps_in=dev(eps, dt))
    # fixture is ill-conditioned (cond ~1e9): compare at the reference's own tolerances
    assert_close(A, g["g_LnT"], dict(rtol=1e-4, atol=1e-4))          # _effective_LT
    assert_close(v, g["g_cov_diag"], dict(rtol=0, atol=1e-4))         # test_gp.py:115-131 atol 1e-4
    ref = O.sparse_samples(O.T(x), O.T(u), O.T(z), O.T(ell), 1e-5, "diagonal", O.T(eps))
    assert_close(f, ref, dict(rtol=1e-4, atol=1e-4))
    # x == z: effective L^T equals chol(K)^T (test_gp.py:68-91, atol 5e-3)
    _, A2, _, _ = H.sgp_fwd(dev(z, dt), dev(z, dt), dev(ell, dt), W, dev(u, dt), eps_in=dev(np.zeros(30), dt))
    assert_close(A2, g["g_cholT"], dict(rtol=0, atol=5e-3))


def test_sgp_experts_batched_and_rng(H):
    dt = torch.float64
    rng = np.random.RandomState(0)
    E, n, M, d, P = 3, 150, 40, 1, 1
    x = rng.uniform(0, 20, (n, d))
    z = np.stack([np.sort(rng.uniform(0, 20, (M, d)), axis=0) for _ in range(E)])
    ell = np.exp(0.2 * rng.randn(E, 1))
    u = rng.randn(E, P, M)
    eps = rng.randn(E, n)
    Ws, fs = [], []
    for e in range(E):
        L = O.kern_cholesky(O.T(z[e]), O.T(ell[e]), 1e-4)
        Ws.append(np.linalg.inv(L.numpy()))
        fs.append(O.sparse_samples(O.T(x), O.T(u[e]), O.T(z[e]), O.T(ell[e]), 1e-4, "diagonal", O.T(eps[e])).numpy())
    f, A, v, eo = H.sgp_fwd(dev(x, dt), dev(z, dt), dev(ell, dt), dev(np.stack(Ws), dt), dev(u, dt), eps_in=dev(eps, dt))
    assert_close(f, np.stack(fs), dict(rtol=1e-6, atol=1e-6))
    # in-kernel noise: eps_out reproduces f through the oracle
    f2, _, _, e2 = H.sgp_fwd(dev(x, dt), dev(z, dt), dev(ell, dt), dev(np.stack(Ws), dt), dev(u, dt), rng=H.Rng(3))
    e2h = host(e2)
    assert abs(e2h.mean()) < 0.2 and 0.8 < e2h.std() < 1.2
    ref = np.stack([O.sparse_samples(O.T(x), O.T(u[e]), O.T(z[e]), O.T(ell[e]), 1e-4, "diagonal", O.T(e2h[e])).numpy()
                    for e in range(E)])
    assert_close(f2, ref, dict(rtol=1e-6, atol=1e-6))


def test_fragment_major_copies_of_the_inverse(H):
    """hb_cholesky_inverse's optional Wfrag output (fragment-major W and W^T, include/henbun_hip.h) has the
    documented element order, and the column-strip contraction gives the same BITS from it as from row-major W."""
    rng = np.random.RandomState(5)
    for B, M in ((1, 64), (2, 96), (1, 512)):
        z = np.sort(rng.uniform(0, M / 2.0, (B, M, 1)), axis=1)
        K = H.gram_fwd(dev(z, torch.float32), dev(z, torch.float32), dev(np.ones(1), torch.float32), diag_add=1e-3)
        K = K.reshape(B, M, M)
        frag = torch.full((2 * B * M * M,), float("nan"), dtype=torch.float32, device="cuda")
        L, W, info = H.cholesky_inverse(K, frag=frag)
        L2, W2, _ = H.cholesky_inverse(K)
        assert not info.cpu().numpy().any() and torch.equal(W, W2) and torch.equal(L, L2)
        Wh = W.cpu().numpy()
        nT = M // 32
        t, Q, v, lane, s = np.meshgrid(np.arange(nT), np.arange(nT), np.arange(4), np.arange(64), np.arange(4), indexing="ij")
        r, k = 32 * t + (lane & 31), 32 * Q + 16 * (lane >> 5) + 4 * v + s
        fr = frag.cpu().numpy().reshape(2, B, nT, nT, 4, 64, 4)
        for b in range(B):
            assert np.array_equal(fr[0, b], Wh[b][r, k])
            assert np.array_equal(fr[1, b], Wh[b].T[r, k])
        x = dev(rng.uniform(0, M / 2.0, (700, 1)), torch.float32)
        u = dev(rng.randn(B, 1, M), torch.float32)
        eps = dev(rng.randn(B, 700), torch.float32)
        zz, ell = dev(z, torch.float32), dev(np.ones((B, 1)), torch.float32)
        if B == 1:
            zz, ell, u, eps = zz[0], ell[0], u[0], eps[0]
            Wd = W.reshape(M, M)
        else:
            Wd = W
        a = H.sgp_fwd(x, zz, ell, Wd, u, eps_in=eps)
        b_ = H.sgp_fwd(x, zz, ell, Wd, u, eps_in=eps, wfrag=frag)
        H.debug_set("sgp_form16", 1)     # (opt-in; M >= 384: the sixteen-wave form on 16x16x4 MFMAs, otherwise the third form again)
        try:
            b16 = H.sgp_fwd(x, zz, ell, Wd, u, eps_in=eps, wfrag=frag)
        finally:
            H.debug_clear()
        # the sixteen-wave form adds four products per MFMA instead of two: A agrees to rounding -- of the terms that cancel
        # in W K (entries of W reach +-30 here), i.e. ~1e-5 of the largest entry, not of each entry
        assert float((b16[1] - b_[1]).abs().max()) <= 1e-4 * float(b_[1].abs().max())      # observed <= 1.1e-5
        assert float((b16[0] - b_[0]).abs().max()) <= 1e-3 * float(b_[0].abs().max())
        # A = W K keeps its bits (same products, same order, whichever operand sits on the lanes); the column sums
        # behind f and v are folded in another fixed order by the transposed-accumulator strip form (observed 2.4e-7 / 6e-7)
        assert torch.equal(a[1], b_[1]) and (a[3] is None or torch.equal(a[3], b_[3]))
        # (v = 1 - sum A^2 is ~1e-3 here, so f = mean + sqrt|v| eps magnifies a rounding of v ~16 times)
        assert float((a[2] - b_[2]).abs().max()) <= 4e-6, float((a[2] - b_[2]).abs().max())     # observed 1.2e-6
        assert float((a[0] - b_[0]).abs().max()) <= 6e-5, float((a[0] - b_[0]).abs().max())     # observed 9.0e-6
        H.debug_set("sgp_strip_form2", 1)   # the second strip form (kept for P > 1) sums like the row-major one
        try:
            b2 = H.sgp_fwd(x, zz, ell, Wd, u, eps_in=eps, wfrag=frag)
        finally:
            H.debug_set("sgp_strip_form2", 0)
        for p_, q_ in zip(a, b2):
            assert p_ is None or torch.equal(p_, q_)
        assert torch.equal(H.sgp_A(x, zz, ell, Wd), H.sgp_A(x, zz, ell, Wd, wfrag=frag))


@pytest.mark.parametrize("E,M,n,d,P,mode", [(1, 512, 3000, 1, 1, "diagonal"), (2, 96, 257, 2, 3, "diagonal"),
                                             (1, 64, 64, 3, 1, "neglected"), (3, 160, 1000, 1, 2, "diagonal"),
                                             (96, 96, 2048, 1, 1, "diagonal")])   # many blocks, ragged 128-row blocks: sgp_lbar_lds_kernel
def test_sgp_backward_column_strip_form(H, E, M, n, d, P, mode):
    """hb_sgp_bwd with the fragment-major W^T (one column-strip kernel: Kbar + the row gradients zbar / ellbar / ubar
    folded as each 32 x 32 tile completes, no second pass over Kbar and A) against torch autograd in fp64 -- at fp32
    tolerances, for the native and the bf16x3 operand forms -- and against the generic fp32 kernels."""
    dt = torch.float32
    rng = np.random.RandomState(5)
    z = np.stack([np.sort(rng.uniform(0, M / 2.0, (M, d)), axis=0) for _ in range(E)])
    ellv = np.exp(0.1 * rng.randn(E, d)) if d > 1 else np.exp(0.1 * rng.randn(E, 1))
    x = rng.uniform(0, M / 2.0, (n, d))
    u, eps, fbar = rng.randn(E, P, M), rng.randn(E, n), rng.randn(E, P, n)
    K = H.gram_fwd(dev(z, dt), dev(z, dt), dev(ellv, dt), diag_add=1e-2).reshape(E, M, M)
    frag = torch.zeros(5 * E * M * M, dtype=dt, device="cuda")
    L, W, info = H.cholesky_inverse(K, frag=frag, frag_bf16x3=True)
    assert not info.cpu().numpy().any()
    m = H.SGP_DIAGONAL if mode == "diagonal" else H.SGP_NEGLECTED
    sq = (lambda a: a if E > 1 else a[0])
    args = (dev(x, dt), dev(sq(z), dt), dev(sq(ellv), dt), W if E > 1 else W[0], dev(sq(u), dt))
    f, A, v, _ = H.sgp_fwd(*args, eps_in=dev(sq(eps), dt), mode=m, wfrag=frag)
    # fp64 reference of the same contract (L an independent leaf, built from the fp32 factor)
    refs = []
    for e in range(E):
        Lt = L[e].double().cpu().clone().requires_grad_(True)
        tz, tl, tu = [torch.as_tensor(t).requires_grad_(True) for t in (z[e], ellv[e], u[e])]
        Ar = torch.linalg.solve_triangular(Lt, O.rbf_K(tz, torch.as_tensor(x), tl), upper=False)
        vr = 1.0 - (Ar * Ar).sum(0)
        fr = tu @ Ar + (torch.sqrt(torch.abs(vr)) * torch.as_tensor(eps[e]) if mode == "diagonal" else 0.0)
        refs.append(torch.autograd.grad((fr * torch.as_tensor(fbar[e])).sum(), [Lt, tu, tz, tl]))
    ref = [np.stack([r[i].numpy() for r in refs]) for i in range(4)]
    ref[0] = np.tril(ref[0])
    bargs = args + (dev(sq(eps), dt), A, v, dev(sq(fbar), dt))
    old = H.sgp_bwd(*bargs, mode=m)                                          # generic fp32 kernels
    new = H.sgp_bwd(*bargs, mode=m, wfrag=frag)                              # column-strip form
    bf3 = H.sgp_bwd(*bargs, mode=m, wfrag=frag, prec=H.PREC_BF16X3)          # ... with bf16x3 operands
    # fragment-major exchange of A and Kbar (no row-major copies), Lbar from the fragment-major contraction
    assert H.sgp_strip_path(E, n, M, d, P)
    a_frag = torch.full((H.sgp_frag_elems(E, n, M),), float("nan"), dtype=dt, device="cuda")
    f2, _, v2, _ = H.sgp_fwd(*args, eps_in=dev(sq(eps), dt), mode=m, wfrag=frag, a_frag=a_frag, skip_a=True)
    assert torch.equal(f2, f) and torch.equal(v2, v)
    nS, nT = (n + 31) // 32, M // 32
    Ah = np.zeros((E, M, 32 * nS))
    Ah[:, :, :n] = host(A).reshape(E, M, n)
    t_, s_, v_, l_, q_ = np.meshgrid(np.arange(nT), np.arange(nS), np.arange(4), np.arange(64), np.arange(4), indexing="ij")
    want = Ah[:, 32 * t_ + (l_ & 31), 32 * s_ + 16 * (l_ >> 5) + 4 * v_ + q_]
    assert np.array_equal(host(a_frag).reshape(E, nT, nS, 4, 64, 4), want), "fragment-major A layout"
    frg = H.sgp_bwd(*(args + (dev(sq(eps), dt), None, v, dev(sq(fbar), dt))), mode=m, wfrag=frag, a_frag=a_frag)
    # the same exchange as bf16x3 planes: forward, Kbar, row gradients AND the Lbar contraction on bf16x3 operands
    a_frag3 = torch.zeros(H.sgp_frag_elems(E, n, M, H.PREC_BF16X3), dtype=dt, device="cuda")
    f3, _, v3, _ = H.sgp_fwd(*args, eps_in=dev(sq(eps), dt), mode=m, wfrag=frag, a_frag=a_frag3, skip_a=True,
                             prec=H.PREC_BF16X3)
    assert np.abs(host(f3) - host(f)).max() < 2e-3 and np.abs(host(v3) - host(v)).max() < 2e-3
    frg3 = H.sgp_bwd(*(args + (dev(sq(eps), dt), None, v3, dev(sq(fbar), dt))), mode=m, wfrag=frag, a_frag=a_frag3,
                     prec=H.PREC_BF16X3)
    names = ("Lbar", "ubar", "zbar", "ellbar")
    for i, nm in enumerate(names):
        r = ref[i].reshape(host(old[i]).shape)
        scale = max(1.0, np.abs(r).max())
        e_old = np.abs(host(old[i]) - r).max() / scale
        for tag, got in (("strip", new), ("bf16x3", bf3), ("fragment-major", frg), ("fragment-major bf16x3", frg3)):
            e_new = np.abs(host(got[i]) - r).max() / scale
            # ellbar is ONE number summed over all M n entries of Kbar o dK (heavy cancellation): with bf16x3 operands
            # its error sits at ~1e-4 of max(1, |ellbar|) whatever the fp32 forms happen to reach (2e-5 .. 9e-5 seen)
            slack = 1.5e-4 if (nm == "ellbar" and "bf16x3" in tag) else 2e-5
            assert e_new <= 3.0 * e_old + slack, (nm, tag, e_old, e_new)
            # absolute caps (the relative form above bounds nothing if the generic kernels are themselves off)
            # observed on MI355X (round 3), worst of the four shapes and of every operand form (jitter 1e-2, cond(L) ~ 1e1..1e2):
            # Lbar 2.7e-4, ubar 5.0e-5, zbar 5.0e-4, ellbar 3.3e-5 (9.1e-5 with bf16x3 operands); generic kernels the same
            cap = {"Lbar": 2e-3, "ubar": 4e-4, "zbar": 4e-3, "ellbar": 8e-4 if "bf16x3" in tag else 3e-4}[nm]
            observe("strip_bwd[E%d,M%d,n%d]/%s/%s" % (E, M, n, nm, tag), e_new, cap)
        observe("strip_bwd[E%d,M%d,n%d]/%s/generic" % (E, M, n, nm), e_old, {"Lbar": 2e-3, "ubar": 4e-4, "zbar": 4e-3, "ellbar": 7e-4}[nm])


def test_side_jobs_ride_on_a_host_launch_and_flush_otherwise(H):
    """hb_side_push_*: the minibatch draw + gather, the diagonal sampler and its VJP recorded instead of launched, then
    (a) adopted by launch 0 of the Cholesky chain, (b) adopted by the in-workgroup split-K GEMM, (c) flushed as a
    launch of their own, (d) a fourth push flushing the first three -- every time the same bits as the stand-alone
    entry points on the same RNG states."""
    dt = torch.float32
    rng_np = np.random.RandomState(3)
    N, n, Msz = 5000, 700, 512
    X, Y = dev(rng_np.randn(N, 1), dt), dev(rng_np.randn(N, 3), dt)
    perm = torch.as_tensor(rng_np.permutation(N)[:4000]).cuda()
    mu, s = dev(0.1 * rng_np.randn(Msz), dt), dev(0.1 * rng_np.randn(Msz) - 1.0, dt)
    xbar, klbar = dev(rng_np.randn(Msz), dt), dev(np.array([-1.0]), dt)
    K = H.gram_fwd(dev(np.linspace(0, 100, 256)[:, None], dt), dev(np.linspace(0, 100, 256)[:, None], dt),
                   torch.ones(1, device="cuda"), diag_add=1e-2)
    A_, B_ = dev(rng_np.randn(128, 256), dt), dev(rng_np.randn(256, 64), dt)

    def jobs(defer, seed):
        r1, r2 = H.Rng(seed, 1, device="cuda"), H.Rng(seed, 2, device="cuda")
        xo, yo = torch.empty(n, 1, dtype=dt, device="cuda"), torch.empty(n, 3, dtype=dt, device="cuda")
        idx = torch.zeros(n, dtype=torch.int64, device="cuda")
        err = torch.zeros(1, dtype=torch.int32, device="cuda")
        mg = H.MultiGather([X, Y], [xo, yo], idx, perm, err)
        mg.launch_draw(r1, 0, 4000, defer=defer)
        x, kl, u = H.diag_sample_kl_fwd(mu, s, rng=r2, defer=defer)
        return dict(xo=xo, yo=yo, idx=idx, x=x, kl=kl, u=u, r1=r1.state, r2=r2.state, keep=(mg, r1, r2))

    def same(a, b):
        for k in ("xo", "yo", "idx", "x", "kl", "u", "r1", "r2"):
            assert torch.equal(a[k], b[k]), k

    ref = jobs(False, 11)
    assert H.side_pending() == 0
    # (a) riding on the Cholesky chain
    got = jobs(True, 11)
    assert H.side_pending() == 2
    L1, W1, _ = H.cholesky_inverse(K)
    assert H.side_pending() == 0
    torch.cuda.synchronize()
    same(ref, got)
    L0, W0, _ = H.cholesky_inverse(K)
    assert torch.equal(L0, L1) and torch.equal(W0, W1)
    # (b) the sampler's VJP riding on the small-GEMM kernel
    mb0, sb0 = H.diag_sample_kl_bwd(s, ref["u"], ref["x"], xbar, klbar)
    mb1, sb1 = H.diag_sample_kl_bwd(s, ref["u"], ref["x"], xbar, klbar, defer=True)
    assert H.side_pending() == 1
    C1 = H.matmul(A_, B_)
    assert H.side_pending() == 0
    torch.cuda.synchronize()
    assert torch.equal(mb0, mb1) and torch.equal(sb0, sb1) and torch.equal(C1, H.matmul(A_, B_))
    # (c) nobody adopts them: flush
    got = jobs(True, 11)
    H.side_flush()
    assert H.side_pending() == 0
    torch.cuda.synchronize()
    same(ref, got)
    H.side_flush()   # nothing pending: a no-op
    # (d) a fourth push runs the first three
    got = jobs(True, 11)
    mb2, sb2 = H.diag_sample_kl_bwd(s, ref["u"], ref["x"], xbar, klbar, defer=True)
    assert H.side_pending() == 3
    mb3, sb3 = H.diag_sample_kl_bwd(s, ref["u"], ref["x"], xbar, klbar, defer=True)
    assert H.side_pending() == 1
    H.side_flush()
    torch.cuda.synchronize()
    same(ref, got)
    assert torch.equal(mb0, mb2) and torch.equal(mb0, mb3) and torch.equal(sb0, sb3)


def test_bf16x3_contraction_has_fp32_accuracy(H):
    """HB_PREC_BF16X3 (BASELINE cfg 5's "fp16-with-fp32-accum" variant in a usable form): A = L^-1 K(z,x) with every
    operand split into three bf16 terms on v_mfma_f32_32x32x16_bf16 is as close to the fp64 result as the fp32-operand
    kernel is (cfg-2-like Kmm, cond ~ 1e5: entries of L^-1 reach +-30 and cancel), and the column statistics /
    draw built on it agree; plain 16-bit operands would be off by ~27 % (profiles/r01_bf16_split_study.txt)."""
    rng = np.random.RandomState(11)
    for E, M, n in ((1, 512, 4096), (2, 256, 1000), (1, 96, 333)):
        z = np.broadcast_to(np.linspace(0, M / 2.0, M)[None, :, None], (E, M, 1)).copy()
        ellv = np.ones((E, 1))
        x = rng.uniform(0, M / 2.0, (n, 1))
        zz, ell, xx = dev(z, torch.float32), dev(ellv, torch.float32), dev(x, torch.float32)
        K = H.gram_fwd(zz, zz, ell, diag_add=1e-4).reshape(E, M, M)
        frag = torch.zeros(5 * E * M * M, dtype=torch.float32, device="cuda")
        L, W, info = H.cholesky_inverse(K, frag=frag, frag_bf16x3=True)
        assert not info.cpu().numpy().any()
        u = dev(rng.randn(E, 1, M), torch.float32)
        eps = dev(rng.randn(E, n), torch.float32)
        if E == 1:
            zz, ell, u, eps, Wd = zz[0], ell[0], u[0], eps[0], W.reshape(M, M)
        else:
            Wd = W
        f32 = H.sgp_fwd(xx, zz, ell, Wd, u, eps_in=eps, wfrag=frag)
        fb3 = H.sgp_fwd(xx, zz, ell, Wd, u, eps_in=eps, wfrag=frag, prec=H.PREC_BF16X3)
        # fp64 reference from the SAME fp32 factor (the comparison is about the contraction, not the factorisation)
        Wh = W.double().cpu().numpy().reshape(E, M, M)
        Kzx = np.exp(-0.5 * (z[:, :, None, 0] - x[None, None, :, 0]) ** 2)
        Aref = Wh @ Kzx
        A32, Ab3 = f32[1].double().cpu().numpy().reshape(E, M, n), fb3[1].double().cpu().numpy().reshape(E, M, n)
        e32, eb3 = np.abs(A32 - Aref).max(), np.abs(Ab3 - Aref).max()
        assert eb3 <= 2.0 * e32 + 1e-6, (E, M, n, e32, eb3)
        assert np.abs(Ab3 - A32).max() <= 4.0 * e32 + 1e-6
        for a_, b_ in ((f32[0], fb3[0]), (f32[2], fb3[2])):      # f and v from the column statistics
            assert np.abs(a_.double().cpu().numpy() - b_.double().cpu().numpy()).max() < 5e-4
        assert torch.equal(H.sgp_A(xx, zz, ell, Wd, wfrag=frag, prec=H.PREC_BF16X3), fb3[1])


# ------------------------------------------------------------------ Adam + graphs
@pytest.mark.parametrize("p", ["f32", "f64"])
def test_adam_matches_tf_formula(H, p):
    dt = DT[p]
    rng = np.random.RandomState(0)
    th = rng.randn(1000)
    ref = O.T(th.copy())
    opt = O.AdamTF([ref], lr=0.01)
    theta, m, v = dev(th, dt), dev(np.zeros(1000), dt), dev(np.zeros(1000), dt)
    t = torch.zeros(1, dtype=torch.int64, device="cuda")
    for step in range(20):
        g = rng.randn(1000)
        opt.step([O.T(g)])
        H.adam_step(theta, dev(g, dt), m, v, t, lr=0.01)
    assert t.item() == 20
    assert_close(theta, ref, TOL[p] if p == "f64" else dict(rtol=1e-4, atol=1e-5))


@pytest.mark.parametrize("p", ["f32", "f64"])
@pytest.mark.parametrize("n", [1000, 40000])   # one-block form (tick inside) and multi-block form (tick kernel)
def test_adam_segments_share_a_step_and_failed_factorisations_block_the_update(H, p, n):
    """hb_adam_step: `tick` lets several segment calls share one step; a non-zero factorisation status word
    (or the other ranks' flag, or the sticky record) makes the call a no-op and records the first blocked step."""
    dt = DT[p]
    rng = np.random.RandomState(1)
    th = rng.randn(n)
    ref = O.T(th.copy())
    opt = O.AdamTF([ref], lr=0.01)
    theta, m, v = dev(th, dt), dev(np.zeros(n), dt), dev(np.zeros(n), dt)
    t = torch.zeros(1, dtype=torch.int64, device="cuda")
    info = torch.zeros(5, dtype=torch.int32, device="cuda")
    fail = torch.zeros(2, dtype=torch.int64, device="cuda")
    flag = torch.zeros(1, dtype=dt, device="cuda")
    h = n // 3
    for step in range(3):
        g = rng.randn(n)
        opt.step([O.T(g)])
        gd = dev(g, dt)
        H.adam_step(theta[:h], gd[:h], m[:h], v[:h], t, lr=0.01, tick=False, info=info, dpflag=flag, fail=fail)
        assert t.item() == step
        H.adam_step(theta[h:], gd[h:], m[h:], v[h:], t, lr=0.01, tick=True, info=info, dpflag=flag, fail=fail)
        assert t.item() == step + 1
    assert_close(theta, ref, TOL[p] if p == "f64" else dict(rtol=1e-4, atol=1e-5))
    before = [x.clone() for x in (theta, m, v)]
    info[3] = 7   # "leading minor 7 is not positive definite"
    H.adam_step(theta, dev(rng.randn(n), dt), m, v, t, lr=0.01, info=info, dpflag=flag, fail=fail)
    torch.cuda.synchronize()
    assert t.item() == 3 and fail.tolist() == [4, 7]
    for a, b in zip(before, (theta, m, v)):
        assert torch.equal(a, b)
    info.zero_()      # the record is sticky: a later clean step is still blocked until the host clears it
    H.adam_step(theta, dev(rng.randn(n), dt), m, v, t, lr=0.01, info=info, dpflag=flag, fail=fail)
    assert t.item() == 3 and fail.tolist() == [4, 7] and torch.equal(before[0], theta)
    fail.zero_()
    flag.fill_(2.0)   # another rank failed
    H.adam_step(theta, dev(rng.randn(n), dt), m, v, t, lr=0.01, info=info, dpflag=flag, fail=fail)
    assert t.item() == 3 and fail.tolist() == [4, -1] and torch.equal(before[0], theta)
    fail.zero_()
    flag.zero_()
    H.adam_step(theta, dev(rng.randn(n), dt), m, v, t, lr=0.01, info=info, dpflag=flag, fail=fail)
    assert t.item() == 4 and fail.tolist() == [0, 0] and not torch.equal(before[0], theta)


def test_graph_capture_replay(H):
    dt = torch.float32
    theta = dev(np.ones(100), dt)
    g, m, v = dev(np.ones(100), dt), dev(np.zeros(100), dt), dev(np.zeros(100), dt)
    t = torch.zeros(1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        gr = H.CapturedGraph()
        gr.begin()
        H.adam_step(theta, g, m, v, t, lr=0.1)
        gr.end()
        for _ in range(5):
            gr.launch()
    s.synchronize()
    assert t.item() == 5  # capture itself does not execute
    ref = O.T(np.ones(100))
    opt = O.AdamTF([ref], lr=0.1)
    for _ in range(5):
        opt.step([O.T(np.ones(100))])
    assert_close(theta, ref, dict(rtol=1e-5, atol=1e-6))


# ------------------------------------------------------------------ fused elementwise program
def _ewise_mode(mode):
    """settings override selecting the compiled (hiprtc) or the interpreted form of a fused elementwise program"""
    import henbun_amd as hb

    cfg = hb.settings.get_settings()
    cfg.runtime.ewise = mode
    return hb.settings.temp_settings(cfg)


@pytest.mark.parametrize("mode", ["jit", "interpret"])
@pytest.mark.parametrize("p", ["f32", "f64"])
def test_ewise_program(H, p, mode):
    """out0 = softplus(a)+1e-6 (scalar, broadcast input); out1 = gauss_logpdf(y, f*sqrt(out0), v) [1,n];
    out2..4 = its gradient w.r.t. (x, mu, var) given g -- one launch, 3 register inputs reused."""
    dt, tol = DT[p], TOL[p]
    rng = np.random.RandomState(0)
    n = 300
    a, v = rng.randn(1), np.abs(rng.randn(1)) + 0.5
    y, f, g = rng.randn(1, n), rng.randn(1, n), rng.randn(1, n)
    ins = [dev(a, dt), dev(v, dt), dev(y, dt), dev(f, dt), dev(g, dt)]
    E = H.EW
    code = [[E["SOFTPLUS"], 5, 0, 0, 0], [E["AFFINE"], 6, 5, 0, 0], [E["SQRT"], 7, 6, 0, 0], [E["MUL"], 8, 3, 7, 0],
            [E["GAUSS_LOGPDF"], 9, 2, 8, 1], [E["GAUSS_LOGPDF_GRAD"], 10, 2, 8, 1]]
    params = [[0, 0], [1.0, 1e-6], [0, 0], [0, 0], [0, 0], [4.0, 0]]  # 4th operand of the grad op = register 4 (g)
    outs = [torch.empty(1, dtype=dt, device="cuda")] + [torch.empty(1, n, dtype=dt, device="cuda") for _ in range(4)]
    istr = [[0], [0], [1], [1], [1]]
    ostr = [[0], [1], [1], [1], [1]]
    with _ewise_mode(mode):
        prog = H.EwiseProgram(code, params, ins, istr, outs, [6, 9, 10, 11, 12], ostr, [n])
    assert (prog.image is None) == (mode == "jit")     # the compiled form must be the one that ran when asked for
    prog.launch()
    ta, tv, ty, tf_, tg = [torch.as_tensor(t) for t in (a, v, y, f, g)]
    kv = torch.nn.functional.softplus(ta) + 1e-6
    mu = (tf_ * kv.sqrt()).requires_grad_(True)
    tyr, tvr = ty.clone().requires_grad_(True), tv.clone().requires_grad_(True)
    lp = O.gaussian(tyr, mu, tvr)
    gx, gmu, gv = torch.autograd.grad(lp, [tyr, mu, tvr], tg)
    assert_close(outs[0], kv, tol)
    assert_close(outs[1], lp, tol)
    assert_close(outs[2], gx, tol if p == "f64" else dict(rtol=1e-3, atol=1e-4))
    assert_close(outs[3], gmu, tol if p == "f64" else dict(rtol=1e-3, atol=1e-4))
    # per-element d/dvar (autograd sums over the broadcast): compare the sum
    assert_close(host(outs[4]).sum().reshape(1), gv, tol if p == "f64" else dict(rtol=1e-3, atol=1e-3))


@pytest.mark.parametrize("kind", ["rbf", "csym"])
def test_gram_symmetric_bwd_and_jitter(H, kind):
    """K(z, z): X2bar == Xbar gives the total point gradient in one pass; diag_add folds the jitter in."""
    rng = np.random.RandomState(5)
    B, n, d = 2, 23, 2
    X = rng.randn(B, n, d)
    ell = np.exp(0.3 * rng.randn(d))
    Kbar = rng.randn(B, n, n)  # deliberately NOT symmetric
    tX, tl = torch.as_tensor(X).requires_grad_(True), torch.as_tensor(ell).requires_grad_(True)
    f = O.rbf_K if kind == "rbf" else O.csym_rbf_K
    K = f(tX, tX, tl)
    gX, gl = torch.autograd.grad((K * torch.as_tensor(Kbar)).sum(), [tX, tl])
    dt = torch.float64
    k = H.KERN_RBF if kind == "rbf" else H.KERN_CSYM_RBF
    Xd, ld, Kd = dev(X, dt), dev(ell, dt), dev(Kbar, dt)
    xb = torch.empty_like(Xd)
    lb = torch.empty_like(ld)
    ws = torch.empty(B * n * d, dtype=dt, device=Xd.device)
    H.gram_bwd_raw(k, Xd, n * d, Xd, n * d, ld, 0, d, Kd, xb, xb, lb, B, n, n, d, ws)
    assert_close(xb, gX, TOL["f64"])
    assert_close(lb, gl, TOL["f64"])
    got = H.gram_fwd(Xd, Xd, ld, kind=k, diag_add=0.25)
    assert_close(got, K.detach() + 0.25 * torch.eye(n, dtype=dt), TOL["f64"])
    # a SYMMETRIC Kbar (what the Cholesky VJP hands over) with the hint that it is: same result, no transposed reads
    Ks = 0.5 * (Kbar + np.transpose(Kbar, (0, 2, 1)))
    Ksd = dev(Ks, dt)
    xb2, lb2 = torch.empty_like(Xd), torch.empty_like(ld)
    H.gram_bwd_raw(k, Xd, n * d, Xd, n * d, ld, 0, d, Ksd, xb, xb, lb, B, n, n, d, ws)
    H.gram_bwd_raw(k | H.KERN_KBAR_SYMMETRIC, Xd, n * d, Xd, n * d, ld, 0, d, Ksd, xb2, xb2, lb2, B, n, n, d, ws)
    assert_close(xb2, xb, TOL["f64"])
    assert_close(lb2, lb, TOL["f64"])


@pytest.mark.parametrize("mode", ["jit", "interpret"])
@pytest.mark.parametrize("p", ["f32", "f64"])
@pytest.mark.parametrize("n", [1, 255, 4097])
def test_ewise_program_sum_outputs(H, p, n, mode):
    """out_regs + EW_PROG_SUM: the sum of a register over the whole space comes out of the same launch."""
    dt = DT[p]
    rng = np.random.RandomState(n)
    a, b = rng.randn(3, n), rng.randn(1, n)
    ins = [dev(a, dt), dev(b, dt)]
    code = [[H.EW["MUL"], 2, 0, 1, 0], [H.EW["ADD"], 3, 2, 0, 0]]
    params = [[0.0, 0.0], [0.0, 0.0]]
    outs = [torch.empty(3, n, dtype=dt, device="cuda"), torch.empty(1, dtype=dt, device="cuda"),
            torch.empty(1, dtype=dt, device="cuda")]
    istr = [[n, 1], [0, 1]]
    ostr = [[n, 1], [0, 0], [0, 0]]
    with _ewise_mode(mode):
        prog = H.EwiseProgram(code, params, ins, istr, outs, [3, 2 + H.EW_PROG_SUM, 3 + H.EW_PROG_SUM], ostr, [3, n])
    prog.launch()
    prod = a * b
    tol = TOL[p] if p == "f64" else dict(rtol=1e-4, atol=1e-3)
    assert_close(outs[0], prod + a, TOL[p])
    assert_close(outs[1], np.array([prod.sum()]), tol)
    assert_close(outs[2], np.array([(prod + a).sum()]), tol)


@pytest.mark.parametrize("M,n", [(512, 8192), (96, 1030), (64, 64)])
def test_likelihood_head_inside_the_forward_contraction(H, M, n):
    """hb_sgp_fwd_gauss + hb_gauss_ll_fold: the Gaussian likelihood head's per-point part (dmu, fbar) computed by the
    forward strip kernel's finishing pass and its three sums folded from per-strip partials -- against hb_sgp_fwd
    followed by hb_gauss_ll_post on the same noise: f, v, dmu and fbar keep their bits, the sums (taken in another fixed
    order) agree to fp32 rounding; with injected and with drawn residual noise."""
    dt = torch.float32
    rng = np.random.RandomState(M + n)
    z = np.sort(rng.uniform(0, M / 2.0, (M, 1)), axis=0)
    x = rng.uniform(0, M / 2.0, (n, 1))
    zz, xx, ell = dev(z, dt), dev(x, dt), dev(np.ones(1), dt)
    K = H.gram_fwd(zz, zz, ell, diag_add=1e-3)
    frag = torch.empty(2 * M * M, dtype=dt, device="cuda")
    L, W, info = H.cholesky_inverse(K, frag=frag)
    u = dev(rng.randn(1, M), dt)
    y = dev(rng.randn(1, n), dt)
    scale, var = dev(np.abs(rng.randn(1)) + 0.5, dt), dev(np.abs(rng.randn(1)) + 0.3, dt)
    eps = dev(rng.randn(n), dt)
    for eps_in, seed in ((eps, None), (None, 7)):
        rngs = [None if seed is None else H.Rng(seed), None if seed is None else H.Rng(seed)]
        units = H.sgp_head_units(xx, zz, u, H.PREC_NATIVE, True, eps_in is None, rngs[0])
        assert units == (n + 31) // 32
        f0, _, v0, e0 = H.sgp_fwd(xx, zz, ell, W, u, eps_in=eps_in, rng=rngs[0], wfrag=frag)
        fb0 = torch.empty(1, n, dtype=dt, device="cuda")
        ll0, dmu0, ds0, dv0 = H.gauss_ll(y, f0, scale, var, post=3.25, fbar=fb0)
        dmu1, fb1 = torch.full((1, n), float("nan"), dtype=dt, device="cuda"), torch.full((1, n), float("nan"), dtype=dt, device="cuda")
        part = torch.full((3 * units,), float("nan"), dtype=dt, device="cuda")
        f1, _, v1, e1 = H.sgp_fwd(xx, zz, ell, W, u, eps_in=eps_in, rng=rngs[1], wfrag=frag,
                                  head=dict(y=y, scale=scale, var=var, post=3.25, dmu=dmu1, fbar=fb1, part=part, units=units))
        ll1, ds1, dv1 = (torch.empty(1, dtype=dt, device="cuda") for _ in range(3))
        H.gauss_ll_fold(part, units, ll1, ds1, dv1)
        for a_, b_ in ((f0, f1), (v0, v1), (e0, e1), (dmu0, dmu1), (fb0, fb1)):
            assert torch.equal(a_.reshape(-1), b_.reshape(-1))
        for a_, b_ in ((ll0, ll1), (ds0, ds1), (dv0, dv1)):
            assert abs(float(a_) - float(b_)) <= 2e-6 * max(abs(float(a_)), float(n) ** 0.5), (float(a_), float(b_))   # observed <= 3e-7 relative


@pytest.mark.parametrize("B,M,d,dl", [(1, 512, 1, 1), (1, 256, 3, 3), (4, 128, 2, 1), (2, 512, 2, 2)])
def test_gram_vjp_inside_the_product_that_computes_kbar(H, B, M, d, dl):
    """hb_matmul_gram_vjp + hb_gram_ell_fold: the last product of the Cholesky VJP (S = T W, symmetric) with the VJP of
    K(X, X) (reference gp/kernels.py:54-101 under TF autodiff) in its epilogue -- against hb_matmul followed by the
    one-pass symmetric hb_gram_bwd on the product's result: the product keeps its bits, Xbar and ellbar agree to fp32
    rounding (other fixed summation order), the counters are left zero (called twice)."""
    dt = torch.float32
    rng = np.random.RandomState(B * 1000 + M + d)
    lead = (B,) if B > 1 else ()
    X = dev(np.sort(rng.uniform(0, 8.0, lead + (M, d)), axis=-2), dt)
    ell = dev(0.7 + rng.rand(*(lead + (dl,))), dt)
    Wl = dev(np.tril(rng.randn(*(lead + (M, M)))) / np.sqrt(M), dt)
    P = rng.randn(*(lead + (M, M)))
    P = dev(P + np.swapaxes(P, -1, -2), dt)
    T1 = H.matmul(Wl, P, transA=True)                      # W^T P
    S0 = H.matmul(T1, Wl)                                  # (W^T P) W : symmetric to rounding
    xb0, lb0 = torch.empty_like(X), torch.empty_like(ell)
    sX = M * d if B > 1 else 0
    sEll = dl if B > 1 else 0
    ws = torch.empty(max(B * M * d, 1), dtype=dt, device="cuda")
    H.gram_bwd_raw(H.KERN_RBF | H.KERN_KBAR_SYMMETRIC, X, sX, X, sX, ell, sEll, dl, S0, xb0, xb0, lb0, B, M, M, d, ws)
    assert H.matmul_gram_vjp_ok(M, M, B, d, dt)
    S1 = torch.full_like(S0, float("nan"))
    xb1, lb1 = torch.full_like(X, float("nan")), torch.full_like(ell, float("nan"))
    lws = torch.full((B * M * d,), float("nan"), dtype=dt, device="cuda")
    part = torch.full((B * (M // 32) ** 2 * 32 * 2 * d,), float("nan"), dtype=dt, device="cuda")
    counters = torch.zeros(B * (M // 32), dtype=torch.int32, device="cuda")
    for _ in range(2):
        H.matmul_gram_vjp(T1, Wl, S1, False, False, X, sX, ell, sEll, dl, d, xb1, lws, part, counters)
        H.gram_ell_fold(lws, M if sEll else B * M, d, dl, B if sEll else 1, lb1)
        torch.cuda.synchronize()
        assert int(counters.abs().sum()) == 0
        assert torch.equal(S0, S1)
        for a_, b_ in ((xb0, xb1), (lb0, lb1)):
            err = float((a_ - b_).abs().max() / a_.abs().max())
            assert err <= 2e-5, err       # observed <= 3e-6
    assert not H.matmul_gram_vjp_ok(M, M, B, 5, dt)


@pytest.mark.parametrize("n,K,N", [(32768, 16, 64), (4096, 32, 96), (2048 + 17, 64, 256), (8192, 128, 33)])
@pytest.mark.parametrize("with_scale,with_bias,post", [(True, True, 3.25), (False, False, None)])
def test_likelihood_head_inside_the_layer_product(H, n, K, N, with_scale, with_bias, post):
    """hb_matmul_gauss + hb_gauss_ll_fold: the Gaussian head of a MatBias layer (reference nn.py:31-32 ->
    densities.py:25-27) computed in the epilogue of the row-streaming product, f never written -- against hb_matmul followed
    by hb_gauss_ll: dmu / fbar and the three sums agree to fp32 rounding (same per-point arithmetic, hb_gauss_point)."""
    dt = torch.float32
    rng = np.random.RandomState(n + K + N)
    x, w = dev(rng.randn(n, K), dt), dev(rng.randn(K, N) / np.sqrt(K), dt)
    b = dev(rng.randn(1, N), dt) if with_bias else None
    y = dev(rng.randn(n, N), dt)
    scale = dev(np.abs(rng.randn(1)) + 0.5, dt) if with_scale else None
    var = dev(np.abs(rng.randn(1)) + 0.3, dt)
    units = H.matmul_gauss_units(n, K, N, dt)
    assert units > 0
    f = H.matmul(x, w, bias=b)
    fb0 = torch.empty(n, N, dtype=dt, device="cuda") if post else None
    ll0, dmu0, ds0, dv0 = H.gauss_ll(y, f, scale, var, **(dict(post=post, fbar=fb0) if post else {}))
    dmu1 = torch.full((n, N), float("nan"), dtype=dt, device="cuda")
    fb1 = torch.full((n, N), float("nan"), dtype=dt, device="cuda") if post else None
    part = torch.full((3 * units,), float("nan"), dtype=dt, device="cuda")
    H.matmul_gauss(x, w, b, dict(y=y, scale=scale, var=var, post=post, dmu=dmu1, fbar=fb1, part=part, units=units))
    ll1, ds1, dv1 = (torch.empty(1, dtype=dt, device="cuda") for _ in range(3))
    H.gauss_ll_fold(part, units, ll1, ds1, dv1)
    # (f = x w + b is rounded once more on its way through memory in the two-launch form and may be contracted differently
    # in the two kernels: dmu agrees to a few ulp of f / var, not bit for bit)
    tol = 4e-7 * float(f.abs().max()) / float(var)
    assert float((dmu0 - dmu1).abs().max()) <= tol, (float((dmu0 - dmu1).abs().max()), tol)
    if post:
        assert float((fb0 - fb1).abs().max()) <= tol * abs(post) * float(scale if scale is not None else 1.0)
    for a_, b_ in ((ll0, ll1), (ds0, ds1), (dv0, dv1)):
        assert abs(float(a_) - float(b_)) <= 3e-6 * max(abs(float(a_)), float(n * N) ** 0.5), (float(a_), float(b_))
    assert H.matmul_gauss_units(1024, K, N, dt) == 0 and H.matmul_gauss_units(n, K, 16, dt) == 0    # shapes it leaves to two launches


@pytest.mark.parametrize("M,n,E", [(512, 8192, 1), (128, 1000, 1), (256, 2048, 2), (64, 64, 1)])
def test_forward_contraction_inside_the_persistent_cholesky_launch(H, M, n, E):
    """Early-start form (hb_sgp_rider_begin; csrc/sgp.hip chol_sgp_fwd_kernel): the forward contraction recorded and
    launched INSIDE the persistent factorisation's grid, taking its row tiles as the rows of W become final -- against the
    two launches it replaces: L, W and the images keep their bits, f / v / A / the head's outputs agree to fp32 rounding
    (the last two row tiles are summed as three partial sums), info is reported, a failed factorisation does not hang, the
    sync words are left zero (the workspace is reused call after call), and a call the factorisation cannot take
    (different image buffer) still runs, behind it."""
    dt = torch.float32
    rng = np.random.RandomState(M + n + E)
    lead = (E,) if E > 1 else ()
    z = dev(np.sort(rng.uniform(0, M / 2.0, lead + (M, 1)), axis=-2), dt)
    x = dev(rng.uniform(0, M / 2.0, (n, 1)), dt)
    ell = dev(np.ones(lead + (1,)), dt)
    u, eps, y = dev(rng.randn(*(lead + (1, M))), dt), dev(rng.randn(*(lead + (n,))), dt), dev(rng.randn(*(lead + (n,))), dt)
    var = dev(np.array([0.3]), dt)
    K = torch.exp(-0.5 * (z - z.transpose(-1, -2)) ** 2) + 1e-3 * torch.eye(M, device="cuda")
    assert H.cholesky_persistent_shape(E, M, dt)

    def run(early, Kmat, other_frag=False):
        L, W = torch.empty_like(Kmat), torch.empty_like(Kmat)
        frag = torch.empty(2 * E * M * M, dtype=dt, device="cuda")
        info = torch.zeros(E, dtype=torch.int32, device="cuda")
        afrag = torch.full((H.sgp_frag_elems(E, n, M, H.PREC_NATIVE),), float("nan"), dtype=dt, device="cuda")
        units = H.sgp_head_units(x, z, u, H.PREC_NATIVE, True, False, None)
        head = dict(y=y, var=var, scale=None, post=0.5, dmu=torch.empty_like(y), fbar=torch.empty_like(y),
                    part=torch.empty(3 * units, dtype=dt, device="cuda"), units=units) if E == 1 else None
        out = (torch.empty(lead + (1, n), device="cuda"), torch.empty(lead + (M, n), device="cuda"),
               torch.empty(lead + (n,), device="cuda"), torch.empty(lead + (n,), device="cuda"))
        fwd = lambda: H.sgp_fwd(x, z, ell, W, u, eps_in=eps, mode=1, out=out, wfrag=frag, prec=H.PREC_NATIVE, a_frag=afrag,
                                skip_a=True, head=head)
        if early:
            assert H.sgp_rider_supported(x, z, u, H.PREC_NATIVE, True, False, None)
            H.sgp_rider_begin()
            fwd()
            assert H.sgp_rider_pending() == 1
            # (other_frag: the factorisation writes its images elsewhere, so it cannot take the recorded forward; the
            # forward then reads an image nobody wrote -- only the sequencing is checked in that case)
            H.cholesky_inverse(Kmat, out=L, inv=W, info=info, frag=torch.empty_like(frag) if other_frag else frag)
            assert H.sgp_rider_pending() == 0
            H.sgp_rider_flush()
        else:
            H.cholesky_inverse(Kmat, out=L, inv=W, info=info, frag=frag)
            fwd()
        torch.cuda.synchronize()
        r = dict(L=L, W=W, frag=frag, info=info, f=out[0], v=out[2], afrag=afrag)
        if head is not None:
            r.update(dmu=head["dmu"], fbar=head["fbar"], part=head["part"])
        return r

    ref, got = run(False, K), run(True, K)
    assert ref["info"].tolist() == [0] * E and got["info"].tolist() == [0] * E
    for k in ("L", "W", "frag"):
        assert torch.equal(ref[k], got[k]), k
    for k in ("f", "v", "afrag") + (("dmu", "fbar", "part") if E == 1 else ()):
        a_, b_ = ref[k].double(), got[k].double()
        err = float((a_ - b_).abs().max() / a_.abs().max())
        # (the two forms sum in different orders -- 16x16x4 batches against 32x32x2, three partial sums in the early form's last
        # two row tiles -- and W K cancels: entries of W reach +-30)
        assert err <= 3e-4, (k, err)     # observed <= 5.4e-5 (f), 2e-6 (A)
    for _ in range(3):                   # the workspace (sync words) is reused call after call
        again = run(True, K)
        assert torch.equal(again["f"], got["f"]) and torch.equal(again["afrag"], got["afrag"])
    Kbad = K.clone()
    Kbad[..., M // 3, M // 3] = -1.0
    gb = run(True, Kbad)
    assert gb["info"].tolist() == [M // 3 + 1] * E
    ok = run(True, K)
    assert ok["info"].tolist() == [0] * E and torch.equal(ok["f"], got["f"])
    run(True, K, other_frag=True)        # not taken: the factorisation, then the forward on its own -- no error, nothing pending
    H.debug_set("sgp_early", 0)          # the switch that keeps the forward a launch of its own
    try:
        assert not H.sgp_rider_supported(x, z, u, H.PREC_NATIVE, True, False, None)
    finally:
        H.debug_clear()


@pytest.mark.parametrize("p", ["f32", "f64"])
@pytest.mark.parametrize("n", [300, 8192, 40000])
def test_gauss_ll_head_writes_the_gradient_for_f_itself(H, p, n):
    """hb_gauss_ll_post: the likelihood head (reference densities.py:25-27 under tf.reduce_sum + TF autodiff) also
    leaves fbar = scale * (post * dmu) -- what the two elementwise ops behind it computed in a launch of their own --
    in all three forms (one-workgroup head, partial sums + finish, with and without a scale); the other outputs keep
    their bits."""
    dt = DT[p]
    rng = np.random.RandomState(n)
    x, f = dev(rng.randn(1, n), dt), dev(rng.randn(1, n), dt)
    var = dev(np.abs(rng.randn(1)) + 0.3, dt)
    for scale in (dev(np.abs(rng.randn(1)) + 0.5, dt), None):
        ref = H.gauss_ll(x, f, scale, var)
        fbar = torch.full((1, n), float("nan"), dtype=dt, device="cuda")
        got = H.gauss_ll(x, f, scale, var, post=122.07, fbar=fbar)
        for a_, b_ in zip(ref, got):
            assert torch.equal(a_, b_)
        want = (1.0 if scale is None else scale.double()) * (122.07 * ref[1].double())
        assert_close(fbar, want, dict(rtol=1e-12, atol=0) if p == "f64" else dict(rtol=3e-7, atol=0))


@pytest.mark.parametrize("n,din,hid,act", [(4096, 64, 256, "sigmoid"), (256, 32, 128, "tanh"), (1024, 64, 128, "relu"),
                                             (32, 32, 256, "sigmoid")])
@pytest.mark.parametrize("inject", [True, False])
def test_fused_encoder_forward_and_backward(H, n, din, hid, act, inject):
    """hb_mlp2_sample_fwd / _bwd (csrc/mlp.hip): the two-layer encoder (reference nn.py:31-32,73-84) feeding a LOCAL diagonal
    Normal (variationals.py:121-129,138-142,225-230) in one launch per direction, the hidden layer never written -- against
    the oracle's op-by-op evaluation (neural_net, feed_split, sample_diag, kl_normal) and torch autograd, in fp64."""
    rng = np.random.RandomState(n + din)
    L = 16
    y = rng.randn(n, din)
    w0, b0 = rng.randn(din, hid) / np.sqrt(din), 0.1 * rng.randn(1, hid)
    w1, b1 = rng.randn(hid, 2 * L) / np.sqrt(hid), 0.1 * rng.randn(1, 2 * L)
    d32 = lambda a: dev(a, torch.float32)
    Y, W0, B0, W1, B1 = d32(y), d32(w0), d32(b0), d32(w1), d32(b1)
    if inject:
        u = rng.randn(n, L)
        x, kl, uo, o = H.mlp2_sample_fwd(Y, W0, B0, W1, B1, act, u_in=d32(u))
        assert torch.equal(uo, d32(u))
    else:
        if not H.mlp2_sample_supported(n, din, hid, 32, 65536, False):
            pytest.skip("needs 2 n generator lanes")
        r1, r2 = H.Rng(5, stream_id=3), H.Rng(5, stream_id=3)
        x, kl, uo, o = H.mlp2_sample_fwd(Y, W0, B0, W1, B1, act, rng=r1)
        x2, kl2, uo2, o2 = H.mlp2_sample_fwd(Y, W0, B0, W1, B1, act, rng=r2)
        assert torch.equal(uo, uo2) and torch.equal(x, x2) and torch.equal(kl, kl2)          # a pure function of the state
        u = host(uo)
        assert abs(u.mean()) < 5.0 / np.sqrt(u.size) and abs(u.std() - 1.0) < 0.02 + 5.0 / np.sqrt(u.size)
        x3 = H.mlp2_sample_fwd(Y, W0, B0, W1, B1, act, rng=r1)[0]
        assert not torch.equal(x, x3)                                                       # and the state has advanced
    actf = {"sigmoid": torch.sigmoid, "tanh": torch.tanh, "relu": torch.relu}[act]
    T = lambda a_: torch.as_tensor(a_, dtype=torch.float64)
    tw0, tb0, tw1, tb1 = [T(a_).clone().requires_grad_(True) for a_ in (w0, b0, w1, b1)]
    oo = actf(T(y) @ tw0 + tb0) @ tw1 + tb1
    mu, sv = O.feed_split(oo, [L, L])
    xs = O.sample_diag(mu, sv, T(u))
    klr = O.kl_normal(sv, T(u), xs, "diagonal")
    assert_close(o, oo.detach(), dict(rtol=2e-4, atol=2e-5))
    assert_close(x, xs.detach(), dict(rtol=2e-4, atol=5e-5))
    assert abs(kl.item() - klr.item()) <= 2e-5 * abs(klr.item()) + 1e-3
    # backward: gradients of  sum(xbar * x) + klbar * kl
    xbar, klbar = rng.randn(n, L) / np.sqrt(n), np.array([-1.0])
    (xs * T(xbar)).sum().add(klr * klbar[0]).backward()
    dw0, db0, dw1, db1 = H.mlp2_sample_bwd(Y, W0, B0, W1, act, o, uo, x, d32(xbar), d32(klbar))
    for got, ref, name in ((dw0, tw0.grad, "dw0"), (db0, tb0.grad.reshape(-1), "db0"), (dw1, tw1.grad, "dw1"),
                           (db1, tb1.grad.reshape(-1), "db1")):
        observe("mlp2_fused[%d,%d,%d,%s]/%s" % (n, din, hid, act, name), tile_err(host(got), ref.numpy()), 5e-6)    # observed 6.6e-8 .. 6.0e-7 over the eight cases
    # only the KL gradient / only the sample gradient
    dw0k = H.mlp2_sample_bwd(Y, W0, B0, W1, act, o, uo, x, None, d32(klbar))[0]
    dw0x = H.mlp2_sample_bwd(Y, W0, B0, W1, act, o, uo, x, d32(xbar), None)[0]
    assert tile_err(host(dw0k + dw0x), host(dw0)) <= 2e-5      # linear in (xbar, klbar), up to fp32 summation order


def test_serial_chain_flushes_before_its_argument_block_is_exhausted(H):
    """hb_chain_push counts the argument slots of the recorded jobs and runs them before a job that would not fit
    (five 9-pointer likelihood heads leave 19 pointer slots; the generated kernel wants 20 free at the start of a job):
    the six calls below used to fail inside hb_chain_flush with 'argument block exhausted', after the jobs were dropped."""
    rng = np.random.RandomState(5)
    n = 1000
    xs = [dev(rng.randn(1, n), torch.float32) for _ in range(6)]
    fs = [dev(rng.randn(1, n), torch.float32) for _ in range(6)]
    var = dev(np.array([0.7]), torch.float32)
    ref = [H.gauss_ll(x, f, None, var) for x, f in zip(xs, fs)]
    outs = [tuple(torch.full_like(t, float("nan")) for t in r) for r in ref]
    H.chain_begin()
    try:
        for x, f, o in zip(xs, fs, outs):
            H.gauss_ll(x, f, None, var, out=o)
        H.chain_end()
    finally:
        H.chain_discard()
    torch.cuda.synchronize()
    for r, o in zip(ref, outs):
        for k in (0, 1, 3):      # ll, dmu, dvar (dscale is not written without a scale)
            assert torch.equal(r[k], o[k]), k


@pytest.mark.parametrize("p", ["f32", "f64"])
@pytest.mark.parametrize("R,n", [(4, 3001), (2, 64), (7, 777)])
def test_column_program_softmax_gate(H, p, R, n):
    """hb_ewise_colprog_*: the softmax gate of the expert mixture (reference notebooks/Expert_GPR.ipynb:139-147) as ONE
    launch over an [R, n] space -- row-block slices of a taller source read in place, tf.reduce_max / reduce_sum over
    axis 0 as in-thread row loops, a scalar operand, a [1, n] result, an [R, n] result and a broadcast store of a
    [1, n] value over the R rows -- against the oracle's op-by-op torch evaluation."""
    dt = DT[p]
    rng = np.random.RandomState(R * 1000 + n)
    f_all = rng.randn(2 * R, 1, n)
    kr = np.abs(rng.randn(1)) + 0.5
    src = dev(f_all, dt)
    E = H.EW
    # r0 = f_all[:R], r1 = f_all[R:], r2 = kr ; g = r1 * sqrt(kr); m = max_rows g; w = exp(g - m); s = sum_rows w;
    # wn = w / s; f = sum_rows(wn * r0)
    code = [[E["SQRT"], 3, 2, -1, -1], [E["MUL"], 4, 1, 3, -1], [H.COLPROG_MAX, 5, 4, -1, -1], [E["SUB"], 6, 4, 5, -1],
            [E["EXP"], 7, 6, -1, -1], [H.COLPROG_SUM, 8, 7, -1, -1], [E["DIV"], 9, 7, 8, -1], [E["MUL"], 10, 9, 0, -1],
            [H.COLPROG_SUM, 11, 10, -1, -1]]
    params = [[0.0, 0.0]] * len(code)
    ins = [(src, 0, n, 1), (src, R * n, n, 1), (dev(kr, dt), 0, 0, 0)]
    wn = torch.empty(R, n, dtype=dt, device="cuda")
    f = torch.empty(1, n, dtype=dt, device="cuda")
    mb = torch.empty(R, n, dtype=dt, device="cuda")     # the row maxima broadcast over the rows
    sq = torch.full((1,), float("nan"), dtype=dt, device="cuda")
    outs = [(wn, 0, n, 1), (f, 0, 0, 1), (mb, 0, n, 1), (sq, 0, 0, 0)]
    prog = H.ColProgram(code, params, ins, outs, [9, 11, 5, 3], R, n)
    assert "for (int q = 0; q < %d; ++q)" % R in prog.source
    prog.launch()
    t = O.T(f_all)
    fe, ge = t[:R, 0, :], t[R:, 0, :] * O.T(kr).sqrt()
    gm = ge.max(0, keepdim=True).values
    w = (ge - gm).exp()
    w = w / w.sum(0, keepdim=True)
    fr = (w * fe).sum(0, keepdim=True)
    tol = TOL[p] if p == "f64" else dict(rtol=2e-6, atol=2e-6)
    assert_close(wn, w, tol)
    assert_close(f, fr, tol)
    assert_close(mb, gm.expand(R, n), tol)
    assert_close(sq, O.T(kr).sqrt(), tol)


@pytest.mark.parametrize("p", ["f32", "f64"])
def test_ewise_program_compiled_equals_interpreted(H, p):
    """The run-time compiled form of a program (hb_ewise_jit_*) against the interpreted one (hb_ewise_prog_run_*): both
    evaluate the library's own ew_apply op by op, so a chain through the arithmetic part of the op table (broadcast
    inputs, a broadcast index-0-writes output, sum-reduced outputs) returns the SAME BITS; ops that go through the
    device math library's log (LOG, LOG1P, SOFTPLUS, LGAMMA, DIGAMMA, POW, the Gaussian log-density) may differ in
    the last place -- the backend's expansion of llvm.log rounds differently from one compilation context to the
    next (tools/jit_vs_interp.py: 1.2e-7 relative in LOG, 1.9e-5 in the digamma series, fp32) -- and are compared at
    a few ulp."""
    dt = DT[p]
    rng = np.random.RandomState(11)
    R, n = 3, 517
    a, b, c = np.abs(rng.randn(R, n)) + 0.3, rng.randn(1, n), rng.randn(R, 1)
    ins = [dev(a, dt), dev(b, dt), dev(c, dt)]
    istr = [[n, 1], [0, 1], [1, 0]]
    E = H.EW
    exact_unary = ["NEG", "SQRT", "SQUARE", "ABS", "SIGN", "RELU", "RECIP", "RSQRT", "STEP", "AFFINE", "CLIP", "CLIPMASK", "COPY"]
    libm_unary = ["EXP", "LOG", "SIGMOID", "SOFTPLUS", "TANH", "LGAMMA", "POWC", "LOG1P", "DIGAMMA"]
    binary = ["ADD", "SUB", "MUL", "DIV", "MAX", "MIN", "GT", "GE", "LT", "LE", "EQ", "SIGMOID_GRAD", "TANH_GRAD",
              "RELU_GRAD", "SOFTPLUS_GRAD", "CLIP_GRAD"]

    def both(code, params, oregs, ostr, oshapes):
        res = []
        for mode in ("jit", "interpret"):
            outs = [torch.empty(*sh, dtype=dt, device="cuda") for sh in oshapes]
            with _ewise_mode(mode):
                prog = H.EwiseProgram(code, params, ins, istr, outs, oregs, ostr, [R, n])
            assert (prog.image is None) == (mode == "jit")
            if mode == "jit":
                assert "ew_apply<T>(" in prog.source
            prog.launch()
            torch.cuda.synchronize()
            res.append([torch.nan_to_num(o, nan=123.0) for o in outs])
        return res

    def chain(unary):
        code, params, reg, acc = [], [], 3, 0
        for f in unary:
            code.append([E[f], reg, 0, 0, 0]); params.append([0.7, 1.3] if f in ("AFFINE", "CLIP", "CLIPMASK") else [1.5, 0.0])
            code.append([E["FMA"], reg + 1, reg, 2, acc if acc else 1]); params.append([0.0, 0.0])
            acc = reg + 1
            reg += 2
        return code, params, acc

    # 1. arithmetic ops: same bits (incl. the broadcast output and the sum-reduced output)
    code, params, acc = chain(exact_unary)
    for f in binary:
        code.append([E[f], acc + 1, acc, 1, 0]); params.append([-0.5, 0.5])
        code.append([E["FMA"], acc + 2, acc + 1, 2, acc]); params.append([0.0, 0.0])
        acc += 2
        if acc > 36:
            break
    code.append([E["WHERE"], acc + 1, 2, acc, 1]); params.append([0.0, 0.0])
    acc += 1
    ja, ia = both(code, params, [acc, 4, acc + H.EW_PROG_SUM], [[n, 1], [1, 0], [0, 0]], [(R, n), (R, 1), (1,)])
    for x, y in zip(ja, ia):
        assert torch.equal(x, y)
    # 2. ops through the math library: a few ulp
    tol = dict(rtol=2e-4, atol=2e-5) if p == "f32" else dict(rtol=1e-11, atol=1e-12)
    code, params, acc = chain(libm_unary)
    jb, ib = both(code, params, [acc, 4, acc + H.EW_PROG_SUM], [[n, 1], [1, 0], [0, 0]], [(R, n), (R, 1), (1,)])
    for x, y in zip(jb, ib):
        assert_close(x, y, tol)
    # 3. the Gaussian log-density and its 4-input, 3-output gradient op
    code = [[E["GAUSS_LOGPDF"], 3, 1, 2, 0], [E["GAUSS_LOGPDF_GRAD"], 4, 1, 2, 0]]
    params = [[0.0, 0.0], [3.0, 0.0]]
    jc, ic = both(code, params, [3, 4, 5, 6], [[n, 1]] * 4, [(R, n)] * 4)
    for x, y in zip(jc, ic):
        assert_close(x, y, tol)


@pytest.mark.parametrize("p", ["f32", "f64"])
def test_gather_rows_multi(H, p):
    """Several arrays gathered by one index vector in one launch == one gather per array."""
    dt = DT[p]
    rng = np.random.RandomState(7)
    N, n = 50, 33
    srcs = [rng.randn(N, 1), rng.randn(N, 5), rng.randn(N, 2, 3)]
    idx = rng.randint(0, 40, n)
    perm = rng.permutation(N)[:40]
    dsrc = [dev(s, dt) for s in srcs]
    outs = [torch.empty((n,) + s.shape[1:], dtype=dt, device="cuda") for s in srcs]
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    di, dp = torch.as_tensor(idx).cuda(), torch.as_tensor(perm).cuda()
    mg = H.MultiGather(dsrc, outs, di, dp, err)
    mg.launch()
    for d, o in zip(dsrc, outs):
        assert np.array_equal(host(o), host(d)[perm[idx]])
    mg.launch(use_perm=False)
    for d, o in zip(dsrc, outs):
        assert np.array_equal(host(o), host(d)[idx])
    assert err.item() == 0
    bad = torch.as_tensor(np.array([0, N + 3] + [1] * (n - 2))).cuda()
    H.MultiGather(dsrc, outs, bad, None, err).launch(use_perm=False)
    assert err.item() == 1 and np.all(host(outs[1])[1] == 0)


@pytest.mark.parametrize("p", ["f32", "f64"])
def test_diag_sample_kl_on_column_blocks_in_place(H, p):
    """The sampler / its VJP reading mu, s as column blocks of a wider [rows, 2L] matrix and writing both gradients as
    the two halves of one [rows, 2L] matrix == the dense form on sliced copies (the encoder-fed LOCAL posterior,
    reference variationals.py:70-80)."""
    dt = DT[p]
    rng = np.random.RandomState(2)
    rows, L = 37, 5
    enc = rng.randn(rows, 2 * L) * 0.3
    u = rng.randn(rows, L)
    xbar, klbar = rng.randn(rows, L), np.array([0.7])
    d_enc, d_u = dev(enc, dt), dev(u, dt)
    mu_c, s_c = dev(enc[:, :L].copy(), dt), dev(enc[:, L:].copy(), dt)
    x0, kl0, _ = H.diag_sample_kl_fwd(mu_c, s_c, u_in=d_u)
    out = (torch.empty(rows, L, dtype=dt, device="cuda"), torch.empty(1, dtype=dt, device="cuda"),
           torch.empty(rows, L, dtype=dt, device="cuda"))
    flat = d_enc.reshape(-1)
    x1, kl1, u1 = H.diag_sample_kl_fwd(flat[0:], flat[L:], u_in=d_u, out=out, rows=(rows, L, 2 * L, 2 * L))
    assert torch.equal(x0, x1) and torch.equal(u1, d_u)
    assert_close(kl1, host(kl0), TOL[p] if p == "f64" else dict(rtol=1e-5, atol=1e-4))
    # halves given the other way round (log-std first), in-kernel noise: same stream as the dense call
    r1, r2 = H.Rng(seed=3, nlanes=256), H.Rng(seed=3, nlanes=256)
    xa, kla, ua = H.diag_sample_kl_fwd(s_c, mu_c, rng=r1)
    xb, klb, ub = H.diag_sample_kl_fwd(flat[L:], flat[0:], rng=r2, out=out, rows=(rows, L, 2 * L, 2 * L))
    assert torch.equal(xa, xb) and torch.equal(ua, ub) and torch.equal(r1.state, r2.state)
    # VJP: s in place, gradients packed
    mb0, sb0 = H.diag_sample_kl_bwd(s_c, d_u, x0, dev(xbar, dt), dev(klbar, dt))
    g = torch.full((rows, 2 * L), float("nan"), dtype=dt, device="cuda")
    gf = g.reshape(-1)
    H.diag_sample_kl_bwd(flat[L:], d_u, x0, dev(xbar, dt), dev(klbar, dt), out=(gf[0:], gf[L:]), rows=(rows, L, 2 * L, 2 * L))
    assert torch.equal(g[:, :L], mb0) and torch.equal(g[:, L:], sb0)
    with pytest.raises(Exception, match="row layout"):
        H.diag_sample_kl_fwd(flat[0:], flat[L:], u_in=d_u, out=out, rows=(rows, L, L - 1, 2 * L))


@pytest.mark.parametrize("p", ["f32", "f64"])
@pytest.mark.parametrize("widths", [(1, 1), (1, 5, 6), (64,), (20, 3)])
def test_gather_rows_multi_with_in_kernel_draw(H, p, widths):
    """Index draw + gather in one launch == hb_rng_randint followed by the gather: same indices, same rows, same
    RNG state afterwards (so a plan may switch between the two forms freely)."""
    dt = DT[p]
    rng = np.random.RandomState(11)
    N, n, hi = 300, 257, 250
    srcs = [rng.randn(N, w) for w in widths]
    perm = rng.permutation(N)[:hi]
    dsrc = [dev(s, dt) for s in srcs]
    dp = torch.as_tensor(perm).cuda()
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    # reference: two launches
    r1 = H.Rng(seed=5, stream_id=2, nlanes=512)
    idx1 = torch.zeros(n, dtype=torch.int64, device="cuda")
    outs1 = [torch.empty((n, w), dtype=dt, device="cuda") for w in widths]
    r1.randint(n, 0, hi, out=idx1)
    H.MultiGather(dsrc, outs1, idx1, dp, err).launch()
    # fused
    r2 = H.Rng(seed=5, stream_id=2, nlanes=512)
    idx2 = torch.zeros(n, dtype=torch.int64, device="cuda")
    outs2 = [torch.empty((n, w), dtype=dt, device="cuda") for w in widths]
    mg = H.MultiGather(dsrc, outs2, idx2, dp, err)
    mg.launch_draw(r2, 0, hi)
    assert torch.equal(idx1, idx2) and torch.equal(r1.state, r2.state)
    for a, b, d in zip(outs1, outs2, dsrc):
        assert torch.equal(a, b)
        assert np.array_equal(host(b), host(d)[perm[host(idx2).astype(np.int64)]])
    assert err.item() == 0 and 0 <= int(idx2.min()) and int(idx2.max()) < hi
    # a second draw continues the stream; without the permutation the raw indices are used
    r1.randint(n, 0, hi, out=idx1)
    mg.launch_draw(r2, 0, hi, use_perm=False)
    assert torch.equal(idx1, idx2) and torch.equal(r1.state, r2.state)
    assert np.array_equal(host(outs2[0]), host(dsrc[0])[host(idx2).astype(np.int64)])
    with pytest.raises(Exception, match="RNG lanes"):
        H.MultiGather(dsrc, outs2, idx2, dp, err).launch_draw(H.Rng(seed=1, nlanes=128), 0, hi)


@pytest.mark.parametrize("p", ["f32", "f64"])
@pytest.mark.parametrize("n,scaled", [(1, True), (1000, True), (8192, False), (5001, True), (16384, True), (16385, False), (100003, True)])
def test_gauss_ll_fused(H, p, n, scaled):
    """hb_gauss_ll: sum of log N(x | f*scale, var) and the pieces of its gradient == the oracle's density + autograd."""
    dt = DT[p]
    rng = np.random.RandomState(n)
    x, f = rng.randn(1, n), rng.randn(1, n)
    s, v = np.array([1.7]), np.array([0.6])
    tf_ = torch.as_tensor(f).requires_grad_(True)
    ts, tv = torch.as_tensor(s).requires_grad_(True), torch.as_tensor(v).requires_grad_(True)
    mu = tf_ * ts if scaled else tf_
    ll = O.gaussian(torch.as_tensor(x), mu, tv).sum()
    gmu, = torch.autograd.grad(ll, [mu], retain_graph=True)
    gs, gv = torch.autograd.grad(ll, [ts, tv], allow_unused=True)
    got = H.gauss_ll(dev(x, dt), dev(f, dt), dev(s, dt) if scaled else None, dev(v, dt))
    tol = TOL[p] if p == "f64" else dict(rtol=2e-4, atol=2e-3)
    assert_close(got[0], ll.detach().reshape(1), tol)
    assert_close(got[1], gmu, TOL[p] if p == "f64" else dict(rtol=1e-4, atol=1e-5))
    if scaled:
        assert_close(got[2], gs.reshape(1), tol)
    assert_close(got[3], gv.reshape(1), tol)
