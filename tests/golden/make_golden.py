"""Generate tests/golden/reference_known_answers.npz.

The reference's tests hold no data files: their known answers are inline numpy
formulas evaluated on `np.random.RandomState(0)` draws.  This script re-derives
those numbers -- same seeds, same draw order, independent loop-style numpy
formulas -- so the oracle (oracle/henbun_oracle.py) and the HIP kernels can be
checked against them without the reference (or TensorFlow) being present.

Sources of the formulas/seeds (all under /root/reference/testing):
  kernels      test_kernels.py:10-63 (RefStationary/RefRBF/RefCsymRBF), :66-88 (draws)
  variationals test_variationals.py:30-52 (draws), :69-106 (logdet, projected
               samples), :326-347 (gaussian_KL)
  sparse gp    test_gp.py:59-66 (fixture), :68-91, :115-131
  densities    test_densities.py:11-32
  log_sum_exp  test_tf_wraps.py:45-59
  transforms   test_transforms.py:39-53

Run:  python tests/golden/make_golden.py   (numpy + scipy only)
"""
import os

import numpy as np
from scipy.special import loggamma

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_known_answers.npz")


def sqdist_loops(X, X2, ell):
    if X.ndim == 3:
        out = np.zeros((X.shape[0], X.shape[1], X2.shape[1]))
        for b in range(X.shape[0]):
            for i in range(X.shape[1]):
                for j in range(X2.shape[1]):
                    dif = (X[b, i] - X2[b, j]) / ell
                    out[b, i, j] = np.sum(dif * dif)
        return out
    out = np.zeros((X.shape[0], X2.shape[0]))
    for i in range(X.shape[0]):
        for j in range(X2.shape[0]):
            dif = (X[i] - X2[j]) / ell
            out[i, j] = np.sum(dif * dif)
    return out


def rbf(X, X2, ell):
    return np.exp(-0.5 * sqdist_loops(X, X2, ell))


def csym(X, X2, ell):
    return np.exp(-0.5 * sqdist_loops(X, X2, ell)) + np.exp(-0.5 * sqdist_loops(X, -X2, ell))


def csym_diag(X, ell):
    Xt = np.sum((X / ell) ** 2, axis=-1)
    return 1.0 + np.exp(-2.0 * Xt)


def main():
    g = {}
    # ---- kernels (test_kernels.py:66-88) ----
    rng = np.random.RandomState(0)
    l1 = np.exp(rng.randn(1))
    l2 = np.exp(rng.randn(2))
    X = rng.randn(5, 2)
    X2 = rng.randn(6, 2)
    Xb = rng.randn(10, 5, 2)
    X2b = rng.randn(10, 6, 2)
    g.update(k_l1=l1, k_l2=l2, k_X=X, k_X2=X2, k_Xb=Xb, k_X2b=X2b)
    g["k_rbf1_XX"] = rbf(X, X, l1)
    g["k_rbf2_XX"] = rbf(X, X, l2)
    g["k_csym_XX"] = csym(X, X, l1)
    g["k_rbf1_XX2"] = rbf(X, X2, l1)
    g["k_rbf2_XX2"] = rbf(X, X2, l2)
    g["k_csym_XX2"] = csym(X, X2, l1)
    g["k_rbf1_b"] = rbf(Xb, Xb, l1)
    g["k_rbf2_b"] = rbf(Xb, Xb, l2)
    g["k_csym_b"] = csym(Xb, Xb, l1)
    g["k_rbf1_b2"] = rbf(Xb, X2b, l1)
    g["k_rbf2_b2"] = rbf(Xb, X2b, l2)
    g["k_csym_b2"] = csym(Xb, X2b, l1)
    g["k_csym_diag"] = csym_diag(X, l1)
    g["k_csym_diag_b"] = csym_diag(Xb, l1)

    # ---- variationals (test_variationals.py:30-52) ----
    rng = np.random.RandomState(0)
    sq_full = rng.randn(3, 10, 10) * 0.5
    sq_diag = rng.randn(3, 10) * 0.5 - 0.5
    for i in range(3):
        for j in range(10):
            sq_full[i, j, j] = np.exp(sq_full[i, j, j])
            for k in range(j + 1, 10):
                sq_full[i, j, k] = 0.0
    vx = rng.randn(3, 10) * 0.3
    iid = rng.randn(3, 10).astype(np.float32).astype(np.float64)
    g.update(v_sq_full=sq_full, v_sq_diag=sq_diag, v_mu=vx, v_iid=iid)
    ld_full = np.zeros((3, 10))
    ld_diag = np.zeros((3, 10))
    post_full = np.zeros((3, 10))
    post_diag = np.zeros((3, 10))
    for i in range(3):
        for j in range(10):
            ld_full[i, j] = 2.0 * np.log(sq_full[i, j, j])
            ld_diag[i, j] = 2.0 * sq_diag[i, j]
        post_full[i] = vx[i] + np.dot(sq_full[i], iid[i])
        post_diag[i] = vx[i] + np.exp(sq_diag[i]) * iid[i]
    g.update(v_logdet_full=ld_full, v_logdet_diag=ld_diag, v_post_full=post_full, v_post_diag=post_diag)
    # analytic KL (test_variationals.py:326-347)
    kl_full = 0.0
    kl_diag = 0.0
    for i in range(3):
        n = 10
        kl_full += 0.5 * (-np.sum(np.log(np.square(np.diagonal(sq_full[i])))) - n + np.sum(np.square(sq_full[i])) + vx[i] @ vx[i])
        kl_diag += 0.5 * (-2.0 * np.sum(sq_diag[i]) - n + np.sum(np.exp(2.0 * sq_diag[i])) + vx[i] @ vx[i])
    g.update(v_kl_full=np.array(kl_full), v_kl_diag=np.array(kl_diag))

    # ---- sparse GP fixture (test_gp.py:59-66,115-131) ----
    rng = np.random.RandomState(0)
    z = np.linspace(-2.0, 2.0, 60).reshape(-1, 2)
    ell = np.ones(1) * 0.5
    xg = rng.randn(20, 2)
    jitter = 1e-5
    Kzz = rbf(z, z, ell) + jitter * np.eye(30)
    Lz = np.linalg.cholesky(Kzz)
    Kzx = rbf(z, xg, ell)
    LnT = np.linalg.solve(Lz, Kzx)
    cov_full = rbf(xg, xg, ell) - LnT.T @ LnT
    cov_diag = 1.0 - np.sum(LnT * LnT, axis=0)
    g.update(g_z=z, g_ell=ell, g_x=xg, g_cholT=Lz.T, g_LnT=LnT, g_cov_full=cov_full, g_cov_diag=cov_diag)

    # ---- densities (test_densities.py:11-32) ----
    rng = np.random.RandomState(0)
    a = rng.randn(2, 3, 4)
    b = rng.randn(2, 3, 4)
    frac = rng.uniform(size=(2, 1, 1))

    def st_ref(x, mu, scale, nu):
        const = loggamma(0.5 * (nu + 1.0)) - loggamma(0.5 * nu) - 0.5 * (np.log(scale * scale) + np.log(nu) + np.log(np.pi))
        return const - 0.5 * (nu + 1.0) * np.log(1.0 + (1.0 / nu) * ((x - mu) / scale) ** 2.0)

    lp0 = -0.5 * np.log(2 * np.pi) - 0.5 * np.log(2.0) - 0.5 * (0.0 - a) ** 2 / 2.0
    lp1 = st_ref(b, 0.0, 2.0, 3.0)
    g.update(d_a=a, d_b=b, d_frac=frac, d_logp0=lp0, d_logp1=lp1)
    g["d_mix"] = np.log(frac * np.exp(lp0) + (1 - frac) * np.exp(lp1))
    rng = np.random.RandomState(0)
    sx = rng.randn(2, 3, 4)
    smu = rng.randn(2, 3, 4)
    sscale = np.exp(rng.randn(2, 3, 4))
    snu = np.exp(rng.randn(2, 3, 4))
    g.update(s_x=sx, s_mu=smu, s_scale=sscale, s_nu=snu)
    g["s_logp_nu3"] = st_ref(sx, smu, sscale, 3.0)
    g["s_logp_nuT"] = st_ref(sx, smu, sscale, snu)

    # ---- log_sum_exp (test_tf_wraps.py:45-59) ----
    rng = np.random.RandomState(0)
    t = rng.randn(3, 4, 5)
    g["lse_in"] = t
    g["lse_axis1"] = np.log(np.sum(np.exp(t), axis=1))
    g["lse_axis2"] = np.log(np.sum(np.exp(t), axis=2))

    # ---- transforms (test_transforms.py:39-53): Log1pe forward values ----
    rng = np.random.RandomState(0)
    tx = rng.randn(10)
    g["t_x"] = tx
    g["t_log1pe"] = np.log(1.0 + np.exp(tx)) + 1e-6

    # ---- MLP (test_nn.py:11-29 shape pattern; weights drawn here) ----
    rng = np.random.RandomState(0)
    nx = rng.randn(5, 6, 3)
    w1 = rng.randn(5, 3, 2)
    b1 = rng.randn(5, 1, 2)
    w2 = rng.randn(5, 2, 4)
    b2 = rng.randn(5, 1, 4)
    h = 1.0 / (1.0 + np.exp(-(np.einsum("lni,lio->lno", nx, w1) + b1)))
    g.update(n_x=nx, n_w1=w1, n_b1=b1, n_w2=w2, n_b2=b2)
    g["n_y"] = np.einsum("lni,lio->lno", h, w2) + b2

    np.savez_compressed(OUT, **g)
    print("wrote", OUT, "(%d arrays)" % len(g))


if __name__ == "__main__":
    main()
