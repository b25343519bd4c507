"""CPU oracle: restatement of the reference's ELBO-path arithmetic.

TEST INFRASTRUCTURE ONLY.  Nothing under henbun_amd/ imports this module; it is
used by tests/, by __graft_entry__.smoke() as the checker, and by bench.py's
`cpu_baseline` leg (kind "port").  The product path never routes through it.

Every function restates, in the reference's own (unfused) op order, what the
reference asks TensorFlow to compute; citations are to /root/reference/Henbun.
The arithmetic itself lives in TensorFlow 1.x (`tensorflow>=1.0`, unpinned,
reference setup.py:34-37), which is absent here (`import Henbun` fails with an
ordinary ModuleNotFoundError at transforms.py:20), so the restatement uses
torch-CPU tensors (float64 by default) -- which also gives reverse-mode
gradients to stand in for `optimizer.minimize`'s TF autodiff (model.py:220).

Pinning.  tests/test_oracle.py checks these functions against the reference's
OWN numpy oracles, executed (not re-typed) in the build container by
tests/golden/make_golden.py: it parses testing/test_kernels.py (RefStationary /
RefRBF / RefCsymRBF, :10-63), testing/test_variationals.py (gaussian_KL,
:326-347), testing/test_densities.py (student_t_ref, :26-32) and
Henbun/transforms.py (Identity / Exp / Log1pe numpy forward/backward, :73-143),
runs those definitions on the reference tests' seeds and draw order and stores
the arrays in tests/golden/reference_known_answers.npz (file hashes and line
ranges recorded in the fixture's `_provenance`; regeneration is byte-identical
and checked by a test).  Known answers that exist only as inline numpy inside
TF test bodies (test_variationals.py:69-106, test_gp.py:59-131,
test_densities.py:23, test_tf_wraps.py:45-59, test_nn.py:11-29) are restated
by that script.  Autograd gradients are cross-checked with central finite
differences.  Gradient VALUES, whole-ELBO values and Adam trajectories are not
pinned by any reference test (its gradient tests assert existence
only): for those, parity is "oracle-pinned", the oracle's forward pieces being
reference-pinned as above.
"""
from __future__ import annotations

import math

import numpy as np
import torch

DT = torch.float64
LOG2PI = math.log(2.0 * math.pi)


def T(x, dtype=DT):
    if isinstance(x, torch.Tensor):
        return x.to(dtype)
    return torch.as_tensor(np.asarray(x), dtype=dtype)


# --------------------------------------------------------------------------
# transforms.py
# --------------------------------------------------------------------------
def log1pe_forward(x, lower=1e-6):
    """transforms.py:133-134  tf.nn.softplus(x) + lower."""
    return torch.nn.functional.softplus(x) + lower


def log1pe_backward_np(y, lower=1e-6):
    """transforms.py:139-140  numpy inverse used by deferred assignment (param.py:247)."""
    return np.log(np.exp(np.asarray(y, dtype=np.float64) - lower) - 1.0)


def log1pe_log_jacobian(x):
    """transforms.py:136-137."""
    return -torch.sum(torch.log(1.0 + torch.exp(-x)))


def logistic_forward(x, a=0.0, b=1.0):
    """transforms.py:153-155."""
    return a + (b - a) / (1.0 + torch.exp(-x))


# --------------------------------------------------------------------------
# tf_wraps.py
# --------------------------------------------------------------------------
def clip(t, enabled=False, vmin=-50.0, vmax=50.0):
    """tf_wraps.py:33-39 (config gated clip_by_value; default off, henbunrc:12-14)."""
    return torch.clamp(t, vmin, vmax) if enabled else t


def log_sum_exp(t, axis=-1):
    """tf_wraps.py:42-48."""
    m = torch.amax(t, dim=axis, keepdim=True)
    return m.squeeze(axis) + torch.log(torch.sum(torch.exp(t - m), dim=axis))


# --------------------------------------------------------------------------
# densities.py / priors.py
# --------------------------------------------------------------------------
def gaussian(x, mu, var):
    """densities.py:25-27  (x, mu, var) with var the VARIANCE."""
    return -0.5 * LOG2PI - 0.5 * torch.log(var) - 0.5 * torch.square(mu - x) / var


def student_t(x, mean, scale, deg_free):
    """densities.py:52-59."""
    deg_free = T(deg_free, x.dtype)
    const = (
        torch.lgamma((deg_free + 1.0) * 0.5)
        - torch.lgamma(deg_free * 0.5)
        - 0.5 * (torch.log(torch.square(scale)) + torch.log(deg_free) + math.log(math.pi))
    )
    return const - 0.5 * (deg_free + 1.0) * torch.log(1.0 + (1.0 / deg_free) * torch.square((x - mean) / scale))


def multivariate_normal(x, mu, L):
    """densities.py:75-91 (columns independent; L = Cholesky of the covariance)."""
    d = x - mu
    d2 = d if d.dim() == 2 else d[:, None]
    alpha = torch.linalg.solve_triangular(L, d2, upper=False)
    num_col = 1.0 if x.dim() == 1 else float(x.shape[1])
    num_dims = float(x.shape[0])
    ret = -0.5 * num_dims * num_col * LOG2PI
    ret = ret - num_col * torch.sum(torch.log(torch.diagonal(L)))
    ret = ret - 0.5 * torch.sum(torch.square(alpha))
    return ret


def bimixture(fraction, logp0, logp1):
    """densities.py:94-103."""
    st = torch.stack([logp0 + torch.log(fraction), logp1 + torch.log(1.0 - fraction)], dim=-1)
    return log_sum_exp(st, axis=-1)


def prior_normal_logp(x):
    """priors.py:44-52."""
    return -0.5 * torch.sum(LOG2PI + torch.square(x))


# --------------------------------------------------------------------------
# variationals.py
# --------------------------------------------------------------------------
def sample_diag(q_mu, q_sqrt, u):
    """variationals.py:138-142  q_mu + exp(q_sqrt) * u  (q_sqrt stores log-std)."""
    return q_mu + torch.exp(q_sqrt) * u


def sample_fullrank(q_mu, q_sqrt, u):
    """variationals.py:144-146  q_mu + tril(q_sqrt) @ u  (batched over leading dims)."""
    sqrt = torch.tril(q_sqrt)
    return q_mu + torch.matmul(sqrt, u[..., None])[..., 0]


def vec_to_tri(v):
    """tf_wraps.py:50-71 (disabled native op) with the element order of transforms.py:225-244
    (LowerTriangular.forward: numpy tril_indices): [..., N(N+1)/2] -> lower-triangular [..., N, N]."""
    T_ = v.shape[-1]
    N = int((8 * T_ + 1) ** 0.5 / 2.0 - 0.5 + 1e-9)
    i, j = np.tril_indices(N)
    out = torch.zeros(v.shape[:-1] + (N, N), dtype=v.dtype)
    out[..., torch.as_tensor(i), torch.as_tensor(j)] = v
    return out


def tri_to_vec(t):
    """transforms.py:246-256 (LowerTriangular.backward order)."""
    i, j = np.tril_indices(t.shape[-1])
    return t[..., torch.as_tensor(i), torch.as_tensor(j)]


def logdet(q_sqrt, q_shape):
    """variationals.py:178-186."""
    if q_shape == "diagonal":
        return 2.0 * q_sqrt
    return torch.log(torch.square(torch.diagonal(q_sqrt, dim1=-2, dim2=-1)))


def kl_normal(q_sqrt, u, x, q_shape):
    """variationals.py:225-230  Normal._KL = -0.5*sum(logdet + u^2 - x^2)."""
    return -0.5 * torch.sum(logdet(q_sqrt, q_shape) + torch.square(u) - torch.square(x))


def kl_generic(q_sqrt, u, x, q_shape, prior_logp=None, transform=None, log_jacobian=None):
    """variationals.py:198-209 generic Monte-Carlo KL."""
    kl = -0.5 * torch.sum(LOG2PI + logdet(q_sqrt, q_shape) + torch.square(u))
    if prior_logp is not None:
        tx = x if transform is None else transform(x)
        kl = kl - torch.sum(prior_logp(tx))
        if log_jacobian is not None:
            kl = kl - torch.sum(log_jacobian(x))
    return kl


def gaussian_kl_analytic(mu, L, q_shape):
    """Closed-form KL[N(mu, LL^T) || N(0, I)] summed over leading rows.

    This is the reference's TEST oracle (testing/test_variationals.py:326-347),
    not what its ELBO evaluates; kept to check the MC estimator's mean.
    """
    mu = np.asarray(mu, dtype=np.float64)
    L = np.asarray(L, dtype=np.float64)
    kl = 0.0
    for i in range(mu.shape[0]):
        n = mu.shape[1]
        if q_shape == "diagonal":
            ld = 2.0 * np.sum(L[i])
            tr = np.sum(np.exp(2.0 * L[i]))
        else:
            ld = np.sum(np.log(np.square(np.diagonal(L[i]))))
            tr = np.sum(np.square(np.tril(L[i])))
        kl += -ld - n + tr + mu[i] @ mu[i]
    return 0.5 * kl


def feed_split(x, sizes):
    """param.py:516-537 Parameterized.feed: slice the LAST axis into consecutive
    chunks, children in sorted-name order ('q_mu' before 'q_sqrt')."""
    out, beg = [], 0
    for s in sizes:
        out.append(x[..., beg : beg + s])
        beg += s
    return out


# --------------------------------------------------------------------------
# gp/kernels.py
# --------------------------------------------------------------------------
def square_dist(X, X2, lengthscales):
    """gp/kernels.py:54-84  |a|^2 + |b|^2 - 2 a b^T on X/ell (2-D or batched 3-D)."""
    Xeff = X / lengthscales
    Xs = torch.sum(torch.square(Xeff), -1)
    if X2 is None:
        return -2.0 * torch.matmul(Xeff, Xeff.transpose(-1, -2)) + Xs[..., :, None] + Xs[..., None, :]
    X2eff = X2 / lengthscales
    X2s = torch.sum(torch.square(X2eff), -1)
    return -2.0 * torch.matmul(Xeff, X2eff.transpose(-1, -2)) + Xs[..., :, None] + X2s[..., None, :]


def rbf_K(X, X2, lengthscales):
    """gp/kernels.py:110-111."""
    return torch.exp(-square_dist(X, X2, lengthscales) / 2.0)


def rbf_K_difference(X, X2, lengthscales):
    """NOT the reference's form: the same kernel from the scaled coordinate DIFFERENCE, sum_d ((x_d - x2_d) / ell_d)^2
    (what the HIP kernels evaluate).  Equal to rbf_K in exact arithmetic; in float32 the reference's
    |a|^2 + |b|^2 - 2 a.b loses ~|a|^2 * 2^-24 absolutely, which at inputs out to 256 lengthscales leaves Kmm + 1e-5 I
    indefinite.  Used only by the float32 evaluation of the oracle that measures what fp32 allows
    (tests/test_model_gpu.py::test_cfg2_full_size_properties_fp32)."""
    if X2 is None:
        X2 = X
    d = (X[..., :, None, :] - X2[..., None, :, :]) / lengthscales
    return torch.exp(-torch.sum(torch.square(d), -1) / 2.0)


def csym_rbf_K(X, X2, lengthscales):
    """gp/kernels.py:122-126."""
    if X2 is None:
        X2 = X
    return torch.exp(-square_dist(X, X2, lengthscales) / 2.0) + torch.exp(-square_dist(X, -X2, lengthscales) / 2.0)


def euclid_dist(X, X2, lengthscales):
    """gp/kernels.py:86-88  sqrt(r^2 + 1e-12)."""
    return torch.sqrt(square_dist(X, X2, lengthscales) + 1e-12)


def matern32_K(X, X2, lengthscales):
    """Matern-3/2 on euclid_dist.  NOT in the reference (it defines the distance, gp/kernels.py:86-88, but no
    Matern class): published formula (Rasmussen & Williams 2006, eq. 4.17), parity unpinned by the reference;
    tests/test_oracle.py checks it against scikit-learn's Matern(nu=1.5)."""
    a = math.sqrt(3.0) * euclid_dist(X, X2, lengthscales)
    return (1.0 + a) * torch.exp(-a)


def matern52_K(X, X2, lengthscales):
    """Matern-5/2 (see matern32_K; scikit-learn Matern(nu=2.5))."""
    r2 = square_dist(X, X2, lengthscales) + 1e-12
    a = math.sqrt(5.0) * torch.sqrt(r2)
    return (1.0 + a + (5.0 / 3.0) * r2) * torch.exp(-a)


def rbf_Kdiag(X):
    """gp/kernels.py:90-91."""
    return torch.ones(X.shape[:-1], dtype=X.dtype)


def csym_rbf_Kdiag(X, lengthscales):
    """gp/kernels.py:128-131."""
    Xs = torch.sum(torch.square(X / lengthscales), -1)
    return torch.ones_like(Xs) + torch.exp(-2.0 * Xs)


def kern_cholesky(X, lengthscales, jitter, K=rbf_K):
    """gp/kernels.py:93-101  chol(K(X) + jitter*I)."""
    n = X.shape[-2]
    return torch.linalg.cholesky(K(X, None, lengthscales) + torch.eye(n, dtype=X.dtype) * jitter)


# --------------------------------------------------------------------------
# gp/gp.py
# --------------------------------------------------------------------------
def gp_samples(x, u, lengthscales, jitter, K=rbf_K):
    """gp/gp.py:37-50  u @ chol(K(x))^T."""
    L = kern_cholesky(x, lengthscales, jitter, K)
    return torch.matmul(u, L.transpose(-1, -2))


def sparse_effective_LT(x, z, lengthscales, jitter, K=rbf_K):
    """gp/gp.py:146-174  Lm^{-1} K(z, x); batched branch uses the explicit inverse."""
    Lm = kern_cholesky(z, lengthscales, jitter, K)
    if x.dim() == 2:
        return torch.linalg.solve_triangular(Lm, K(z, x, lengthscales), upper=False)
    N = x.shape[0]
    Lminv = torch.linalg.solve_triangular(Lm, torch.eye(z.shape[0], dtype=x.dtype), upper=False)
    zt = z[None].expand(N, -1, -1)
    return torch.matmul(Lminv[None].expand(N, -1, -1), K(zt, x, lengthscales))


def sparse_additional_cov(x, LnT, lengthscales, q_shape, K=rbf_K, Kdiag=None):
    """gp/gp.py:177-192."""
    if q_shape == "diagonal":
        kd = rbf_Kdiag(x) if Kdiag is None else Kdiag(x)
        return kd - torch.sum(torch.square(LnT), -2)
    return K(x, None, lengthscales) - torch.matmul(LnT.transpose(-1, -2), LnT)


def sparse_samples(x, u, z, lengthscales, jitter, q_shape="diagonal", eps=None, K=rbf_K, Kdiag=None):
    """gp/gp.py:99-143 SparseGP.samples.  `eps` is the injected standard-normal
    draw: shape x.shape[:-1] for 'diagonal' (ONE vector shared by all rows of u,
    gp.py:131-132), [N,n] / [N,1,n] for 'fullrank'."""
    assert q_shape in ("diagonal", "neglected", "fullrank")
    LnT = sparse_effective_LT(x, z, lengthscales, jitter, K)
    if x.dim() == 2:
        samples = torch.matmul(u, LnT)
    else:
        samples = torch.matmul(u[:, None, :], LnT)[:, 0, :]
    if q_shape == "neglected":
        return samples
    if q_shape == "diagonal":
        diag_cov = sparse_additional_cov(x, LnT, lengthscales, "diagonal", K, Kdiag)
        return samples + torch.sqrt(torch.abs(diag_cov)) * eps
    n = x.shape[-2]
    cov = sparse_additional_cov(x, LnT, lengthscales, "fullrank", K) + torch.eye(n, dtype=x.dtype) * jitter
    chol = torch.linalg.cholesky(cov)
    if x.dim() == 2:
        return samples + torch.matmul(eps, chol.transpose(-1, -2))
    return samples + torch.matmul(eps, chol.transpose(-1, -2))[:, 0, :]


# --------------------------------------------------------------------------
# nn.py
# --------------------------------------------------------------------------
def matbias(x, w, b, clip_enabled=False):
    """nn.py:31-32  clip(x @ w + b)."""
    return clip(torch.matmul(x, w) + b, clip_enabled)


def neural_net(x, ws, bs, acts=None, clip_enabled=False):
    """nn.py:73-84: activation (default sigmoid) after every layer but the last."""
    y = x
    n = len(ws)
    for i in range(n - 1):
        act = torch.sigmoid if acts is None else acts[i]
        y = act(matbias(y, ws[i], bs[i], clip_enabled))
    return matbias(y, ws[-1], bs[-1], clip_enabled)


# --------------------------------------------------------------------------
# model.py: Indexer + TF-1 Adam
# --------------------------------------------------------------------------
class Indexer:
    """model.py:126-153: shuffle once, hold out floor(0.1*N), with-replacement draws."""

    def __init__(self, data_size, rng, test_frac=0.1):
        self.data_size = data_size
        self.test_size = int(np.floor(data_size * test_frac))
        self.train_size = data_size - self.test_size
        index = np.arange(data_size)
        rng.shuffle(index)
        self._train_index = index[: self.train_size]
        self._test_index = index[self.train_size :]
        self.rng = rng

    def train_index(self, n):
        return self._train_index[self.rng.randint(0, self.train_size, n)]

    def test_index(self, n):
        return self._test_index[self.rng.randint(0, self.test_size, n)]


class AdamTF:
    """tf.train.AdamOptimizer update rule (third party).

    t <- t+1; lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; theta -= lr_t*m/(sqrt(v)+eps).
    """

    def __init__(self, params, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.t = 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    @torch.no_grad()
    def step(self, grads):
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            p.sub_(lr_t * m / (torch.sqrt(v) + self.eps))


# --------------------------------------------------------------------------
# model compositions (notebooks/GaussianProcess.ipynb:109-159)
# --------------------------------------------------------------------------
def svgp_elbo(params, X, Y, N_total, u_noise, eps, jitter=1e-5, q_shape="diagonal", residual="diagonal", K=rbf_K):
    """Sparse variational GP regression ELBO at injected noise.

    params: dict of RAW (pre-transform) leaves
       z [M,d], ell_raw [dl], q_mu [P,M] (flattened size P*M in the reference),
       q_sqrt [P*M] (diag, log-std) or [P*M,P*M] (fullrank), k_var_raw [1], var_raw [1]
    X [n,d], Y [n,1]; u_noise [P*M]; eps [n].
    ELBO = (N/n) * sum gaussian(Y^T, f*sqrt(k_var), var) - KL   (user-side rescale).
    """
    z = params["z"]
    ell = log1pe_forward(params["ell_raw"])
    k_var = log1pe_forward(params["k_var_raw"])
    var = log1pe_forward(params["var_raw"])
    P, M = params["q_mu"].shape
    mu = params["q_mu"].reshape(-1)
    if q_shape == "diagonal":
        xs = sample_diag(mu, params["q_sqrt"].reshape(-1), u_noise)
        kl = kl_normal(params["q_sqrt"].reshape(-1), u_noise, xs, "diagonal")
    else:
        xs = sample_fullrank(mu, params["q_sqrt"], u_noise)
        kl = kl_normal(params["q_sqrt"], u_noise, xs, "fullrank")
    u = xs.reshape(P, M)
    f = sparse_samples(X, u, z, ell, jitter, residual, eps, K=K) * torch.sqrt(k_var)
    n = X.shape[0]
    ll = torch.sum(gaussian(Y.transpose(0, 1), f, var))
    return (N_total / n) * ll - kl


def amortised_elbo(params, Y, u_noise):
    """cfg-4 composition: NeuralNet encoder -> LOCAL Normal -> linear Gaussian decoder.

    params: enc_w0,enc_b0,enc_w1,enc_b1, dec_w0,dec_b0, var_raw.  The encoder
    output columns are [q_mu (L), q_sqrt (L, log-std)] (param.py:516-537 order).
    ELBO = sum gaussian(Y, dec(z), var) - KL(local).
    """
    h = neural_net(Y, [params["enc_w0"], params["enc_w1"]], [params["enc_b0"], params["enc_b1"]])
    L = h.shape[-1] // 2
    q_mu, q_sqrt = feed_split(h, [L, L])
    zs = sample_diag(q_mu, q_sqrt, u_noise)
    kl = kl_normal(q_sqrt, u_noise, zs, "diagonal")
    rec = neural_net(zs, [params["dec_w0"]], [params["dec_b0"]])
    var = log1pe_forward(params["var_raw"])
    ll = torch.sum(gaussian(Y, rec, var))
    return ll - kl


def grads_of(fn, params):
    """ELBO value and d ELBO / d leaf for every leaf in `params` (torch autograd
    standing in for TF autodiff, model.py:220)."""
    leaves = {k: v.clone().detach().requires_grad_(True) for k, v in params.items()}
    val = fn(leaves)
    gs = torch.autograd.grad(val, list(leaves.values()), allow_unused=True)
    return val.detach(), {k: (g if g is not None else torch.zeros_like(leaves[k])) for k, g in zip(leaves, gs)}


def finite_difference(fn, params, key, idx, h=1e-6):
    """Central difference of fn w.r.t. params[key].flat[idx] (cross-check of autograd)."""
    p = {k: v.clone() for k, v in params.items()}
    flat = p[key].reshape(-1)
    old = flat[idx].item()
    flat[idx] = old + h
    fp = fn(p).item()
    flat[idx] = old - h
    fm = fn(p).item()
    return (fp - fm) / (2 * h)


def expert_elbo(params, X, Y, N_total, noises, eps, jitter=1e-5):
    """Two sparse-GP experts + sparse-GP gate (notebooks/Expert_GPR.ipynb:139-147, sparse form).

    params: for g in (s, l, r): z_g [M,d], ell_raw_g [1], q_mu_g [M], q_sqrt_g [M]; k_var_raw, k_var_r_raw,
    var_raw.  noises: {g: u [M]}; eps: [n,3] residual noise columns (s, l, r)."""
    fs, kl = {}, 0.0
    for i, g in enumerate(("s", "l", "r")):
        mu, sq = params["q_mu_" + g], params["q_sqrt_" + g]
        xs = sample_diag(mu, sq, noises[g])
        kl = kl + kl_normal(sq, noises[g], xs, "diagonal")
        ell = log1pe_forward(params["ell_raw_" + g])
        fs[g] = sparse_samples(X, xs.reshape(1, -1), params["z_" + g], ell, jitter, "diagonal", eps[:, i])
    k_var = log1pe_forward(params["k_var_raw"])
    k_var_r = log1pe_forward(params["k_var_r_raw"])
    var = log1pe_forward(params["var_raw"])
    frac = torch.sigmoid(fs["r"] * torch.sqrt(k_var_r))
    f = (frac * fs["s"] + (1.0 - frac) * fs["l"]) * k_var
    n = X.shape[0]
    ll = torch.sum(gaussian(Y.transpose(0, 1), f, var))
    return (N_total / n) * ll - kl


def experts_elbo(params, X, Y, N_total, u_noise, eps, jitter=1e-5):
    """E experts + E softmax gates (cfg 5; builder-defined generalisation of the 2-expert sigmoid form).

    params: z [2E,M,d], ell_raw [2E,1], q_mu [2E*M], q_sqrt [2E*M], k_var_raw, k_var_r_raw, var_raw;
    u_noise [2E*M]; eps [n, 2E]."""
    E2, M = params["z"].shape[0], params["z"].shape[1]
    E = E2 // 2
    xs = sample_diag(params["q_mu"], params["q_sqrt"], u_noise)
    kl = kl_normal(params["q_sqrt"], u_noise, xs, "diagonal")
    us = xs.reshape(E2, 1, M)
    ell = log1pe_forward(params["ell_raw"])
    fs = [sparse_samples(X, us[e], params["z"][e], ell[e], jitter, "diagonal", eps[:, e])[0] for e in range(E2)]
    f_e = torch.stack(fs[:E])
    g_e = torch.stack(fs[E:]) * torch.sqrt(log1pe_forward(params["k_var_r_raw"]))
    w = torch.softmax(g_e, dim=0)
    f = torch.sum(w * f_e, 0, keepdim=True) * log1pe_forward(params["k_var_raw"])
    n = X.shape[0]
    ll = torch.sum(gaussian(Y.transpose(0, 1), f, log1pe_forward(params["var_raw"])))
    return (N_total / n) * ll - kl
