"""The build's own model counterparts of the BASELINE configurations: SVGP (cfg 1/2/3),
Amortised (cfg 4), ExpertsGPR (cfg 5), plus the dense GPR and the two-expert form of the notebooks.
Used by bench.py, __graft_entry__.smoke() and the test suites.

Written the way a reference user would write them (SURVEY.md Appendix C;
reference notebooks/GaussianProcess.ipynb:109-159), with `tf = hb.tf`.
"""
import numpy as np

import henbun_amd as hb  # noqa: E402  (the package is fully imported before this module is)

tf = hb.tf


class SVGP(hb.model.Model):
    """Sparse variational GP regression: cfg 1/2 (q_shape='diagonal') and cfg 3 ('fullrank')."""

    def setUp(self, X, Y, Z, q_shape="diagonal", residual="diagonal", eps=None):
        self.N = X.shape[0]
        self.X = hb.param.MinibatchData(X)
        self.Y = hb.param.MinibatchData(Y)
        self.gp = hb.gp.SparseGP(kern=hb.gp.kernels.UnitRBF(np.ones(1)), z=Z)
        self.u = hb.variationals.Normal(shape=[1, Z.shape[0]], q_shape=q_shape)
        self.k_var = hb.param.Variable([1], transform=hb.transforms.positive)
        self.var = hb.param.Variable([1], transform=hb.transforms.positive)
        self.residual = residual
        self.eps = None if eps is None else hb.param.MinibatchData(eps)  # injected residual noise (parity runs)

    @hb.model.AutoOptimize()
    def ELBO(self):
        f = self.gp.samples(self.X, self.u, q_shape=self.residual, eps=self.eps) * tf.sqrt(self.k_var)
        n = tf.shape(self.X)[0]
        ll = tf.reduce_sum(hb.densities.gaussian(tf.transpose(self.Y), f, self.var))
        return (self.N / n) * ll - self.KL()


class Amortised(hb.model.Model):
    """cfg 4: NeuralNet encoder -> LOCAL Normal -> linear Gaussian decoder."""

    def setUp(self, Y, L=16, H=256, stddev=None):
        Din = Y.shape[1]
        self.Y = hb.param.MinibatchData(Y)
        self.z = hb.variationals.Normal([L], collections=hb.param.graph_key.LOCAL)
        self.enc = hb.nn.NeuralNet([Din, H, 2 * L], stddev=stddev or 1.0 / np.sqrt(Din))
        self.dec = hb.nn.NeuralNet([L, Din], stddev=stddev or 1.0 / np.sqrt(L))
        self.var = hb.param.Variable([1], transform=hb.transforms.positive)

    @hb.model.AutoOptimize()
    def ELBO(self):
        self.z = self.enc(self.Y)
        ll = tf.reduce_sum(hb.densities.gaussian(self.Y, self.dec(self.z), self.var))
        return ll - self.KL()


class DenseGPR(hb.model.Model):
    """Dense variational GP regression, the form of notebooks/GaussianProcess.ipynb:109-148."""

    def setUp(self, X, Y):
        self.X = hb.param.Data(X)
        self.Y = hb.param.Data(Y)
        self.kern = hb.gp.kernels.UnitRBF(np.ones(1))
        self.k_var = hb.param.Variable([1], transform=hb.transforms.positive)
        self.var = hb.param.Variable([1], transform=hb.transforms.positive)
        self.q = hb.variationals.Normal(shape=[X.shape[0], 1], q_shape="fullrank")

    @hb.model.AutoOptimize()
    def ELBO(self):
        f = tf.matmul(self.kern.Cholesky(self.X), self.q) * tf.sqrt(self.k_var)
        return tf.reduce_sum(hb.densities.gaussian(self.Y, f, self.var)) - self.KL()


def svgp_data(N, M, seed=0, domain=None, dtype=np.float64):
    """Synthetic regression set of the BASELINE configs: X ~ U(0, domain),
    Y = sin X + 0.3 eps, Z = linspace(0, domain, M) (spacing 0.5 lengthscales)."""
    rng = np.random.RandomState(seed)
    domain = 0.5 * M if domain is None else domain
    X = rng.uniform(0, domain, (N, 1))
    Y = np.sin(X) + 0.3 * rng.randn(N, 1)
    Z = np.linspace(0, domain, M)[:, None]
    return X.astype(dtype), Y.astype(dtype), Z.astype(dtype)


class ExpertGPR(hb.model.Model):
    """Mixture of two sparse-GP experts with a sparse-GP gate: the sparse form of
    notebooks/Expert_GPR.ipynb:101-160 (three independent GPs with their own kernels;
    f = (sigmoid(f_r) f_s + (1 - sigmoid(f_r)) f_l) * k_var  -- times k_var, not its root,
    as the notebook writes it)."""

    def setUp(self, X, Y, Z, ells=(0.3, 2.0, 1.0), eps=None):
        self.N = X.shape[0]
        self.X = hb.param.MinibatchData(X)
        self.Y = hb.param.MinibatchData(Y)
        self.gp_s = hb.gp.SparseGP(kern=hb.gp.kernels.UnitRBF(np.ones(1) * ells[0]), z=Z)
        self.gp_l = hb.gp.SparseGP(kern=hb.gp.kernels.UnitRBF(np.ones(1) * ells[1]), z=Z)
        self.gp_r = hb.gp.SparseGP(kern=hb.gp.kernels.UnitRBF(np.ones(1) * ells[2]), z=Z)
        M = Z.shape[0]
        self.u_s = hb.variationals.Normal(shape=[1, M])
        self.u_l = hb.variationals.Normal(shape=[1, M])
        self.u_r = hb.variationals.Normal(shape=[1, M])
        self.k_var = hb.param.Variable([1], transform=hb.transforms.positive)
        self.k_var_r = hb.param.Variable([1], transform=hb.transforms.positive)
        self.var = hb.param.Variable([1], transform=hb.transforms.positive)
        # injected residual noise, one column per GP (parity runs)
        self.eps = None if eps is None else hb.param.MinibatchData(eps)

    @hb.model.AutoOptimize()
    def ELBO(self):
        e = self.eps
        es, el, er = (None, None, None) if e is None else (e[:, 0], e[:, 1], e[:, 2])
        f_s = self.gp_s.samples(self.X, self.u_s, eps=es)
        f_l = self.gp_l.samples(self.X, self.u_l, eps=el)
        f_r = self.gp_r.samples(self.X, self.u_r, eps=er) * tf.sqrt(self.k_var_r)
        frac = tf.sigmoid(f_r)
        f = (frac * f_s + (1.0 - frac) * f_l) * self.k_var
        n = tf.shape(self.X)[0]
        ll = tf.reduce_sum(hb.densities.gaussian(tf.transpose(self.Y), f, self.var))
        return (self.N / n) * ll - self.KL()


class ExpertsGPR(hb.model.Model):
    """cfg 5: E sparse-GP experts with E sparse-GP gates (softmax gating; E = 2 with r = g_2 - g_1
    reduces to the sigmoid form of notebooks/Expert_GPR.ipynb:139-147).  The 2E independent GPs
    -- own inducing points, own lengthscale, own q(u) -- are ONE batched SparseGP, so their Gram
    matrices, Cholesky factors and M^2 n contractions run as single expert-batched launches."""

    def setUp(self, X, Y, Z, ells, eps=None):
        E2 = len(ells)
        self.E = E2 // 2
        self.N = X.shape[0]
        self.X = hb.param.MinibatchData(X)
        self.Y = hb.param.MinibatchData(Y)
        M = Z.shape[0]
        z = np.broadcast_to(Z, (E2,) + Z.shape).copy()
        self.gp = hb.gp.SparseGP(kern=hb.gp.kernels.UnitRBF(np.asarray(ells, dtype=np.float64).reshape(E2, 1)), z=z)
        self.u = hb.variationals.Normal(shape=[E2, 1, M])
        self.k_var = hb.param.Variable([1], transform=hb.transforms.positive)
        self.k_var_r = hb.param.Variable([1], transform=hb.transforms.positive)
        self.var = hb.param.Variable([1], transform=hb.transforms.positive)
        self.eps = None if eps is None else hb.param.MinibatchData(eps)  # [N, 2E] injected residual noise

    @hb.model.AutoOptimize()
    def ELBO(self):
        E = self.E
        eps = None if self.eps is None else tf.transpose(self.eps)        # [2E, n]
        f_all = self.gp.samples(self.X, self.u, eps=eps)                   # [2E, 1, n]
        f_e = f_all[:E, 0, :]                                              # [E, n]
        g_e = f_all[E:, 0, :] * tf.sqrt(self.k_var_r)
        g_max = tf.reduce_max(g_e, 0, keep_dims=True)
        w = tf.exp(g_e - g_max)
        w = w / tf.reduce_sum(w, 0, keep_dims=True)
        f = tf.reduce_sum(w * f_e, 0, keep_dims=True) * self.k_var         # [1, n]
        n = tf.shape(self.X)[0]
        ll = tf.reduce_sum(hb.densities.gaussian(tf.transpose(self.Y), f, self.var))
        return (self.N / n) * ll - self.KL()
