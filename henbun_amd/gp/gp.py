"""GP / SparseGP posterior draws (reference Henbun/gp/gp.py:9-192).

`SparseGP.samples` with a 2-D x, the UnitRBF kernel and q_shape in
{'diagonal','neglected'} lowers to the fused HIP path (hb_sgp_fwd/bwd: the RBF
cross-covariance block is never materialised and the M^2 n contraction runs on
MFMA); every other case is composed from the generic graph ops the way the
reference composes TensorFlow ops.
"""
from __future__ import annotations

import numpy as np

from .. import graph as G
from .._settings import settings
from ..param import Parameterized, Variable, graph_key
from .kernels import UnitRBF


class GP(Parameterized):
    """Dense GP: samples = u @ chol(K(x))^T (reference gp/gp.py:9-50)."""

    def __init__(self, kern):
        Parameterized.__init__(self)
        self.kern = kern

    def _kern(self):
        return object.__getattribute__(self, "kern")

    def samples(self, x, u):
        L = self._kern().Cholesky(x)
        return G.matmul(u, L, transpose_b=True)


class SparseGP(GP):
    def __init__(self, kern, z, collections=[graph_key.VARIABLES]):
        GP.__init__(self, kern)
        z = np.asarray(z)
        self.z = Variable(shape=z.shape, collections=collections)
        self.z = z  # deferred assignment of the initial inducing locations
        self.m = z.shape[-2]  # z is [m,d], or [E,m,d] for E independent GPs evaluated as one batch

    def _z(self):
        return object.__getattribute__(self, "z").tensor()

    def samples(self, x, u, q_shape="diagonal", eps=None):
        """reference gp/gp.py:99-143.  `eps` optionally injects the standard-normal
        draw of the residual term (shape x.shape[:-1] for 'diagonal')."""
        assert q_shape in ["diagonal", "neglected", "fullrank"]
        x, u = G.as_tensor(x), G.as_tensor(u)
        kern = self._kern()
        z = self._z()
        if len(x.shape) == 2 and isinstance(kern, UnitRBF) and q_shape in ("diagonal", "neglected"):
            Lm = kern.Cholesky(z)
            f, _, _, _ = G.sgp_samples(x, z, kern._ell(), Lm, u, mode=q_shape, eps=eps)
            return f
        # generic composition
        jitter = settings.numerics.jitter_level
        LnT = self._effective_LT(x)
        if len(x.shape) == 2:
            samples = G.matmul(u, LnT)
        else:
            samples = G.squeeze(G.matmul(G.expand_dims(u, 1), LnT), [1])
        if q_shape == "neglected":
            return samples
        if q_shape == "diagonal":
            diag_cov = self._additional_cov(x, LnT, "diagonal")
            noise = G.random_normal(x.shape[:-1]) if eps is None else G.as_tensor(eps)
            return G.add(samples, G.mul(G.unary("SQRT", G.unary("ABS", diag_cov)), noise))
        n = x.shape[-2]
        N = u.shape[0]
        chol = G.cholesky(G.add_eye(self._additional_cov(x, LnT, "fullrank"), jitter))
        if len(x.shape) == 2:
            noise = G.random_normal([N, n]) if eps is None else G.as_tensor(eps)
            return G.add(samples, G.matmul(noise, chol, transpose_b=True))
        noise = G.random_normal([N, 1, n]) if eps is None else G.as_tensor(eps)
        return G.add(samples, G.squeeze(G.matmul(noise, chol, transpose_b=True), [1]))

    def _effective_LT(self, x):
        """Lm^{-1} K(z, x) (reference gp/gp.py:146-174)."""
        x = G.as_tensor(x)
        kern = self._kern()
        z = self._z()
        Lm = kern.Cholesky(z)
        if len(x.shape) == 2:
            return G.triangular_solve(Lm, kern.K(z, x))
        if len(x.shape) == 3:
            # batched branch: explicit inverse, broadcast over the batch (no tiling needed)
            return G.matmul(G.trinv(Lm), kern.K(z, x))
        raise ValueError("shape is not specified for tensor x")

    def _additional_cov(self, x, LnT, q_shape):
        """K(x,x) - K(x,z) Kmm^-1 K(z,x) (reference gp/gp.py:177-192)."""
        kern = self._kern()
        if q_shape == "diagonal":
            return G.sub(kern.Kdiag(x), G.reduce_sum(G.square(LnT), -2))
        return G.sub(kern.K(x), G.matmul(LnT, LnT, transpose_a=True))
