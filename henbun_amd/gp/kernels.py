"""Unit-variance stationary kernels (reference Henbun/gp/kernels.py:28-131).

`K` lowers to the HIP Gram kernel (hb_gram_*), which forms the squared
distance directly as sum_d((x_d - x2_d)/ell_d)^2 -- the same value as the
reference's |a|^2+|b|^2-2ab^T on X/ell, better conditioned."""
from __future__ import annotations

import numpy as np

from .. import graph as G
from .. import transforms
from .._settings import settings
from ..param import Parameterized, Variable, graph_key
from ..variationals import Variational


class Kern(Parameterized):
    def __init__(self):
        Parameterized.__init__(self)
        self.scoped_keys.extend(["K", "Kdiag"])


class UnitStationary(Kern):
    kind = None

    def __init__(self, lengthscales=np.ones(1), n_batch=None, collections=[graph_key.VARIABLES]):
        Kern.__init__(self)
        if isinstance(lengthscales, np.ndarray):
            self.lengthscales = Variable(lengthscales.shape, transform=transforms.positive, collections=collections)
            self.lengthscales = lengthscales  # deferred assignment of the initial value
        elif isinstance(lengthscales, (Variable, Variational)):
            self.lengthscales = lengthscales
        else:
            raise TypeError
        self.scoped_keys.extend(["square_dist", "euclid_dist", "Cholesky"])

    def _ell(self):
        """Lengthscales as [dl], or [E, dl] when a 2-D array was given: E independent kernels
        evaluated as one batch (one per expert; builder extension, SURVEY.md cfg 5)."""
        ls = object.__getattribute__(self, "lengthscales")
        t = ls.tensor()
        if len(t.shape) == 2 and t.shape[0] > 1:
            return t
        return G.reshape(t, [-1])

    def square_dist(self, X, X2=None):
        """r^2 between X [n,d]/[N,n,d] and X2 (reference gp/kernels.py:54-84)."""
        X = G.as_tensor(X)
        X2 = X if X2 is None else G.as_tensor(X2)
        return G.gram(X, X2, self._ell(), "sqdist")

    def euclid_dist(self, X, X2=None):
        """sqrt(r^2 + 1e-12) (reference gp/kernels.py:86-88)."""
        return G.unary("SQRT", G.affine(self.square_dist(X, X2), 1.0, 1e-12))

    def Kdiag(self, X):
        X = G.as_tensor(X)
        return G.constant(np.ones(X.shape[:-1]))

    def K(self, X, X2=None):
        X = G.as_tensor(X)
        X2 = X if X2 is None else G.as_tensor(X2)
        return G.gram(X, X2, self._ell(), self.kind)

    def Cholesky(self, X):
        """chol(K(X) + jitter*I), jitter read at trace time (reference gp/kernels.py:93-101)."""
        return G.cholesky(G.add_eye(self.K(X), settings.numerics.jitter_level))


class UnitRBF(UnitStationary):
    """exp(-r^2/2) (reference gp/kernels.py:103-111)."""

    kind = "rbf"


class UnitCsymRBF(UnitStationary):
    """exp(-|x-x2|^2/2) + exp(-|x+x2|^2/2) (reference gp/kernels.py:113-131)."""

    kind = "csym_rbf"

    def Kdiag(self, X):
        X = G.as_tensor(X)
        ell = self._ell()
        Xs = G.reduce_sum(G.square(G.div(X, ell)), -1)
        return G.affine(G.unary("EXP", G.affine(Xs, -2.0)), 1.0, 1.0)


class UnitMatern32(UnitStationary):
    """Matern-3/2 on the reference's `euclid_dist` (gp/kernels.py:86-88):  (1 + sqrt(3) r) exp(-sqrt(3) r).
    The reference defines the distance but no Matern class: builder extension, parity pinned
    by the oracle restatement and scikit-learn's Matern only.  Composed from the pairwise-distance kernel
    (hb_gram 'sqdist') and fused elementwise ops; SparseGP uses it through the generic composition."""

    kind = "matern32"

    def K(self, X, X2=None):
        a = G.affine(self.euclid_dist(X, X2), float(np.sqrt(3.0)))
        return G.mul(G.affine(a, 1.0, 1.0), G.unary("EXP", G.unary("NEG", a)))


class UnitMatern52(UnitStationary):
    """Matern-5/2:  (1 + sqrt(5) r + 5/3 r^2) exp(-sqrt(5) r)  (see UnitMatern32)."""

    kind = "matern52"

    def K(self, X, X2=None):
        r2 = G.affine(self.square_dist(X, X2), 1.0, 1e-12)
        a = G.affine(G.unary("SQRT", r2), float(np.sqrt(5.0)))
        poly = G.add(G.affine(a, 1.0, 1.0), G.affine(r2, 5.0 / 3.0))
        return G.mul(poly, G.unary("EXP", G.unary("NEG", a)))
