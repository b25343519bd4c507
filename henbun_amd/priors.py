"""Priors (reference Henbun/priors.py:29-116)."""
from __future__ import annotations

import math

import numpy as np

from . import densities
from . import graph as G
from .param import Parameterized


class Prior(Parameterized):
    def logp(self, x):
        raise NotImplementedError

    def __str__(self):
        raise NotImplementedError


class Normal(Prior):
    """Zero-mean unit-variance Gaussian: -0.5*sum(log 2pi + x^2)  (reference priors.py:44-52)."""

    def logp(self, x):
        return G.affine(G.reduce_sum(G.affine(G.square(x), 1.0, math.log(2 * math.pi))), -0.5)

    def __str__(self):
        return "N(0,1)"


class Gaussian(Prior):
    def __init__(self, mu, var):
        Prior.__init__(self)
        self.mu = np.atleast_1d(np.array(mu, np.float64))
        self.var = np.atleast_1d(np.array(var, np.float64))

    def logp(self, x):
        return G.reduce_sum(densities.gaussian(x, self.mu, self.var))

    def __str__(self):
        return "N(%s,%s)" % (self.mu, self.var)


class LogNormal(Prior):
    def __init__(self, mu, var):
        Prior.__init__(self)
        self.mu = np.atleast_1d(np.array(mu, np.float64))
        self.var = np.atleast_1d(np.array(var, np.float64))

    def logp(self, x):
        return G.reduce_sum(densities.lognormal(x, self.mu, self.var))

    def __str__(self):
        return "logN(%s,%s)" % (self.mu, self.var)


class Gamma(Prior):
    def __init__(self, shape, scale):
        Prior.__init__(self)
        self.shape = np.atleast_1d(np.array(shape, np.float64))
        self.scale = np.atleast_1d(np.array(scale, np.float64))

    def logp(self, x):
        return G.reduce_sum(densities.gamma(self.shape, self.scale, x))

    def __str__(self):
        return "Ga(%s,%s)" % (self.shape, self.scale)


class Laplace(Prior):
    def __init__(self, mu, sigma):
        Prior.__init__(self)
        self.mu = np.atleast_1d(np.array(mu, np.float64))
        self.sigma = np.atleast_1d(np.array(sigma, np.float64))

    def logp(self, x):
        return G.reduce_sum(densities.laplace(self.mu, self.sigma, x))

    def __str__(self):
        return "Lap.(%s,%s)" % (self.mu, self.sigma)


class Uniform(Prior):
    def __init__(self, lower=0, upper=1):
        Prior.__init__(self)
        self.log_height = -math.log(upper - lower)
        self.lower, self.upper = lower, upper

    def logp(self, x):
        return G.constant(self.log_height * G.as_tensor(x).size)

    def __str__(self):
        return "U(%s,%s)" % (self.lower, self.upper)
