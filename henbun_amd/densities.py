"""Elementwise log-densities (reference Henbun/densities.py:25-103), expressed
on the henbun_amd graph.  `gaussian` lowers to one fused HIP kernel (and one
fused 3-output gradient kernel)."""
from __future__ import annotations

import math

import numpy as np

from . import graph as G
from .tf_wraps import log_sum_exp


def _t(x):
    return G.as_tensor(x)


def gaussian(x, mu, var):
    """reference densities.py:25-27; argument order (x, mu, var), var = VARIANCE."""
    return G.gauss_logpdf(x, mu, var)


def lognormal(x, mu, var):
    lnx = G.unary("LOG", x)
    return G.sub(gaussian(lnx, mu, var), lnx)


def bernoulli(p, y):
    """log(p if y == 1 else 1 - p)  (reference densities.py:35-36)."""
    p, y = _t(p), _t(y)
    is1 = G.binary("EQ", y, G.constant(1.0))
    return G.unary("LOG", G.where(is1, p, G.affine(p, -1.0, 1.0)))


def poisson(lamb, y):
    lamb, y = _t(lamb), _t(y)
    return G.sub(G.sub(G.mul(y, G.unary("LOG", lamb)), lamb), G.unary("LGAMMA", G.affine(y, 1.0, 1.0)))


def exponential(lamb, y):
    lamb, y = _t(lamb), _t(y)
    return G.sub(G.unary("NEG", G.div(y, lamb)), G.unary("LOG", lamb))


def gamma(shape, scale, x):
    shape, scale, x = _t(shape), _t(scale), _t(x)
    return G.add(G.sub(G.sub(G.unary("NEG", G.mul(shape, G.unary("LOG", scale))), G.unary("LGAMMA", shape)),
                       G.div(x, scale)), G.mul(G.affine(shape, 1.0, -1.0), G.unary("LOG", x)))


def student_t(x, mean, scale, deg_free):
    """reference densities.py:52-59."""
    x, mean, scale, nu = _t(x), _t(mean), _t(scale), _t(deg_free)
    const = G.sub(G.sub(G.unary("LGAMMA", G.affine(nu, 0.5, 0.5)), G.unary("LGAMMA", G.affine(nu, 0.5))),
                  G.affine(G.add(G.add(G.unary("LOG", G.square(scale)), G.unary("LOG", nu)), math.log(math.pi)), 0.5))
    zz = G.square(G.div(G.sub(x, mean), scale))
    body = G.unary("LOG", G.affine(G.div(zz, nu), 1.0, 1.0))
    return G.sub(const, G.mul(G.affine(nu, 0.5, 0.5), body))


def beta(alpha, beta, y):
    alpha, beta, y = _t(alpha), _t(beta), _t(y)
    y = G.unary("CLIP", y, (1e-6, 1.0 - 1e-6))
    return G.add(G.add(G.mul(G.affine(alpha, 1.0, -1.0), G.unary("LOG", y)),
                       G.mul(G.affine(beta, 1.0, -1.0), G.unary("LOG", G.affine(y, -1.0, 1.0)))),
                 G.sub(G.sub(G.unary("LGAMMA", G.add(alpha, beta)), G.unary("LGAMMA", alpha)), G.unary("LGAMMA", beta)))


def laplace(mu, sigma, y):
    mu, sigma, y = _t(mu), _t(sigma), _t(y)
    return G.sub(G.unary("NEG", G.div(G.unary("ABS", G.sub(mu, y)), sigma)), G.unary("LOG", G.affine(sigma, 2.0)))


def multivariate_normal(x, mu, L):
    """reference densities.py:75-91: columns independent, L = chol(cov)."""
    x, mu, L = _t(x), _t(mu), _t(L)
    d = G.sub(x, mu)
    vec = len(d.shape) == 1
    d2 = G.expand_dims(d, 1) if vec else d
    alpha = G.triangular_solve(L, d2, lower=True)
    num_col = 1.0 if vec else float(x.shape[1])
    num_dims = float(x.shape[0])
    ret = -0.5 * num_dims * num_col * math.log(2 * math.pi)
    logdet = G.reduce_sum(G.unary("LOG", G.diag_part(L)))
    return G.affine(G.add(G.affine(logdet, num_col), G.affine(G.reduce_sum(G.square(alpha)), 0.5)), -1.0, ret)


def bimixture(fraction, logp0, logp1):
    """log(fraction*exp(logp0) + (1-fraction)*exp(logp1))  (reference densities.py:94-103)."""
    fraction = _t(fraction)
    a = G.add(_t(logp0), G.unary("LOG", fraction))
    b = G.add(_t(logp1), G.unary("LOG", G.affine(fraction, -1.0, 1.0)))
    shp = G.bshape(a.shape, b.shape)
    st = G.stack([G.broadcast_to(a, shp), G.broadcast_to(b, shp)], axis=-1)
    return log_sum_exp(st, axis=-1)
