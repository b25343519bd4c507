"""Priors on the hot path (reference Henbun/priors.py:29-52): the base class and the unit Normal that
`Variational.__init__` installs by default (variationals.py:66).  The reference's other prior classes
(priors.py:55-116) are out of scope and are not provided."""
from __future__ import annotations

import math

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


def _out_of_scope(name, where):
    class _Missing(Prior):
        def __init__(self, *args, **kwargs):
            raise NotImplementedError(
                "henbun_amd.priors.%s: the reference's %s (Henbun/priors.py:%s) is outside the accelerated path "
                "(SURVEY.md section 2, INTEGRATION.md 'Not provided'); write the term with hb.densities.* inside the "
                "objective, or use priors.Normal with a transform" % (name, name, where))

    _Missing.__name__ = name
    return _Missing


# reference prior classes that are NOT provided: constructing one says so instead of an AttributeError
Gaussian = _out_of_scope("Gaussian", "55-65")
LogNormal = _out_of_scope("LogNormal", "68-78")
Gamma = _out_of_scope("Gamma", "81-91")
Laplace = _out_of_scope("Laplace", "94-104")
Uniform = _out_of_scope("Uniform", "107-116")
