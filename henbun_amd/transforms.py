"""Constraint transforms between the free (optimised) space and the variable
space.  Mirrors reference Henbun/transforms.py:27-180: numpy `forward` /
`backward`, graph `tf_forward` / `tf_log_jacobian`."""
from __future__ import annotations

import numpy as np

from . import graph as G


class Transform:
    def forward(self, x):
        raise NotImplementedError

    def backward(self, y):
        raise NotImplementedError

    def tf_forward(self, x):
        raise NotImplementedError

    def tf_log_jacobian(self, x):
        raise NotImplementedError

    def free_state_size(self, variable_shape):
        return int(np.prod(variable_shape))

    def __str__(self):
        raise NotImplementedError


class Identity(Transform):
    """reference transforms.py:73-87."""

    def forward(self, x):
        return x

    def backward(self, y):
        return y

    def tf_forward(self, x):
        return G.as_tensor(x)

    def tf_log_jacobian(self, x):
        return G.constant(np.zeros((1,)))

    def __str__(self):
        return "(none)"


class Exp(Transform):
    """y = exp(x) + lower (reference transforms.py:90-107)."""

    def __init__(self, lower=1e-6):
        self._lower = lower

    def forward(self, x):
        return np.exp(x) + self._lower

    def backward(self, y):
        return np.log(np.asarray(y, dtype=np.float64) - self._lower)

    def tf_forward(self, x):
        return G.affine(G.unary("EXP", x), 1.0, self._lower)

    def tf_log_jacobian(self, x):
        return G.reduce_sum(x)

    def __str__(self):
        return "+ve"


class Log1pe(Transform):
    """y = log(1 + exp(x)) + lower, i.e. softplus (reference transforms.py:110-143)."""

    def __init__(self, lower=1e-6):
        self._lower = lower

    def forward(self, x):
        x = np.asarray(x, dtype=np.float64)
        return np.logaddexp(0.0, x) + self._lower

    def backward(self, y):
        y = np.asarray(y, dtype=np.float64) - self._lower
        # log(exp(y) - 1), stable for large y
        return y + np.log(-np.expm1(-y))

    def tf_forward(self, x):
        return G.affine(G.unary("SOFTPLUS", x), 1.0, self._lower)

    def tf_log_jacobian(self, x):
        # -sum(log(1 + exp(-x))) = -sum(softplus(-x))
        return G.unary("NEG", G.reduce_sum(G.unary("SOFTPLUS", G.unary("NEG", x))))

    def __str__(self):
        return "+ve"


class Logistic(Transform):
    """y = a + (b - a) / (1 + exp(-x)) (reference transforms.py:146-180)."""

    def __init__(self, a=0.0, b=1.0):
        assert b > a
        self.a, self.b = float(a), float(b)

    def forward(self, x):
        return self.a + (self.b - self.a) / (1.0 + np.exp(-np.asarray(x, dtype=np.float64)))

    def backward(self, y):
        y = np.asarray(y, dtype=np.float64)
        return -np.log((self.b - self.a) / (y - self.a) - 1.0)

    def tf_forward(self, x):
        return G.affine(G.unary("SIGMOID", x), self.b - self.a, self.a)

    def tf_log_jacobian(self, x):
        # sum(x - 2 log(exp(x) + 1) + log(b - a))
        x = G.as_tensor(x)
        t = G.sub(x, G.affine(G.unary("SOFTPLUS", x), 2.0))
        return G.reduce_sum(G.affine(t, 1.0, float(np.log(self.b - self.a))))

    def __str__(self):
        return "[%s, %s]" % (self.a, self.b)


positive = Log1pe()
