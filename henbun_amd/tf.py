"""A TensorFlow-1-flavoured op namespace over the henbun_amd graph.

Reference user code calls raw `tf.*` inside objective methods (e.g.
notebooks/GaussianProcess.ipynb:135-148: tf.sqrt, tf.reduce_sum, tf.transpose;
Expert_GPR.ipynb:139-147: tf.sigmoid).  `import henbun_amd as hb; tf = hb.tf`
keeps such methods working: the names below build graph nodes that lower to HIP
kernels.
"""
from __future__ import annotations

import numpy as np

from . import graph as G
from .graph import (Tensor, concat, expand_dims, matmul, reduce_max, reduce_mean, reduce_sum, reshape, squeeze,
                    stack, stop_gradient, tile, transpose, where)
from . import model as _model

float32, float64 = np.float32, np.float64


def _u(name):
    def f(x, name=None):
        return G.unary(_NAMES[f], x)
    return f


def convert_to_tensor(value, dtype=None, name=None):
    return G.as_tensor(value)


constant = convert_to_tensor


def cast(x, dtype=None, name=None):
    return G.as_tensor(x)


def identity(x, name=None):
    return G.as_tensor(x)


def shape(x):
    return list(G.as_tensor(x).shape)


def negative(x, name=None):
    return G.unary("NEG", x)


def exp(x, name=None):
    return G.unary("EXP", x)


def log(x, name=None):
    return G.unary("LOG", x)


def sqrt(x, name=None):
    return G.unary("SQRT", x)


def square(x, name=None):
    return G.unary("SQUARE", x)


def abs(x, name=None):
    return G.unary("ABS", x)


def sign(x, name=None):
    return G.unary("SIGN", x)


def sigmoid(x, name=None):
    return G.unary("SIGMOID", x)


def tanh(x, name=None):
    return G.unary("TANH", x)


def lgamma(x, name=None):
    return G.unary("LGAMMA", x)


def log1p(x, name=None):
    return G.unary("LOG1P", x)


def reciprocal(x, name=None):
    return G.unary("RECIP", x)


def rsqrt(x, name=None):
    return G.unary("RSQRT", x)


def pow(x, y, name=None):
    return G.as_tensor(x) ** y


def add(x, y, name=None):
    return G.add(x, y)


def subtract(x, y, name=None):
    return G.sub(x, y)


def multiply(x, y, name=None):
    return G.mul(x, y)


def divide(x, y, name=None):
    return G.div(x, y)


def maximum(x, y, name=None):
    return G.binary("MAX", x, y)


def minimum(x, y, name=None):
    return G.binary("MIN", x, y)


def clip_by_value(t, clip_value_min, clip_value_max, name=None):
    return G.unary("CLIP", t, (clip_value_min, clip_value_max))


def add_n(inputs, name=None):
    return G.add_n([G.as_tensor(t) for t in inputs])


def ones(shape, dtype=None):
    return G.constant(np.ones(shape))


def zeros(shape, dtype=None):
    return G.constant(np.zeros(shape))


def ones_like(x, dtype=None):
    return G.unary("AFFINE", x, (0.0, 1.0))


def zeros_like(x, dtype=None):
    return G.unary("AFFINE", x, (0.0, 0.0))


def eye(n, dtype=None):
    return G.constant(np.eye(int(n)))


def diag(x):
    x = G.as_tensor(x)
    n = x.shape[0]
    return G.mul(G.constant(np.eye(n)), G.reshape(x, [n, 1]))


def slice(x, begin, size, name=None):
    return G.slice_(x, begin, size)


def random_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None, name=None):
    return G.affine(G.random_normal(shape), stddev, mean)


def cholesky(x, name=None):
    return G.cholesky(x)


def matrix_triangular_solve(matrix, rhs, lower=True, adjoint=False, name=None):
    return G.triangular_solve(matrix, rhs, lower=lower, adjoint=adjoint)


def matrix_band_part(x, num_lower, num_upper, name=None):
    return G.band_part(x, num_lower, num_upper)


def matrix_diag_part(x, name=None):
    return G.diag_part(x)


diag_part = matrix_diag_part


def matrix_transpose(x, name=None):
    return G.matrix_transpose(x)


def gradients(ys, xs):
    xs = list(xs) if isinstance(xs, (list, tuple)) else [xs]
    return G.gradients(G.reshape(G.as_tensor(ys), []), xs)


class nn:  # noqa: N801  (tf.nn.*)
    sigmoid = staticmethod(sigmoid)
    tanh = staticmethod(tanh)

    @staticmethod
    def relu(x, name=None):
        return G.unary("RELU", x)

    @staticmethod
    def softplus(x, name=None):
        return G.unary("SOFTPLUS", x)


class train:  # noqa: N801  (tf.train.*)
    AdamOptimizer = _model.AdamOptimizer
