"""The three op shims the reference keeps in Henbun/tf_wraps.py:26-48 (`eye`,
`clip`, `log_sum_exp`), on the henbun_amd graph.  (The rest of that module's
role -- being the door to the numeric backend -- is henbun_amd/_lib.py.)"""
from __future__ import annotations

import numpy as np

from . import graph as G
from ._settings import settings


def eye(N):
    return G.constant(np.eye(int(N)))


def clip(tensor):
    """Config-gated clip_by_value, read at trace time (reference tf_wraps.py:33-39)."""
    if settings.numerics.clip_by_value:
        return G.unary("CLIP", tensor, (settings.numerics.clip_value_min, settings.numerics.clip_value_max))
    return G.as_tensor(tensor)


def log_sum_exp(tensor, axis=-1):
    """reference tf_wraps.py:42-48."""
    tensor = G.as_tensor(tensor)
    m = G.reduce_max(tensor, axis, keepdims=True)
    s = G.reduce_sum(G.unary("EXP", G.sub(tensor, m)), axis, keepdims=False)
    return G.add(G.squeeze(m, axis), G.unary("LOG", s))


def vec_to_tri(vectors):
    """[B, N(N+1)/2] -> [B, N, N] lower-triangular matrices, entries in numpy tril_indices order: the native op the
    reference declares and leaves disabled (tf_wraps.py:50-71; `tf.load_op_library('tfops/matpackops.so')`), here
    hb_vec_to_tri.  Gradient: tri_to_vec (tf_wraps.py:56-58)."""
    return G.vec_to_tri(vectors)


def tri_to_vec(matrices):
    """Inverse packing (hb_tri_to_vec)."""
    return G.tri_to_vec(matrices)
