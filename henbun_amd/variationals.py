"""Variational posteriors (reference Henbun/variationals.py:31-381).

`Variational` keeps `q_mu` and `q_sqrt` (diagonal: log-std vector; fullrank:
dense [size,size], lower triangle used) and reads, in tf_mode, as one
reparameterised sample.  Sampling and the Monte-Carlo KL of `Normal` lower to
the fused HIP kernels (hb_diag_sample_kl_* / hb_fullrank_sample_kl_*).
"""
from __future__ import annotations

import math
from functools import reduce

import numpy as np

from . import graph as G
from . import priors, transforms
from ._settings import settings
from .param import Parameterized, TriPackedVariable, Variable, graph_key
from .tf_wraps import clip


class Variational(Parameterized):
    def __init__(self, shape, n_layers=[], n_batch=None, q_shape="diagonal", mean=0.0, stddev=1.0,
                 prior=None, transform=transforms.Identity(), collections=[graph_key.VARIABLES], tri_pack=None):
        Parameterized.__init__(self)
        self._shape = [int(shape)] if isinstance(shape, (int, np.integer)) else [int(s) for s in shape]
        self.n_layers = [int(n_layers)] if isinstance(n_layers, (int, np.integer)) else [int(s) for s in n_layers]
        self.n_batch = n_batch
        self.size = int(reduce(np.multiply, self._shape, 1))
        self.collections = collections
        assert q_shape in ["diagonal", "fullrank"]
        self.q_shape = q_shape
        # initial values as reference variationals.py:84-96
        self.q_mu = Variable(self.size, n_layers=self.n_layers, n_batch=n_batch, mean=mean, stddev=0.1 * stddev,
                             collections=collections)
        if q_shape == "diagonal":
            self.q_sqrt = Variable(self.size, n_layers=self.n_layers, n_batch=n_batch, mean=math.log(stddev),
                                   stddev=0.1, collections=collections)
        elif self._use_tri_pack(tri_pack, collections, n_batch):
            # global full-rank q_sqrt kept as its packed lower triangle (tf_wraps.py:50-71)
            self.q_sqrt = TriPackedVariable(self.size, n_layers=self.n_layers, mean=stddev, stddev=0.1 * stddev,
                                            collections=collections)
        else:
            self.q_sqrt = Variable([self.size, self.size], n_layers=self.n_layers, n_batch=n_batch, mean=stddev,
                                   stddev=0.1 * stddev, collections=collections)
        self.transform = transform
        self.prior = prior
        self._draw = None       # (trace id, x, kl, u) of the sample drawn for the trace in progress
        self._injected_u = None

    # -- helpers
    @staticmethod
    def _use_tri_pack(tri_pack, collections, n_batch):
        if collections == graph_key.LOCAL or n_batch is not None:
            return False            # fed / batched q_sqrt stays dense: its columns come from an encoder output
        if tri_pack is None:
            tri_pack = bool(getattr(settings.numerics, "tri_pack", False))
        return bool(tri_pack)

    @property
    def packed(self):
        return isinstance(object.__getattribute__(self, "q_sqrt"), TriPackedVariable)

    def _dense_sqrt(self, sq):
        """q_sqrt as a dense [.., size, size] graph tensor (generic consumers: logdet, closed-form KL)."""
        return G.vec_to_tri(sq) if self.packed else sq

    @property
    def is_local(self):
        return self.collections == graph_key.LOCAL

    def _trace_id(self):
        root = self.highest_parent
        return getattr(getattr(root, "_session", None), "trace_id", 0)

    def inject_noise(self, u):
        """Use a fixed standard-normal draw `u` instead of the in-kernel RNG
        (numpy array / tensor of the sample's shape, or None to go back)."""
        self._injected_u = u
        self._draw = None

    def _raw_params(self):
        """(q_mu, q_sqrt) as graph tensors, regardless of tf_mode."""
        mu = object.__getattribute__(self, "q_mu")
        sq = object.__getattribute__(self, "q_sqrt")
        return mu.tensor(), sq.tensor()

    def _sample(self, u=None):
        """(x, kl_normal, u) for the current parameters (reference variationals.py:131-153)."""
        mu, sq = self._raw_params()
        if mu is None or sq is None:
            raise ValueError("local variable " + self.long_name + " is not fed.")
        stream = "local" if self.is_local else "global"
        if u is not None:
            u = G.reshape(G.as_tensor(u), mu.shape)
        if self.q_shape == "diagonal":
            if self.is_local and bool(getattr(settings.runtime, "fused_encoder", True)):
                # fed by a two-layer encoder on a data operand (the amortised model, SURVEY.md App. C cfg 4): encoder,
                # sample and KL become ONE op whose kernels never write the hidden layer (csrc/mlp.hip)
                pat = G.match_mlp2_encoder(mu, sq)
                if pat is not None:
                    y, w0, b0, w1, b1, act = pat
                    x, kl, uu, _ = G.mlp2_sample_kl(y, w0, b0, w1, b1, act, u=u, stream=stream)
                    return x, kl, uu
            return G.diag_sample_kl(mu, sq, u, stream=stream)
        return G.fullrank_sample_kl(mu, sq, u, stream=stream, packed=self.packed)

    def _current(self):
        tid = self._trace_id()
        if self._draw is None or self._draw[0] != tid:
            root = self.highest_parent
            probing = getattr(getattr(root, "_session", None), "probing", False)
            # the shape-validation trace of compile() uses a placeholder minibatch size: no injection there
            x, kl, u = self._sample(None if probing else self._injected_u)
            self._draw = (tid, x, kl, u)
        return self._draw

    @property
    def u(self):
        return self._current()[3]

    @property
    def _tensor(self):
        return self._current()[1]

    @property
    def transformed_tensor(self):
        return self.transform.tf_forward(self._current()[1])

    def tensor(self):
        """One sample, shaped n_layers + [N] + shape (reference variationals.py:112-119)."""
        t = self.transformed_tensor
        if not self.is_local and self.n_batch is None:
            return clip(G.reshape(t, self.n_layers + self._shape))
        return clip(G.reshape(t, self.n_layers + [-1] + self._shape))

    def feed(self, x):
        """LOCAL: split the encoder output over q_mu / q_sqrt and redraw
        (reference variationals.py:121-129)."""
        Parameterized.feed(self, x)
        if self.is_local:
            self._draw = None

    @property
    def logdet(self):
        """reference variationals.py:178-186."""
        mu, sq = self._raw_params()
        if self.q_shape == "diagonal":
            return G.affine(sq, 2.0)
        return G.unary("LOG", G.square(G.diag_part(self._dense_sqrt(sq))))

    def KL(self, collection=None):
        if collection is None or collection in self.collections:
            return self._KL()
        return np.zeros([], dtype=np.float64)

    def _KL(self):
        """Generic Monte-Carlo KL (reference variationals.py:198-209)."""
        _, x, _, u = self._current()
        kl = G.affine(G.reduce_sum(G.add(G.affine(self.logdet, 1.0, math.log(2.0 * math.pi)), G.square(u))), -0.5)
        if self.prior is not None:
            kl = G.sub(kl, G.reduce_sum(self.prior.logp(self.transform.tf_forward(x))))
            kl = G.sub(kl, G.reduce_sum(self.transform.tf_log_jacobian(x)))
        return kl


class Normal(Variational):
    """Standard-normal prior, identity transform (reference variationals.py:213-230)."""

    def __init__(self, shape, n_layers=[], n_batch=None, q_shape="diagonal", mean=0.0, stddev=1.0,
                 collections=[graph_key.VARIABLES], kl_form=None, tri_pack=None):
        Variational.__init__(self, shape, q_shape=q_shape, n_layers=n_layers, n_batch=n_batch, mean=mean,
                             stddev=stddev, prior=priors.Normal(), transform=transforms.Identity(),
                             collections=collections, tri_pack=tri_pack)
        if kl_form not in (None, "mc", "analytic"):
            raise ValueError("kl_form must be 'mc' or 'analytic'")
        self.kl_form = kl_form

    def _KL(self):
        """kl_form 'mc' (the reference's estimator, variationals.py:225-230): -0.5*sum(logdet + u^2 - x^2),
        produced by the sampler kernel itself.  kl_form 'analytic' (this build's extra mode; `kl_form=None` reads
        settings.numerics.kl_form): the closed form KL[N(mu, L L^T) || N(0, I)], see _KL_analytic."""
        form = self.kl_form or str(getattr(settings.numerics, "kl_form", "mc"))
        if form == "analytic":
            return _KL_analytic(self)
        return G.reshape(self._current()[2], [])


def _KL_analytic(v):
    """Closed-form KL[q || N(0, I)] of a diagonal / full-covariance Gaussian q = N(mu, L L^T),
        0.5 * sum( -logdet - 1 + trace + mu^2 ),
    logdet = 2 s, trace = exp(2 s) (diagonal: s is the log-std) or logdet = log S_kk^2, trace = sum tril(S)^2
    (full rank) -- the formula the reference uses as the expectation of its Monte-Carlo estimator
    (testing/test_variationals.py:326-347, compared there at rtol 0.1 over 100 draws).  It has no sampling
    noise, so its gradient w.r.t. (mu, s/S) is exact; the reparameterised sample itself is unchanged."""
    mu, sq = v._raw_params()
    half_mu2 = G.reduce_sum(G.square(mu))
    if v.q_shape == "diagonal":
        # sum(exp(2s) - 2s - 1)
        rest = G.reduce_sum(G.sub(G.unary("EXP", G.affine(sq, 2.0)), G.affine(sq, 2.0, 1.0)))
    else:
        d = G.diag_part(v._dense_sqrt(sq))
        rest = G.sub(G.reduce_sum(G.square(sq if v.packed else G.band_part(sq, -1, 0))),
                     G.reduce_sum(G.affine(G.unary("LOG", G.square(d)), 1.0, 1.0)))
    return G.reshape(G.affine(G.add(half_mu2, rest), 0.5), [])


class Gaussian(Normal):
    """Normal times a positive `scale` parameter (reference variationals.py:232-291)."""

    def __init__(self, shape, n_layers=[], n_batch=None, q_shape="diagonal", mean=0.0, stddev=1.0,
                 collections=[graph_key.VARIABLES], scale_shape=None, scale_n_layers=None):
        if np.abs(mean) < stddev:
            scale_mean, q_mean, q_std = stddev, mean / stddev, 1.0
        else:
            scale_mean, q_mean, q_std = np.abs(mean), 1.0, stddev / np.abs(mean)
        Variational.__init__(self, shape, q_shape=q_shape, n_layers=n_layers, n_batch=n_batch, mean=q_mean,
                             stddev=q_std, prior=priors.Normal(), transform=transforms.Identity(),
                             collections=collections)
        self.kl_form = None
        scale_shape = scale_shape or [1 for _ in self._shape]
        scale_layer = scale_n_layers or [1 for _ in self.n_layers]
        self.scale = Variable(scale_shape, n_layers=scale_layer, n_batch=n_batch, mean=scale_mean,
                              stddev=0.1 * scale_mean, transform=transforms.positive, collections=collections)

    def tensor(self):
        scale = object.__getattribute__(self, "scale").tensor()
        return G.mul(scale, Normal.tensor(self))


class OffsetGaussian(Gaussian):
    """Gaussian plus an offset parameter (reference variationals.py:293-314)."""

    def __init__(self, shape, n_layers=[], n_batch=None, q_shape="diagonal", mean=0.0, stddev=1.0,
                 collections=[graph_key.VARIABLES], scale_shape=None, scale_n_layers=None):
        Gaussian.__init__(self, shape=shape, n_layers=n_layers, n_batch=n_batch, q_shape=q_shape, mean=0.0,
                          stddev=stddev, collections=collections, scale_shape=scale_shape,
                          scale_n_layers=scale_n_layers)
        offset_shape = scale_shape or [1 for _ in self._shape]
        offset_layer = scale_n_layers or [1 for _ in self.n_layers]
        self.offset = Variable(offset_shape, n_layers=offset_layer, n_batch=n_batch, mean=mean,
                               stddev=0.1 * abs(mean), collections=collections)

    def tensor(self):
        offset = object.__getattribute__(self, "offset").tensor()
        return G.add(Gaussian.tensor(self), offset)
