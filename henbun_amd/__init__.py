"""henbun_amd: MI355X-native stochastic variational inference engine.

Keeps the authoring surface of fujii-team/Henbun (Parameterized / Variable /
Variational / gp / nn / densities / settings; reference Henbun/__init__.py:1-8)
and replaces TensorFlow with hand-written HIP kernels behind a C ABI
(include/henbun_hip.h, loaded by henbun_amd/_lib.py).  `hb.tf` offers the
handful of TensorFlow names that reference-style objective methods use.
"""
from . import densities, gp, graph, model, nn, param, priors, tf, tf_wraps, transforms, variationals
from ._settings import settings
from .graph import CholeskyError

__version__ = "0.1"
