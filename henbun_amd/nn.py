"""MatBias / NeuralNet (reference Henbun/nn.py:10-87).  A MatBias layer followed
by a known activation lowers to ONE fused HIP launch (MFMA GEMM + bias +
activation epilogue)."""
from __future__ import annotations

from . import graph as G
from ._settings import settings
from .param import Parameterized, Variable, graph_key
from .tf_wraps import clip


def sigmoid(x, name=None):
    return G.unary("SIGMOID", x)


def relu(x, name=None):
    return G.unary("RELU", x)


def tanh(x, name=None):
    return G.unary("TANH", x)


_FUSABLE = {sigmoid: "sigmoid", relu: "relu", tanh: "tanh"}


class MatBias(Parameterized):
    def __init__(self, nodes, n_layers=[], mean=0.0, stddev=1.0, variable=Variable,
                 collections=[graph_key.VARIABLES]):
        assert len(nodes) == 2
        Parameterized.__init__(self)
        self.w = variable(shape=[nodes[0], nodes[1]], n_layers=n_layers, mean=mean, stddev=stddev,
                          collections=collections)
        self.b = variable(shape=[1, nodes[1]], n_layers=n_layers, mean=mean, stddev=stddev, collections=collections)

    def _wb(self):
        return object.__getattribute__(self, "w").tensor(), object.__getattribute__(self, "b").tensor()

    def __call__(self, x, act=None):
        """clip(x @ w + b), optionally with a fused activation (reference nn.py:31-32)."""
        w, b = self._wb()
        fuse = _FUSABLE.get(act) if not settings.numerics.clip_by_value else None
        if fuse is not None:
            return G.matmul(x, w, bias=b, act=fuse)
        y = clip(G.matmul(x, w, bias=b))
        return y if act is None else act(y)


class NeuralNet(Parameterized):
    def __init__(self, nodes, n_layers=[], mean=0.0, stddev=1.0, variable_types=Variable, neuron_types=sigmoid,
                 collections=[graph_key.VARIABLES]):
        Parameterized.__init__(self)
        self.nodes = nodes
        if not isinstance(variable_types, list):
            variable_types = [variable_types for _ in range(len(nodes) - 1)]
        if not isinstance(neuron_types, list):
            self.neuron_types = [neuron_types for _ in range(len(nodes) - 2)]
        else:
            self.neuron_types = neuron_types
        self._matbias_list = []
        for i in range(len(nodes) - 1):
            mb = MatBias(nodes=[nodes[i], nodes[i + 1]], n_layers=n_layers, mean=mean, stddev=stddev,
                         variable=variable_types[i], collections=collections)
            self._matbias_list.append(mb)
            setattr(self, "matbias" + str(i), mb)

    def __call__(self, x):
        """Activation after every layer but the last (reference nn.py:73-84)."""
        y = x
        mbs = object.__getattribute__(self, "_matbias_list")
        for i in range(len(self.nodes) - 2):
            y = mbs[i](y, act=self.neuron_types[i])
        return mbs[-1](y)

    def __getitem__(self, i):
        return object.__getattribute__(self, "_matbias_list")[i]
