"""Parameter tree: Parentable / Variable / Parameterized / ParamList / Data /
MinibatchData and the `tf_mode` context.

API surface of reference Henbun/param.py:29-739, re-implemented without
TensorFlow: a global `Variable` is a slice of the model's flat device parameter
buffer (raw / free-space values, so one fused Adam kernel updates everything);
in `tf_mode` every child reads as a `henbun_amd.graph.Tensor`; assigning a tensor
to a LOCAL child in `tf_mode` feeds it (column slicing in sorted-name order).
"""
from __future__ import annotations

from contextlib import contextmanager
from functools import reduce

import numpy as np

from . import graph as G
from . import transforms
from ._settings import settings


class _GraphKey:
    """Collection flags (reference param.py:29-47)."""

    VARIABLES = "variables"  # tf.GraphKeys.GLOBAL_VARIABLES has this value too
    LOCAL = "LOCAL"
    DATA = "DATA"

    @property
    def not_parameters(self):
        return [self.LOCAL, self.DATA]


graph_key = _GraphKey()


def _in_collection(collection, collections):
    """`collection in collections` as the reference evaluates it (list membership,
    or substring test when `collections` is the string 'LOCAL'/'DATA')."""
    if collection is None:
        return True
    return collection in collections


def truncated_normal(shape, mean=0.0, stddev=1.0):
    """tf.truncated_normal: redraw until within two standard deviations
    (reference param.py:206-208; bounds pinned by testing/test_param.py:286-296)."""
    shape = tuple(int(s) for s in shape)
    out = np.random.randn(*shape) if shape else np.array(np.random.randn())
    out = np.asarray(out, dtype=np.float64)
    bad = np.abs(out) > 2.0
    while np.any(bad):
        out[bad] = np.random.randn(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return mean + stddev * out


class Parentable:
    """Tree node that knows its parent and derives its name from the parent's
    attribute dict (reference param.py:49-95)."""

    def __init__(self):
        self._parent = None

    @property
    def highest_parent(self):
        return self if self._parent is None else self._parent.highest_parent

    @property
    def name(self):
        if self._parent is None:
            return "unnamed"
        if isinstance(self._parent, ParamList):
            return "item%i" % self._parent._list.index(self)
        matches = [k for k, v in self._parent.__dict__.items() if v is self]
        if len(matches) == 0:
            raise ValueError("mis-specified parent: the parent holds no reference to this object")
        if len(matches) > 1:
            raise ValueError("this object is referenced twice by its parent")
        return matches[0]

    @property
    def long_name(self):
        if self._parent is None:
            return self.name
        return self._parent.long_name + "." + self.name


class Variable(Parentable):
    """A parameter (global), an encoder-fed LOCAL quantity, or DATA, by `collections`
    (reference param.py:97-304)."""

    def __init__(self, shape, n_layers=[], n_batch=None, mean=0.0, stddev=1.0,
                 transform=transforms.Identity(), collections=[graph_key.VARIABLES]):
        Parentable.__init__(self)
        if isinstance(shape, (int, np.integer)):
            shape = [shape]
        self.transform = transform
        self.collections = collections
        self.n_batch = n_batch
        self.shape = [int(s) for s in shape]
        self.n_layers = [int(s) for s in n_layers]
        self._assigned = True
        self._tensor = None      # fed tensor (LOCAL) / placeholder leaf (DATA)
        self._leaf = None        # graph leaf of a global parameter (raw values)
        self._host_raw = None    # pending free-space value to upload at initialize()
        if not self.is_parameter:
            return
        self._full_shape = self.n_layers + ([int(n_batch)] if n_batch is not None else []) + self.shape
        self._host_raw = truncated_normal(self._full_shape, mean=mean, stddev=stddev)
        self._leaf = G.leaf("param", tuple(self._full_shape), var=self)

    # -- classification
    @property
    def is_parameter(self):
        return self.collections not in graph_key.not_parameters

    @property
    def is_local(self):
        return self.collections == graph_key.LOCAL

    # -- graph view
    def tensor(self):
        """In tf_mode this object reads as transform(raw) (reference param.py:211-218)."""
        if self.is_parameter:
            return self.transform.tf_forward(self._leaf)
        if self._tensor is None:
            return None
        return self.transform.tf_forward(self._tensor)

    def get_tf_variables(self, collection=None):
        if _in_collection(collection, self.collections):
            return [self._leaf if self.is_parameter else self._tensor]
        return []

    def get_variables(self, collection=None):
        return [self] if _in_collection(collection, self.collections) else []

    # -- values
    def assign(self, value):
        """Deferred assignment of a variable-space value (reference param.py:241-248)."""
        if self.is_parameter:
            value = np.asarray(value, dtype=np.float64)
            raw = np.asarray(self.transform.backward(value), dtype=np.float64)
            self._host_raw = np.broadcast_to(raw, self._full_shape).copy() if raw.shape != tuple(self._full_shape) \
                else raw.copy()
            self._assigned = True

    @property
    def initialize_ops(self):
        return [self] if (self.is_parameter and self._assigned) else []

    def finalize(self):
        self._assigned = False

    @property
    def value(self):
        """Current variable-space value (reference param.py:268-279)."""
        root = self.highest_parent
        assert hasattr(root, "_session"), "value needs the variable to be part of a Model"
        if not self.is_parameter:
            raise ValueError("only global parameters hold a value")
        return root._session.read_value(self)

    @property
    def feed_size(self):
        if self.is_local:
            return int(reduce(np.multiply, self.shape, 1))
        return 0

    def feed(self, x):
        """LOCAL: take x [*n_layers, N, feed_size] as this variable's raw value
        (reference param.py:291-304)."""
        if self.is_local:
            x = G.as_tensor(x)
            if self.n_batch is not None:
                assert x.shape[-2] == self.n_batch
            self._tensor = G.reshape(x, self.n_layers + [x.shape[-2]] + self.shape)

    def get_feed_dict(self, minibatch_index):
        if self.collections == graph_key.DATA:
            raise NotImplementedError
        return {}


def tri_pack(dense):
    """Lower triangle of [..., N, N] as [..., N(N+1)/2] in numpy tril_indices (row-major) order -- the layout of
    the reference's LowerTriangular transform (transforms.py:225-244)."""
    dense = np.asarray(dense)
    i, j = np.tril_indices(dense.shape[-1])
    return dense[..., i, j]


def tri_unpack(packed):
    packed = np.asarray(packed)
    T = packed.shape[-1]
    N = int((8 * T + 1) ** 0.5 / 2.0 - 0.5 + 1e-9)
    assert N * (N + 1) // 2 == T, "not a triangular number"
    out = np.zeros(packed.shape[:-1] + (N, N), dtype=packed.dtype)
    i, j = np.tril_indices(N)
    out[..., i, j] = packed
    return out


class TriPackedVariable(Variable):
    """A lower-triangular [size, size] matrix parameter stored as its packed lower triangle [size(size+1)/2]
    (the storage the reference sketched with its disabled vec_to_tri hook, tf_wraps.py:50-71).  Reads (`.value`)
    and deferred assignment speak dense [..., size, size] matrices -- entries above the diagonal do not exist:
    they read as zero and are dropped on assignment (in the dense form they are stored, masked at use and get a
    zero gradient, reference variationals.py:94-96,145).  The graph sees the packed vector (`dense_shape` tells
    consumers the matrix shape): parameter bytes, gradient bytes and the data-parallel all-reduce are halved."""

    def __init__(self, size, n_layers=[], mean=0.0, stddev=1.0, collections=[graph_key.VARIABLES]):
        size = int(size)
        self.dense_shape = [size, size]
        Variable.__init__(self, [size * (size + 1) // 2], n_layers=n_layers, mean=mean, stddev=stddev,
                          collections=collections)

    def assign(self, value):
        value = np.asarray(value, dtype=np.float64)
        if value.shape[-2:] == tuple(self.dense_shape):
            value = tri_pack(value)
        Variable.assign(self, value)

    @property
    def value(self):
        return tri_unpack(Variable.value.fget(self))


class Parameterized(Parentable):
    """Holds Variables / other Parameterized as attributes (reference param.py:306-603)."""

    def __init__(self):
        Parentable.__init__(self)
        self._tf_mode = False
        self.scoped_keys = []

    def __getattribute__(self, key):
        o = object.__getattribute__(self, key)
        try:
            if not object.__getattribute__(self, "_tf_mode"):
                return o
        except AttributeError:
            return o
        if key == "_parent":
            return o
        if isinstance(o, (Parameterized, Variable)) and hasattr(o, "tensor"):
            return o.tensor()
        return o

    def __setattr__(self, key, value):
        if key in self.__dict__.keys():
            p = object.__getattribute__(self, key)
            try:
                if object.__getattribute__(self, "_tf_mode"):
                    if isinstance(p, (Variable, Parameterized)):
                        p.feed(value)
                        return
            except AttributeError:
                pass
            if isinstance(p, Variable):
                if isinstance(value, (float, int)):
                    value = np.array([value], dtype=np.float64)
                if isinstance(value, np.ndarray):
                    p.assign(value)
                    return
            if isinstance(p, (Variable, Parameterized)) and isinstance(value, (Variable, Parameterized)):
                p._parent = None
        object.__setattr__(self, key, value)
        if isinstance(value, Parentable) and key != "_parent":
            value._parent = self
            root = self.highest_parent
            if hasattr(root, "_session") and root._session is not None:
                root._session.invalidate()

    @contextmanager
    def tf_mode(self):
        """Inside, child parameters read as graph tensors (reference param.py:419-453)."""
        self._begin_tf_mode()
        try:
            yield
        finally:
            self._end_tf_mode()

    def _begin_tf_mode(self):
        for child in self.sorted_variables:
            if isinstance(child, Parameterized):
                child._begin_tf_mode()
        self._tf_mode = True

    def _end_tf_mode(self):
        for child in self.sorted_variables:
            if isinstance(child, Parameterized):
                child._end_tf_mode()
        self._tf_mode = False

    @property
    def sorted_variables(self):
        """Children sorted by attribute name (reference param.py:455-465)."""
        d = object.__getattribute__(self, "__dict__")
        items = [(k, v) for k, v in d.items() if isinstance(v, (Variable, Parameterized)) and k != "_parent"]
        return [v for _, v in sorted(items, key=lambda kv: kv[0])]

    def get_tf_variables(self, collection=None):
        out = []
        for p in self.sorted_variables:
            out += p.get_tf_variables(collection)
        return out

    def get_variables(self, collection=None):
        out = []
        for p in self.sorted_variables:
            out += p.get_variables(collection)
        return out

    @property
    def initialize_ops(self):
        out = []
        for p in self.sorted_variables:
            out += p.initialize_ops
        return out

    def finalize(self):
        for p in self.sorted_variables:
            p.finalize()

    @property
    def feed_size(self):
        return int(np.sum([p.feed_size for p in self.get_variables(graph_key.LOCAL)], dtype=int))

    def feed(self, x):
        """Split x's last axis over the children in sorted-name order
        (reference param.py:516-537: 'q_mu' before 'q_sqrt')."""
        local = self.get_variables(graph_key.LOCAL)
        if len(local) == 0:
            return
        n_layers = local[0].n_layers
        for p in local:
            assert list(p.n_layers) == list(n_layers), \
                "n_layers of all LOCAL variables must agree to use feed(); feed them separately otherwise"
        x = G.as_tensor(x)
        begin = 0
        for p in self.sorted_variables:
            size = p.feed_size
            p.feed(x[..., begin:begin + size])
            begin += size

    def get_feed_dict(self, minibatch_index=None):
        fd = {}
        for p in self.sorted_variables:
            fd.update(p.get_feed_dict(minibatch_index))
        return fd

    def KL(self, collection=None):
        """Sum of the children's KL (reference param.py:549-560)."""
        kls = [p.KL(collection) for p in self.sorted_variables if hasattr(p, "KL")]
        kls = [k for k in kls if isinstance(k, G.Tensor)]
        if len(kls) == 0:
            return np.zeros([], dtype=np.float64)
        return reduce(G.add, kls)

    # -- checkpointing: {long_name: raw array} of global parameters (reference param.py:562-603)
    def _saved_variables(self):
        return {v.long_name: v for v in self.get_variables() if v.is_parameter}

    def save(self, save_path=None):
        """Write the sub-tree's parameters to `save_path` (.npz); returns the path."""
        root = self.highest_parent
        if save_path is None:
            save_path = self.name + ".ckpt"
        vd = self._saved_variables()
        if len(vd) == 0:
            raise ValueError("This class does not contain any global variables.")
        root.initialize()
        arrays = {k: root._session.read_raw(v) for k, v in vd.items()}
        path = save_path if save_path.endswith(".npz") else save_path + ".npz"
        np.savez(path, **arrays)
        return save_path

    def restore(self, save_path=None):
        root = self.highest_parent
        if save_path is None:
            save_path = self.name + ".ckpt"
        path = save_path if save_path.endswith(".npz") else save_path + ".npz"
        vd = self._saved_variables()
        with np.load(path) as f:
            missing = [k for k in vd if k not in f.files]
            if missing:
                raise KeyError("checkpoint %s lacks %s" % (path, missing))
            root._session.ensure_layout()
            for k, v in vd.items():
                root._session.write_raw(v, f[k])
        for v in self.get_variables():
            v.finalize()


class ParamList(Parameterized):
    """A list of parameters visible to the tree (reference param.py:605-674)."""

    def __init__(self, list_of_params=[]):
        Parameterized.__init__(self)
        list_of_params = list(list_of_params)
        for item in list_of_params:
            assert isinstance(item, (Variable, Parameterized))
            item._parent = self
        self._list = list_of_params

    @property
    def sorted_variables(self):
        return object.__getattribute__(self, "_list")

    def __getitem__(self, key):
        o = self.sorted_variables[key]
        if isinstance(o, Variable) and object.__getattribute__(self, "_tf_mode"):
            return o.tensor()
        return o

    def __len__(self):
        return len(self.sorted_variables)

    def append(self, item):
        assert isinstance(item, (Variable, Parameterized)), "this object is for containing parameters"
        item._parent = self
        self.sorted_variables.append(item)

    def __setitem__(self, key, value):
        p = self.sorted_variables[key]
        if isinstance(value, np.ndarray):
            p.assign(value)
        elif isinstance(value, (float, int)):
            p.assign(np.array([value], dtype=np.float64))
        else:
            raise TypeError


class Data(Variable):
    """Full data fed every run (reference param.py:676-714)."""

    def __init__(self, data):
        data = np.asarray(data)
        Variable.__init__(self, data.shape, n_layers=[], n_batch=None, collections=graph_key.DATA)
        self._check_dtype(data)
        self.data = data
        self._tensor = G.leaf("data", tuple(data.shape), var=self)

    @staticmethod
    def _check_dtype(array):
        if array.dtype in (np.float32, np.float64):
            return "float"
        if array.dtype in (np.int16, np.int32, np.int64):
            return "int"
        raise NotImplementedError("unknown dtype")

    def tensor(self):
        return self._tensor

    def get_feed_dict(self, minibatch_index=None):
        return {self._tensor: self.data}

    def assign(self, value):
        value = np.asarray(value)
        if not np.all(value.shape == self.data.shape):
            raise ValueError("The shape of data must be the same.")
        self.data = value
        root = self.highest_parent
        if hasattr(root, "_session") and root._session is not None:
            root._session.data_changed(self)

    @property
    def value(self):
        return self.data


class MinibatchData(Data):
    """Data whose first axis is the (minibatched) data axis (reference param.py:716-739).
    The whole array is kept resident on the device; each step gathers rows by
    index there (K0) instead of the reference's host fancy-index + feed."""

    def __init__(self, data):
        data = np.asarray(data)
        Variable.__init__(self, data.shape[1:], n_layers=[], n_batch=None, collections=graph_key.DATA)
        self._check_dtype(data)
        self.data = data
        self._tensor = None
        self._leaves = {}

    @property
    def data_size(self):
        return self.data.shape[0]

    def tensor(self):
        """A [n, *shape] leaf for the minibatch size of the trace in progress."""
        root = self.highest_parent
        n = root._session.trace_minibatch if hasattr(root, "_session") else None
        if n is None:
            n = self.data_size
        t = self._leaves.get(n)
        if t is None:
            t = G.leaf("minibatch", (int(n),) + tuple(self.shape), var=self)
            self._leaves[n] = t
        return t

    def get_feed_dict(self, minibatch_index):
        if minibatch_index is None:
            return {}
        return {self: self.data[minibatch_index]}
