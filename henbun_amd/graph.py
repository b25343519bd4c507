"""Lazy tensor graph, graph-level reverse-mode autodiff and the HIP executor.

This module stands where the reference leans on TensorFlow's graph + autodiff
+ session (reference Henbun/model.py:206-230 `Optimizer.compile`:
`likelihood_method(model)` traced in tf_mode, `optimizer.minimize(-objective)`;
model.py:265-266 `session.run`).  User code written against `henbun_amd.tf`
builds `Tensor` nodes; `gradients()` extends the graph with the backward pass
(every VJP is expressed with the same primitive ops, fused HIP kernels included);
`Plan` lowers the graph to a fixed list of C-ABI kernel launches on
preallocated device buffers, which `Plan.capture()` turns into one hipGraph.

Nothing here computes on the host: evaluation happens only through
`henbun_amd.hip_ops` (ctypes -> libhenbun_hip.so).
"""
from __future__ import annotations

import itertools
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from ._lib import HipBackendError as _HipBackendError

_uid = itertools.count()

# ------------------------------------------------------------------------------
# graph objects
# ------------------------------------------------------------------------------


class Node:
    __slots__ = ("op", "inputs", "attrs", "outputs", "id")

    def __init__(self, op, inputs, attrs, out_shapes):
        self.op = op
        self.inputs = tuple(inputs)
        self.attrs = attrs
        self.id = next(_uid)
        self.outputs = [Tensor(self, i, tuple(int(d) for d in s)) for i, s in enumerate(out_shapes)]


class Tensor:
    """A symbolic value: output `index` of `node`, static `shape`."""

    __slots__ = ("node", "index", "shape")
    __array_priority__ = 1000  # numpy defers to our reflected operators

    def __init__(self, node, index, shape):
        self.node, self.index, self.shape = node, index, shape

    # -- tf.Tensor look-alikes
    def get_shape(self):
        return _Shape(self.shape)

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1

    def __repr__(self):
        return "<hb.Tensor %s#%d:%d shape=%s>" % (self.node.op, self.node.id, self.index, self.shape)

    def __hash__(self):
        return id(self)

    def __eq__(self, other):
        return self is other

    # -- operators
    def __add__(self, o):
        return add(self, o)

    def __radd__(self, o):
        return add(o, self)

    def __sub__(self, o):
        return sub(self, o)

    def __rsub__(self, o):
        return sub(o, self)

    def __mul__(self, o):
        return mul(self, o)

    def __rmul__(self, o):
        return mul(o, self)

    def __truediv__(self, o):
        return div(self, o)

    def __rtruediv__(self, o):
        return div(o, self)

    def __neg__(self):
        return unary("NEG", self)

    def __pow__(self, p):
        if isinstance(p, (int, float)):
            return square(self) if p == 2 else unary("POWC", self, (float(p),))
        return binary("POW", self, p)

    def __getitem__(self, key):
        return getitem(self, key)


class _Shape(tuple):
    """tuple with the tf.TensorShape bits user code touches (`.ndims`, `.as_list()`)."""

    @property
    def ndims(self):
        return len(self)

    def as_list(self):
        return list(self)


_intern: Dict[tuple, Node] = {}


def _freeze(v):
    if isinstance(v, dict):
        return tuple(sorted((k, _freeze(x)) for k, x in v.items()))
    if isinstance(v, (list, tuple)):
        return tuple(_freeze(x) for x in v)
    if isinstance(v, np.ndarray):
        return ("nd", v.shape, v.tobytes())
    return v


def make(op, inputs, attrs, out_shapes, unique=False) -> Node:
    """Create (or reuse: structural hash-consing) a node.  The reference builds
    e.g. kern.Cholesky(z) twice per sample (gp/gp.py:116,159); interning makes
    the duplicate free."""
    attrs = dict(attrs or {})
    if unique:
        return Node(op, inputs, attrs, out_shapes)
    key = (op, tuple(id(t) for t in inputs), _freeze(attrs))
    n = _intern.get(key)
    if n is None:
        n = Node(op, inputs, attrs, out_shapes)
        _intern[key] = n
    return n


def reset_interning():
    _intern.clear()


# ------------------------------------------------------------------------------
# op registry
# ------------------------------------------------------------------------------
class OpDef:
    def __init__(self, emit=None, vjp=None):
        self.emit = emit  # (plan, node) -> None (appends launch closures to plan.steps)
        self.vjp = vjp    # (node, out_grads[list[Tensor|None]]) -> list[Tensor|None] per input


OPS: Dict[str, OpDef] = {}


def defop(name, emit=None, vjp=None):
    OPS[name] = OpDef(emit, vjp)


# ------------------------------------------------------------------------------
# leaves
# ------------------------------------------------------------------------------
def leaf(kind, shape, **attrs) -> Tensor:
    """kind in {'param','data','minibatch','const','noise'}; always unique."""
    return Node("leaf:" + kind, (), attrs, [shape]).outputs[0]


def constant(value) -> Tensor:
    arr = np.asarray(value, dtype=np.float64)
    return make("leaf:const", (), {"value": arr}, [arr.shape]).outputs[0]


def as_tensor(x) -> Tensor:
    if isinstance(x, Tensor):
        return x
    if hasattr(x, "tensor") and callable(x.tensor):  # Variable / Variational used outside tf_mode
        return x.tensor()
    return constant(x)


def is_scalar_const(x):
    return isinstance(x, (int, float, np.floating, np.integer)) or (
        isinstance(x, np.ndarray) and x.ndim == 0
    )


# ------------------------------------------------------------------------------
# elementwise
# ------------------------------------------------------------------------------
def bshape(*shapes):
    return tuple(np.broadcast_shapes(*shapes))


def _cval(t):
    """the value of a single-element constant tensor, else None"""
    if isinstance(t, Tensor) and t.node.op == "leaf:const" and t.size == 1:
        return float(np.asarray(t.node.attrs["value"]).reshape(-1)[0])
    return None


def unary(opname, x, params=()) -> Tensor:
    x = as_tensor(x)
    c = _cval(x)
    if c is not None and opname in ("NEG", "AFFINE"):
        # single-element constants fold at trace time: the gradient seed and its sign flips never reach the device
        v = -c if opname == "NEG" else float(params[0]) * c + float(params[1])
        return constant(np.full(x.shape, v))
    return make("ew", (x,), {"f": opname, "p": tuple(float(p) for p in params)}, [x.shape]).outputs[0]


def binary(opname, a, b, params=()) -> Tensor:
    a, b = as_tensor(a), as_tensor(b)
    return make("ew", (a, b), {"f": opname, "p": tuple(float(p) for p in params)}, [bshape(a.shape, b.shape)]).outputs[0]


def affine(x, scale=1.0, shift=0.0) -> Tensor:
    if scale == 1.0 and shift == 0.0:
        return as_tensor(x)
    return unary("AFFINE", x, (scale, shift))


def add(a, b):
    if is_scalar_const(b):
        return affine(a, 1.0, float(b))
    if is_scalar_const(a):
        return affine(b, 1.0, float(a))
    return binary("ADD", a, b)


def sub(a, b):
    if is_scalar_const(b):
        return affine(a, 1.0, -float(b))
    if is_scalar_const(a):
        return affine(b, -1.0, float(a))
    return binary("SUB", a, b)


def mul(a, b):
    if is_scalar_const(b):
        return affine(a, float(b), 0.0)
    if is_scalar_const(a):
        return affine(b, float(a), 0.0)
    a, b = as_tensor(a), as_tensor(b)
    ca, cb = _cval(a), _cval(b)
    shp = bshape(a.shape, b.shape)
    if ca is not None and cb is not None:
        return constant(np.full(shp, ca * cb))
    if ca is not None and shp == tuple(b.shape):
        return affine(b, ca, 0.0)
    if cb is not None and shp == tuple(a.shape):
        return affine(a, cb, 0.0)
    return binary("MUL", a, b)


def div(a, b):
    if is_scalar_const(b):
        return affine(a, 1.0 / float(b), 0.0)
    if is_scalar_const(a):
        return affine(unary("RECIP", b), float(a), 0.0)
    return binary("DIV", a, b)


def square(x):
    return unary("SQUARE", x)


def gauss_logpdf(x, mu, var) -> Tensor:
    """Fused densities.gaussian (reference densities.py:25-27)."""
    x, mu, var = as_tensor(x), as_tensor(mu), as_tensor(var)
    return make("ew", (x, mu, var), {"f": "GAUSS_LOGPDF", "p": ()}, [bshape(x.shape, mu.shape, var.shape)]).outputs[0]


def where(c, a, b) -> Tensor:
    c, a, b = as_tensor(c), as_tensor(a), as_tensor(b)
    return make("ew", (c, a, b), {"f": "WHERE", "p": ()}, [bshape(c.shape, a.shape, b.shape)]).outputs[0]


def stop_gradient(x) -> Tensor:
    x = as_tensor(x)
    return make("stop_gradient", (x,), {}, [x.shape]).outputs[0]


def sum_to_shape(g: Tensor, shape) -> Tensor:
    """Reduce a broadcast gradient back to `shape`."""
    shape = tuple(shape)
    if g.shape == shape:
        return g
    nd = len(g.shape)
    lead = nd - len(shape)
    axes = list(range(lead))
    for i, s in enumerate(shape):
        if s == 1 and g.shape[lead + i] != 1:
            axes.append(lead + i)
    r = reduce_sum(g, axes, keepdims=True) if axes else g
    return reshape(r, shape)


def _ew_vjp(node, gs):
    g = gs[0]
    if g is None:
        return [None] * len(node.inputs)
    f, p = node.attrs["f"], node.attrs["p"]
    ins = node.inputs
    y = node.outputs[0]
    x = ins[0]
    one = lambda t, s: sum_to_shape(t, s.shape)
    if f == "NEG":
        return [unary("NEG", g)]
    if f == "EXP":
        return [mul(g, y)]
    if f == "LOG":
        return [div(g, x)]
    if f == "SQRT":
        return [div(affine(g, 0.5), y)]
    if f == "SQUARE":
        return [mul(affine(g, 2.0), x)]
    if f == "ABS":
        return [mul(g, unary("SIGN", x))]
    if f in ("SIGN", "STEP", "CLIPMASK", "GT", "GE", "LT", "LE", "EQ"):
        return [None] * len(ins)
    if f == "SIGMOID":
        return [binary("SIGMOID_GRAD", y, g)]
    if f == "TANH":
        return [binary("TANH_GRAD", y, g)]
    if f == "RELU":
        return [binary("RELU_GRAD", x, g)]
    if f == "SOFTPLUS":
        return [binary("SOFTPLUS_GRAD", x, g)]
    if f == "RECIP":
        return [unary("NEG", mul(g, square(y)))]
    if f == "RSQRT":
        return [affine(mul(g, mul(y, square(y))), -0.5)]
    if f == "AFFINE":
        return [affine(g, p[0])]
    if f == "CLIP":
        return [binary("CLIP_GRAD", x, g, p)]
    if f == "LGAMMA":
        return [mul(g, unary("DIGAMMA", x))]
    if f == "POWC":
        return [mul(affine(g, p[0]), unary("POWC", x, (p[0] - 1.0,)))]
    if f == "LOG1P":
        return [div(g, affine(x, 1.0, 1.0))]
    if f == "COPY":
        return [g]
    a, b = ins[0], ins[1] if len(ins) > 1 else None
    if f == "ADD":
        return [one(g, a), one(g, b)]
    if f == "SUB":
        return [one(g, a), one(unary("NEG", g), b)]
    if f == "MUL":
        return [one(mul(g, b), a), one(mul(g, a), b)]
    if f == "DIV":
        return [one(div(g, b), a), one(unary("NEG", div(mul(g, y), b)), b)]
    if f == "MAX":
        m = binary("GE", a, b)
        return [one(mul(g, m), a), one(mul(g, affine(m, -1.0, 1.0)), b)]
    if f == "MIN":
        m = binary("LE", a, b)
        return [one(mul(g, m), a), one(mul(g, affine(m, -1.0, 1.0)), b)]
    if f == "POW":
        ga = mul(g, mul(b, binary("POW", a, affine(b, 1.0, -1.0))))
        gb = mul(g, mul(y, unary("LOG", a)))
        return [one(ga, a), one(gb, b)]
    if f == "WHERE":
        c, aa, bb = ins
        z = affine(g, 0.0)
        return [None, one(where(c, g, z), aa), one(where(c, z, g), bb)]
    if f == "GAUSS_LOGPDF":
        xx, mu, var = ins
        n = make("ew", (xx, mu, var, g), {"f": "GAUSS_LOGPDF_GRAD", "p": ()}, [y.shape] * 3)
        return [one(n.outputs[0], xx), one(n.outputs[1], mu), one(n.outputs[2], var)]
    if f in ("SIGMOID_GRAD", "TANH_GRAD", "RELU_GRAD", "SOFTPLUS_GRAD", "CLIP_GRAD", "GAUSS_LOGPDF_GRAD", "DIGAMMA"):
        raise NotImplementedError("second-order gradients are not supported (%s)" % f)
    raise NotImplementedError("no VJP for elementwise op " + f)


def _ew_emit(plan, node):
    H = plan.H
    ins = [plan.buf(t) for t in node.inputs]
    outs = [plan.out(t) for t in node.outputs]
    f, p = node.attrs["f"], list(node.attrs["p"])
    nout = len(outs)
    plan.steps.append(lambda: H.ewise(f, ins, nout=nout, params=p, out=outs))


defop("ew", _ew_emit, _ew_vjp)
defop("stop_gradient", lambda plan, node: plan.alias(node.outputs[0], plan.buf(node.inputs[0])),
      lambda node, gs: [None])


# ------------------------------------------------------------------------------
# reductions
# ------------------------------------------------------------------------------
def _norm_axes(axes, nd):
    if axes is None:
        return tuple(range(nd))
    if isinstance(axes, (int, np.integer)):
        axes = [axes]
    return tuple(sorted(set(int(a) % nd if nd else 0 for a in axes)))


def _reduce(kind, x, axes=None, keepdims=False) -> Tensor:
    x = as_tensor(x)
    nd = len(x.shape)
    axes = _norm_axes(axes, nd)
    if not axes:
        return x
    if kind == "sum" and len(axes) == nd and x.size > 1:
        ll = _try_gauss_ll(x)
        if ll is not None:  # tf.reduce_sum(densities.gaussian(y, f * s, var)): one fused op (value + gradient pieces)
            return reshape(ll, [1] * nd if keepdims else [])
    # chain contiguous runs of reduced axes (each run is one [K1,R,K2] launch)
    runs, cur = [], [axes[0]]
    for a in axes[1:]:
        if a == cur[-1] + 1:
            cur.append(a)
        else:
            runs.append(cur)
            cur = [a]
    runs.append(cur)
    t = x
    for run in reversed(runs):  # reduce trailing runs first so earlier axis numbers stay valid
        shp = list(t.shape)
        K1 = int(np.prod(shp[: run[0]])) if run[0] > 0 else 1
        R = int(np.prod(shp[run[0] : run[-1] + 1]))
        K2 = int(np.prod(shp[run[-1] + 1 :])) if run[-1] + 1 < len(shp) else 1
        out_shape = shp[: run[0]] + [1] * len(run) + shp[run[-1] + 1 :]
        t = make("reduce", (t,), {"kind": kind, "K1": K1, "R": R, "K2": K2}, [tuple(out_shape)]).outputs[0]
    if not keepdims:
        t = reshape(t, [s for i, s in enumerate(x.shape) if i not in axes])
    return t


# ------------------------------------------------------------------------------
# fused Gaussian log-likelihood head
def _try_gauss_ll(x):
    """x = GAUSS_LOGPDF(y, mu, var) about to be summed over everything, var a single element and y, mu of full size
    (mu optionally f * scale with a single-element scale): return ll[1] of the fused op, else None."""
    n = x.node
    if n.op != "ew" or n.attrs["f"] != "GAUSS_LOGPDF":
        return None
    y, mu, var = n.inputs
    if var.size != 1 or y.size != x.size or mu.size != x.size or _squeeze_shape(y.shape) != _squeeze_shape(mu.shape):
        return None
    f, scale = mu, None
    mn = mu.node
    if mn.op == "ew" and mn.attrs["f"] == "MUL":
        a, b = mn.inputs
        if b.size == 1 and a.size == mu.size:
            f, scale = a, b
        elif a.size == 1 and b.size == mu.size:
            f, scale = b, a
    ins = (y, f, var) + ((scale,) if scale is not None else ())
    return make("gauss_ll", ins, {}, [(1,), tuple(f.shape), (1,), (1,)]).outputs[0]


def _gauss_ll_post(node, consumers, outputs):
    """The elementwise tail TF autodiff appends to the likelihood head -- gmu = c * dmu (c: the constant upstream factor
    of the likelihood term, N / n), then scale * gmu -- when nothing else reads those values: (c, the tensor that is
    d objective / d f, [absorbed nodes]), else None.  The head then writes that gradient itself (hb_gauss_ll_post)."""
    dmu = node.outputs[1]
    if dmu in outputs:
        return None
    c1 = consumers.get(dmu, [])
    if len(c1) != 1 or c1[0].op != "ew" or c1[0].attrs["f"] != "AFFINE" or len(c1[0].inputs) != 1:
        return None
    m1 = c1[0]
    pp = list(m1.attrs["p"]) + [0.0, 0.0]
    if pp[1] != 0.0:
        return None
    gmu = m1.outputs[0]
    if len(node.inputs) <= 3:
        return (float(pp[0]), gmu, [m1]) if gmu.shape == dmu.shape else None
    if gmu in outputs:
        return None
    c2 = consumers.get(gmu, [])
    if len(c2) != 1 or c2[0].op != "ew" or c2[0].attrs["f"] != "MUL" or len(c2[0].inputs) != 2:
        return None
    m2 = c2[0]
    other = m2.inputs[0] if m2.inputs[1] is gmu else m2.inputs[1]

    def root(t):
        while t.node.op == "reshape":
            t = t.node.inputs[0]
        return t

    if other.size != 1 or root(other) is not root(node.inputs[3]) or m2.outputs[0].size != dmu.size:
        return None
    return (float(pp[0]), m2.outputs[0], [m1, m2])


def _gauss_ll_emit(plan, node):
    H = plan.H
    fusedh = plan._gll_fused.get(node.id)
    if fusedh is not None:
        # the per-point part ran inside the forward strip kernel (hb_sgp_fwd_gauss, see _sgp_emit): what is left is the
        # fold of its partial sums, emitted right in front of the first reader of (ll, dscale, dvar)
        part, units = fusedh
        ll, _, ds, dv = (plan.out(t) for t in node.outputs)
        step = lambda: H.gauss_ll_fold(part, units, ll, ds, dv)
        plan.chain_kind[id(step)] = "full"
        plan._pending.append((step, "gauss_ll_fold", node, {node.outputs[0], node.outputs[2], node.outputs[3]}))
        return
    y, f, var = (plan.buf(t) for t in node.inputs[:3])
    scale = plan.buf(node.inputs[3]) if len(node.inputs) > 3 else None
    outs = tuple(plan.out(t) for t in node.outputs)
    post = plan._gll_post.get(node.id)
    if post is not None:
        c, t_fbar, _ = post
        fbar = plan.out(t_fbar)
        step = lambda: H.gauss_ll(y, f, scale, var, out=outs, post=c, fbar=fbar)
    else:
        step = lambda: H.gauss_ll(y, f, scale, var, out=outs)
    plan.steps.append(step)
    plan.chain_kind[id(step)] = "full"     # hb_gauss_ll records itself into a serial chain (csrc/chain.cuh)


def _gauss_ll_vjp(node, gs):
    g = gs[0]
    if any(x is not None for x in gs[1:]):
        raise NotImplementedError("gradients through the saved derivative outputs of gauss_ll")
    if g is None:
        return [None] * len(node.inputs)
    y, f, var = node.inputs[:3]
    scale = node.inputs[3] if len(node.inputs) > 3 else None
    ll, dmu, dsc, dvr = node.outputs
    g1 = reshape(g, [1])
    gmu = mul(g1, dmu)  # d/d mu, shape of f
    res = [reshape(unary("NEG", gmu), y.shape),
           reshape(mul(reshape(scale, [1]), gmu) if scale is not None else gmu, f.shape),
           reshape(mul(g1, dvr), var.shape)]
    if scale is not None:
        res.append(reshape(mul(g1, dsc), scale.shape))
    return res


defop("gauss_ll", _gauss_ll_emit, _gauss_ll_vjp)


def reduce_sum(x, axis=None, keepdims=False, keep_dims=None) -> Tensor:
    if keep_dims is not None:
        keepdims = keep_dims
    return _reduce("sum", x, axis, keepdims)


def reduce_max(x, axis=None, keepdims=False, keep_dims=None) -> Tensor:
    if keep_dims is not None:
        keepdims = keep_dims
    return _reduce("max", x, axis, keepdims)


def reduce_mean(x, axis=None, keepdims=False, keep_dims=None) -> Tensor:
    x = as_tensor(x)
    if keep_dims is not None:
        keepdims = keep_dims
    axes = _norm_axes(axis, len(x.shape))
    cnt = int(np.prod([x.shape[a] for a in axes])) if axes else 1
    return affine(_reduce("sum", x, axis, keepdims), 1.0 / max(cnt, 1))


def _reduce_emit(plan, node):
    H = plan.H
    if node.id in plan._fused_colsum:
        plan.out(node.outputs[0])    # written by the weight-gradient GEMM that streams the same operand (hb_matmul_colsum)
        return
    x, out = plan.buf(node.inputs[0]), plan.out(node.outputs[0])
    a = node.attrs
    op = H.RED_SUM if a["kind"] == "sum" else H.RED_MAX
    plan.steps.append(lambda: H.reduce_mid(x, a["K1"], a["R"], a["K2"], op=op, out=out))


def _reduce_vjp(node, gs):
    g = gs[0]
    if g is None:
        return [None]
    x = node.inputs[0]
    if node.attrs["kind"] == "sum":
        return [broadcast_to(g, x.shape)]
    mask = binary("EQ", x, node.outputs[0])
    return [mul(mask, g)]


defop("reduce", _reduce_emit, _reduce_vjp)


# ------------------------------------------------------------------------------
# shape ops
# ------------------------------------------------------------------------------
def reshape(x, shape) -> Tensor:
    x = as_tensor(x)
    shape = [int(s) for s in shape]
    if -1 in shape:
        known = int(np.prod([s for s in shape if s != -1])) if len(shape) > 1 else 1
        shape[shape.index(-1)] = x.size // max(known, 1)
    shape = tuple(shape)
    if int(np.prod(shape)) != x.size:
        raise ValueError("cannot reshape %s to %s" % (x.shape, shape))
    if shape == x.shape:
        return x
    if x.node.op == "leaf:const":
        return constant(np.asarray(x.node.attrs["value"]).reshape(shape))
    return make("reshape", (x,), {"shape": shape}, [shape]).outputs[0]


defop("reshape", lambda plan, node: plan.alias(node.outputs[0], plan.buf(node.inputs[0]).view(node.outputs[0].shape)),
      lambda node, gs: [None if gs[0] is None else reshape(gs[0], node.inputs[0].shape)])


def expand_dims(x, axis) -> Tensor:
    x = as_tensor(x)
    shp = list(x.shape)
    axis = axis if axis >= 0 else axis + len(shp) + 1
    shp.insert(axis, 1)
    return reshape(x, shp)


def squeeze(x, axis=None) -> Tensor:
    x = as_tensor(x)
    shp = list(x.shape)
    if axis is None:
        shp = [s for s in shp if s != 1]
    else:
        axs = _norm_axes(axis, len(shp))
        shp = [s for i, s in enumerate(shp) if i not in axs]
    return reshape(x, shp)


def _contig_strides(shape):
    st, acc = [0] * len(shape), 1
    for d in range(len(shape) - 1, -1, -1):
        st[d] = acc
        acc *= shape[d]
    return st


def strided_view(x, out_shape, strides, offset=0) -> Tensor:
    """out[idx] = x.flat[offset + sum idx*strides] (materialised copy; covers
    transpose / slice / broadcast / tile).  Linear in x."""
    x = as_tensor(x)
    out_shape = tuple(int(d) for d in out_shape)
    # a view that keeps the memory order of every non-unit dim (e.g. the transpose of an [n,1] column) is a reshape
    if int(offset) == 0 and int(np.prod(out_shape)) == x.size and all(
            d == 1 or int(s) == c for d, s, c in zip(out_shape, strides, _contig_strides(out_shape))):
        return reshape(x, out_shape)
    return make("strided", (x,), {"shape": tuple(out_shape), "strides": tuple(int(s) for s in strides),
                                  "offset": int(offset)}, [tuple(out_shape)]).outputs[0]


def _column_block(node):
    """(L, ld, offset) when a strided node is a column block [rows, L] of a row-major [rows, ld] matrix, else None."""
    a = node.attrs
    xs = node.inputs[0].shape
    if len(a["shape"]) != 2 or len(xs) != 2:
        return None
    rows, L = a["shape"]
    if a["strides"] != (xs[1], 1) or rows != xs[0] or not 0 <= a["offset"] or a["offset"] + L > xs[1]:
        return None
    return L, xs[1], a["offset"]


def _strided_emit(plan, node):
    H = plan.H
    y = node.outputs[0]
    cb = _column_block(node)
    cons = plan._consumers.get(y, [])
    if (cb is not None and cons and y not in plan.outputs and y not in plan._bind
            and all((c.op == "diag_sample_kl" and y in c.inputs[:2] and (len(c.inputs) < 3 or c.inputs[2] is not y))
                    or (c.op == "diag_sample_kl_grad" and c.inputs[0] is y and y not in c.inputs[1:]) for c in cons)):
        # the mean / log-std halves of an encoder output feeding the diagonal sampler (and `s` again in its VJP):
        # the kernels read the column block in place (row stride = the source's width), no copy is made
        plan._lazy_cols[y] = (plan.buf(node.inputs[0]), cb)
        return
    x, out = plan.buf(node.inputs[0]), plan.out(y)
    a = node.attrs
    shp = list(a["shape"])
    ostr = _contig_strides(shp)
    plan.steps.append(lambda: H.copy_nd(x, list(a["strides"]), out, ostr, shp, src_off=a["offset"]))


def _strided_vjp(node, gs):
    g = gs[0]
    if g is None:
        return [None]
    x = node.inputs[0]
    a = node.attrs
    if 0 in [s for s, d in zip(a["strides"], a["shape"]) if d > 1]:
        # broadcast dims: sum them first, then scatter the rest
        keep = [i for i, (s, d) in enumerate(zip(a["strides"], a["shape"])) if not (s == 0 and d > 1)]
        red = [i for i in range(len(a["shape"])) if i not in keep]
        g = reduce_sum(g, red, keepdims=False)
        shape = [a["shape"][i] for i in keep]
        strides = [a["strides"][i] for i in keep]
    else:
        shape, strides = list(a["shape"]), list(a["strides"])
    return [make("scatter_strided", (g,), {"xshape": x.shape, "shape": tuple(shape), "strides": tuple(strides),
                                          "offset": a["offset"]}, [x.shape]).outputs[0]]


defop("strided", _strided_emit, _strided_vjp)


def _scatter_emit(plan, node):
    H = plan.H
    g, out = plan.buf(node.inputs[0]), plan.out(node.outputs[0])
    a = node.attrs
    shp = list(a["shape"])
    istr = _contig_strides(shp)
    covers = int(np.prod(shp)) == int(np.prod(a["xshape"]))

    def step():
        if not covers:
            H.fill(out, 0.0)
        H.copy_nd(g, istr, out, list(a["strides"]), shp, dst_off=a["offset"])

    plan.steps.append(step)


defop("scatter_strided", _scatter_emit,
      lambda node, gs: [None if gs[0] is None else strided_view(gs[0], node.attrs["shape"], node.attrs["strides"],
                                                                 node.attrs["offset"])])


def transpose(x, perm=None) -> Tensor:
    x = as_tensor(x)
    nd = len(x.shape)
    perm = list(range(nd - 1, -1, -1)) if perm is None else [int(p) % nd for p in perm]
    if perm == list(range(nd)):
        return x
    st = _contig_strides(x.shape)
    return strided_view(x, [x.shape[p] for p in perm], [st[p] for p in perm])


def matrix_transpose(x) -> Tensor:
    nd = len(as_tensor(x).shape)
    return transpose(x, list(range(nd - 2)) + [nd - 1, nd - 2])


def broadcast_to(x, shape) -> Tensor:
    """Broadcast (numpy rules).  Elementwise consumers read the small operand in place; a copy is
    only materialised for consumers that need the full array."""
    x = as_tensor(x)
    shape = tuple(int(s) for s in shape)
    if x.shape == shape:
        return x
    nd = len(shape)
    xs = (1,) * (nd - len(x.shape)) + tuple(x.shape)
    for i in range(nd):
        if xs[i] != shape[i] and xs[i] != 1:
            raise ValueError("cannot broadcast %s to %s" % (x.shape, shape))
    return make("bcast", (x,), {"shape": shape}, [shape]).outputs[0]


def _bcast_emit(plan, node):
    H = plan.H
    t = node.outputs[0]
    src = plan.buf(node.inputs[0])
    nd = len(t.shape)
    xs = (1,) * (nd - len(src.shape)) + tuple(src.shape)
    cons = plan._consumers.get(t, [])
    if cons and all(c.op == "ew" for c in cons) and t not in plan.outputs and t not in plan._bind:
        plan.alias(t, src.view(xs))  # every consumer broadcasts its operands itself
        return
    out = plan.out(t)
    st = _contig_strides(xs)
    strides = [0 if (xs[i] == 1 and t.shape[i] != 1) else st[i] for i in range(nd)]
    shp = list(t.shape)
    ostr = _contig_strides(shp)
    plan.steps.append(lambda: H.copy_nd(src, strides, out, ostr, shp))


defop("bcast", _bcast_emit, lambda node, gs: [None if gs[0] is None else sum_to_shape(gs[0], node.inputs[0].shape)])


def tile(x, multiples) -> Tensor:
    x = as_tensor(x)
    mult = [int(m) for m in multiples]
    assert len(mult) == len(x.shape)
    # view x as [1,s0,1,s1,...] broadcast to [m0,s0,m1,s1,...] then merge
    st = _contig_strides(x.shape)
    shape, strides = [], []
    for m, s, t in zip(mult, x.shape, st):
        shape += [m, s]
        strides += [0, t]
    v = strided_view(x, shape, strides)
    return reshape(v, [m * s for m, s in zip(mult, x.shape)])


def slice_(x, begin, size) -> Tensor:
    x = as_tensor(x)
    begin = [int(b) for b in begin]
    size = [int(x.shape[i] - begin[i]) if int(s) == -1 else int(s) for i, s in enumerate(size)]
    st = _contig_strides(x.shape)
    for b, s, d in zip(begin, size, x.shape):
        if b < 0 or b + s > d:
            raise ValueError("slice out of range")
    if list(size) == list(x.shape):
        return x
    return strided_view(x, size, st, offset=sum(b * t for b, t in zip(begin, st)))


def getitem(x, key) -> Tensor:
    x = as_tensor(x)
    if not isinstance(key, tuple):
        key = (key,)
    if any(k is Ellipsis for k in key):
        i = key.index(Ellipsis)
        nfill = len(x.shape) - sum(1 for k in key if k is not None and k is not Ellipsis)
        key = key[:i] + (slice(None),) * nfill + key[i + 1 :]
    begin, size, final = [], [], []
    dim = 0
    for k in key:
        if k is None:
            final.append(1)
            continue
        d = x.shape[dim]
        if isinstance(k, (int, np.integer)):
            kk = int(k) % d
            begin.append(kk)
            size.append(1)
        elif isinstance(k, slice):
            s, e, stp = k.indices(d)
            if stp != 1:
                raise NotImplementedError("strided slicing")
            begin.append(s)
            size.append(max(e - s, 0))
            final.append(max(e - s, 0))
        else:
            raise TypeError("unsupported index %r" % (k,))
        dim += 1
    while dim < len(x.shape):
        begin.append(0)
        size.append(x.shape[dim])
        final.append(x.shape[dim])
        dim += 1
    return reshape(slice_(x, begin, size), final)


def concat(values, axis) -> Tensor:
    vals = [as_tensor(v) for v in values]
    nd = len(vals[0].shape)
    axis = axis % nd
    oshape = list(vals[0].shape)
    oshape[axis] = sum(v.shape[axis] for v in vals)
    return make("concat", tuple(vals), {"axis": axis}, [tuple(oshape)]).outputs[0]


def _concat_emit(plan, node):
    H = plan.H
    if node.id in plan._fused_concat:
        return  # its producer wrote the parts in place (see _diag_skl_grad_emit)
    out = plan.out(node.outputs[0])
    axis = node.attrs["axis"]
    ostr = _contig_strides(node.outputs[0].shape)
    off = 0
    for t in node.inputs:
        src = plan.buf(t)
        shp = list(t.shape)
        istr = _contig_strides(shp)
        o = off * ostr[axis]
        off += t.shape[axis]
        if src.data_ptr() == out.data_ptr() + o * out.element_size() and all(d == 1 for d in node.outputs[0].shape[:axis]):
            continue   # written in place by its producer (Plan._place)
        plan.steps.append(lambda src=src, istr=istr, shp=shp, o=o: H.copy_nd(src, istr, out, ostr, shp, dst_off=o))


def _concat_vjp(node, gs):
    g = gs[0]
    if g is None:
        return [None] * len(node.inputs)
    axis = node.attrs["axis"]
    res, off = [], 0
    for t in node.inputs:
        begin = [0] * len(t.shape)
        begin[axis] = off
        res.append(slice_(g, begin, t.shape))
        off += t.shape[axis]
    return res


defop("concat", _concat_emit, _concat_vjp)


def stack(values, axis=0) -> Tensor:
    vals = [as_tensor(v) for v in values]
    nd = len(vals[0].shape) + 1
    axis = axis % nd
    return concat([expand_dims(v, axis) for v in vals], axis)


# ------------------------------------------------------------------------------
# linear algebra
# ------------------------------------------------------------------------------
def matmul(a, b, transpose_a=False, transpose_b=False, bias=None, act="none") -> Tensor:
    """tf.matmul with batch dims (equal, or absent on one side); optional fused
    bias (+activation) epilogue = the reference MatBias layer (nn.py:31-32)."""
    a, b = as_tensor(a), as_tensor(b)
    if len(a.shape) < 2 or len(b.shape) < 2:
        raise ValueError("matmul needs rank >= 2 operands")
    m, k = (a.shape[-1], a.shape[-2]) if transpose_a else (a.shape[-2], a.shape[-1])
    k2, n = (b.shape[-1], b.shape[-2]) if transpose_b else (b.shape[-2], b.shape[-1])
    if k != k2:
        raise ValueError("matmul inner dimensions differ: %s x %s" % (a.shape, b.shape))
    la, lb = a.shape[:-2], b.shape[:-2]
    if la and lb and la != lb:
        raise ValueError("matmul batch dimensions differ: %s vs %s" % (la, lb))
    lead = la if la else lb
    ins = (a, b) if bias is None else (a, b, as_tensor(bias))
    return make("matmul", ins, {"ta": bool(transpose_a), "tb": bool(transpose_b), "act": act},
                [tuple(lead) + (m, n)]).outputs[0]


def _matmul_emit(plan, node):
    H = plan.H
    a, b = plan.buf(node.inputs[0]), plan.buf(node.inputs[1])
    bias = plan.buf(node.inputs[2]) if len(node.inputs) > 2 else None
    at = node.attrs
    y = node.outputs[0]
    g = getattr(plan, "_mm_head", {}).get(node.id)
    if g is not None:
        # the Gaussian likelihood head of this layer runs in the product's epilogue (hb_matmul_gauss): f is never written
        units = H.matmul_gauss_units(y.shape[0], node.inputs[0].shape[1], y.shape[1], plan.dtype)
        ok = (units > 0 and a.stride(1) == 1 and a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0 and b.stride(1) == 1
              and (bias is None or bias.numel() == y.shape[1]))
        if ok:
            def early(t):
                shp = t.shape
                while t.node.op == "reshape" and t not in plan._buf:
                    t = t.node.inputs[0]
                return plan.buf(t).view(shp)

            post = plan._gll_post.get(g.id)
            head = dict(y=early(g.inputs[0]), var=early(g.inputs[2]), scale=early(g.inputs[3]) if len(g.inputs) > 3 else None,
                        dmu=plan.out(g.outputs[1]), post=post[0] if post else 0.0, fbar=plan.out(post[1]) if post else None,
                        part=plan.scratch((3 * units,)), units=units)
            plan._gll_fused[g.id] = (head["part"], units)
            plan.pin_side_reads([g.inputs[0], g.inputs[2]] + list(g.inputs[3:4]))
            plan.steps.append(lambda: H.matmul_gauss(a, b, bias, head))
            return
        # (the head was paired at plan time but this call cannot carry it: the plain product, and the head as a launch of its own)
        plan._gll_in_sgp.pop(g.id, None)
    # a triangular / Phi / symmetrising matutil that is the only consumer becomes the GEMM's epilogue
    cons = plan._consumers.get(y, [])
    epi = 0
    if (not at.get("actgrad") and len(cons) == 1 and cons[0].op == "matutil" and bias is None and at["act"] == "none" and y.shape[-1] == y.shape[-2]
            and y not in plan.outputs and y not in plan._bind):
        ma = cons[0].attrs
        if ma["mode"] == 2:
            epi = H.MM_PHI_OUT
        elif ma["mode"] == 3:
            epi = H.MM_SYM_OUT
        elif ma["mode"] == 4:
            epi = H.MM_SYMLOW_OUT
        elif ma["mode"] == 0 and ma["lower"] < 0 and ma["upper"] == 0:
            epi = H.MM_TRIL_OUT
    if epi:
        out = plan.out(cons[0].outputs[0])
        plan._fused_matutil.add(cons[0].id)
    else:
        out = plan.out(y)
    csn = plan._colsum_of.get(node.id)
    if csn is not None and not epi:
        cout = plan.out(csn.outputs[0])
        plan.steps.append(lambda: H.matmul_colsum(a, b, out=out, colsum=cout))
        return
    if at.get("actgrad"):
        # third input = the activation output Y: C = (op(A) op(B)) * act'(Y)
        plan.steps.append(lambda: H.matmul(a, b, transA=at["ta"], transB=at["tb"], act=at["actgrad"], actgrad=bias, out=out))
        return
    # the in-workgroup split-K form (small square-ish fp32 products: the Cholesky VJP) hosts pending side jobs
    ysh = y.shape
    kk = node.inputs[0].shape[-2] if at["ta"] else node.inputs[0].shape[-1]
    wgk_like = (plan.dtype == plan.torch.float32 and bias is None and ysh[-1] % 32 == 0 and ysh[-2] % 32 == 0 and kk % 128 == 0
                and 128 <= kk <= 4096 and (ysh[-1] // 32) * (ysh[-2] // 32) * int(np.prod(ysh[:-2]) if len(ysh) > 2 else 1) <= 1024)
    host = wgk_like and plan.attach_side(node)
    # The product whose result is Kbar of a Gram matrix K(X, X) and feeds nothing but that Gram matrix's VJP (the last
    # product of the Cholesky VJP): the VJP runs in the product's epilogue (hb_matmul_gram_vjp), gram_grad keeps the fold.
    # settings.runtime.gram_vjp_in_product, OFF by default: one launch less, but the tile partials' way through memory
    # (write-through, drain, counter, last arriver's loads) is a longer dependent chain than the launch it replaces --
    # cfg 2: 197.3 -> 202.4 us per step (tools/ab_step.py)
    gg = cons[0] if (len(cons) == 1 and cons[0].op == "gram_grad") else None
    gp = "Gram VJP inside the product that computes Kbar (hb_matmul_gram_vjp)"
    if gg is not None and not epi and bias is None:
        from ._settings import settings as _st
        tX, tX2, tell, tg = gg.inputs
        B, BX, BX2, n, n2, d, sX, sX2, sEll, dl = _gram_layout(tX, tX2, tell)
        ok = (bool(getattr(_st.runtime, "gram_vjp_in_product", False)) and wgk_like and gg.attrs.get("sym") and tg is y
              and node.attrs.get("sym_result") and gg.attrs["kind"] == "rbf" and not (BX == 1 and B > 1) and y not in plan.outputs
              and y not in plan._bind and n == ysh[-1] and H.matmul_gram_vjp_ok(n, kk, B, d, plan.dtype))
        plan.note(gp, gg, ok, "" if ok else "needs the in-workgroup split-K form, the symmetric one-pass VJP of a UnitRBF Gram and d <= 4")
        if ok:
            X, ell = plan.buf(tX), plan.buf(tell)
            oX = plan.out(gg.outputs[0])
            lws = plan.scratch((max(B * n * d, 1),))
            part = plan.scratch((B * (n // 32) * (n // 32) * 32 * 2 * d,))
            counters = plan.torch.zeros((B * (n // 32),), dtype=plan.torch.int32, device=plan.device)
            plan._gram_in_mm[gg.id] = (lws, (n if sEll != 0 else B * n), d, dl, (B if sEll != 0 else 1))
            plan.steps.append(lambda: H.matmul_gram_vjp(a, b, out, at["ta"], at["tb"], X, sX, ell, sEll, dl, d, oX, lws, part, counters))
            if host:
                plan.steps.append(lambda: H.side_flush())
            return
    plan.steps.append(lambda: H.matmul(a, b, transA=at["ta"], transB=at["tb"], bias=bias, act=at["act"], out=out,
                                       epilogue=epi))
    if host:
        plan.steps.append(lambda: H.side_flush())
    if csn is not None:
        # the planner paired this product with the column sums of its right operand (Plan: _colsum_of), but the product
        # takes a matutil epilogue, which hb_matmul_colsum does not have: the absorbed reduction runs as a launch of its
        # own (its node emits nothing by itself, _reduce_emit)
        cout, ra = plan.out(csn.outputs[0]), csn.attrs
        plan.steps.append(lambda: H.reduce_mid(b, ra["K1"], ra["R"], ra["K2"], op=H.RED_SUM, out=cout))


def _sum_lead(g, t):
    """sum a batched gradient over leading dims the operand `t` does not have."""
    extra = len(g.shape) - len(t.shape)
    return reduce_sum(g, list(range(extra))) if extra > 0 else g


def _matmul_vjp(node, gs):
    g = gs[0]
    if g is None:
        return [None] * len(node.inputs)
    a, b = node.inputs[0], node.inputs[1]
    ta, tb, act = node.attrs["ta"], node.attrs["tb"], node.attrs["act"]
    y = node.outputs[0]
    if node.attrs.get("actgrad"):
        raise NotImplementedError("second derivatives through a fused activation-gradient GEMM are not supported")
    gn = g.node
    if (act != "none" and gn.op == "matmul" and len(gn.inputs) == 2 and gn.attrs["act"] == "none"
            and not gn.attrs.get("actgrad") and tuple(g.shape) == tuple(y.shape)):
        # the incoming gradient is itself a GEMM (the next layer's dx): the activation derivative becomes that
        # GEMM's epilogue instead of a separate pass over the [n, H] activations (reference nn.py:79-84 backward)
        g = make("matmul", (gn.inputs[0], gn.inputs[1], y), {"ta": gn.attrs["ta"], "tb": gn.attrs["tb"], "act": "none",
                                                             "actgrad": act}, [tuple(g.shape)]).outputs[0]
    elif act == "sigmoid":
        g = binary("SIGMOID_GRAD", y, g)
    elif act == "tanh":
        g = binary("TANH_GRAD", y, g)
    elif act == "relu":
        g = binary("RELU_GRAD", y, g)  # y > 0  <=>  pre-activation > 0
    if not ta:
        ga = matmul(g, b, transpose_b=not tb)
    else:
        ga = matmul(b, g, transpose_a=tb, transpose_b=True)
    if not tb:
        gb = matmul(a, g, transpose_a=not ta)
    else:
        gb = matmul(g, a, transpose_a=True, transpose_b=ta)
    res = [_sum_lead(ga, a), _sum_lead(gb, b)]
    if len(node.inputs) > 2:
        res.append(sum_to_shape(g, node.inputs[2].shape))
    return res


defop("matmul", _matmul_emit, _matmul_vjp)


def matutil(x, mode, lower=-1, upper=-1, alpha=0.0) -> Tensor:
    x = as_tensor(x)
    return make("matutil", (x,), {"mode": mode, "lower": int(lower), "upper": int(upper), "alpha": float(alpha)},
                [x.shape]).outputs[0]


def _known_lower(t) -> bool:
    """True when t is lower-triangular by construction (its strict upper part is exactly zero)."""
    n = t.node
    if n.op in ("cholesky", "trinv"):
        return True
    if n.op == "sgp_grad":
        return t.index == 0
    if n.op == "matutil":
        return n.attrs["mode"] == 2 or (n.attrs["mode"] == 0 and n.attrs["upper"] == 0)
    if n.op == "stop_gradient":
        return _known_lower(n.inputs[0])
    return False


def band_part(x, num_lower, num_upper):
    x = as_tensor(x)
    if int(num_lower) < 0 and int(num_upper) == 0 and _known_lower(x):
        return x  # tril of a matrix that is already lower-triangular
    return matutil(x, 0, num_lower, num_upper)


def add_eye(x, alpha):
    return matutil(x, 1, alpha=alpha)


def _matutil_emit(plan, node):
    if node.id in plan._fused_matutil:
        return  # folded into the producing GEMM's epilogue
    H = plan.H
    x, out = plan.buf(node.inputs[0]), plan.out(node.outputs[0])
    a = node.attrs
    plan.steps.append(lambda: H.matutil(x, a["mode"], a["lower"], a["upper"], a["alpha"], out=out))


def _matutil_vjp(node, gs):
    g = gs[0]
    if g is None:
        return [None]
    a = node.attrs
    if a["mode"] == 0:
        return [matutil(g, 0, a["lower"], a["upper"])]
    if a["mode"] == 1:
        return [g]
    if a["mode"] == 2:
        return [matutil(g, 2)]
    if a["mode"] == 4:
        return [matutil(matutil(g, 3), 2)]   # y_ij = x_{max,min} / 2:  xbar = Phi((g + g^T) / 2)
    return [matutil(g, 3)]


defop("matutil", _matutil_emit, _matutil_vjp)


def diag_part(x) -> Tensor:
    x = as_tensor(x)
    n = x.shape[-1]
    st = _contig_strides(x.shape)
    return strided_view(x, x.shape[:-2] + (n,), st[:-2] + [n + 1])


def cholesky(a) -> Tensor:
    """Lower Cholesky factor, batched (tf.cholesky; reference gp/kernels.py:101)."""
    a = as_tensor(a)
    if a.shape[-1] != a.shape[-2]:
        raise ValueError("cholesky needs square matrices")
    return make("cholesky", (a,), {}, [a.shape]).outputs[0]


def _cholesky_emit(plan, node):
    H = plan.H
    # (the matrix itself does not exist when the Gram launch was folded into this one: plan._gram_for_chol)
    a = None if node.id in plan._gram_for_chol else plan.buf(node.inputs[0])
    out = plan.out(node.outputs[0])
    B = int(np.prod(node.outputs[0].shape[:-2])) if len(node.outputs[0].shape) > 2 else 1
    info = plan.new_info(B, "cholesky#%d" % node.id)
    # trinv(cholesky(a)) -- SparseGP's whitening, the Cholesky VJP -- is lowered to the fused
    # factor+inverse launches: the inverse rides along for free (csrc/linalg.hip)
    inv_node = next((c for c in plan._consumers.get(node.outputs[0], ()) if c.op == "trinv"), None)
    if inv_node is not None and (a is None or a.data_ptr() != out.data_ptr()):
        w = plan.out(inv_node.outputs[0])
        # exchange area + sync words of the persistent launch: zero-filled once, left zero by every call
        ws = plan.torch.zeros((max(H.cholesky_ws_elems(B, node.outputs[0].shape[-1], plan.dtype), 1),), dtype=plan.dtype,
                              device=plan.device)
        plan._fused_trinv.add(inv_node.id)
        # fragment-major copies of W / W^T for the M^2 n contractions that consume this inverse (csrc/sgp.hip)
        frag = None
        M = node.outputs[0].shape[-1]
        users = list(plan._consumers.get(inv_node.outputs[0], ()))
        users += [c2 for c in users if c.op == "stop_gradient" for c2 in plan._consumers.get(c.outputs[0], ())]
        bf3 = False
        if plan.dtype == plan.torch.float32 and M % 32 == 0 and M >= 32 and any(c.op in ("sgp", "sgp_grad") for c in users):
            from ._settings import settings as _st

            # settings.numerics.contraction = bf16x3: the M^2 n contractions take three-term bf16 operands
            # (fp32-level accuracy at the bf16 MFMA rate, include/henbun_hip.h HB_PREC_BF16X3); M <= 512 only
            bf3 = str(getattr(_st.numerics, "contraction", "native")) == "bf16x3" and M <= 512
            frag = plan.scratch(((5 if bf3 else 2) * max(int(np.prod(node.outputs[0].shape)), 1),))
            plan._wfrag[inv_node.outputs[0]] = (frag, bf3)
        # launch 0 of the 64-column chain hosts pending side jobs (minibatch gather, the sample of q(u))
        host = plan.dtype == plan.torch.float32 and M % 64 == 0 and plan.attach_side(node)
        plan.note("factor + inverse in one persistent launch (chol_persist_kernel)", node,
                  H.cholesky_persistent_shape(B, M, plan.dtype), "needs fp32 and M % 64 == 0: the launch-chain form runs")
        gk = plan._gram_for_chol.get(node.id)
        if gk is not None:
            gX, gell, gkind, gjit = gk
            factor = lambda: H.gram_cholesky_inverse(gX, gell, gjit, kind=gkind, out=out, inv=w, info=info, ws=ws, frag=frag,
                                                     frag_bf16x3=bf3)
        else:
            factor = lambda: H.cholesky_inverse(a, out=out, inv=w, info=info, ws=ws, frag=frag, frag_bf16x3=bf3)
        # Early-start forward (csrc/sgp.hip): _sgp_emit may hand the forward contraction that consumes this inverse to
        # this step -- it is then recorded first and launched INSIDE the persistent factorisation's grid.
        cell = {"rider": None}

        def chol_step():
            r = cell["rider"]
            if r is None:
                return factor()
            H.sgp_rider_begin()
            r()                       # recorded, not launched
            factor()                  # ONE launch: factorisation + pending side jobs + the recorded forward
            H.sgp_rider_flush()       # (a no-op unless the factorisation could not take it)

        cell["step"] = chol_step
        cell["epos"] = len(plan._emitted)
        if frag is not None and not bf3 and H.cholesky_persistent_shape(B, M, plan.dtype):
            plan._chol_rider[inv_node.outputs[0]] = cell
        plan.steps.append(chol_step)
        if host:
            plan.steps.append(lambda: H.side_flush())
        return
    plan.steps.append(lambda: H.cholesky(a, out=out, info=info))


def _sgp_frag_exchange(plan, node, prec):
    """True when the A output of sparse-GP op `node` only feeds its own VJP in column-strip form (no x gradient): A is
    then exchanged fragment-major and its row-major copy is never written."""
    H = plan.H
    zsh, xsh = node.inputs[1].shape, node.inputs[0].shape
    E, M, n, d, P = int(np.prod(zsh[:-2])) if len(zsh) > 2 else 1, zsh[-2], xsh[-2], xsh[-1], node.inputs[5].shape[-2]
    if plan.dtype != plan.torch.float32 or not H.sgp_strip_path(E, n, M, d, P, prec):
        return False
    users = list(plan._consumers.get(node.outputs[1], ()))
    return bool(users) and all(c.op == "sgp_grad" and not plan.needed(c.outputs[4]) for c in users)


def _cholesky_vjp(node, gs):
    """Murray (2016) / TF _CholeskyGrad:  P = Phi(L^T Lbar);  S = L^-T P L^-1;  Abar = (S+S^T)/2.

    Since (S + S^T)/2 = L^-T ((P + P^T)/2) L^-1, the symmetrisation is applied to the M x M operand instead of the
    result: Psym = (P + P^T)/2, i.e. Psym_ij = Q_{max(i,j),min(i,j)} / 2 with Q = L^T Lbar (matutil mode 4, the
    epilogue of the first product), and Abar = L^-T Psym L^-1 comes out symmetric from two plain products -- the
    symmetrising third product carried a second accumulator per workgroup and took 11.2 us against 6.3 at cfg 2."""
    g = gs[0]
    if g is None:
        return [None]
    L = node.outputs[0]
    W = trinv(L)
    P = matutil(matmul(L, band_part(g, -1, 0), transpose_a=True), 4)
    res = matmul(matmul(W, P, transpose_a=True), W)
    res.node.attrs["sym_result"] = True   # L^-T Psym L^-1: symmetric (to rounding); gram_grad then skips the transposed reads
    return [res]


defop("cholesky", _cholesky_emit, _cholesky_vjp)


def trinv(L) -> Tensor:
    """Inverse of a lower-triangular matrix (batched)."""
    L = as_tensor(L)
    return make("trinv", (L,), {}, [L.shape]).outputs[0]


def _trinv_vjp(node, gs):
    g = gs[0]
    if g is None:
        return [None]
    W = node.outputs[0]
    # d(L^-1) = -W dL W  ->  Lbar = -tril(W^T Wbar W^T)
    t = matmul(matmul(W, band_part(g, -1, 0), transpose_a=True), W, transpose_b=True)
    return [band_part(unary("NEG", t), -1, 0)]


def _trinv_emit(plan, node):
    if node.id in plan._fused_trinv:
        return  # produced by the fused cholesky+inverse step
    H, l, o = plan.H, plan.buf(node.inputs[0]), plan.out(node.outputs[0])
    plan.steps.append(lambda: H.trinv(l, out=o))


defop("trinv", _trinv_emit, _trinv_vjp)


def triangular_solve(L, b, lower=True, adjoint=False) -> Tensor:
    """tf.matrix_triangular_solve(L, b): formed as L^{-1} b with the explicit
    triangular inverse (the reference's batched SparseGP branch does the same,
    gp/gp.py:169-172)."""
    if not lower:
        raise NotImplementedError("upper-triangular solve")
    return matmul(trinv(L), b, transpose_a=adjoint)


# ------------------------------------------------------------------------------
# fused probabilistic ops
# ------------------------------------------------------------------------------
def random_normal(shape, stream="local") -> Tensor:
    """tf.random_normal: fresh N(0,1) draw per run.  stream 'global' draws are
    identical on every data-parallel rank, 'local' ones are rank-distinct."""
    return Node("leaf:noise", (), {"stream": stream}, [tuple(int(s) for s in shape)]).outputs[0]


def diag_sample_kl(mu, s, u=None, stream="global") -> Tuple[Tensor, Tensor, Tensor]:
    """(x, kl, u):  x = mu + exp(s) u ;  kl = -0.5 sum(2 s + u^2 - x^2)
    (reference variationals.py:138-142, :225-230).  `u` = injected noise tensor
    or None (drawn in-kernel)."""
    mu, s = as_tensor(mu), as_tensor(s)
    assert mu.shape == s.shape
    ins = (mu, s) if u is None else (mu, s, as_tensor(u))
    n = make("diag_sample_kl", ins, {"stream": stream}, [mu.shape, (1,), mu.shape], unique=u is None)
    return n.outputs[0], n.outputs[1], n.outputs[2]


def _col_view(plan, t):
    """(1-D view starting at the operand's first element, row stride) for a lazily sliced column block, else (buffer, None)."""
    lz = plan._lazy_cols.get(t)
    if lz is None:
        return plan.buf(t), None
    src, (L, ld, off) = lz
    return src.reshape(-1)[off:], ld


def _diag_skl_emit(plan, node):
    H = plan.H
    (mu, ldm), (s, lds) = _col_view(plan, node.inputs[0]), _col_view(plan, node.inputs[1])
    u_in = plan.buf(node.inputs[2]) if len(node.inputs) > 2 else None
    outs = tuple(plan.out(t) for t in node.outputs)
    x, kl, u = outs
    rng = None if u_in is not None else plan.rng(node.attrs["stream"])
    rows = None
    if ldm is not None or lds is not None:
        nrows, L = node.inputs[0].shape
        rows = (nrows, L, ldm or L, lds or L)
    cell = plan.side_candidate(list(node.outputs))
    plan.steps.append(lambda: H.diag_sample_kl_fwd(mu, s, u_in=u_in, rng=rng, out=(x, kl, u), rows=rows, defer=cell["defer"]))


def _diag_skl_vjp(node, gs):
    gx, gkl = gs[0], gs[1]
    if gx is None and gkl is None:
        return [None] * len(node.inputs)
    x, kl, u = node.outputs
    s = node.inputs[1]
    ins = [s, u, x]
    flags = {"has_x": gx is not None, "has_kl": gkl is not None}
    if gx is not None:
        ins.append(gx)
    if gkl is not None:
        ins.append(gkl)
    n = make("diag_sample_kl_grad", tuple(ins), flags, [s.shape, s.shape])
    return [n.outputs[0], n.outputs[1]] + [None] * (len(node.inputs) - 2)


def _diag_skl_grad_emit(plan, node):
    H = plan.H
    s, lds = _col_view(plan, node.inputs[0])
    bufs = [None] + [plan.buf(t) for t in node.inputs[1:]]
    u, x = bufs[1:3]
    k = 3
    xbar = klbar = None
    if node.attrs["has_x"]:
        xbar = bufs[k]
        k += 1
    if node.attrs["has_kl"]:
        klbar = bufs[k]
    o0, o1 = node.outputs
    c0, c1 = plan._consumers.get(o0, []), plan._consumers.get(o1, [])
    packed = None
    if (len(c0) == 1 and len(c1) == 1 and c0[0] is c1[0] and c0[0].op == "concat" and len(o0.shape) == 2
            and c0[0].attrs["axis"] == 1 and len(c0[0].inputs) == 2 and set(c0[0].inputs) == {o0, o1}
            and not any(t in plan.outputs or t in plan._bind for t in (o0, o1))):
        # both gradients are the two halves of ONE [rows, 2L] matrix (the gradient of the encoder output): written
        # in place with the wide row stride, the concat that would assemble them is not emitted
        cat = c0[0]
        L = o0.shape[1]
        dst = plan.out(cat.outputs[0]).reshape(-1)
        first = cat.inputs[0] is o0
        packed = (dst[0:] if first else dst[L:], dst[L:] if first else dst[0:])
        plan._fused_concat.add(cat.id)
    if packed is not None or lds is not None:
        nrows, L = o0.shape if len(o0.shape) == 2 else (1, o0.size)
        outs = packed if packed is not None else tuple(plan.out(t) for t in node.outputs)
        rows = (nrows, L, lds or L, 2 * L if packed is not None else L)
    else:
        outs, rows = tuple(plan.out(t) for t in node.outputs), None
    cand_outs = list(node.outputs) + ([c0[0].outputs[0]] if packed is not None else [])
    cell = plan.side_candidate(cand_outs)
    plan.steps.append(lambda: H.diag_sample_kl_bwd(s, u, x, xbar, klbar, out=outs, rows=rows, defer=cell["defer"]))


defop("diag_sample_kl", _diag_skl_emit, _diag_skl_vjp)
defop("diag_sample_kl_grad", _diag_skl_grad_emit, None)


# ------------------------------------------------------------------------------
# the amortised encoder as one op: two MatBias layers feeding a LOCAL diagonal Normal (csrc/mlp.hip)
# ------------------------------------------------------------------------------
def mlp2_sample_kl(y, w0, b0, w1, b1, act, u=None, stream="local"):
    """(x, kl, u, o):  o = act(y w0 + b0) w1 + b1 = [mu | s];  x = mu + exp(s) u;  kl = -0.5 sum(2 s + u^2 - x^2)
    (reference nn.py:31-32,73-84 feeding variationals.py:121-129,138-142,225-230).  The hidden layer is an internal of the
    op: the fused kernels never write it, the backward recomputes it.  No gradient flows into `y` (a data operand)."""
    y, w0, b0, w1, b1 = (as_tensor(t) for t in (y, w0, b0, w1, b1))
    n, L2 = y.shape[0], w1.shape[-1]
    assert len(y.shape) == 2 and L2 % 2 == 0 and w0.shape[0] == y.shape[1] and w1.shape[0] == w0.shape[1]
    ins = (y, w0, b0, w1, b1) if u is None else (y, w0, b0, w1, b1, as_tensor(u))
    nd = make("mlp2_sample_kl", ins, {"act": act, "stream": stream}, [(n, L2 // 2), (1,), (n, L2 // 2), (n, L2)], unique=u is None)
    return nd.outputs[0], nd.outputs[1], nd.outputs[2], nd.outputs[3]


def _mlp2_dims(node):
    n, din = node.inputs[0].shape
    return n, din, node.inputs[1].shape[1], node.inputs[3].shape[1]


def _mlp2_emit(plan, node):
    H = plan.H
    y, w0, b0, w1, b1 = (plan.buf(t) for t in node.inputs[:5])
    u_in = plan.buf(node.inputs[5]) if len(node.inputs) > 5 else None
    x, kl, u, o = (plan.out(t) for t in node.outputs)
    act = node.attrs["act"]
    n, din, hid, L2 = _mlp2_dims(node)
    rng = None if u_in is not None else plan.rng(node.attrs["stream"])
    from ._settings import settings as _st

    fused = (plan.dtype == plan.torch.float32 and bool(getattr(_st.runtime, "fused_encoder", True))
             and H.mlp2_sample_supported(n, din, hid, L2, rng.nlanes if rng is not None else 0, u_in is not None))
    plan._mlp2[node.id] = {"fused": fused}
    plan.note("fused encoder (hb_mlp2_sample_fwd / _bwd)", node, fused,
              "" if fused else "needs fp32, Din in {32, 64}, H in {128, 256}, 2 L = 32 outputs, n % 32 == 0, 2 n generator "
              "lanes for in-kernel noise and settings.runtime.fused_encoder: lowered to the op-by-op launches")
    if fused:
        ws = plan.torch.empty(H.mlp2_sample_ws_elems(n, din, hid), dtype=plan.dtype, device=plan.device)
        plan._mlp2[node.id]["ws"] = ws
        plan.steps.append(lambda: H.mlp2_sample_fwd(y, w0, b0, w1, b1, act, u_in=u_in, rng=rng, out=(x, kl, u, o), ws=ws))
        return
    # lowered form (fp64, shapes the fused kernels do not take): the op-by-op launches, hidden layer in a scratch buffer
    h = plan.scratch((n, hid))
    plan._mlp2[node.id]["h"] = h
    L = L2 // 2
    of = o.reshape(-1)

    def step():
        H.matmul(y, w0, bias=b0, act=act, out=h)
        H.matmul(h, w1, bias=b1, out=o)
        H.diag_sample_kl_fwd(of[0:], of[L:], u_in=u_in, rng=rng, out=(x, kl, u), rows=(n, L, L2, L2))

    plan.steps.append(step)


def _mlp2_vjp(node, gs):
    gx, gkl = gs[0], gs[1]
    if gs[2] is not None or gs[3] is not None:
        raise NotImplementedError("gradients through the noise / encoder-output results of mlp2_sample_kl")
    if gx is None and gkl is None:
        return [None] * len(node.inputs)
    y, w0, b0, w1, b1 = node.inputs[:5]
    x, kl, u, o = node.outputs
    ins = [y, w0, b0, w1, o, u, x] + ([gx] if gx is not None else []) + ([gkl] if gkl is not None else [])
    g = make("mlp2_sample_kl_grad", tuple(ins), {"act": node.attrs["act"], "has_x": gx is not None, "has_kl": gkl is not None},
             [w0.shape, b0.shape, w1.shape, b1.shape])
    return [None, g.outputs[0], g.outputs[1], g.outputs[2], g.outputs[3]] + [None] * (len(node.inputs) - 5)


def _mlp2_grad_emit(plan, node):
    H = plan.H
    y, w0, b0, w1, o, u, x = (plan.buf(t) for t in node.inputs[:7])
    k = 7
    xbar = klbar = None
    if node.attrs["has_x"]:
        xbar = plan.buf(node.inputs[k])
        k += 1
    if node.attrs["has_kl"]:
        klbar = plan.buf(node.inputs[k])
    dw0, db0, dw1, db1 = (plan.out(t) for t in node.outputs)
    act = node.attrs["act"]
    fwd = node.inputs[4].node
    n, din, hid, L2 = _mlp2_dims(fwd)
    rec = plan._mlp2[fwd.id]
    if rec["fused"]:
        ws = rec["ws"]
        plan.steps.append(lambda: H.mlp2_sample_bwd(y, w0, b0, w1, act, o, u, x, xbar, klbar,
                                                    out=(dw0, db0.reshape(-1), dw1, db1.reshape(-1)), ws=ws))
        return
    h, L = rec["h"], L2 // 2
    do, dh = plan.scratch((n, L2)), plan.scratch((n, hid))
    of, dof = o.reshape(-1), do.reshape(-1)

    def step():
        H.diag_sample_kl_bwd(of[L:], u, x, xbar, klbar, out=(dof[0:], dof[L:]), rows=(n, L, L2, L2))
        H.matmul_colsum(h, do, out=dw1, colsum=db1.reshape(-1))
        H.matmul(do, w1, transB=True, act=act, actgrad=h, out=dh)
        H.matmul_colsum(y, dh, out=dw0, colsum=db0.reshape(-1))

    plan.steps.append(step)


defop("mlp2_sample_kl", _mlp2_emit, _mlp2_vjp)
defop("mlp2_sample_kl_grad", _mlp2_grad_emit, None)


def match_mlp2_encoder(mu, sq):
    """(y, w0, b0, w1, b1, act) when `mu` / `sq` are the two column halves of o = matmul(matmul(y, w0, b0, act), w1, b1) with
    `y` a data operand (no gradient path into it) -- the amortised-encoder pattern mlp2_sample_kl replaces -- else None."""
    def base(t):
        while t.node.op == "reshape" and tuple(t.node.inputs[0].shape) == tuple(t.shape):
            t = t.node.inputs[0]
        return t

    bm, bs = base(mu), base(sq)
    if bm.node.op != "strided" or bs.node.op != "strided" or len(mu.shape) != 2 or tuple(mu.shape) != tuple(sq.shape):
        return None
    o = bm.node.inputs[0]
    cm, cs = _column_block(bm.node), _column_block(bs.node)
    L = mu.shape[1]
    if bs.node.inputs[0] is not o or cm != (L, 2 * L, 0) or cs != (L, 2 * L, L) or tuple(bm.shape) != tuple(mu.shape):
        return None
    n2 = o.node
    if n2.op != "matmul" or len(n2.inputs) != 3 or n2.attrs["ta"] or n2.attrs["tb"] or n2.attrs["act"] != "none" or n2.attrs.get("actgrad"):
        return None
    h, w1, b1 = n2.inputs
    n1 = h.node
    if (n1.op != "matmul" or len(n1.inputs) != 3 or n1.attrs["ta"] or n1.attrs["tb"] or n1.attrs.get("actgrad")
            or n1.attrs["act"] not in ("sigmoid", "relu", "tanh")):
        return None
    y, w0, b0 = n1.inputs
    if len(y.shape) != 2 or len(w0.shape) != 2 or len(w1.shape) != 2 or y.node.op not in ("leaf:minibatch", "leaf:data", "leaf:const"):
        return None
    return y, w0, b0, w1, b1, n1.attrs["act"]


def fullrank_sample_kl(mu, S, u=None, stream="global", packed=False) -> Tuple[Tensor, Tensor, Tensor]:
    """(x, kl, u):  x = mu + tril(S) u ; kl = -0.5 sum(log S_kk^2 + u^2 - x^2)
    (reference variationals.py:144-146, :186, :225-230).  packed: S holds the lower triangle only,
    [..., size(size+1)/2] in tril_indices order (half the bytes read; the gradient comes back packed)."""
    mu, S = as_tensor(mu), as_tensor(S)
    size = mu.shape[-1]
    if packed:
        assert S.shape == mu.shape[:-1] + (size * (size + 1) // 2,)
    else:
        assert S.shape == mu.shape + (size,)
    ins = (mu, S) if u is None else (mu, S, as_tensor(u))
    n = make("fullrank_sample_kl", ins, {"stream": stream, "packed": bool(packed)}, [mu.shape, (1,), mu.shape],
             unique=u is None)
    return n.outputs[0], n.outputs[1], n.outputs[2]


def _fr_skl_emit(plan, node):
    H = plan.H
    mu, S = plan.buf(node.inputs[0]), plan.buf(node.inputs[1])
    u_in = plan.buf(node.inputs[2]) if len(node.inputs) > 2 else None
    outs = tuple(plan.out(t) for t in node.outputs)
    rng = None if u_in is not None else plan.rng(node.attrs["stream"])
    packed = bool(node.attrs.get("packed", False))
    plan.steps.append(lambda: H.fullrank_sample_kl_fwd(mu, S, u_in=u_in, rng=rng, out=outs, packed=packed))


def _fr_skl_vjp(node, gs):
    gx, gkl = gs[0], gs[1]
    if gx is None and gkl is None:
        return [None] * len(node.inputs)
    x, kl, u = node.outputs
    S = node.inputs[1]
    ins = [S, u, x]
    flags = {"has_x": gx is not None, "has_kl": gkl is not None, "packed": bool(node.attrs.get("packed", False))}
    if gx is not None:
        ins.append(gx)
    if gkl is not None:
        ins.append(gkl)
    n = make("fullrank_sample_kl_grad", tuple(ins), flags, [x.shape, S.shape])
    return [n.outputs[0], n.outputs[1]] + [None] * (len(node.inputs) - 2)


def _fr_skl_grad_emit(plan, node):
    H = plan.H
    bufs = [plan.buf(t) for t in node.inputs]
    S, u, x = bufs[:3]
    k = 3
    xbar = klbar = None
    if node.attrs["has_x"]:
        xbar = bufs[k]
        k += 1
    if node.attrs["has_kl"]:
        klbar = bufs[k]
    outs = tuple(plan.out(t) for t in node.outputs)
    packed = bool(node.attrs.get("packed", False))
    plan.steps.append(lambda: H.fullrank_sample_kl_bwd(S, u, x, xbar, klbar, out=outs, packed=packed))


defop("fullrank_sample_kl", _fr_skl_emit, _fr_skl_vjp)
defop("fullrank_sample_kl_grad", _fr_skl_grad_emit, None)


def vec_to_tri(v) -> Tensor:
    """[..., N(N+1)/2] -> lower-triangular [..., N, N]: the reference's disabled native op (tf_wraps.py:50-71);
    its gradient is tri_to_vec (tf_wraps.py:56-58)."""
    v = as_tensor(v)
    T = v.shape[-1]
    N = int((8 * T + 1) ** 0.5 / 2.0 - 0.5 + 1e-9)
    if N * (N + 1) // 2 != T:
        raise ValueError("vec_to_tri: %d is not a triangular number" % T)
    return make("vec_to_tri", (v,), {}, [v.shape[:-1] + (N, N)]).outputs[0]


def tri_to_vec(t) -> Tensor:
    t = as_tensor(t)
    N = t.shape[-1]
    assert t.shape[-2] == N
    return make("tri_to_vec", (t,), {}, [t.shape[:-2] + (N * (N + 1) // 2,)]).outputs[0]


defop("vec_to_tri", lambda plan, node: (lambda a, o: plan.steps.append(lambda: plan.H.vec_to_tri(a, out=o)))(
    plan.buf(node.inputs[0]), plan.out(node.outputs[0])), lambda node, gs: [None if gs[0] is None else tri_to_vec(gs[0])])
defop("tri_to_vec", lambda plan, node: (lambda a, o: plan.steps.append(lambda: plan.H.tri_to_vec(a, out=o)))(
    plan.buf(node.inputs[0]), plan.out(node.outputs[0])), lambda node, gs: [None if gs[0] is None else vec_to_tri(gs[0])])

KERN_KINDS = {"rbf": 0, "csym_rbf": 1, "sqdist": 2}


def gram(X, X2, ell, kind="rbf") -> Tensor:
    """Stationary Gram matrix K(X, X2) (reference gp/kernels.py:54-131); X, X2
    are [n,d] or [B,n,d] (a 2-D operand is shared by the batch); ell is [dl]
    (one kernel) or [B,dl] (an independent kernel per batch entry: experts)."""
    X, X2, ell = as_tensor(X), as_tensor(X2), as_tensor(ell)
    if X.shape[-1] != X2.shape[-1]:
        raise ValueError("gram: input dimensions differ")
    lead = X.shape[:-2] if len(X.shape) >= len(X2.shape) else X2.shape[:-2]
    if len(ell.shape) == 2 and ell.shape[0] > 1:
        if lead and int(np.prod(lead)) != ell.shape[0]:
            raise ValueError("gram: %d lengthscale sets for a batch of %s" % (ell.shape[0], lead))
        lead = lead or (ell.shape[0],)
    return make("gram", (X, X2, ell), {"kind": kind}, [tuple(lead) + (X.shape[-2], X2.shape[-2])]).outputs[0]


def _gram_layout(tX, tX2, tell):
    n, d = tX.shape[-2], tX.shape[-1]
    n2 = tX2.shape[-2]
    BX = int(np.prod(tX.shape[:-2])) if len(tX.shape) > 2 else 1
    BX2 = int(np.prod(tX2.shape[:-2])) if len(tX2.shape) > 2 else 1
    BE = tell.shape[0] if (len(tell.shape) == 2 and tell.shape[0] > 1) else 1
    B = max(BX, BX2, BE)
    sX = n * d if (BX == B and B > 1) else 0
    sX2 = n2 * d if (BX2 == B and B > 1) else 0
    dl = tell.size // BE
    sEll = dl if BE > 1 else 0
    return B, BX, BX2, n, n2, d, sX, sX2, sEll, dl


def _gram_emit(plan, node):
    H = plan.H
    X, X2, ell = (plan.buf(t) for t in node.inputs)
    k = KERN_KINDS[node.attrs["kind"]]
    y = node.outputs[0]
    # K + jitter*I (kern.Cholesky, reference gp/kernels.py:101) as the Gram kernel's own diagonal term
    cons = plan._consumers.get(y, [])
    if (len(cons) == 1 and cons[0].op == "matutil" and cons[0].attrs["mode"] == 1 and y.shape[-1] == y.shape[-2]
            and y not in plan.outputs and y not in plan._bind):
        jit = cons[0].attrs["alpha"]
        # K(z, z) + jitter I whose ONLY reader is a factor + inverse in persistent form (fp32, M % 64 == 0): the Cholesky
        # launch synthesises its tiles from the points (hb_gram_cholesky_inverse), K is never written, this launch goes
        kj = cons[0].outputs[0]
        cc = plan._consumers.get(kj, [])
        M = y.shape[-1]
        Bk = int(np.prod(y.shape[:-2])) if len(y.shape) > 2 else 1
        from ._settings import settings as _st

        if (bool(getattr(_st.runtime, "fold_gram", True)) and node.inputs[0] is node.inputs[1] and len(cc) == 1
                and cc[0].op == "cholesky" and kj not in plan.outputs
                and kj not in plan._bind and plan.dtype == plan.torch.float32 and k in (H.KERN_RBF, H.KERN_CSYM_RBF)
                and any(c.op == "trinv" for c in plan._consumers.get(cc[0].outputs[0], ()))
                and H.cholesky_persistent_shape(Bk, M, plan.dtype)):
            plan._fused_matutil.add(cons[0].id)
            plan._gram_for_chol[cc[0].id] = (X, ell, k, jit)
            plan.note("Gram folded into the persistent Cholesky (hb_gram_cholesky_inverse)", node, True)
            return
        plan.note("Gram folded into the persistent Cholesky (hb_gram_cholesky_inverse)", node, False,
                  "needs fp32, K(z, z) with one reader (a factor + inverse), an RBF-family kernel, M % 64 == 0 and "
                  "settings.runtime.fold_gram")
        out = plan.out(cons[0].outputs[0])
        plan._fused_matutil.add(cons[0].id)
        plan.steps.append(lambda: H.gram_fwd(X, X2, ell, kind=k, out=out, diag_add=jit))
        return
    out = plan.out(y)
    plan.steps.append(lambda: H.gram_fwd(X, X2, ell, kind=k, out=out))


def _gram_vjp(node, gs):
    g = gs[0]
    if g is None:
        return [None, None, None]
    X, X2, ell = node.inputs
    if X is X2:
        # K(z, z): one pass gives the total gradient w.r.t. the shared points
        n = make("gram_grad", (X, X2, ell, g), {"kind": node.attrs["kind"], "sym": True}, [X.shape, ell.shape])
        return [n.outputs[0], None, n.outputs[1]]
    n = make("gram_grad", (X, X2, ell, g), {"kind": node.attrs["kind"], "sym": False}, [X.shape, X2.shape, ell.shape])
    return list(n.outputs)


def _gram_grad_emit(plan, node):
    H = plan.H
    fused = plan._gram_in_mm.get(node.id)
    if fused is not None:
        # the point gradient and the lengthscale row partials came out of the product that computed Kbar (_matmul_emit)
        lws, rows, d_, dl_, groups = fused
        oL = plan.out(node.outputs[1])
        plan.out(node.outputs[0])
        fold = lambda: H.gram_ell_fold(lws, rows, d_, dl_, groups, oL)
        plan.steps.append(fold)
        plan.chain_kind[id(fold)] = "full"
        return
    tX, tX2, tell, tg = node.inputs
    X, X2, ell, g = (plan.buf(t) for t in node.inputs)
    k = KERN_KINDS[node.attrs["kind"]]
    B, BX, BX2, n, n2, d, sX, sX2, sEll, dl = _gram_layout(tX, tX2, tell)
    if node.attrs.get("sym"):
        oX, oL = [plan.out(t) for t in node.outputs]
        tmp = plan.scratch((B, n, d)) if (BX == 1 and B > 1) else None
        ws = plan.scratch((max(B * n * d, 1),))

        ksym = k | (H.KERN_KBAR_SYMMETRIC if tg.node.attrs.get("sym_result") else 0)

        def sym_step():
            dst = tmp if tmp is not None else oX
            H.gram_bwd_raw(ksym, X, sX, X2, sX2, ell, sEll, dl, g, dst, dst, oL, B, n, n2, d, ws)
            if tmp is not None:
                H.reduce_mid(tmp, 1, B, n * d, out=oX)

        plan.steps.append(sym_step)
        if tmp is None:
            plan.chain_kind[id(sym_step)] = "tail"   # its last launch (the lengthscale fold) may open a serial chain
        return
    oX, oX2, oL = [plan.out(t) for t in node.outputs]
    # an operand shared by the batch gets its per-batch gradients summed
    tmpX = plan.scratch((B, n, d)) if (BX == 1 and B > 1) else None
    tmpX2 = plan.scratch((B, n2, d)) if (BX2 == 1 and B > 1) else None
    ws = plan.scratch((max(B * n * d, 1),))

    def step():
        H.gram_bwd_raw(k, X, sX, X2, sX2, ell, sEll, dl, g, tmpX if tmpX is not None else oX,
                       tmpX2 if tmpX2 is not None else oX2, oL, B, n, n2, d, ws)
        if tmpX is not None:
            H.reduce_mid(tmpX, 1, B, n * d, out=oX)
        if tmpX2 is not None:
            H.reduce_mid(tmpX2, 1, B, n2 * d, out=oX2)

    plan.steps.append(step)
    if tmpX is None and tmpX2 is None:
        plan.chain_kind[id(step)] = "tail"


defop("gram", _gram_emit, _gram_vjp)
defop("gram_grad", _gram_grad_emit, None)

SGP_MODES = {"neglected": 0, "diagonal": 1}


def sgp_samples(x, z, ell, L, u, mode="diagonal", eps=None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Fused SparseGP.samples (reference gp/gp.py:99-143) for the UnitRBF kernel
    and a 2-D x:  A = L^-1 K(z,x); f = u A + sqrt|1 - colsum(A^2)| eps.
    Returns (f [P,n], A [M,n], v [n], eps [n]).  L = chol(K(z,z)+jitter I)."""
    x, z, ell, L, u = (as_tensor(t) for t in (x, z, ell, L, u))
    W = stop_gradient(trinv(L))  # helper operand: the VJP differentiates through L directly
    M, n, P = z.shape[-2], x.shape[-2], u.shape[-2]
    lead = z.shape[:-2]
    ins = [x, z, ell, L, W, u] + ([as_tensor(eps)] if eps is not None else [])
    nd = make("sgp", tuple(ins), {"mode": mode}, [lead + (P, n), lead + (M, n), lead + (n,), lead + (n,)],
              unique=(eps is None and mode == "diagonal"))
    return tuple(nd.outputs)


def _through_stop_gradient(t):
    while t.node.op == "stop_gradient":
        t = t.node.inputs[0]
    return t


def _sgp_emit(plan, node):
    H = plan.H
    x, z, ell, L, W, u = (plan.buf(t) for t in node.inputs[:6])
    eps_in = plan.buf(node.inputs[6]) if len(node.inputs) > 6 else None
    outs = tuple(plan.out(t) for t in node.outputs)
    mode = SGP_MODES[node.attrs["mode"]]
    rng = plan.rng("local") if (eps_in is None and mode == 1) else None
    wfrag, bf3 = plan._wfrag.get(_through_stop_gradient(node.inputs[4]), (None, False))
    # bf16x3 operands (settings.numerics.contraction): column-strip kernel only (d <= 4, at most 4 latent functions)
    bf3 = bf3 and node.inputs[0].shape[-1] <= 4 and node.inputs[5].shape[-2] <= 4
    prec = H.PREC_BF16X3 if bf3 else H.PREC_NATIVE
    # A = L^-1 K(z,x) is only an intermediate between this op and its VJP: when both run in column-strip form it is
    # exchanged in fragment-major layout (every load and store of the three M^2 n contractions a contiguous KB),
    # and the row-major copy is not written at all
    zsh, xsh = node.inputs[1].shape, node.inputs[0].shape
    E, M, n, d, P = int(np.prod(zsh[:-2])) if len(zsh) > 2 else 1, zsh[-2], xsh[-2], xsh[-1], node.inputs[5].shape[-2]
    a_frag, skip_a = None, False
    if wfrag is not None and _sgp_frag_exchange(plan, node, prec):
        a_frag = plan.scratch((H.sgp_frag_elems(E, n, M, prec),))
        plan._afrag[node.outputs[1]] = (a_frag, prec)
        skip_a = node.outputs[1] not in plan.outputs
    head = None
    g = plan._sgp_head.get(node.id)
    if g is not None and wfrag is not None:
        units = H.sgp_head_units(x, z, u, prec, True, eps_in is None and mode == 1, rng)
        if units > 0:
            post = plan._gll_post.get(g.id)
            def early(t):
                # (an operand may be a view whose reshape node sits later in the emission order: take it from its source)
                shp = t.shape
                while t.node.op == "reshape" and t not in plan._buf:
                    t = t.node.inputs[0]
                return plan.buf(t).view(shp)

            head = dict(y=early(g.inputs[0]), var=early(g.inputs[2]),
                        scale=early(g.inputs[3]) if len(g.inputs) > 3 else None,
                        dmu=plan.out(g.outputs[1]), post=post[0] if post else 0.0,
                        fbar=plan.out(post[1]) if post else None,
                        part=plan.scratch((3 * units,)), units=units)
            plan._gll_fused[g.id] = (head["part"], units)
            # the head makes THIS step read y, var and scale, which are not inputs of the sgp node: a pending side job
            # that produces one of them (the hoisted minibatch gather of Y) must not ride on a launch behind this one
            plan.pin_side_reads([g.inputs[0], g.inputs[2]] + ([g.inputs[3]] if len(g.inputs) > 3 else []))
    step = lambda: H.sgp_fwd(x, z, ell, W, u, eps_in=eps_in, rng=rng, mode=mode, out=outs, wfrag=wfrag,
                             prec=prec, a_frag=a_frag, skip_a=skip_a, head=head)
    # Early-start form: the contraction starts inside the launch of the persistent factorisation that produces W, when
    # everything else it reads exists before that launch (or is written by a side job riding on it)
    ep = "forward contraction starts inside the persistent Cholesky's launch (hb_sgp_rider_begin)"
    cell = plan._chol_rider.get(_through_stop_gradient(node.inputs[4]))
    from ._settings import settings as _st
    if not bool(getattr(_st.runtime, "early_forward", False)):
        cell = None
    if cell is not None and cell["rider"] is None:
        if not H.sgp_rider_supported(x, z, u, prec, wfrag is not None, eps_in is None and mode == 1, rng):
            plan.note(ep, node, False, "needs the fused finishing pass (fp32, one latent function, d <= 2, M % 64 == 0)")
        else:
            reads = [node.inputs[0], node.inputs[1], node.inputs[2], node.inputs[5]] + list(node.inputs[6:])
            if head is not None:
                reads += [g.inputs[0], g.inputs[2]] + list(g.inputs[3:4])
            early_nodes = {id(nd) for nd in plan._emitted[:cell["epos"]]}

            def ready(t):
                while t.node.op in ("reshape", "stop_gradient") and t.node.inputs:
                    t = t.node.inputs[0]
                return t.node.op.startswith("leaf:") or id(t.node) in early_nodes

            if all(ready(t) for t in reads):
                cell["rider"] = step
                plan.step_labels[id(cell["step"])] = "cholesky+sgp"
                plan.step_riders[id(cell["step"])] = node
                plan.note(ep, node, True)
                return
            plan.note(ep, node, False, "an operand is produced after the factorisation has been launched")
    plan.steps.append(step)
    plan.chain_kind[id(step)] = "tail"       # its finishing pass may open a serial chain


def _sgp_vjp(node, gs):
    gf = gs[0]
    if gf is None:
        return [None] * len(node.inputs)
    if any(g is not None for g in gs[1:3]):
        raise NotImplementedError("gradients through the A / v outputs of sgp_samples")
    x, z, ell, L, W, u = node.inputs[:6]
    f, A, v, eps = node.outputs
    n = make("sgp_grad", (x, z, ell, W, u, eps, A, v, gf), {"mode": node.attrs["mode"]},
             [L.shape, u.shape, z.shape, ell.shape, x.shape])
    Lb, ub, zb, lb, xb = n.outputs
    return [xb, zb, lb, Lb, None, ub] + [None] * (len(node.inputs) - 6)


def _sgp_grad_emit(plan, node):
    H = plan.H
    x, z, ell, W, u, eps, A, v, gf = (plan.buf(t) for t in node.inputs)
    Lb, ub, zb, lb, xb = (plan.out(t) for t in node.outputs)
    mode = SGP_MODES[node.attrs["mode"]]
    need_x = plan.needed(node.outputs[4])
    if need_x and len(node.inputs[1].shape) > 2 and len(node.inputs[0].shape) == 2:
        raise NotImplementedError("gradient w.r.t. an x shared by several experts")
    xbuf = xb.reshape((-1,) + tuple(xb.shape[-2:])) if need_x else None
    wfrag, bf3 = plan._wfrag.get(_through_stop_gradient(node.inputs[3]), (None, False))
    bf3 = bf3 and not need_x and node.inputs[0].shape[-1] <= 4 and node.inputs[4].shape[-2] <= 4
    prec = H.PREC_BF16X3 if bf3 else H.PREC_NATIVE
    a_frag, fprec = plan._afrag.get(node.inputs[6], (None, None))   # fragment-major A from the forward op (see _sgp_emit)
    if a_frag is not None:
        prec = fprec          # the layout (fp32 image / bf16x3 planes) was fixed by the forward op
    kbar_frag = plan.scratch((a_frag.numel(),)) if a_frag is not None else None
    Kbar = plan.scratch(A.shape) if a_frag is None else None
    plan.steps.append(lambda: H.sgp_bwd(x, z, ell, W, u, eps, A, v, gf, mode=mode, need_xbar=need_x,
                                        out=(Kbar, Lb, ub, zb, lb, xbuf), wfrag=wfrag, prec=prec, a_frag=a_frag,
                                        kbar_frag=kbar_frag))


defop("sgp", _sgp_emit, _sgp_vjp)
defop("sgp_grad", _sgp_grad_emit, None)


# ------------------------------------------------------------------------------
# autodiff
# ------------------------------------------------------------------------------
def topo_order(outputs: Sequence[Tensor]) -> List[Node]:
    seen, order = set(), []
    stack = [(t.node, False) for t in outputs]
    while stack:
        n, done = stack.pop()
        if done:
            order.append(n)
            continue
        if n.id in seen:
            continue
        seen.add(n.id)
        stack.append((n, True))
        for t in n.inputs:
            if t.node.id not in seen:
                stack.append((t.node, False))
    return order


def _scatter_partition(ts):
    """The gradients of slices along ONE axis that tile their source exactly -- an encoder output split into mean and
    log-std halves (last axis; reference variationals.py:70-80), the experts / gates halves of a batched GP draw
    (leading axis; notebooks/Expert_GPR.ipynb:139-147): their sum is the concatenation, not N zero-filled buffers added up."""
    if len(ts) < 2 or any(t.node.op != "scatter_strided" for t in ts):
        return None
    xshape = tuple(ts[0].node.attrs["xshape"])
    if len(xshape) < 1:
        return None
    cst = tuple(_contig_strides(xshape))
    parts, axis = [], None
    for t in ts:
        a = t.node.attrs
        if tuple(a["xshape"]) != xshape or len(a["shape"]) != len(xshape) or tuple(a["strides"]) != cst:
            return None
        diff = [i for i in range(len(xshape)) if a["shape"][i] != xshape[i]]
        if len(diff) != 1 or (axis is not None and diff[0] != axis):
            return None
        axis = diff[0]
        if a["offset"] % cst[axis] != 0:
            return None
        start = a["offset"] // cst[axis]
        if not 0 <= start or start + a["shape"][axis] > xshape[axis]:
            return None
        parts.append((start, a["shape"][axis], t.node.inputs[0]))
    parts.sort(key=lambda p: p[0])
    pos = 0
    for off, w, _ in parts:
        if off != pos:
            return None
        pos += w
    if pos != xshape[axis]:
        return None
    return concat([p[2] for p in parts], axis)


def add_n(ts: List[Tensor]) -> Tensor:
    packed = _scatter_partition(ts)
    if packed is not None:
        return packed
    acc = ts[0]
    for t in ts[1:]:
        acc = binary("ADD", acc, t)
    return acc


def gradients(loss: Tensor, wrt: Sequence[Tensor]) -> List[Optional[Tensor]]:
    """d loss / d wrt[i] as new graph tensors (None where independent).

    Stands in for TF's autodiff inside optimizer.minimize (reference model.py:220)."""
    if loss.size != 1:
        raise ValueError("gradients() needs a scalar objective, got shape %s" % (loss.shape,))
    order = topo_order([loss])
    # nodes that depend on any wrt tensor
    wrt_set = set(wrt)
    dep = set()
    for n in order:
        if any(o in wrt_set for o in n.outputs) or any(t.node.id in dep for t in n.inputs):
            dep.add(n.id)
    grads: Dict[Tensor, List[Tensor]] = {loss: [constant(np.ones(loss.shape))]}
    for n in reversed(order):
        if n.id not in dep or n.op.startswith("leaf:"):
            continue
        outg = []
        anyg = False
        for o in n.outputs:
            lst = grads.get(o)
            if lst:
                g = add_n(lst) if len(lst) > 1 else lst[0]
                grads[o] = [g]
                outg.append(g)
                anyg = True
            else:
                outg.append(None)
        if not anyg:
            continue
        d = OPS.get(n.op)
        if d is None or d.vjp is None:
            raise NotImplementedError("op %s has no gradient" % n.op)
        ing = d.vjp(n, outg)
        for t, g in zip(n.inputs, ing):
            if g is None or t.node.id not in dep and t not in wrt_set:
                continue
            if g.shape != t.shape:
                g = sum_to_shape(g, t.shape)
            grads.setdefault(t, []).append(g)
    res = []
    for w in wrt:
        lst = grads.get(w)
        if not lst:
            res.append(None)
        else:
            res.append(add_n(lst) if len(lst) > 1 else lst[0])
    return res


# ------------------------------------------------------------------------------
# HIP executor
# ------------------------------------------------------------------------------
# ------------------------------------------------------------------------------
# elementwise clustering: chains / groups of elementwise nodes -> one interpreted launch
# ------------------------------------------------------------------------------
EW_CLUSTER_MAX_ELEMS = 1 << 16   # interpreted programs: larger tensors keep one specialised launch per op (the LDS
                                 # register file caps the interpreter at 256 threads x 40 registers per workgroup)
EW_CLUSTER_MAX_ELEMS_JIT = 1 << 26   # compiled programs run from registers at full occupancy: no such limit
EW_CLUSTER_MAX_INSTR = 44
EW_CLUSTER_MAX_IN = 12
EW_CLUSTER_MAX_OUT = 6
EW_CLUSTER_MAX_REGS = 38


def _squeeze_shape(shape):
    return tuple(d for d in shape if d != 1)


def _merge_space(S, shape):
    """Iteration space containing both (one must broadcast to the other), or None."""
    try:
        b = tuple(np.broadcast_shapes(S, shape))
    except ValueError:
        return None
    size = lambda t: int(np.prod(t)) if t else 1
    if size(b) != max(size(S), size(shape)):
        return None
    return b


def _fusable(n):
    if n.op == "ew":
        return True
    return n.op == "reshape" and n.outputs[0].size == 1


# A chain ending in reduce_sum runs as ONE workgroup (no cross-workgroup hand-off), i.e. one dependent global-load
# round trip per 256 elements: measured 45-90 us for 8192 elements against 4.6 us for a separate reduce launch,
# so only spaces a single pass covers are fused.
EW_REDUCE_MAX_ELEMS = 512


def _full_sum(n):
    """reduce_sum over every element (one result)."""
    return n.op == "reduce" and n.attrs["kind"] == "sum" and n.attrs["K1"] == 1 and n.attrs["K2"] == 1


class _Cluster:
    __slots__ = ("nodes", "space", "sealed", "ninstr", "reduced")

    def __init__(self):
        self.nodes, self.space, self.sealed, self.ninstr, self.reduced = [], (), False, 0, set()


def cluster_elementwise(order, enabled=True, max_elems=EW_CLUSTER_MAX_ELEMS, skip=()):
    """Greedy clustering over a topological order.  A node joins the cluster of one of its
    elementwise producers (or, failing that, any open cluster with a compatible iteration
    space); a cluster is sealed as soon as a non-member consumes one of its values, so the
    cluster can be emitted at its last member's position.  Returns {node id: cluster}."""
    member = {}
    if not enabled:
        return member
    open_clusters = []
    for n in order:
        joined = None
        if n.id not in skip and _fusable(n) and len(_squeeze_shape(n.outputs[0].shape)) <= 4 and n.outputs[0].size <= max_elems:
            cands = []
            for t in n.inputs:
                c = member.get(t.node.id)
                if c is not None and not c.sealed and c not in cands:
                    cands.append(c)
            cands += [c for c in reversed(open_clusters) if not c.sealed and c not in cands]
            for c in cands:
                if any(t in c.reduced for t in n.inputs):
                    continue  # a reduced value is only complete when its whole program has run
                sp = _merge_space(c.space, n.outputs[0].shape)
                if sp is None or len(_squeeze_shape(sp)) > 4 or int(np.prod(sp) if sp else 1) > max_elems:
                    continue
                if c.ninstr + 1 > EW_CLUSTER_MAX_INSTR:
                    continue
                # operand / result counts are re-checked when the program is built; keep a margin here
                ext_in = {t for m in c.nodes + [n] for t in m.inputs if member.get(t.node.id) is not c and t.node is not n}
                if len(ext_in) > EW_CLUSTER_MAX_IN or len(c.nodes) + 1 > EW_CLUSTER_MAX_OUT * 4:
                    continue
                joined = c
                c.space = sp
                break
            if joined is None:
                joined = _Cluster()
                joined.space = n.outputs[0].shape
                open_clusters.append(joined)
            joined.nodes.append(n)
            joined.ninstr += 1
            member[n.id] = joined
        elif n.id not in skip and _full_sum(n):
            # reduce_sum of a cluster value over the cluster's whole space: becomes a sum-reduced program output
            t = n.inputs[0]
            c = member.get(t.node.id)
            size = lambda s: int(np.prod(s)) if s else 1
            if (c is not None and not c.sealed and t not in c.reduced and t.size == size(c.space)
                    and size(c.space) <= EW_REDUCE_MAX_ELEMS and len(c.reduced) < 3):
                joined = c
                c.nodes.append(n)
                c.reduced.add(n.outputs[0])
                member[n.id] = c
        # whoever consumes a cluster value from outside closes that cluster
        for t in n.inputs:
            c = member.get(t.node.id)
            if c is not None and c is not joined:
                c.sealed = True
    return member


# ------------------------------------------------------------------------------
# column clusters: fused programs over a short-and-wide space [R, n] with reductions over the ROW axis inside
# (hb_ewise_colprog_build: one thread per column, row loops unrolled in the thread).  The softmax gate of the expert
# mixture (notebooks/Expert_GPR.ipynb:139-147) and its VJP are chains of elementwise ops on [E, n] arrays broken up by
# tf.reduce_max / tf.reduce_sum over axis 0 and by row-block slices of the batched GP draw: op by op that is ~20 launches
# of 4-60 us on [4, 65536] arrays at cfg 5; as column programs it is one launch forward and one backward.
# ------------------------------------------------------------------------------
EW_COL_MAX_ROWS = 8
EW_COL_MIN_COLS = 512
EW_COL_MAX_NODES = 44
EW_COL_MAX_IN = 12
EW_COL_MAX_OUT = 12


def _colform(shape, R, n):
    """How a tensor of `shape` sits in the column space [R, n]: 'full' ([R, n] up to unit dims), 'row' ([1, n]),
    'scalar' (one element), or None."""
    sq = tuple(int(d) for d in shape if d != 1)
    if sq == (R, n):
        return "full"
    if sq == (n,):
        return "row"
    if sq == ():
        return "scalar"
    return None


def _col_space_of(shape):
    sq = tuple(int(d) for d in shape if d != 1)
    if len(sq) == 2 and 2 <= sq[0] <= EW_COL_MAX_ROWS and sq[1] >= EW_COL_MIN_COLS:
        return sq
    return None


def _col_strided_view(node, R, n):
    """(form, element offset, row stride) when a strided node reads a row block [R, n] (or one row) of its source."""
    a = node.attrs
    nz = [(int(d), int(st)) for d, st in zip(a["shape"], a["strides"]) if d != 1]
    if len(nz) == 2 and nz[0][0] == R and nz[1] == (n, 1) and nz[0][1] >= n:
        return ("full", int(a["offset"]), nz[0][1])
    if len(nz) == 1 and nz[0] == (n, 1):
        return ("row", int(a["offset"]), 0)
    return None


class _ColCluster:
    __slots__ = ("nodes", "R", "n", "sealed", "absorbed")

    def __init__(self, R, n):
        self.nodes, self.R, self.n, self.sealed, self.absorbed = [], R, n, False, {}


def cluster_columns(order, outputs=(), enabled=True, skip=()):
    """Greedy clustering over a topological order, like cluster_elementwise, of the nodes a column program can hold:
    elementwise ops whose operands are [R, n], [1, n] or single elements, reductions over the row axis, reshapes /
    broadcasts between those forms, and row-block slices of a taller source (read in place).  Only clusters that contain
    a row reduction or a slice are kept -- the rest is plain elementwise work for cluster_elementwise.
    Returns {node id: cluster}."""
    member = {}
    if not enabled:
        return member
    consumers = {}
    for nd in order:
        for t in nd.inputs:
            consumers.setdefault(t, []).append(nd)
    outputs = set(outputs)
    open_clusters = []

    def can_join(nd, c):
        R, n = c.R, c.n
        if len(c.nodes) >= EW_COL_MAX_NODES:
            return False
        fo = _colform(nd.outputs[0].shape, R, n)
        if nd.op == "ew":
            if nd.attrs["f"] == "GAUSS_LOGPDF_GRAD" or len(nd.outputs) != 1 or len(nd.inputs) > 3 or fo is None:
                return False
            forms = [_colform(t.shape, R, n) for t in nd.inputs]
            if any(f is None for f in forms):
                return False
            # the broadcast of the operand forms must be the output's form
            best = "full" if "full" in forms else ("row" if "row" in forms else "scalar")
            return best == fo
        if nd.op == "reduce":
            a = nd.attrs
            return (a["kind"] in ("sum", "max") and a["K1"] == 1 and a["R"] == R and a["K2"] == n
                    and _colform(nd.inputs[0].shape, R, n) == "full")
        if nd.op == "reshape":
            return member.get(nd.inputs[0].node.id) is c and fo is not None and fo == _colform(nd.inputs[0].shape, R, n)
        if nd.op == "bcast":
            return member.get(nd.inputs[0].node.id) is c and fo == "full" and _colform(nd.inputs[0].shape, R, n) in ("row", "scalar")
        if nd.op == "strided":
            return member.get(nd.inputs[0].node.id) is not c and _col_strided_view(nd, R, n) is not None
        return False

    def seed_space(nd):
        if nd.op == "ew" and nd.attrs["f"] != "GAUSS_LOGPDF_GRAD" and len(nd.outputs) == 1:
            return _col_space_of(nd.outputs[0].shape)
        if nd.op == "strided":
            sp = _col_space_of(nd.outputs[0].shape)
            return sp if sp and _col_strided_view(nd, sp[0], sp[1]) is not None else None
        if nd.op == "reduce" and nd.attrs["K1"] == 1 and nd.attrs["kind"] in ("sum", "max"):
            sp = _col_space_of(nd.inputs[0].shape)
            return sp if sp and sp == (nd.attrs["R"], nd.attrs["K2"]) else None
        return None

    for nd in order:
        joined = None
        if nd.id not in skip and nd.op in ("ew", "reduce", "reshape", "bcast", "strided"):
            cands = []
            for t in nd.inputs:
                c = member.get(t.node.id)
                if c is not None and not c.sealed and c not in cands:
                    cands.append(c)
            cands += [c for c in reversed(open_clusters) if not c.sealed and c not in cands]
            for c in cands:
                if can_join(nd, c):
                    joined = c
                    break
            if joined is None:
                sp = seed_space(nd)
                if sp is not None:
                    joined = _ColCluster(sp[0], sp[1])
                    open_clusters.append(joined)
                    if not can_join(nd, joined):
                        open_clusters.pop()
                        joined = None
            if joined is not None:
                joined.nodes.append(nd)
                member[nd.id] = joined
        for t in nd.inputs:
            c = member.get(t.node.id)
            if c is not None and c is not joined:
                c.sealed = True
    # post-pass: slices whose value somebody outside needs are ordinary copies again; clusters without a row reduction
    # or an in-place slice, or with too many operands, dissolve
    for c in open_clusters:
        changed = True
        while changed:
            changed = False
            for nd in list(c.nodes):
                if nd.op != "strided":
                    continue
                o = nd.outputs[0]
                if o in outputs or any(member.get(x.id) is not c for x in consumers.get(o, [])) or not consumers.get(o):
                    c.nodes.remove(nd)
                    del member[nd.id]
                    changed = True
        ids = {nd.id for nd in c.nodes}
        ext_in = {t for nd in c.nodes for t in nd.inputs if t.node.id not in ids}
        outs = [o for nd in c.nodes for o in nd.outputs if nd.op != "strided"
                and (o in outputs or any(x.id not in ids for x in consumers.get(o, [])))]
        keep = (any(nd.op in ("reduce", "strided") for nd in c.nodes) and sum(1 for nd in c.nodes if nd.op in ("ew", "reduce")) >= 2
                and len(ext_in) <= EW_COL_MAX_IN and 1 <= len(outs) <= EW_COL_MAX_OUT)
        if not keep:
            for nd in c.nodes:
                del member[nd.id]
            c.nodes = []
    return member


INFO_POOL_WORDS = 65536


class CholeskyError(ArithmeticError):
    """A Cholesky factorisation met a non-positive pivot (TensorFlow raises
    InvalidArgumentError from tf.cholesky at the same point)."""


class Plan:
    """A compiled launch sequence for a set of output tensors.

    `leaf_resolver(tensor) -> torch.Tensor` supplies device buffers for param /
    data / minibatch leaves; `rngs` maps stream name -> hip_ops.Rng; `binds`
    lets the caller alias chosen outputs (e.g. leaf gradients) onto its own
    flat buffers so no copy is needed."""

    def __init__(self, outputs, dtype, device, leaf_resolver, rngs, binds=None, prologue=None, stream=None):
        from . import hip_ops

        import torch

        self.H = hip_ops
        self.torch = torch
        self.dtype, self.device = dtype, device
        self.outputs = list(outputs)
        self.steps: List = []
        self._buf: Dict[Tensor, object] = {}
        self._bind: Dict[Tensor, object] = {}
        self._rngs = rngs
        self._infos: List[Tuple[object, str]] = []
        self._nocap: List[str] = []
        self._graph = None
        self._graphs: Dict[object, object] = {}   # injection state -> captured graph (see _state_key)
        self._capture_enabled = False
        self._info_pool = None          # every factorisation status word of the plan, contiguous (Adam reads them)
        self._info_used = 0
        self.stream = stream          # torch.cuda.Stream the plan runs on (None = current)
        self.side_effect_steps = set()  # steps skipped by the capture warm-up (e.g. the Adam update)
        self.explain = []               # (pass, node label, fired, why): the planner's fusion decisions (tools/dump_plan.py --explain)
        self.param_only_steps = []      # elementwise programs whose every operand is a parameter leaf
        self.prologue = []              # ... moved behind the optimiser update (they then serve the NEXT replay)
        self.trail_version = None       # session.param_version the moved steps' outputs correspond to
        self.chain_kind: Dict[int, str] = {}    # id(step) -> "full" | "tail": what a serial chain may take of it (fuse_chains)
        self.chain_members: Dict[int, list] = {}   # id(fused step) -> the steps it runs between hb_chain_begin / hb_chain_end
        self._chains_fused = False
        self.step_labels = {}           # id(step closure) -> op label (profiling)
        self.step_nodes = {}            # id(step closure) -> graph Node that emitted it (work models in bench.py)
        self.noise_inputs: Dict[Tensor, object] = {}
        self._injected: Dict[Tensor, bool] = {}
        self._leaf_resolver = leaf_resolver
        order = topo_order(self.outputs)
        self._order_pos = {n.id: i for i, n in enumerate(order)}
        self._order = order
        self._needed = set()
        for n in order:
            for t in n.inputs:
                self._needed.add(t)
        for t in self.outputs:
            self._needed.add(t)
        self._extra_copies = []
        self._fused_trinv = set()
        self._wfrag: Dict[Tensor, object] = {}   # W tensor -> fragment-major copies written by the fused factorisation
        self._afrag: Dict[Tensor, object] = {}   # A tensor of an sgp op -> its fragment-major copy (read by sgp_grad)
        self._fused_matutil = set()
        self._fused_concat = set()
        self._lazy_cols: Dict[Tensor, object] = {}
        for t, b in (binds or []):
            self._prebind(t, b)
        if prologue:
            prologue(self)
        from ._settings import settings as _st

        fuse = bool(getattr(_st.runtime, "fuse_elementwise", True))
        consumers0 = {}
        for n in order:
            for t in n.inputs:
                consumers0.setdefault(t, []).append(n)
        # the likelihood head writes d objective / d f itself when the two elementwise ops behind it feed nothing else
        self._gll_post: Dict[int, tuple] = {}
        self._absorbed = set()
        if fuse:
            outs_set = set(self.outputs)
            for n in order:
                if n.op == "gauss_ll":
                    post = _gauss_ll_post(n, consumers0, outs_set)
                    if post is not None and not any(m.id in self._absorbed for m in post[2]):
                        self._gll_post[n.id] = post
                        self._absorbed.update(m.id for m in post[2])
                        self.note("likelihood head writes dobjective/df (hb_gauss_ll_post)", n, True)
                    else:
                        self.note("likelihood head writes dobjective/df (hb_gauss_ll_post)", n, False,
                                  "dmu is not followed by exactly scale * (post * dmu) feeding nothing else" if post is None
                                  else "its elementwise tail is already absorbed by another head")
        else:
            self.note("elementwise fusion", None, False, "settings.runtime.fuse_elementwise is off")
        # column programs first (short-and-wide spaces with row reductions inside: they need the compiled form), then
        # the plain elementwise clusters over what is left
        self._colclusters = cluster_columns(order, outputs=self.outputs, skip=self._absorbed,
                                            enabled=fuse and hip_ops.ewise_jit_enabled() and bool(getattr(_st.runtime, "column_programs", True)))
        self._clusters = cluster_elementwise(order, enabled=fuse, skip=set(self._colclusters) | self._absorbed,
                                             max_elems=EW_CLUSTER_MAX_ELEMS_JIT if hip_ops.ewise_jit_enabled() else EW_CLUSTER_MAX_ELEMS)
        consumers = {}
        for n in order:
            for t in n.inputs:
                consumers.setdefault(t, []).append(n)
        self._consumers = consumers
        # The likelihood head riding in the forward contraction (hb_sgp_fwd_gauss): a gauss_ll whose f is the single latent
        # function of an sgp draw, and whose other operands exist before that draw is launched.  The head's per-point part
        # then runs in the strip kernel's finishing pass; its three sums are folded by a small step that is emitted right in
        # front of their first reader (the step's last serial chain).
        self._sgp_head: Dict[int, Node] = {}      # sgp node id -> gauss_ll node
        self._gll_in_sgp: Dict[int, Node] = {}    # gauss_ll node id -> sgp node
        self._gll_fused: Dict[int, tuple] = {}    # gauss_ll node id -> (partials buffer, units): set by _sgp_emit
        self._pending: List[tuple] = []           # (step, label, node, tensors it writes): emitted before their first reader
        if fuse and bool(getattr(_st.runtime, "head_in_contraction", True)):
            hoist_mb = self.side_jobs_enabled()

            def emitted_before(t, pos):
                while t.node.op == "reshape":      # a view: what matters is where its source comes from
                    t = t.node.inputs[0]
                nd = t.node
                if nd.op == "leaf:minibatch" and hoist_mb:
                    return True                    # minibatch gathers are emitted first (see below)
                cl = self._clusters.get(nd.id) or self._colclusters.get(nd.id)
                last = cl.nodes[-1] if (cl is not None and len(cl.nodes) >= 2) else nd
                return self._order_pos[last.id] < pos
            for n in order:
                if n.op != "gauss_ll" or len(n.inputs[1].shape) == 0:
                    continue
                ft = n.inputs[1]
                while ft.node.op == "reshape":
                    ft = ft.node.inputs[0]
                sg = ft.node
                hp = "likelihood head inside the forward contraction (hb_sgp_fwd_gauss)"
                if sg.op != "sgp" or ft is not sg.outputs[0] or sg.id in self._sgp_head or sg.inputs[5].shape[-2] != 1:
                    self.note(hp, n, False, "f is not the single latent function of an sgp draw" if sg.op != "sgp" or ft is not sg.outputs[0]
                              else ("the draw already carries a head" if sg.id in self._sgp_head else "more than one latent function (P > 1)"))
                    continue
                if n.inputs[0].size != ft.size:
                    self.note(hp, n, False, "y and f differ in size (a broadcast)")
                    continue
                pos = self._order_pos[sg.id]
                others = [n.inputs[0], n.inputs[2]] + list(n.inputs[3:4])
                if all(emitted_before(t, pos) for t in others):
                    self._sgp_head[sg.id] = n
                    self._gll_in_sgp[n.id] = sg
                    self.note(hp, n, True)
                else:
                    self.note(hp, n, False, "y / var / scale are produced after the draw is launched")
            # The same head riding in the MatBias product that produces its f (hb_matmul_gauss: a decoder layer feeding
            # densities.gaussian): f is then never written.  Conditions: a plain 2-D product (bias allowed, no activation,
            # nothing transposed) whose result feeds ONLY this head, and the head's other operands exist before it.
            self._mm_head: Dict[int, Node] = {}       # matmul node id -> gauss_ll node
            if bool(getattr(_st.runtime, "head_in_decoder", True)):
                hp = "likelihood head inside the layer's product (hb_matmul_gauss)"
                for n in order:
                    if n.op != "gauss_ll" or n.id in self._gll_in_sgp or len(n.inputs[1].shape) != 2:
                        continue
                    ft, single = n.inputs[1], True
                    while ft.node.op == "reshape":
                        single = single and len(consumers.get(ft, [])) == 1
                        ft = ft.node.inputs[0]
                    mm = ft.node
                    if mm.op != "matmul" or ft is not mm.outputs[0]:
                        continue      # (not a layer's output: nothing to report)
                    at = mm.attrs
                    why = None
                    if at["ta"] or at["tb"] or at.get("actgrad") or at["act"] != "none" or len(mm.inputs[0].shape) != 2 or len(mm.inputs[1].shape) != 2:
                        why = "the product is transposed, batched or carries an activation"
                    elif not single or len(consumers.get(ft, [])) != 1 or ft in self.outputs or ft in self._bind:
                        why = "f is read by something else as well"
                    elif n.inputs[0].shape != n.inputs[1].shape or mm.id in self._mm_head:
                        why = "y and f differ in shape"
                    elif not all(emitted_before(t, self._order_pos[mm.id]) for t in [n.inputs[0], n.inputs[2]] + list(n.inputs[3:4])):
                        why = "y / var / scale are produced after the product is launched"
                    elif hip_ops.matmul_gauss_units(ft.shape[0], mm.inputs[0].shape[1], ft.shape[1], dtype) == 0:
                        why = "shape outside the row-streaming form (fp32, n >= 2048, 32 < N <= 256, K = 16 / 32 / 64 / 128)"
                    if why is None:
                        self._mm_head[mm.id] = n
                        self._gll_in_sgp[n.id] = mm      # (same deferral of the fold as for a head riding in an sgp draw)
                    self.note(hp, n, why is None, why or "")
        # Concatenation in place: the parts of a concat along its leading non-unit axis are contiguous blocks of the
        # result, so a part that a fused program (or a single elementwise launch) produces is WRITTEN there -- the
        # gradient of a batched GP draw whose expert / gate halves come out of one column program needs no
        # scatter + add (three launches over [2E, n] arrays at cfg 5).  _concat_emit copies whatever did not land in place.
        self._place: Dict[Tensor, object] = {}
        for n in order:
            if n.op != "concat":
                continue
            o, ax = n.outputs[0], n.attrs["axis"]
            if any(d != 1 for d in o.shape[:ax]):
                continue
            dst, off = None, 0
            for t in n.inputs:
                tgt = self._place_target(t)
                if (tgt is not None and tgt not in self._bind and tgt not in self._place and tgt not in self.outputs
                        and len(consumers.get(tgt, [])) == 1):
                    if dst is None:
                        dst = self.out(o).reshape(-1)
                    self._place[tgt] = dst[off:off + t.size].view(tgt.shape)
                off += t.size
        # Weight gradient + bias gradient of a MatBias layer from one pass over the incoming gradient G: a 2-D product
        # X^T G whose right operand is also column-summed (reduce over axis 0) becomes hb_matmul_colsum -- the GEMM folds
        # the columns of G while it streams them, and the stand-alone reduction launches (two per layer) disappear.
        self._gram_for_chol: Dict[int, tuple] = {}   # cholesky node id -> (points, lengthscales, kind, jitter) of its folded Gram
        self._gram_in_mm: Dict[int, tuple] = {}      # gram_grad node id -> (lengthscale partials, rows, d, dl, groups): set by _matmul_emit
        self._chol_rider: Dict[Tensor, dict] = {}    # inverse tensor -> the persistent factorisation step's rider cell (_cholesky_emit)
        self.step_riders: Dict[int, Node] = {}       # id(step closure) -> node whose work rides in that step's launch
        self._mlp2: Dict[int, dict] = {}             # mlp2_sample_kl node id -> {fused, ws | h}
        self._colsum_of: Dict[int, Node] = {}    # matmul node id -> the reduce node it absorbs
        self._fused_colsum = set()               # ids of absorbed reduce nodes
        for n in order:
            if (n.op != "matmul" or not n.attrs["ta"] or n.attrs["tb"] or n.attrs.get("actgrad") or len(n.inputs) != 2
                    or n.attrs["act"] != "none" or len(n.inputs[0].shape) != 2 or len(n.inputs[1].shape) != 2):
                continue
            gmat = n.inputs[1]
            for r in consumers.get(gmat, []):
                ra = r.attrs if r.op == "reduce" else None
                if (ra is None or r.id in self._fused_colsum or ra["kind"] != "sum" or ra["K1"] != 1
                        or ra["R"] != gmat.shape[0] or ra["K2"] != gmat.shape[1]):
                    continue
                cl = self._clusters.get(r.id)
                if cl is not None and len(cl.nodes) > 1:
                    continue       # the reduction is part of a fused elementwise program
                rout = r.outputs[0]
                if any(self._order_pos[c.id] < self._order_pos[n.id] for c in consumers.get(rout, [])):
                    continue       # somebody reads the sums before the GEMM has run
                self._colsum_of[n.id] = r
                self._fused_colsum.add(r.id)
                self.note("weight + bias gradient from one pass (hb_matmul_colsum)", n, True)
                break
        self._emitted: List[Node] = []      # nodes in emission order
        self._side_cands: List[dict] = []   # small independent steps that may ride on a later host launch (side jobs)
        # minibatch gathers first: they depend on nothing, and emitted early they can ride on the first launch of the
        # Cholesky chain instead of being a launch of their own right before their first consumer
        hoisted = set()
        if self.side_jobs_enabled():
            for n in order:
                if n.op == "leaf:minibatch":
                    self._emit(n)
                    hoisted.add(n.id)
        for n in order:
            if n.id in hoisted:
                continue
            if n.id in self._absorbed:      # written by the likelihood head (hb_gauss_ll_post)
                self._emitted.append(n)
                continue
            cc = self._colclusters.get(n.id)
            if cc is not None:
                if n is cc.nodes[-1]:
                    self._flush_pending([t for m in cc.nodes for t in m.inputs])
                    if self._emit_colcluster(cc):
                        self._emitted.extend(cc.nodes)
                continue
            c = self._clusters.get(n.id)
            if c is None or len(c.nodes) < 2:
                self._flush_pending(n.inputs)
                self._emit(n)
            elif n is c.nodes[-1]:
                self._flush_pending([t for m in c.nodes for t in m.inputs])
                self._emit_cluster(c)
                self._emitted.extend(c.nodes)
        self._flush_pending(None)
        # outputs that could not be bound in place: explicit copy
        for t, b in list(self._bind.items()) + self._extra_copies:
            got = self._buf.get(t)
            if got is not None and got.data_ptr() != b.data_ptr():
                src = got
                self.steps.append(lambda src=src, b=b: hip_ops.ewise("COPY", [src.reshape(b.shape)], out=b))

    # -- buffer management
    def _prebind(self, t, b):
        n = t.node
        if n.op == "reshape":
            self._prebind(n.inputs[0], b.view(n.inputs[0].shape))
            return
        b = b.view(t.shape) if tuple(b.shape) != t.shape else b
        if t in self._bind:
            self._extra_copies.append((t, b))  # same tensor feeds two destinations
        else:
            self._bind[t] = b

    def _flush_pending(self, reads):
        """Append the deferred steps whose results `reads` (a list of tensors; None = all of them) needs."""
        if not self._pending:
            return
        keep = []
        for step, label, node, writes in self._pending:
            if reads is None or any(t in writes for t in reads):
                self.steps.append(step)
                self.step_labels[id(step)] = label
                self.step_nodes[id(step)] = node
            else:
                keep.append((step, label, node, writes))
        self._pending = keep

    def _place_target(self, t):
        """The tensor whose buffer a concat part really is: looks through reshapes (views) down to a value that a fused
        program or a single elementwise launch writes through Plan.out; None when the producer keeps its own buffer."""
        n = t.node
        if n.id in self._colclusters:
            return t if n.op != "strided" else None
        if n.op == "reshape":
            return self._place_target(n.inputs[0])
        if n.op == "ew":
            return t
        return None

    def needed(self, t):
        return t in self._needed

    def buf(self, t):
        b = self._buf.get(t)
        if b is None:
            raise RuntimeError("buffer of %r requested before it was produced" % (t,))
        return b

    def out(self, t):
        b = self._buf.get(t)
        if b is None:
            if t in self._place:
                b = self._place[t]     # a part of a concatenation, written in place
            elif t in self._bind and t.node.op not in ("reshape", "stop_gradient") and not t.node.op.startswith("leaf:"):
                b = self._bind[t]
            else:
                b = self.torch.empty(t.shape, dtype=self.dtype, device=self.device)
            self._buf[t] = b
        return b

    def alias(self, t, b):
        self._buf[t] = b

    def scratch(self, shape):
        return self.torch.empty(tuple(shape), dtype=self.dtype, device=self.device)

    # -- side jobs (csrc/side_jobs.cuh): small independent steps recorded instead of launched, riding on a host launch
    def side_jobs_enabled(self):
        from ._settings import settings as _st

        return bool(getattr(_st.runtime, "side_jobs", True)) and self.dtype == self.torch.float32

    def side_candidate(self, outs):
        """Register the step about to be appended as deferrable; returns the cell its closure must consult
        (`cell["defer"]` turns True when a later host launch adopts the step).  `outs`: the tensors it writes (a list
        that may still grow)."""
        cell = {"defer": False}
        self._side_cands.append(dict(cell=cell, outs=outs, epos=len(self._emitted), node=outs[0].node if outs else None))
        return cell

    def attach_side(self, host_node):
        """Called by a host-capable op before it appends its step: adopt the pending candidates whose outputs nobody
        reads between their position and the host (the host included).  True when at least one was adopted -- the
        caller then appends H.side_flush() after its own step (a no-op when the host launch took the jobs)."""
        if not self.side_jobs_enabled():
            return False
        adopted = 0
        for c in self._side_cands:
            if c["cell"]["defer"] or c.get("dead"):
                continue
            outs = set(c["outs"])
            busy = False
            for nd in self._emitted[c["epos"] + 1:]:
                if not any(t in outs for t in nd.inputs):
                    continue
                if nd.op in ("reshape", "stop_gradient"):
                    outs.update(nd.outputs)   # a view of the same buffer: nothing is read yet
                else:
                    busy = True
                    break
            if busy or any(t in outs for t in host_node.inputs):
                c["dead"] = True   # somebody needs it before any later host could run it
                self.note("side job rides on a host launch", c.get("node"), False, "its output is read before (or by) the next host launch")
                continue
            c["cell"]["defer"] = True
            self.note("side job rides on a host launch", c.get("node"), True, "host: " + host_node.op)
            adopted += 1
            if adopted == 3:
                break
        return adopted > 0

    def note(self, pass_name, node, fired, why=""):
        """Record a fusion decision of the planner: which pass looked at which node, whether it fired, and why not."""
        lab = "%s#%d%s" % (node.op, node.id, "x".join(str(d) for d in node.outputs[0].shape).join(("[", "]"))) if node is not None else "-"
        self.explain.append((pass_name, lab, bool(fired), why))

    def pin_side_reads(self, tensors):
        """A step about to be emitted reads `tensors` although they are not inputs of its node (operands of a fused
        epilogue): pending side candidates that write one of them -- directly or behind reshape / stop_gradient views --
        stop being adoptable, i.e. they run as a launch of their own at their original position."""
        src = set()
        for t in tensors:
            while t is not None:
                src.add(t)
                t = t.node.inputs[0] if t.node.op in ("reshape", "stop_gradient") and t.node.inputs else None
        for c in self._side_cands:
            if not c["cell"]["defer"] and not c.get("dead") and any(o in src for o in c["outs"]):
                c["dead"] = True

    def new_info(self, n, label):
        """`n` LAPACK-style status words for one (batched) factorisation, carved from one pool so that the
        optimiser step can look at all of them in one launch (hb_adam_step's failure containment)."""
        n = max(int(n), 1)
        if self._info_pool is None:
            self._info_pool = self.torch.zeros(INFO_POOL_WORDS, dtype=self.torch.int32, device=self.device)
        if self._info_used + n > INFO_POOL_WORDS:
            raise ValueError("more than %d factorisations in one plan" % INFO_POOL_WORDS)
        info = self._info_pool[self._info_used:self._info_used + n]
        self._info_used += n
        self._infos.append((info, label))
        return info

    def info_words(self):
        """The used part of the status pool (None when the plan factorises nothing)."""
        return None if self._info_used == 0 else self._info_pool[:self._info_used]

    def rng(self, stream):
        return self._rngs[stream]

    def not_capturable(self, why):
        self._nocap.append(why)

    # -- lowering
    def _emit(self, n: Node):
        if n.op.startswith("leaf:"):
            t = n.outputs[0]
            kind = n.op[5:]
            if kind == "const":
                arr = np.ascontiguousarray(n.attrs["value"])
                self._buf[t] = self.torch.as_tensor(arr).to(dtype=self.dtype, device=self.device).reshape(t.shape)
            elif kind == "noise":
                b = self.torch.empty(t.shape, dtype=self.dtype, device=self.device)
                self._buf[t] = b
                self.noise_inputs[t] = b
                rng = self._rngs[n.attrs["stream"]]
                plan = self

                def step(b=b, rng=rng, t=t):
                    if t not in plan._injected:
                        rng.normal(b.shape, out=b)

                self.steps.append(step)
            else:
                self._buf[t] = self._leaf_resolver(t)
            self._emitted.append(n)
            return
        d = OPS.get(n.op)
        if d is None or d.emit is None:
            raise NotImplementedError("op %s cannot be lowered" % n.op)
        before = len(self.steps)
        d.emit(self, n)
        label = n.op if n.op != "ew" else "ew:" + n.attrs["f"]
        for s in self.steps[before:]:
            if id(s) not in self.step_labels:   # (a deferred step appended on behalf of another node keeps its own label)
                self.step_labels[id(s)] = label
                self.step_nodes[id(s)] = n
        self._emitted.append(n)

    def _emit_cluster(self, c):
        """One hb_ewise_prog launch for the whole cluster."""
        H = self.H
        members = {n.id for n in c.nodes}
        space = tuple(c.space)
        keep = [i for i, d in enumerate(space) if d != 1]
        shape = [space[i] for i in keep]

        def strides_of(tshape):
            full = (1,) * (len(space) - len(tshape)) + tuple(tshape)
            st, acc = [0] * len(space), 1
            for d in range(len(space) - 1, -1, -1):
                st[d] = acc if full[d] != 1 else 0
                acc *= full[d]
            return [st[i] for i in keep]

        in_regs, inputs, istr = {}, [], []
        reg_of = {}
        code, params = [], []
        next_reg = [0]

        def operand(t):
            if t.node.id in members:
                return reg_of[t]
            r = in_regs.get(t)
            if r is None:
                r = len(inputs)
                in_regs[t] = r
                inputs.append(self.buf(t))
                istr.append(strides_of(tuple(self.buf(t).shape)))  # buffer shape: a lazy broadcast aliases the small operand
            return r

        # first pass: collect external inputs so that their registers come first
        for n in c.nodes:
            for t in n.inputs:
                if t.node.id not in members:
                    operand(t)
        next_reg[0] = len(inputs)
        for n in c.nodes:
            ops = [operand(t) for t in n.inputs]
            dst = next_reg[0]
            if n.op == "reduce":
                reg_of[n.outputs[0]] = ops[0] + H.EW_PROG_SUM  # no instruction: a flag on the output register
                continue
            if n.op == "reshape":
                code.append([H.EW["COPY"], dst, ops[0], 0, 0])
                params.append([0.0, 0.0])
                reg_of[n.outputs[0]] = dst
                next_reg[0] += 1
                continue
            f, p = n.attrs["f"], list(n.attrs["p"]) + [0.0, 0.0]
            if f == "GAUSS_LOGPDF_GRAD":
                code.append([H.EW[f], dst, ops[0], ops[1], ops[2]])
                params.append([float(ops[3]), 0.0])
                for k, o in enumerate(n.outputs):
                    reg_of[o] = dst + k
                next_reg[0] += 3
            else:
                ops = (ops + [0, 0, 0])[:3]
                code.append([H.EW[f], dst, ops[0], ops[1], ops[2]])
                params.append([p[0], p[1]])
                reg_of[n.outputs[0]] = dst
                next_reg[0] += 1
        outs, out_regs, ostr = [], [], []
        for n in c.nodes:
            for o in n.outputs:
                used_outside = o in self.outputs or any(x.id not in members for x in self._consumers.get(o, []))
                if used_outside:
                    outs.append(self.out(o))
                    out_regs.append(reg_of[o])
                    ostr.append([0] * len(keep) if o in c.reduced else strides_of(o.shape))
        ok = (len(code) <= 48 and len(inputs) <= EW_CLUSTER_MAX_IN and 1 <= len(outs) <= EW_CLUSTER_MAX_OUT
              and next_reg[0] <= EW_CLUSTER_MAX_REGS and len(shape) <= 4)
        if not ok:
            for n in c.nodes:  # too wide for one program: fall back to one launch per node
                self._emit(n)
            return
        prog = H.EwiseProgram(code, params, inputs, istr, outs, out_regs, ostr, shape)
        step = prog.launch
        self.steps.append(step)
        self.step_labels[id(step)] = "ew_cluster[%d]" % len(c.nodes)
        if in_regs and all(t.node.op in ("leaf:param", "leaf:const") for t in in_regs):
            # every operand is a parameter: the start-of-step transforms (softplus of the raw hyper-parameters, ...).  An
            # optimiser may run this step at the END of its plan instead, for the next replay (model.py: trailing transforms)
            self.param_only_steps.append(step)
        if prog.image is None:
            self.chain_kind[id(step)] = "full"   # compiled programs record themselves into a serial chain

    def _emit_colcluster(self, c):
        """One hb_ewise_colprog launch for a column cluster (falls back to one launch per node when the program
        cannot be built)."""
        H = self.H
        R, n = c.R, c.n
        ids = {m.id for m in c.nodes}
        steps_before = len(self.steps)
        try:
            inputs, in_reg = [], {}

            def ext_operand(t):
                r = in_reg.get(t)
                if r is not None:
                    return r
                if t.node.id in ids:          # an absorbed row-block slice: its source is read in place
                    form, off, rs = _col_strided_view(t.node, R, n)
                    src_t = t.node.inputs[0]
                    src = self.buf(src_t)
                    if src.numel() != src_t.size or not src.is_contiguous():
                        raise ValueError("slice of a lazily broadcast operand")
                    entry = (src, off, rs if form == "full" else 0, 1)
                else:
                    b = self.buf(t)
                    form = _colform(tuple(b.shape), R, n)    # (buffer shape: a lazy broadcast aliases the small operand)
                    if form is None or not b.is_contiguous():
                        raise ValueError("operand %s does not fit the column space" % (tuple(b.shape),))
                    entry = (b, 0, n, 1) if form == "full" else ((b, 0, 0, 1) if form == "row" else (b, 0, 0, 0))
                in_reg[t] = len(inputs)
                inputs.append(entry)
                return in_reg[t]

            for m in c.nodes:
                if m.op == "strided":
                    ext_operand(m.outputs[0])
                else:
                    for t in m.inputs:
                        if t.node.id not in ids:
                            ext_operand(t)
            reg_of, code, params = {}, [], []
            next_reg = len(inputs)

            def operand(t):
                if t.node.id in ids and t.node.op != "strided":
                    return reg_of[t]
                return ext_operand(t)

            for m in c.nodes:
                o = m.outputs[0]
                if m.op == "strided":
                    continue
                if m.op in ("reshape", "bcast"):
                    reg_of[o] = operand(m.inputs[0])
                    continue
                if m.op == "reduce":
                    code.append([H.COLPROG_SUM if m.attrs["kind"] == "sum" else H.COLPROG_MAX, next_reg, operand(m.inputs[0]), -1, -1])
                    params.append([0.0, 0.0])
                else:
                    ops = ([operand(t) for t in m.inputs] + [-1, -1, -1])[:3]
                    pp = list(m.attrs["p"]) + [0.0, 0.0]
                    code.append([H.EW[m.attrs["f"]], next_reg] + ops)
                    params.append([pp[0], pp[1]])
                reg_of[o] = next_reg
                next_reg += 1
            outs, out_regs = [], []
            for m in c.nodes:
                if m.op == "strided":
                    continue
                for o in m.outputs:
                    if o in self.outputs or any(x.id not in ids for x in self._consumers.get(o, [])):
                        form = _colform(o.shape, R, n)
                        b = self.out(o)
                        outs.append((b, 0, n, 1) if form == "full" else ((b, 0, 0, 1) if form == "row" else (b, 0, 0, 0)))
                        out_regs.append(reg_of[o])
            if not code or not outs or len(inputs) > EW_COL_MAX_IN + 4 or len(outs) > EW_COL_MAX_OUT:
                raise ValueError("column program too wide")
            prog = H.ColProgram(code, params, inputs, outs, out_regs, R, n)
        except (ValueError, _HipBackendError):
            del self.steps[steps_before:]
            for m in c.nodes:   # one launch per node, as without column programs
                self._emit(m)
            return False
        step = prog.launch
        self.steps.append(step)
        self.step_labels[id(step)] = "col_cluster[%d]" % len(c.nodes)
        return True

    def inject_noise(self, t: Tensor, value):
        """Overwrite a random_normal leaf with a fixed draw (parity runs)."""
        b = self.noise_inputs[t]
        b.copy_(self.torch.as_tensor(np.asarray(value)).to(dtype=self.dtype, device=self.device).reshape(b.shape))
        self._injected[t] = True
        self._graph = self._graphs.get(self._state_key())

    # -- execution
    def _on_stream(self):
        import contextlib

        return self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _state_key(self):
        """What the recorded launch sequence depends on besides buffer contents: which noise leaves are injected
        (their draw launches are left out).  Subclasses add their own state (injected minibatch indices)."""
        return frozenset(self._injected)

    def fuse_chains(self):
        """Serial chains (csrc/chain.cuh): a run of consecutive steps that are small dependent launches -- the tail of a
        "tail" step (the sparse-GP finishing pass, the lengthscale fold) followed by "full" steps (the likelihood head,
        compiled elementwise programs, a one-workgroup Adam update) -- becomes ONE step that brackets them with
        hb_chain_begin / hb_chain_end: the entry points record instead of launching and the whole run executes as one
        generated kernel.  Called once, when the step list is complete (capture() / the first run())."""
        if self._chains_fused:
            return
        self._chains_fused = True
        from ._settings import settings as _st

        if not bool(getattr(_st.runtime, "serial_chains", True)) or not self.H.ewise_jit_enabled():
            return
        H = self.H
        out, i, steps = [], 0, self.steps
        while i < len(steps):
            kind = self.chain_kind.get(id(steps[i]))
            j = i + 1
            if kind in ("tail", "full"):
                while j < len(steps) and self.chain_kind.get(id(steps[j])) == "full":
                    j += 1
            if j - i < 2:
                if kind in ("tail", "full"):
                    self.explain.append(("serial chain", self.step_labels.get(id(steps[i]), "other"), False,
                                         "no chainable step directly behind it (chain-aware small launches only: compiled "
                                         "elementwise programs, likelihood head, folds, one-workgroup Adam)"))
                out.append(steps[i])
                i += 1
                continue
            members = steps[i:j]
            self.explain.append(("serial chain", "+".join(self.step_labels.get(id(m), "other") for m in members), True, ""))

            def fused(members=members):
                H.chain_begin()
                try:
                    for m in members:
                        m()
                    H.chain_end()
                except BaseException:
                    H.chain_discard()
                    raise

            self.chain_members[id(fused)] = members
            self.step_labels[id(fused)] = "chain[" + "+".join(self.step_labels.get(id(m), "other") for m in members) + "]"
            self.step_nodes[id(fused)] = self.step_nodes.get(id(members[0]))
            if any(m in self.side_effect_steps for m in members):
                self.side_effect_steps.add(fused)
            out.append(fused)
            i = j
        self.steps = out

    def run(self):
        self.fuse_chains()
        with self._on_stream():
            if self._graph is None and self._capture_enabled and not self._nocap:
                # the injection state changed since the last capture: one hipGraph per state, captured on demand
                self._graph = self._graphs.get(self._state_key())
                if self._graph is None:
                    self._capture_now()
            if self._graph is not None:
                self._graph.launch()
            else:
                self._run_eager(self.steps)

    def _run_eager(self, steps):
        """The launch sequence, step by step.  Side jobs (csrc/side_jobs.cuh) are per-thread state of the library, not
        of a plan: whatever is pending when a plan starts belongs to a sequence somebody abandoned and is dropped, a
        plan must leave nothing pending, and a step that raises takes its recorded jobs with it -- raw device pointers
        never ride on a launch of another plan."""
        H = self.H
        H.side_discard()
        H.chain_discard()
        try:
            for s in steps:
                s()
            if H.side_pending():
                raise RuntimeError("plan left %d side job(s) pending: a deferred step has no flushing host" % H.side_pending())
        except BaseException:
            H.side_discard()
            raise

    def capture(self):
        """Record the launch sequence into one hipGraph (replayed by run()).  A
        warm-up pass first lets lazily sized workspaces allocate outside the
        capture; steps with side effects beyond the plan's own buffers (the Adam
        update, the data-parallel pack + all-reduce) are skipped in it and the RNG
        streams are put back afterwards, so a capture -- also a later re-capture
        on one rank only -- is invisible to the trajectory and issues no collective.
        Later changes of the injection state (set_indices / clear_indices /
        inject_noise) re-capture on the next run; graphs are kept per state."""
        if self._nocap:
            return False
        self.fuse_chains()
        self._capture_enabled = True
        self.torch.cuda.synchronize()
        with self._on_stream():
            self._capture_now()
        return True

    def _capture_now(self):
        st = self.torch.cuda.current_stream()
        saved = {k: r.state.clone() for k, r in self._rngs.items()} if self._rngs else {}
        # (a fused chain with a side-effect member runs its other members, unchained: lazily sized workspaces are theirs)
        warm = []
        for s in self.steps:
            if s not in self.side_effect_steps:
                warm.append(s)
            else:
                warm += [m for m in self.chain_members.get(id(s), ()) if m not in self.side_effect_steps]
        self._run_eager(warm)
        for k, r in self._rngs.items():
            r.state.copy_(saved[k])
        st.synchronize()
        g = self.H.CapturedGraph()
        self.H.side_discard()
        g.begin()
        try:
            try:
                for s in self.steps:
                    s()
            finally:
                g.end()
        except BaseException:
            self.H.side_discard()
            raise
        if self.H.side_pending():
            n = self.H.side_discard()
            raise RuntimeError("capture left %d side job(s) pending: a deferred step has no flushing host" % n)
        self._graph = g
        self._graphs[self._state_key()] = g

    @property
    def is_captured(self):
        return self._graph is not None

    def profile(self, iters=20):
        """Per-op device time (us) from an eager, event-bracketed replay of the
        plan: {label: (avg_us_per_run, launches_per_run)}.  Diagnostic; the
        captured graph is what run() replays."""
        torch = self.torch
        acc, cnt = {}, {}
        with self._on_stream():
            st = torch.cuda.current_stream()
            for it in range(iters + 2):
                evs = []
                for s in self.steps:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    s()
                    e1.record(st)
                    evs.append((self.step_labels.get(id(s), "other"), e0, e1))
                st.synchronize()
                if it < 2:
                    continue
                for lab, e0, e1 in evs:
                    acc[lab] = acc.get(lab, 0.0) + e0.elapsed_time(e1) * 1e3
                    cnt[lab] = cnt.get(lab, 0) + 1
        return {k: (acc[k] / iters, cnt[k] // iters) for k in acc}

    def check(self):
        """Synchronise and raise if any Cholesky in the plan failed."""
        if self.stream is not None:
            self.stream.synchronize()
        else:
            self.torch.cuda.synchronize()
        if self._info_used == 0:
            return
        pool = self._info_pool[:self._info_used].cpu().numpy()  # ONE small copy for every factorisation of the plan
        if not pool.any():
            return
        off = 0
        for info, label in self._infos:
            host = pool[off:off + info.numel()]
            off += info.numel()
            bad = np.flatnonzero(host)
            if bad.size:
                raise CholeskyError("%s: leading minor %d is not positive definite (matrix %d)"
                                    % (label, int(host[bad[0]]), int(bad[0])))

    def value(self, t: Tensor):
        return self.buf(t).detach().cpu().numpy()
