"""Device-buffer level wrappers over the C ABI (include/henbun_hip.h).

PyTorch is used here ONLY as plumbing: `torch.empty(device='cuda')` for device
memory, `tensor.data_ptr()` for the raw pointer and the current stream handle.
No torch op computes anything on this path -- every function below is one (or
a fixed few) launches of hand-written HIP kernels through ctypes.  If the
backend library is missing, `_lib.lib()` raises; there is no fallback.
"""
from __future__ import annotations

import ctypes
from ctypes import c_double, c_int, c_long, c_void_p

import numpy as np
import torch

from . import _lib

# ---- enums mirrored from include/henbun_hip.h -------------------------------
EW = dict(
    NEG=1, EXP=2, LOG=3, SQRT=4, SQUARE=5, ABS=6, SIGN=7, SIGMOID=8, RELU=9, SOFTPLUS=10, TANH=11,
    RECIP=12, RSQRT=13, STEP=14, AFFINE=15, CLIP=16, CLIPMASK=17, LGAMMA=18, POWC=19, LOG1P=20,
    COPY=21, DIGAMMA=22,
    ADD=32, SUB=33, MUL=34, DIV=35, MAX=36, MIN=37, POW=38, GT=39, GE=40, LT=41, LE=42, EQ=43,
    SIGMOID_GRAD=44, TANH_GRAD=45, RELU_GRAD=46, SOFTPLUS_GRAD=47, CLIP_GRAD=48,
    WHERE=64, FMA=65, GAUSS_LOGPDF=66,
    GAUSS_LOGPDF_GRAD=80,
)
RED_SUM, RED_MAX = 0, 1
KERN_RBF, KERN_CSYM_RBF, KERN_SQDIST = 0, 1, 2
KERN_KBAR_SYMMETRIC = 256   # OR-ed into the kind of gram_bwd: Kbar is symmetric (no transposed reads)
EW_PROG_SUM = 256
MM_LOWER_OUT = 1
MM_TRIL_OUT = 2
MM_PHI_OUT = 4
MM_SYM_OUT = 8
MM_SYMLOW_OUT = 32
MM_ACTGRAD = 16
ACT = dict(none=0, sigmoid=1, relu=2, tanh=3)
SGP_NEGLECTED, SGP_DIAGONAL = 0, 1
MATUTIL_BAND, MATUTIL_ADD_EYE, MATUTIL_PHI, MATUTIL_SYM = 0, 1, 2, 3

WS_ELEMS = 1 << 16  # generic scratch (elements) for reductions / KL partials


def _suf(t: torch.Tensor) -> str:
    if t.dtype == torch.float32:
        return "_f32"
    if t.dtype == torch.float64:
        return "_f64"
    raise TypeError("henbun_amd kernels compute in float32 or float64, got %s" % t.dtype)


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def _chk(t: torch.Tensor, name="tensor"):
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU" % name)
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def _larr(vals):
    return (c_long * len(vals))(*[int(v) for v in vals])


_capturing = False


def set_capturing(flag):
    """While a hipGraph is being captured every buffer must already exist: a tensor allocated inside
    the capture would be returned to torch's caching allocator afterwards and the graph would keep
    replaying into memory that now belongs to something else."""
    global _capturing
    _capturing = bool(flag)


def _empty(*a, **k):
    if _capturing:
        raise RuntimeError("henbun_amd: device allocation during hipGraph capture (an op was emitted without "
                           "preallocated outputs/workspace)")
    return torch.empty(*a, **k)


def _empty_like(t):
    if _capturing:
        raise RuntimeError("henbun_amd: device allocation during hipGraph capture (an op was emitted without "
                           "preallocated outputs/workspace)")
    return torch.empty_like(t)


_ws_cache = {}


_ws_retired = []


def workspace(dtype, device, elems=WS_ELEMS) -> torch.Tensor:
    """Shared scratch for the stream-ordered launches of one stream.  A workspace that has to grow is
    retired, never freed: captured hipGraphs may still hold its address."""
    key = (dtype, str(device), torch.cuda.current_stream().cuda_stream)
    w = _ws_cache.get(key)
    if w is None or w.numel() < elems:
        if w is not None:
            _ws_retired.append(w)
        w = _empty(max(elems, WS_ELEMS), dtype=dtype, device=device)
        _ws_cache[key] = w
    return w


def _bcast_strides(shape, out_shape):
    """element strides of a contiguous tensor of `shape` viewed as `out_shape` (0 on broadcast dims)."""
    nd = len(out_shape)
    shape = (1,) * (nd - len(shape)) + tuple(shape)
    strides, acc = [0] * nd, 1
    for d in range(nd - 1, -1, -1):
        if shape[d] == 1 and out_shape[d] != 1:
            strides[d] = 0
        elif shape[d] == out_shape[d]:
            strides[d] = acc if shape[d] != 1 else 0
        else:
            raise ValueError("shape %s does not broadcast to %s" % (shape, out_shape))
        acc *= shape[d]
    return strides


def ewise(op, inputs, nout=1, params=None, out=None):
    """n-ary broadcasting elementwise op (hb_ewise_*).  Returns a tensor or a tuple."""
    opc = EW[op] if isinstance(op, str) else int(op)
    inputs = [_chk(t, "input") for t in inputs]
    out_shape = tuple(torch.broadcast_shapes(*[tuple(t.shape) for t in inputs]))
    if out is not None:
        # explicit result buffers may be larger than the operands' common shape (operands broadcast into them)
        o0 = out[0] if isinstance(out, (list, tuple)) else out
        out_shape = tuple(torch.broadcast_shapes(out_shape, tuple(o0.shape)))
    if len(out_shape) > 6:
        raise ValueError("elementwise ops support at most 6 dims")
    nin = len(inputs)
    nd = len(out_shape)
    strides = []
    for t in inputs:
        strides += _bcast_strides(tuple(t.shape), out_shape)
    if out is None:
        outs = [_empty(out_shape, dtype=inputs[0].dtype, device=inputs[0].device) for _ in range(nout)]
    else:
        outs = list(out) if isinstance(out, (list, tuple)) else [out]
        for o in outs:
            assert tuple(o.shape) == out_shape and o.is_contiguous()
    in_arr = (c_void_p * nin)(*[t.data_ptr() for t in inputs])
    out_arr = (c_void_p * nout)(*[o.data_ptr() for o in outs])
    pr = (c_double * 4)(*(list(params or []) + [0.0] * 4)[:4])
    _lib.lib().call("hb_ewise" + _suf(inputs[0]), opc, nin, in_arr, _larr(strides) if strides else _larr([0]),
                    nout, out_arr, nd, _larr(out_shape) if nd else _larr([1]), pr, stream())
    return outs[0] if nout == 1 else tuple(outs)


EWISE_JIT_SOURCE_BYTES = 8192


def ewise_jit_enabled():
    """Fused elementwise programs are compiled at plan-build time (hiprtc) unless `settings.runtime.ewise = interpret`
    or hiprtc cannot be loaded in this process; the interpreted form is the same program run by ew_prog_image_kernel."""
    from ._settings import settings

    if getattr(settings.runtime, "ewise", "jit") != "jit":
        return False
    return bool(_lib.lib().raw("hb_ewise_jit_available")())


class EwiseProgram:
    """A prepared hb_ewise_prog launch (all host-side argument arrays built once)."""

    def __init__(self, code, params, inputs, istrides, outputs, out_regs, ostrides, shape):
        from ctypes import c_int

        self.suf = _suf(outputs[0])
        self.ninstr = len(code)
        self.code = (c_int * (5 * len(code)))(*[int(v) for ins in code for v in ins])
        self.params = (c_double * (2 * len(code)))(*[float(v) for pr in params for v in pr])
        self.nin = len(inputs)
        self.inputs = (c_void_p * max(self.nin, 1))(*[t.data_ptr() for t in inputs])
        self.nd = len(shape)
        flat_is = [s for st in istrides for s in st]
        self.istr = _larr(flat_is) if flat_is else _larr([0])
        self.nout = len(outputs)
        self.outputs = (c_void_p * self.nout)(*[t.data_ptr() for t in outputs])
        self.out_regs = (c_int * self.nout)(*[int(r) for r in out_regs])
        flat_os = [s for st in ostrides for s in st]
        self.ostr = _larr(flat_os) if flat_os else _larr([0])
        self.shape = _larr(shape) if shape else _larr([1])
        self._keep = (inputs, outputs)
        lib = _lib.lib()
        self.handle = None
        self.source = None
        if ewise_jit_enabled():
            # compiled form: the program becomes a gfx950 kernel of its own at plan-build time (hb_ewise_jit_*)
            n_out, red_out = c_long(0), c_int(0)
            handle = c_void_p(None)
            src = ctypes.create_string_buffer(EWISE_JIT_SOURCE_BYTES)
            lib.call("hb_ewise_jit_build" + self.suf, self.ninstr, self.code, self.params, self.nin, self.inputs, self.istr,
                     self.nout, self.outputs, self.out_regs, self.ostr, self.nd, self.shape, ctypes.byref(handle),
                     ctypes.byref(n_out), ctypes.byref(red_out), src, EWISE_JIT_SOURCE_BYTES)
            self.n, self.reduces = int(n_out.value), int(red_out.value)
            self.handle = handle
            self.source = src.value.decode()
            self.image = None
            return
        # interpreted form: the launch descriptor is built and uploaded once; a (replayed) launch carries one pointer
        # and the kernel pulls the descriptor into LDS in one parallel load
        nbytes = int(lib.raw("hb_ewise_prog_image_bytes")())
        host = ctypes.create_string_buffer(nbytes)
        n_out, red_out = c_long(0), c_int(0)
        lib.call("hb_ewise_prog_build", self.ninstr, self.code, self.params, self.nin, self.inputs, self.istr, self.nout,
                 self.outputs, self.out_regs, self.ostr, self.nd, self.shape, host, ctypes.byref(n_out),
                 ctypes.byref(red_out))
        self.n, self.reduces = int(n_out.value), int(red_out.value)
        self.image = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(outputs[0].device)
        torch.cuda.synchronize()

    def launch(self):
        if self.image is None:
            _lib.lib().call("hb_ewise_jit_run", self.handle, stream())
        else:
            _lib.lib().call("hb_ewise_prog_run" + self.suf, _p(self.image), self.n, self.reduces, stream())

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            try:
                _lib.lib().raw("hb_ewise_jit_destroy")(h)
            except Exception:
                pass


COLPROG_SUM, COLPROG_MAX = -1, -2   # row reductions of a column program (HB_COLPROG_* in include/henbun_hip.h)


class ColProgram:
    """A prepared column program (hb_ewise_colprog_build): a fused program over an [R, n] space, one thread per
    column, reductions over the row axis as ordinary instructions.

    `inputs` / `outputs`: lists of (tensor, element offset, row stride, column stride); code rows are
    [op, dst, a, b, c] with -1 for an unused operand slot."""

    def __init__(self, code, params, inputs, outputs, out_regs, R, n, dry=False):
        from ctypes import c_int

        ref = (outputs[0][0] if outputs else inputs[0][0])
        self.suf = _suf(ref)
        isz = ref.element_size()
        self.ninstr = len(code)
        self.code = (c_int * (5 * len(code)))(*[int(v) for ins in code for v in ins])
        self.params = (c_double * (2 * len(code)))(*[float(v) for pr in params for v in pr])
        self.nin, self.nout = len(inputs), len(outputs)
        self.inputs = (c_void_p * max(self.nin, 1))(*[t.data_ptr() + int(off) * isz for t, off, _, _ in inputs])
        self.istr = _larr([v for _, _, rs, cs in inputs for v in (rs, cs)] or [0])
        self.outputs = (c_void_p * self.nout)(*[t.data_ptr() + int(off) * isz for t, off, _, _ in outputs])
        self.ostr = _larr([v for _, _, rs, cs in outputs for v in (rs, cs)])
        self.out_regs = (c_int * self.nout)(*[int(r) for r in out_regs])
        self._keep = (inputs, outputs)
        self.R, self.n = int(R), int(n)
        src = ctypes.create_string_buffer(EWISE_JIT_SOURCE_BYTES)
        handle = c_void_p(None)
        _lib.lib().call("hb_ewise_colprog_build" + self.suf, self.ninstr, self.code, self.params, self.nin, self.inputs, self.istr,
                        self.nout, self.outputs, self.out_regs, self.ostr, self.R, self.n,
                        None if dry else ctypes.byref(handle), src, EWISE_JIT_SOURCE_BYTES)
        self.handle = None if dry else handle
        self.source = src.value.decode()

    def launch(self):
        _lib.lib().call("hb_ewise_jit_run", self.handle, stream())

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            try:
                _lib.lib().raw("hb_ewise_jit_destroy")(h)
            except Exception:
                pass


def gauss_ll(x, f, scale, var, out=None, post=None, fbar=None):
    """(ll[1], dmu[like x], dscale[1], dvar[1]) of sum_j log N(x_j | f_j*scale, var)  (hb_gauss_ll; scale may be None).
    `post`, `fbar`: also fbar = scale * (post * dmu), the gradient of post * ll w.r.t. f (hb_gauss_ll_post)."""
    _chk(x), _chk(f), _chk(var)
    assert x.numel() == f.numel() and var.numel() == 1 and (scale is None or scale.numel() == 1)
    n = x.numel()
    if out is None:
        ll = _empty(1, dtype=x.dtype, device=x.device)
        dmu = _empty_like(x)
        ds = _empty(1, dtype=x.dtype, device=x.device)
        dv = _empty(1, dtype=x.dtype, device=x.device)
    else:
        ll, dmu, ds, dv = out
    ws = workspace(x.dtype, x.device, 3 * max((n + 1023) // 1024, 1))
    if fbar is not None:
        _chk(fbar)
        assert fbar.numel() == n
        _lib.lib().call("hb_gauss_ll_post" + _suf(x), _p(x), _p(f), _p(scale), _p(var), n, _p(ll), _p(dmu), _p(ds), _p(dv),
                        float(post), _p(fbar), _p(ws), ws.numel(), stream())
        return ll, dmu, ds, dv
    _lib.lib().call("hb_gauss_ll" + _suf(x), _p(x), _p(f), _p(scale), _p(var), n, _p(ll), _p(dmu), _p(ds), _p(dv), _p(ws),
                    ws.numel(), stream())
    return ll, dmu, ds, dv


def reduce_mid(x, K1, R, K2, op=RED_SUM, out=None):
    """out[K1,K2] = reduce_R x[K1,R,K2] (x contiguous, any shape with K1*R*K2 elements)."""
    _chk(x)
    assert x.numel() == K1 * R * K2
    if out is None:
        out = _empty(K1 * K2, dtype=x.dtype, device=x.device)
    ws = workspace(x.dtype, x.device)
    _lib.lib().call("hb_reduce" + _suf(x), op, _p(x), _p(out), K1, R, K2, _p(ws), ws.numel(), stream())
    return out


def copy_nd(src, src_strides, dst, dst_strides, shape, src_off=0, dst_off=0):
    """strided element copy; strides/offsets in elements over the flat buffers."""
    es = src.element_size()
    sp = c_void_p(src.data_ptr() + src_off * es)
    dp = c_void_p(dst.data_ptr() + dst_off * es)
    nd = len(shape)
    _lib.lib().call("hb_copy_nd" + _suf(src), sp, _larr(src_strides) if nd else _larr([0]), dp,
                    _larr(dst_strides) if nd else _larr([0]), nd, _larr(shape) if nd else _larr([1]), stream())
    return dst


def fill(t, value):
    _lib.lib().call("hb_fill" + _suf(t), _p(t), t.numel(), float(value), stream())
    return t


def gather_rows(src, idx, perm=None, out=None, err=None):
    """out[i,:] = src[perm[idx[i]],:]  (K0)."""
    _chk(src)
    nsrc = src.shape[0]
    row = src.numel() // max(nsrc, 1)
    n = idx.numel()
    if out is None:
        out = _empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    assert idx.dtype == torch.int64 and (perm is None or perm.dtype == torch.int64)
    _lib.lib().call("hb_gather_rows" + _suf(src), _p(src), nsrc, row, _p(idx), _p(perm), n, _p(out), _p(err), stream())
    return out


class MultiGather:
    """out_a[i,:] = src_a[perm[idx[i]],:] for several arrays sharing idx, one launch (argument arrays built once)."""

    def __init__(self, srcs, outs, idx, perm, err):
        assert 1 <= len(srcs) <= 8 and len(srcs) == len(outs)
        nsrc = srcs[0].shape[0]
        for s in srcs:
            _chk(s)
            assert s.shape[0] == nsrc and s.dtype == srcs[0].dtype
        self.suf = _suf(srcs[0])
        self.narr, self.nsrc, self.n = len(srcs), nsrc, idx.numel()
        self.srcs = (c_void_p * self.narr)(*[s.data_ptr() for s in srcs])
        self.dsts = (c_void_p * self.narr)(*[o.data_ptr() for o in outs])
        self.rows = _larr([s.numel() // max(nsrc, 1) for s in srcs])
        self.idx, self.perm, self.err = idx, perm, err
        self._keep = (srcs, outs)

    def launch(self, use_perm=True):
        _lib.lib().call("hb_gather_rows_multi" + self.suf, self.narr, self.srcs, self.rows, self.dsts, self.nsrc,
                        _p(self.idx), _p(self.perm if use_perm else None), self.n, _p(self.err), stream())

    def launch_draw(self, rng, lo, hi, use_perm=True, defer=False):
        """draw idx ~ U{lo..hi-1} from `rng` (as rng.randint would) and gather, in one launch; needs n <= rng.nlanes.
        defer: recorded as a side job of the next host launch (fp32 only) instead of launched now."""
        name = "hb_side_push_gather_draw_f32" if (defer and self.suf == "_f32") else "hb_gather_rows_multi_draw" + self.suf
        _lib.lib().call(name, self.narr, self.srcs, self.rows, self.dsts, self.nsrc,
                        _p(rng.state), rng.nlanes, int(lo), int(hi), _p(self.idx), _p(self.perm if use_perm else None),
                        self.n, _p(self.err), stream())


def matutil(x, mode, lower=-1, upper=-1, alpha=0.0, out=None):
    _chk(x)
    R, C = x.shape[-2], x.shape[-1]
    B = x.numel() // max(R * C, 1)
    if out is None:
        out = _empty_like(x)
    _lib.lib().call("hb_matutil" + _suf(x), _p(x), _p(out), B, R, C, mode, lower, upper, float(alpha), stream())
    return out


# ---- RNG --------------------------------------------------------------------
class Rng:
    """xoroshiro128+ per-lane state on the device (hb_rng_*)."""

    def __init__(self, seed, stream_id=0, nlanes=65536, device="cuda"):
        self.nlanes = int(nlanes)
        self.state = _empty(2 * self.nlanes, dtype=torch.int64, device=device)
        self.seed, self.stream_id = int(seed), int(stream_id)
        self.reseed(seed, stream_id)

    def reseed(self, seed, stream_id=0):
        self.seed, self.stream_id = int(seed), int(stream_id)
        _lib.lib().call("hb_rng_init", _p(self.state), self.nlanes, int(seed) & (2**64 - 1),
                        int(stream_id) & (2**64 - 1), stream())

    def normal(self, shape, dtype=torch.float32, out=None):
        if out is None:
            out = _empty(shape, dtype=dtype, device=self.state.device)
        _lib.lib().call("hb_rng_normal" + _suf(out), _p(self.state), self.nlanes, _p(out), out.numel(), stream())
        return out

    def randint(self, n, lo, hi, out=None):
        if out is None:
            out = _empty(n, dtype=torch.int64, device=self.state.device)
        _lib.lib().call("hb_rng_randint", _p(self.state), self.nlanes, _p(out), out.numel(), int(lo), int(hi), stream())
        return out


def _rng_args(rng):
    if rng is None:
        return None, 0
    return _p(rng.state), rng.nlanes


# ---- K1/K2 variational sampler + MC-KL --------------------------------------
def side_flush():
    """Launch whatever side jobs are still pending on this thread (hb_side_flush); a no-op when a host launch took them."""
    _lib.lib().call("hb_side_flush", stream())


def side_pending():
    return int(_lib.lib().raw("hb_side_pending")())


def chain_begin():
    """Start recording a serial chain on this thread (hb_chain_begin): the small launches of the chain-aware entry points
    that follow run as one generated kernel at chain_end()."""
    _lib.lib().call("hb_chain_begin")


def chain_end():
    _lib.lib().call("hb_chain_end", stream())


def chain_discard():
    return int(_lib.lib().raw("hb_chain_discard")())


def chain_source():
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.lib().call("hb_chain_source", buf, 1 << 16)
    return buf.value.decode()


def side_discard():
    """Drop this thread's recorded side jobs without running them (hb_side_discard); returns how many there were."""
    return int(_lib.lib().raw("hb_side_discard")())


def diag_sample_kl_fwd(mu, s, u_in=None, rng=None, out=None, rows=None, defer=False):
    """x = mu + exp(s)*u ; kl = -0.5*sum(2s + u^2 - x^2).  Returns (x, kl, u).
    rows=(nrows, L, ld_mu, ld_s): mu and s are column blocks of wider row-major matrices (1-D views starting at their
    first element), read in place; x and u are dense [nrows, L] (out= is then required)."""
    _chk(mu), _chk(s)
    if rows is None:
        n = mu.numel()
        L = ldm = lds = max(n, 1)
    else:
        nrows, L, ldm, lds = (int(v) for v in rows)
        n = nrows * L
        if out is None:
            raise ValueError("diag_sample_kl_fwd: strided inputs need explicit outputs")
    if out is None:
        x, kl, u = _empty_like(mu), _empty(1, dtype=mu.dtype, device=mu.device), _empty_like(mu)
    else:
        x, kl, u = out
    ws = workspace(mu.dtype, mu.device)
    rp, rl = _rng_args(rng)
    # defer: recorded as a side job of the next host launch (hb_side_push_*; fp32 only) instead of launched now
    name = "hb_side_push_diag_fwd_f32" if (defer and mu.dtype == torch.float32) else "hb_diag_sample_kl_fwd" + _suf(mu)
    _lib.lib().call(name, _p(mu), _p(s), _p(u_in), rp, rl, _p(u), _p(x), _p(kl), n, L, ldm, lds, _p(ws), stream())
    return x, kl, u


def diag_sample_kl_bwd(s, u, x, xbar, klbar, out=None, rows=None, defer=False):
    """rows=(nrows, L, ld_s, ld_out): s read, mubar/sbar written, as column blocks of wider matrices (see _fwd)."""
    if rows is None:
        n = s.numel()
        L = lds = ldo = max(n, 1)
    else:
        nrows, L, lds, ldo = (int(v) for v in rows)
        n = nrows * L
        if out is None:
            raise ValueError("diag_sample_kl_bwd: strided layout needs explicit outputs")
    if out is None:
        mubar, sbar = _empty_like(s), _empty_like(s)
    else:
        mubar, sbar = out
    name = "hb_side_push_diag_bwd_f32" if (defer and s.dtype == torch.float32) else "hb_diag_sample_kl_bwd" + _suf(s)
    _lib.lib().call(name, _p(s), _p(u), _p(x), _p(xbar), _p(klbar), _p(mubar), _p(sbar), n, L, lds, ldo, stream())
    return mubar, sbar


def fullrank_sample_kl_fwd(mu, S, u_in=None, rng=None, out=None, packed=False):
    """x_r = mu_r + tril(S_r) u_r over rows; S [..., size, size], or packed [..., size(size+1)/2] (lower triangle,
    tril_indices order)."""
    _chk(mu), _chk(S)
    size = mu.shape[-1]
    rows = mu.numel() // max(size, 1)
    assert S.numel() == rows * (size * (size + 1) // 2 if packed else size * size)
    if out is None:
        x, kl, u = _empty_like(mu), _empty(1, dtype=mu.dtype, device=mu.device), _empty_like(mu)
    else:
        x, kl, u = out
    ws = workspace(mu.dtype, mu.device)
    rp, rl = _rng_args(rng)
    # (one launch for a block of up to 1024 dimensions; the entry falls back to the three-launch form by itself)
    _lib.lib().call("hb_fullrank_sample_kl_fwd1" + _suf(mu), _p(mu), _p(S), _p(u_in), rp, rl, _p(u), _p(x), _p(kl),
                    rows, size, int(bool(packed)), _p(ws), _p(sync_word(mu.device)), stream())
    return x, kl, u


_SYNC_WORDS = {}


def sync_word(device):
    """One zero 32-bit word per (device, stream) for the kernels that meet at an arrival counter (zero at entry, left zero
    by every call; launches on one stream are ordered, so they can share it)."""
    key = (str(device), stream())
    b = _SYNC_WORDS.get(key)
    if b is None:
        b = _SYNC_WORDS[key] = torch.zeros(4, dtype=torch.int32, device=device)
    return b


def fullrank_sample_kl_bwd(S, u, x, xbar, klbar, out=None, packed=False):
    size = u.shape[-1]
    rows = u.numel() // max(size, 1)
    if out is None:
        mubar, Sbar = _empty_like(u), _empty_like(S)
    else:
        mubar, Sbar = out
    _lib.lib().call("hb_fullrank_sample_kl_bwd" + _suf(S), _p(S), _p(u), _p(x), _p(xbar), _p(klbar), _p(mubar),
                    _p(Sbar), rows, size, int(bool(packed)), stream())
    return mubar, Sbar


def tri_size(n_packed):
    """N with N(N+1)/2 == n_packed."""
    N = int((8 * n_packed + 1) ** 0.5 / 2.0 - 0.5 + 1e-9)
    if N * (N + 1) // 2 != n_packed:
        raise ValueError("%d is not a triangular number" % n_packed)
    return N


def vec_to_tri(v, out=None):
    """[..., N(N+1)/2] -> lower-triangular [..., N, N] (reference tf_wraps.py:50-71)."""
    _chk(v)
    N = tri_size(v.shape[-1])
    B = v.numel() // max(v.shape[-1], 1)
    if out is None:
        out = _empty(tuple(v.shape[:-1]) + (N, N), dtype=v.dtype, device=v.device)
    _lib.lib().call("hb_vec_to_tri" + _suf(v), _p(v), _p(out), B, N, stream())
    return out


def tri_to_vec(tri, out=None):
    """Lower triangle of [..., N, N] -> [..., N(N+1)/2] (the gradient of vec_to_tri)."""
    _chk(tri)
    N = tri.shape[-1]
    B = tri.numel() // max(N * N, 1)
    if out is None:
        out = _empty(tuple(tri.shape[:-2]) + (N * (N + 1) // 2,), dtype=tri.dtype, device=tri.device)
    _lib.lib().call("hb_tri_to_vec" + _suf(tri), _p(tri), _p(out), B, N, stream())
    return out


# ---- K3 Gram ------------------------------------------------------------------
def _batch_view(X, nd_tail=2):
    """(B, stride) of a [..., n, d] tensor flattened over leading dims."""
    n, d = X.shape[-2], X.shape[-1]
    B = X.numel() // max(n * d, 1)
    return B, n, d


def _ell_layout(ell, B, d):
    """(sEll, dl): a lengthscale tensor [dl] is shared by the batch, [B, dl] gives one kernel per batch entry."""
    if ell.dim() >= 2 and ell.shape[0] == B and B > 1:
        dl = ell.numel() // B
        return dl, dl
    return 0, ell.numel()


def gram_fwd(X, X2, ell, kind=KERN_RBF, out=None, diag_add=0.0):
    """K[b,i,j] = k(X[b,i], X2[b,j]); X, X2: [n,d] or [B,n,d] (a 2-D operand is shared over B);
    ell: [dl] (shared) or [B, dl] (one kernel per batch entry)."""
    _chk(X), _chk(X2), _chk(ell)
    BX, n, d = _batch_view(X)
    BX2, n2, d2 = _batch_view(X2)
    assert d == d2
    B = max(BX, BX2)
    if ell.dim() >= 2 and ell.shape[0] > 1:
        B = max(B, ell.shape[0])
    assert BX in (1, B) and BX2 in (1, B)
    sX = n * d if (BX == B and B > 1) else 0
    sX2 = n2 * d if (BX2 == B and B > 1) else 0
    sEll, dl = _ell_layout(ell, B, d)
    batched = X.dim() > 2 or X2.dim() > 2 or sEll != 0
    lead = tuple(X.shape[:-2]) if X.dim() > 2 else (tuple(X2.shape[:-2]) if X2.dim() > 2 else ((B,) if sEll else ()))
    if out is None:
        out = _empty((lead if batched else ()) + (n, n2), dtype=X.dtype, device=X.device)
    _lib.lib().call("hb_gram_fwd" + _suf(X), kind, _p(X), sX, _p(X2), sX2, _p(ell), sEll, dl, _p(out), B, n, n2, d,
                    float(diag_add), stream())
    return out


def gram_bwd_raw(kind, X, sX, X2, sX2, ell, sEll, dl, Kbar, Xbar, X2bar, ellbar, B, n, n2, d, ws):
    """hb_gram_bwd on caller-provided buffers (no allocation: graph-capturable)."""
    _lib.lib().call("hb_gram_bwd" + _suf(X), kind, _p(X), sX, _p(X2), sX2, _p(ell), sEll, dl, _p(Kbar), _p(Xbar),
                    _p(X2bar), _p(ellbar), B, n, n2, d, _p(ws), stream())


def gram_bwd(X, X2, ell, Kbar, kind=KERN_RBF, need=(True, True, True)):
    """VJP of gram_fwd: returns (Xbar, X2bar, ellbar) (None where not needed).
    Operands shared over the batch get their gradient summed over it."""
    _chk(X), _chk(X2), _chk(ell), _chk(Kbar)
    BX, n, d = _batch_view(X)
    BX2, n2, _ = _batch_view(X2)
    B = max(BX, BX2)
    if ell.dim() >= 2 and ell.shape[0] > 1:
        B = max(B, ell.shape[0])
    sX = n * d if (BX == B and B > 1) else 0
    sX2 = n2 * d if (BX2 == B and B > 1) else 0
    sEll, dl = _ell_layout(ell, B, d)
    dev, dt = X.device, X.dtype
    Xbar = _empty((B, n, d), dtype=dt, device=dev) if need[0] else None
    X2bar = _empty((B, n2, d), dtype=dt, device=dev) if need[1] else None
    ellbar = _empty(ell.numel(), dtype=dt, device=dev) if need[2] else None
    ws = workspace(dt, dev, max(B * n * d, 1))
    gram_bwd_raw(kind, X, sX, X2, sX2, ell, sEll, dl, Kbar, Xbar, X2bar, ellbar, B, n, n2, d, ws)
    if Xbar is not None:
        Xbar = reduce_mid(Xbar, 1, B, n * d).reshape(X.shape) if (BX == 1 and B > 1) else Xbar.reshape(X.shape)
    if X2bar is not None:
        X2bar = reduce_mid(X2bar, 1, B, n2 * d).reshape(X2.shape) if (BX2 == 1 and B > 1) else X2bar.reshape(X2.shape)
    if ellbar is not None:
        ellbar = ellbar.reshape(ell.shape)
    return Xbar, X2bar, ellbar


# ---- dense linear algebra ------------------------------------------------------
def matmul(A, B, transA=False, transB=False, alpha=1.0, bias=None, act="none", lower_out=False, out=None,
           beta=0.0, tril_out=False, epilogue=0, actgrad=None):
    """C = act(alpha*op(A)@op(B) + bias) (+ beta*C).  A:[...,m,k], B:[...,k,n]; a 2-D operand broadcasts
    over the other's leading (batch) dims.  bias: [n] or [batch..., n]/[batch...,1,n].
    actgrad=Y: C = alpha*op(A)@op(B) * act'(Y), Y = the output of activation `act` ([..., m, n], contiguous)."""
    _chk(A), _chk(B)
    am, ak = (A.shape[-1], A.shape[-2]) if transA else (A.shape[-2], A.shape[-1])
    bk, bn = (B.shape[-1], B.shape[-2]) if transB else (B.shape[-2], B.shape[-1])
    if ak != bk:
        raise ValueError("matmul inner dims differ: %s vs %s" % (tuple(A.shape), tuple(B.shape)))
    la, lb = tuple(A.shape[:-2]), tuple(B.shape[:-2])
    ba, bb = int(np.prod(la)) if la else 1, int(np.prod(lb)) if lb else 1
    if ba != bb and ba != 1 and bb != 1:
        raise ValueError("matmul batch dims must match or be absent: %s vs %s" % (la, lb))
    batch = max(ba, bb)
    lead = la if ba >= bb else lb
    sA = A.shape[-2] * A.shape[-1] if (ba == batch and batch > 1) else 0
    sB = B.shape[-2] * B.shape[-1] if (bb == batch and batch > 1) else 0
    if out is None:
        out = _empty(lead + (am, bn), dtype=A.dtype, device=A.device)
    sBias = 0
    if actgrad is not None:
        _chk(actgrad)
        if bias is not None or beta != 0.0 or lower_out or tril_out or epilogue:
            raise ValueError("matmul: actgrad excludes bias, beta and the triangular epilogues")
        if tuple(actgrad.shape[-2:]) != (am, bn) or actgrad.numel() not in (am * bn, batch * am * bn):
            raise ValueError("matmul: actgrad operand %s does not match the result [%d,%d]" % (tuple(actgrad.shape), am, bn))
        if not actgrad.is_contiguous() or actgrad.dtype != A.dtype:
            raise ValueError("matmul: actgrad operand must be contiguous and of the operands' dtype")
        bias, epilogue = actgrad, MM_ACTGRAD
        sBias = am * bn if (actgrad.numel() == batch * am * bn and batch > 1) else 0
    elif bias is not None:
        _chk(bias)
        if bias.numel() == bn:
            sBias = 0
        elif bias.numel() == batch * bn:
            sBias = bn
        else:
            raise ValueError("bias shape %s does not match [%d] or [%d,%d]" % (tuple(bias.shape), bn, batch, bn))
    ws = workspace(A.dtype, A.device, 1 << 22)
    _lib.lib().call("hb_matmul" + _suf(A), _p(A), _p(B), _p(out), batch, am, bn, ak, A.shape[-1], B.shape[-1], bn,
                    sA, sB, am * bn, int(transA), int(transB), float(alpha), float(beta), _p(bias), sBias, ACT[act],
                    (MM_LOWER_OUT if lower_out else 0) | (MM_TRIL_OUT if tril_out else 0) | int(epilogue), _p(ws), ws.numel(), stream())
    return out


def matmul_colsum(A, B, out=None, colsum=None):
    """(A^T B, column sums of B) for A [K, M], B [K, N]: the weight and bias gradients of a MatBias layer from one pass over
    the incoming gradient (hb_matmul_colsum)."""
    _chk(A), _chk(B)
    K, M = A.shape[-2], A.shape[-1]
    N = B.shape[-1]
    assert A.dim() == 2 and B.dim() == 2 and B.shape[0] == K
    if out is None:
        out = _empty((M, N), dtype=A.dtype, device=A.device)
    if colsum is None:
        colsum = _empty((N,), dtype=A.dtype, device=A.device)
    assert colsum.numel() == N and colsum.is_contiguous()
    ws = workspace(A.dtype, A.device, 1 << 22)
    _lib.lib().call("hb_matmul_colsum" + _suf(A), _p(A), _p(B), _p(out), _p(colsum), M, N, K, M, N, N, _p(ws), ws.numel(),
                    stream())
    return out, colsum


ACTS = {"none": 0, "sigmoid": 1, "relu": 2, "tanh": 3}


def mlp2_sample_supported(n, din, hid, nout, rng_lanes=0, has_u=True):
    return bool(_lib.lib().raw("hb_mlp2_sample_supported")(n, din, hid, nout, int(rng_lanes), int(bool(has_u))))


def mlp2_sample_ws_elems(n, din, hid):
    return int(_lib.lib().raw("hb_mlp2_sample_ws_elems")(n, din, hid))


def mlp2_sample_ws(n, din, hid, device):
    return _empty(mlp2_sample_ws_elems(n, din, hid), dtype=torch.float32, device=device)


def mlp2_sample_fwd(y, w0, b0, w1, b1, act, u_in=None, rng=None, out=None, ws=None):
    """(x, kl, u, o) of the fused two-layer encoder + LOCAL diagonal sample + MC-KL (hb_mlp2_sample_fwd_f32)."""
    _chk(y), _chk(w0), _chk(w1)
    n, din = y.shape
    hid = w0.shape[1]
    if out is None:
        out = (_empty((n, 16), dtype=y.dtype, device=y.device), _empty(1, dtype=y.dtype, device=y.device),
               _empty((n, 16), dtype=y.dtype, device=y.device), _empty((n, 32), dtype=y.dtype, device=y.device))
    x, kl, u, o = out
    if ws is None:
        ws = mlp2_sample_ws(n, din, hid, y.device)
    _lib.lib().call("hb_mlp2_sample_fwd_f32", _p(y), _p(w0), _p(b0), _p(w1), _p(b1), ACTS[act], _p(u_in),
                    _p(rng.state) if (rng is not None and u_in is None) else None, rng.nlanes if rng is not None else 0,
                    _p(x), _p(kl), _p(u), _p(o), n, din, hid, _p(ws), stream())
    return x, kl, u, o


def mlp2_sample_bwd(y, w0, b0, w1, act, o, u, x, xbar, klbar, out=None, ws=None):
    """(dw0, db0, dw1, db1) of the fused encoder + sampler (hb_mlp2_sample_bwd_f32): h recomputed from y."""
    n, din = y.shape
    hid = w0.shape[1]
    if out is None:
        out = (_empty_like(w0), _empty(hid, dtype=y.dtype, device=y.device), _empty_like(w1), _empty(32, dtype=y.dtype, device=y.device))
    dw0, db0, dw1, db1 = out
    if ws is None:
        ws = mlp2_sample_ws(n, din, hid, y.device)
    _lib.lib().call("hb_mlp2_sample_bwd_f32", _p(y), _p(w0), _p(b0), _p(w1), ACTS[act], _p(o), _p(u), _p(x), _p(xbar), _p(klbar),
                    _p(dw0), _p(db0), _p(dw1), _p(db1), n, din, hid, _p(ws), stream())
    return dw0, db0, dw1, db1


def cholesky(A, out=None, info=None):
    """L = chol(A) (lower), batched over leading dims.  Returns (L, info[B] int32 device tensor)."""
    _chk(A)
    M = A.shape[-1]
    assert A.shape[-2] == M
    B = A.numel() // max(M * M, 1)
    if out is None:
        out = _empty_like(A)
    if info is None:
        info = _empty(max(B, 1), dtype=torch.int32, device=A.device)
    if out.data_ptr() == A.data_ptr() and A.numel():
        # the C entry point forbids aliasing; in-place requests go through a copy (eager use only)
        tmp = _empty_like(A)
        _lib.lib().call("hb_cholesky" + _suf(A), _p(A), _p(tmp), B, M, _p(info), stream())
        ewise("COPY", [tmp], out=out)
        return out, info
    _lib.lib().call("hb_cholesky" + _suf(A), _p(A), _p(out), B, M, _p(info), stream())
    return out, info


PREC_NATIVE, PREC_BF16X3 = 0, 1

_chol_ws_cache = {}


def debug_set(key, value):
    """Diagnostic switch of the native library (hb_debug_set, include/henbun_hip.h): tests and tools only."""
    _lib.lib().call("hb_debug_set", key.encode(), int(value))


def debug_clear():
    _lib.lib().call("hb_debug_clear")


def cholesky_ws_elems(B, M, dtype):
    return int(_lib.lib().raw("hb_cholesky_inverse_ws_elems")(B, M, 4 if dtype == torch.float32 else 8))


def cholesky_workspace(dtype, device, B, M):
    """ZERO-FILLED workspace of hb_cholesky_inverse for (B, M) on the current stream: the persistent launch keeps its
    sync words there and leaves them zero after every call (include/henbun_hip.h, workspace contract)."""
    key = (dtype, str(device), torch.cuda.current_stream().cuda_stream, B, M)
    w = _chol_ws_cache.get(key)
    if w is None:
        w = torch.zeros(max(cholesky_ws_elems(B, M, dtype), 1), dtype=dtype, device=device)
        _chol_ws_cache[key] = w
    return w


def cholesky_inverse(A, out=None, inv=None, info=None, ws=None, frag=None, frag_bf16x3=False):
    """(L, W, info): L = chol(A) and W = L^-1 from one fused launch sequence (batched over leading dims).
    `frag` (2*B*M*M elements, M % 32 == 0): receives the fragment-major copies of W and W^T (see the header)."""
    _chk(A)
    M = A.shape[-1]
    assert A.shape[-2] == M
    B = A.numel() // max(M * M, 1)
    if out is None:
        out = _empty_like(A)
    if inv is None:
        inv = _empty_like(A)
    if info is None:
        info = _empty(max(B, 1), dtype=torch.int32, device=A.device)
    if ws is None:
        ws = cholesky_workspace(A.dtype, A.device, B, M)
    assert ws.numel() >= cholesky_ws_elems(B, M, A.dtype), "hb_cholesky_inverse: workspace too small (cholesky_ws_elems)"
    if frag is not None:
        assert frag.numel() >= (5 if frag_bf16x3 else 2) * B * M * M and M % 32 == 0 and frag.dtype == A.dtype
    _lib.lib().call("hb_cholesky_inverse" + _suf(A), _p(A), _p(out), _p(inv), B, M, _p(info), _p(ws), _p(frag),
                    int(bool(frag_bf16x3 and frag is not None)), stream())
    return out, inv, info


def cholesky_persistent_shape(B, M, dtype):
    """hb_cholesky_inverse takes its one-launch persistent form for this (B, M, dtype)."""
    return bool(_lib.lib().raw("hb_cholesky_persistent_shape")(B, M, 4 if dtype == torch.float32 else 8))


def gram_cholesky_inverse(X, ell, diag_add, kind=KERN_RBF, out=None, inv=None, info=None, ws=None, frag=None, frag_bf16x3=False):
    """(L, W, info) of K(X, X) + diag_add I without K being written: hb_gram_cholesky_inverse_f32 (the persistent
    Cholesky synthesises its tiles from the points).  X: [M, d] or [B, M, d]; ell as in gram_fwd."""
    _chk(X), _chk(ell)
    assert X.dtype == torch.float32
    BX, M, d = _batch_view(X)
    B = BX
    if ell.dim() >= 2 and ell.shape[0] > 1:
        B = max(B, ell.shape[0])
    assert BX in (1, B)
    sX = M * d if (BX == B and B > 1) else 0
    sEll, dl = _ell_layout(ell, B, d)
    batched = X.dim() > 2 or sEll != 0
    shape = ((B,) if batched else ()) + (M, M)
    if out is None:
        out = _empty(shape, dtype=X.dtype, device=X.device)
    if inv is None:
        inv = _empty(shape, dtype=X.dtype, device=X.device)
    if info is None:
        info = _empty(max(B, 1), dtype=torch.int32, device=X.device)
    if ws is None:
        ws = cholesky_workspace(X.dtype, X.device, B, M)
    assert ws.numel() >= cholesky_ws_elems(B, M, X.dtype)
    if frag is not None:
        assert frag.numel() >= (5 if frag_bf16x3 else 2) * B * M * M and frag.dtype == X.dtype
    _lib.lib().call("hb_gram_cholesky_inverse_f32", kind, _p(X), sX, _p(ell), sEll, dl, d, float(diag_add), _p(out), _p(inv), B, M,
                    _p(info), _p(ws), _p(frag), int(bool(frag_bf16x3 and frag is not None)), stream())
    return out, inv, info


def trinv(L, out=None):
    """W = L^{-1} for lower-triangular L, batched."""
    _chk(L)
    M = L.shape[-1]
    B = L.numel() // max(M * M, 1)
    if out is None:
        out = _empty_like(L)
    ws = workspace(L.dtype, L.device, max(B * M * M, 1))
    _lib.lib().call("hb_trinv" + _suf(L), _p(L), _p(out), B, M, _p(ws), stream())
    return out


# ---- K5/K6 fused sparse GP -----------------------------------------------------
def _sgp_dims(x, z, u):
    E = z.shape[0] if z.dim() == 3 else 1
    M, d = z.shape[-2], z.shape[-1]
    n = x.shape[-2]
    P = u.shape[-2]
    sx = n * d if (x.dim() == 3 and x.shape[0] == E and E > 1) else 0
    return E, n, M, d, P, sx


def sgp_strip_path(E, n, M, d, P, prec=PREC_NATIVE):
    """True when sgp_fwd / sgp_bwd (given wfrag) run in column-strip form and may exchange A / Kbar fragment-major."""
    return bool(_lib.lib().raw("hb_sgp_strip_path")(E, n, M, d, P, int(prec)))


def sgp_frag_elems(E, n, M, prec=PREC_NATIVE):
    """fp32 elements of a fragment-major [E, M, n] operand buffer (columns padded to whole strips of 32); the bf16x3
    form holds three bf16 planes instead of one fp32 image: 1.5 times the bytes."""
    base = E * M * 32 * ((n + 31) // 32)
    return base * 3 // 2 if prec == PREC_BF16X3 else base


def sgp_head_units(x, z, u, prec, has_wfrag, draw, rng):
    """Number of partial-sum units hb_sgp_fwd_gauss leaves for this call (0: the likelihood head cannot ride in it)."""
    E, n, M, d, P, _ = _sgp_dims(x, z, u)
    if x.dtype != torch.float32:
        return 0
    return int(_lib.lib().raw("hb_sgp_head_units")(E, n, M, d, P, int(prec), int(bool(has_wfrag)), int(bool(draw)),
                                                  rng.nlanes if rng is not None else 0))


def matmul_gram_vjp_ok(M, K, batch, d, dtype):
    """The square product [batch, M, K] x [batch, K, M] can carry the Gram VJP of its result (hb_matmul_gram_vjp_ok)."""
    if dtype != torch.float32:
        return False
    return bool(_lib.lib().raw("hb_matmul_gram_vjp_ok")(int(M), int(K), int(batch), int(d)))


def matmul_gram_vjp(a, b, out, transA, transB, X, sX, ell, sEll, dl, d, xbar, ell_partial, part, counters):
    """out = op(a) op(b) (batched square product) and, from the same launch, the one-pass symmetric Gram VJP of `out` as
    Kbar: xbar and the lengthscale row partials (hb_matmul_gram_vjp_f32)."""
    M = out.shape[-1]
    batch = out.numel() // (M * M)
    K = a.shape[-2] if transA else a.shape[-1]
    sA = a.stride(0) if a.dim() == 3 and a.shape[0] > 1 else 0
    sB = b.stride(0) if b.dim() == 3 and b.shape[0] > 1 else 0
    _lib.lib().call("hb_matmul_gram_vjp_f32", _p(a), _p(b), _p(out), batch, M, K, a.stride(-2), b.stride(-2), out.stride(-2),
                    sA, sB, M * M, int(bool(transA)), int(bool(transB)), _p(X), int(sX), _p(ell), int(sEll), int(dl), int(d),
                    _p(xbar), _p(ell_partial), _p(part), _p(counters), stream())


def gram_ell_fold(partial, rows, d, dl, groups, ellbar):
    """ellbar from the lengthscale row partials of a Gram VJP (hb_gram_ell_fold; chain-aware)."""
    _lib.lib().call("hb_gram_ell_fold" + _suf(partial), _p(partial), int(rows), int(d), int(dl), int(groups), _p(ellbar), stream())


def matmul_gauss_units(n, K, N, dtype):
    """Partial-sum units of hb_matmul_gauss for an [n, K] x [K, N] MatBias layer feeding a Gaussian head (0: not this shape)."""
    if dtype != torch.float32:
        return 0
    return int(_lib.lib().raw("hb_matmul_gauss_units")(int(n), int(K), int(N)))


def matmul_gauss(x, w, bias, head):
    """The Gaussian likelihood head of x @ w + bias in the product's own launch (hb_matmul_gauss_f32): head = dict(y, scale,
    var, post, dmu, fbar, part, units); writes head['dmu'] (and head['fbar']) and the partial sums head['part']."""
    for t in (x, w):
        _chk(t)
    n, K = x.shape
    N = w.shape[1]
    _lib.lib().call("hb_matmul_gauss_f32", _p(x), x.stride(0), _p(w), w.stride(0), _p(bias), _p(head["y"]), _p(head.get("scale")),
                    _p(head["var"]), float(head.get("post") or 0.0), _p(head["dmu"]), _p(head.get("fbar")), _p(head["part"]),
                    int(head["units"]), int(n), int(K), int(N), stream())


def sgp_rider_supported(x, z, u, prec, has_wfrag, draw, rng):
    """1 when this forward call can be recorded (sgp_rider_begin) and start inside the launch of the persistent
    factorisation that produces its W (hb_sgp_rider_supported)."""
    E, n, M, d, P, _ = _sgp_dims(x, z, u)
    if x.dtype != torch.float32:
        return False
    return bool(_lib.lib().raw("hb_sgp_rider_supported")(E, n, M, d, P, int(prec), int(bool(has_wfrag)), int(bool(draw)),
                                                         rng.nlanes if rng is not None else 0))


def sgp_rider_begin():
    """The next sgp_fwd call of this thread is recorded instead of launched; the next cholesky_inverse /
    gram_cholesky_inverse launches it inside its own grid (include/henbun_hip.h: early-start forward)."""
    _lib.lib().call("hb_sgp_rider_begin")


def sgp_rider_flush():
    """Launch a recorded forward that no factorisation picked up (a no-op otherwise)."""
    _lib.lib().call("hb_sgp_rider_flush", stream())


def sgp_rider_pending():
    return int(_lib.lib().raw("hb_sgp_rider_pending")())


def gauss_ll_fold(part, nb, ll, ds, dv):
    """(ll, dscale, dvar) from the partial sums part[3][nb] of a head whose per-point part ran elsewhere (hb_gauss_ll_fold)."""
    _lib.lib().call("hb_gauss_ll_fold" + _suf(part), _p(part), int(nb), _p(ll), _p(ds), _p(dv), stream())


def sgp_fwd(x, z, ell, W, u, eps_in=None, rng=None, mode=SGP_DIAGONAL, out=None, wfrag=None, prec=PREC_NATIVE,
            a_frag=None, skip_a=False, head=None):
    """Returns (f[E?,P,n], A[E?,M,n], v[E?,n], eps[E?,n]).  `wfrag`: cholesky_inverse's fragment-major copies of W.
    `head`: dict(y, scale, var, post, dmu, fbar, part, units) -- the Gaussian likelihood head's per-point part in the same
    launch (hb_sgp_fwd_gauss; units = sgp_head_units(...) > 0)."""
    for t in (x, z, ell, W, u):
        _chk(t)
    E, n, M, d, P, sx = _sgp_dims(x, z, u)
    lead = (E,) if z.dim() == 3 else ()
    dev, dt = x.device, x.dtype
    if out is None:
        f = _empty(lead + (P, n), dtype=dt, device=dev)
        A = _empty(lead + (M, n), dtype=dt, device=dev)
        v = _empty(lead + (n,), dtype=dt, device=dev)
        eps = _empty(lead + (n,), dtype=dt, device=dev)
    else:
        f, A, v, eps = out
    dl = ell.numel() // E
    rp, rl = _rng_args(rng)
    ws = workspace(dt, dev, int(_lib.lib().raw("hb_sgp_ws_elems")(E, n, M, d, P)))
    # a_frag: also (skip_a: only) leave A fragment-major for sgp_bwd (column-strip form, see the header)
    if head is not None:
        _lib.lib().call("hb_sgp_fwd_gauss_f32", KERN_RBF, mode, _p(x), sx, _p(z), _p(ell), dl, _p(W), _p(wfrag), int(prec), _p(u),
                        _p(eps_in), rp, rl, _p(eps), None if (skip_a and a_frag is not None) else _p(A), _p(a_frag), _p(f), _p(v),
                        E, n, M, d, P, _p(ws), _p(head["y"]), _p(head.get("scale")), _p(head["var"]), float(head.get("post") or 0.0),
                        _p(head["dmu"]), _p(head.get("fbar")), _p(head["part"]), int(head["units"]), stream())
        return f, A, v, eps
    _lib.lib().call("hb_sgp_fwd" + _suf(x), KERN_RBF, mode, _p(x), sx, _p(z), _p(ell), dl, _p(W), _p(wfrag), int(prec), _p(u), _p(eps_in),
                    rp, rl, _p(eps), None if (skip_a and a_frag is not None) else _p(A), _p(a_frag), _p(f), _p(v), E, n,
                    M, d, P, _p(ws), stream())
    return f, A, v, eps


def sgp_A(x, z, ell, W, out=None, wfrag=None, prec=PREC_NATIVE):
    """A = W K(z,x): the M^2 n contraction alone (hb_sgp_A)."""
    E = z.shape[0] if z.dim() == 3 else 1
    M, d = z.shape[-2], z.shape[-1]
    n = x.shape[-2]
    sx = n * d if (x.dim() == 3 and x.shape[0] == E and E > 1) else 0
    if out is None:
        out = _empty(((E,) if z.dim() == 3 else ()) + (M, n), dtype=x.dtype, device=x.device)
    _lib.lib().call("hb_sgp_A" + _suf(x), KERN_RBF, _p(x), sx, _p(z), _p(ell), ell.numel() // E, _p(W), _p(wfrag), int(prec),
                    _p(out), E, n, M, d, stream())
    return out


def sgp_bwd(x, z, ell, W, u, eps, A, v, fbar, mode=SGP_DIAGONAL, need_xbar=False, out=None, wfrag=None,
            prec=PREC_NATIVE, a_frag=None, kbar_frag=None):
    """Returns (Lbar, ubar, zbar, ellbar, xbar|None)."""
    E, n, M, d, P, sx = _sgp_dims(x, z, u)
    dev, dt = x.device, x.dtype
    dl = ell.numel() // E
    if out is None:
        Kbar = _empty_like(A) if a_frag is None else None
        Lbar = _empty_like(W)
        ubar = _empty_like(u)
        zbar = _empty_like(z)
        ellbar = _empty_like(ell)
        xbar = _empty((E, n, d), dtype=dt, device=dev) if need_xbar else None
    else:
        Kbar, Lbar, ubar, zbar, ellbar, xbar = out
    wse = _lib.lib().raw("hb_sgp_ws_elems")(E, n, M, d, P)
    ws = workspace(dt, dev, wse)
    if a_frag is not None and kbar_frag is None:
        kbar_frag = _empty(a_frag.numel(), dtype=dt, device=dev)
    _lib.lib().call("hb_sgp_bwd" + _suf(x), KERN_RBF, mode, _p(x), sx, _p(z), _p(ell), dl, _p(W), _p(wfrag), int(prec), _p(u), _p(eps),
                    _p(A) if a_frag is None else None, _p(a_frag), _p(v), _p(fbar), _p(Kbar) if a_frag is None else None,
                    _p(kbar_frag), _p(Lbar), _p(ubar), _p(zbar), _p(ellbar), _p(xbar), E, n, M, d, P, _p(ws), stream())
    return Lbar, ubar, zbar, ellbar, xbar


# ---- K9 Adam ---------------------------------------------------------------------
def adam_step(theta, g, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, gscale=1.0, tick=True, info=None, dpflag=None,
              fail=None):
    """In-place TF-1 Adam on flat buffers; `t` is a 1-element int64 device tensor (advanced when `tick`).
    `info` (int32 status words of this step's factorisations), `dpflag` (one element of theta's dtype: the
    all-reduced failure flag) and `fail` (int64[2], sticky) make the update a no-op when a factorisation failed."""
    assert t.dtype == torch.int64
    n_info = 0
    if info is not None:
        assert info.dtype == torch.int32 and info.is_contiguous()
        n_info = info.numel()
    if fail is not None:
        assert fail.dtype == torch.int64 and fail.numel() == 2
    if dpflag is not None:
        assert dpflag.dtype == theta.dtype and dpflag.numel() == 1
    _lib.lib().call("hb_adam_step" + _suf(theta), _p(theta), _p(g), _p(m), _p(v), theta.numel(), float(lr), float(b1),
                    float(b2), float(eps), float(gscale), _p(t), int(bool(tick)), _p(info) if n_info else None, n_info,
                    _p(dpflag) if dpflag is not None else None, _p(fail) if fail is not None else None, stream())


# ---- data-parallel exchange ---------------------------------------------------------
def allreduce_sum(flat, comm_handle):
    """In-place RCCL all-reduce (sum) of a contiguous device buffer on the current stream."""
    _chk(flat)
    _lib.lib().call("hb_allreduce_sum" + _suf(flat), _p(flat), flat.numel(), comm_handle, stream())


def dp_pack(tail, objective, info):
    """tail[0] = objective, tail[1] = any(info != 0)  (hb_dp_pack)."""
    assert tail.numel() == 2
    n_info = 0 if info is None else info.numel()
    _lib.lib().call("hb_dp_pack" + _suf(tail), _p(tail), _p(objective), _p(info) if n_info else None, n_info, stream())


# ---- hipGraph capture ---------------------------------------------------------------
class CapturedGraph:
    """A replayable hipGraph of whatever was launched between begin() and end()."""

    def __init__(self):
        self._exec = c_void_p(None)

    def begin(self):
        _lib.lib().call("hb_graph_begin_capture", stream())
        set_capturing(True)

    def end(self):
        set_capturing(False)
        _lib.lib().call("hb_graph_end_capture", stream(), ctypes.byref(self._exec))

    def launch(self):
        _lib.lib().call("hb_graph_launch", self._exec, stream())

    def __del__(self):
        try:
            if self._exec:
                _lib.lib().raw("hb_graph_destroy")(self._exec)
        except Exception:
            pass


def device_info():
    buf = ctypes.create_string_buffer(256)
    cus = ctypes.c_int(0)
    _lib.lib().call("hb_device_info", buf, 256, ctypes.byref(cus))
    return buf.value.decode(), cus.value
