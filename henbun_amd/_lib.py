"""ctypes binding of libhenbun_hip.so (declared in include/henbun_hip.h).

This is the module that stands where the reference has `tf_wraps.py` + the
TensorFlow runtime (reference Henbun/tf_wraps.py:26-48, model.py:265-266): the
only way numerics happen in henbun_amd.  There is NO fallback: if the shared
library is missing the import-time loader raises, and a call that returns a
non-zero status raises `HipBackendError`.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_long, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhenbun_hip.so")


class HipBackendError(RuntimeError):
    """A libhenbun_hip.so entry point returned a non-zero status."""


P, L, I, D, U64 = c_void_p, c_long, c_int, c_double, c_uint64

# name -> argument ctypes (return type is int unless listed in _RESTYPES)
_SIGS = {
    "hb_version": [],
    "hb_debug_set": [c_char_p, L],
    "hb_debug_clear": [],
    "hb_cholesky_inverse_ws_elems": [L, L, I],
    "hb_cholesky_persistent_shape": [L, L, I],
    "hb_mlp2_sample_supported": [L, L, L, L, L, I],
    "hb_mlp2_sample_ws_elems": [L, L, L],
    "hb_mlp2_sample_fwd_f32": [P, P, P, P, P, I, P, P, L, P, P, P, P, L, L, L, P, P],
    "hb_mlp2_sample_bwd_f32": [P, P, P, P, I, P, P, P, P, P, P, P, P, P, L, L, L, P, P],
    "hb_gram_cholesky_inverse_f32": [I, P, L, P, L, L, L, D, P, P, L, L, P, P, P, I, P],
    "hb_last_error_string": [],
    "hb_device_info": [P, I, P],
    "hb_graph_begin_capture": [P],
    "hb_graph_end_capture": [P, P],
    "hb_graph_launch": [P, P],
    "hb_graph_destroy": [P],
    "hb_rng_init": [P, L, U64, U64, P],
    "hb_rng_randint": [P, L, P, L, L, L, P],
    "hb_sgp_ws_elems": [L, L, L, L, L],
    "hb_sgp_strip_path": [L, L, L, L, L, I],
    "hb_sgp_head_units": [L, L, L, L, L, I, I, I, L],
    "hb_sgp_rider_supported": [L, L, L, L, L, I, I, I, L],
    "hb_matmul_gauss_units": [L, L, L],
    "hb_matmul_gram_vjp_ok": [L, L, L, L],
    "hb_matmul_gram_vjp_ws_elems": [L, L, L],
    "hb_matmul_gram_vjp_f32": [P, P, P, L, L, L, L, L, L, L, L, L, I, I, P, L, P, L, L, L, P, P, P, P, P],
    "hb_fullrank_one_launch_shape": [L, L],
    "hb_matmul_gauss_f32": [P, L, P, L, P, P, P, P, D, P, P, P, L, L, L, L, P],
    "hb_sgp_rider_begin": [],
    "hb_sgp_rider_pending": [],
    "hb_sgp_rider_flush": [P],
    "hb_sgp_fwd_gauss_f32": [I, I, P, L, P, P, L, P, P, I, P, P, P, L, P, P, P, P, P, L, L, L, L, L, P, P, P, P, D, P, P, P, L, P],
    "hb_ewise_prog_image_bytes": [],
    "hb_ewise_prog_build": [I, P, P, I, P, P, I, P, P, P, I, P, P, P, P],
    "hb_ewise_jit_available": [],
    "hb_ewise_jit_run": [P, P],
    "hb_ewise_jit_destroy": [P],
    "hb_side_push_gather_draw_f32": [I, P, P, P, L, P, L, L, L, P, P, L, P, P],
    "hb_side_push_diag_fwd_f32": [P, P, P, P, L, P, P, P, L, L, L, L, P, P],
    "hb_side_push_diag_bwd_f32": [P, P, P, P, P, P, P, L, L, L, L, P],
    "hb_side_pending": [],
    "hb_side_flush": [P],
    "hb_side_discard": [],
    "hb_chain_begin": [],
    "hb_chain_end": [P],
    "hb_chain_discard": [],
    "hb_chain_source": [P, L],
    "hb_chain_compile_dry": [],
    "hb_comm_available": [],
    "hb_comm_unique_id": [P],
    "hb_comm_init": [P, I, I, P],
    "hb_comm_destroy": [P],
}
_RESTYPES = {"hb_last_error_string": c_char_p, "hb_sgp_ws_elems": c_long, "hb_ewise_prog_image_bytes": c_long,
             "hb_sgp_head_units": c_long, "hb_matmul_gauss_units": c_long, "hb_matmul_gram_vjp_ws_elems": c_long, "hb_cholesky_inverse_ws_elems": c_long, "hb_mlp2_sample_ws_elems": c_long}

# entry points that exist as _f32 and _f64
_TYPED = {
    "hb_ewise": [I, I, P, P, I, P, I, P, P, P],
    "hb_ewise_prog": [I, P, P, I, P, P, I, P, P, P, I, P, P],
    "hb_ewise_prog_run": [P, L, I, P],
    "hb_ewise_jit_build": [I, P, P, I, P, P, I, P, P, P, I, P, P, P, P, P, L],
    "hb_ewise_colprog_build": [I, P, P, I, P, P, I, P, P, P, L, L, P, P, L],
    "hb_gauss_ll": [P, P, P, P, L, P, P, P, P, P, L, P],
    "hb_gauss_ll_post": [P, P, P, P, L, P, P, P, P, D, P, P, L, P],
    "hb_gauss_ll_fold": [P, L, P, P, P, P],
    "hb_reduce": [I, P, P, L, L, L, P, L, P],
    "hb_copy_nd": [P, P, P, P, I, P, P],
    "hb_fill": [P, L, D, P],
    "hb_gather_rows": [P, L, L, P, P, L, P, P, P],
    "hb_gather_rows_multi": [I, P, P, P, L, P, P, L, P, P],
    "hb_gather_rows_multi_draw": [I, P, P, P, L, P, L, L, L, P, P, L, P, P],
    "hb_matutil": [P, P, L, L, L, I, L, L, D, P],
    "hb_rng_normal": [P, L, P, L, P],
    "hb_diag_sample_kl_fwd": [P, P, P, P, L, P, P, P, L, L, L, L, P, P],
    "hb_diag_sample_kl_bwd": [P, P, P, P, P, P, P, L, L, L, L, P],
    "hb_fullrank_sample_kl_fwd": [P, P, P, P, L, P, P, P, L, L, I, P, P],
    "hb_fullrank_sample_kl_fwd1": [P, P, P, P, L, P, P, P, L, L, I, P, P, P],
    "hb_fullrank_sample_kl_bwd": [P, P, P, P, P, P, P, L, L, I, P],
    "hb_vec_to_tri": [P, P, L, L, P],
    "hb_tri_to_vec": [P, P, L, L, P],
    "hb_gram_fwd": [I, P, L, P, L, P, L, L, P, L, L, L, L, D, P],
    "hb_gram_bwd": [I, P, L, P, L, P, L, L, P, P, P, P, L, L, L, L, P, P],
    "hb_gram_ell_fold": [P, L, L, L, L, P, P],
    "hb_matmul": [P, P, P, L, L, L, L, L, L, L, L, L, L, I, I, D, D, P, L, I, I, P, L, P],
    "hb_matmul_colsum": [P, P, P, P, L, L, L, L, L, L, P, L, P],
    "hb_cholesky": [P, P, L, L, P, P],
    "hb_cholesky_inverse": [P, P, P, L, L, P, P, P, I, P],
    "hb_trinv": [P, P, L, L, P, P],
    "hb_sgp_A": [I, P, L, P, P, L, P, P, I, P, L, L, L, L, P],
    "hb_sgp_fwd": [I, I, P, L, P, P, L, P, P, I, P, P, P, L, P, P, P, P, P, L, L, L, L, L, P, P],
    "hb_sgp_bwd": [I, I, P, L, P, P, L, P, P, I, P, P, P, P, P, P, P, P, P, P, P, P, P, L, L, L, L, L, P, P],
    "hb_adam_step": [P, P, P, P, L, D, D, D, D, D, P, I, P, L, P, P, P],
    "hb_allreduce_sum": [P, L, P, P],
    "hb_dp_pack": [P, P, P, L, P],
}


def declared_symbols():
    """Every symbol include/henbun_hip.h declares (used by the ABI test)."""
    names = list(_SIGS)
    for base in _TYPED:
        names += [base + "_f32", base + "_f64"]
    return names


class _Lib:
    def __init__(self, path: str):
        if not os.path.exists(path):
            raise ImportError(
                "henbun_amd: %s is missing.  Build it with `python -m henbun_amd._build` "
                "(hipcc, gfx950).  There is no CPU fallback." % path
            )
        # PyTorch (device memory / streams) ships its own libamdhip64.so: it must be the HIP runtime of the process.
        # Loading this library first would pull in /opt/rocm's copy, and kernels launched through that second runtime
        # see no device ("no ROCm-capable device is detected" when henbun_amd was imported before torch).
        import torch  # noqa: F401

        self._dll = ctypes.CDLL(path)
        self._fns = {}
        for name, args in _SIGS.items():
            self._bind(name, args, _RESTYPES.get(name, c_int))
        for base, args in _TYPED.items():
            for suf in ("_f32", "_f64"):
                self._bind(base + suf, args, c_int)
        if self.raw("hb_version")() != 2:
            raise ImportError("henbun_amd: ABI version mismatch in " + path)

    def _bind(self, name, args, restype):
        fn = getattr(self._dll, name)
        fn.argtypes = args
        fn.restype = restype
        self._fns[name] = fn

    def raw(self, name):
        return self._fns[name]

    def last_error(self) -> str:
        s = self._fns["hb_last_error_string"]()
        return s.decode() if s else ""

    def call(self, name, *args):
        """Call an int-returning entry point; raise on a non-zero status."""
        rc = self._fns[name](*args)
        if rc != 0:
            raise HipBackendError("%s failed (status %d): %s" % (name, rc, self.last_error()))
        return 0


_lib = None


def lib() -> _Lib:
    """The loaded backend (loads on first use; raises ImportError if absent)."""
    global _lib
    if _lib is None:
        _lib = _Lib(LIB_PATH)
    return _lib
