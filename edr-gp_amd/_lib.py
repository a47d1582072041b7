"""ctypes binding of ``libedrgp_hip.so`` (C ABI declared in ``include/edrgp_hip.h``).

There is no CPU fallback: if the shared object is missing, or no gfx950 device
is visible, the calls below raise.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EDRGP_LIB", os.path.join(_HERE, "libedrgp_hip.so"))   # EDRGP_LIB: another build of the same library (A/B runs)

# status codes (include/edrgp_hip.h)
OK, ERR_INVALID, ERR_NOT_PD, ERR_HIP, ERR_STATE, ERR_COMM = 0, -1, -2, -3, -4, -5
KERN_CODES = {"RBF": 0, "Matern32": 1, "Matern52": 2}
MAX_PARTS = 4
MAX_DIM = 256

GET = dict(MODE=14, PHI=1, BVEC=2, LM=3, LB=4, A=5, WV=6, WINV=7, KMM=8, KNM=9, LMI=10, LBI=11, DLDKMM=12, SCALARS=13)
PROF = dict(SYRK=0, KBUILD=1, GRADGEMM=2, SMALL=3, CHOL=4)


class EdrgpError(RuntimeError):
    """Any non-zero status from the native library."""

    def __init__(self, status, message):
        super().__init__("edrgp status %d: %s" % (status, message))
        self.status = status


class NotPositiveDefiniteError(EdrgpError, np.linalg.LinAlgError):
    """Cholesky failed at every level of the jitchol jitter ladder."""


_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_float_p = ctypes.POINTER(ctypes.c_float)
_c_int_p = ctypes.POINTER(ctypes.c_int)
_c_i64_p = ctypes.POINTER(ctypes.c_int64)
_ctx_p = ctypes.c_void_p

# name -> argtypes  (restype is int unless listed in _RESTYPES)
PROTOTYPES = {
    "edrgp_last_error": [],
    "edrgp_version": [],
    "edrgp_device_count": [_c_int_p],
    "edrgp_ctx_create": [ctypes.c_int, ctypes.POINTER(_ctx_p)],
    "edrgp_ctx_destroy": [_ctx_p],
    "edrgp_sync": [_ctx_p],
    "edrgp_set_option": [_ctx_p, ctypes.c_char_p, ctypes.c_double],
    "edrgp_set_data": [_ctx_p, _c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int],
    "edrgp_set_data_projected": [_ctx_p, _c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int, _c_double_p,
                                 ctypes.c_int],
    "edrgp_set_inducing": [_ctx_p, _c_double_p, ctypes.c_int],
    "edrgp_set_kernel": [_ctx_p, ctypes.c_int, _c_int_p, _c_double_p, _c_double_p, ctypes.c_double],
    "edrgp_kernel_matrix": [_ctx_p, ctypes.c_int, _c_int_p, _c_double_p, _c_double_p, _c_double_p, ctypes.c_int64,
                            _c_double_p, ctypes.c_int64, ctypes.c_int, _c_double_p],
    "edrgp_sparse_lml": [_ctx_p, _c_double_p],
    "edrgp_sparse_lml_grad": [_ctx_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p],
    "edrgp_sparse_posterior": [_ctx_p, _c_double_p],
    "edrgp_dense_lml_grad": [_ctx_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p],
    "edrgp_dense_posterior": [_ctx_p, _c_double_p],
    "edrgp_predict": [_ctx_p, _c_double_p, ctypes.c_int64, _c_double_p, _c_double_p],
    "edrgp_predict_grad": [_ctx_p, _c_double_p, ctypes.c_int64, _c_double_p],
    "edrgp_predict_train": [_ctx_p, _c_double_p, _c_double_p],
    "edrgp_predict_train_gram": [_ctx_p, _c_double_p, _c_double_p, _c_double_p],
    "edrgp_ep_begin": [_ctx_p, _c_double_p],
    "edrgp_ep_sweep": [_ctx_p, _c_int_p, ctypes.c_double, ctypes.c_double, _c_double_p, _c_double_p],
    "edrgp_ep_finish": [_ctx_p, _c_double_p],
    "edrgp_ep_get_sites": [_ctx_p, _c_double_p, _c_double_p],
    "edrgp_ep_set_sites": [_ctx_p, _c_double_p, _c_double_p, ctypes.c_double],
    "edrgp_ep_lml_grad": [_ctx_p, _c_double_p, _c_double_p, _c_double_p],
    "edrgp_ep_posterior": [_ctx_p, _c_double_p],
    "edrgp_epdtc_begin": [_ctx_p, _c_double_p],
    "edrgp_epdtc_sweep": [_ctx_p, _c_int_p, ctypes.c_double, ctypes.c_double, _c_double_p, _c_double_p],
    "edrgp_epdtc_lml_grad": [_ctx_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p],
    "edrgp_epdtc_posterior": [_ctx_p, _c_double_p],
    "edrgp_sweep_scales": [_ctx_p, ctypes.c_int, _c_double_p, ctypes.c_double, ctypes.c_double, ctypes.c_int, _c_double_p,
                           _c_double_p, _c_double_p, _c_int_p],
    "edrgp_get": [_ctx_p, ctypes.c_int, _c_double_p, ctypes.c_int64],
    "edrgp_profile_reset": [_ctx_p],
    "edrgp_profile_get": [_ctx_p, ctypes.c_int, _c_double_p, _c_i64_p],
    "edrgp_comm_unique_id": [ctypes.c_void_p],
    "edrgp_comm_init": [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p],
    "edrgp_comm_init_host": [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p],
    "edrgp_comm_destroy": [_ctx_p],
    "edrgp_dbg_gemm": [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                       ctypes.c_double, _c_double_p, ctypes.c_int, _c_double_p, ctypes.c_int,
                       ctypes.c_double, _c_double_p, ctypes.c_int],
    "edrgp_dbg_gemm32": [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                         _c_float_p, ctypes.c_int, _c_float_p, ctypes.c_int, _c_float_p, ctypes.c_int],
    "edrgp_dbg_syrk32": [_ctx_p, _c_float_p, ctypes.c_int64, ctypes.c_int, _c_double_p],
    "edrgp_dbg_potrf": [_ctx_p, _c_double_p, ctypes.c_int, _c_int_p],
    "edrgp_dbg_potrf_inv": [_ctx_p, _c_double_p, ctypes.c_int, _c_int_p, _c_double_p],
    "edrgp_dbg_potrf_time": [_ctx_p, _c_double_p, ctypes.c_int, ctypes.c_int, _c_double_p],
    "edrgp_dbg_trtri": [_ctx_p, _c_double_p, _c_double_p, ctypes.c_int],
    "edrgp_dbg_syrk": [_ctx_p, _c_double_p, ctypes.c_int64, ctypes.c_int, _c_double_p],
    "edrgp_dbg_syrk_time": [_ctx_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_double_p, _c_double_p],
    "edrgp_dbg_mfma_layout": [_ctx_p, _c_double_p, _c_double_p, _c_double_p],
    "edrgp_calib": [_ctx_p, _c_double_p, _c_double_p],
}
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, _c_double_p, ctypes.c_int64)
_RESTYPES = {"edrgp_last_error": ctypes.c_char_p}

_lib = None


def load():
    """Load the shared library once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `python edr-gp_amd/build.py` (hipcc, gfx950). "
            "The MI355X path has no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, ctypes.c_int)
    _lib = lib
    return lib


def last_error():
    msg = load().edrgp_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(status):
    if status == OK:
        return
    msg = last_error()
    if status == ERR_NOT_PD:
        raise NotPositiveDefiniteError(status, msg)
    raise EdrgpError(status, msg)


def dptr(a):
    """double* of a C-contiguous float64 array (None -> NULL)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_c_double_p)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def device_count():
    n = ctypes.c_int(0)
    status = load().edrgp_device_count(ctypes.byref(n))
    if status != OK:
        return 0
    return n.value
