"""ctypes binding of ``libxmris_hip.so`` (C ABI declared in ``include/xmris_hip.h``).

The product path has NO CPU fallback: if the shared library is missing or a call fails, an
exception is raised.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C xmris_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
import threading

XM_C64 = 0
XM_C128 = 1

XM_FFT_INVERSE = 1
XM_FFT_ORTHO = 2
XM_FFT_SHIFT_IN = 4
XM_FFT_SHIFT_OUT = 8
XM_AMAX_VALUE_ONLY = 16
XM_AMAX_GLOBAL_KEY = 32

XM_ERR_INVALID_ARG = -1
XM_ERR_UNSUPPORTED_N = -2
XM_ERR_HIP = -3
XM_ERR_NO_DEVICE = -4

LIB_NAME = "libxmris_hip.so"
# (XMRIS_AMD_LIB: another build of the same library, for A/B runs of kernel variants on one box)
LIB_PATH = os.environ.get("XMRIS_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

_p = ctypes.c_void_p
_i = ctypes.c_int
_u = ctypes.c_uint
_l = ctypes.c_int64

# name -> (restype, argtypes); mirrors include/xmris_hip.h one-to-one
SIGNATURES = {
    "xm_version": (_i, []),
    "xm_last_error_string": (ctypes.c_char_p, []),
    "xm_clear_cache": (_i, []),
    "xm_last_kernel_string": (ctypes.c_char_p, []),
    "xm_fft_supported": (_i, [_i, _i]),
    "xm_plan_prepare": (_i, [_i, _i]),
    "xm_zero_fill": (_i, [_p, _p, _l, _i, _i, _i, _i, _p]),
    "xm_apodize": (_i, [_p, _p, _p, _l, _i, _i, _p]),
    "xm_fft1d_batched": (_i, [_p, _p, _l, _i, _u, _i, _p]),
    "xm_zf_apod": (_i, [_p, _l, _p, _p, _l, _i, _i, _i, _i, _i, _p]),
    "xm_roll": (_i, [_p, _p, _l, _i, _i, _i, _p]),
    "xm_phase_apply": (_i, [_p, _p, _p, _l, _i, _i, _p]),
    "xm_absmax_rows": (_i, [_p, _l, _i, _p, _p, _i, _p]),
    "xm_argmax_reduce": (_i, [_p, _p, _l, _i, _p, _p, _i, _p]),
    "xm_baseline_als_workspace_bytes": (_l, [_l, _i]),
    "xm_baseline_als": (_i, [_p, _i, _l, _i, ctypes.c_double, ctypes.c_double, _i, _p, _p, _l, _i, _p]),
    "xm_gather_row_c128": (_i, [_p, _l, _i, _p, _i, _p, _i, _p]),
    "xm_pipeline_fused": (_i, [_p, _l, _p, _p, _p, _l, _i, _i, _i, _u, _p, _p, _i, _p]),
    "xm_pipeline_fused_ramp": (_i, [_p, _l, _p, _p, ctypes.c_double, ctypes.c_double, _l, _i, _i, _i, _u, _p, _p, _i, _p]),
    "xm_pipeline_ramp_native": (_i, [_p, _l, _i, _i, _i, _u, _i]),
    "xm_pipeline_key_native": (_i, [_p, _l, _i, _i, _i, _u, _i]),
    "xm_row_l1": (_i, [_p, _l, _p, _l, _i, _i, _i, _p, _p, _i, _p]),
    "xm_argmax_key_take": (_i, [_p, _i, _p, _p, _p, _l, _i, _p, _i, _p]),
    "xm_guess_supported": (_i, [_p, _l, _i, _i, _i, _u, _i]),
    "xm_guess_rows": (_i, [_p, _l, _p, _l, _i, _i, _i, _u, _p, _p, _i, _p]),
    "xm_guess_refine": (_i, [_p, _l, _p, _l, _i, _i, _u, _p, _p, ctypes.c_float, _p, _p, _p, _p, _i, _p]),
    "xm_phase_table": (_i, [_p, _i, ctypes.c_double, ctypes.c_double, ctypes.c_double, _p, _i]),
    "xm_solver_create": (_p, [_p, _p, _i, ctypes.c_double, _i, _i, _i]),
    "xm_solver_destroy": (None, [_p]),
    "xm_solver_score": (ctypes.c_double, [_p, _p, _i]),
    "xm_solver_score_batch": (None, [_p, _p, _i, _i, _p]),
    "xm_solver_fg": (_i, [_p, _p, _i, _p, _p, _p, _p]),
    "xm_solver_pool_backups": (ctypes.c_long, []),
    "xm_solver_nfev": (ctypes.c_long, [_p]),
    "xm_solver_set_threads": (_i, [_p, _i]),
    "xm_solver_de": (_i, [_p, _i, _u, ctypes.c_double, _i, _p, _p, _p, _p]),
    "xm_search_supported": (_i, [_i, _i, ctypes.c_double]),
    "xm_search_launch": (_i, [_p, _i, ctypes.c_double, ctypes.c_double, ctypes.c_double, _i, _i, _u, ctypes.c_double, _i,
                              ctypes.c_uint64, _p, _p]),
    "xm_search_eval": (_i, [_p, _i, ctypes.c_double, ctypes.c_double, ctypes.c_double, _i, _i, _p, _i, _p, _p]),
    "xm_hostsearch_submit": (_i, [_p, _i, _p, _i, _i, _i, _i, _u, ctypes.c_double, _i, _i, ctypes.c_uint64, _p]),
    "xm_hostsearch_set_workers": (_i, [_i]),
    "xm_stream_create": (_i, [_p, _i, _i]),
    "xm_stream_destroy": (_i, [_p]),
    "xm_stream_cus": (_i, [_p]),
    "xm_atomic_load_acquire_i64": (_l, [_p]),
    "xm_atomic_store_release_i64": (None, [_p, _l]),
    "xm_atomic_wait_all_ge_i64": (_i, [_p, _i, _i, _l, _i]),
}


class XmrisHipError(RuntimeError):
    """A call into libxmris_hip.so returned a negative status."""

    def __init__(self, func: str, code: int, detail: str):
        self.func, self.code, self.detail = func, code, detail
        super().__init__(f"{func} failed with status {code}: {detail}")


class UnsupportedLengthError(XmrisHipError, ValueError):
    """The transform length has no in-LDS plan on the device (status XM_ERR_UNSUPPORTED_N)."""


_lock = threading.Lock()
_lib = None


def load() -> ctypes.CDLL:
    """Load the shared library once; raise loudly when it is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing. The xmris_amd hot path has no CPU fallback: build the HIP "
                f"library first (`python -c \"import __graft_entry__ as g; g.build()\"` or "
                f"`make -C xmris_amd/csrc`)."
            )
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI and the header drifted apart
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(func: str, status: int) -> None:
    if status == 0:
        return
    detail = load().xm_last_error_string()
    detail = detail.decode("utf-8", "replace") if detail else ""
    if status == XM_ERR_UNSUPPORTED_N:
        raise UnsupportedLengthError(func, status, detail)
    raise XmrisHipError(func, status, detail)


def call(func: str, *args) -> None:
    check(func, getattr(load(), func)(*args))
