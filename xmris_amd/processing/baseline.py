"""Asymmetric-least-squares baseline correction on the GPU (the step that
follows the hot path on a phased spectrum).

Host-side mirror of the reference's ``src/xmris/processing/baseline.py:43-119``: works on the REAL part,
returns a real float64 array, keeps the input attrs and stamps ``baseline_method/lam/p/iter``.  The
per-spectrum pentadiagonal solves run in ``xm_baseline_als`` (fp64 band LDL', one thread per spectrum).
"""
from __future__ import annotations

import copy as _copy

import numpy as np

from .. import device as dev
from ..config import ATTRS, DIMS
from ..dims import _check_dims
from ._common import as_labeled, like_input


def baseline_als(da, dim: str = DIMS.frequency, lam: float = 1e5, p: float = 0.001, n_iter: int = 10):
    src = as_labeled(da)
    _check_dims(src, dim, "baseline_als")
    if src.is_device_resident:
        x = src.data
    else:  # upload as is: real data stays real (the kernel reads the real part of complex data itself)
        import torch

        x = torch.from_numpy(np.ascontiguousarray(src.data)).to("cuda")
    if not (x.is_complex() or x.is_floating_point()):
        x = x.double()
    y = dev.baseline_als(x, src.get_axis_num(dim), lam, p, n_iter)
    out = src.copy(data=y)
    out.attrs = _copy.copy(src.attrs)
    out.attrs[ATTRS.baseline_method] = "als"
    out.attrs[ATTRS.baseline_lam] = lam
    out.attrs[ATTRS.baseline_p] = p
    out.attrs[ATTRS.baseline_iter] = n_iter
    return like_input(out, da)
