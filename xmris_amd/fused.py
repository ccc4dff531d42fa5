"""Labelled front end of the fused hot path (``.xmr.spectral_pipeline``).

Produces exactly the DataArray that ``da.xmr.zero_fill(target_points=T).xmr.apodize_exp(lb=L)
.xmr.to_spectrum().xmr.autophase()`` produces (dims, coordinates with attrs, lineage attrs, name),
but the data makes two passes through HBM instead of six (see ``xmris_amd.pipeline``).
"""
from __future__ import annotations

import copy as _copy

import numpy as np

from . import pipeline as pl
from .config import ATTRS, COORDS, DIMS
from .labeled import Coordinate, LabeledArray, as_labeled, like_input
from .processing._common import device_data
from .dims import MSG_METHOD, MSG_MODE, MSG_MODE_ALL, MSG_POSITION, _check_dims, term_attrs


def spectral_pipeline(da, target_points: int = 1024, lb: float = 1.0, dim: str = DIMS.time,
                      out_dim: str = DIMS.frequency, position: str = "end", method: str = "acme",
                      peak_width=100, target_coord=None, p0_only: bool = False, mode: str = "single", _window=None,
                      _apodization_attrs=None, _promote: bool = False, **kwargs):
    """`_window` / `_apodization_attrs` (private: the lazy chain's `apodize_lg`): the weights over the zero-filled axis
    and the lineage attrs that replace `apodization_lb = lb`.  `_promote` (private: the lazy chain on host data): widen
    complex64 to complex128 like the staged chain does.

    Data that live in HOST memory (a numpy-backed array of at least `hostpath.min_bytes()`, FID axis last) take
    `hostpath.run_host`: upload, both passes and the download overlap chunk by chunk and the result comes back as a
    host array."""
    src = as_labeled(da)
    _check_dims(src, dim, "zero_fill")
    if position not in ("end", "symmetric"):
        raise ValueError(MSG_POSITION)
    if mode == "all":
        raise NotImplementedError(MSG_MODE_ALL)
    elif mode != "single":
        raise ValueError(MSG_MODE)
    if method not in ("acme", "peak_minima", "positivity"):
        raise ValueError(MSG_METHOD)
    t = src.coords[dim].values  # apodize_exp needs the coordinate (KeyError otherwise)
    host = _host_rows(src, dim)
    if host is not None:
        from . import hostpath

        x2h, lead = host
        y2, res, plan = hostpath.run_host(x2h, t, target_points, lb, position, window_host=_window, method=method,
                                          peak_width=peak_width, target_coord=target_coord, p0_only=p0_only,
                                          promote=_promote)
        return like_input(_label_result(src, y2.reshape(lead + (plan.n_out,)), res, plan, dim, out_dim, target_points,
                                        position, lb, _apodization_attrs), da)
    x, _ = device_data(src)
    if _promote:
        from .processing._common import promote_for_float64_operand

        x = promote_for_float64_operand(x)
    ax = src.get_axis_num(dim)
    nd = x.dim()
    xm = x.movedim(ax, -1) if ax != nd - 1 else x
    lead = tuple(xm.shape[:-1])
    x2 = xm.reshape(-1, xm.shape[-1])
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    plan = pl.make_plan(x2, t, target_points, lb, position, window_host=_window)
    if ax != nd - 1:
        # the global arg-max must follow the ORIGINAL C order (phasing.py:229); with the FID axis moved
        # the fused pre-pass would break ties differently, so fall back to the staged calls
        from .processing import apodize_exp, autophase, to_spectrum, zero_fill

        out = autophase(to_spectrum(apodize_exp(zero_fill(src, dim, target_points, position), dim, lb), dim, out_dim),
                        out_dim, method=method, peak_width=peak_width, target_coord=target_coord,
                        p0_only=p0_only, **kwargs)
        return like_input(out, da)
    y2, res, plan = pl.run(x2, t, target_points, lb, method=method, peak_width=peak_width,
                           target_coord=target_coord, p0_only=p0_only, plan=plan)
    return like_input(_label_result(src, y2.reshape(lead + (plan.n_out,)), res, plan, dim, out_dim, target_points, position,
                                    lb, _apodization_attrs), da)


def _host_rows(src, dim):
    """([n_batch, n] host rows, leading shape) when `src` is numpy-backed complex data large enough for the chunked
    host path with the FID axis last; else None."""
    from . import hostpath

    if src.is_device_resident or src.get_axis_num(dim) != src.ndim - 1:
        return None
    x = src.data
    if not isinstance(x, np.ndarray) or x.dtype not in (np.complex64, np.complex128) or x.nbytes < hostpath.min_bytes():
        return None
    try:
        import torch

        if not torch.cuda.is_available():
            return None
    except ImportError:
        return None
    x = np.ascontiguousarray(x)
    return x.reshape(-1, x.shape[-1]), tuple(x.shape[:-1])


def _label_result(src, y, res, plan, dim, out_dim, target_points, position, lb, _apodization_attrs):
    """Dims, coordinates, lineage attrs and name of the fused result (fid.py:254-283, fourier.py:92-111, phasing.py:76-94)."""
    n, n_out = src.sizes[dim], plan.n_out
    new_dims = tuple(out_dim if d == dim else d for d in src.dims)
    coords = {}
    for k, c in src.coords.items():
        if c.dim != dim:
            coords[k] = c
        elif k != dim:  # bystander coordinate along the FID axis: NaN-padded, then rolled with the data
            v = np.full(n_out, np.nan)
            v[plan.pad_left:plan.pad_left + n] = c.values
            coords[k] = Coordinate(out_dim, np.roll(v, n_out // 2), c.attrs)
    term = COORDS.frequency if (dim == DIMS.time and out_dim in (None, DIMS.frequency)) else None
    coords[out_dim] = Coordinate(out_dim, plan.freq, term_attrs(term) if term is not None else {})
    attrs = _copy.copy(src.attrs)
    if target_points > n:
        attrs[ATTRS.zero_fill_target] = target_points
        attrs[ATTRS.zero_fill_position] = position
    if _apodization_attrs is not None:
        attrs.update(_apodization_attrs)
    else:
        attrs[ATTRS.apodization_lb] = lb
    attrs[ATTRS.phase_p0] = res.p0
    attrs[ATTRS.phase_p1] = res.p1
    attrs[ATTRS.phase_pivot] = res.pivot
    attrs[ATTRS.phase_pivot_coord] = out_dim
    name = src.name if src.name == dim == out_dim else None
    return LabeledArray(y, new_dims, coords, attrs, name)
