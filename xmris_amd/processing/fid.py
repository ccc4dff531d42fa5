"""FID-domain operations on the GPU: zero filling, apodisation, FID <-> spectrum.

Host-side mirror of the reference's ``src/xmris/processing/fid.py`` (same function names, keyword
names, defaults, error behaviour, coordinate and lineage-attr handling); the array arithmetic runs
in hand-written HIP kernels through ``xmris_amd.device`` (no CPU fallback).
"""
from __future__ import annotations

import copy as _copy

import numpy as np

from .. import device as dev
from ..config import ATTRS, COORDS, DIMS
from ..dims import MSG_POSITION, _check_dims, term_attrs
from ._common import (Coordinate, LabeledArray, as_labeled, binary_op_name, deferred, device_data, like_input,
                      maybe_real, promote_for_float64_operand, promoted_dtype)
from .fourier import fft, fftshift, ifft, ifftshift


def to_spectrum(da, dim: str = DIMS.time, out_dim: str = DIMS.frequency):
    """FID -> centred spectrum: ortho FFT + fftshift (reference ``fid.py:9-42``).

    One kernel launch: the fftshift roll is folded into the FFT's store addresses."""
    src = as_labeled(da)
    _check_dims(src, dim, "to_spectrum")
    return like_input(fft(src, dim=dim, out_dim=out_dim, _shift_out=True), da)


def to_fid(da, dim: str = DIMS.frequency, out_dim: str = DIMS.time):
    """Spectrum -> FID: ifftshift + ortho IFFT, time axis rebuilt from the frequency step
    (reference ``fid.py:45-102``)."""
    src = as_labeled(da)
    _check_dims(src, dim, "to_fid")
    out = ifft(src, dim=dim, out_dim=out_dim, _shift_in=True)
    if dim in src.coords:  # fid.py:80-100
        freqs = src.coords[dim].values
        n = len(freqs)
        if n > 1:
            df = abs(freqs[1] - freqs[0])
            dt = 1.0 / (n * df)
            attrs = term_attrs(COORDS.time) if out_dim == DIMS.time else {}
            out.coords[out_dim] = Coordinate(out_dim, np.arange(n) * dt, attrs)
    return like_input(out, da)


def _apodize(src: LabeledArray, dim: str, weight: np.ndarray, step=None) -> LabeledArray:
    def compute():
        x, was_real = device_data(src)
        x = promote_for_float64_operand(x)  # complex64 * float64 window -> complex128 (fid.py:136-139)
        return maybe_real(dev.apodize(x, src.get_axis_num(dim), weight), was_real)

    # the result's dtype follows numpy's promotion; a recorded step lets `autophase` fuse the chain (phasing.py)
    y = deferred(src, compute, src.shape, promoted_dtype(src.dtype), step) if step is not None else compute()
    out = LabeledArray(y, src.dims, src.coords, _copy.copy(src.attrs), binary_op_name(src, dim))
    return out


def apodize_exp(da, dim: str = DIMS.time, lb: float = 1.0):
    """Multiply by exp(-pi*lb*t), t = the coordinate VALUES of `dim` (reference ``fid.py:105-144``)."""
    src = as_labeled(da)
    _check_dims(src, dim, "apodize_exp")
    t = src.coords[dim].values  # KeyError when `dim` has no coordinate, like the reference
    w = np.exp(-np.pi * lb * t)
    out = _apodize(src, dim, w, step=("apodize_exp", {"dim": dim, "lb": lb, "_weight": w}))
    out.attrs[ATTRS.apodization_lb] = lb
    return like_input(out, da)


def apodize_lg(da, dim: str = DIMS.time, lb: float = 1.0, gb: float = 1.0):
    """Lorentz-to-Gauss window exp(+pi*lb*t) * exp(-t^2/T_G^2) (reference ``fid.py:147-198``)."""
    src = as_labeled(da)
    _check_dims(src, dim, "apodize_lg")
    t = src.coords[dim].values
    w = np.exp(np.pi * lb * t)
    if gb != 0:
        t_g = (2 * np.sqrt(np.log(2))) / (np.pi * gb)
        w = w * np.exp(-(t**2) / (t_g**2))
    out = _apodize(src, dim, w, step=("apodize_lg", {"dim": dim, "lb": lb, "gb": gb, "_weight": w}))
    out.attrs[ATTRS.apodization_lb] = lb
    out.attrs[ATTRS.apodization_gb] = gb
    return like_input(out, da)


def zero_fill(da, dim: str = DIMS.time, target_points: int = 1024, position: str = "end"):
    """Pad `dim` with zeros to `target_points`, extrapolate its coordinate linearly and stamp the
    lineage attrs (reference ``fid.py:201-285``)."""
    src = as_labeled(da)
    _check_dims(src, dim, "zero_fill")
    n = src.sizes[dim]
    if target_points <= n:  # fid.py:235-236: plain copy, no lineage
        return like_input(src.copy(), da)
    pad = target_points - n
    if position == "end":
        pad_left = 0
    elif position == "symmetric":
        pad_left = pad // 2
    else:
        raise ValueError(MSG_POSITION)
    def compute():
        x, was_real = device_data(src)
        return maybe_real(dev.zero_fill(x, src.get_axis_num(dim), int(target_points), pad_left), was_real)

    shape = tuple(int(target_points) if d == dim else s_ for d, s_ in zip(src.dims, src.shape))
    y = deferred(src, compute, shape, src.dtype,
                 ("zero_fill", {"dim": dim, "target_points": int(target_points), "position": position}))
    coords = {}
    for k, c in src.coords.items():
        if c.dim != dim:
            coords[k] = c
        else:  # xarray's pad NaN-fills coordinates along the padded dim
            v = np.full(target_points, np.nan)
            v[pad_left:pad_left + n] = c.values
            coords[k] = Coordinate(c.dim, v, c.attrs)
    if dim in src.coords:  # fid.py:254-278
        old = src.coords[dim].values
        if len(old) > 1:
            delta = old[1] - old[0]
            if position == "end":
                new = old[0] + np.arange(target_points) * delta
            else:
                new = (old[0] - (pad_left * delta)) + np.arange(target_points) * delta
            term = next((c for c in (COORDS.time, COORDS.frequency, COORDS.chemical_shift) if c == dim), None)
            attrs = term_attrs(term) if term is not None else dict(src.coords[dim].attrs)
            coords[dim] = Coordinate(dim, new, attrs)
    out = LabeledArray(y, src.dims, coords, _copy.copy(src.attrs), src.name)
    out.attrs[ATTRS.zero_fill_target] = target_points
    out.attrs[ATTRS.zero_fill_position] = position
    return like_input(out, da)


__all__ = ["to_spectrum", "to_fid", "apodize_exp", "apodize_lg", "zero_fill", "fft", "fftshift", "ifft",
           "ifftshift"]
