"""Ortho-normalised FFT / IFFT on named dims with coordinate-aware shifts, on the GPU.

Host-side mirror of the reference's ``src/xmris/processing/fourier.py``.  Each transformed dim is
one batched 1-D launch of ``xm_fft1d_batched``; (i)fftshift rolls adjacent to a transform are folded
into that launch by the internal ``_shift_in`` / ``_shift_out`` switches.
"""
from __future__ import annotations

import numpy as np

from .. import device as dev
from ..config import COORDS, DIMS
from ..dims import _check_dims, term_attrs
from ._common import Coordinate, LabeledArray, as_labeled, deferred, device_data, like_input, maybe_real


def _as_list(dim):
    return [dim] if isinstance(dim, str) else list(dim)


def _roll(src: LabeledArray, dims, shift_of) -> LabeledArray:
    x, was_real = device_data(src)
    coords = dict(src.coords)
    for d in dims:
        s = shift_of(src.sizes[d])
        x = dev.roll(x, src.get_axis_num(d), s)
        for k, c in list(coords.items()):  # roll_coords=True
            if c.dim == d:
                coords[k] = Coordinate(c.dim, np.roll(c.values, s), c.attrs)
    return LabeledArray(maybe_real(x, was_real), src.dims, coords, dict(src.attrs), src.name)


def fftshift(da, dim):
    """Roll data and coordinates by n//2 (reference ``fourier.py:10-32``)."""
    src = as_labeled(da)
    dims = _as_list(dim)
    _check_dims(src, dims, "fftshift")
    return like_input(_roll(src, dims, lambda n: n // 2), da)


def ifftshift(da, dim):
    """Roll data and coordinates by (n+1)//2 (reference ``fourier.py:35-58``)."""
    src = as_labeled(da)
    dims = _as_list(dim)
    _check_dims(src, dims, "ifftshift")
    return like_input(_roll(src, dims, lambda n: (n + 1) // 2), da)


def _transform(da, dim, out_dim, inverse: bool, name: str, shift_in: bool, shift_out: bool):
    src = as_labeled(da)
    dims = _as_list(dim)
    _check_dims(src, dims, name)
    out_dims = [out_dim] if isinstance(out_dim, str) else out_dim
    if out_dims is not None and len(dims) != len(out_dims):
        raise ValueError("`dim` and `out_dim` lists must have the same length.")
    for d in dims:  # fourier.py:93 reads da.coords[dim]: KeyError without a coordinate
        src.coords[d]
    def compute():
        y, _ = device_data(src)
        for d in dims:
            y = dev.fft(y, src.get_axis_num(d), inverse=inverse, ortho=True, shift_in=shift_in, shift_out=shift_out)
        return y

    if len(dims) == 1 and not inverse and shift_out and not shift_in:  # to_spectrum: a step `autophase` can fuse
        cdt = {np.dtype(np.float32): np.complex64, np.dtype(np.float64): np.complex128}.get(np.dtype(src.dtype), src.dtype)
        x = deferred(src, compute, src.shape, cdt, ("to_spectrum", {"dim": dims[0], "out_dim": out_dims[0] if out_dims else None}))
    else:
        x = compute()
    new_dims = list(src.dims)
    coords = dict(src.coords)
    for i, d in enumerate(dims):
        o = out_dims[i] if out_dims else None
        if inverse:
            term = COORDS.time if (d == DIMS.frequency and o in (None, DIMS.time)) else None
        else:
            term = COORDS.frequency if (d == DIMS.time and o in (None, DIMS.frequency)) else None
        n = src.sizes[d]
        old = src.coords[d].values
        if shift_in:  # the coordinate the un-fused ifftshift would have produced
            old = np.roll(old, (n + 1) // 2)
        delta = (old[1] - old[0]) if len(old) > 1 else 1.0  # fourier.py:95
        new = np.fft.fftfreq(n, d=delta)  # fourier.py:98
        if shift_out:
            new = np.roll(new, n // 2)
        target = o if o is not None else d
        if o is not None and o != d:  # fourier.py:108-109 rename
            new_dims[new_dims.index(d)] = o
            renamed = {}
            for k, c in coords.items():
                renamed[o if k == d else k] = Coordinate(o if c.dim == d else c.dim, c.values, c.attrs)
            coords = renamed
        # other coordinates along the transformed dim would be rolled by the un-fused shifts
        for k, c in list(coords.items()):
            if c.dim == target and k != target:
                v = c.values
                if shift_in:
                    v = np.roll(v, (n + 1) // 2)
                if shift_out:
                    v = np.roll(v, n // 2)
                coords[k] = Coordinate(c.dim, v, c.attrs)
        coords[target] = Coordinate(target, new, term_attrs(term) if term is not None else {})
    out = LabeledArray(x, new_dims, coords, dict(src.attrs), src.name)  # fft writes no attrs
    return like_input(out, da)


def fft(da, dim=DIMS.time, out_dim=None, _shift_in: bool = False, _shift_out: bool = False):
    """N-D ortho FFT, unshifted, reciprocal coordinates ``fftfreq(n, d=c1-c0)``; ``time ->
    frequency`` gets the Hz metadata (reference ``fourier.py:117-173``)."""
    return _transform(da, dim, out_dim, False, "fft", _shift_in, _shift_out)


def ifft(da, dim=DIMS.frequency, out_dim=None, _shift_in: bool = False, _shift_out: bool = False):
    """N-D ortho IFFT (reference ``fourier.py:176-226``)."""
    return _transform(da, dim, out_dim, True, "ifft", _shift_in, _shift_out)


def fftc(da, dim=DIMS.time, out_dim=None):
    """ifftshift -> fft -> fftshift in one launch per dim (reference ``fourier.py:232-264``)."""
    _check_dims(as_labeled(da), _as_list(dim), "ifftshift")
    return _transform(da, dim, out_dim, False, "fft", True, True)


def ifftc(da, dim=DIMS.frequency, out_dim=None):
    """ifftshift -> ifft -> fftshift in one launch per dim (reference ``fourier.py:267-298``)."""
    _check_dims(as_labeled(da), _as_list(dim), "ifftshift")
    return _transform(da, dim, out_dim, True, "ifft", True, True)
