"""TEST DOUBLE (not product code): a numpy stand-in for ``xmris_amd.device`` so that the host layer's
metadata logic (dims, coords, attrs, names, errors) can be exercised on a CPU-only box.  The product
never imports this; GPU parity of the real kernels is covered by the ``-m gpu`` tests."""
import numpy as np


def to_device(a, device="cuda", dtype=None):
    a = np.asarray(a)
    if not np.iscomplexobj(a):
        a = a.astype(np.complex64 if a.dtype == np.float32 else np.complex128)
    return a


def zero_fill(x, axis, target_points, pad_left=0):
    pads = [(0, 0)] * x.ndim
    pads[axis] = (pad_left, target_points - x.shape[axis] - pad_left)
    return np.pad(x, pads)


def _along(x, v, axis):
    shape = [1] * x.ndim
    shape[axis] = len(v)
    return x * np.asarray(v).reshape(shape)


def apodize(x, axis, window):
    return _along(x, window, axis)


def phase_apply(x, axis, table):
    return _along(x, table, axis)


def roll(x, axis, shift):
    return np.roll(x, shift, axis=axis)


def fft(x, axis, inverse=False, ortho=True, shift_in=False, shift_out=False):
    n = x.shape[axis]
    if shift_in:
        x = np.roll(x, (n + 1) // 2, axis=axis)
    f = np.fft.ifft if inverse else np.fft.fft
    y = f(x, axis=axis, norm="ortho" if ortho else None)
    if shift_out:
        y = np.roll(y, n // 2, axis=axis)
    return y


def slice_axis(x, axis, start):
    idx = [slice(None)] * x.ndim
    idx[axis] = slice(int(start), None)
    return np.ascontiguousarray(x[tuple(idx)])


def shift_fractional(x, axis, start, table):
    y = slice_axis(x, axis, start)
    return np.fft.ifft(_along(np.fft.fft(y, axis=axis), table, axis), axis=axis)


def absmax_argmax(x):
    flat = int(np.argmax(np.abs(x)))
    return float(np.abs(x).reshape(-1)[flat]), flat


def install(monkeypatch):
    from xmris_amd import device

    for name in ("to_device", "zero_fill", "apodize", "phase_apply", "roll", "fft", "absmax_argmax", "slice_axis",
                 "shift_fractional"):
        monkeypatch.setattr(device, name, globals()[name])
