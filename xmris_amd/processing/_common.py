"""Shared host helpers of the processing functions: device upload, dim bookkeeping."""
from __future__ import annotations

import numpy as np

from .. import device as dev
from ..labeled import Coordinate, Deferred, LabeledArray, as_labeled, like_input  # noqa: F401


def lazy_enabled() -> bool:
    """The hot path's three array steps (zero_fill, apodize_exp, to_spectrum) record themselves instead of running
    when this is on (default); `XMRIS_AMD_EAGER=1` makes every call compute at once."""
    import os

    return not os.environ.get("XMRIS_AMD_EAGER")


def deferred(src: LabeledArray, compute, shape, dtype, step):
    """`compute()` now (eager mode) or a `Deferred` that runs it on first use."""
    if not lazy_enabled():
        return compute()
    return Deferred(compute, shape, dtype, step, src)


def promoted_dtype(dt):
    """numpy's result dtype of (array of dtype `dt`) * (float64 / complex128 operand), see promote_for_float64_operand."""
    import os

    dt = np.dtype(dt)
    if os.environ.get("XMRIS_AMD_KEEP_COMPLEX64"):
        return dt
    return {np.dtype(np.complex64): np.dtype(np.complex128), np.dtype(np.float32): np.dtype(np.float64)}.get(dt, dt)


def device_data(da: LabeledArray):
    """The array's data as a complex tensor in HBM (uploads host data once; real data is promoted
    to complex of the same precision) and whether the input was real-valued."""
    was_real = not np.issubdtype(da.dtype, np.complexfloating)
    if da.is_device_resident:
        x = da.data
        if was_real:
            x = dev.to_device(x)
        return x, was_real
    return dev.to_device(da.data), was_real


def promote_for_float64_operand(x):
    """numpy's promotion when an array meets a float64 / complex128 operand (the reference multiplies by
    ``np.exp(-pi*lb*t)`` -- float64, fid.py:136-139 -- or by ``np.exp(1j*phi)`` -- complex128, phasing.py:73,
    vendor/bruker.py:83): single precision becomes double, so from the first such step on the reference's chain is
    complex128 whatever the FID's storage was.  The staged calls follow it (their results feed `autophase`, whose
    search is sensitive to the last bits of its input); the fused `spectral_pipeline` keeps the storage precision and
    recomputes the one arg-max spectrum in complex128 instead.  ``XMRIS_AMD_KEEP_COMPLEX64=1`` switches the
    promotion off."""
    import os

    if os.environ.get("XMRIS_AMD_KEEP_COMPLEX64"):
        return x
    if hasattr(x, "detach"):
        import torch

        return x.to(torch.complex128) if x.dtype == torch.complex64 else x
    return x.astype(np.complex128) if x.dtype == np.complex64 else x


def maybe_real(x, was_real: bool):
    """Ops that keep a real array real in the reference (pad, roll, real window) return the real part."""
    if not was_real:
        return x
    r = x.real
    return r.contiguous() if hasattr(r, "contiguous") else np.ascontiguousarray(r)


def to_host(x) -> np.ndarray:
    """Device tensor (or ndarray) -> host ndarray."""
    return dev.to_host(x) if hasattr(x, "detach") else np.asarray(x)


def binary_op_name(da: LabeledArray, dim: str):
    """xarray keeps a binary op's name only when both operands share it; the other operand here is
    the coordinate of `dim` (named `dim`)."""
    return da.name if da.name == dim else None


def fused_materialise(la: LabeledArray):
    """The data of `to_spectrum(apodize_exp | apodize_lg([zero_fill](fid)))` / `to_spectrum(zero_fill(fid))` when the whole chain is
    still recorded and its END is asked for: ONE launch of the fused kernel (zero fill + window + ortho FFT +
    fftshift, `xm_pipeline_fused`) on the root instead of three staged passes with two intermediates (22 GiB of
    traffic instead of 6 for 32,768 x 4096 -> 8192 complex128).  None when the pattern does not apply -- another chain,
    an intermediate someone has looked at, real input, the FID axis not last -- and the steps run one by one."""
    nodes, node = [], la
    while node._data is None:
        nodes.append(node._lazy.step)
        node = node._lazy.parent
    root = node
    # either window (the kernel takes the weights as they are)
    names = ["apodize" if s_[0] in ("apodize_exp", "apodize_lg") else s_[0] for s_ in nodes]
    if names == ["apodize", "zero_fill"]:  # the chain stops before the FFT: zero fill + window in one launch
        return _zf_apod_materialise(root, nodes)
    if names not in (["to_spectrum", "apodize", "zero_fill"], ["to_spectrum", "apodize"], ["to_spectrum", "zero_fill"]):
        return None
    steps = {name: s_[1] for name, s_ in zip(names, nodes)}
    d0 = steps["to_spectrum"]["dim"]
    if any(steps[k]["dim"] != d0 for k in steps) or d0 not in root.dims or root.get_axis_num(d0) != root.ndim - 1:
        return None
    if not np.issubdtype(root.dtype, np.complexfloating) or root.ndim < 1:
        return None
    x, _ = device_data(root)
    if not hasattr(x, "detach") or not x.is_contiguous():
        return None
    import torch

    n = root.sizes[d0]
    n_out, pad_left = n, 0
    if "zero_fill" in steps:
        n_out = int(steps["zero_fill"]["target_points"])
        pad_left = (n_out - n) // 2 if steps["zero_fill"]["position"] == "symmetric" else 0
    window = None
    if "apodize" in steps:
        x = promote_for_float64_operand(x)  # complex64 * float64 window -> complex128 (fid.py:136-139)
        rd = torch.float64 if x.dtype == torch.complex128 else torch.float32
        window = torch.from_numpy(np.ascontiguousarray(steps["apodize"]["_weight"], dtype=np.float64)).to(x.device, rd)
    if x.numel() == 0 or not dev.fft_supported(n_out, complex128=x.dtype == torch.complex128):
        return None  # the staged steps raise the reference's own errors
    x2 = x.reshape(-1, n)
    out = dev.pipeline_fused(x2, n_out, pad_left, window=window).out
    return out.reshape(tuple(root.shape[:-1]) + (n_out,))


def _zf_apod_materialise(root: LabeledArray, nodes):
    """`apodize_exp | apodize_lg(zero_fill(fid))` asked for its values: `xm_zf_apod` on the root -- one read of the FIDs,
    one write of the result, no zero-filled intermediate (fid.py:251, 136-139).  None -> the staged steps."""
    apod, zf = nodes[0][1], nodes[1][1]
    d0 = apod["dim"]
    if zf["dim"] != d0 or d0 not in root.dims or root.get_axis_num(d0) != root.ndim - 1:
        return None
    if not np.issubdtype(root.dtype, np.complexfloating) or root.ndim < 1:
        return None
    x, _ = device_data(root)
    if not hasattr(x, "detach") or not x.is_contiguous() or x.numel() == 0:
        return None
    import os

    import torch

    n = root.sizes[d0]
    n_out = int(zf["target_points"])
    pad_left = (n_out - n) // 2 if zf["position"] == "symmetric" else 0
    promote = x.dtype == torch.complex64 and not os.environ.get("XMRIS_AMD_KEEP_COMPLEX64")
    if n_out <= n or not dev.zf_apod_supported(n_out, promote or x.dtype == torch.complex128):
        return None
    out = dev.zf_apod(x.reshape(-1, n), n_out, pad_left, apod["_weight"], promote=promote)
    return out.reshape(tuple(root.shape[:-1]) + (n_out,))
