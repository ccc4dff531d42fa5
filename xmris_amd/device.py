"""Array-level device operations of the xmris spectral hot path (MI355X / gfx950).

Thin host wrappers around the C ABI of ``libxmris_hip.so``.  Inputs are complex torch tensors
resident in HBM (``complex64`` or ``complex128``); PyTorch is used only for device memory and
streams.  Each function names the reference statement it replaces.  There is no CPU fallback:
a tensor that is not on a GPU raises.

The FID / frequency axis may be any axis: it is moved last (one transposing copy) so that the
kernels see ``[n_batch, n]`` C-contiguous rows, and moved back afterwards, exactly like
``da.get_axis_num(dim)`` + ``axes=(axis,)`` in ``processing/fourier.py:152-153``.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def _torch():
    import torch

    return torch


def _dtype_code(x) -> int:
    torch = _torch()
    if x.dtype == torch.complex64:
        return _lib.XM_C64
    if x.dtype == torch.complex128:
        return _lib.XM_C128
    raise TypeError(f"xmris_amd kernels need complex64/complex128 data, got {x.dtype}")


def _real_dtype(x):
    torch = _torch()
    return torch.float32 if x.dtype == torch.complex64 else torch.float64


def _require_device(x):
    torch = _torch()
    if not isinstance(x, torch.Tensor):
        raise TypeError("expected a torch.Tensor resident on the GPU")
    if not x.is_cuda:
        raise RuntimeError(
            "xmris_amd has no CPU path: the tensor must live on a HIP device (use xmris_amd.to_device)"
        )


def _stream(x):
    torch = _torch()
    return torch.cuda.current_stream(x.device).cuda_stream


def to_device(a, device="cuda", dtype=None):
    """Host ndarray (or tensor) -> complex tensor in HBM; real input is promoted to complex."""
    torch = _torch()
    if isinstance(a, torch.Tensor):
        t = a
    else:
        a = np.asarray(a)
        if not np.iscomplexobj(a):
            a = a.astype(np.complex128 if a.dtype != np.float32 else np.complex64)
        t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    if not t.is_complex():
        t = t.to(torch.complex128 if t.dtype == torch.float64 else torch.complex64)
    return t.to(device)


_HOST_STAGE = {}  # device index -> (pinned staging tensors, copy stream, worker pool)
_HOST_STAGE_BUFS = 4
_HOST_STAGE_LOCK = __import__("threading").Lock()  # one large download at a time shares the staging buffers


def to_host(x, chunk_bytes: int = 32 << 20):
    """Device tensor -> host ndarray.  Large results go through a few pinned staging buffers on a side stream:
    the DMA of the next chunks overlaps the host memcpys (worker threads; first-touch page faults of the fresh
    result dominate them) of the previous ones, instead of the runtime's own pageable path (measured 6.9 GB/s
    for a 1 GiB result)."""
    torch = _torch()
    x = x.detach()
    nbytes = x.numel() * x.element_size()
    if not x.is_cuda or nbytes < 4 * chunk_bytes:
        return x.cpu().numpy()
    with _HOST_STAGE_LOCK:
        return _to_host_staged(x.contiguous(), chunk_bytes)


def _to_host_staged(x, chunk_bytes: int):
    torch = _torch()
    from concurrent.futures import ThreadPoolExecutor

    nbytes = x.numel() * x.element_size()
    dev_idx = x.device.index or 0
    if dev_idx not in _HOST_STAGE:
        _HOST_STAGE[dev_idx] = ([torch.empty(chunk_bytes, dtype=torch.uint8, pin_memory=True)
                                 for _ in range(_HOST_STAGE_BUFS)],
                                torch.cuda.Stream(device=x.device), ThreadPoolExecutor(max_workers=_HOST_STAGE_BUFS))
    stage, side, pool = _HOST_STAGE[dev_idx]
    np_dtype = {torch.complex64: np.complex64, torch.complex128: np.complex128, torch.float32: np.float32,
                torch.float64: np.float64}.get(x.dtype)
    if np_dtype is None or chunk_bytes != stage[0].numel():
        return x.cpu().numpy()
    out = np.empty(tuple(x.shape), dtype=np_dtype)
    dst = out.reshape(-1).view(np.uint8)
    src = (torch.view_as_real(x) if x.is_complex() else x).reshape(-1).view(torch.uint8)
    stage_np = [b.numpy() for b in stage]
    side.wait_stream(torch.cuda.current_stream(x.device))  # the producer kernels
    pending = [None] * len(stage)

    def drain(ev, lo, n, k):
        ev.synchronize()
        np.copyto(dst[lo:lo + n], stage_np[k][:n])

    for i, lo in enumerate(range(0, nbytes, chunk_bytes)):
        k, n = i % len(stage), min(chunk_bytes, nbytes - lo)
        if pending[k] is not None:
            pending[k].result()  # the staging buffer is free again
        with torch.cuda.stream(side):
            stage[k][:n].copy_(src[lo:lo + n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        pending[k] = pool.submit(drain, ev, lo, n, k)
    for f in pending:
        if f is not None:
            f.result()
    x.record_stream(side)
    return out


def _rows(x, axis):
    """Move `axis` last and flatten the rest -> ([n_batch, n] contiguous, restore(y, n_new))."""
    nd = x.dim()
    axis = axis % nd
    xm = x.movedim(axis, -1) if axis != nd - 1 else x
    lead = tuple(xm.shape[:-1])
    n = xm.shape[-1]
    x2 = xm.reshape(-1, n)
    if not x2.is_contiguous():
        x2 = x2.contiguous()

    def restore(y2):
        y = y2.reshape(lead + (y2.shape[-1],))
        if axis != nd - 1:
            y = y.movedim(-1, axis).contiguous()
        return y

    return x2, restore


def _table(values, like, complex_table: bool):
    """fp64 host table -> device tensor of the storage precision (rounded once)."""
    torch = _torch()
    if isinstance(values, torch.Tensor):
        t = values
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(values)))
    want = like.dtype if complex_table else _real_dtype(like)
    return t.to(device=like.device, dtype=want).contiguous()


# ---------------------------------------------------------------------------------------------
def zero_fill(x, axis: int, target_points: int, pad_left: int = 0):
    """fid.py:251 ``da.pad({dim: (pad_left, pad_right)}, constant_values=0)`` -- bit-exact copy."""
    _require_device(x)
    torch = _torch()
    x2, restore = _rows(x, axis)
    nb, n_in = x2.shape
    out = torch.empty((nb, target_points), dtype=x.dtype, device=x.device)
    _lib.call("xm_zero_fill", x2.data_ptr(), out.data_ptr(), nb, n_in, target_points, pad_left,
              _dtype_code(x), _stream(x))
    return restore(out)


def apodize(x, axis: int, window):
    """fid.py:139 ``da * weight`` with a real window over `axis` (host-computed in fp64)."""
    _require_device(x)
    torch = _torch()
    x2, restore = _rows(x, axis)
    nb, n = x2.shape
    w = _table(window, x, complex_table=False)
    if w.numel() != n:
        raise ValueError(f"window has {w.numel()} points, axis has {n}")
    out = torch.empty_like(x2)
    _lib.call("xm_apodize", x2.data_ptr(), out.data_ptr(), w.data_ptr(), nb, n, _dtype_code(x), _stream(x))
    return restore(out)


def zf_apod_supported(n_out: int, out_complex128: bool) -> bool:
    """True when `zf_apod` takes this output length (its window must fit the LDS)."""
    return int(n_out) * (8 if out_complex128 else 4) <= 96 * 1024


def zf_apod(x2, n_out: int, pad_left: int, window, promote: bool = False):
    """fid.py:251 + fid.py:136-139 in one launch (`xm_zf_apod`): ``x2`` = [n_batch, n_in] contiguous rows -> [n_batch,
    n_out] rows, zero filled and multiplied by `window` (n_out weights, fp64 host values or a device tensor).
    `promote`: complex64 rows give complex128 results (numpy's promotion against the float64 window)."""
    _require_device(x2)
    torch = _torch()
    if x2.dim() != 2 or not x2.is_contiguous():
        raise ValueError("zf_apod expects a contiguous [n_batch, n_in] tensor")
    nb, n_in = x2.shape
    out_dt = torch.complex128 if (promote or x2.dtype == torch.complex128) else torch.complex64
    rd = torch.float64 if out_dt == torch.complex128 else torch.float32
    if isinstance(window, torch.Tensor):
        w = window.to(device=x2.device, dtype=rd).contiguous()
    else:
        w = torch.from_numpy(np.ascontiguousarray(np.asarray(window, dtype=np.float64))).to(device=x2.device, dtype=rd)
    if w.numel() != n_out:
        raise ValueError(f"window has {w.numel()} points, the zero-filled axis has {n_out}")
    out = torch.empty((nb, n_out), dtype=out_dt, device=x2.device)
    _lib.call("xm_zf_apod", x2.data_ptr(), n_in, out.data_ptr(), w.data_ptr(), nb, n_in, int(n_out), int(pad_left),
              _dtype_code(x2), _lib.XM_C128 if out_dt == torch.complex128 else _lib.XM_C64, _stream(x2))
    return out


def phase_apply(x, axis: int, table):
    """phasing.py:73 ``da * np.exp(1j * phase_array)`` with a complex table over `axis`."""
    _require_device(x)
    torch = _torch()
    x2, restore = _rows(x, axis)
    nb, n = x2.shape
    ph = _table(table, x, complex_table=True)
    if ph.numel() != n:
        raise ValueError(f"phase table has {ph.numel()} points, axis has {n}")
    out = torch.empty_like(x2)
    _lib.call("xm_phase_apply", x2.data_ptr(), out.data_ptr(), ph.data_ptr(), nb, n, _dtype_code(x),
              _stream(x))
    return restore(out)


def roll(x, axis: int, shift: int):
    """fourier.py:31-32 / 57-58 ``da.roll({dim: shift})`` on the data -- bit-exact."""
    _require_device(x)
    torch = _torch()
    x2, restore = _rows(x, axis)
    nb, n = x2.shape
    out = torch.empty_like(x2)
    _lib.call("xm_roll", x2.data_ptr(), out.data_ptr(), nb, n, int(shift) % n, _dtype_code(x), _stream(x))
    return restore(out)


def fft(x, axis: int, inverse: bool = False, ortho: bool = True, shift_in: bool = False,
        shift_out: bool = False):
    """fourier.py:153 / 210 ``np.fft.(i)fftn(values, axes=(axis,), norm="ortho")`` with the
    surrounding (i)fftshift rolls optionally folded into the same launch."""
    _require_device(x)
    torch = _torch()
    x2, restore = _rows(x, axis)
    nb, n = x2.shape
    if n == 1:  # length-1 transform is the identity for every norm used on this path
        return restore(x2.clone())
    flags = ((_lib.XM_FFT_INVERSE if inverse else 0) | (_lib.XM_FFT_ORTHO if ortho else 0)
             | (_lib.XM_FFT_SHIFT_IN if shift_in else 0) | (_lib.XM_FFT_SHIFT_OUT if shift_out else 0))
    out = torch.empty_like(x2)
    _lib.call("xm_fft1d_batched", x2.data_ptr(), out.data_ptr(), nb, n, flags, _dtype_code(x), _stream(x))
    return restore(out)


def slice_axis(x, axis: int, start: int):
    """``da.isel({dim: slice(start, None)})`` on the data (bruker.py:66-67): contiguous copy."""
    _require_device(x)
    idx = [slice(None)] * x.dim()
    idx[axis % x.dim()] = slice(int(start), None)
    return x[tuple(idx)].contiguous()


def shift_fractional(x, axis: int, start: int, table):
    """bruker.py:79-84 on rows that start at sample `start`:  ifft(fft(x[start:]) * table).

    Two launches: the forward transform reads the rows in place (pointer offset + row stride, no slicing
    copy) and multiplies by `table` (complex, n - start values, fp64-computed) on the way out; the
    inverse transform carries numpy's 1/n scale."""
    _require_device(x)
    torch = _torch()
    x2, restore = _rows(x, axis)
    nb, n = x2.shape
    m = n - int(start)
    ph = _table(table, x, complex_table=True)
    if ph.numel() != m:
        raise ValueError(f"table has {ph.numel()} points, sliced axis has {m}")
    view = x2[:, int(start):]
    spec = torch.empty((nb, m), dtype=x.dtype, device=x.device)
    code, st = _dtype_code(x), _stream(x)
    _lib.call("xm_pipeline_fused", view.data_ptr(), n, spec.data_ptr(), None, ph.data_ptr(), nb, m, m, 0, 0,
              None, None, code, st)
    out = torch.empty_like(spec)
    _lib.call("xm_fft1d_batched", spec.data_ptr(), out.data_ptr(), nb, m, _lib.XM_FFT_INVERSE, code, st)
    return restore(out)


def baseline_als(x, axis: int, lam: float, p: float, n_iter: int):
    """baseline.py:10-40 along `axis`: returns real(x) - AsLS baseline as float64 (any real or complex
    float32/float64 input).  fp64 band LDL' solves, one thread per spectrum on transposed scratch."""
    torch = _torch()
    if not isinstance(x, torch.Tensor) or not x.is_cuda:
        raise RuntimeError("xmris_amd has no CPU path: the tensor must live on a HIP device")
    is_complex = x.is_complex()
    if x.dtype in (torch.complex64, torch.float32):
        code = _lib.XM_C64
    elif x.dtype in (torch.complex128, torch.float64):
        code = _lib.XM_C128
    else:
        raise TypeError(f"baseline_als needs float32/float64 (complex) data, got {x.dtype}")
    x2, restore = _rows(x, axis)
    nb, n = x2.shape
    need = int(_lib.load().xm_baseline_als_workspace_bytes(nb, n))
    work = torch.empty(max(need // 8, 1), dtype=torch.float64, device=x.device)
    out = torch.empty((nb, n), dtype=torch.float64, device=x.device)
    _lib.call("xm_baseline_als", x2.data_ptr(), int(is_complex), nb, n, float(lam), float(p), int(n_iter),
              out.data_ptr(), work.data_ptr(), need, code, torch.cuda.current_stream(x.device).cuda_stream)
    return restore(out)


def absmax_argmax(x):
    """phasing.py:229 ``int(np.argmax(np.abs(values)))``: (max |x|, first flat C-order index).

    The flat arg-max does not depend on which axis is the FID axis, so the array is viewed as
    [prod(shape[:-1]), shape[-1]] rows in its own layout.
    """
    _require_device(x)
    torch = _torch()
    xc = x if x.is_contiguous() else x.contiguous()
    n = xc.shape[-1] if xc.dim() else 1
    x2 = xc.reshape(-1, n)
    nb = x2.shape[0]
    rd = _real_dtype(x)
    amax = torch.empty(nb, dtype=rd, device=x.device)
    aidx = torch.empty(nb, dtype=torch.int32, device=x.device)
    gmax = torch.empty(1, dtype=rd, device=x.device)
    gflat = torch.empty(1, dtype=torch.int64, device=x.device)
    code, st = _dtype_code(x), _stream(x)
    _lib.call("xm_absmax_rows", x2.data_ptr(), nb, n, amax.data_ptr(), aidx.data_ptr(), code, st)
    _lib.call("xm_argmax_reduce", amax.data_ptr(), aidx.data_ptr(), nb, n, gmax.data_ptr(), gflat.data_ptr(),
              code, st)
    return float(gmax.item()) ** 0.5, int(gflat.item())


class FusedResult:
    """Outputs of one fused launch: `out` ([n_batch, n_out] or None) and the arg-max pairs."""

    __slots__ = ("out", "absmax2", "argidx")

    def __init__(self, out, absmax2, argidx):
        self.out, self.absmax2, self.argidx = out, absmax2, argidx


def ramp_native(x2, n_out: int, pad_left: int = 0, shift_out: bool = True, ortho: bool = True) -> bool:
    """True when the fused kernel of this geometry applies a linear phase natively (`phase_ramp=` of
    `pipeline_fused` then costs no table and no per-output load); otherwise callers upload a phase table."""
    _require_device(x2)
    flags = (_lib.XM_FFT_ORTHO if ortho else 0) | (_lib.XM_FFT_SHIFT_OUT if shift_out else 0)
    return bool(_lib.load().xm_pipeline_ramp_native(x2.data_ptr(), x2.shape[1], x2.shape[1], int(n_out), int(pad_left),
                                                    flags, _dtype_code(x2)))


def key_native(x2, n_out: int, pad_left: int = 0, shift_out: bool = True, ortho: bool = True) -> bool:
    """True when `pipeline_fused(global_key=, key_result=)` is available for this geometry and dtype (complex64: the
    geometries of `ramp_native` -- without a >= 2x zero fill only in the `phase_ramp=` form; complex128: half lengths
    4096 and 8192)."""
    _require_device(x2)
    flags = (_lib.XM_FFT_ORTHO if ortho else 0) | (_lib.XM_FFT_SHIFT_OUT if shift_out else 0)
    # (the kernels without a >= 2x zero fill pack two rows per lane pair: a single row takes another kernel)
    return x2.shape[0] >= 2 and bool(_lib.load().xm_pipeline_key_native(x2.data_ptr(), x2.shape[1], x2.shape[1], int(n_out),
                                                                       int(pad_left), flags, _dtype_code(x2)))


def pipeline_fused(x2, n_out: int, pad_left: int = 0, window=None, phase_table=None, shift_out: bool = True,
                   ortho: bool = True, want_out: bool = True, want_argmax: bool = False, out=None,
                   absmax2=None, argidx=None, argmax_value_only: bool = False, phase_ramp=None, global_key=None,
                   key_result=None):
    """One launch of zero-fill + window + FFT(+fftshift) [+ |X|^2 arg-max] [+ phase] on
    ``x2`` = [n_batch, n_in] contiguous rows (FID axis last).  `window` / `phase_table` are
    device tensors of the storage precision (real n_out / complex n_out) or None.
    `phase_ramp` = (phase0, dphase) in radians multiplies output k by e^{i (phase0 + dphase k)} instead of a
    table (phasing.py:62-73 on a uniform axis).  `global_key` (`new_argmax_key`, geometries
    with `ramp_native` only) receives the launch's global arg-max (value bits, row) instead of per-row outputs;
    `argmax_key_take` decodes and clears it -- or, with `key_result` (`new_key_result`: 16 bytes of pinned host
    memory), the kernel's last workgroup does that itself and no further launch is needed."""
    _require_device(x2)
    torch = _torch()
    if x2.dim() != 2 or not x2.is_contiguous():
        raise ValueError("pipeline_fused expects a contiguous [n_batch, n_in] tensor")
    nb, n_in = x2.shape
    rd = _real_dtype(x2)
    if want_out and out is None:
        out = torch.empty((nb, n_out), dtype=x2.dtype, device=x2.device)
    if global_key is not None:
        want_argmax, absmax2 = True, global_key  # the key rides in the absmax2 slot, the result record in argidx's
        argidx = key_result
    if want_argmax:
        if absmax2 is None:
            absmax2 = torch.empty(nb, dtype=rd, device=x2.device)
        if argidx is None and global_key is None:
            argidx = torch.empty(nb, dtype=torch.int32, device=x2.device)
    flags = (_lib.XM_FFT_ORTHO if ortho else 0) | (_lib.XM_FFT_SHIFT_OUT if shift_out else 0)
    if global_key is not None:
        flags |= _lib.XM_AMAX_GLOBAL_KEY | _lib.XM_AMAX_VALUE_ONLY
    if argmax_value_only:  # hint: kernels may skip the first-index scan (argidx then holds 0)
        flags |= _lib.XM_AMAX_VALUE_ONLY
    if phase_ramp is not None:
        if phase_table is not None or not want_out:
            raise ValueError("phase_ramp excludes phase_table and needs an output")
        _lib.call(
            "xm_pipeline_fused_ramp", x2.data_ptr(), n_in, out.data_ptr(),
            window.data_ptr() if window is not None else None, float(phase_ramp[0]), float(phase_ramp[1]), nb, n_in,
            n_out, pad_left, flags, absmax2.data_ptr() if want_argmax else None,
            argidx.data_ptr() if (want_argmax and argidx is not None) else None, _dtype_code(x2), _stream(x2))
        return FusedResult(out, absmax2 if want_argmax else None, argidx if want_argmax else None)
    _lib.call(
        "xm_pipeline_fused", x2.data_ptr(), n_in, out.data_ptr() if want_out else None,
        window.data_ptr() if window is not None else None,
        phase_table.data_ptr() if phase_table is not None else None, nb, n_in, n_out, pad_left, flags,
        absmax2.data_ptr() if want_argmax else None,
        argidx.data_ptr() if (want_argmax and argidx is not None) else None, _dtype_code(x2), _stream(x2))
    return FusedResult(out if want_out else None, absmax2 if want_argmax else None,
                       argidx if want_argmax else None)


def row_l1(x2, window=None, pad_left: int = 0, out=None, n_used: int | None = None, sub_step: int = 1, key=None):
    """Windowed L1 norm of every row of ``x2`` = [n_batch, n_in] (`xm_row_l1`): the cheap streaming guess for
    the row that holds the global maximum of the spectra.  `n_used` < n_in sums only the leading samples of
    every row (a caller that knows the window's tail carries no weight skips reading it); `sub_step` > 1 sums
    every sub_step-th 128-sample block of those (a ranking statistic on 1/sub_step of the bytes).  `key` (complex64:
    `new_argmax_key`) receives the row with the largest norm in the same launch
    (`argmax_key_take` decodes and clears it); the per-row norms are then not written unless `out` is given."""
    _require_device(x2)
    torch = _torch()
    if x2.dim() != 2 or not x2.is_contiguous():
        raise ValueError("row_l1 expects a contiguous [n_batch, n_in] tensor")
    nb, n_in = x2.shape
    if out is None and key is None:
        out = torch.empty(nb, dtype=_real_dtype(x2), device=x2.device)
    n_sum = n_in if n_used is None else max(1, min(int(n_used), n_in))
    _lib.call("xm_row_l1", x2.data_ptr(), n_in, window.data_ptr() if window is not None else None, nb, n_sum,
              int(pad_left), max(1, int(sub_step)), out.data_ptr() if out is not None else None,
              key.data_ptr() if key is not None else None, _dtype_code(x2), _stream(x2))
    return out


def guess_supported(x2, n_out: int, pad_left: int = 0, shift_out: bool = True, ortho: bool = True) -> bool:
    """True when the coarse-spectra guess stage (`guess_rows` + `guess_refine`) takes this geometry ("end" zero fill to
    >= 2x with an in-LDS half-length plan; complex64 rows 16-byte aligned)."""
    _require_device(x2)
    if x2.dim() != 2 or not x2.is_contiguous():
        return False
    flags = (_lib.XM_FFT_ORTHO if ortho else 0) | (_lib.XM_FFT_SHIFT_OUT if shift_out else 0)
    return bool(_lib.load().xm_guess_supported(x2.data_ptr(), x2.shape[1], x2.shape[1], int(n_out), int(pad_left), flags,
                                               _dtype_code(x2)))


def guess_rows(x2, n_out: int, window32, est, key, n_guess: int = 0, shift_out: bool = True, ortho: bool = True):
    """`xm_guess_rows`: est[b] = max |X_c|^2 of the coarse spectrum of row b (its first <= 512 windowed samples on a
    1024-bin grid), the largest estimate merged into `key`.  `window32`: float32 weights over the zero-filled axis
    (for complex128 rows too); `est`: float32 [n_batch]."""
    _require_device(x2)
    flags = (_lib.XM_FFT_ORTHO if ortho else 0) | (_lib.XM_FFT_SHIFT_OUT if shift_out else 0)
    nb, n_in = x2.shape
    _lib.call("xm_guess_rows", x2.data_ptr(), n_in, window32.data_ptr() if window32 is not None else None, nb, n_in,
              int(n_out), int(n_guess), flags, est.data_ptr(), key.data_ptr(), _dtype_code(x2), _stream(x2))
    return est


def guess_refine(x2, n_out: int, window32, est, guess_key, work_key, gmax, gflat, out_row, band: float = 0.75,
                 shift_out: bool = True, ortho: bool = True):
    """`xm_guess_refine`: every row whose estimate is within `band` of the largest one is transformed exactly; the
    winner's max |X|^2 -> `gmax` (float32), row * n_out -> `gflat` (int64), its FID as complex128 -> `out_row`
    ([1, n_in]); both keys are left zero."""
    _require_device(x2)
    flags = (_lib.XM_FFT_ORTHO if ortho else 0) | (_lib.XM_FFT_SHIFT_OUT if shift_out else 0)
    nb, n_in = x2.shape
    _lib.call("xm_guess_refine", x2.data_ptr(), n_in, window32.data_ptr() if window32 is not None else None, nb, n_in,
              int(n_out), flags, est.data_ptr(), guess_key.data_ptr(), float(band), work_key.data_ptr(),
              gmax.data_ptr(), gflat.data_ptr(), out_row.data_ptr(), _dtype_code(x2), _stream(x2))
    return out_row


def new_argmax_key(device):
    """A zeroed arg-max key buffer (XM_KEY_BYTES) for `row_l1(key=)` / `pipeline_fused(global_key=)`."""
    return _torch().zeros(131072 // 8, dtype=_torch().int64, device=device)


def new_key_result():
    """Pinned host record (xm_argmax_result: float32 max |X|^2, pad, int64 flat index) a kernel can fill directly."""
    return _torch().zeros(2, dtype=_torch().int64, pin_memory=True)


def read_key_result(rec, complex128: bool = False):
    """(max |X|^2, flat index) of a `new_key_result` record (after the producing launch has completed); a complex128
    launch leaves the maximum as a double."""
    m2 = rec.view(_torch().float64)[0] if complex128 else rec.view(_torch().float32)[0]
    return float(m2.item()), int(rec[1].item())


def argmax_key_take(key, n_per_row: int, gmax, gflat, x2=None, out_row=None):
    """Decode a global arg-max key (`row_l1(key=)` / `pipeline_fused(global_key=)`) into `gmax` (float32, the value)
    and `gflat` (int64, row * n_per_row) -- device-accessible one-element tensors -- and clear it; with `x2` also
    gather the winning row as complex128 into `out_row` ([1, n_in])."""
    torch = _torch()
    if x2 is not None:
        _require_device(x2)
        if out_row is None:
            out_row = torch.empty((1, x2.shape[1]), dtype=torch.complex128, device=x2.device)
    st = torch.cuda.current_stream(key.device).cuda_stream
    _lib.call("xm_argmax_key_take", key.data_ptr(), int(n_per_row), gmax.data_ptr(), gflat.data_ptr(),
              x2.data_ptr() if x2 is not None else None, x2.shape[1] if x2 is not None else 0,
              x2.shape[1] if x2 is not None else 0, out_row.data_ptr() if x2 is not None else None,
              _dtype_code(x2) if x2 is not None else _lib.XM_C64, st)
    return out_row


def argmax_reduce(absmax2, argidx, n: int):
    """Global (max |X|, flat index) from the per-spectrum pairs of a fused launch."""
    torch = _torch()
    nb = absmax2.numel()
    gmax = torch.empty(1, dtype=absmax2.dtype, device=absmax2.device)
    gflat = torch.empty(1, dtype=torch.int64, device=absmax2.device)
    code = _lib.XM_C64 if absmax2.dtype == torch.float32 else _lib.XM_C128
    _lib.call("xm_argmax_reduce", absmax2.data_ptr(), argidx.data_ptr(), nb, n, gmax.data_ptr(),
              gflat.data_ptr(), code, torch.cuda.current_stream(absmax2.device).cuda_stream)
    return float(gmax.item()) ** 0.5, int(gflat.item())


def argmax_reduce_async(absmax2, argidx, n: int, gmax=None, gflat=None):
    """Same reduction, results left on the device (no host synchronisation)."""
    torch = _torch()
    nb = absmax2.numel()
    if gmax is None:
        gmax = torch.empty(1, dtype=absmax2.dtype, device=absmax2.device)
    if gflat is None:
        gflat = torch.empty(1, dtype=torch.int64, device=absmax2.device)
    code = _lib.XM_C64 if absmax2.dtype == torch.float32 else _lib.XM_C128
    _lib.call("xm_argmax_reduce", absmax2.data_ptr(), argidx.data_ptr(), nb, n, gmax.data_ptr(),
              gflat.data_ptr(), code, torch.cuda.current_stream(absmax2.device).cuda_stream)
    return gmax, gflat


def gather_row_c128(x2, gflat, n_per_row: int, out=None):
    """phasing.py:241-242: the spectrum's source row, selected by a flat index that lives on the DEVICE,
    upcast to complex128 ([1, n_in])."""
    _require_device(x2)
    torch = _torch()
    nb, n_in = x2.shape
    if out is None:
        out = torch.empty((1, n_in), dtype=torch.complex128, device=x2.device)
    _lib.call("xm_gather_row_c128", x2.data_ptr(), n_in, n_in, gflat.data_ptr(), int(n_per_row), out.data_ptr(),
              _dtype_code(x2), _stream(x2))
    return out


def last_kernel() -> str:
    """`xm_last_kernel_string`: the kernel the fused dispatcher launched last on this thread, as the profiler names it."""
    v = _lib.load().xm_last_kernel_string()
    return v.decode("utf-8", "replace") if v else ""


def fft_supported(n: int, complex128: bool = False) -> bool:
    if n == 1:
        return True
    return bool(_lib.load().xm_fft_supported(int(n), _lib.XM_C128 if complex128 else _lib.XM_C64))


# ---------------------------------------------------------------------------------------------
# A7 on the device: the (p0, p1) search as one workgroup beside the streaming kernels (csrc/xm_search.hip)
# ---------------------------------------------------------------------------------------------
SEARCH_RECORD_WORDS = 16  # xm_search_result: 128 bytes


def search_supported(n: int, method: str = "acme", x_range: float = 1.0) -> bool:
    """True when `search_launch` takes this slice length / objective (ACME, n <= 16576, a non-degenerate axis)."""
    from .autophase_solver import METHODS

    return method in METHODS and bool(_lib.load().xm_search_supported(int(n), METHODS.index(method), float(x_range)))


def new_search_record():
    """Pinned host record (`xm_search_result`) that a search kernel fills when it ends, `seq` (word 7) last."""
    return _torch().zeros(SEARCH_RECORD_WORDS, dtype=_torch().int64, pin_memory=True)


def uniform_axis(coords):
    """(c0, cstep, x_range) of a coordinate axis when it is uniform to a few ulp (the fftfreq axis of this path), else
    None: phasing.py:56-69's (c - pivot) / (max c - min c) is then linear in the bin index."""
    c = np.asarray(coords, dtype=np.float64)
    if c.size < 2:
        return None
    step = (c[-1] - c[0]) / (c.size - 1)
    rng = float(c.max() - c.min())
    if step == 0 or rng <= 0 or not np.all(np.abs(c - (c[0] + step * np.arange(c.size))) <= 4e-15 * rng):
        return None
    return float(c[0]), float(step), rng


def search_launch(slice_c128, axis, record, seq: int, p0_only: bool = False, seed: int = 42, tol: float = 0.01,
                  maxiter: int = 1000, stream=None):
    """`xm_search_launch`: phasing.py:276-284's differential evolution for the ACME objective on `slice_c128` (n
    complex128 bins, device-accessible: device memory or pinned host memory), asynchronous on `stream` (a
    torch.cuda.Stream; default: the current one).  `axis` = `uniform_axis(coords)`.  `record` (`new_search_record`)
    receives the result; `search_done(record, seq)` tells when."""
    torch = _torch()
    n = slice_c128.numel()
    st = (stream if stream is not None else torch.cuda.current_stream()).cuda_stream
    _lib.call("xm_search_launch", slice_c128.data_ptr(), int(n), float(axis[0]), float(axis[1]), float(axis[2]), 0,
              int(bool(p0_only)), int(seed), float(tol), int(maxiter), int(seq), record.data_ptr(), st)


def search_done(record, seq: int) -> bool:
    return int(_lib.load().xm_atomic_load_acquire_i64(record.data_ptr() + 56)) == int(seq)


def read_search_record(record) -> dict:
    """The fields of a finished `xm_search_result`."""
    f = record.view(_torch().float64)
    i = record.view(_torch().int32)
    return dict(x=(float(f[0]), float(f[1])), fun=float(f[2]), pg_norm=float(f[3]), nfev=int(i[8]), nit=int(i[9]),
                status=int(i[10]), needs_polish=bool(int(i[11])), target_idx=int(i[12]),
                t_us=[float(v) for v in f[8:14]])


def search_eval(slice_c128, axis, xs, target_idx: int = -1, p0_only: bool = False):
    """`xm_search_eval`: the device objective at the rows of `xs` ([count, 2] degrees); synchronous (tests)."""
    torch = _torch()
    xs_t = torch.as_tensor(np.ascontiguousarray(xs, dtype=np.float64)).reshape(-1, 2).to(slice_c128.device if slice_c128.is_cuda else "cuda")
    fs = torch.empty(xs_t.shape[0], dtype=torch.float64, device=xs_t.device)
    _lib.call("xm_search_eval", slice_c128.data_ptr(), int(slice_c128.numel()), float(axis[0]), float(axis[1]),
              float(axis[2]), int(target_idx), int(bool(p0_only)), xs_t.data_ptr(), int(xs_t.shape[0]), fs.data_ptr(),
              torch.cuda.current_stream().cuda_stream)
    return fs.cpu().numpy()


class ChipPartition:
    """`xm_stream_create`: one compute stream that owns all CUs but `reserved`, and `n_search` streams that own the
    reserved ones (spread over the eight XCDs), as torch stream objects.  Kept for the life of the process."""

    def __init__(self, device, reserved: int, n_search: int):
        import ctypes

        torch = _torch()
        self.device, self.reserved = device, int(reserved)
        self._handles = []

        def make(partition):
            h = ctypes.c_void_p()
            with torch.cuda.device(device):
                _lib.call("xm_stream_create", ctypes.byref(h), int(reserved), partition)
            self._handles.append(h.value)
            return torch.cuda.ExternalStream(h.value, device=device)

        self.compute = make(0)
        self.search = [make(1) for _ in range(int(n_search))]

    def another_search_stream(self):
        import ctypes

        h = ctypes.c_void_p()
        with _torch().cuda.device(self.device):
            _lib.call("xm_stream_create", ctypes.byref(h), self.reserved, 1)
        self._handles.append(h.value)
        st = _torch().cuda.ExternalStream(h.value, device=self.device)
        self.search.append(st)
        return st


_PARTITIONS = {}


def chip_partition(device, reserved: int, n_search: int) -> ChipPartition:
    torch = _torch()
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), int(reserved))
    part = _PARTITIONS.get(key)
    if part is None:
        part = _PARTITIONS[key] = ChipPartition(device, reserved, n_search)
    while len(part.search) < n_search:
        part.another_search_stream()
    return part


def replacement_search_stream(device, partition_streams):
    """A fresh stream for searches (one whose kernel is still running was retired): of the search partition when the
    chip is split, an ordinary one otherwise."""
    torch = _torch()
    if partition_streams is None:
        return torch.cuda.Stream(device=device)
    for part in _PARTITIONS.values():
        if partition_streams and partition_streams[0] in part.search:
            return part.another_search_stream()
    return torch.cuda.Stream(device=device)
