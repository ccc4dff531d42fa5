"""TEST DOUBLE (not product code): numpy/torch-CPU stand-ins for the device entry points and the CUDA stream / event
objects that `xmris_amd.pipeline.run_stream` touches, so that the streaming EXECUTOR itself -- its look-ahead, the order
of its exchange / broadcast calls, verification, cross-rank repairs, hedged searches, the device-search hand-off -- can
run on a CPU-only box, in one process or in eight.  The arithmetic is numpy's on complex128 rows; the real kernels are
covered by the `-m gpu` tests.  `install()` patches the current process for good (worker processes of the multi-rank
tests call it first thing); `install(monkeypatch)` undoes itself with the fixture."""
import numpy as np
import torch

COARSE = 512  # the guess stage sees the first 512 samples of a row on a 1024-bin grid (xm_guess_rows)
COUNTS = {"guess_rows": 0, "guess_refine": 0, "main": 0, "prepass": 0, "search_launch": 0}


class FakeEvent:
    def __init__(self, enable_timing=False, blocking=False):
        pass

    def record(self, stream=None):
        pass

    def synchronize(self):
        pass

    def query(self):
        return True

    def elapsed_time(self, other):
        return 0.0


class FakeStream:
    cuda_stream = 0

    def __init__(self, device=None, priority=0):
        pass

    def wait_event(self, ev):
        pass

    def wait_stream(self, s):
        pass

    def synchronize(self):
        pass


class FusedResult:
    __slots__ = ("out", "absmax2", "argidx")

    def __init__(self, out, absmax2, argidx):
        self.out, self.absmax2, self.argidx = out, absmax2, argidx


def _spectra(x2, n_out, pad_left, window):
    x = x2.numpy().astype(np.complex128)
    z = np.zeros((x.shape[0], n_out), dtype=np.complex128)
    z[:, pad_left:pad_left + x.shape[1]] = x
    if window is not None:
        z = z * window.numpy().astype(np.float64)[None, :]
    return np.roll(np.fft.fft(z, axis=1, norm="ortho"), n_out // 2, axis=1)


def pipeline_fused(x2, n_out, pad_left=0, window=None, phase_table=None, shift_out=True, ortho=True, want_out=True,
                   want_argmax=False, out=None, absmax2=None, argidx=None, argmax_value_only=False, phase_ramp=None,
                   global_key=None, key_result=None):
    spec = _spectra(x2, n_out, pad_left, window)
    m2 = (spec.real ** 2 + spec.imag ** 2)
    if want_out:
        COUNTS["main"] += 1
        y = spec
        if phase_ramp is not None:
            y = spec * np.exp(1j * (phase_ramp[0] + phase_ramp[1] * np.arange(n_out)))[None, :]
        elif phase_table is not None:
            y = spec * phase_table.numpy()[None, :]
        if out is None:
            out = torch.empty((x2.shape[0], n_out), dtype=x2.dtype)
        out.copy_(torch.from_numpy(y).to(out.dtype))
    else:
        COUNTS["prepass"] += 1
    if global_key is not None:  # the launch's global arg-max straight into the result record (first row on ties)
        rowmax = m2.max(axis=1)
        row = int(np.argmax(rowmax))
        if x2.dtype == torch.complex128:
            key_result.view(torch.float64)[0] = float(rowmax[row])
        else:
            key_result.view(torch.float32)[0] = float(rowmax[row])
        key_result[1] = row * n_out
        return FusedResult(out, global_key, key_result)
    if want_argmax:
        rd = torch.float32 if x2.dtype == torch.complex64 else torch.float64
        if absmax2 is None:
            absmax2 = torch.empty(x2.shape[0], dtype=rd)
        if argidx is None:
            argidx = torch.empty(x2.shape[0], dtype=torch.int32)
        absmax2.copy_(torch.from_numpy(m2.max(axis=1)).to(rd))
        argidx.copy_(torch.from_numpy(np.zeros(x2.shape[0]) if argmax_value_only else m2.argmax(axis=1)).to(torch.int32))
    return FusedResult(out if want_out else None, absmax2 if want_argmax else None, argidx if want_argmax else None)


def argmax_reduce_async(absmax2, argidx, n, gmax=None, gflat=None):
    a = absmax2.numpy()
    b = int(np.argmax(a))
    if gmax is None:
        gmax = torch.empty(1, dtype=absmax2.dtype)
    if gflat is None:
        gflat = torch.empty(1, dtype=torch.int64)
    gmax[0] = float(a[b])
    gflat[0] = b * n + int(argidx[b])
    return gmax, gflat


def argmax_reduce(absmax2, argidx, n):
    g, f = argmax_reduce_async(absmax2, argidx, n)
    return float(g.item()) ** 0.5, int(f.item())


def gather_row_c128(x2, gflat, n_per_row, out=None):
    row = int(gflat.item()) // n_per_row
    if out is None:
        out = torch.empty((1, x2.shape[1]), dtype=torch.complex128)
    out.copy_(x2[row:row + 1].to(torch.complex128))
    return out


def guess_rows(x2, n_out, window32, est, key, n_guess=0, shift_out=True, ortho=True):
    """Coarse spectra: the first <= 512 windowed samples of every row on a 1024-bin grid."""
    COUNTS["guess_rows"] += 1
    ng = min(COARSE, x2.shape[1])
    z = x2.numpy()[:, :ng].astype(np.complex128) * window32.numpy().astype(np.float64)[None, :ng]
    c = np.fft.fft(z, n=2 * COARSE, axis=1) / np.sqrt(n_out)
    est.copy_(torch.from_numpy((c.real ** 2 + c.imag ** 2).max(axis=1)).to(torch.float32))
    return est


def guess_refine(x2, n_out, window32, est, guess_key, work_key, gmax, gflat, out_row, band=0.75, shift_out=True, ortho=True):
    """Rows whose estimate is within the band of the largest one are transformed exactly; winner = first arg-max."""
    COUNTS["guess_refine"] += 1
    e = est.numpy().astype(np.float64)
    cand = np.nonzero(e >= band * band * e.max())[0]
    spec = _spectra(x2[torch.from_numpy(cand)], n_out, 0, window32)
    m2 = (spec.real ** 2 + spec.imag ** 2).max(axis=1)
    row = int(cand[int(np.argmax(m2))])
    gmax[0] = float(m2.max())
    gflat[0] = row * n_out
    out_row.copy_(x2[row:row + 1].to(torch.complex128))
    return out_row


def new_argmax_key(device):
    return torch.zeros(16, dtype=torch.int64)


def new_key_result():
    return torch.zeros(2, dtype=torch.int64)


def read_key_result(rec, complex128=False):
    m2 = rec.view(torch.float64)[0] if complex128 else rec.view(torch.float32)[0]
    return float(m2.item()), int(rec[1].item())


def new_search_record():
    return torch.zeros(16, dtype=torch.int64)


def search_supported(n, method="acme", x_range=1.0):
    return method == "acme" and n >= 2 and x_range > 0


def search_launch(slice_c128, axis, record, seq, p0_only=False, seed=42, tol=0.01, maxiter=1000, stream=None):
    """The device search's CONTRACT on the host engine: scipy's generations (xm_solver_de), then the projected-gradient
    test -- `record` is complete when this returns."""
    from xmris_amd import autophase_solver as aps

    COUNTS["search_launch"] += 1
    sl = slice_c128.numpy().reshape(-1).astype(np.complex128)
    n = sl.size
    coords = axis[0] + axis[1] * np.arange(n)
    k = int(np.argmax(sl.real ** 2 + sl.imag ** 2))
    obj = aps.NativeObjective(sl, coords, float(coords[k]), k, 1, "acme")
    obj.set_threads(1)
    rc, x, fun, nfev, nit = obj.de(p0_only, seed=seed, tol=tol, maxiter=maxiter)
    lo, hi = np.array([-180.0, -4000.0])[:len(x)], np.array([180.0, 4000.0])[:len(x)]
    _, g0 = obj.fg(np.clip(x, lo, hi), lo, hi)
    pg = np.where(g0 < 0, np.maximum(x - hi, g0), np.minimum(x - lo, g0))
    f = record.view(torch.float64)
    i = record.view(torch.int32)
    f[0], f[1], f[2], f[3] = float(x[0]), (float(x[1]) if len(x) > 1 else 0.0), float(fun), float(np.abs(pg).max())
    i[8], i[9], i[10], i[11], i[12] = int(nfev), int(nit), int(rc), int(np.abs(pg).max() > 0.5e-5), k
    record[7] = int(seq)


def search_done(record, seq):
    return int(record[7]) == int(seq)


def install(monkeypatch=None):
    from xmris_amd import device

    def put(obj, name, value):
        if monkeypatch is not None:
            monkeypatch.setattr(obj, name, value)
        else:
            setattr(obj, name, value)

    for name in ("pipeline_fused", "argmax_reduce_async", "argmax_reduce", "gather_row_c128", "guess_rows", "guess_refine",
                 "new_argmax_key", "new_key_result", "read_key_result", "new_search_record", "search_supported",
                 "search_launch", "search_done"):
        put(device, name, globals()[name])
    put(device, "last_kernel", lambda: "numpy double")
    put(device, "key_native", lambda *a, **k: True)
    put(device, "guess_supported", lambda *a, **k: True)
    put(device, "ramp_native", lambda *a, **k: True)
    put(device, "_require_device", lambda x: None)
    import contextlib
    import types

    put(device, "chip_partition", lambda dev_, reserved, n_search: types.SimpleNamespace(
        compute=FakeStream(), search=[FakeStream() for _ in range(n_search)]))
    put(device, "replacement_search_stream", lambda dev_, streams: FakeStream())
    put(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    put(torch.cuda, "Event", FakeEvent)
    put(torch.cuda, "Stream", FakeStream)
    put(torch.cuda, "current_stream", lambda device=None: FakeStream())
    real_empty, real_zeros = torch.empty, torch.zeros

    def empty(*a, pin_memory=False, **k):
        return real_empty(*a, **k)

    def zeros(*a, pin_memory=False, **k):
        return real_zeros(*a, **k)

    put(torch, "empty", empty)
    put(torch, "zeros", zeros)
    for k_ in COUNTS:
        COUNTS[k_] = 0
