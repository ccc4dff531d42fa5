"""The fused hot path for callers whose FIDs live in HOST memory (an `xarray.DataArray` / numpy user of the drop-in,
reference boundary ``core/accessor.py:452-550, 630-683``): upload, the two passes and the download overlapped instead
of run one after the other.

    upload        the rows go to HBM in chunks: worker threads copy pageable -> pinned staging buffers, the copy
                  engine moves staging -> HBM on a side stream, and
    pre-pass      the arg-max pass of chunk k runs on the compute stream as soon as chunk k has landed (the upload of
                  chunk k+1 is in flight meanwhile);
    search        global arg-max -> the winning row's spectrum in complex128 -> (p0, p1) on the host (O(1));
    main pass     chunk by chunk into a small ring of device buffers, and
    download      each finished chunk is copied into the RESULT -- a pinned host array that the caller gets as a numpy
                  view (PyTorch's caching host allocator recycles the pages between calls) -- while the next chunk is
                  being transformed.

The FIDs are uploaded once and stay resident for the main pass; nothing is computed on the host but the O(1) search.
PyTorch is used for device memory, pinned memory, streams and events only.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import device as dev
from . import pipeline as pl

_STAGE = {}   # (device index, bytes) -> [pinned uint8 tensors]
_POOL = None  # worker threads of the pageable -> pinned copies
_LOCK = __import__("threading").Lock()  # the staging buffers are shared: one `run_host` at a time per process


def _pool():
    global _POOL
    if _POOL is None:
        _POOL = ThreadPoolExecutor(max_workers=6, thread_name_prefix="xm-host")
    return _POOL


def min_bytes() -> int:
    """Host inputs below this size take the plain path (one upload, one download): the chunk machinery only pays for
    transfers of tens of megabytes."""
    return int(os.environ.get("XMRIS_AMD_HOST_STREAM_MIN", str(64 << 20)))


def run_host(x_host: np.ndarray, t, target_points: int, lb, position: str = "end", window_host=None, method="acme",
             peak_width=100, target_coord=None, p0_only=False, polish="exact", chunk_bytes: int = 128 << 20,
             device="cuda", timing: dict | None = None, promote: bool = False, pinned_result: bool | None = None):
    """`pipeline.run` for ``x_host`` = [n_batch, n_time] complex64 / complex128 rows in host memory.  Returns
    (phased spectra as a host ndarray [n_batch, n_out], AutophaseResult, plan).  `timing` (optional dict) receives the
    wall-clock split.  `promote`: complex64 rows are uploaded as they are and widened to complex128 in HBM (numpy's
    promotion in the staged chain, fid.py:136-139: the result is complex128).  `pinned_result`: the result array is
    page-locked host memory handed out as a numpy view (fastest: the download lands in it directly; default, and
    `XMRIS_AMD_PINNED_RESULT=0` switches it off); False -- or a failed page-locked allocation -- downloads through two
    pinned staging buffers into an ordinary (pageable) array.  Concurrent calls are serialised (shared staging)."""
    with _LOCK:
        return _run_host(x_host, t, target_points, lb, position, window_host, method, peak_width, target_coord, p0_only,
                         polish, chunk_bytes, device, timing, promote, pinned_result)


def _run_host(x_host, t, target_points, lb, position, window_host, method, peak_width, target_coord, p0_only, polish,
              chunk_bytes, device, timing, promote, pinned_result):
    import time

    import torch

    x_host = np.ascontiguousarray(x_host)
    if x_host.ndim != 2 or not np.issubdtype(x_host.dtype, np.complexfloating):
        raise ValueError("run_host expects a [n_batch, n_time] complex array")
    up_dt = torch.complex64 if x_host.dtype == np.complex64 else torch.complex128
    tdt = torch.complex128 if promote else up_dt
    nb, n_in = x_host.shape
    t0 = time.perf_counter()
    xd = torch.empty((nb, n_in), dtype=tdt, device=device)
    x_up = xd if up_dt == tdt else torch.empty((nb, n_in), dtype=up_dt, device=device)  # lands here, widened below
    plan = pl.make_plan(xd, t, target_points, lb, position, window_host=window_host)
    n = plan.n_out
    rd = torch.float32 if tdt == torch.complex64 else torch.float64
    row_bytes = n_in * x_host.itemsize
    rows = max(1, min(nb, chunk_bytes // row_bytes))
    chunks = [(lo, min(nb, lo + rows)) for lo in range(0, nb, rows)]
    compute = torch.cuda.current_stream(xd.device)
    copy_in = torch.cuda.Stream(device=xd.device)
    # xd / x_up come from the caching allocator of the COMPUTE stream: a recycled block may still have kernels queued
    # there (a just-freed intermediate of a preceding device op) -- the uploads must not overtake them (advisor, round 3)
    copy_in.wait_stream(compute)
    x_up.record_stream(copy_in)
    stage_key = (xd.device.index or 0, rows * row_bytes)
    stage = _STAGE.get(stage_key)
    if stage is None:
        stage = _STAGE[stage_key] = [torch.empty(rows * row_bytes, dtype=torch.uint8, pin_memory=True) for _ in range(3)]
    src_bytes = x_host.reshape(-1).view(np.uint8)
    dst_bytes = torch.view_as_real(x_up).reshape(-1).view(torch.uint8)
    stage_np = [s.numpy() for s in stage]
    absmax2 = torch.empty(nb, dtype=rd, device=xd.device)
    argidx = torch.empty(nb, dtype=torch.int32, device=xd.device)

    # ---- upload || pre-pass ---------------------------------------------------------------------
    pool = _pool()
    fills = {}

    def fill(k, lo_b, n_b):
        np.copyto(stage_np[k][:n_b], src_bytes[lo_b:lo_b + n_b])

    for i, (lo, hi) in enumerate(chunks[:len(stage)]):  # the first fills start at once
        fills[i] = pool.submit(fill, i % len(stage), lo * row_bytes, (hi - lo) * row_bytes)
    for i, (lo, hi) in enumerate(chunks):
        k = i % len(stage)
        fills.pop(i).result()
        n_b = (hi - lo) * row_bytes
        with torch.cuda.stream(copy_in):
            dst_bytes[lo * row_bytes:lo * row_bytes + n_b].copy_(stage[k][:n_b], non_blocking=True)
            landed = torch.cuda.Event()
            landed.record(copy_in)
        j = i + len(stage)
        if j < len(chunks):  # refill this staging buffer once its DMA is done (the worker waits, not this thread)
            lo2, hi2 = chunks[j]

            def refill(ev=landed, k=k, lo_b=lo2 * row_bytes, n_b2=(hi2 - lo2) * row_bytes):
                ev.synchronize()
                fill(k, lo_b, n_b2)

            fills[j] = pool.submit(refill)
        compute.wait_event(landed)
        if x_up is not xd:
            xd[lo:hi].copy_(x_up[lo:hi])
        dev.pipeline_fused(xd[lo:hi], n, plan.pad_left, window=plan.window, want_out=False, want_argmax=True,
                           absmax2=absmax2[lo:hi], argidx=argidx[lo:hi], argmax_value_only=True)
    t1 = time.perf_counter()

    # ---- selection + search (phasing.py:226-287) --------------------------------------------------
    sel = pl.Selection(xd, plan, absmax2, argidx, index_from_slice=True)
    res, _ = pl.select_and_solve(xd, plan, absmax2, argidx, method, peak_width, target_coord, p0_only, selection=sel,
                                 threads=pl.aps.burst_threads(), polish=polish)
    t2 = time.perf_counter()

    # ---- main pass || download ----------------------------------------------------------------------
    if pinned_result is None:
        pinned_result = os.environ.get("XMRIS_AMD_PINNED_RESULT", "1") != "0"
    out_host = None
    if pinned_result:
        try:
            out_host = torch.empty((nb, n), dtype=tdt, pin_memory=True)  # (recycled by torch's caching host allocator)
        except RuntimeError:  # no page-locked memory of that size to be had: staged download below
            out_host = None
    out_np = out_host.numpy() if out_host is not None else np.empty((nb, n), dtype=np.complex64 if tdt == torch.complex64
                                                                   else np.complex128)
    bounce = None if out_host is not None else [torch.empty((rows, n), dtype=tdt, pin_memory=True) for _ in range(2)]
    copy_out = torch.cuda.Stream(device=xd.device)
    copy_out.wait_stream(compute)  # (the ring below comes from the compute stream's allocator, as above)
    ring = [torch.empty((rows, n), dtype=tdt, device=xd.device) for _ in range(2)]
    for r_ in ring:
        r_.record_stream(copy_out)
    drained = [None, None]
    copies = [None, None]  # staged download: the host memcpy of the chunk that used this bounce buffer last
    for i, (lo, hi) in enumerate(chunks):
        b = i % 2
        if drained[b] is not None:
            compute.wait_event(drained[b])  # its previous contents are on their way out
        pl.main_pass(plan, xd[lo:hi], ring[b][:hi - lo], res.p0, res.p1, res.pivot)
        done = torch.cuda.Event()
        done.record(compute)
        if copies[b] is not None:
            copies[b].result()  # the bounce buffer is free again
        with torch.cuda.stream(copy_out):
            copy_out.wait_event(done)
            dst = out_host[lo:hi] if out_host is not None else bounce[b][:hi - lo]
            dst.copy_(ring[b][:hi - lo], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_out)
        drained[b] = ev
        if out_host is None:
            def unload(ev=ev, b=b, lo=lo, hi=hi):
                ev.synchronize()
                np.copyto(out_np[lo:hi], bounce[b][:hi - lo].numpy())

            copies[b] = pool.submit(unload)
    copy_out.synchronize()
    for c in copies:
        if c is not None:
            c.result()
    t3 = time.perf_counter()
    if timing is not None:
        timing.update(upload_prepass_s=t1 - t0, search_s=t2 - t1, main_download_s=t3 - t2, total_s=t3 - t0, chunks=len(chunks),
                      bytes=int(x_host.nbytes + out_np.nbytes), pinned_result=out_host is not None)
    return out_np, res, plan
