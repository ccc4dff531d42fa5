"""Array-level fused hot path:  zero_fill -> apodize_exp -> to_spectrum -> autophase.

This is what the four chained accessor calls of the reference's quick start
(``README.md:66-73``) amount to on the data, done in two data-parallel launches instead of six
full-size numpy passes:

  pre-pass   read the FIDs, FFT in LDS, emit only (max |X|^2, arg-max) per spectrum
  exchange   device reduce -> 16 bytes to the host (over ranks: caller-provided gather)
  solve      the one arg-max spectrum (64 KiB D2H) -> differential evolution on the host
  main pass  read the FIDs again, FFT, multiply by e^{i phi}, write the phased spectra

All coordinate arithmetic stays on the host in fp64, restating ``processing/fid.py:254-263``
(zero-fill coords), ``:136`` (window), ``processing/fourier.py:95-98, 31`` (frequency coords,
roll) and ``processing/phasing.py:226-247, 56-73`` (selection, phase ramp).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import autophase_solver as aps
from . import device as dev
from .dims import MSG_POSITION


@dataclass
class PipelinePlan:
    """Host-side metadata of one pipeline configuration (depends on coords, not on the data)."""

    n_in: int
    n_out: int
    pad_left: int
    time: np.ndarray          # zero-fill-extrapolated time coordinate (fid.py:257-263)
    freq: np.ndarray          # fftshifted frequency coordinate (fourier.py:98, 31-32)
    window_host: np.ndarray   # exp(-pi*lb*t) in fp64 (fid.py:136) or ones
    window: object = None     # device tensor, storage precision
    window64: object = None   # device tensor, float64 (for the complex128 recomputation of the arg-max slice)
    extra: dict = field(default_factory=dict)


def zero_fill_coords(t: np.ndarray, target_points: int, pad_left: int) -> np.ndarray:
    """fid.py:257-263."""
    delta = t[1] - t[0]
    if pad_left == 0:
        return t[0] + np.arange(target_points) * delta
    return (t[0] - (pad_left * delta)) + np.arange(target_points) * delta


def make_plan(x2, t: np.ndarray, target_points: int, lb, position: str = "end", window_host=None) -> PipelinePlan:
    """`window_host`: the apodisation weights over the zero-filled axis when they are not exp(-pi lb t) (`apodize_lg`)."""
    import torch

    n_in = x2.shape[-1]
    t = np.asarray(t, dtype=np.float64)
    if target_points <= n_in:  # fid.py:235-236: no-op zero fill
        n_out, pad_left, tt = n_in, 0, t
    else:
        n_out = int(target_points)
        if position == "end":
            pad_left = 0
        elif position == "symmetric":
            pad_left = (n_out - n_in) // 2
        else:
            raise ValueError(MSG_POSITION)
        tt = zero_fill_coords(t, n_out, pad_left) if len(t) > 1 else t
    win = np.exp(-np.pi * lb * tt) if lb is not None else np.ones(n_out)
    if window_host is not None:
        win = np.asarray(window_host, dtype=np.float64)
        if win.shape != (n_out,):
            raise ValueError("window_host must cover the zero-filled axis")
    delta = (tt[1] - tt[0]) if len(tt) > 1 else 1.0
    freq = np.roll(np.fft.fftfreq(n_out, d=delta), n_out // 2)
    rd = torch.float32 if x2.dtype == torch.complex64 else torch.float64
    wdev = torch.from_numpy(np.ascontiguousarray(win)).to(device=x2.device, dtype=rd)
    return PipelinePlan(n_in, n_out, pad_left, tt, freq, win, wdev)


@dataclass
class AutophaseResult:
    p0: float
    p1: float
    pivot: float
    flat_index: int
    target_idx: int
    max_abs: float
    nfev: int = 0
    fun: float = float("nan")
    timing: dict = field(default_factory=dict)
    owner: int = 0      # rank that owns the winning spectrum (run_stream)
    mine: bool = True   # ... and whether that is this rank
    speculation: str = ""  # run_stream(speculate=True): "hit" or "repaired"
    hedged: bool = False   # run_stream(speculate=True): the search ran late and was started a second time (see there)


def slice_on_host() -> bool:
    """Where the winning row's spectrum -- the slice the (p0, p1) search runs on (phasing.py:241-247) -- is computed.
    On the HOST (default): the row comes back as complex128 (64 KiB) and `winner_spectrum` restates the reference's own
    numpy statements on it, so the search sees the reference's slice BIT FOR BIT (numpy's pocketfft transforms a row of
    a batch exactly as it transforms the row alone) and, with generations and polish that replicate scipy's, returns
    the reference's (p0, p1).  `XMRIS_AMD_SLICE=device` keeps rounds 1-3's fp64 kernel (one workgroup, 17 us of the
    step; its spectrum differs from pocketfft's in the last bit, which the flat landscape of a noise-only dataset
    amplified to 3e-6 of the spectrum's maximum, profiles/r03/c1_tolerance.txt)."""
    import os

    return os.environ.get("XMRIS_AMD_SLICE", "host") != "device"


def winner_spectrum(plan: "PipelinePlan", row) -> np.ndarray:
    """zero_fill -> apodize -> ortho FFT -> fftshift of ONE row in numpy, statement for statement what the reference does
    to every row (fid.py:251 `da.pad(..., constant_values=0)`, fid.py:136-139 `da * weight` with float64 weights --
    numpy promotes a complex64 FID to complex128 there --, fourier.py:153 `np.fft.fftn(..., norm="ortho")`,
    fourier.py:31-32 `roll(n // 2)`)."""
    n_out, n_in, pl = plan.n_out, plan.n_in, plan.pad_left
    row = np.asarray(row).reshape(-1)[:n_in]
    buf = np.zeros(n_out, dtype=np.complex128)
    buf[pl:pl + n_in] = row
    if plan.window_host is not None:
        buf = buf * np.asarray(plan.window_host, dtype=np.float64)
    return np.roll(np.fft.fft(buf, norm="ortho"), n_out // 2)


class Selection:
    """Device-side selection stage of autophase, queued without any host synchronisation right behind a
    pre-pass: global arg-max reduction -> gather of the winning FID (row index read from device memory)
    -> its spectrum recomputed in complex128, with (max, flat index, slice) written by the kernels straight
    into pinned host memory, closed by an event.  `wait()` blocks on that event only, so kernels queued
    later on the same stream (another dataset's main pass) do not delay the host solver."""

    @staticmethod
    def new_slot(x2, plan: "PipelinePlan", real_dtype):
        """Pinned host buffers (max, flat index, fp64 spectrum) + the device staging row of one selection in flight."""
        import torch

        return (torch.empty(1, dtype=real_dtype, pin_memory=True), torch.empty(1, dtype=torch.int64, pin_memory=True),
                torch.empty((1, plan.n_out), dtype=torch.complex128, pin_memory=True),
                torch.empty((1, x2.shape[1]), dtype=torch.complex128, device=x2.device),
                torch.empty((1, x2.shape[1]), dtype=torch.complex128, pin_memory=True))  # (the winning FID, host-slice mode)

    def __init__(self, x2, plan: "PipelinePlan", absmax2, argidx, index_from_slice: bool = False, key=None, slot=None,
                 refine=None, blocking=None):
        import torch

        n = plan.n_out
        self.index_from_slice = index_from_slice  # pre-pass ran with argmax_value_only: only the ROW is known
        if plan.window64 is None:
            plan.window64 = torch.from_numpy(np.ascontiguousarray(plan.window_host)).to(x2.device, torch.float64)
        self.n = n
        # Results go straight into pinned host memory (device-accessible at the same address on ROCm): the
        # reduction writes (max, flat index) there, the gather reads the index back from there, and the
        # complex128 spectrum kernel stores its 128 KiB row there -- no memcpy nodes on the stream.  The
        # buffers are reused across datasets (two sets: a streaming caller keeps at most two selections
        # in flight); allocating pinned memory per call costs more than the transfers.
        rdt = absmax2.dtype if (key is None and refine is None) else torch.float32
        if slot is None:
            pool = plan.extra.setdefault("pinned", [])
            turn = plan.extra["turn"] = (plan.extra.get("turn", -1) + 1) % 2
            if len(pool) <= turn:
                pool.append(Selection.new_slot(x2, plan, rdt))
            slot = pool[turn]
        self.h_max, self.h_flat, self.h_slice, x1 = slot[:4]
        self.plan = plan
        self.h_row = slot[4] if (len(slot) > 4 and slice_on_host()) else None
        if self.h_row is not None:
            x1 = self.h_row  # (the kernels below write the winning FID straight into pinned host memory)
        if refine is not None:  # coarse estimates -> exact check of the candidates -> winner decoded + gathered
            window32, est, gkey, wkey, band = refine
            dev.guess_refine(x2, n, window32, est, gkey, wkey, self.h_max, self.h_flat, x1, band=band)
        elif key is not None:  # the producer left the winner in a 64-bit key: decode + gather in one small launch
            dev.argmax_key_take(key, n, self.h_max, self.h_flat, x2, out_row=x1)
        else:
            dev.argmax_reduce_async(absmax2, argidx, n, gmax=self.h_max, gflat=self.h_flat)
            dev.gather_row_c128(x2, self.h_flat, n, out=x1)
        if self.h_row is None:
            dev.pipeline_fused(x1, n, plan.pad_left, window=plan.window64, out=self.h_slice)
        self.event = torch.cuda.Event(blocking=aps.scarce_cpus() if blocking is None else blocking)
        self.event.record()
        self._done = False

    def wait(self):
        self.event.synchronize()
        if self.h_row is not None and not self._done:
            # (into the pinned slice buffer: the search service, the device search and the polish read it from there)
            self.h_slice[0].numpy()[:] = winner_spectrum(self.plan, self.h_row[0].numpy())
            self._done = True
        sl = self.h_slice[0].numpy().copy()
        flat = int(self.h_flat.item())
        if self.index_from_slice:  # index along the axis = first arg-max of the (fp64) winning spectrum
            flat = (flat // self.n) * self.n + int(np.argmax(np.abs(sl)))
        return float(self.h_max.item()) ** 0.5, flat, sl


def select_and_solve(x2, plan: PipelinePlan, absmax2, argidx, method="acme", peak_width=100, target_coord=None,
                     p0_only=False, exchange=None, rank_offset_rows=0, disp=False, on_host_phase=None,
                     selection: "Selection | None" = None, threads=None, polish="exact"):
    """phasing.py:226-287 on the outputs of the pre-pass.  `exchange(max_abs, flat)` may merge the
    per-rank winners (returns (owner_is_me, global_flat)); default = single device.
    `on_host_phase()` is called once the device has nothing left to do for this dataset until the
    solver returns (a streaming caller queues the next dataset's pre-pass there)."""
    n = plan.n_out
    sl_ready = None
    if selection is not None:  # everything was queued behind the pre-pass already
        amax, flat, sl_ready = selection.wait()
    else:
        amax, flat = dev.argmax_reduce(absmax2, argidx, n)
    gflat = rank_offset_rows * n + flat
    mine = True
    if exchange is not None:
        mine, gflat = exchange(amax, gflat)
    k = gflat % n
    if target_coord is not None:  # phasing.py:233-235
        target_idx = int(np.argmin(np.abs(plan.freq - target_coord)))
        pivot = float(target_coord)
    else:  # phasing.py:237-238
        target_idx = int(k)
        pivot = float(plan.freq[k])
    res = AutophaseResult(0.0, 0.0, pivot, int(gflat), target_idx, amax)
    if mine:
        # The optimiser is chaotic in its input (a 1e-7 perturbation of the slice can steer the search
        # into another local minimum of the ACME landscape), so the ONE spectrum it works on is
        # recomputed in complex128 from the stored samples, as the reference's float64 path would.
        import torch

        if sl_ready is not None:
            sl = sl_ready
        else:
            row = gflat // n - rank_offset_rows
            x1 = x2[row:row + 1].to(torch.complex128)
            if plan.window64 is None:
                plan.window64 = torch.from_numpy(np.ascontiguousarray(plan.window_host)).to(x2.device, torch.float64)
            w64 = plan.window64
            sl = dev.pipeline_fused(x1, n, plan.pad_left, window=w64).out[0].cpu().numpy()
        if on_host_phase is not None:
            on_host_phase()
        iw = aps.index_width_of(plan.freq, peak_width)
        p0, p1, opt = aps.solve(sl, plan.freq, pivot, target_idx, iw, method=method, p0_only=p0_only, disp=disp,
                                threads=threads, polish=polish)
        res.p0, res.p1, res.nfev, res.fun = p0, p1, int(opt.nfev), float(opt.fun)
        res.timing = {"generations_ms": 1e3 * opt.get("t_generations", 0.0), "polish_ms": 1e3 * opt.get("t_polish", 0.0)}
    elif on_host_phase is not None:
        on_host_phase()
    return res, mine


def run(x2, t, target_points: int, lb: float, method: str = "acme", peak_width=100, target_coord=None,
        p0_only: bool = False, out=None, plan: PipelinePlan | None = None, params=None, polish: str | None = None):
    """Fused hot path on ``x2`` = [n_batch, n_time] complex rows resident in HBM.

    Returns (phased [n_batch, n_out] tensor, AutophaseResult, plan).  `params=(p0, p1)` skips the
    solver (used by parity tests that inject the oracle's parameters).  `polish` (`XMRIS_AMD_POLISH` overrides the
    default): "exact" -- the projected-gradient test that scipy's polish starts (and usually ends) with is made
    natively, a search that does not pass it is polished on the reference's own route, scipy's minimiser on the numpy
    objective: (p0, p1) equal the reference's to the last bit also on flat landscapes; "numpy" -- always that route
    (~5-10 ms); "native" -- scipy's L-BFGS-B core on the native objective (~0.1 ms; its end point differs from the
    reference's by ~1e-3 degrees when the polish iterates).  `run_stream` has the same default."""
    import os

    import torch

    if polish is None:
        polish = os.environ.get("XMRIS_AMD_POLISH", "exact")

    if plan is None:
        plan = make_plan(x2, t, target_points, lb)
    n = plan.n_out
    # One dataset, own solve: where the guess stage and the arg-max key apply, the speculative schedule replaces the
    # FFT pre-pass (0.89 ms on the roofline shape) by coarse spectra + an exact check of the candidates (0.14 ms); the
    # main pass verifies the guess and a wrong one is repaired -- same result as the classic order below
    # (XMRIS_AMD_RUN_CLASSIC=1 keeps that order).
    if (params is None and target_coord is None and os.environ.get("XMRIS_AMD_RUN_CLASSIC") is None
            and x2.dim() == 2 and x2.is_contiguous() and plan.window is not None
            and dev.guess_supported(x2, n, plan.pad_left) and dev.key_native(x2, n, plan.pad_left)):
        if out is None:
            out = torch.empty((x2.shape[0], n), dtype=x2.dtype, device=x2.device)
        res = run_stream([x2], [out], plan, method=method, peak_width=peak_width, p0_only=p0_only, speculate=True,
                         polish=polish)[0]
        return out, res, plan
    # with a solve, only the winning ROW is needed from the pre-pass (its index along the axis comes from
    # the winning spectrum itself, recomputed in fp64); injected parameters need the full arg-max
    pre = dev.pipeline_fused(x2, n, plan.pad_left, window=plan.window, want_out=False, want_argmax=True,
                             argmax_value_only=params is None)
    if params is None:  # arg-max reduction, row gather, fp64 slice and D2H all queued without host syncs
        sel = Selection(x2, plan, pre.absmax2, pre.argidx, index_from_slice=True)
        res, _ = select_and_solve(x2, plan, pre.absmax2, pre.argidx, method, peak_width, target_coord, p0_only,
                                  selection=sel, threads=aps.burst_threads(), polish=polish)  # one search, nothing beside it
    else:
        res = _selection_only(pre, plan, target_coord)
    if params is not None:
        res.p0, res.p1 = float(params[0]), float(params[1])
    main = main_pass(plan, x2, out, res.p0, res.p1, res.pivot)
    return main.out, res, plan


def run_stream(inputs, outputs, plan: PipelinePlan, *, exchange=None, broadcast=None, rank_offset_rows: int = 0,
               overlap: bool = True, method: str = "acme", peak_width=100, target_coord=None, p0_only: bool = False,
               trace: list | None = None, speculate: bool = False, polish: str | None = None):
    """The fused hot path over a SEQUENCE of independent datasets of one shape, software-pipelined.

    ``inputs[i]`` ([n_batch, n_in] complex rows in HBM) is transformed into ``outputs[i]`` ([n_batch, n_out]);
    the lists may repeat tensors.  Per dataset the device does a pre-pass (per-spectrum max |X|^2), the
    selection stage (`Selection`) and the main pass; the host does the O(1) exchange, the (p0, p1) search on
    the winning spectrum and the phase table.  With `overlap` the pre-pass + selection of dataset i+1 is queued
    before the host starts searching for dataset i, so the device works on dataset i+1 (and on the main pass
    of dataset i-1) while the host searches -- every dataset still gets all of its own work, nothing is
    reused across datasets.

    Multi-device: `exchange(max_abs, global_flat) -> (owner_rank_is_me, winning_global_flat, owner)` merges
    the per-rank winners and `broadcast(values, owner) -> values` hands the owner's (p0, p1) to every rank
    (`xmris_amd.sharding`); `rank_offset_rows` = first global row of this rank's shard.

    `speculate=True` replaces the arg-max pre-pass (an FFT of every row, instruction bound) by a GUESS of the
    winning row from the windowed L1 norm of the FIDs (`xm_row_l1`, a streaming read at HBM speed:
    sum|z|/sqrt(N) bounds every |X[k]| of a row and equals the peak of a single decaying resonance).  (p0, p1)
    is searched on the guessed row's spectrum, the main pass applies it AND returns the true per-row maxima, and
    the true global arg-max row is compared with the guess before the next main pass is queued.  A wrong guess
    is repaired exactly: the true row's spectrum is fetched, (p0, p1) searched again and the dataset's main pass
    run again with them; the result equals the non-speculative schedule's.  How often the guess is right depends on the data (rows of similar
    spectral shape: always); correctness never does.

    `trace`, if given, receives one dict per dataset: host timestamps (`t_start`, `t_exchanged`, `t_solved`,
    `t_table`; speculative schedule: also `t_search_begin`, `t_search_end` on the search's thread and `t_collect`) and torch events around the two kernels (`pre0`, `pre1`, `main0`, `main1`).
    Returns the list of AutophaseResult (p0, p1 filled on every rank; `.speculation` = "hit" / "repaired" with
    `speculate`)."""
    import time

    import torch

    n_sets = len(inputs)
    if polish is None:
        import os

        polish = os.environ.get("XMRIS_AMD_POLISH", "exact")
    if n_sets != len(outputs):
        raise ValueError("inputs and outputs must have the same length")
    if n_sets == 0:
        return []
    # The loop is latency sensitive (one dataset every ~2 ms) and creates a little cyclic garbage per dataset; a
    # generation-2 collection over an interpreter that has torch and numpy loaded costs 10+ ms.  The cyclic
    # collector is paused for the duration of the call (reference counting still frees almost everything).
    import gc

    gc_was_enabled = gc.isenabled()
    gc.disable()
    try:
        return _run_stream(inputs, outputs, plan, exchange, broadcast, rank_offset_rows, overlap, method, peak_width,
                           target_coord, p0_only, trace, speculate, polish)
    finally:
        if gc_was_enabled:
            gc.enable()


def _run_stream(inputs, outputs, plan, exchange, broadcast, rank_offset_rows, overlap, method, peak_width,
                target_coord, p0_only, trace, speculate, polish="exact"):
    import time

    import torch

    n_sets = len(inputs)
    if speculate:
        if target_coord is not None:
            raise ValueError("speculate=True needs the arg-max pivot (target_coord=None)")
        return _run_stream_speculative(inputs, outputs, plan, exchange, broadcast, rank_offset_rows, overlap, method,
                                       peak_width, p0_only, trace, polish)
    n = plan.n_out
    x0 = inputs[0]
    rd = torch.float32 if x0.dtype == torch.complex64 else torch.float64
    bufs = plan.extra.get(("stream_bufs", x0.shape[0], str(rd)))
    if bufs is None:  # two sets of pre-pass outputs: dataset i+1's pre-pass runs while dataset i is being solved
        bufs = plan.extra[("stream_bufs", x0.shape[0], str(rd))] = (
            [torch.empty(x0.shape[0], dtype=rd, device=x0.device) for _ in range(2)],
            [torch.empty(x0.shape[0], dtype=torch.int32, device=x0.device) for _ in range(2)])
    absmax2, argidx = bufs
    sel = [None, None]
    events = [dict() for _ in range(n_sets)]

    def prepass(i):
        b = i & 1
        ev = events[i]
        if trace is not None:
            ev["pre0"], ev["pre1"] = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev["pre0"].record()
        dev.pipeline_fused(inputs[i], n, plan.pad_left, window=plan.window, want_out=False, want_argmax=True,
                           absmax2=absmax2[b], argidx=argidx[b], argmax_value_only=True)
        if trace is not None:
            ev["pre1"].record()
        # selection stage queued right behind it (device-side arg-max -> fp64 slice -> pinned host memory)
        sel[b] = Selection(inputs[i], plan, absmax2[b], argidx[b], index_from_slice=True)

    results = []
    prepass(0)
    for i in range(n_sets):
        b = i & 1
        ev = events[i]
        ev["t_start"] = time.perf_counter()
        owner_box = [0]

        def merged(amax, gflat):
            if exchange is None:
                ev["t_exchanged"] = time.perf_counter()
                return True, gflat
            mine, gwin, owner = exchange(amax, gflat)
            owner_box[0] = owner
            ev["t_exchanged"] = time.perf_counter()
            return mine, gwin

        def queue_next():
            if overlap and i + 1 < n_sets:
                prepass(i + 1)

        res, mine = select_and_solve(inputs[i], plan, absmax2[b], argidx[b], method, peak_width, target_coord,
                                     p0_only, exchange=merged, rank_offset_rows=rank_offset_rows,
                                     on_host_phase=queue_next, selection=sel[b], polish=polish)
        if broadcast is not None:
            res.p0, res.p1 = broadcast([res.p0, res.p1], owner_box[0])
        res.owner, res.mine = owner_box[0], mine
        ev["t_solved"] = ev["t_table"] = time.perf_counter()
        if trace is not None:
            ev["main0"], ev["main1"] = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev["main0"].record()
        main_pass(plan, inputs[i], outputs[i], res.p0, res.p1, res.pivot)
        if trace is not None:
            ev["main1"].record()
        if not overlap and i + 1 < n_sets:
            prepass(i + 1)
        results.append(res)
        if trace is not None:
            trace.append(ev)
    return results


def _search_workers(plan: PipelinePlan, n_rows: int, elem_bytes: int, threads: int | None = None):
    """(searches in flight, threads per search) for the streaming executor.  A search is O(1) per dataset on the host
    (measured, n_out = 8192, ACME: 3.2 / 1.9 / 1.1 / 0.76 ms of generations with 1 / 2 / 4 / 8 threads + 0.3 ms of
    polish; generations scale with n_out); the device period is the dataset's compulsory traffic at ~5.5 TB/s plus
    ~0.12 ms of small launches.  Three searches at a time with a third of the team each where that keeps up with the
    device; otherwise four with a quarter each (a smaller team spends fewer core-milliseconds per search).  Measured (16-CPU share; ms per step with 2 / 4 / 8 in flight): 16,384 x 2048 -> 4096: 0.73 / 0.60 /
    0.79, 32,768 x 1536: 0.49 / 0.47 / 0.66, 65,536 x 4096 -> 8192: 1.17 / 1.19 / 1.23 -- eight single-thread searches
    lose to the interpreter lock (every search ends in scipy's polish, ~0.3 ms of Python)."""
    import os

    def team_of(w):  # an explicit budget (tests, tuning) is divided evenly
        return max(1, threads // w) if given else _search_team(w)

    given = threads is not None
    if not given:
        threads = aps.stream_threads()
    if os.environ.get("XM_SEARCH_WORKERS"):  # tuning switch
        w = max(2, int(os.environ["XM_SEARCH_WORKERS"]))
        return w, team_of(w)
    device_ms = n_rows * (plan.n_in + plan.n_out) * elem_bytes / 5.5e9 + 0.12
    speedup = {1: 1.0, 2: 1.73, 4: 3.05, 8: 4.25, 16: 5.7}
    # Round 3: THREE in flight where the device paces the steps (teams of four out of twelve threads): a search then
    # has three device periods, ~2 ms of slack instead of ~1 for a search that runs late (a contended host), at the
    # same throughput on a quiet one -- six A/B pairs at K = 20: 51.7 vs 51.4 M spectra/s, three at K = 100: 55.35 vs
    # 55.20, 5.7 instead of 6.8 cores busy.
    w = 3
    team = max(1, threads // w)
    gain = speedup[max(k for k in speedup if k <= team)]
    if (0.3 + 3.2 * (plan.n_out / 8192.0) / gain) / w > 0.8 * device_ms and threads >= 4:
        w = 4
    return w, team_of(w)


_ABANDONED_RECORDS = []  # result records of searches nobody waits for any more (a hedged search's loser, a test hook's
                          # late submission): their searches still write them when they end, so they must not be freed


def _abandon(rec):
    _ABANDONED_RECORDS.append(rec)
    del _ABANDONED_RECORDS[:-256]


def _look_ahead(workers: int, n_sets: int, overlap: bool, polish: str) -> int:
    """Datasets whose searches are SUBMITTED ahead of the main pass being queued.  `workers` of them run side by side;
    with the native search service (polish="exact") three more wait in its queue: a search then ends five or six device
    periods before its result is needed instead of two, which is what hides the rare search whose polish has to run on
    the reference's route (scipy's minimiser on the numpy objective: 3-8 ms) -- on the heterogeneous dataset family
    one search in seven needs it, and with the polish done by the launch thread when the result was collected the rate
    fell from 46 to 33 M spectra/s (profiles/r04/hetero_steps.txt)."""
    if not overlap:
        return 0
    if n_sets <= 2:
        return 1
    import os

    extra = int(os.environ.get("XM_SEARCH_QUEUE_EXTRA", "3")) if polish == "exact" else 0  # (tuning switch)
    return min(workers + max(0, extra), n_sets - 1)


def _search_team(workers: int) -> int:
    """Threads per search with `workers` searches in flight (`autophase_solver.stream_threads`: more than three in
    flight means the host paces the steps)."""
    return max(1, aps.stream_threads(host_paced=workers > 3) // max(1, workers))


def _run_stream_speculative(inputs, outputs, plan, exchange, broadcast, rank_offset_rows, overlap, method,
                            peak_width, p0_only, trace, polish="exact"):
    """`run_stream(speculate=True)`: guess pass -> search -> main pass with true maxima -> verify (-> repair).

    Software pipeline over the datasets (with `overlap`): while the main pass of dataset i is queued, the guess
    kernels + selection stages of datasets up to i+4 are already on the stream and the (p0, p1) searches of
    datasets i+1 ... i+3 run on worker threads (each search is one native call that releases the GIL and brings its
    own small team, `xm_solver_de`), so a search has three device periods to finish instead of racing one
    (`_search_workers`: three in flight where the device paces the steps, four where the host does).  Every
    dataset still gets all of its own work; the collective-like calls (`exchange`, `broadcast`) are made by this
    thread in dataset order, identically on every rank."""
    import os
    import time
    from concurrent.futures import ThreadPoolExecutor

    import torch

    n_sets, n = len(inputs), plan.n_out
    x0 = inputs[0]
    nb = x0.shape[0]
    rd = torch.float32 if x0.dtype == torch.complex64 else torch.float64
    # searches running ahead of the main pass being queued (= worker threads): two where the device period is longer
    # than a search, more -- with smaller teams, which use the cores better -- where the host would pace the steps
    workers, team = _search_workers(plan, nb, x0.element_size())
    # ---- search engine ------------------------------------------------------------------------------------------
    # "device": the search of a dataset runs as ONE workgroup on a side stream right behind
    # the dataset's selection stage (`xm_search_launch`, csrc/xm_search.hip: scipy's generations bit for bit, then the
    # projected-gradient test scipy's polish starts with) -- no host core computes anything, no team spins, and what a
    # shared host does to its threads no longer reaches the device's schedule; a search that does not pass the test is
    # polished by this thread on the reference's route.  A search takes milliseconds on its one CU, so the guess stages
    # run `dev_ahead` datasets in front, and the first datasets of a call (the pipeline is still filling: nothing hides
    # a search there) are searched by the host engine as before.  "host": round 3's worker threads + native teams.
    axis = plan.extra.get("uniform_axis")
    if axis is None:
        axis = plan.extra["uniform_axis"] = dev.uniform_axis(plan.freq) or False
    # Which one?  Measured on one rank with 16 CPUs (profiles/r04/search_engines.txt): the host engine is FASTER -- a
    # search kernel needs a whole CU's registers for 2.5-4 ms, so the chip is split (`xm_stream_create`), and the
    # streaming kernels lose more than the CUs' share (65,536 x 4096 -> 8192: 52.3 vs 48.0 M spectra/s; 16,384 x 2048
    # -> 4096: 0.35 vs 0.49 ms per dataset).  What the device engine buys is independence from the host: 2 instead of
    # 6 busy cores, and a schedule that a contended or core-starved host cannot disturb.  "auto" therefore takes it
    # only where the host cannot carry the searches: fewer than TWO CPUs per rank of this node (searches are per
    # dataset, not per rank: eight ranks on 16 CPUs still carry them -- one core per launch thread, eight for the
    # teams, profiles/r03/rehearsal_6ranks.txt).
    want = os.environ.get("XMRIS_AMD_SEARCH", "auto")
    if want == "auto":
        local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
        # ... and only where every rank has a GPU of its own: the partition is made of CU-masked queues, and those of
        # several processes on ONE card reserve the same CUs and oversubscribe its hardware queues -- six ranks sharing
        # a GPU fell into the scheduler's 10.7 ms process time slices, 124 instead of 1.4 ms per step
        # (profiles/r04/rehearsal_6ranks.txt; four ranks still ran at full speed)
        own_gpu = torch.cuda.device_count() >= local_world
        want = "device" if (aps._cpu_share() < 2 * local_world and own_gpu) else "host"
    use_dev = (want == "device" and polish == "exact" and method == "acme"
               and overlap and axis is not False and dev.search_supported(n, method, axis[2]))
    dev_ahead = 0
    if use_dev:
        # ~600 objective evaluations; measured per evaluation: 2.3 us + 0.4 us per 1000 bins (profiles/r04/device_search.txt)
        est_ms = 600 * (2.3 + 0.4 * n / 1000.0) * 1e-3
        device_ms = nb * (plan.n_in + plan.n_out) * x0.element_size() / 5.5e9 + 0.12
        dev_ahead = int(min(24, max(3, -(-est_ms // device_ms) + 2)))
        if os.environ.get("XM_SEARCH_AHEAD"):  # tuning switch
            dev_ahead = max(1, int(os.environ["XM_SEARCH_AHEAD"]))
    if exchange is not None:
        # several ranks: the look-ahead fixes the ORDER of the exchange calls, which every rank must make alike --
        # it may not depend on anything a rank measures or owns (its shard size, its share of the host's cores).
        # Rank 0's choice goes to everyone (one more broadcast at the start of the call); without a broadcast
        # callable: two.
        if broadcast is not None:
            got = broadcast([float(workers), float(dev_ahead if use_dev else 0)], 0)
            workers, use_dev, dev_ahead = int(round(got[0])), use_dev and got[1] > 0, int(round(got[1]))
        else:
            workers, use_dev, dev_ahead = 2, False, 0
    s_ahead = _look_ahead(workers, n_sets, overlap, polish)
    cpu_fill = min(n_sets, min(workers, s_ahead) + 1) if use_dev else n_sets
    use_dev = use_dev and cpu_fill < n_sets
    eng = dict(workers=workers, team=team, use_dev=use_dev, dev_ahead=dev_ahead, est_ms=est_ms if use_dev else 0.0,
               axis=axis, search_streams=None)
    args = (inputs, outputs, plan, exchange, broadcast, rank_offset_rows, overlap, method, peak_width, p0_only, trace,
            polish, eng)
    if os.environ.get("XM_FORCE_PARTITION") and not use_dev:  # tuning switch: the compute partition without any search kernel
        part = dev.chip_partition(x0.device, int(os.environ["XM_FORCE_PARTITION"]), n_search=1)
        caller = torch.cuda.current_stream(x0.device)
        part.compute.wait_stream(caller)
        try:
            with torch.cuda.stream(part.compute):
                return _spec_loop(*args)
        finally:
            caller.wait_stream(part.compute)
    if not use_dev or os.environ.get("XM_SEARCH_PARTITION", "1") == "0":  # (tuning switch: searches share the chip)
        return _spec_loop(*args)
    # Search kernels need a whole CU's registers for milliseconds, the streaming kernels are persistent grids sized to
    # fill every CU: sharing one pool, a search waits for a kernel boundary to start and the main pass then finds CUs
    # taken (measured: main pass +7 %, stalls of milliseconds).  So the chip is split for the duration of the call --
    # `reserved` CUs, spread over the eight XCDs, for the searches; the streaming kernels run on a stream that owns
    # the rest and size their grids by it (`xm_stream_create`).
    reserved = int(os.environ.get("XM_SEARCH_CUS", "0")) or int(min(32, 8 * -(-max(dev_ahead, 1) // 8)))
    part = dev.chip_partition(x0.device, reserved, n_search=min(16, max(dev_ahead, 1) + 2))
    eng["search_streams"] = part.search
    caller = torch.cuda.current_stream(x0.device)
    part.compute.wait_stream(caller)
    try:
        with torch.cuda.stream(part.compute):
            return _spec_loop(*args)
    finally:
        caller.wait_stream(part.compute)


def _spec_loop(inputs, outputs, plan, exchange, broadcast, rank_offset_rows, overlap, method, peak_width, p0_only, trace,
               polish, eng):
    """The software pipeline of `_run_stream_speculative` (engine and look-ahead chosen there)."""
    import os
    import time

    t_call = time.perf_counter()
    from concurrent.futures import ThreadPoolExecutor

    import torch

    n_sets, n = len(inputs), plan.n_out
    x0 = inputs[0]
    nb = x0.shape[0]
    rd = torch.float32 if x0.dtype == torch.complex64 else torch.float64
    workers, team, use_dev, dev_ahead, est_ms, axis = (eng["workers"], eng["team"], eng["use_dev"], eng["dev_ahead"],
                                                        eng["est_ms"], eng["axis"])
    s_ahead = _look_ahead(workers, n_sets, overlap, polish)
    # (round 3: the selection stage of dataset j on a stream of its own beside the coarse spectra of dataset j + 1,
    # gated so that it never shares the chip with a main pass, hides nothing -- the coarse-spectra kernel fills the
    # chip, 1.20 vs 1.20 ms per step; let loose beside the main kernel it costs 6 %.  Only the winner's fp64 spectrum
    # (one workgroup, 18 us) on a high-priority stream beside the next main pass: that pass gets 15 us longer, -0.6 % at
    # K = 20 and +0.4 % at K = 100 over six A/B pairs.  Everything stays on one stream.)
    g_ahead = s_ahead + 1 if overlap else 0                # guess kernels queued ahead of it
    # device engine: the first `cpu_fill` datasets of the call are searched by the host engine (s_ahead at a time);
    # from then on a dataset's search is a kernel that starts `dev_ahead` datasets before its main pass
    cpu_fill = min(n_sets, min(workers, s_ahead) + 1) if use_dev else n_sets
    use_dev = use_dev and cpu_fill < n_sets
    if use_dev:
        g_ahead = max(g_ahead, min(dev_ahead, n_sets - 1))
    ring = g_ahead + 2
    distinct = list({id(x): x for x in inputs}.values())

    def geometry(x):  # (key_native, guess_supported) of a dataset: they depend on its alignment, shape and dtype only
        k = ("geometry", x.data_ptr() & 15, tuple(x.shape), str(x.dtype), x.is_contiguous())
        g = plan.extra.get(k)
        if g is None:
            g = plan.extra[k] = (dev.key_native(x, plan.n_out, plan.pad_left), dev.guess_supported(x, plan.n_out, plan.pad_left))
        return g

    # the main pass leaves its true global arg-max in a key (no per-row arrays): every dataset must take that kernel
    use_keys = all(geometry(x)[0] for x in distinct)
    c128 = x0.dtype == torch.complex128
    # guess stage: coarse spectra + exact check of the candidates (both precisions); else the windowed L1 norm's winner
    use_guess = (os.environ.get("XM_GUESS_L1") is None and plan.window is not None
                 and all(geometry(x)[1] for x in distinct))
    # Candidate band of the guess stage.  A coarse spectrum (first 512 samples, 1024 bins) underestimates a line's height
    # by the part of its windowed FID beyond sample 512 -- at most the window's own weight out there, for a line that
    # does not decay by itself -- and by the grid's scalloping (>= 0.9 for a 2x zero-filled truncated line): rows whose
    # estimate reaches 0.9 x (window weight inside the first 512 samples) of the largest estimate are checked exactly
    # (lb = 5 Hz at 5 kHz: 0.72; measured on the heterogeneous family: estimates within [0.83, 0.97] of the true peaks,
    # scripts/study_guess_statistics.py).  A wider band only costs exact transforms (<= 16 per workgroup).
    band = plan.extra.get("guess_band")
    if band is None:
        wabs = np.abs(np.asarray(plan.window_host, dtype=np.float64)[plan.pad_left:plan.pad_left + plan.n_in])
        inside = float(wabs[:512].sum()) / max(float(wabs.sum()), 1e-300)
        # ... floored at 0.4: the window bound assumes a line that does not decay by itself; measured on the
        # heterogeneous family WITHOUT apodisation (lb = 0, inside = 0.125): 12/12 hits with 0.25 and with 0.4 (device
        # period 1.43 / 1.31 ms: a wide band costs exact transforms), 11/12 with 0.5 (scripts/time_hetero_lb.py)
        band = plan.extra["guess_band"] = float(min(0.95, max(0.4, 0.9 * inside)))
    if os.environ.get("XM_GUESS_BAND"):  # tuning switch
        band = float(os.environ["XM_GUESS_BAND"])
    l1_keys = use_keys and not c128 and not use_guess  # round 2's guess stage leaves its winner in a key (complex64)
    # (use_keys / l1_keys depend on the alignment of every input: buffers cached for one combination hold None where
    # another needs arrays -- advisor, round 3)
    key = ("spec_bufs", nb, str(rd), ring, use_guess, use_keys, l1_keys)
    bufs = plan.extra.get(key)
    if bufs is None:
        sel_rd = torch.float32 if use_guess else rd
        bufs = plan.extra[key] = dict(
            norm=[None if (l1_keys or use_guess) else torch.empty(nb, dtype=rd, device=x0.device) for _ in range(ring)],
            est=[torch.empty(nb, dtype=torch.float32, device=x0.device) if use_guess else None for _ in range(ring)],
            wkey=dev.new_argmax_key(x0.device) if use_guess else None,
            zero_idx=torch.zeros(nb, dtype=torch.int32, device=x0.device),
            tmax=[None if use_keys else torch.empty(nb, dtype=rd, device=x0.device) for _ in range(ring)],
            tidx=[None if use_keys else torch.empty(nb, dtype=torch.int32, device=x0.device) for _ in range(ring)],
            vmax=[torch.empty(1, dtype=rd, pin_memory=True) for _ in range(ring)],
            vflat=[torch.empty(1, dtype=torch.int64, pin_memory=True) for _ in range(ring)],
            # complex64: the guess kernel and the main kernel leave their winners in arg-max key buffers (no per-row
            # arrays, no separate reductions); every key is cleared by the launch that decodes it
            gkey=[dev.new_argmax_key(x0.device) if (use_keys or use_guess) else None for _ in range(ring)],
            vkey=[dev.new_argmax_key(x0.device) if use_keys else None for _ in range(ring)],
            vres=[dev.new_key_result() if use_keys else None for _ in range(ring)],
            sel_slots=[Selection.new_slot(x0, plan, sel_rd) for _ in range(ring)])
    if use_guess and plan.extra.get("window32") is None:
        plan.extra["window32"] = plan.window.to(torch.float32).contiguous()
    sel = [None] * ring
    events = [dict() for _ in range(n_sets)]
    results = [None] * n_sets
    blocking = aps.scarce_cpus()
    iw = aps.index_width_of(plan.freq, peak_width)
    # The guess needs a ranking, not the norm itself: samples whose window weight is negligible are not read.
    # n_used = the leading samples that carry all but 1e-3 of the window's total weight (rounded up to 256).
    n_used = plan.extra.get("guess_n_used")
    if n_used is None:
        wabs = np.abs(np.asarray(plan.window_host, dtype=np.float64)[plan.pad_left:plan.pad_left + plan.n_in])
        total = float(wabs.sum())
        n_used = plan.n_in
        if total > 0:
            n_used = int(np.searchsorted(np.cumsum(wabs), (1.0 - 1e-3) * total)) + 1
            n_used = min(plan.n_in, max(256, -(-n_used // 256) * 256))
        plan.extra["guess_n_used"] = n_used
    # ... and of those, every `sub_step`-th 1-KiB block (default 8: whole cache lines at the start, middle and end of the window's support): the
    # guess is verified by the main pass anyway, and a regular subset ranks rows like the full sum does
    sub_step = plan.extra.get("guess_sub_step")
    if sub_step is None:
        sub_step = plan.extra["guess_sub_step"] = max(1, int(os.environ.get("XM_GUESS_SUBSTEP", "8")))
    # searches in flight at once share the host: each gets an equal part of the team
    n_workers = min(workers, s_ahead) if s_ahead >= 2 else 0  # searches RUNNING side by side (more may be queued)
    team = _search_team(n_workers) if n_workers else aps.stream_threads()  # per search in flight
    pool = None
    if n_workers:
        pool = plan.extra.get(("search_pool", n_workers))
        if pool is None:
            # (one thread more than searches in flight: a search that was hedged keeps its thread until it ends)
            pool = plan.extra[("search_pool", n_workers)] = ThreadPoolExecutor(max_workers=n_workers + 1,
                                                                              thread_name_prefix="xm-search")

    # The host engine's searches run on NATIVE threads of the library (`xm_hostsearch_submit`: generations + the
    # projected-gradient test, the result in a record this thread polls) -- a search on a Python thread costs 60-100 us
    # of interpreter under the lock this thread needs to queue kernels, which is what paced the small configurations.
    # Python threads remain for the other polish modes ("native" / "numpy": scipy's minimiser in the loop).
    use_service = polish == "exact" and not os.environ.get("XM_SEARCH_PYTHON_THREADS")
    hsearch = None
    if use_service:
        hsearch = plan.extra.get(("host_search", ring))
        if hsearch is None:
            hsearch = plan.extra[("host_search", ring)] = dict(
                recs=[torch.zeros(dev.SEARCH_RECORD_WORDS, dtype=torch.int64) for _ in range(ring)], seq=[0])
        if plan.extra.get("freq_c") is None:
            plan.extra["freq_c"] = np.ascontiguousarray(plan.freq, dtype=np.float64)
        from . import _lib as _lib_mod

        # (as many as run side by side: with spinning teams, one search more than the thread budget was cut for
        # oversubscribes the cores; a hedged second start raises the cap for itself)
        _lib_mod.load().xm_hostsearch_set_workers(max(1, n_workers))
    # Searches that do not pass scipy's projected-gradient test are polished on the reference's route (numpy objective,
    # milliseconds of interpreter): a helper thread starts on that as soon as the search's record says so -- the launch
    # thread looks at the records of the searches in flight once per dataset -- instead of the launch thread doing it
    # when it needs the result.
    # ... and that helper is a worker PROCESS (`autophase_solver.PolishWorkers`): a polish is milliseconds of small numpy
    # operations, and on helper THREADS they were taken out of this thread's share of the interpreter lock -- on the
    # heterogeneous family, where 13 searches of 16 need the polish, the launch thread fell from 1.4 to 2.3 ms per
    # dataset (profiles/r04/hetero_polish.txt).  XM_POLISH_THREADS=1 keeps them on threads of this process.
    polish_pool = plan.extra.get("polish_pool")
    if polish_pool is None:
        if os.environ.get("XM_POLISH_THREADS"):
            polish_pool = ThreadPoolExecutor(max_workers=max(1, int(os.environ["XM_POLISH_THREADS"])), thread_name_prefix="xm-polish")
        elif n_sets > 4:  # (a stream: the workers start now, in the background -- a process start + scipy's import is ~1 s)
            polish_pool = aps.polish_workers()
        if polish_pool is not None:
            plan.extra["polish_pool"] = polish_pool
    polish_futs = {}

    def advance_polishes():
        for j, (_, fut_j, _) in pending.items():
            if j in polish_futs or not (isinstance(fut_j, tuple) and len(fut_j) == 2 and fut_j[0] in ("host", "dev")):
                continue
            rec = (hsearch if fut_j[0] == "host" else dsearch)["recs"][j % ring]
            if not dev.search_done(rec, fut_j[1]):
                continue
            r = dev.read_search_record(rec)
            if not r["needs_polish"]:
                polish_futs[j] = None
                continue
            k = r["target_idx"]
            sl = sel[j % ring].h_slice[0].numpy().copy()
            args = (sl, plan.freq, float(plan.freq[k]), k, iw, method, p0_only, r["x"])
            if polish_pool is None:  # (a short call: the launch thread polishes when it collects the result)
                polish_futs[j] = None
            else:
                polish_futs[j] = (polish_pool.submit(aps.polish_reference, *args) if isinstance(polish_pool, ThreadPoolExecutor)
                                  else polish_pool.submit(*args))

    def polished(i, r, k, b):
        """(x, fun, nfev of the polish) of a search that needs one: the helper thread's, or done here."""
        fut = polish_futs.pop(i, None)
        if fut is not None:
            x, fun, nfev_p, _ = fut.result()
            return x, fun, nfev_p
        sl = sel[b].h_slice[0].numpy().copy()
        x, fun, nfev_p, _ = aps.polish_reference(sl, plan.freq, float(plan.freq[k]), k, iw, method, p0_only, r["x"])
        return x, fun, nfev_p

    def submit_host_search(j, k, threads, rec):
        """`xm_hostsearch_submit` of dataset j's slice (pinned, in its selection slot); returns the sequence number."""
        from . import _lib

        hsearch["seq"][0] += 1
        seq = hsearch["seq"][0]
        rec[7] = 0
        _lib.call("xm_hostsearch_submit", sel[j % ring].h_slice.data_ptr(), n, plan.extra["freq_c"].ctypes.data,
                  aps.METHODS.index(method), int(k), int(iw), int(bool(p0_only)), 42, 0.01, 1000, int(threads), seq,
                  rec.data_ptr())
        return seq

    def collect_host(i, seq, k, ev):
        """Result of dataset i's search on the service -> (p0, p1, nfev, fun, timing, hedged); a search later than twice
        the typical time is submitted a second time with the whole team and the first to finish is taken (see `collect`)."""
        b = i % ring
        rec = hsearch["recs"][b]
        recent = (fill_hist if i == 0 else run_hist)[-9:]
        if len(recent) < 3:  # (see `collect`: no history yet, or -- several ranks -- never)
            gain = {1: 1.0, 2: 1.73, 4: 3.05, 8: 4.25, 16: 5.7}
            th = max(1, fill_team if i == 0 else team)
            # (x 3: a process's first searches also start the service's threads and their teams)
            recent = [3e-3 * (0.3 + 3.2 * (plan.n_out / 8192.0) / gain[max(k_ for k_ in gain if k_ <= th)]) + 1e-3]
        typical = sorted(recent)[len(recent) // 2]
        deadline = ev["t_exchanged"] + 2.0 * typical + 0.5e-3
        can_hedge = hedging and i - last_hedge[0] >= hedge_gap
        second, hedged, nap = None, False, 0.0
        give_up = time.perf_counter() + 120.0
        while True:
            if dev.search_done(rec, seq):
                break
            if second is not None and dev.search_done(second[0], second[1]):
                # the first submission is still on its way: it will write its record whenever it ends, so that record
                # leaves the ring (a later dataset's result in the same slot must not be overwritten by it)
                _abandon(rec)
                hsearch["recs"][b] = torch.zeros(dev.SEARCH_RECORD_WORDS, dtype=torch.int64)
                rec = second[0]
                break
            now = time.perf_counter()
            if can_hedge and second is None and now > deadline:
                last_hedge[0] = i
                spare = torch.zeros(dev.SEARCH_RECORD_WORDS, dtype=torch.int64)  # (its own record: see above)
                _abandon(spare)
                _lib_mod.load().xm_hostsearch_set_workers(max(1, n_workers) + 1)  # (it must not wait in the queue)
                second = (spare, submit_host_search(i, k, fill_team, spare))
                _lib_mod.load().xm_hostsearch_set_workers(max(1, n_workers))
                hedged = True
            if now > give_up:
                raise RuntimeError("the search service did not answer within two minutes")
            if blocking:
                nap = min(1e-4, nap + 1e-5)
                time.sleep(nap)
            # (no time.sleep(0) here to hand over the interpreter lock: tried at the end of round 4 -- with the search teams
            # spinning on every core of the quota a yielding launch thread lost its CPU for 2-3 ms at a time, K = 20
            # went from 1.22 to 1.39-1.51 ms per step in two collections)
        r = dev.read_search_record(rec)
        hist = fill_hist if i == 0 else run_hist
        hist.append(1e-6 * r["t_us"][5])
        del hist[:-16]
        p0, p1 = r["x"]
        nfev, fun = r["nfev"] + (1 if p0_only else 2) + 1, r["fun"]
        timing = {"generations_ms": 1e-3 * r["t_us"][0], "polish_ms": 1e-3 * (r["t_us"][5] - r["t_us"][0])}
        if r["needs_polish"]:
            t0 = time.perf_counter()
            x, fun, nfev_p = polished(i, r, k, b)
            p0, p1 = float(x[0]), (float(x[1]) if not p0_only else 0.0)
            nfev = r["nfev"] + nfev_p
            timing["polish_ms"] = 1e3 * (time.perf_counter() - t0)  # (what the launch thread still waited for)
            timing["polish_route"] = "numpy"
        else:
            polish_futs.pop(i, None)
        return p0, (p1 if not p0_only else 0.0), nfev, fun, timing, hedged

    # (tuning switch XM_GUESS_STREAM=1: the guess + selection chain on a stream of its own beside the main passes.
    # Round 3 measured it on the roofline shape: -7 %, the coarse-spectra kernel fills the chip and the main kernel
    # slows down beside it; round 4 on the small shapes, whose 0.17 ms main passes leave ramp-up and tail bubbles:
    # profiles/r04/guess_stream.txt.)
    guess_stream = None
    if os.environ.get("XM_GUESS_STREAM", "0") not in ("", "0") and torch.cuda.is_available():
        guess_stream = plan.extra.get("guess_stream")
        if guess_stream is None:
            guess_stream = plan.extra["guess_stream"] = torch.cuda.Stream(device=x0.device)
        guess_stream.wait_stream(torch.cuda.current_stream(x0.device))  # (the inputs are ready on the caller's stream)

    def guess(j):
        if guess_stream is None:
            return guess_on_current_stream(j)
        with torch.cuda.stream(guess_stream):
            return guess_on_current_stream(j)

    def guess_on_current_stream(j):  # coarse spectra (or streaming L1 norms) + the selection stage on the winning row
        b = j % ring
        ev = events[j]
        if trace is not None:
            ev["pre0"], ev["pre1"] = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev["pre0"].record()
        if use_guess:
            dev.guess_rows(inputs[j], n, plan.extra["window32"], bufs["est"][b], bufs["gkey"][b])
        else:
            dev.row_l1(inputs[j], plan.window, plan.pad_left, out=bufs["norm"][b], n_used=n_used, sub_step=sub_step,
                       key=bufs["gkey"][b] if l1_keys else None)
        if trace is not None:
            ev["pre1"].record()
            ev["guess_kernel"] = dev.last_kernel()
        sel[b] = Selection(inputs[j], plan, bufs["norm"][b], bufs["zero_idx"], index_from_slice=True,
                           key=bufs["gkey"][b] if l1_keys else None, slot=bufs["sel_slots"][b],
                           refine=(plan.extra["window32"], bufs["est"][b], bufs["gkey"][b], bufs["wkey"], band)
                           if use_guess else None, blocking=blocking)
        if trace is not None:  # (the selection stage: exact check of the candidates + the winner's fp64 spectrum)
            ev["sel1"] = torch.cuda.Event(enable_timing=True)
            ev["sel1"].record()
        if use_dev and exchange is None and j >= cpu_fill and not slice_on_host():
            # one rank, slice computed by the device: the winner needs no exchange -- the search kernel is queued at
            # once, gated on the selection stage by an event; this thread does not wait for either
            dev_seq[j] = launch_dev_search(j, after=sel[b].event)

    # ---- searches on the device (`use_dev`) ------------------------------------------------------------------------
    dsearch, dev_seq = None, {}
    if use_dev:
        dsearch = plan.extra.get(("dev_search", ring))
        if dsearch is None:
            dsearch = plan.extra[("dev_search", ring)] = dict(
                recs=[dev.new_search_record() for _ in range(ring)],
                streams=[torch.cuda.Stream(device=x0.device) for _ in range(min(ring, 16))], seq=[0])
        if eng["search_streams"] is not None:  # the search partition of the chip (see _run_stream_speculative)
            dsearch["streams"] = list(eng["search_streams"])

    def launch_dev_search(j, after=None):
        """`xm_search_launch` for dataset j on a side stream; returns the sequence number its record will carry."""
        b = j % ring
        st = dsearch["streams"][j % len(dsearch["streams"])]
        if after is not None:
            st.wait_event(after)
        dsearch["seq"][0] += 1
        seq = dsearch["seq"][0]
        dev.search_launch(sel[b].h_slice[0], axis, dsearch["recs"][b], seq, p0_only=p0_only, stream=st)
        events[j]["t_search_begin"] = time.perf_counter()
        return seq

    def collect_dev(i, seq, ev):
        """Result of dataset i's search kernel -> (p0, p1, k, nfev, fun, timing, hedged).  A search that does not pass
        scipy's projected-gradient test is polished here on the reference's route; one that runs far beyond the usual
        time (a landscape that keeps the generations going for tens of thousands of evaluations) is overtaken by the
        host engine -- the search is a pure function of the slice."""
        b = i % ring
        rec = dsearch["recs"][b]
        deadline = time.perf_counter() + max(4.0 * est_ms * 1e-3, 8e-3)
        nap = 0.0
        while not dev.search_done(rec, seq):  # (bounded: past the deadline the host engine takes over)
            if time.perf_counter() > deadline:
                sl = sel[b].h_slice[0].numpy().copy()
                k = int(np.argmax(np.abs(sl)))
                p0, p1, opt = search(sl, k, float(plan.freq[k]), fill_team, {})
                # the kernel is still running: its record and its stream are retired (it will write the record when it
                # ends; the stream's later searches would queue behind it)
                _abandon((rec, dsearch["streams"][i % len(dsearch["streams"])]))
                dsearch["recs"][b] = dev.new_search_record()
                dsearch["streams"][i % len(dsearch["streams"])] = dev.replacement_search_stream(x0.device, eng["search_streams"])
                return p0, p1, k, int(opt.nfev), float(opt.fun), {"generations_ms": 1e3 * opt.get("t_generations", 0.0),
                                                                 "polish_ms": 1e3 * opt.get("t_polish", 0.0)}, True
            if blocking:  # few cores per rank: do not spin beside another rank's launch thread
                nap = min(1e-4, nap + 1e-5)
                time.sleep(nap)
        ev["t_search_end"] = time.perf_counter()
        r = dev.read_search_record(rec)
        k = r["target_idx"]
        p0, p1 = r["x"]
        nfev, fun = r["nfev"] + (1 if p0_only else 2) + 1, r["fun"]  # (+ the gradient test's evaluations, like scipy's count)
        timing = {"generations_ms": 1e-3 * r["t_us"][5], "polish_ms": 0.0, "device": True}
        if r["needs_polish"]:
            t0 = time.perf_counter()
            x, fun, nfev_p = polished(i, r, k, b)
            p0, p1 = float(x[0]), (float(x[1]) if not p0_only else 0.0)
            nfev = r["nfev"] + nfev_p
            timing["polish_ms"] = 1e3 * (time.perf_counter() - t0)
            timing["polish_route"] = "numpy"
        else:
            polish_futs.pop(i, None)
        return p0, (p1 if not p0_only else 0.0), k, nfev, fun, timing, False

    # the pipeline-filling search (the first main pass waits for it) takes the whole CPU share for its millisecond:
    # four A/B pairs at the driver's K = 20: 53.1 -> 53.8 M spectra/s
    fill_team = aps.burst_threads()

    slow_hook = os.environ.get("XM_TEST_SLOW_SEARCH")  # test hook "<dataset>,<ms>": that dataset's search thread naps first
    slow_j, slow_ms = (int(slow_hook.split(",")[0]), float(slow_hook.split(",")[1])) if slow_hook else (-1, 0.0)

    def search(sl, k, pivot, threads, ev, j=-1):
        ev["t_search_begin"] = time.perf_counter()  # (on the worker thread: dispatch latency = this - t_exchanged)
        if j == slow_j and j >= 0:
            time.sleep(slow_ms * 1e-3)
        out = aps.solve(sl, plan.freq, pivot, k, iw, method=method, p0_only=p0_only, threads=threads, polish=polish)
        ev["t_search_end"] = time.perf_counter()
        return out

    # Hedged searches.  A search is O(1) work on ONE Python thread plus its team; when that thread loses its CPU (or
    # is dispatched late) on a contended host the search ends milliseconds late and the device waits (seen: 3-10 ms
    # searches with next to no team shares taken over, `profiles/r03/box_spread.txt`).  The result is a pure function
    # of the slice, so when the launch thread needs a result that is later than twice the typical run time it has the
    # SAME search started again on a spare thread with the whole team and takes whichever of the two ends first.  At
    # most one dataset in eight (a uniformly slow host gains nothing from doing everything twice).
    hedging = os.environ.get("XMRIS_AMD_HEDGE", "1") != "0"
    # typical run times, kept with the plan from call to call: the pipeline-filling search of a call (whole team, the
    # device idle behind it) has a history of its own
    run_hist, fill_hist = plan.extra.setdefault("search_run_hist", []), plan.extra.setdefault("fill_run_hist", [])
    last_hedge = [-100]
    hedge_gap = max(1, int(os.environ.get("XM_HEDGE_SPACING", "8")))  # (tuning / test switch: datasets between two hedges)

    def collect(i, fut, args, ev):
        from concurrent.futures import TimeoutError as FutureTimeout

        recent = (fill_hist if i == 0 else run_hist)[-9:]
        if not hedging or i - last_hedge[0] < hedge_gap:
            return fut.result(), False
        if len(recent) < 3:
            # no history yet -- with several ranks a rank only searches the datasets it owns, so it may never have
            # one (found by the eight-rank executor test): the measured cost model of `_search_workers` stands in,
            # generously (x 1.5)
            gain = {1: 1.0, 2: 1.73, 4: 3.05, 8: 4.25, 16: 5.7}
            th = max(1, fill_team if i == 0 else team)
            recent = [1.5e-3 * (0.3 + 3.2 * (plan.n_out / 8192.0) / gain[max(k_ for k_ in gain if k_ <= th)])]
        typical = sorted(recent)[len(recent) // 2]
        wait = ev["t_exchanged"] + 2.0 * typical + 0.5e-3 - time.perf_counter()
        try:
            return fut.result(timeout=max(wait, 0.0)), False
        except FutureTimeout:
            from concurrent.futures import FIRST_COMPLETED, wait as wait_first

            last_hedge[0] = i
            hedge_pool = plan.extra.get("hedge_pool")
            if hedge_pool is None:
                hedge_pool = plan.extra["hedge_pool"] = ThreadPoolExecutor(max_workers=1, thread_name_prefix="xm-hedge")
            again = hedge_pool.submit(search, *args, fill_team, {})
            done, _ = wait_first([fut, again], return_when=FIRST_COMPLETED)
            return (fut.result() if fut in done else again.result()), True

    pending = {}  # dataset -> (partial result, future or None)

    def start_search(j):
        """Selection of dataset j -> (exchange) -> its search, inline or on a worker."""
        ev = events[j]
        ev["t_start"] = time.perf_counter()
        on_dev = use_dev and j >= cpu_fill
        if on_dev and exchange is None and not slice_on_host():  # queued behind its selection stage already (guess)
            ev["t_exchanged"] = ev["t_start"]
            pending[j] = (None, ("dev", dev_seq.pop(j)), None)
            return
        amax, flat, sl = sel[j % ring].wait()  # (largest L1 norm, guessed row * n + arg-max of its fp64 spectrum, spectrum)
        ev["t_selected"] = time.perf_counter()
        gflat, mine, owner = rank_offset_rows * n + flat, True, 0
        if exchange is not None:  # the guess is global too: the rank with the largest norm owns it
            mine, gflat, owner = exchange(amax, gflat)
        ev["t_exchanged"] = time.perf_counter()
        k = gflat % n
        res = AutophaseResult(0.0, 0.0, float(plan.freq[k]), int(gflat), int(k), amax)
        res.owner, res.mine = owner, mine
        fut = None
        if mine and on_dev:  # several ranks: the owner of the winning row queues the search kernel
            pending[j] = (res, ("dev", launch_dev_search(j)), (sl, int(k), res.pivot))
            return
        if mine:
            # the first search fills the pipeline (the first main pass waits for it): whole team; the others run
            # three (four) at a time and have as many device periods each
            th = fill_team if j == 0 else team  # (smaller teams for the searches right behind the first: slower, -1...3 %)
            if use_service:
                rec = hsearch["recs"][j % ring]
                if j == slow_j:  # test hook: this search reaches the service late
                    import threading

                    hsearch["seq"][0] += 1
                    seq = hsearch["seq"][0]
                    rec[7] = 0

                    def late(seq=seq, rec=rec, k=int(k), th=th):
                        from . import _lib

                        _lib.call("xm_hostsearch_submit", sel[j % ring].h_slice.data_ptr(), n, plan.extra["freq_c"].ctypes.data,
                                  aps.METHODS.index(method), k, int(iw), int(bool(p0_only)), 42, 0.01, 1000, int(th), seq, rec.data_ptr())

                    _abandon(rec)  # (the late search writes it whenever it ends, maybe after this call has returned)
                    tm = threading.Timer(slow_ms * 1e-3, late)
                    tm.daemon = True
                    tm.start()
                else:
                    seq = submit_host_search(j, int(k), th, rec)
                pending[j] = (res, ("host", seq), (sl, int(k), res.pivot))
                return
            fut = (pool.submit(search, sl, int(k), res.pivot, th, ev, j) if pool is not None
                   else search(sl, int(k), res.pivot, th, ev))
        pending[j] = (res, fut, (sl, int(k), res.pivot))

    def verify(i):
        """True global arg-max row of dataset i (its main pass has been queued) against the guess; repair."""
        b = i % ring
        res, ev = results[i], events[i]
        ev["verify_event"].synchronize()
        if use_keys:
            m2, fl = dev.read_key_result(bufs["vres"][b], c128)
            tmax, trow = m2 ** 0.5, fl // n
        else:
            tmax, trow = float(bufs["vmax"][b].item()) ** 0.5, int(bufs["vflat"][b].item()) // n
        g_row, owner = rank_offset_rows + trow, 0
        mine = True
        if exchange is not None:
            mine, gflat, owner = exchange(tmax, (rank_offset_rows + trow) * n)
            g_row = gflat // n
        if g_row == res.flat_index // n:
            res.speculation = "hit"
            if mine:
                res.max_abs = tmax  # the guess stage only knew the L1 norm
            return
        # wrong guess: the owner of the true row fetches its spectrum (fp64), searches again, everyone rotates
        vals = [0.0, 0.0, 0.0, 0.0]
        if mine:
            row = g_row - rank_offset_rows
            x1 = inputs[i][row:row + 1].to(torch.complex128)
            if plan.window64 is None:
                plan.window64 = torch.from_numpy(np.ascontiguousarray(plan.window_host)).to(x1.device, torch.float64)
            if slice_on_host():
                sl = winner_spectrum(plan, x1[0].cpu().numpy())
            else:
                sl = dev.pipeline_fused(x1, n, plan.pad_left, window=plan.window64).out[0].cpu().numpy()
            k = int(np.argmax(np.abs(sl)))
            p0, p1, opt = aps.solve(sl, plan.freq, float(plan.freq[k]), k, iw, method=method, p0_only=p0_only, polish=polish)
            vals = [p0, p1, float(k), float(opt.nfev)]
        if broadcast is not None:
            vals = broadcast(vals, owner)
        p0, p1, k = float(vals[0]), float(vals[1]), int(vals[2])
        pivot = float(plan.freq[k])
        # the dataset's FIDs are still there: run its main pass again with the right parameters (one read + one
        # write of the dataset, and the result is the classic schedule's to the bit; rotating the wrong output in
        # place by the phase ratio reads AND writes the spectra and adds two roundings)
        main_pass(plan, inputs[i], outputs[i], p0, p1, pivot)
        res.p0, res.p1, res.pivot, res.target_idx = p0, p1, pivot, k
        res.flat_index, res.max_abs, res.owner, res.mine = g_row * n + k, tmax, owner, mine
        res.nfev = int(vals[3]) if mine else 0
        res.speculation = "repaired"

    guessed = started = -1
    events[0]["t_call"] = t_call
    unverified = []  # datasets whose main pass is queued and whose guess is not settled yet (ascending)
    # (the order of the exchange calls must be the same on every rank: with several ranks the fill keeps its fixed order)
    fast_fill = (exchange is None and not use_dev and overlap and n_sets > 2 and s_ahead >= 2
                 and os.environ.get("XM_FAST_FILL", "1") != "0")

    # (XM_FILL_RAMP=0: the whole look-ahead in front of the first main pass, rounds 1-3.  Measured through the multi-rank
    # code path on a GPU of its own -- `bench.py` with XM_BENCH_SOLO_EXCHANGE=1, driver command, four rounds,
    # profiles/r04/fill.txt: 1.302 -> 1.244 ms per step, i.e. what one rank reaches with the event-driven fill
    # (1.248); four and six ranks SHARING the box's one GPU: 1.61 vs 1.59 and 1.38 vs 1.40, nothing either way.)
    fill_ramp = overlap and os.environ.get("XM_FILL_RAMP", "1") != "0"

    # searches started beside the pipeline-filling one (it has the whole team for its millisecond; the others have
    # device periods of slack and start right behind it) -- tuning switch, A/B in profiles/r04/fill.txt
    fill_searches = int(os.environ.get("XM_FILL_SEARCHES", "1"))

    def first_search_done():
        _, fut0, _ = pending[0]
        if isinstance(fut0, tuple) and len(fut0) == 2 and fut0[0] == "host":
            return dev.search_done(hsearch["recs"][0], fut0[1])
        return fut0 is None or not hasattr(fut0, "done") or fut0.done()
    for i in range(n_sets):
        b = i % ring
        ev = events[i]
        # keep the guess kernels g_ahead and the searches s_ahead datasets in front; while the pipeline fills, every
        # search starts right behind its own guess (the first main pass waits for the first search)
        # (device engine: a search is started as soon as its selection stage is queued -- one rank -- or has ended --
        # several ranks, whose exchange needs the stage's result: one dataset behind the newest guess)
        s_look = s_ahead if not use_dev else (g_ahead if exchange is None else max(s_ahead, g_ahead - 1))
        if i == 0 and fast_fill:
            # Filling the pipeline, one rank: every guess of the look-ahead is queued at once (the device works through
            # them while the first search runs), searches start as their selection stages END -- this thread polls the
            # events instead of blocking on each in turn -- and the moment the FIRST search is there its main pass is
            # queued.  Starting all the look-ahead's searches first, each behind a blocking wait for its selection
            # (8 x (0.13 ms of kernels + the host's turnaround)), had the first main pass queued at 1.5-1.7 ms with
            # the first search done at 1.0 and the device idle in between (profiles/r04/fill.txt).
            while guessed < min(n_sets - 1, 1):  # (two guesses, then the first search: nothing else delays its start)
                guessed += 1
                guess(guessed)
            started = 0
            start_search(0)
            while not first_search_done():
                more_guesses = guessed < min(n_sets - 1, g_ahead)
                if more_guesses:  # one per turn: a launch takes this thread half the time the device needs for it
                    guessed += 1
                    guess(guessed)
                nxt = started + 1
                if nxt > min(n_sets - 1, s_look, fill_searches):
                    if more_guesses:
                        continue
                    break  # (`collect` waits for the first search -- and hedges it if it is late)
                if nxt <= guessed and sel[nxt % ring].event.query():
                    started = nxt
                    start_search(nxt)
                elif blocking and not more_guesses:
                    time.sleep(2e-5)
        # (several ranks -- where the order of the exchange calls may not depend on anything a rank observes -- and the
        # other engines: the look-ahead is built up over the first datasets, three searches before the first main pass
        # and three more with every dataset, instead of all of it in front of the first main pass)
        ramp = 2 + 3 * i if (not use_dev and not fast_fill and fill_ramp) else n_sets
        while started < min(n_sets - 1, i + s_look, ramp):
            while guessed < min(n_sets - 1, started + 1):
                guessed += 1
                guess(guessed)
            started += 1
            start_search(started)
        while guessed < min(n_sets - 1, i + g_ahead):
            guessed += 1
            guess(guessed)
        if use_service or use_dev:
            advance_polishes()
        res, fut, search_args = pending.pop(i)
        ev["t_collect"] = time.perf_counter()
        if isinstance(fut, tuple) and len(fut) == 2 and fut[0] == "dev":  # a search kernel
            p0, p1, k, nfev, fun, timing, hedged = collect_dev(i, fut[1], ev)
            if res is None:  # one rank: the record is the first the host hears of this dataset's winner
                slot = bufs["sel_slots"][b]
                gflat = (rank_offset_rows + int(slot[1].item()) // n) * n + k
                res = AutophaseResult(0.0, 0.0, float(plan.freq[k]), int(gflat), int(k), float(slot[0].item()) ** 0.5)
            res.p0, res.p1, res.nfev, res.fun, res.timing, res.hedged = p0, p1, nfev, fun, timing, hedged
        elif isinstance(fut, tuple) and len(fut) == 2 and fut[0] == "host":  # a search on the library's native threads
            p0, p1, nfev, fun, timing, hedged = collect_host(i, fut[1], search_args[1], ev)
            res.p0, res.p1, res.nfev, res.fun, res.timing, res.hedged = p0, p1, nfev, fun, timing, hedged
        elif fut is not None:
            if pool is not None:
                (p0, p1, opt), res.hedged = collect(i, fut, search_args, ev)
                if "t_search_end" in ev and "t_search_begin" in ev:
                    hist = fill_hist if i == 0 else run_hist
                    hist.append(ev["t_search_end"] - ev["t_search_begin"])
                    del hist[:-16]
            else:
                p0, p1, opt = fut
            res.p0, res.p1, res.nfev, res.fun = p0, p1, int(opt.nfev), float(opt.fun)
            res.timing = {"generations_ms": 1e3 * opt.get("t_generations", 0.0), "polish_ms": 1e3 * opt.get("t_polish", 0.0)}
        if broadcast is not None:
            res.p0, res.p1 = broadcast([res.p0, res.p1], res.owner)
        ev["t_solved"] = ev["t_table"] = time.perf_counter()
        results[i] = res
        # Settle earlier guesses: always those two or more datasets back (their main pass finished a device period
        # ago, so this never waits -- neither for this GPU nor, through the exchange, for another rank's), and at once
        # any whose output buffer is about to be overwritten (a repair must still find its spectra there).
        while unverified and (unverified[0] <= i - 2 or any(outputs[j].data_ptr() == outputs[i].data_ptr() for j in unverified)):
            verify(unverified.pop(0))
        if trace is not None:
            ev["main0"], ev["main1"] = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev["main0"].record()
        if use_keys:
            main_pass(plan, inputs[i], outputs[i], res.p0, res.p1, res.pivot, global_key=bufs["vkey"][b],
                      key_result=bufs["vres"][b])  # the kernel's last workgroup decodes + clears the key
        else:
            main_pass(plan, inputs[i], outputs[i], res.p0, res.p1, res.pivot, want_argmax=True, absmax2=bufs["tmax"][b],
                      argidx=bufs["tidx"][b], argmax_value_only=True)
        if trace is not None:
            ev["main1"].record()
            ev["main_kernel"] = dev.last_kernel()
        if not use_keys:
            dev.argmax_reduce_async(bufs["tmax"][b], bufs["tidx"][b], n, gmax=bufs["vmax"][b], gflat=bufs["vflat"][b])
        ev["verify_event"] = torch.cuda.Event(blocking=blocking)
        ev["verify_event"].record()
        unverified.append(i)
        if trace is not None:
            trace.append(ev)
    events[-1]["t_last_queued"] = time.perf_counter()
    while unverified:
        verify(unverified.pop(0))
    events[-1]["t_return"] = time.perf_counter()
    return results


def phase_ramp_of(plan: PipelinePlan, p0: float, p1: float, pivot: float):
    """phasing.py:56-69 on the plan's (uniform) frequency axis in closed form: phi[k] = phase0 + dphase * k.
    Returns (phase0, dphase) in radians, or None when the axis is not uniform (a table is needed then)."""
    lin = plan.extra.get("freq_linear")
    if lin is None:
        f = np.asarray(plan.freq, dtype=np.float64)
        ok = f.size >= 2
        if ok:
            step = (f[-1] - f[0]) / (f.size - 1)
            ok = step != 0 and bool(np.all(np.abs(f - (f[0] + step * np.arange(f.size))) <= 1e-12 * np.abs(f).max()))
        lin = plan.extra["freq_linear"] = (float(f[0]), float(step), float(f.max() - f.min())) if ok else False
    if lin is False:
        return None
    c0, step, rng = lin
    if rng == 0:
        return float(np.deg2rad(p0)), 0.0
    return (float(np.deg2rad(p0) + np.deg2rad(p1) * (c0 - pivot) / rng), float(np.deg2rad(p1) * step / rng))


def main_pass(plan: PipelinePlan, x2, out, p0: float, p1: float, pivot: float, **kw):
    """The fused main pass with the autophase ramp: in closed form where the kernel applies it natively (no table
    is built, nothing is uploaded), through a phase table otherwise."""
    native = plan.extra.get(("ramp_native", x2.data_ptr() & 15, x2.shape[1], str(x2.dtype)))
    if native is None:
        native = plan.extra[("ramp_native", x2.data_ptr() & 15, x2.shape[1], str(x2.dtype))] = dev.ramp_native(
            x2, plan.n_out, plan.pad_left)
    ramp = phase_ramp_of(plan, p0, p1, pivot) if native else None
    if ramp is not None:
        return dev.pipeline_fused(x2, plan.n_out, plan.pad_left, window=plan.window, phase_ramp=ramp, out=out, **kw)
    ph = upload_phase_table(plan, x2, p0, p1, pivot)
    return dev.pipeline_fused(x2, plan.n_out, plan.pad_left, window=plan.window, phase_table=ph, out=out, **kw)


def upload_phase_table(plan: PipelinePlan, like, p0: float, p1: float, pivot: float):
    """e^{i phi} over the frequency axis (fp64 on the host, phasing.py:56-73), rounded once to the storage
    precision into a reused pinned staging buffer and copied to the device asynchronously."""
    import torch

    n = plan.n_out
    key = ("phase_stage", str(like.dtype))
    stage = plan.extra.get(key)
    if stage is None:
        stage = plan.extra[key] = [torch.empty(n, dtype=like.dtype, pin_memory=True) for _ in range(2)] + [0]
    stage[2] ^= 1
    host = stage[stage[2]]
    # fp64 cos / sin of the ramp, rounded once, written straight into the pinned staging buffer by the host library
    from . import _lib

    freq = plan.extra.get("freq_c")
    if freq is None:
        freq = plan.extra["freq_c"] = np.ascontiguousarray(plan.freq, dtype=np.float64)
    rc = _lib.load().xm_phase_table(freq.ctypes.data, n, float(p0), float(p1), float(pivot), host.data_ptr(),
                                    1 if like.dtype == torch.complex64 else 0)
    if rc:
        raise ValueError("xm_phase_table rejected its arguments")
    dev_t = plan.extra.setdefault(("phase_dev", str(like.dtype)),
                                  [torch.empty(n, dtype=like.dtype, device=like.device) for _ in range(2)])
    out = dev_t[stage[2]]
    out.copy_(host, non_blocking=True)
    return out


def _selection_only(pre, plan, target_coord):
    amax, flat = dev.argmax_reduce(pre.absmax2, pre.argidx, plan.n_out)
    k = flat % plan.n_out
    if target_coord is not None:
        return AutophaseResult(0.0, 0.0, float(target_coord), flat,
                               int(np.argmin(np.abs(plan.freq - target_coord))), amax)
    return AutophaseResult(0.0, 0.0, float(plan.freq[k]), flat, int(k), amax)
