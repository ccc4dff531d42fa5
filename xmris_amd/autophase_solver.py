"""Host side of ``autophase``: the O(1)-per-dataset search for (p0, p1) on the ONE 1-D slice
that holds the global |X| maximum (reference ``processing/phasing.py:100-157, 257-287``).

The reference drives ``scipy.optimize.differential_evolution`` (best1bin, tol=0.01, seed=42)
with objectives that go through ``phase()`` -> xarray on every evaluation.  Here the same
optimiser is driven with the same arithmetic on plain ndarrays (no xarray object per
evaluation).  The data-parallel parts of autophase -- the global arg-max and the broadcast
phase multiply over the whole dataset -- run on the GPU (``xm_pipeline_fused``).
"""
from __future__ import annotations

import numpy as np

METHODS = ("acme", "peak_minima", "positivity")
MSG_METHOD = "Method must be 'acme', 'peak_minima', or 'positivity'"  # phasing.py:268 (kept in dims.py too)


def phase_angles(coords: np.ndarray, p0: float, p1: float, pivot: float):
    """phasing.py:56-69: phi = rad(p0) + rad(p1) * (c - pivot) / (max c - min c); scalar if range 0."""
    x_min = float(coords.min())
    x_max = float(coords.max())
    x_range = x_max - x_min
    p0_rad = np.radians(p0)
    p1_rad = np.radians(p1)
    if x_range == 0:
        return p0_rad
    return p0_rad + p1_rad * ((coords - pivot) / x_range)


def phase_table(coords: np.ndarray, p0: float, p1: float, pivot: float) -> np.ndarray:
    """e^{i phi} over the axis in fp64 (phasing.py:73); always an array of len(coords)."""
    ph = phase_angles(np.asarray(coords, dtype=np.float64), p0, p1, pivot)
    f = np.exp(1.0j * ph)
    if np.ndim(f) == 0:
        f = np.full(len(coords), f, dtype=np.complex128)
    return f


def _phased_real(ph, sl, coords, pivot):
    p0 = ph[0]
    p1 = ph[1] if len(ph) > 1 else 0.0
    return np.real(sl * np.exp(1.0j * phase_angles(coords, p0, p1, pivot)))


def acme_score(ph, sl, coords, pivot):
    """phasing.py:100-122 -- entropy of the first derivative + negativity penalty."""
    data = _phased_real(ph, sl, coords, pivot)
    ds1 = np.abs((data[1:] - data[:-1]) / 2)
    p1_prob = ds1 / np.sum(ds1)
    p1_prob[p1_prob == 0] = 1
    h1s = np.sum(-p1_prob * np.log(p1_prob))
    as_ = data - np.abs(data)
    pfun = 0.0
    if np.sum(as_) < 0:
        pfun = np.sum((as_ / 2) ** 2)
    return (h1s + 1000 * pfun) / data.shape[-1] / np.max(data)


def peak_minima_score(ph, sl, coords, pivot, target_idx, index_width):
    """phasing.py:125-139."""
    data = _phased_real(ph, sl, coords, pivot)
    start = max(0, target_idx - index_width)
    end = min(len(data), target_idx + index_width)
    mina = np.min(data[start:target_idx]) if start < target_idx else data[target_idx]
    minb = np.min(data[target_idx:end]) if end > target_idx else data[target_idx]
    return np.abs(mina - minb)


def roi_positivity_score(ph, sl, coords, pivot, target_idx, index_width):
    """phasing.py:142-157."""
    data = _phased_real(ph, sl, coords, pivot)
    start = max(0, target_idx - index_width)
    end = min(len(data), target_idx + index_width)
    roi = data[start:end]
    return np.sum(np.abs(roi[roi < 0])) * 5.0 - np.sum(roi[roi > 0])


def index_width_of(coords: np.ndarray, peak_width: float) -> int:
    """phasing.py:245-247."""
    step = np.abs(coords[1] - coords[0])
    return max(1, int(round((peak_width / 2.0) / step)))


_CPU_SHARE = None


def _cpu_share() -> int:
    """CPUs this process may use: scheduler affinity, capped by the cgroup v2 quota.  Read once per process (the
    executor asks on every dataset: a sched_getaffinity call and a file read on the launch thread; advisor, round 3)."""
    global _CPU_SHARE
    if _CPU_SHARE is None:
        _CPU_SHARE = _read_cpu_share()
    return _CPU_SHARE


def _read_cpu_share() -> int:
    import os

    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            cpus = min(cpus, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cpus


def scarce_cpus() -> bool:
    """Several ranks on few cores (fewer than four per rank): host waits on device events should BLOCK (interrupt)
    instead of spinning -- a spinning wait of one rank takes the core another rank's search team is running on."""
    import os

    if os.environ.get("XM_BLOCKING_SYNC"):  # tuning switch
        return os.environ["XM_BLOCKING_SYNC"] != "0"
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    return local_world > 1 and _cpu_share() < 4 * local_world


def default_threads() -> int:
    """Team size for the native objective: at most 16 and at most HALF of this process's share of the CPUs it may use
    (scheduler affinity and the cgroup v2 quota), as a power of two; several ranks on one node: see below."""
    import os

    if os.environ.get("XM_SOLVER_THREADS"):  # tuning switch
        return max(1, min(32, int(os.environ["XM_SOLVER_THREADS"])))
    cpus = _cpu_share()
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    if local_world > 1:
        # Several ranks on one node share its cores.  Searches are per DATASET, not per rank: only the rank that owns
        # a dataset's winning spectrum searches, the streaming executor keeps its look-ahead's worth of searches (two to four)
        # in flight on the whole node (`pipeline._run_stream_speculative`), and the other ranks wait for the
        # broadcast sleep-polling.  So the node-wide budget is what the ranks' launch threads leave: one core per rank
        # is reserved for launching and polling (a launch thread is busy for ~0.2 ms of a 1.2 ms step), the searches
        # in flight share the rest -- 16 CPUs and 8 ranks: 8 cores, 4 per search (1.1 ms of generations; round 2's
        # (cpus - 2 ranks) / 2 left ONE thread there: 3.2 ms against a 1.2 ms device period).  Any team size works (a
        # batch of evaluations is cut into 16+ work units).
        return max(1, min(16, cpus - local_world))
    # Half of the share, as a power of two (the work units of a batch are 4 evaluations x 4 parts): a team that
    # fills the whole CPU quota while it spins leaves no headroom for the HIP runtime's threads, and a cgroup that
    # overdraws its quota is frozen until the next 100 ms period.
    share = max(1, cpus // 2)
    team = 1
    while team * 2 <= min(16, share):
        team *= 2
    return team


def burst_threads() -> int:
    """Team size for ONE search with nothing beside it (a single accessor call, or the search that fills a streaming
    call's pipeline: the device waits for it): the whole share of the CPUs this process may use, up to 16 --
    `default_threads` keeps half of it free because a streaming executor's teams spin for as long as it runs; a
    millisecond does not reach the quota."""
    import os

    if os.environ.get("XM_SOLVER_THREADS") or int(os.environ.get("LOCAL_WORLD_SIZE", "1")) > 1:
        return default_threads()
    cpus = _cpu_share()
    team = 1
    while team * 2 <= min(16, cpus):
        team *= 2
    return max(team, default_threads())


def stream_threads(host_paced: bool = False) -> int:
    """Thread budget of ALL the searches a streaming executor keeps in flight (`pipeline._search_workers` divides it).
    One rank on its node, two or three searches in flight (the device paces the steps; each search is busy for
    about half of the device periods it has): THREE QUARTERS of the share, up to 12 -- with two in flight, six threads
    apiece ran the generations in 1.25 instead of 1.45 ms at the same throughput (round 3, six interleaved pairs at the
    driver's K = 20: 51.5 vs 51.6 M spectra/s, 6.5 vs 5.3 cores busy); the executor now keeps THREE in flight with four
    threads each (`pipeline._search_workers`: the same throughput again, a third device period of slack for a search
    that runs late).  The WHOLE share (two teams of eight, 1.05 ms) is 2 % faster when nothing goes wrong and stalled
    for 2-5 ms in three of seven runs: sixteen spinning threads plus the launch thread oversubscribe a 16-CPU quota.
    `host_paced` (more than three searches in flight: every team spins all the time): the whole share (see below; round
    3, with the searches on Python threads: half).  Several ranks on one node, or XM_SOLVER_THREADS: `default_threads`."""
    import os

    if os.environ.get("XM_SOLVER_THREADS") or int(os.environ.get("LOCAL_WORLD_SIZE", "1")) > 1:
        return default_threads()
    cpus = min(16, _cpu_share())
    if host_paced:
        # Round 4: the searches run on native threads of the library now (`xm_hostsearch_submit`), no interpreter lock
        # is fought over, and where the searches pace the steps the WHOLE share is theirs -- 16,384 x 2048 -> 4096 with
        # four searches in flight: 0.38 ms per dataset with teams of two (round 3's half share), 0.31 with three,
        # 0.26 with four (profiles/r04/search_workers.txt).
        return max(default_threads(), cpus)
    return max(default_threads(), cpus - cpus // 4)


class NativeObjective:
    """The three objectives evaluated by libxmris_hip.so's host solver (vectorised C++, fp64)."""

    def __init__(self, sl, coords, pivot, target_idx, index_width, method):
        from . import _lib

        self._lib = _lib.load()
        self._sl = np.ascontiguousarray(sl, dtype=np.complex128)
        self._c = np.ascontiguousarray(coords, dtype=np.float64)
        self._h = self._lib.xm_solver_create(self._sl.ctypes.data, self._c.ctypes.data, len(self._sl), float(pivot),
                                             METHODS.index(method), int(target_idx), int(index_width))
        if not self._h:
            raise ValueError("xm_solver_create rejected the slice (needs n >= 2 and a valid target index)")
        self.set_threads(default_threads())

    def __call__(self, x):
        xx = np.ascontiguousarray(x, dtype=np.float64)
        return self._lib.xm_solver_score(self._h, xx.ctypes.data, len(xx))

    def set_threads(self, n):
        return self._lib.xm_solver_set_threads(self._h, int(n))

    def evaluations(self):
        """Objective evaluations performed so far (>= the sequential algorithm's nfev: xm_solver_de
        evaluates trials speculatively in batches)."""
        return int(self._lib.xm_solver_nfev(self._h))

    def score_batch(self, xs):
        xs = np.asarray(xs, dtype=np.float64)
        buf = np.zeros((xs.shape[0], 2))
        buf[:, :xs.shape[1]] = xs
        out = np.empty(xs.shape[0])
        self._lib.xm_solver_score_batch(self._h, buf.ctypes.data, int(xs.shape[1]), int(xs.shape[0]), out.ctypes.data)
        return out

    def fg(self, x, lb, ub):
        """(f, forward-difference gradient) at x inside the box [lb, ub] in ONE native call (`xm_solver_fg`): what
        `polish_lbfgsb` otherwise spells out in ~20 numpy operations per request, under the interpreter lock that the
        other searches in flight and the launch thread are waiting for."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        g = np.empty(len(x))
        f = np.empty(1)
        rc = self._lib.xm_solver_fg(self._h, x.ctypes.data, len(x), lb.ctypes.data, ub.ctypes.data, f.ctypes.data,
                                    g.ctypes.data)
        if rc:
            raise ValueError("xm_solver_fg rejected its arguments")
        return float(f[0]), g

    def de(self, p0_only, seed=42, tol=0.01, maxiter=1000):
        import ctypes

        x = (ctypes.c_double * 2)()
        fun, nfev, nit = ctypes.c_double(), ctypes.c_int(), ctypes.c_int()
        rc = self._lib.xm_solver_de(self._h, int(bool(p0_only)), seed, tol, maxiter, x, ctypes.byref(fun),
                                    ctypes.byref(nfev), ctypes.byref(nit))
        return rc, np.array(x[:1 if p0_only else 2]), fun.value, nfev.value, nit.value

    def __del__(self):
        try:
            self._lib.xm_solver_destroy(self._h)
        except Exception:
            pass


def polish_lbfgsb(obj, x0, bounds, force_scipy: bool = False):
    """`scipy.optimize.minimize(obj, x0, method="L-BFGS-B", bounds=bounds)` -- the polish step of
    `differential_evolution` (phasing.py:276-284 runs it with scipy's defaults) -- without the per-call Python
    machinery around it (`ScalarFunction`, bounds standardisation, `approx_derivative`'s generic front end: 0.3-0.5 ms
    of interpreter time per search, held under the GIL while other searches and the launch thread wait).  Same
    compiled core (`scipy.optimize._lbfgsb.setulb`), same defaults (maxcor 10, ftol 2.22e-9, gtol 1e-5, eps 1e-8,
    maxls 20), same forward-difference gradient as `approx_derivative(method="2-point", abs_step=1e-8, bounds=...)`
    with its exactly-representable step and its flip at the upper bound, same evaluation count (the gradient's
    evaluations included, the repeated request for f at x0 answered from the cache like `ScalarFunction` does):
    x, fun, nfev, nit, success are scipy's to the bit (tests/test_abi_and_host.py).  The three points of one
    f-and-gradient request go to the native objective in ONE call.  Any surprise in scipy's private interface
    (this follows 1.15.3, the version SURVEY pins) -> the public `minimize`."""
    import os

    import scipy
    import scipy.optimize

    try:
        # the private entry point is followed as scipy 1.15 has it (SURVEY 8c pins 1.15.3); any other release takes
        # the public route
        if force_scipy or not scipy.__version__.startswith("1.15."):
            raise ImportError
        from scipy.optimize import _lbfgsb
        from scipy.optimize._lbfgsb_py import _minimize_lbfgsb  # noqa: F401 -- same module layout as the code followed
    except ImportError:
        return scipy.optimize.minimize(obj, np.copy(x0), method="L-BFGS-B", bounds=bounds)
    lb = np.array([b[0] for b in bounds], dtype=np.float64)
    ub = np.array([b[1] for b in bounds], dtype=np.float64)
    n = len(lb)
    m, maxls, maxfun, maxiter = 10, 20, 15000, 15000
    factr = 2.2204460492503131e-09 / np.finfo(float).eps
    pgtol, abs_step = 1e-5, 1e-8
    x0 = np.clip(np.asarray(x0, dtype=np.float64).ravel(), lb, ub)
    state = {"nfev": 0, "x": None, "f": None, "g": None}
    native_fg = hasattr(obj, "fg") and n <= 2 and not os.environ.get("XM_POLISH_NUMPY_FG")  # (switch: cross-checks)

    def func_and_grad(x):
        if state["x"] is not None and np.array_equal(x, state["x"]):
            return state["f"], state["g"]
        xc = np.array(x, dtype=np.float64)
        if native_fg:  # the same arithmetic in one native call (bit-equal: tests/test_abi_and_host.py)
            f, g = obj.fg(xc, lb, ub)
            state["nfev"] += n + 1
            state.update(x=xc, f=f, g=g)
            return f, g
        # approx_derivative: absolute step, relative fallback when it vanishes, flipped where it leaves the bounds
        sign = (xc >= 0).astype(float) * 2 - 1
        h = np.full(n, abs_step)
        h = np.where(((xc + h) - xc) == 0, np.finfo(np.float64).eps ** 0.5 * sign * np.maximum(1.0, np.abs(xc)), h)
        lower, upper = xc - lb, ub - xc
        xt = xc + h
        violated = (xt < lb) | (xt > ub)
        fitting = np.abs(h) <= np.maximum(lower, upper)
        h[violated & fitting] *= -1
        forward = (upper >= lower) & ~fitting
        h[forward] = upper[forward]
        backward = (upper < lower) & ~fitting
        h[backward] = -lower[backward]
        pts = np.tile(xc, (n + 1, 1))
        dx = np.empty(n)
        for i in range(n):
            pts[1 + i, i] += h[i]
            dx[i] = pts[1 + i, i] - xc[i]  # the step as an exactly representable number
        vals = obj.score_batch(pts)
        state["nfev"] += n + 1
        f = float(vals[0])
        g = (vals[1:] - f) / dx
        state.update(x=xc, f=f, g=g)
        return f, g

    func_and_grad(x0)  # ScalarFunction evaluates f and the gradient at x0 when it is built
    nbd = np.full(n, 2, dtype=np.int32)
    x = np.array(x0, dtype=np.float64)
    f = np.array(0.0, dtype=np.int32)
    g = np.zeros((n,), dtype=np.int32)
    wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
    iwa = np.zeros(3 * n, dtype=np.int32)
    task = np.zeros(2, dtype=np.int32)
    ln_task = np.zeros(2, dtype=np.int32)
    lsave = np.zeros(4, dtype=np.int32)
    isave = np.zeros(44, dtype=np.int32)
    dsave = np.zeros(29, dtype=np.float64)
    nit = 0
    try:
        while True:
            g = g.astype(np.float64)
            _lbfgsb.setulb(m, x, lb, ub, nbd, f, g, factr, pgtol, wa, iwa, task, lsave, isave, dsave, maxls, ln_task)
            if task[0] == 3:
                f, g = func_and_grad(x)
            elif task[0] == 1:
                nit += 1
                if nit >= maxiter:
                    task[0], task[1] = 5, 504
                elif state["nfev"] > maxfun:
                    task[0], task[1] = 5, 502
            else:
                break
    except (TypeError, ValueError):  # another scipy: its private entry point takes other arguments
        return scipy.optimize.minimize(obj, np.copy(x0), method="L-BFGS-B", bounds=bounds)
    return scipy.optimize.OptimizeResult(x=x, fun=f, jac=g, nfev=state["nfev"], nit=nit, success=bool(task[0] == 4),
                                         status=0 if task[0] == 4 else (1 if (state["nfev"] > maxfun or nit >= maxiter) else 2))


def _numpy_objective(sl, coords, pivot, target_idx, index_width, method):
    """The objective as the reference evaluates it (phasing.py:100-157 on plain ndarrays): numpy ufuncs, numpy's
    summation order, the platform's libm / SVML -- what `engine="scipy"` drives."""
    if method == "acme":
        return lambda x: acme_score(x, sl, coords, pivot)
    if method == "peak_minima":
        return lambda x: peak_minima_score(x, sl, coords, pivot, target_idx, index_width)
    return lambda x: roi_positivity_score(x, sl, coords, pivot, target_idx, index_width)


class _PolishObjective:
    """The numpy objective for the polish on the reference's route, with what does not depend on (p0, p1) computed
    ONCE: `phase_angles` evaluates (coords - pivot) / (max - min) -- the same two numpy statements on the same
    operands, hence the same bits -- on every call (a min, a max, a subtraction and a division over the axis: 25 us
    of a 390 us evaluation at 8192 bins).  Everything that touches (p0, p1) is the statements of `acme_score` etc.
    unchanged.  `score_batch` is the interface `polish_lbfgsb`'s forward differences use."""

    def __init__(self, sl, coords, pivot, target_idx, index_width, method):
        self.sl, self.method, self.ti, self.iw = sl, method, target_idx, index_width
        x_range = float(coords.max()) - float(coords.min())
        self.x_range = x_range
        self.u = None if x_range == 0 else (coords - pivot) / x_range
        self.nfev = 0

    def _data(self, ph):
        p0_rad = np.radians(ph[0])
        p1_rad = np.radians(ph[1] if len(ph) > 1 else 0.0)
        ang = p0_rad if self.u is None else p0_rad + p1_rad * self.u
        return np.real(self.sl * np.exp(1.0j * ang))

    def __call__(self, ph):
        self.nfev += 1
        data = self._data(ph)
        if self.method == "acme":  # phasing.py:100-122, the statements of `acme_score`
            ds1 = np.abs((data[1:] - data[:-1]) / 2)
            p1_prob = ds1 / np.sum(ds1)
            p1_prob[p1_prob == 0] = 1
            h1s = np.sum(-p1_prob * np.log(p1_prob))
            as_ = data - np.abs(data)
            pfun = 0.0
            if np.sum(as_) < 0:
                pfun = np.sum((as_ / 2) ** 2)
            return (h1s + 1000 * pfun) / data.shape[-1] / np.max(data)
        start = max(0, self.ti - self.iw)
        end = min(len(data), self.ti + self.iw)
        if self.method == "peak_minima":  # phasing.py:125-139
            mina = np.min(data[start:self.ti]) if start < self.ti else data[self.ti]
            minb = np.min(data[self.ti:end]) if end > self.ti else data[self.ti]
            return np.abs(mina - minb)
        roi = data[start:end]  # phasing.py:142-157
        return np.sum(np.abs(roi[roi < 0])) * 5.0 - np.sum(roi[roi > 0])

    def score_batch(self, pts):
        return np.array([self(p) for p in np.asarray(pts, dtype=np.float64)])


def polish_reference(sl, coords, pivot, target_idx, index_width, method, p0_only, x):
    """The polish of `differential_evolution` on the reference's own route (phasing.py:276-284 with scipy's defaults):
    scipy's L-BFGS-B minimiser on the NUMPY objective from the generations' best member `x`; accepted when it lowers
    that objective.  Returns (x, fun, nfev, polished).  Used by every engine whose generations ended with a member
    that does not pass scipy's projected-gradient test (`polish="exact"`; the device search's `needs_polish`)."""
    import scipy.optimize

    bounds = [(-180.0, 180.0)] if p0_only else [(-180.0, 180.0), (-4000.0, 4000.0)]
    x = np.asarray(x, dtype=np.float64)[:len(bounds)]
    fn = _PolishObjective(np.asarray(sl, dtype=np.complex128), np.asarray(coords, dtype=np.float64), pivot, target_idx,
                          index_width, method)
    fun = float(fn(x))
    # scipy's minimiser itself, driven without its per-call Python front end (`polish_lbfgsb`: the same compiled
    # L-BFGS-B core, the same forward differences, bit for bit -- tests/test_abi_and_host.py); 4.9 -> 2.6 ms per polish
    res = polish_lbfgsb(fn, np.copy(x), bounds)
    lo = np.array([b_[0] for b_ in bounds])
    hi = np.array([b_[1] for b_ in bounds])
    polished = bool(res.fun < fun and res.success and np.all(res.x <= hi) and np.all(lo <= res.x))
    if polished:
        return np.asarray(res.x, dtype=np.float64), float(res.fun), int(res.nfev), True
    return x, fun, int(res.nfev), False


class PolishWorkers:
    """A few worker PROCESSES (`xmris_amd/_polish_worker.py`, started as plain children: `python -c ...`, no fork of
    this process, no re-import of its main module) that run `polish_reference` away from this process's interpreter
    lock.  `submit(...)` takes `polish_reference`'s arguments and returns a future.  Nobody ever waits for a worker to
    come up: a worker joins the free list when it has reported ready (its imports take ~1 s of CPU), and a request that
    finds no free worker -- none started yet, all busy, one died -- is polished by the future's own thread.
    `start()` launches the children (idempotent); with `lazy` the first request does (several ranks on one node: 6 x 4
    interpreters importing scipy at the start of a stream ate a 16-CPU quota and the cgroup was throttled for
    40-60 ms inside the timed region, profiles/r04/rehearsal_6ranks.txt)."""

    def __init__(self, n: int = 4, lazy: bool = False):
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor

        self._n = max(1, int(n))
        self._alive = 0  # workers that have reported ready and have not been found dead
        self._free = queue.Queue()
        self._procs = []
        self._started = False
        self._lock = threading.Lock()
        # (threads that mostly wait on a pipe: they hold the interpreter lock for microseconds per request)
        self._pool = ThreadPoolExecutor(max_workers=self._n, thread_name_prefix="xm-polish")
        import atexit

        atexit.register(self.close)
        if not lazy:
            self.start()

    def start(self):
        import os
        import subprocess
        import sys
        import threading

        with self._lock:
            if self._started:
                return
            self._started = True
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1",
                       OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
            for _ in range(self._n):
                try:
                    pr = subprocess.Popen([sys.executable, "-c", "from xmris_amd import _polish_worker as w; w.main()"],
                                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env, cwd=root)
                except OSError:
                    break
                self._procs.append(pr)
                threading.Thread(target=self._await_ready, args=(pr,), daemon=True, name="xm-polish-ready").start()

    def _await_ready(self, pr):
        try:
            if pr.stdout.read(8) == b"XMREADY\n":  # (written by the worker once numpy, scipy and the objective are imported)
                with self._lock:
                    self._alive += 1
                self._free.put(pr)
        except Exception:  # noqa: BLE001 -- a worker that never reports is never used
            pass

    def submit(self, *args):
        if not self._started:
            self.start()
        return self._pool.submit(self._call, args)

    def _call(self, args):
        import pickle
        import queue
        import struct

        # no worker is up (yet, or any more): this thread does it; otherwise wait for one to come free -- a polish in
        # this process costs the launch thread its share of the interpreter lock, which is what the workers are for
        pr = None
        while pr is None:
            if self._alive <= 0:
                return polish_reference(*args)
            try:
                pr = self._free.get(timeout=0.02)
            except queue.Empty:
                pass
        try:
            if pr.poll() is None:
                blob = pickle.dumps(args, protocol=pickle.HIGHEST_PROTOCOL)
                pr.stdin.write(struct.pack("<q", len(blob)))
                pr.stdin.write(blob)
                pr.stdin.flush()
                head = pr.stdout.read(8)
                if len(head) == 8:
                    status, value = pickle.loads(pr.stdout.read(struct.unpack("<q", head)[0]))
                    if status == "ok":
                        return value
        except Exception:  # noqa: BLE001 -- any trouble with a worker: this thread does the polish
            pass
        finally:
            if pr.poll() is None:
                self._free.put(pr)
            else:  # (found dead, before or during the request: it leaves the count, nobody waits for it again)
                with self._lock:
                    self._alive -= 1
        return polish_reference(*args)

    def close(self):
        for pr in self._procs:
            try:
                pr.stdin.close()
            except Exception:  # noqa: BLE001
                pass
        for pr in self._procs:
            try:
                pr.wait(timeout=2.0)
            except Exception:  # noqa: BLE001
                pr.kill()
            try:
                pr.stdout.close()
            except Exception:  # noqa: BLE001
                pass
        self._procs = []


_POLISH_WORKERS = None


def polish_workers():
    """The process-wide `PolishWorkers`, created on first use.  How many (`XM_POLISH_WORKERS`): up to four, and at most
    one per two CPUs of this rank's share; one rank alone starts them at once, several ranks on a node start theirs
    with the first search that needs a polish."""
    import os

    global _POLISH_WORKERS
    if _POLISH_WORKERS is None:
        local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
        n = int(os.environ.get("XM_POLISH_WORKERS", "0")) or max(1, min(4, _cpu_share() // (2 * local_world)))
        _POLISH_WORKERS = PolishWorkers(n, lazy=local_world > 1)
    return _POLISH_WORKERS


def _solve_native(sl, coords, pivot, target_idx, index_width, method, p0_only, threads=None, polish="exact"):
    """scipy's differential_evolution(best1bin, tol=0.01, seed=42) restated natively: the generations
    run in libxmris_hip.so (same RandomState stream, same trial vectors as scipy given equal objective
    values, objectives vectorised over host cores), the final L-BFGS-B polish is scipy's, exactly as
    `DifferentialEvolutionSolver.solve` does it."""
    import scipy.optimize

    import time

    t0 = time.perf_counter()
    obj = NativeObjective(sl, coords, pivot, target_idx, index_width, method)
    if threads is not None:  # several searches in flight share the host's cores
        obj.set_threads(max(1, int(threads)))
    rc, x, fun, nfev, nit = obj.de(p0_only)  # worker pool spins for the duration of the generations
    t1 = time.perf_counter()
    # the polish's isolated evaluations below run serially (the pool is parked outside xm_solver_de)
    bounds = [(-180.0, 180.0)] if p0_only else [(-180.0, 180.0), (-4000.0, 4000.0)]
    skipped = False
    if polish == "exact":
        # What scipy's polish does in the usual case is NOTHING: L-BFGS-B evaluates f and the forward-difference
        # gradient at the generations' best member, finds the projected gradient below pgtol = 1e-5 (the scores are
        # ~1e-3 and the parameters are degrees: gradients of 1e-8 ... 1e-6 at a converged population) and returns that
        # member -- `result.fun < fun` fails and differential_evolution keeps it (measured: 19 of 20 ACME searches,
        # nit = 0, nfev = 3).  That test is made here with the native objective (three evaluations, one native call;
        # the noise of a difference quotient over steps of 1e-8 is ~1e-11, five orders below pgtol); only when it does
        # not hold with a factor of two to spare does the polish run -- and then on the reference's own route, scipy's
        # minimiser on the numpy objective, whose end point the last bits of the objective decide.  Either way the
        # result is the reference's whenever the generations took the same decisions.
        lo_b = np.array([b_[0] for b_ in bounds])
        hi_b = np.array([b_[1] for b_ in bounds])
        xc = np.clip(np.asarray(x, dtype=np.float64), lo_b, hi_b)
        _, g0 = obj.fg(xc, lo_b, hi_b)
        pg = np.where(g0 < 0, np.maximum(xc - hi_b, g0), np.minimum(xc - lo_b, g0))  # L-BFGS-B's projgr, both bounds set
        if float(np.max(np.abs(pg))) <= 0.5 * 1e-5:
            skipped = True
            res = scipy.optimize.OptimizeResult(x=xc, fun=fun, nfev=len(bounds) + 1, nit=0, success=True)
        else:
            polish = "numpy"
    if skipped:
        pass
    elif polish == "numpy":
        # The polish walks a finite-difference gradient (steps of 1e-8 degrees): on a flat landscape (pure noise, the
        # README quick start) the LAST BITS of the objective decide where it ends, and the native objective's differ
        # from numpy's (vectorised log / sincos recurrence, its own summation order).  Driving scipy's minimiser with
        # the numpy objective from the generations' best member reproduces the reference's polish bit for bit
        # whenever the generations took the same decisions -- at ~0.15 ms per evaluation instead of 6 us, which a
        # single accessor call can afford and a stream of datasets cannot.
        fn = _numpy_objective(sl, coords, pivot, target_idx, index_width, method)
        fun = float(fn(x))
        res = scipy.optimize.minimize(fn, np.copy(x), method="L-BFGS-B", bounds=bounds)
    else:
        res = polish_lbfgsb(obj, np.copy(x), bounds)
    nfev += res.nfev
    lo = np.array([b[0] for b in bounds])
    hi = np.array([b[1] for b in bounds])
    polished = bool(not skipped and res.fun < fun and res.success and np.all(res.x <= hi) and np.all(lo <= res.x))
    if polished:
        x, fun = res.x, float(res.fun)
    opt = scipy.optimize.OptimizeResult(x=x, fun=fun, nfev=nfev, nit=nit, success=(rc == 0), polished=polished,
                                        polish_route="none" if skipped else polish,
                                        t_generations=t1 - t0, t_polish=time.perf_counter() - t1)
    return opt


def solve(sl: np.ndarray, coords: np.ndarray, pivot: float, target_idx: int, index_width: int,
          method: str = "acme", p0_only: bool = False, disp: bool = False, engine: str = "native", threads=None,
          polish: str = "exact"):
    """phasing.py:257-287.  Returns (p0, p1, OptimizeResult).  engine="native" (default) runs the
    optimiser's generations in libxmris_hip.so; engine="scipy" calls scipy's driver with the numpy
    objectives above (the reference's own route, ~15x slower; kept for cross-checks).  `polish` (native engine):
    "exact" (default) -- the projected-gradient test scipy's polish starts with is made natively, and only a search
    that does not pass it is polished, on the reference's route (numpy objective); "numpy" -- always that route;
    "native" -- scipy's compiled L-BFGS-B core on the native objective (round 3's streaming default: its end point
    can differ from the reference's by ~1e-3 degrees when a polish iterates).  See `_solve_native`."""
    import scipy.optimize

    sl = np.asarray(sl, dtype=np.complex128)
    coords = np.asarray(coords, dtype=np.float64)
    if method not in METHODS:
        raise ValueError(MSG_METHOD)
    if engine == "native" and len(sl) >= 2:
        opt = _solve_native(sl, coords, pivot, target_idx, index_width, method, p0_only, threads, polish)
        return float(opt.x[0]), (float(opt.x[1]) if not p0_only else 0.0), opt
    if method == "acme":
        fn, args = acme_score, (sl, coords, pivot)
    elif method == "peak_minima":
        fn, args = peak_minima_score, (sl, coords, pivot, target_idx, index_width)
    elif method == "positivity":
        fn, args = roi_positivity_score, (sl, coords, pivot, target_idx, index_width)
    else:
        raise ValueError(MSG_METHOD)
    bounds = [(-180.0, 180.0)] if p0_only else [(-180.0, 180.0), (-4000.0, 4000.0)]
    opt = scipy.optimize.differential_evolution(
        fn, bounds=bounds, args=args, strategy="best1bin", tol=0.01, seed=42, disp=disp
    )
    p0 = float(opt.x[0])
    p1 = float(opt.x[1]) if not p0_only else 0.0
    return p0, p1, opt
