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


def default_threads() -> int:
    """Team size for the native objective: a power of two, at most 16 and at most HALF of this process's share of
    the CPUs it may use (scheduler affinity and the cgroup v2 quota, divided by the ranks torchrun started on
    this node)."""
    import os

    if os.environ.get("XM_SOLVER_THREADS"):  # tuning switch
        return max(1, min(32, int(os.environ["XM_SOLVER_THREADS"])))
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
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    # Half of the share, as a power of two (the work units of a batch are 4 evaluations x 4 parts): a team that
    # fills the whole CPU quota while it spins leaves no headroom for the HIP runtime's threads, and a cgroup that
    # overdraws its quota is frozen until the next 100 ms period -- measured as rare 15-25 ms stalls of the whole
    # pipeline with 16 threads on a 16-CPU share, none with 8 (the search takes 0.95 instead of 0.71 ms, still
    # hidden behind the device).
    # Several ranks on one node: only the rank that OWNS a dataset's winning spectrum searches (the others wait,
    # sleep-polling, for the broadcast), so the team is not divided by the number of ranks -- two cores per rank stay
    # reserved for launching and polling, the owner's searches may take half of the rest.
    share = max(1, cpus // 2) if local_world == 1 else max(1, (cpus - 2 * local_world) // 2)
    team = 1
    while team * 2 <= min(16, share):
        team *= 2
    return team


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


def _solve_native(sl, coords, pivot, target_idx, index_width, method, p0_only, threads=None):
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
    res = scipy.optimize.minimize(obj, np.copy(x), method="L-BFGS-B", bounds=bounds)
    nfev += res.nfev
    lo = np.array([b[0] for b in bounds])
    hi = np.array([b[1] for b in bounds])
    polished = bool(res.fun < fun and res.success and np.all(res.x <= hi) and np.all(lo <= res.x))
    if polished:
        x, fun = res.x, float(res.fun)
    opt = scipy.optimize.OptimizeResult(x=x, fun=fun, nfev=nfev, nit=nit, success=(rc == 0), polished=polished,
                                        t_generations=t1 - t0, t_polish=time.perf_counter() - t1)
    return opt


def solve(sl: np.ndarray, coords: np.ndarray, pivot: float, target_idx: int, index_width: int,
          method: str = "acme", p0_only: bool = False, disp: bool = False, engine: str = "native", threads=None):
    """phasing.py:257-287.  Returns (p0, p1, OptimizeResult).  engine="native" (default) runs the
    optimiser's generations in libxmris_hip.so; engine="scipy" calls scipy's driver with the numpy
    objectives above (the reference's own route, ~15x slower; kept for cross-checks)."""
    import scipy.optimize

    sl = np.asarray(sl, dtype=np.complex128)
    coords = np.asarray(coords, dtype=np.float64)
    if method not in METHODS:
        raise ValueError("Method must be 'acme', 'peak_minima', or 'positivity'")
    if engine == "native" and len(sl) >= 2:
        opt = _solve_native(sl, coords, pivot, target_idx, index_width, method, p0_only, threads)
        return float(opt.x[0]), (float(opt.x[1]) if not p0_only else 0.0), opt
    if method == "acme":
        fn, args = acme_score, (sl, coords, pivot)
    elif method == "peak_minima":
        fn, args = peak_minima_score, (sl, coords, pivot, target_idx, index_width)
    elif method == "positivity":
        fn, args = roi_positivity_score, (sl, coords, pivot, target_idx, index_width)
    else:
        raise ValueError("Method must be 'acme', 'peak_minima', or 'positivity'")
    bounds = [(-180.0, 180.0)] if p0_only else [(-180.0, 180.0), (-4000.0, 4000.0)]
    opt = scipy.optimize.differential_evolution(
        fn, bounds=bounds, args=args, strategy="best1bin", tol=0.01, seed=42, disp=disp
    )
    p0 = float(opt.x[0])
    p1 = float(opt.x[1]) if not p0_only else 0.0
    return p0, p1, opt
