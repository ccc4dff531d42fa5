"""(A measurement tool that uses the CPU oracle as its checker: it lives under tests/, the only place besides
__graft_entry__.smoke() and bench.py's cpu_baseline leg that may import oracle/.)
Host-only timing of xm_solver_de on a benchmark-like slice (no GPU needed).
XM_SOLVER_BATCH=1 reproduces the one-trial-per-hand-off schedule."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import xmris_oracle as orc
from xmris_amd import autophase_solver as aps
nt, N = 4096, 8192
t = np.arange(nt) / 5000.0
rng = np.random.default_rng(1)
x = 2.0 * sum(a * np.exp(-d * t) * np.exp(2j * np.pi * f * t) for a, d, f in zip((1.0, .5, .3), (20., 33., 25.), (300., -800., 1100.)))
x = (x + 0.014 * (rng.standard_normal(nt) + 1j * rng.standard_normal(nt)))[None, :]
spec, inf = orc.pipeline_values(x, t, N, 5.0, solve=False)
sl, fr, pv, ti = inf["slice"], inf["freq"], inf["pivot"], inf["target_idx"]
for batch in (int(b) for b in os.environ.get("BATCHES", "1,2,4,6,8,15,30").split(",")):
    os.environ["XM_SOLVER_BATCH"] = str(batch)
    obj = aps.NativeObjective(sl, fr, pv, ti, 1, "acme")
    for thr in (int(a) for a in (sys.argv[1:] or ["0"])):
        n = obj.set_threads(thr)
        best = 1e9
        for rep in range(8):
            e0 = obj.evaluations()
            ta = time.perf_counter(); rc, xx, fun, nfev, nit = obj.de(False); tb = time.perf_counter()
            best = min(best, tb - ta)
        print(f"batch={batch:2d} threads={n:2d}: de {1e3*best:.3f} ms, nfev {nfev} ({obj.evaluations() - e0} evaluated), nit {nit}, x=({xx[0]:.4f},{xx[1]:.4f})")
