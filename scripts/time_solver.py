"""Phase timing of the host part of autophase on the benchmark slice (8192 points)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmris_amd import autophase_solver as aps, device as dev, pipeline
import scipy.optimize
nv, nt, N = 4096, 4096, 8192
t = np.arange(nt) / 5000.0
base = sum(a * np.exp(-d * t) * np.exp(2j * np.pi * f * t) for a, d, f in zip((1.0, .5, .3), (20., 33., 25.), (300., -800., 1100.)))
x = torch.from_numpy(base).to("cuda", torch.complex64)[None, :] * (0.5 + torch.rand(nv, 1, device="cuda"))
x = x + torch.view_as_complex(torch.randn(nv, nt, 2, device="cuda") * 0.014)
plan = pipeline.make_plan(x, t, N, 5.0)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pre = dev.pipeline_fused(x, N, 0, window=plan.window, want_out=False, want_argmax=True)
    amax, flat = dev.argmax_reduce(pre.absmax2, pre.argidx, N); t1 = time.perf_counter()
    row, k = flat // N, flat % N
    x1 = x[row:row + 1].to(torch.complex128)
    w64 = torch.from_numpy(np.ascontiguousarray(plan.window_host)).to(x.device, torch.float64)
    sl = dev.pipeline_fused(x1, N, 0, window=w64).out[0].cpu().numpy(); t2 = time.perf_counter()
    obj = aps.NativeObjective(sl, plan.freq, float(plan.freq[k]), k, 1, "acme"); t3 = time.perf_counter()
    for thr in ((0,) if rep else (0, 1, 4, 8, 12, 16)):
        ta = time.perf_counter(); n = obj.set_threads(thr); rc, xx, fun, nfev, nit = obj.de(False); tb = time.perf_counter()
        print(f"   threads={n}: de {1e3*(tb-ta):.2f} ms for {nfev} evals = {1e6*(tb-ta)/nfev:.1f} us/eval")
    t4 = time.perf_counter(); obj.set_threads(1)
    res = scipy.optimize.minimize(obj, np.copy(xx), method="L-BFGS-B", bounds=[(-180., 180.), (-4000., 4000.)]); t5 = time.perf_counter()
    tab = aps.phase_table(plan.freq, res.x[0], res.x[1], float(plan.freq[k])); ph = torch.from_numpy(tab).to("cuda", torch.complex64); torch.cuda.synchronize(); t6 = time.perf_counter()
    print(f"rep {rep}: prepass+argmax {1e3*(t1-t0):.2f}  slice(c128)+D2H {1e3*(t2-t1):.2f}  create {1e3*(t3-t2):.2f}  de {1e3*(t4-t3):.2f}  polish {1e3*(t5-t4):.2f} ({res.nfev} evals)  table+H2D {1e3*(t6-t5):.2f}")
print("host cpus", os.cpu_count())
