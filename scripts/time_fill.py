"""Host-side timeline of the first datasets of a run_stream(speculate=True) call that starts with an empty pipeline
(the driver's --steps 20): when each selection is ready, when each search is done, when each main pass starts."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xmris_amd import pipeline
nv, nt, N = 65536, 4096, 8192
x, t = bench.synth_fids(torch, nv, nt, 1.0 / 5000.0, 0, nv, torch.device("cuda"), torch.complex64)
plan = pipeline.make_plan(x, t, N, 5.0)
out = [torch.empty((nv, N), dtype=torch.complex64, device="cuda") for _ in range(2)]
def run(k):
    trace = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    res = pipeline.run_stream([x] * k, [out[i % 2] for i in range(k)], plan, trace=trace, speculate=True)
    torch.cuda.synchronize()
    return t0, e0, trace, res, time.perf_counter() - t0
for _ in range(3): run(8)
print("65,536 x 4096 -> 8192 c64, 20 datasets per call, pipeline empty at the start; host clock / device events in ms from the call")
for rep in range(3):
    t0, e0, trace, res, el = run(20)
    print(f"run {rep}: {el*1e3:.3f} ms for 20 datasets = {el*1e3/20:.4f} ms/step")
    for i in range(5):
        e, r = trace[i], res[i]
        print(f"  dataset {i}: selection waited from {(e['t_start']-t0)*1e3:6.3f} to {(e['t_exchanged']-t0)*1e3:6.3f}  "
              f"search used at {(e['t_solved']-t0)*1e3:6.3f}  (generations {r.timing.get('generations_ms',0):.3f} + polish {r.timing.get('polish_ms',0):.3f})  "
              f"main pass on the device {e0.elapsed_time(e['main0']):6.3f} .. {e0.elapsed_time(e['main1']):6.3f}")
