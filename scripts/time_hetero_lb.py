#!/usr/bin/env python3
"""Hit rate and step time of the speculative schedule on the heterogeneous family for other apodisations than the
benchmark's lb = 5 Hz (the candidate band follows the window: 0.9 x its weight inside the first 512 samples)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xmris_amd import pipeline
nv, nt, N = int(os.environ.get("NV", 65536)), 4096, 8192
n_sets = int(os.environ.get("NSETS", 12))
sets = [bench.synth_hetero(torch, nv, nt, 2e-4, 50 + s, "cuda", torch.complex64)[0] for s in range(n_sets)]
t = np.arange(nt) * 2e-4
outs = [torch.empty((nv, N), dtype=torch.complex64, device="cuda") for _ in range(2)]
for lb in (0.0, 1.0, 2.0, 5.0, 20.0):
    plan = pipeline.make_plan(sets[0], t, N, lb)
    pipeline.run_stream(sets[:3], [outs[k % 2] for k in range(3)], plan, speculate=True)
    torch.cuda.synchronize()
    trace = []
    t0 = time.perf_counter()
    res = pipeline.run_stream(sets, [outs[k % 2] for k in range(n_sets)], plan, speculate=True, trace=trace)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n_sets * 1e3
    per = [trace[i]["main0"].elapsed_time(trace[i + 1]["main0"]) for i in range(n_sets - 1)]
    print(f"lb {lb:5.1f} Hz: band {plan.extra['guess_band']:.2f}  hits {sum(r.speculation == 'hit' for r in res)}/{n_sets}  "
          f"{ms:.3f} ms/step  device period median {np.median(per):.3f} ms")
