#!/usr/bin/env python3
"""Where does a step of the speculative schedule go on the heterogeneous family (bench.synth_hetero)?  Per dataset:
search cost (nfev, generations, polish), guess/selection kernels, main kernel, device period."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from xmris_amd import pipeline  # noqa: E402

nv, nt, N = int(os.environ.get("NV", 65536)), 4096, 8192
n_sets = int(os.environ.get("NSETS", 12))
dtype = torch.complex64
sets = [bench.synth_hetero(torch, nv, nt, 2e-4, s, "cuda", dtype)[0] for s in range(n_sets)]
t = np.arange(nt) * 2e-4
plan = pipeline.make_plan(sets[0], t, N, 5.0)
outs = [torch.empty((nv, N), dtype=dtype, device="cuda") for _ in range(2)]
pipeline.run_stream(sets[:4], [outs[k % 2] for k in range(4)], plan, speculate=True)
torch.cuda.synchronize()
if not os.environ.get("XM_POLISH_THREADS") and os.environ.get("XMRIS_AMD_POLISH", "exact") == "exact":
    # steady state: the polish workers are up (a stream started cold polishes in-process until they are, ~1 s)
    from xmris_amd import autophase_solver as aps

    pw = aps.polish_workers()
    pw.start()
    t_w = time.time()
    while pw._alive < pw._n and time.time() - t_w < 30:
        time.sleep(0.05)
    print(f"polish workers ready: {pw._alive} of {pw._n} after {time.time() - t_w:.1f} s")
for rep in range(2):
    trace = []
    t0 = time.perf_counter()
    res = pipeline.run_stream(sets, [outs[k % 2] for k in range(n_sets)], plan, speculate=True, trace=trace)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    print(f"rep {rep}: {wall / n_sets:.3f} ms/step")
for i, (e, r) in enumerate(zip(trace, res)):
    period = e["main0"].elapsed_time(trace[i + 1]["main0"]) if i + 1 < n_sets else float("nan")
    print(f"  set {i:2d} {r.speculation:8s} row {r.flat_index // N:6d} nfev {r.nfev:5d} gen {r.timing.get('generations_ms', 0):6.2f} ms "
          f"polish {r.timing.get('polish_ms', 0):5.2f} ms  guess {e['pre0'].elapsed_time(e['pre1']) * 1e3:6.1f} us  "
          f"main {e['main0'].elapsed_time(e['main1']):.3f} ms  period {period:.3f} ms  "
          f"wait {1e3 * (e['t_exchanged'] - e['t_start']):.2f} ms  exchange->use {1e3 * (e['t_solved'] - e['t_exchanged']):.2f} ms")
