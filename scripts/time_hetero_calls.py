"""Wall time of run_stream(speculate=True) over calls of 16 distinct heterogeneous datasets (bench.py's `heterogeneous`
note), per call, with the per-dataset marks of the slowest call."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xmris_amd import pipeline  # noqa: E402

nv, nt, N = 65536, 4096, 8192
dt = 1 / 5000.0
device = torch.device("cuda", 0)
x0, t = bench.synth_fids(torch, nv, nt, dt, 0, nv, device, torch.complex64, seed=42, star=nv // 3)
outs = [torch.empty((nv, N), dtype=torch.complex64, device=device) for _ in range(2)]
plan = pipeline.make_plan(x0, t, N, 5.0)
pipeline.run_stream([x0, x0], outs, plan, speculate=True)
seed = 0
for call in range(4):
    sets = [bench.synth_hetero(torch, nv, nt, dt, seed + k, device, torch.complex64)[0] for k in range(16)]
    seed += 16
    torch.cuda.synchronize()
    trace = []
    t0 = time.perf_counter()
    res = pipeline.run_stream(sets, [outs[k % 2] for k in range(16)], plan, speculate=True, trace=trace)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"call {call}: {1e3 * wall:.2f} ms = {1e3 * wall / 16:.3f} ms per dataset; polished on the reference's route: "
          f"{sum(1 for r in res if r.timing.get('polish_route') == 'numpy')}, hedged {sum(bool(r.hedged) for r in res)}; "
          f"waited for a polish (ms): {[round(r.timing.get('polish_ms', 0.0), 2) for r in res if r.timing.get('polish_route') == 'numpy']}; "
          f"collect->solved (ms): {[round(1e3 * (e['t_solved'] - e['t_collect']), 2) for e in trace]}", flush=True)
    del sets
