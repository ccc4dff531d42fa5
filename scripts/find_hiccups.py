#!/usr/bin/env python3
"""Repeat the driver-sized call (20 datasets of the roofline shape) and show, for the slowest repetitions, where the
extra milliseconds sit: host-side gaps between consecutive datasets (t_start) and device-side periods (main0 events)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xmris_amd import pipeline
nv, nt, N = 65536, 4096, 8192
xs = [bench.synth_fids(torch, nv, nt, 2e-4, 0, nv, torch.device("cuda"), torch.complex64, seed=42 + k, star=(nv // 3 + k * 1000) % nv)[0] for k in range(4)]
t = np.arange(nt) * 2e-4
plan = pipeline.make_plan(xs[0], t, N, 5.0)
outs = [torch.empty((nv, N), dtype=torch.complex64, device="cuda") for _ in range(2)]
import gc
run = lambda k, tr: pipeline.run_stream([xs[i % 4] for i in range(k)], [outs[i % 2] for i in range(k)], plan, speculate=True, trace=tr)
for _ in range(4):
    run(8, None)
gc.collect(); gc.freeze()
torch.cuda.synchronize()
reps = []
for rep in range(int(os.environ.get("REPS", 30))):
    tr = []
    torch.cuda.synchronize(); t0 = time.perf_counter(); res = run(20, tr); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    reps.append((dt, tr, res, t0))
ms = np.array([r[0] for r in reps]) / 20
print("ms/step per repetition:", " ".join(f"{v:.3f}" for v in ms))
print(f"median {np.median(ms):.4f}  min {ms.min():.4f}  max {ms.max():.4f}  >1.35: {(ms > 1.35).sum()} of {len(ms)}")
for dt, tr, res, t0 in sorted(reps, key=lambda r: -r[0])[:2]:
    per = [tr[i]["main0"].elapsed_time(tr[i + 1]["main0"]) for i in range(19)]
    host = [(tr[i + 1]["t_solved"] - tr[i]["t_solved"]) * 1e3 for i in range(19)]
    first = (tr[0]["t_solved"] - t0) * 1e3
    e0 = tr[0]
    print(f"   dataset 0: selection waited from {(e0['t_start'] - t0) * 1e3:.2f} to {(e0['t_exchanged'] - t0) * 1e3:.2f} ms, "
          f"search (generations {res[0].timing.get('generations_ms', 0):.2f} + polish {res[0].timing.get('polish_ms', 0):.2f} ms) "
          f"consumed at {(e0['t_solved'] - t0) * 1e3:.2f} ms")
    print(f"rep of {dt:.2f} ms: first (p0, p1) ready at {first:.2f} ms; device periods", " ".join(f"{v:.2f}" for v in per))
    print("   host: interval between consecutive datasets' (p0, p1) becoming available", " ".join(f"{v:.2f}" for v in host))
    print("   search gen ms", " ".join(f"{r.timing.get('generations_ms', 0):.2f}" for r in res))
