"""BASELINE configs[1] (c2) or configs[4] (c5) through the streaming executor, for rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xmris_amd import pipeline  # noqa: E402

nv, nt, N = (16384, 2048, 4096) if (len(sys.argv) < 2 or sys.argv[1] == "c2") else (32768, 1536, 1536)
device = torch.device("cuda", 0)
xs = []
for k in range(4):
    xk, t = bench.synth_fids(torch, nv, nt, 1 / 5000.0, 0, nv, device, torch.complex64, seed=77 + 1009 * k, star=(nv // 3 + k * (nv // 5) + 7 * k) % nv)
    xs.append(xk)
outs = [torch.empty((nv, N), dtype=torch.complex64, device=device) for _ in range(2)]
plan = pipeline.make_plan(xs[0], t, N, 5.0)
K = 100
ins = [xs[k % 4] for k in range(K)]
ots = [outs[k % 2] for k in range(K)]
pipeline.run_stream(ins[:20], ots[:20], plan, speculate=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
pipeline.run_stream(ins, ots, plan, speculate=True)
torch.cuda.synchronize()
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'c2'}: {(time.perf_counter() - t0) / K * 1e3:.4f} ms per dataset")
