import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from xmris_amd import device as dev
n_in, N = 4096, 8192
x1 = torch.view_as_complex(torch.randn(1, n_in, 2, device="cuda", dtype=torch.float64))
w = torch.rand(N, device="cuda", dtype=torch.float64)
out_dev = torch.empty((1, N), dtype=torch.complex128, device="cuda")
out_pin = torch.empty((1, N), dtype=torch.complex128, pin_memory=True)
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("1-row c128 spectrum -> device memory: %.1f us" % t(lambda: dev.pipeline_fused(x1, N, 0, window=w, out=out_dev)))
print("1-row c128 spectrum -> pinned host:   %.1f us" % t(lambda: dev.pipeline_fused(x1, N, 0, window=w, out=out_pin)))
x2 = x1.expand(2, n_in).contiguous(); o2 = torch.empty((2, N), dtype=torch.complex128, device="cuda")
print("2-row (two workgroups) -> device:     %.1f us" % t(lambda: dev.pipeline_fused(x2, N, 0, window=w, out=o2)))
