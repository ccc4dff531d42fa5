"""Kernel times of the speculative schedule's pieces next to the pre-pass it replaces (65536 x 4096 -> 8192 c64)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmris_amd import device as dev
nv, nt, N = 65536, 4096, 8192
x = torch.view_as_complex(torch.randn(nv, nt, 2, device="cuda"))
w = torch.rand(N, device="cuda"); ph = torch.view_as_complex(torch.randn(N, 2, device="cuda"))
out = torch.empty(nv, N, dtype=torch.complex64, device="cuda")
am = torch.empty(nv, device="cuda"); ai = torch.empty(nv, dtype=torch.int32, device="cuda"); nrm = torch.empty(nv, device="cuda")
def timeit(f, reps=15):
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
ref = (x.abs() * w[:nt]).sum(dim=1)
dev.row_l1(x, w, 0, out=nrm); torch.cuda.synchronize()
print("row_l1 rel err", float(((nrm - ref).abs() / ref).max()))
print(f"row_l1 (guess)                : {timeit(lambda: dev.row_l1(x, w, 0, out=nrm)):.4f} ms")
print(f"pre-pass (value only)         : {timeit(lambda: dev.pipeline_fused(x, N, 0, window=w, want_out=False, want_argmax=True, absmax2=am, argidx=ai, argmax_value_only=True)):.4f} ms")
print(f"main pass                     : {timeit(lambda: dev.pipeline_fused(x, N, 0, window=w, out=out, phase_table=ph)):.4f} ms")
print(f"main pass + maxima (value)    : {timeit(lambda: dev.pipeline_fused(x, N, 0, window=w, out=out, phase_table=ph, want_argmax=True, absmax2=am, argidx=ai, argmax_value_only=True)):.4f} ms")
print(f"main pass + maxima (+ index)  : {timeit(lambda: dev.pipeline_fused(x, N, 0, window=w, out=out, phase_table=ph, want_argmax=True, absmax2=am, argidx=ai)):.4f} ms")
