"""Streaming ceiling for the main pass's traffic pattern: xm_zero_fill reads 65536 x 4096 c64 and writes
65536 x 8192 c64 (the same 2 GiB in / 4 GiB out as the fused main kernel, no arithmetic), next to a plain
device copy of 4 GiB."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmris_amd import device as dev
nv, nt, N = 65536, 4096, 8192
x = torch.view_as_complex(torch.randn(nv, nt, 2, device="cuda"))
def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
out = torch.empty(nv, N, dtype=torch.complex64, device="cuda")
import ctypes
from xmris_amd import _lib
def zf():
    _lib.call("xm_zero_fill", x.data_ptr(), out.data_ptr(), nv, nt, N, 0, _lib.XM_C64, torch.cuda.current_stream().cuda_stream)
ms = timeit(zf); print(f"xm_zero_fill 2 GiB -> 4 GiB : {ms:.4f} ms  {nv*8*(nt+N)/ms/1e6:.1f} GB/s")
y = torch.empty_like(out)
ms = timeit(lambda: y.copy_(out)); print(f"device copy 4 GiB         : {ms:.4f} ms  {2*nv*8*N/ms/1e6:.1f} GB/s")
ms = timeit(lambda: out.zero_()); print(f"memset 4 GiB              : {ms:.4f} ms  {nv*8*N/ms/1e6:.1f} GB/s")
w = torch.rand(N, device="cuda"); ph = torch.view_as_complex(torch.randn(N, 2, device="cuda"))
ms = timeit(lambda: dev.pipeline_fused(x, N, 0, window=w, out=out, phase_table=ph)); print(f"fused main pass           : {ms:.4f} ms  {nv*8*(nt+N)/ms/1e6:.1f} GB/s")
