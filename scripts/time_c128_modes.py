"""complex128 hot shape (32,768 x 4096 -> 8192): the first-generation kernel's modes, 8 back-to-back launches each."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmris_amd import device as dev
nv, nt, N = 32768, 4096, 8192
x = torch.view_as_complex(torch.randn(nv, nt, 2, device="cuda", dtype=torch.float64))
w = torch.rand(N, device="cuda", dtype=torch.float64)
ph = torch.view_as_complex(torch.randn(N, 2, device="cuda", dtype=torch.float64))
out = torch.empty(nv, N, dtype=torch.complex128, device="cuda")
am = torch.empty(nv, device="cuda", dtype=torch.float64); ai = torch.empty(nv, dtype=torch.int32, device="cuda")
def run(name, **kw):
    f = lambda: dev.pipeline_fused(x, N, 0, window=w, out=out, absmax2=am, argidx=ai, **kw)
    for _ in range(2): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 8)
    ms = float(np.median(ts))
    by = nv * 16 * (nt + (N if kw.get("want_out", True) else 0))
    print(f"{name:34s} {ms:8.4f} ms  {by/ms/1e6:8.1f} GB/s")
run("write", want_out=True)
run("write + table", want_out=True, phase_table=ph)
run("write + table + max (value only)", want_out=True, phase_table=ph, want_argmax=True, argmax_value_only=True)
run("max only (value only)", want_out=False, want_argmax=True, argmax_value_only=True)
run("write + ramp", want_out=True, phase_ramp=(0.3, -1e-3))
run("write + ramp + max (value only)", want_out=True, phase_ramp=(0.3, -1e-3), want_argmax=True, argmax_value_only=True)
