"""Time the fused-kernel variants on the C3 shape (65536 x 4096 -> 8192, c64)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmris_amd import device as dev
nv, nt, N = int(os.environ.get("NV", 65536)), 4096, 8192
x = torch.view_as_complex(torch.randn(nv, nt, 2, device="cuda"))
w = torch.rand(N, device="cuda")
ph = torch.view_as_complex(torch.randn(N, 2, device="cuda"))
out = torch.empty(nv, N, dtype=torch.complex64, device="cuda")
am = torch.empty(nv, device="cuda"); ai = torch.empty(nv, dtype=torch.int32, device="cuda")
def run(name, **kw):
    f = lambda: dev.pipeline_fused(x, N, 0, window=w, out=out, absmax2=am, argidx=ai, **kw)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    print(f"{name:28s} {ms:7.3f} ms   read-only {nv*nt*8/ms/1e6:7.1f} GB/s   r+w {nv*(nt+N)*8/ms/1e6:7.1f} GB/s")
run("write", want_out=True)
run("write+amax", want_out=True, want_argmax=True)
run("amax only (prepass)", want_out=False, want_argmax=True)
run("write+phase (main)", want_out=True, phase_table=ph)
run("write+phase+amax", want_out=True, phase_table=ph, want_argmax=True)
# copy kernels for reference
y = torch.empty_like(x)
for _ in range(2): y.copy_(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); y.copy_(x); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1); print(f"torch copy 2GiB: {ms:.3f} ms  {2*nv*nt*8/ms/1e6:.1f} GB/s")
