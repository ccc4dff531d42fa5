"""Main kernel of the speculative schedule alone (65,536 x 4096 -> 8192 complex64, write + ramp + arg-max key): mean of
back-to-back launches, for same-box A/B runs of builds (XMRIS_AMD_LIB=<another libxmris_hip.so>)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmris_amd import device as dev
nv, nt, N = 65536, 4096, 8192
x = torch.view_as_complex(torch.randn(nv, nt, 2, device="cuda"))
w = torch.rand(N, device="cuda")
outs = [torch.empty(nv, N, dtype=torch.complex64, device="cuda") for _ in range(2)]
key, rec = dev.new_argmax_key("cuda"), dev.new_key_result()
f = lambda i: dev.pipeline_fused(x, N, 0, window=w, out=outs[i % 2], phase_ramp=(0.7, 0.0085), global_key=key, key_result=rec)
for i in range(6): f(i)
torch.cuda.synchronize(); ts = []
for rep in range(int(os.environ.get("REPS", 5))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(16): f(i)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 16)
print(f"{os.environ.get('XMRIS_AMD_LIB', 'default'):60s} median {np.median(ts):.4f} ms  min {min(ts):.4f}  max {max(ts):.4f}  ({nv * 8 * (nt + N) / np.median(ts) / 1e9:.2f} TB/s)")
