"""Run one fused-kernel variant a few times (for rocprofv3 counter passes).
VARIANT in {write, pre, main, all, guess};  NV voxels (default 65536).
  all   = the main pass of the speculative schedule (write + phase + per-row maxima, value only)
  guess = the windowed L1 norms (xm_row_l1) that replace the pre-pass in that schedule"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmris_amd import device as dev
nv, nt, N = int(os.environ.get("NV", 65536)), int(os.environ.get("NT", 4096)), int(os.environ.get("NOUT", 8192))
var = os.environ.get("VARIANT", "main")
x = torch.view_as_complex(torch.randn(nv, nt, 2, device="cuda"))
w = torch.rand(N, device="cuda")
ph = torch.view_as_complex(torch.randn(N, 2, device="cuda"))
out = torch.empty(nv, N, dtype=torch.complex64, device="cuda")
am = torch.empty(nv, device="cuda"); ai = torch.empty(nv, dtype=torch.int32, device="cuda")
kw = {"guess": {}, "write": dict(want_out=True), "pre": dict(want_out=False, want_argmax=True, argmax_value_only=True),
      "main": dict(want_out=True, phase_table=ph), "all": dict(want_out=True, phase_table=ph, want_argmax=True, argmax_value_only=True)}[var]
for _ in range(int(os.environ.get("REPS", 3))):
    if var == "guess":
        dev.row_l1(x, w, 0, out=am)
    else:
        dev.pipeline_fused(x, N, 0, window=w, out=out, absmax2=am, argidx=ai, **kw)
torch.cuda.synchronize()
