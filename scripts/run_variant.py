"""Run one fused-kernel variant a few times (for rocprofv3 counter passes) on the C3 shape (NV x 4096 -> 8192, c64).
VARIANT:
  all   = main pass of the speculative schedule: k_zf2p mode 13 (write + phase ramp + global arg-max key)
  main  = main pass of the classic schedule: k_zf2p mode 9 (write + phase ramp)
  table = write + phase TABLE (k_zf2p mode 3; the first-generation kernel with XM_ZF2_GEN1=1)
  write = write only;  pre = arg-max pre-pass of the classic schedule
  guess = the sub-sampled windowed L1 norms (xm_row_l1 with an arg-max key) that replace the pre-pass (round 2)
  coarse = the guess stage: xm_guess_rows (coarse spectra: k_coarse_mfma, or k_zf2p<512-plan, 4, 17> with XM_GUESS_FFT=1) + xm_guess_refine
  rows  = write + phase ramp + per-row maxima (value only): the complex128 main pass of the speculative schedule
DTYPE=c128 runs the complex128 kernels (`all`: k_zf2d with the arg-max key and the prefetch, mode 221)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmris_amd import device as dev
nv, nt, N = int(os.environ.get("NV", 65536)), int(os.environ.get("NT", 4096)), int(os.environ.get("NOUT", 8192))
var = os.environ.get("VARIANT", "all")
rd = torch.float64 if os.environ.get("DTYPE", "c64") == "c128" else torch.float32
x = torch.view_as_complex(torch.randn(nv, nt, 2, device="cuda", dtype=rd))
w = torch.rand(N, device="cuda", dtype=rd)
ph = torch.view_as_complex(torch.randn(N, 2, device="cuda", dtype=rd))
out = torch.empty(nv, N, dtype=x.dtype, device="cuda")
am = torch.empty(nv, device="cuda", dtype=rd); ai = torch.empty(nv, dtype=torch.int32, device="cuda")
key = dev.new_argmax_key("cuda")
gmax = torch.empty(1, device="cuda"); gflat = torch.empty(1, dtype=torch.int64, device="cuda")
ramp = (0.7, 0.0085)
kw = {"guess": {}, "coarse": {}, "write": dict(want_out=True), "pre": dict(want_out=False, want_argmax=True, argmax_value_only=True),
      "table": dict(want_out=True, phase_table=ph), "main": dict(want_out=True, phase_ramp=ramp),
      "all": dict(want_out=True, phase_ramp=ramp, global_key=key),
      "rows": dict(want_out=True, phase_ramp=ramp, want_argmax=True, argmax_value_only=True)}[var]
est = torch.empty(nv, device="cuda", dtype=torch.float32)
wkey = dev.new_argmax_key("cuda")
hflat = torch.zeros(1, dtype=torch.int64, pin_memory=True)
hmax = torch.zeros(1, dtype=torch.float32, pin_memory=True)
row = torch.empty((1, nt), dtype=torch.complex128, device="cuda")
w32 = w.to(torch.float32)
res128 = dev.new_key_result()
for _ in range(int(os.environ.get("REPS", 3))):
    if var == "coarse":
        dev.guess_rows(x, N, w32, est, key)
        dev.guess_refine(x, N, w32, est, key, wkey, hmax, hflat, row)
    elif var == "guess":
        dev.row_l1(x, w, 0, n_used=2304, sub_step=8, key=key)
        dev.argmax_key_take(key, N, gmax, gflat)
    elif var == "all" and rd == torch.float64:  # complex128: the launch decodes its key itself (result record)
        dev.pipeline_fused(x, N, 0, window=w, out=out, key_result=res128, **kw)
    else:
        dev.pipeline_fused(x, N, 0, window=w, out=out, absmax2=None if var == "all" else am, argidx=None if var == "all" else ai, **kw)
        if var == "all":
            dev.argmax_key_take(key, N, gmax, gflat)
torch.cuda.synchronize()
