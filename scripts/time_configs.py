"""Pre-pass (per-row maxima, value only) and main-pass (phase ramp where the kernel applies it natively, else a phase
table) timing of the BASELINE.json parity configs (C1, C2, C5) next to C3, complex64.  The rate column is the
device's time per launch with launches back to back; the figure behind it is what rounds 2 and 3 quoted -- an event pair
around ONE launch on an idle queue, which also times the Python wrapper (3-15 % more on the 0.1-0.2 ms launches)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmris_amd import device as dev
REPS = 20
def run(name, nv, nt, N):
    x = torch.view_as_complex(torch.randn(nv, nt, 2, device="cuda"))
    w = torch.rand(N, device="cuda"); ph = torch.view_as_complex(torch.randn(N, 2, device="cuda"))
    out = torch.empty(nv, N, dtype=torch.complex64, device="cuda")
    am = torch.empty(nv, device="cuda"); ai = torch.empty(nv, dtype=torch.int32, device="cuda")
    native = dev.ramp_native(x, N, 0)
    for label, kw in (("pre ", dict(want_out=False, want_argmax=True, argmax_value_only=True)),
                      ("main", dict(want_out=True, phase_ramp=(0.7, 0.0085)) if native else dict(want_out=True, phase_table=ph))):
        f = lambda: dev.pipeline_fused(x, N, 0, window=w, out=out, absmax2=am, argidx=ai, **kw)
        for _ in range(3): f()
        torch.cuda.synchronize(); ts, tb = [], []
        for _ in range(7):  # one launch between two events on an idle queue: includes the wrapper's ~30 us of host work
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        for _ in range(5):  # REPS launches back to back (the queue never drains): the device's time per launch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f(); e0.record()
            for _ in range(REPS): f()
            e1.record(); torch.cuda.synchronize(); tb.append(e0.elapsed_time(e1) / REPS)
        ms1, ms = float(np.median(ts)), float(np.median(tb)); by = nv * 8 * (nt + (N if label == "main" else 0))
        print(f"{name:34s} {label} {ms:8.4f} ms  {by/ms/1e6:8.1f} GB/s  {nv/ms/1e3:8.2f} M spectra/s   (single launch on an idle queue: {ms1:.4f} ms)")
run("C1 5 x 1024 -> 2048", 5, 1024, 2048)
run("C2 16384 x 2048 -> 4096", 16384, 2048, 4096)
run("C2' 16384 x 2048 (no-op zero fill)", 16384, 2048, 2048)
run("C3 65536 x 4096 -> 8192", 65536, 4096, 8192)
run("C5 32768 x 1536", 32768, 1536, 1536)
run("prime 32768 x 1531 (Bluestein)", 32768, 1531, 1531)
run("65536 x 4096 (no zero fill)", 65536, 4096, 4096)
run("32768 x 8192 -> 16384", 32768, 8192, 16384)
