"""PCIe-inclusive rate of the drop-in accessor path on HOST ndarrays (never used as bench `value`):
numpy FIDs -> LabeledArray -> .xmr.spectral_pipeline(...) -> host ndarray, one dataset, next to what the link does
(pinned copies of the same sizes, both directions at once) and to the staged route of round 2 (upload, compute, download
one after the other: XMRIS_AMD_HOST_STREAM_MIN set beyond the input's size)."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xmris_amd as xm
from xmris_amd import hostpath
nv, nt, N = int(os.environ.get("NV", 16384)), 4096, 8192
t = np.arange(nt) / 5000.0
rng = np.random.default_rng(0)
base = sum(a * np.exp(-d * t) * np.exp(2j * np.pi * f * t) for a, d, f in zip((1.0, .5, .3), (20., 33., 25.), (300., -800., 1100.)))
amp = 0.5 + rng.random(nv); amp[nv // 3] = 2.0
x = (amp[:, None] * base[None, :]).astype(np.complex64)
x += (0.014 * (rng.standard_normal((nv, nt)) + 1j * rng.standard_normal((nv, nt)))).astype(np.complex64)
# the link: pinned copies of the input's and the result's sizes, one direction at a time and both at once
hin = torch.empty(x.nbytes, dtype=torch.uint8, pin_memory=True); din = torch.empty(x.nbytes, dtype=torch.uint8, device="cuda")
hout = torch.empty(2 * x.nbytes, dtype=torch.uint8, pin_memory=True); dout = torch.empty(2 * x.nbytes, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for label, fn in (("H2D pinned", lambda: din.copy_(hin, non_blocking=True)), ("D2H pinned", lambda: hout.copy_(dout, non_blocking=True))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label}: {(x.nbytes if label[0] == 'H' else 2 * x.nbytes) / dt / 1e9:.1f} GB/s")
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.cuda.stream(s1): din.copy_(hin, non_blocking=True)
with torch.cuda.stream(s2): hout.copy_(dout, non_blocking=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"both at once: {3 * x.nbytes / dt / 1e9:.1f} GB/s in total; the link's time for this dataset (in, then out): "
      f"{1e3 * dt:.1f} ms overlapped")
del hin, din, hout, dout
for mode in ("chunked (hostpath.run_host)", "staged (round 2)"):
    if mode.startswith("staged"):
        os.environ["XMRIS_AMD_HOST_STREAM_MIN"] = str(1 << 60)
    for rep in range(4):
        tm = {}
        t0 = time.perf_counter()
        fid = xm.LabeledArray(x, dims=["voxel", "time"], coords={"voxel": np.arange(nv), "time": t})
        spec = fid.xmr.spectral_pipeline(target_points=N, lb=5.0)
        out = spec.values
        dt = time.perf_counter() - t0
        print(f"{mode} rep {rep}: {nv} spectra in {1e3 * dt:.1f} ms -> {nv / dt / 1e6:.3f} M spectra/s  "
              f"({(x.nbytes + out.nbytes) / dt / 1e9:.1f} GB/s over the host link), p0={spec.attrs['phase_p0']:.4f}")
        del spec, out
tm = {}
y, res, plan = hostpath.run_host(x, t, N, 5.0, timing=tm)
print("split of one chunked call:", {k: (round(1e3 * v, 1) if k.endswith("_s") else v) for k, v in tm.items()})
