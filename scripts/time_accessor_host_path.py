"""PCIe-inclusive rate of the drop-in accessor path on HOST ndarrays (never used as bench `value`):
numpy complex64 FIDs -> LabeledArray -> .xmr.spectral_pipeline(...) -> .values (numpy), one dataset."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xmris_amd as xm
nv, nt, N = int(os.environ.get("NV", 16384)), 4096, 8192
t = np.arange(nt) / 5000.0
rng = np.random.default_rng(0)
base = sum(a * np.exp(-d * t) * np.exp(2j * np.pi * f * t) for a, d, f in zip((1.0, .5, .3), (20., 33., 25.), (300., -800., 1100.)))
amp = 0.5 + rng.random(nv); amp[nv // 3] = 2.0
x = (amp[:, None] * base[None, :]).astype(np.complex64)
x += (0.014 * (rng.standard_normal((nv, nt)) + 1j * rng.standard_normal((nv, nt)))).astype(np.complex64)
for rep in range(3):
    t0 = time.perf_counter()
    fid = xm.LabeledArray(x, dims=["voxel", "time"], coords={"voxel": np.arange(nv), "time": t})
    spec = fid.xmr.spectral_pipeline(target_points=N, lb=5.0)
    t1 = time.perf_counter()
    out = spec.values
    t2 = time.perf_counter()
    print(f"rep {rep}: {nv} spectra  upload+pipeline {1e3*(t1-t0):.1f} ms, download {1e3*(t2-t1):.1f} ms -> "
          f"{nv/(t2-t0)/1e6:.3f} M spectra/s  ({(x.nbytes+out.nbytes)/(t2-t0)/1e9:.1f} GB/s over the host link), p0={spec.attrs['phase_p0']:.4f}")
# large-result download check: chunked pinned path == plain copy
from xmris_amd import device as dev
y = spec.data
ref = y.cpu().numpy()
t0 = time.perf_counter(); got = dev.to_host(y); t1 = time.perf_counter()
print("to_host equal:", np.array_equal(ref, got), f"{got.nbytes/(t1-t0)/1e9:.1f} GB/s")
