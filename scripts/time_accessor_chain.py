"""Wall time of the four-call accessor chain on a device-resident LabeledArray (what a user of the drop-in sees):
da.xmr.zero_fill(...).xmr.apodize_exp(...).xmr.to_spectrum().xmr.autophase(), complex128 and complex64 input."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import xmris_amd as xm
from xmris_amd import device as dev
nv, nt, N = int(os.environ.get("NV", 32768)), 4096, 8192
for cdtype in (torch.complex128, torch.complex64):
    x, t = bench.synth_fids(torch, nv, nt, 1.0 / 5000.0, 0, nv, torch.device("cuda"), cdtype)
    fid = xm.LabeledArray(x, dims=["voxel", "time"], coords={"voxel": np.arange(nv), "time": t})
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        spec = fid.xmr.zero_fill(target_points=N).xmr.apodize_exp(lb=5.0).xmr.to_spectrum().xmr.autophase()
        y = spec.data  # materialise (device)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    print(f"{str(cdtype):18s} {nv} x {nt} -> {N}: {1e3*(t1-t0):.3f} ms per chain ({nv/(t1-t0)/1e6:.2f} M spectra/s), result {y.dtype}, "
          f"p0 {spec.attrs['phase_p0']:.4f} p1 {spec.attrs['phase_p1']:.3f}")
    del x, fid, spec, y
