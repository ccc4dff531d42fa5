"""Latency of the reference's README quick start (BASELINE configs[0]: 5 x 1024 -> 2048, lb = 5) through the
accessor chain and through the fused call, host ndarray in -> host ndarray out, next to the CPU oracle."""
import sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import xmris_amd as xm
import xmris_oracle as orc
rng = np.random.default_rng(42)
data = rng.standard_normal((5, 1024)) + 1j * rng.standard_normal((5, 1024))
t = np.linspace(0, 1, 1024)
def chain():
    fid = xm.LabeledArray(data, dims=["voxel", "time"], coords={"voxel": np.arange(5), "time": t})
    return fid.xmr.zero_fill(target_points=2048).xmr.apodize_exp(lb=5.0).xmr.to_spectrum().xmr.autophase().values
def fused():
    fid = xm.LabeledArray(data, dims=["voxel", "time"], coords={"voxel": np.arange(5), "time": t})
    return fid.xmr.spectral_pipeline(target_points=2048, lb=5.0).values
def oracle():
    return orc.pipeline_values(data, t, 2048, 5.0, peak_width=100)[0]
for name, f, reps in (("accessor chain (4 calls)", chain, 20), ("fused spectral_pipeline", fused, 20), ("CPU oracle (numpy + scipy DE)", oracle, 3)):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); y = f(); ts.append(time.perf_counter() - t0)
    print(f"{name:32s} median {1e3*np.median(ts):8.2f} ms   min {1e3*min(ts):8.2f} ms")
