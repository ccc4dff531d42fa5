"""Real-data fixture from the reference's own test data: the raw 1H NSPECT acquisition
`/root/reference/tests/data/nspect_slab_1H/rawdatajob0.nc` (NetCDF-3; variable (raw=10240, component=2) float64 =
interleaved re/im of 5 averages x 2048 points, Bruker order: time fastest) -> `tests/golden/bruker_1h.npz`.

Only DATA is copied (the samples and the four acquisition parameters the chain needs, as recorded next to the file in
`ground_truth.toml`): PVM_SpecMatrix 2048, PVM_NAverages 5, PVM_SpecSWH 5000 Hz, groupDelay 76.125 samples, and the
expected water peak position -2.58 Hz (`[nspect_1h.spectrum_view] water_main`, tolerance +-2.5 Hz from
`docs/notebooks/vendor/testonly_bruker_fid_loader_13C.md:170-178`).  Run in the build container:
    python tests/golden/make_bruker_golden.py
"""
import os

import numpy as np
import scipy.io

SRC = "/root/reference/tests/data/nspect_slab_1H/rawdatajob0.nc"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bruker_1h.npz")


def main():
    with scipy.io.netcdf_file(SRC, "r", mmap=False) as f:
        raw = np.array(f.variables["__xarray_dataarray_variable__"].data, dtype=np.float64)  # (10240, 2)
    z = raw[:, 0] + 1j * raw[:, 1]
    n_points, n_avg = 2048, 5
    assert z.size == n_points * n_avg
    fid = z.reshape(n_avg, n_points)  # [average, time], time fastest (vendor/bruker.py:195-197 reshapes the same way)
    np.savez_compressed(DST, fid=fid, sw_hz=5000.0, group_delay=76.125, water_main_hz=-2.58, tol_hz=2.5)
    print(f"wrote {DST}: fid {fid.shape} {fid.dtype}, |fid|max {np.abs(fid).max():.4g}")


if __name__ == "__main__":
    main()
