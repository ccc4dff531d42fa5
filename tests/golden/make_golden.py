"""Generates the golden vectors under tests/golden/ from the CPU oracle (numpy/scipy restatement).

The reference itself cannot be imported in the build container (ModuleNotFoundError: xarray --
SURVEY.md section 8c), so these vectors pin (a) the oracle against regressions (numpy / scipy upgrades,
edits) and (b) the HIP path on fixed inputs.  Every stage is a closed-form numpy expression that the
reference's notebook assert cells state themselves (see tests/test_oracle_kat.py).
Run:  python tests/golden/make_golden.py      (numpy 2.2.6, scipy 1.15.3)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import xmris_oracle as orc  # noqa: E402


def three_peak(nv, nt, dt, seed):
    t = np.arange(nt) * dt
    base = sum(a * np.exp(-d * t) * np.exp(2j * np.pi * f * t)
               for a, d, f in zip((1.0, 0.5, 0.3), (20.0, 33.0, 25.0), (300.0, -800.0, 1100.0)))
    rng = np.random.default_rng(seed)
    amp = 0.5 + (np.arange(nv) % 997) / 997.0
    amp[nv // 3] = 2.0
    x = amp[:, None] * base[None, :] + 0.02 * (rng.standard_normal((nv, nt)) + 1j * rng.standard_normal((nv, nt))) / np.sqrt(2)
    return x, t


def main():
    # (1) README quick start (C1): 5 x 1024 noise, zero_fill(2048), lb=5, to_spectrum, autophase
    rng = np.random.default_rng(42)
    t = np.linspace(0, 1, 1024)
    x = rng.standard_normal((5, 1024)) + 1j * rng.standard_normal((5, 1024))
    out, info = orc.pipeline_values(x, t, 2048, 5.0, peak_width=100)
    np.savez_compressed(os.path.join(HERE, "c1_quickstart.npz"), x=x, t=t, spectrum=info["spectrum"], freq=info["freq"],
                        flat_idx=info["flat_idx"], pivot=info["pivot"], p0=info["p0"], p1=info["p1"], nfev=info["nfev"],
                        phased=out)
    # (2) 3-peak FIDs 16 x 4096 -> 8192 (C3-shaped): first/last rows + checksums, both precisions
    x, t = three_peak(16, 4096, 1 / 5000.0, 7)
    out, info = orc.pipeline_values(x, t, 8192, 5.0, peak_width=100)
    x32 = x.astype(np.complex64)
    out32, info32 = orc.pipeline_values(x32.astype(np.complex128), t, 8192, 5.0, peak_width=100)
    np.savez_compressed(os.path.join(HERE, "c3_three_peak.npz"), x=x, t=t, rows=np.array([0, 5, 15]),
                        phased_rows=out[[0, 5, 15]], phased32_rows=out32[[0, 5, 15]],
                        checksum=np.array([out.sum(), np.abs(out).sum()]), flat_idx=info["flat_idx"],
                        p0=info["p0"], p1=info["p1"], pivot=info["pivot"], p0_32=info32["p0"], p1_32=info32["p1"])
    # (3) FFT-only vectors: radix-3 length and a prime length
    rng = np.random.default_rng(3)
    a = rng.standard_normal((4, 1536)) + 1j * rng.standard_normal((4, 1536))
    b = rng.standard_normal((3, 1531)) + 1j * rng.standard_normal((3, 1531))
    np.savez_compressed(os.path.join(HERE, "fft_mixed.npz"), a=a, fa=orc.to_spectrum_values(a, 1), b=b,
                        fb=orc.fft_values(b, 1))
    # (4) symmetric zero fill 32 -> 128 and odd/even fftshift
    k = rng.standard_normal((6, 32)) + 1j * rng.standard_normal((6, 32))
    zf, pad_left = orc.zero_fill_values(k, 1, 128, "symmetric")
    s7 = rng.standard_normal((2, 7)) + 0j
    s8 = rng.standard_normal((2, 8)) + 0j
    np.savez_compressed(os.path.join(HERE, "zero_fill_roll.npz"), k=k, zf=zf, pad_left=pad_left, s7=s7, s8=s8,
                        r7=np.roll(s7, 3, axis=1), r8=np.roll(s8, 4, axis=1), ir7=np.roll(s7, 4, axis=1))
    # (5) objective values of the three scores at fixed (p0, p1) on one slice
    sl, fr, pv, ti = info["slice"], info["freq"], info["pivot"], info["target_idx"]
    pts = [(0.0, 0.0), (30.0, -200.0), (-120.0, 1500.0), (179.0, -3999.0), (2.5, 40.0)]
    iw = max(1, int(round(50.0 / abs(fr[1] - fr[0]))))
    np.savez_compressed(os.path.join(HERE, "scores.npz"), slice=sl, freq=fr, pivot=pv, target_idx=ti, index_width=iw,
                        points=np.array(pts),
                        acme=np.array([orc.acme_score(p, sl, fr, pv) for p in pts]),
                        peak_minima=np.array([orc.peak_minima_score(p, sl, fr, pv, ti, iw) for p in pts]),
                        positivity=np.array([orc.roi_positivity_score(p, sl, fr, pv, ti, iw) for p in pts]))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
