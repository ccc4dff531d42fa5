"""baseline_als (reference processing/baseline.py).  CPU: the oracle against the
notebook's known-answer cell (pipeline/baseline.md:143-167).  GPU: the fp64 band-LDL' kernel vs the oracle's
scipy.sparse spsolve (tolerance 1e-6 of the spectrum scale: the system's condition number is ~1e9)."""
import numpy as np
import pytest


def _spectrum(n=1024, sw=2000.0, seed=1, nv=1):
    """Two sharp lines on a rolling baseline of three very broad lines (baseline.md:60-95, own generator)."""
    t = np.arange(n) / sw
    rng = np.random.default_rng(seed)
    rows = []
    for v in range(nv):
        fid = sum(a * np.exp(-d * t) * np.exp(2j * np.pi * f * t) for a, d, f in
                  [(10.0, 30.0, 616.0), (5.0, 40.0, -308.0), (35.0, 1200.0, 369.6), (45.0, 1800.0, 61.6),
                   (30.0, 1500.0, -184.8)])
        fid = fid * (1 + 0.1 * v) + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        fid[0] *= 0.5
        rows.append(np.roll(np.fft.fft(fid, norm="ortho"), n // 2))
    freq = np.roll(np.fft.fftfreq(n, d=1 / sw), n // 2)
    return np.stack(rows) if nv > 1 else rows[0], freq


def test_oracle_kat(oracle):
    spec, freq = _spectrum()
    da = oracle.Labeled(spec, ("frequency",), {"frequency": oracle.Coord("frequency", freq)}, {"reference_frequency": 123.2})
    cor = oracle.baseline_als(da, lam=1e5, p=0.01)
    assert np.iscomplexobj(da.values) and not np.iscomplexobj(cor.values)
    assert cor.attrs["reference_frequency"] == 123.2 and cor.attrs["baseline_method"] == "als"
    assert cor.attrs["baseline_lam"] == 1e5 and cor.attrs["baseline_p"] == 0.01 and cor.attrs["baseline_iter"] == 10
    i = int(np.argmin(np.abs(freq - 123.2)))  # metabolite-free region: pure baseline
    assert spec.real[i] > 0.5 and abs(cor.values[i]) < 0.2 * abs(spec.real[i])
    with pytest.raises(ValueError, match="baseline_als"):
        oracle.baseline_als(da, dim="time")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["complex128", "complex64", "float64"])
def test_hip_matches_oracle(oracle, dtype):
    import xmris_amd as xm

    spec, freq = _spectrum(nv=5)
    vals = spec.real.astype(dtype) if dtype == "float64" else spec.astype(dtype)
    for lam, p, n_iter in [(1e5, 0.01, 10), (1e5, 0.001, 10), (1e7, 0.001, 3)]:
        a = xm.LabeledArray(vals, ("voxel", "frequency"), {"frequency": freq}, {"reference_frequency": 123.2}, name="s")
        o = oracle.Labeled(vals, ("voxel", "frequency"), {"frequency": oracle.Coord("frequency", freq)},
                           {"reference_frequency": 123.2}, "s")
        r, ro = a.xmr.baseline_als(lam=lam, p=p, n_iter=n_iter), oracle.baseline_als(o, lam=lam, p=p, n_iter=n_iter)
        assert r.dims == ro.dims and r.attrs == ro.attrs and r.name == ro.name
        assert r.values.dtype == np.float64 and not np.iscomplexobj(r.values)
        np.testing.assert_array_equal(r.coords["frequency"].values, freq)
        assert np.abs(r.values - ro.values).max() < 1e-6 * np.abs(np.real(vals)).max(), (lam, p)
    # other axis + 8192 points + the notebook's known-answer region
    big, f8 = _spectrum(n=8192, sw=5000.0, seed=3)
    a = xm.LabeledArray(np.stack([big, 2 * big]).T.copy(), ("frequency", "rep"), {"frequency": f8})
    o = oracle.Labeled(np.stack([big, 2 * big]).T.copy(), ("frequency", "rep"), {"frequency": oracle.Coord("frequency", f8)})
    r, ro = a.xmr.baseline_als(dim="frequency", lam=1e6, p=0.01), oracle.baseline_als(o, dim="frequency", lam=1e6, p=0.01)
    assert np.abs(r.values - ro.values).max() < 1e-6 * np.abs(big.real).max()
    with pytest.raises(ValueError, match="baseline_als"):
        a.xmr.baseline_als(dim="time")
    # chain: the reference's real-data pipeline ends ... -> autophase -> baseline
    chain = xm.LabeledArray(spec, ("voxel", "frequency"), {"frequency": freq}).xmr.autophase().xmr.baseline_als(p=0.01)
    assert chain.attrs["baseline_method"] == "als" and "phase_p0" in chain.attrs
