"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the CPU oracle).
CPU part: the oracle and the product's host-side solver still reproduce them.  GPU part (-m gpu): the
HIP path reproduces them through the C ABI."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def _relerr(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


# ------------------------------------------------------------------ CPU
def test_oracle_reproduces_c1(oracle):
    g = _load("c1_quickstart.npz")
    out, info = oracle.pipeline_values(g["x"], g["t"], 2048, 5.0, peak_width=100)
    assert info["flat_idx"] == int(g["flat_idx"]) and info["pivot"] == float(g["pivot"])
    assert _relerr(info["spectrum"], g["spectrum"]) < 1e-13
    assert abs(info["p0"] - float(g["p0"])) < 1e-3 and abs(info["p1"] - float(g["p1"])) < 1e-3  # noise: flat optimum
    assert _relerr(out, g["phased"]) < 1e-4


def test_host_solver_matches_golden_scores_and_solution():
    from xmris_amd import autophase_solver as aps

    g = _load("scores.npz")
    sl, fr, pv, ti, iw = g["slice"], g["freq"], float(g["pivot"]), int(g["target_idx"]), int(g["index_width"])
    for i, p in enumerate(g["points"]):
        assert aps.acme_score(p, sl, fr, pv) == pytest.approx(g["acme"][i], rel=1e-13)
        assert aps.peak_minima_score(p, sl, fr, pv, ti, iw) == pytest.approx(g["peak_minima"][i], rel=1e-12, abs=1e-15)
        assert aps.roi_positivity_score(p, sl, fr, pv, ti, iw) == pytest.approx(g["positivity"][i], rel=1e-12)
    c3 = _load("c3_three_peak.npz")
    p0, p1, opt = aps.solve(sl, fr, pv, ti, iw)
    assert abs(p0 - float(c3["p0"])) < 1e-6 and abs(p1 - float(c3["p1"])) < 1e-6
    tab = aps.phase_table(fr, 33.0, -250.0, 12.5)
    x = (fr - 12.5) / (fr.max() - fr.min())
    np.testing.assert_allclose(tab, np.exp(1j * (np.radians(33.0) + np.radians(-250.0) * x)), rtol=1e-15)
    assert np.all(aps.phase_table(np.array([5.0]), 10.0, 99.0, 5.0) == np.exp(1j * np.radians(10.0)))  # range 0


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [("complex128", 1e-12), ("complex64", 1e-5)])
def test_hip_reproduces_c1_and_c3(dtype, tol):
    from xmris_amd import device as dev
    from xmris_amd import pipeline as pipe

    g = _load("c1_quickstart.npz")
    xd = dev.to_device(g["x"].astype(dtype))
    out, res, plan = pipe.run(xd, g["t"], 2048, 5.0, params=(float(g["p0"]), float(g["p1"])))
    assert res.flat_index == int(g["flat_idx"]) and res.pivot == float(g["pivot"])
    assert _relerr(out.cpu().numpy(), g["phased"]) < tol
    c3 = _load("c3_three_peak.npz")
    xd = dev.to_device(c3["x"].astype(dtype))
    out, res, plan = pipe.run(xd, c3["t"], 8192, 5.0)  # own solve
    key = ("p0", "p1", "phased_rows") if dtype == "complex128" else ("p0_32", "p1_32", "phased32_rows")
    assert res.flat_index == int(c3["flat_idx"]) and res.pivot == float(c3["pivot"])
    assert abs(res.p0 - float(c3[key[0]])) < 1e-6 and abs(res.p1 - float(c3[key[1]])) < 1e-6
    assert _relerr(out[c3["rows"].tolist()].cpu().numpy(), c3[key[2]]) < tol
    if dtype == "complex128":
        o = out.cpu().numpy()
        assert abs(o.sum() - c3["checksum"][0]) < 1e-9 * abs(c3["checksum"][1])
        assert abs(np.abs(o).sum() - c3["checksum"][1].real) < 1e-9 * abs(c3["checksum"][1])


@pytest.mark.gpu
def test_hip_reproduces_mixed_radix_prime_and_index_vectors():
    from xmris_amd import device as dev

    g = _load("fft_mixed.npz")
    assert _relerr(dev.fft(dev.to_device(g["a"]), 1, shift_out=True).cpu().numpy(), g["fa"]) < 1e-13
    assert _relerr(dev.fft(dev.to_device(g["b"]), 1).cpu().numpy(), g["fb"]) < 4e-13
    assert _relerr(dev.fft(dev.to_device(g["a"].astype(np.complex64)), 1, shift_out=True).cpu().numpy(), g["fa"]) < 2e-6
    z = _load("zero_fill_roll.npz")
    np.testing.assert_array_equal(dev.zero_fill(dev.to_device(z["k"]), 1, 128, int(z["pad_left"])).cpu().numpy(), z["zf"])
    np.testing.assert_array_equal(dev.roll(dev.to_device(z["s7"]), 1, 7 // 2).cpu().numpy(), z["r7"])
    np.testing.assert_array_equal(dev.roll(dev.to_device(z["s8"]), 1, 8 // 2).cpu().numpy(), z["r8"])
    np.testing.assert_array_equal(dev.roll(dev.to_device(z["s7"]), 1, (7 + 1) // 2).cpu().numpy(), z["ir7"])
