"""BASELINE.json configs[1], configs[2] (full size) and configs[4] at their literal shapes through the drop-in surface:
  C2  (32, 32, 16, 2048)  3-D MRSI grid, zero-filled to 4096 (and the literal no-op target_points=2048)
  C5  (8, 64, 64, 1536)   multi-coil, non-power-of-two length (2^9 * 3), no zero fill, both storage precisions
Each goes through the chained `.xmr` calls AND `.xmr.spectral_pipeline`; the oracle runs on 64 sampled voxels plus the
designated brightest one (which fixes the global arg-max, hence (p0, p1), for the subset exactly as for the whole array),
and size-independent properties are checked on every voxel: |phased| == |unphased| (unit-modulus phase), Parseval
against the apodised FID, the arg-max voxel, lineage attrs and coordinates.  Synthetic data as SURVEY section 8(d)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _synth(shape_vox, nt, seed, dtype):
    nv = int(np.prod(shape_vox))
    t = np.arange(nt) / 5000.0
    base = sum(a * np.exp(-d * t) * np.exp(2j * np.pi * f * t)
               for a, d, f in zip((1.0, 0.5, 0.3), (20.0, 33.0, 25.0), (300.0, -800.0, 1100.0)))
    rng = np.random.default_rng(seed)
    amp = 0.5 + (np.arange(nv) % 997) / 997.0
    star = nv // 3
    amp[star] = 2.0
    x = amp[:, None] * base[None, :] + 0.02 / np.sqrt(2) * (rng.standard_normal((nv, nt)) + 1j * rng.standard_normal((nv, nt)))
    return x.astype(dtype).reshape(shape_vox + (nt,)), t, star


def _subset(nv, star, seed):
    rng = np.random.default_rng(seed)
    return np.unique(np.concatenate([rng.choice(nv, 64, replace=False), [star, 0, nv - 1]]))


def _check_against_oracle(oracle, got, x, t, dims_vox, target, lb, sub, star, nv, rtol, p_tol):
    """Oracle chain on the sampled voxels (a 2-D [voxel, time] array that contains the brightest voxel)."""
    n_out = max(target, x.shape[-1])
    xs = x.reshape(nv, -1)[sub]
    o = oracle.Labeled(xs, ("voxel", "time"), {"time": oracle.Coord("time", t)}, {})
    oc = oracle.autophase(oracle.to_spectrum(oracle.apodize_exp(oracle.zero_fill(o, target_points=target), lb=lb)), peak_width=100)
    assert got.dims == dims_vox + ("frequency",)
    np.testing.assert_array_equal(got.coords["frequency"].values, oc.coords["frequency"].values)
    assert got.attrs["phase_pivot"] == oc.attrs["phase_pivot"] and got.attrs["phase_pivot_coord"] == "frequency"
    assert abs(got.attrs["phase_p0"] - oc.attrs["phase_p0"]) < p_tol and abs(got.attrs["phase_p1"] - oc.attrs["phase_p1"]) < p_tol
    g = got.values.reshape(nv, n_out)[sub]
    assert np.abs(g - oc.values).max() <= rtol * np.abs(oc.values).max()
    return oc


def _properties(got, x, t, lb, nv, star, n_out, rel):
    """Size-independent checks on EVERY voxel."""
    g = got.values.reshape(nv, n_out)
    flat = int(np.argmax(np.abs(g)))
    assert flat // n_out == star, "the designated brightest voxel holds the global maximum"
    w = np.exp(-np.pi * lb * t)
    e_fid = (np.abs(x.reshape(nv, -1).astype(np.complex128) * w) ** 2).sum(axis=1)
    e_spec = (np.abs(g.astype(np.complex128)) ** 2).sum(axis=1)
    np.testing.assert_allclose(e_spec, e_fid, rtol=rel)  # ortho FFT + unit-modulus phase: Parseval per voxel


@pytest.mark.parametrize("target", [4096, 2048])
def test_c2_mrsi_grid_32x32x16x2048(oracle, target):
    import xmris_amd as xm

    shape_vox, nt, lb = (32, 32, 16), 2048, 5.0
    nv = int(np.prod(shape_vox))
    x, t, star = _synth(shape_vox, nt, 21, np.complex64)
    dims = ("x", "y", "z", "time")
    a = xm.LabeledArray(x, dims, {"time": t, "x": np.arange(32), "z": np.arange(16) * 2.0}, {"B0": 3.0})
    sub = _subset(nv, star, 5)
    n_out = max(target, nt)
    chain = a.xmr.zero_fill(target_points=target).xmr.apodize_exp(lb=lb).xmr.to_spectrum().xmr.autophase()
    fused = a.xmr.spectral_pipeline(target_points=target, lb=lb)
    for got, rtol, p_tol in ((chain, 1e-9, 1e-6), (fused, 1e-5, 1e-6)):
        oc = _check_against_oracle(oracle, got, x, t, dims[:-1], target, lb, sub, star, nv, rtol, p_tol)
        _properties(got, x, t, lb, nv, star, n_out, 2e-5)
        assert got.attrs["B0"] == 3.0 and got.attrs["apodization_lb"] == lb
        assert ("zero_fill_target" in got.attrs) == (target > nt)  # fid.py:235-236: the no-op fill stamps nothing
        np.testing.assert_array_equal(got.coords["z"].values, np.arange(16) * 2.0)
    assert chain.values.dtype == np.complex128 and fused.values.dtype == np.complex64
    assert chain.attrs["phase_p0"] == fused.attrs["phase_p0"] or abs(chain.attrs["phase_p0"] - fused.attrs["phase_p0"]) < 1e-6


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_c5_multicoil_8x64x64x1536(oracle, dtype):
    import xmris_amd as xm

    shape_vox, nt, lb = (8, 64, 64), 1536, 5.0
    nv = int(np.prod(shape_vox))
    x, t, star = _synth(shape_vox, nt, 22, dtype)
    dims = ("coil", "x", "y", "time")
    a = xm.LabeledArray(x, dims, {"time": t, "coil": np.arange(8)}, {"nucleus": "1H"})
    sub = _subset(nv, star, 6)
    chain = a.xmr.zero_fill(target_points=1536).xmr.apodize_exp(lb=lb).xmr.to_spectrum().xmr.autophase()
    fused = a.xmr.spectral_pipeline(target_points=1536, lb=lb)
    tol_fused = 1e-5 if dtype == np.complex64 else 1e-9
    for got, rtol in ((chain, 1e-9), (fused, tol_fused)):
        _check_against_oracle(oracle, got, x, t, dims[:-1], 1536, lb, sub, star, nv, rtol, 1e-6)
        _properties(got, x, t, lb, nv, star, 1536, 2e-5)
        assert "zero_fill_target" not in got.attrs and got.attrs["nucleus"] == "1H"
    assert fused.values.dtype == dtype


def test_c2_shape_with_the_fid_axis_first(oracle):
    """The same grid stored time-first (2048, 32, 32, 16), as Bruker files are: `dim` is not the last axis, every call
    moves it (processing/fourier.py:152 `get_axis_num`) and the global arg-max follows the ORIGINAL C order."""
    import xmris_amd as xm

    shape_vox, nt, lb, target = (32, 32, 16), 2048, 5.0, 4096
    nv = int(np.prod(shape_vox))
    x, t, star = _synth(shape_vox, nt, 23, np.complex64)
    xt = np.ascontiguousarray(np.moveaxis(x, -1, 0))
    a = xm.LabeledArray(xt, ("time", "x", "y", "z"), {"time": t}, {})
    got = a.xmr.zero_fill(target_points=target).xmr.apodize_exp(lb=lb).xmr.to_spectrum().xmr.autophase()
    assert got.dims == ("frequency", "x", "y", "z") and got.shape == (target, 32, 32, 16)
    sub = _subset(nv, star, 7)
    xs = x.reshape(nv, nt)[sub]
    o = oracle.Labeled(np.ascontiguousarray(xs.T), ("time", "voxel"), {"time": oracle.Coord("time", t)}, {})
    oc = oracle.autophase(oracle.to_spectrum(oracle.apodize_exp(oracle.zero_fill(o, target_points=target), lb=lb)), peak_width=100)
    assert got.attrs["phase_pivot"] == oc.attrs["phase_pivot"]
    assert abs(got.attrs["phase_p0"] - oc.attrs["phase_p0"]) < 1e-6 and abs(got.attrs["phase_p1"] - oc.attrs["phase_p1"]) < 1e-6
    g = got.values.reshape(target, nv)[:, sub]
    assert np.abs(g - oc.values).max() <= 1e-9 * np.abs(oc.values).max()
    fused = a.xmr.spectral_pipeline(target_points=target, lb=lb)  # falls back to the staged calls for this layout
    assert fused.dims == got.dims and np.abs(fused.values - got.values).max() <= 1e-9 * np.abs(got.values).max()


@pytest.mark.parametrize("dtype,nv", [("complex64", 65536), ("complex128", 65536)])
def test_c3_roofline_config_65536x4096_full_size(oracle, dtype, nv):
    """BASELINE configs[2] at its full size (65,536 voxels x 4096 -> 8192, complex64, the bench's synthetic data
    generated on the device; and in complex128 -- the reference's arithmetic, the bench's `c128` record -- at the same 65,536 voxels): both schedules of the streaming executor and the accessor's fused pipeline.  The oracle
    runs on the first 64, the last 64 and the voxels around the designated brightest one (which fixes the global
    arg-max, hence (p0, p1), for the subset as for the whole array); every voxel is checked through size-independent
    properties on the device: Parseval against the apodised FID, the arg-max voxel, speculative == classic."""
    import os
    import sys

    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import xmris_amd as xm
    from xmris_amd import pipeline

    nt, n_out, lb = 4096, 8192, 5.0
    cdt = getattr(torch, dtype)
    tol = 1e-5 if dtype == "complex64" else 1e-10
    x, t = bench.synth_fids(torch, nv, nt, 1.0 / 5000.0, 0, nv, torch.device("cuda"), cdt)
    star = nv // 3
    plan = pipeline.make_plan(x, t, n_out, lb)
    outs = [torch.empty((nv, n_out), dtype=cdt, device="cuda") for _ in range(2)]
    classic = pipeline.run_stream([x], [outs[0]], plan)[0]
    spec = pipeline.run_stream([x], [outs[1]], plan, speculate=True)[0]
    torch.cuda.synchronize()
    assert spec.speculation == "hit" and classic.flat_index // n_out == star == spec.flat_index // n_out
    assert (classic.p0, classic.p1, classic.pivot, classic.flat_index) == (spec.p0, spec.p1, spec.pivot, spec.flat_index)
    scale = float(outs[0].abs().max())
    assert float((outs[0] - outs[1]).abs().max()) <= (2.5e-7 if dtype == "complex64" else 1e-14) * scale  # contraction only
    # every voxel: Parseval (ortho FFT, unit-modulus phase) and the global maximum
    w = torch.from_numpy(np.exp(-np.pi * lb * t)).to("cuda", torch.float64)
    e_fid = ((x.to(torch.complex128) * w).abs() ** 2).sum(dim=1)
    for o in outs:
        e_spec = (o.to(torch.complex128).abs() ** 2).sum(dim=1)
        assert float(((e_spec - e_fid).abs() / e_fid).max()) < (2e-5 if dtype == "complex64" else 1e-12)
        assert int(torch.argmax(o.abs().reshape(-1))) // n_out == star
    # the oracle on a subset that contains the brightest voxel
    sub = np.unique(np.concatenate([np.arange(64), np.arange(nv - 64, nv), np.arange(star - 2, star + 3)]))
    xs = x[torch.from_numpy(sub).cuda()].cpu().numpy()
    ref, info = oracle.pipeline_values(xs.astype(np.complex128), t, n_out, lb, peak_width=100)
    assert info["pivot"] == classic.pivot and abs(info["p0"] - classic.p0) < 1e-6 and abs(info["p1"] - classic.p1) < 1e-6
    got = outs[1][torch.from_numpy(sub).cuda()].cpu().numpy()
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max()
    # the accessor's fused pipeline on the device-resident array
    del outs
    a = xm.LabeledArray(x, ("voxel", "time"), {"time": t}, {})
    fused = a.xmr.spectral_pipeline(target_points=n_out, lb=lb)
    assert abs(fused.attrs["phase_p0"] - info["p0"]) < 1e-6 and abs(fused.attrs["phase_p1"] - info["p1"]) < 1e-6
    g = fused.data[torch.from_numpy(sub).cuda()].cpu().numpy()
    assert np.abs(g - ref).max() <= tol * np.abs(ref).max()
