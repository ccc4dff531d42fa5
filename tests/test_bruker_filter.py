"""remove_digital_filter (reference vendor/bruker.py:7-118).
CPU: the oracle against the notebook's known-answer cell (vendor/bruker_filter_removal.md:196-239) and
the host layer against the oracle (numpy test double for the kernels).  GPU: the HIP path vs the oracle."""
import numpy as np
import pytest


def _bruker_like(n_points=1000, delay_points=76.125):
    """The notebook's synthetic hardware FID (bruker_filter_removal.md:84-126)."""
    dt = 0.001
    time = np.arange(n_points) * dt
    true_fid = np.exp(-time * 10.0) * np.exp(1j * 2 * np.pi * 50.0 * time)
    int_delay = int(np.floor(delay_points))
    frac = delay_points - int_delay
    fir_length = 2 * int_delay + 1
    n = np.arange(fir_length)
    window = 0.54 - 0.46 * np.cos(2 * np.pi * n / (fir_length - 1))
    sinc_filter = np.sinc(0.5 * (n - int_delay)) * window
    sinc_filter /= np.sum(sinc_filter)
    hardware = np.convolve(true_fid, sinc_filter, mode="full")[:n_points]
    spec = np.fft.fft(hardware) * np.exp(-1j * 2 * np.pi * np.fft.fftfreq(n_points) * frac)
    return np.fft.ifft(spec), time


def _kat(clean, to_spectrum, raw_attrs):
    assert clean.attrs["digital_filter_removed"] is True
    assert clean.attrs["group_delay_removed"] == 76.125
    assert clean.attrs["length_retained_with_zeros"] is True
    assert clean.attrs["description"] == "Raw Bruker Data" and "digital_filter_removed" not in raw_attrs
    assert clean.values.shape[-1] == 1000
    np.testing.assert_allclose(clean.coords["Time"].values[0], 0.0)
    np.testing.assert_allclose(clean.values[..., -76:], 0.0, atol=1e-12)
    first = clean.values[..., 0].reshape(-1)[0]
    assert first.real > 0.5 and abs(first.imag) < 0.2
    spec = to_spectrum(clean)
    row = spec.values.reshape(-1, spec.values.shape[-1])[0]
    peak = row[np.argmax(np.abs(row))]
    assert peak.real > 0 and abs(peak.imag) < peak.real * 0.15


def test_oracle_kat(oracle):
    fid, time = _bruker_like()
    raw = oracle.Labeled(fid, ("Time",), {"Time": oracle.Coord("Time", time)}, {"units": "a.u.", "description": "Raw Bruker Data"})
    clean = oracle.remove_digital_filter(raw, 76.125, dim="Time", keep_length=True)
    _kat(clean, lambda d: oracle.to_spectrum(d, dim="Time", out_dim="Frequency"), raw.attrs)
    short = oracle.remove_digital_filter(raw, 76.125, dim="Time", keep_length=False)
    assert short.values.shape == (924,) and short.coords["Time"].values[0] == 0.0
    np.testing.assert_allclose(short.values, clean.values[:924])
    assert oracle.remove_digital_filter(raw, 0.0, dim="Time").attrs == raw.attrs
    with pytest.raises(ValueError, match="missing in DataArray"):
        oracle.remove_digital_filter(raw, 5.0, dim="time")


def _cases(oracle, xm):
    fid, time = _bruker_like()
    rng = np.random.default_rng(4)
    stack = np.stack([fid * a for a in (1.0, 0.5, 2.0)]) + 0.01 * rng.standard_normal((3, 1000))
    for values, dims, coords in [(fid, ("Time",), {"Time": time}),
                                 (stack, ("avg", "Time"), {"Time": time, "avg": np.arange(3)}),
                                 (stack.T.copy(), ("Time", "avg"), {"Time": time})]:
        attrs = {"units": "a.u.", "description": "Raw Bruker Data"}
        a = xm.LabeledArray(values, dims, coords, attrs)
        o = oracle.Labeled(values, dims, {k: oracle.Coord(k, np.asarray(v)) for k, v in coords.items()}, dict(attrs))
        for gd, keep in [(76.125, True), (76.125, False), (76.0, True), (0.4, True), (0.0, True), (12.5, False)]:
            yield a, o, gd, keep


def _compare(a, o, rtol):
    assert a.dims == o.dims and a.attrs == o.attrs and set(a.coords) == set(o.coords)
    for k in o.coords:
        np.testing.assert_array_equal(a.coords[k].values, o.coords[k].values)
    assert a.values.shape == o.values.shape
    assert np.abs(a.values - o.values).max() <= rtol * max(np.abs(o.values).max(), 1e-300)


def test_host_layer_matches_oracle(oracle, monkeypatch):
    import _numpy_device

    import xmris_amd as xm

    _numpy_device.install(monkeypatch)
    for a, o, gd, keep in _cases(oracle, xm):
        _compare(a.xmr.remove_digital_filter(group_delay=gd, dim="Time", keep_length=keep),
                 oracle.remove_digital_filter(o, gd, dim="Time", keep_length=keep), 1e-12)
    a = xm.LabeledArray(np.zeros(8, complex), ("Time",))
    with pytest.raises(ValueError, match="Dimension 'time' missing in DataArray."):
        a.xmr.remove_digital_filter(group_delay=3.5)
    with pytest.raises(KeyError):
        a.xmr.remove_digital_filter(group_delay=3.5, dim="Time")  # no coordinate on the dim


@pytest.mark.gpu
def test_hip_matches_oracle_and_notebook_kat(oracle):
    import xmris_amd as xm

    for a, o, gd, keep in _cases(oracle, xm):
        _compare(a.xmr.remove_digital_filter(group_delay=gd, dim="Time", keep_length=keep),
                 oracle.remove_digital_filter(o, gd, dim="Time", keep_length=keep), 1e-12)
    fid, time = _bruker_like()
    raw = xm.LabeledArray(fid.astype(np.complex64), ("Time",), {"Time": time},
                          {"units": "a.u.", "description": "Raw Bruker Data"})
    clean = raw.xmr.remove_digital_filter(group_delay=76.125, dim="Time", keep_length=True)
    _kat(clean, lambda d: d.xmr.to_spectrum(dim="Time", out_dim="Frequency"), raw.attrs)
    ref = oracle.remove_digital_filter(oracle.Labeled(fid.astype(np.complex64), ("Time",), {"Time": oracle.Coord("Time", time)}, {}),
                                       76.125, dim="Time")
    assert np.abs(clean.values - ref.values).max() < 1e-5 * np.abs(ref.values).max()
    # the reference's real-data chain: filter removal -> apodize -> spectrum -> autophase (bruker_fid_loader.md:113-123)
    chain = clean.xmr.apodize_exp(dim="Time", lb=5.0).xmr.to_spectrum(dim="Time", out_dim="frequency").xmr.autophase()
    assert "phase_p0" in chain.attrs and chain.attrs["digital_filter_removed"] is True
