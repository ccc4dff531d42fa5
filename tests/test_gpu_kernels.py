"""GPU parity tests: every entry point of libxmris_hip.so (through the C ABI) against the CPU
oracle on the same seeded inputs.  Integer/index work is bit-exact; floating point is compared
with the tolerance the north-star states (1e-5 relative to the spectrum's largest magnitude for
complex64, 1e-12 for complex128)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = {"complex64": 1e-5, "complex128": 1e-12}
# observed fp32 error is ~2e-7; keep a tighter regression guard as well
TIGHT = {"complex64": 2e-6, "complex128": 1e-13}


@pytest.fixture(scope="module")
def dev():
    import torch

    from xmris_amd import device

    assert torch.cuda.is_available(), "GPU tests need a HIP device (no CPU fallback exists)"
    return device


def _rand(shape, dtype, seed=0):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)


def _relerr(got, ref):
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300))


POW2 = [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192]
MIXED = [384, 768, 1536, 3072, 6144, 640, 1280, 2560, 5120]
BLUESTEIN = [3, 5, 7, 17, 100, 1000, 1531, 2000, 2049, 4095, 6000, 8191]  # the last two: chirp-z with M = 16384


# beyond the in-LDS plans: four-step over global memory (n = n1 n2), and chirp-z on top of it for everything else
LONG = [16384, 32768, 65536, 12288, 10240, 24576, 10000, 9001, 20011]


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
@pytest.mark.parametrize("n", POW2 + MIXED + BLUESTEIN + LONG)
def test_fft_matches_numpy(dev, oracle, n, dtype):
    assert dev.fft_supported(n, complex128=dtype == "complex128")
    nb = (3 if n > 8192 else 7) if n > 64 else 37  # ragged vs. spectra-per-workgroup
    x = _rand((nb, n), dtype, seed=n)
    ref = oracle.fft_values(x.astype(np.complex128), 1)
    got = dev.fft(dev.to_device(x), 1).cpu().numpy()
    assert got.dtype == np.dtype(dtype)
    tol = TIGHT[dtype] * (4 if (n in BLUESTEIN or n in LONG) else 1)
    assert _relerr(got, ref) < tol


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
@pytest.mark.parametrize("n", [7, 8, 256, 1000, 1536, 3072, 4096, 5120])
def test_fft_flags(dev, oracle, n, dtype):
    x = _rand((5, n), dtype, seed=3)
    xd = dev.to_device(x)
    x128 = x.astype(np.complex128)
    tol = TIGHT[dtype] * 4
    # to_spectrum = fftshift(fft)
    ref = oracle.to_spectrum_values(x128, 1)
    assert _relerr(dev.fft(xd, 1, shift_out=True).cpu().numpy(), ref) < tol
    # to_fid = ifft(ifftshift)   (fid.py:73-77)
    ref = np.fft.ifftn(np.roll(x128, (n + 1) // 2, axis=1), axes=(1,), norm="ortho")
    assert _relerr(dev.fft(xd, 1, inverse=True, shift_in=True).cpu().numpy(), ref) < tol
    # fftc / ifftc  (fourier.py:258-264, 292-298)
    ref = np.roll(np.fft.fftn(np.roll(x128, (n + 1) // 2, axis=1), axes=(1,), norm="ortho"), n // 2, axis=1)
    assert _relerr(dev.fft(xd, 1, shift_in=True, shift_out=True).cpu().numpy(), ref) < tol
    ref = np.roll(np.fft.ifftn(np.roll(x128, (n + 1) // 2, axis=1), axes=(1,), norm="ortho"), n // 2, axis=1)
    assert _relerr(dev.fft(xd, 1, inverse=True, shift_in=True, shift_out=True).cpu().numpy(), ref) < tol
    # un-normalised forward / 1/N inverse
    assert _relerr(dev.fft(xd, 1, ortho=False).cpu().numpy(), np.fft.fft(x128, axis=1)) < tol
    assert _relerr(dev.fft(xd, 1, inverse=True, ortho=False).cpu().numpy(), np.fft.ifft(x128, axis=1)) < tol


def test_fft_axis_and_roundtrip(dev, oracle):
    x = _rand((6, 256, 5), "complex128", seed=11)
    xd = dev.to_device(x)
    got = dev.fft(xd, 1, shift_out=True)
    assert _relerr(got.cpu().numpy(), oracle.to_spectrum_values(x, 1)) < 1e-13
    back = dev.fft(got, 1, inverse=True, shift_in=True)  # to_fid(to_spectrum(x)) == x
    np.testing.assert_allclose(back.cpu().numpy(), x, atol=1e-10)  # fid_transformations.md:151-157
    one = dev.fft(dev.to_device(_rand((4, 1), "complex64")), 1)
    assert one.shape == (4, 1)


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
def test_zero_fill_bit_exact(dev, oracle, dtype):
    x = _rand((9, 100), dtype, seed=5)
    xd = dev.to_device(x)
    for target, position in [(512, "end"), (257, "end"), (128, "symmetric"), (131, "symmetric")]:
        ref, pad_left = oracle.zero_fill_values(x, 1, target, position)
        got = dev.zero_fill(xd, 1, target, pad_left).cpu().numpy()
        np.testing.assert_array_equal(got, ref)
    # another axis (k-space style, zero_fill.md:257-295)
    y = _rand((32, 6), dtype, seed=6)
    ref, pad_left = oracle.zero_fill_values(y, 0, 128, "symmetric")
    np.testing.assert_array_equal(dev.zero_fill(dev.to_device(y), 0, 128, pad_left).cpu().numpy(), ref)


def test_apodize_and_phase_complex128_bit_exact(dev, oracle):
    n = 777
    x = _rand((4, n), "complex128", seed=8)
    t = np.arange(n) * 2e-4 + 1e-3
    w = oracle.exp_window(t, 5.0)
    got = dev.apodize(dev.to_device(x), 1, w).cpu().numpy()
    np.testing.assert_array_equal(got, x * w)  # complex x real: same two products as numpy
    coords = np.linspace(-2500, 2500, n)
    ph = np.exp(1j * oracle.phase_array(coords, 33.0, -250.0, 12.5))
    got = dev.phase_apply(dev.to_device(x), 1, ph).cpu().numpy()
    assert _relerr(got, oracle.phase_values(x, coords, 1, 33.0, -250.0, 12.5)) < 1e-15


def test_apodize_phase_complex64(dev, oracle):
    n = 2048
    x = _rand((3, n), "complex64", seed=9)
    t = np.arange(n) * 2e-4
    w = oracle.exp_window(t, 5.0)
    got = dev.apodize(dev.to_device(x), 1, w).cpu().numpy()
    assert _relerr(got, x.astype(np.complex128) * w) < 2e-7
    coords = np.linspace(-2500, 2500, n)
    ref = oracle.phase_values(x.astype(np.complex128), coords, 1, 33.0, -250.0, 12.5)
    ph = np.exp(1j * oracle.phase_array(coords, 33.0, -250.0, 12.5))
    assert _relerr(dev.phase_apply(dev.to_device(x), 1, ph).cpu().numpy(), ref) < 3e-7


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
def test_roll_bit_exact(dev, dtype):
    for n in (7, 8, 1000):
        x = _rand((5, n), dtype, seed=n)
        xd = dev.to_device(x)
        np.testing.assert_array_equal(dev.roll(xd, 1, n // 2).cpu().numpy(), np.roll(x, n // 2, axis=1))
        np.testing.assert_array_equal(dev.roll(xd, 1, (n + 1) // 2).cpu().numpy(), np.roll(x, (n + 1) // 2, axis=1))


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
def test_absmax_argmax_first_occurrence(dev, oracle, dtype):
    x = _rand((33, 300), dtype, seed=21)
    flat_ref, _ = oracle.global_argmax(x)
    amax, flat = dev.absmax_argmax(dev.to_device(x))
    assert flat == flat_ref
    assert abs(amax - np.abs(x).max()) < 1e-5 * np.abs(x).max()
    # exact ties: the FIRST maximum in C order wins (np.argmax semantics, phasing.py:229)
    y = np.zeros((16, 64), dtype)
    for pos in [(9, 3), (4, 60), (4, 17), (12, 0)]:
        y[pos] = 3.0 + 4.0j
    y[4, 17] = -5.0  # same magnitude, different phase
    _, flat = dev.absmax_argmax(dev.to_device(y))
    assert flat == int(np.argmax(np.abs(y))) == 4 * 64 + 17
    # 3-D array, arg-max is independent of which axis is the FID axis
    z = _rand((4, 5, 6), dtype, seed=2)
    assert dev.absmax_argmax(dev.to_device(z))[1] == int(np.argmax(np.abs(z)))


CASES = [
    # (n_batch, n_in, n_out, pad_left)  -- README C1, C2-shaped, >=2x, <2x, symmetric, mixed, prime
    (5, 1024, 2048, 0),
    (16, 2048, 4096, 0),
    (3, 4096, 8192, 0),
    (6, 1000, 4096, 0),
    (6, 3000, 4096, 0),
    (5, 700, 4096, 300),   # >=2x zero fill with a left pad: persistent kernel, clamped loads
    (2, 4096, 16384, 0),   # half length 8192 (1024-thread workgroups), complex64 only
    (4, 32, 128, 48),
    (7, 1536, 1536, 0),
    (5, 1200, 1536, 0),
    (3, 1531, 1531, 0),
    (3, 700, 1531, 415),
    (9, 100, 128, 0),
    (6, 2500, 3072, 0),    # 3*2^k: 12 points per thread (k_fft2 pairs / k_fft1)
    (4, 5120, 5120, 0),    # 5*2^k: 20 points per thread
    (3, 3000, 6144, 0),
    (5, 1972, 1972, 0),    # a Bruker FID after the group-delay cut: chirp-z with M = 4096 (k_blue)
    (3, 900, 1000, 50),    # chirp-z with M = 2048, left pad
    (2, 3000, 3001, 0),    # chirp-z with M = 8192 (one-spectrum kernel)
    (2, 8192, 16384, 0),   # 2x zero fill of an 8k FID: in-LDS for complex64, four-step (128 x 128) for complex128
    (3, 12000, 32768, 100),  # four-step 128 x 256, left pad
    (2, 9000, 20000, 0),   # no two-factor split: chirp-z on top of the four-step (M = 65536)
    (2, 12288, 12288, 0),  # 3 * 2^12 = 8 x 1536
    (3, 12000, 16384, 50),  # 16384 without the >= 2x zero fill: one in-LDS transform (complex128: plane-by-plane exchange)
    (3, 5000, 6000, 100),   # chirp-z with M = 16384
]


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
@pytest.mark.parametrize("nb,n_in,n_out,pad_left", CASES)
def test_fused_pipeline_matches_staged_oracle(dev, oracle, nb, n_in, n_out, pad_left, dtype):
    import torch


    x = _rand((nb, n_in), dtype, seed=n_in + n_out)
    x128 = x.astype(np.complex128)
    pads = [(0, 0), (pad_left, n_out - n_in - pad_left)]
    t = (np.arange(n_out) - pad_left) * 2e-4
    w = oracle.exp_window(t, 5.0)
    spec = oracle.to_spectrum_values(np.pad(x128, pads) * w, 1)
    freq = np.roll(np.fft.fftfreq(n_out, d=2e-4), n_out // 2)
    ph = np.exp(1j * oracle.phase_array(freq, 41.0, -333.0, float(freq[n_out // 3])))
    ref = spec * ph
    xd = dev.to_device(x)
    rd = torch.float32 if dtype == "complex64" else torch.float64
    wd = torch.from_numpy(w).to("cuda", rd)
    phd = torch.from_numpy(ph).to("cuda", xd.dtype)
    tol = TIGHT[dtype] * (4 if n_out in (1531, 1972, 1000, 3001, 6000, 16384, 32768, 20000, 12288) else 1)
    # (1) arg-max pre-pass only
    pre = dev.pipeline_fused(xd, n_out, pad_left, window=wd, want_out=False, want_argmax=True)
    amax, flat = dev.argmax_reduce(pre.absmax2, pre.argidx, n_out)
    assert flat == int(np.argmax(np.abs(spec)))
    assert abs(amax - np.abs(spec).max()) < 1e-5 * np.abs(spec).max()
    np.testing.assert_array_equal(pre.argidx.cpu().numpy(), np.argmax(np.abs(spec), axis=1))
    # (1b) value-only pre-pass (the bench's mode): same per-spectrum maxima, bit for bit -- except on the complex128
    # shapes of k_zf2d (-> 8192, -> 16384), which takes the value-only modes only: another kernel than the index modes'
    vo = dev.pipeline_fused(xd, n_out, pad_left, window=wd, want_out=False, want_argmax=True, argmax_value_only=True)
    if dtype == "complex128" and n_out in (8192, 16384) and pad_left == 0 and 2 * n_in <= n_out:
        assert torch.allclose(vo.absmax2, pre.absmax2, rtol=1e-13, atol=0.0)
    else:
        assert torch.equal(vo.absmax2, pre.absmax2)
    # (2) unphased spectrum + arg-max in one launch
    both = dev.pipeline_fused(xd, n_out, pad_left, window=wd, want_argmax=True)
    assert _relerr(both.out.cpu().numpy(), spec) < tol
    np.testing.assert_array_equal(both.argidx.cpu().numpy(), pre.argidx.cpu().numpy())
    # (3) main pass: phased spectrum
    main = dev.pipeline_fused(xd, n_out, pad_left, window=wd, phase_table=phd)
    assert _relerr(main.out.cpu().numpy(), ref) < tol
    assert _relerr(main.out.cpu().numpy(), ref) < TOL[dtype]
    # (4) the fused result equals the staged device ops composed (zero_fill -> apodize -> fft+shift -> phase)
    staged = dev.phase_apply(dev.fft(dev.apodize(dev.zero_fill(xd, 1, n_out, pad_left), 1, w), 1, shift_out=True),
                             1, ph)
    assert _relerr(main.out.cpu().numpy(), staged.cpu().numpy().astype(np.complex128)) < 4 * tol


def test_randomised_parity_sweep(dev):
    """Random (n_in, n_out, pad_left, batch) over every transform family -- in-LDS powers of two up to 16384 (complex128
    16384: plane-by-plane exchange), 3*2^k / 5*2^k in both stage orders, chirp-z incl. the convolution lengths 3072 and
    16384, four-step -- window + phase table + per-row arg-max in one launch against numpy in fp64
    (`scripts/fuzz_parity.py` runs the same sweep with more cases and other seeds)."""
    import torch

    rng = np.random.default_rng(11)
    direct = [512, 768, 1024, 1280, 1536, 2048, 2560, 3072, 4096, 5120, 6144, 8192, 16384, 384, 640, 256, 64]
    for case in range(36):
        fam = ["direct", "direct", "chirp", "chirp3072", "chirp16k", "long"][case % 6]
        if fam == "direct":
            n_out = int(rng.choice(direct))
        elif fam == "chirp":
            n_out = int(rng.integers(3, 4096)) | 1
        elif fam == "chirp3072":
            n_out = int(rng.integers(1025, 1536))
        elif fam == "chirp16k":
            n_out = int(rng.integers(4097, 8192)) | 1
        else:
            n_out = int(rng.choice([12288, 10240, 24576, 20000]))
        n_in = int(rng.integers(2, n_out + 1)) if rng.random() < 0.7 else n_out
        pad = int(rng.integers(0, n_out - n_in + 1)) if rng.random() < 0.4 else 0
        nb = int(rng.integers(1, 40)) if n_out <= 8192 else int(rng.integers(1, 6))
        t = (np.arange(n_out) - pad) * 2e-4
        w = np.exp(-np.pi * 5.0 * np.abs(t))
        ph = np.exp(1j * (0.3 + 1e-3 * np.arange(n_out)))
        for dtype in ("complex64", "complex128"):
            x = (rng.standard_normal((nb, n_in)) + 1j * rng.standard_normal((nb, n_in))).astype(dtype)
            xp = np.zeros((nb, n_out), dtype=np.complex128)
            xp[:, pad:pad + n_in] = x
            spec = np.fft.fftshift(np.fft.fft(xp * w, axis=1, norm="ortho"), axes=1)
            xd = dev.to_device(x)
            rd = torch.float32 if dtype == "complex64" else torch.float64
            got = dev.pipeline_fused(xd, n_out, pad, window=torch.from_numpy(w).to("cuda", rd),
                                     phase_table=torch.from_numpy(ph).to("cuda", xd.dtype), want_argmax=True)
            tag = (fam, dtype, nb, n_in, n_out, pad)
            assert _relerr(got.out.cpu().numpy().astype(np.complex128), spec * ph) < (3e-6 if dtype == "complex64" else 1e-13), tag
            if dtype == "complex128":
                np.testing.assert_array_equal(got.argidx.cpu().numpy(), np.argmax(np.abs(spec), axis=1), err_msg=str(tag))


def test_unsupported_length_raises(dev):
    from xmris_amd import _lib

    assert not dev.fft_supported((1 << 22) + 1) and dev.fft_supported(1 << 22)
    x = dev.to_device(_rand((1, (1 << 22) + 1), "complex64"))
    with pytest.raises(_lib.UnsupportedLengthError):
        dev.fft(x, 1)
    with pytest.raises(RuntimeError):
        dev.fft(x.cpu(), 1)  # no CPU path


@pytest.mark.parametrize("dtype", ["complex64", "float64"])
def test_to_host_staged_download_is_exact(dev, dtype):
    """Large results leave HBM through pinned staging buffers + worker-thread memcpys (device.to_host): the bytes
    must be the ones a plain copy delivers, including a ragged last chunk and a non-contiguous source."""
    import torch

    n = (5 * (32 << 20) + 12345) // 8  # five and a bit staging chunks of complex64
    x = torch.randn(n, 2, device="cuda")
    x = torch.view_as_complex(x) if dtype == "complex64" else x.double().reshape(-1)
    assert np.array_equal(dev.to_host(x), x.cpu().numpy())
    y = x[: (x.numel() // 6) * 6].reshape(-1, 6)[:, ::2]  # strided view
    assert np.array_equal(dev.to_host(y), y.cpu().numpy())


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
@pytest.mark.parametrize("n_in,n_out", [(4096, 8192), (512, 1024), (1000, 4096), (1536, 1536), (1972, 1972)])
def test_fused_pipeline_without_window_or_phase(dev, oracle, n_in, n_out, dtype):
    """`window == NULL` / `phase_table == NULL` through the C ABI: plain zero-fill + ortho FFT + fftshift, with
    and without the arg-max outputs (the >= 2x kernels fold the window into per-thread constants: the
    no-window case must still scale by 1/sqrt(n) only)."""
    x = _rand((6, n_in), dtype, seed=n_in)
    spec = oracle.to_spectrum_values(np.pad(x.astype(np.complex128), [(0, 0), (0, n_out - n_in)]), 1)
    xd = dev.to_device(x)
    tol = TIGHT[dtype] * (4 if n_out == 1972 else 1)
    pre = dev.pipeline_fused(xd, n_out, 0, want_out=False, want_argmax=True)
    np.testing.assert_array_equal(pre.argidx.cpu().numpy(), np.argmax(np.abs(spec), axis=1))
    np.testing.assert_allclose(np.sqrt(pre.absmax2.cpu().numpy().astype(np.float64)), np.abs(spec).max(axis=1),
                               rtol=20 * tol)
    both = dev.pipeline_fused(xd, n_out, 0, want_argmax=True)
    assert _relerr(both.out.cpu().numpy(), spec) < tol
    np.testing.assert_array_equal(both.argidx.cpu().numpy(), pre.argidx.cpu().numpy())
    assert _relerr(dev.pipeline_fused(xd, n_out, 0).out.cpu().numpy(), spec) < tol


MODE_SHAPES = [(4096, 8192), (1024, 2048), (2048, 2048), (3072, 3072), (700, 1000), (100, 128), (3000, 3001)]


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
@pytest.mark.parametrize("n_in,n_out", MODE_SHAPES)
def test_every_output_combination_of_the_fused_entry_point(dev, oracle, n_in, n_out, dtype):
    """xm_pipeline_fused with every combination of {window, phase table, spectrum output, arg-max outputs}: each
    combination instantiates a different kernel mode (and a different instruction schedule around the stores), so
    each is checked on its own -- for every kernel family (>= 2x zero fill, pair / scalar persistent FFT, chirp-z
    persistent and one-spectrum, small-n generic)."""
    import itertools

    import torch

    nb = 5
    x = _rand((nb, n_in), dtype, seed=7 * n_in + n_out)
    xpad = np.pad(x.astype(np.complex128), [(0, 0), (0, n_out - n_in)])
    w = oracle.exp_window(np.arange(n_out) * 2e-4, 3.0)
    freq = np.roll(np.fft.fftfreq(n_out, d=2e-4), n_out // 2)
    ph = np.exp(1j * oracle.phase_array(freq, -17.0, 250.0, float(freq[n_out // 5])))
    xd = dev.to_device(x)
    rd = torch.float32 if dtype == "complex64" else torch.float64
    wd = torch.from_numpy(w).to("cuda", rd)
    phd = torch.from_numpy(ph).to("cuda", xd.dtype)
    tol = TIGHT[dtype] * (4 if n_out in (1000, 3001) else 1)
    for use_w, use_ph, want_out, want_amax in itertools.product((False, True), repeat=4):
        if not want_out and (use_ph or not want_amax):
            continue  # nothing to produce / phase without an output
        spec = oracle.to_spectrum_values(xpad * (w if use_w else 1.0), 1)
        r = dev.pipeline_fused(xd, n_out, 0, window=wd if use_w else None, phase_table=phd if use_ph else None,
                               want_out=want_out, want_argmax=want_amax)
        tag = (use_w, use_ph, want_out, want_amax)
        if want_out:
            assert _relerr(r.out.cpu().numpy(), spec * (ph if use_ph else 1.0)) < tol, tag
        if want_amax:  # the arg-max is taken BEFORE the phase multiply (|X e^{i phi}| = |X|)
            np.testing.assert_array_equal(r.argidx.cpu().numpy(), np.argmax(np.abs(spec), axis=1), err_msg=str(tag))
            np.testing.assert_allclose(np.sqrt(r.absmax2.cpu().numpy().astype(np.float64)), np.abs(spec).max(axis=1),
                                       rtol=20 * tol, err_msg=str(tag))


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
@pytest.mark.parametrize("n", [64, 512, 1536, 2048, 4096, 5120, 8192, 1000, 1972, 4093])
def test_single_row_and_two_row_transforms(dev, oracle, n, dtype):
    """n_batch = 1 takes the one-spectrum kernels (the pair / persistent kernels need two rows), n_batch = 2 is
    the smallest input of the persistent ones: both must match numpy, with the rolls folded in."""
    for nb in (1, 2):
        x = _rand((nb, n), dtype, seed=n + nb)
        xd = dev.to_device(x)
        tol = TIGHT[dtype] * (4 if n in (1000, 1972, 4093) else 1)
        assert _relerr(dev.fft(xd, 1, shift_out=True).cpu().numpy(), oracle.to_spectrum_values(x.astype(np.complex128), 1)) < tol
        back = dev.fft(dev.fft(xd, 1, shift_out=True), 1, inverse=True, shift_in=True).cpu().numpy()
        assert _relerr(back, x.astype(np.complex128)) < 2 * tol


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
@pytest.mark.parametrize("n_in,n_out", [(4096, 8192), (2048, 2048), (1536, 1536), (1972, 1972), (100, 128)])
def test_strided_input_rows_through_the_c_abi(dev, oracle, n_in, n_out, dtype):
    """`in_row_stride` > n_in: the rows of the input are slices of wider rows (what the Bruker group-delay cut
    hands over); every kernel family must honour the stride in its first load AND in its prefetch."""
    import torch

    from xmris_amd import _lib

    nb, stride = 9, n_in + 24
    wide = _rand((nb, stride), dtype, seed=n_in)
    x = wide[:, 5:5 + n_in]
    spec = oracle.to_spectrum_values(np.pad(x.astype(np.complex128), [(0, 0), (0, n_out - n_in)]), 1)
    wd = dev.to_device(wide)
    out = torch.empty((nb, n_out), dtype=wd.dtype, device="cuda")
    am = torch.empty(nb, dtype=torch.float32 if dtype == "complex64" else torch.float64, device="cuda")
    ai = torch.empty(nb, dtype=torch.int32, device="cuda")
    code = _lib.XM_C64 if dtype == "complex64" else _lib.XM_C128
    first = wd.data_ptr() + 5 * wd.element_size()
    flags = _lib.XM_FFT_ORTHO | _lib.XM_FFT_SHIFT_OUT
    _lib.call("xm_pipeline_fused", first, stride, out.data_ptr(), None, None, nb, n_in, n_out, 0, flags,
              am.data_ptr(), ai.data_ptr(), code, torch.cuda.current_stream().cuda_stream)
    tol = TIGHT[dtype] * (4 if n_out == 1972 else 1)
    assert _relerr(out.cpu().numpy(), spec) < tol
    np.testing.assert_array_equal(ai.cpu().numpy(), np.argmax(np.abs(spec), axis=1))
    _lib.call("xm_pipeline_fused", first, stride, None, None, None, nb, n_in, n_out, 0, flags,
              am.data_ptr(), ai.data_ptr(), code, torch.cuda.current_stream().cuda_stream)
    np.testing.assert_array_equal(ai.cpu().numpy(), np.argmax(np.abs(spec), axis=1))


@pytest.mark.parametrize("dtype,n_in,n_out", [("complex64", 4096, 8192), ("complex128", 4096, 8192),
                                              ("complex64", 2048, 2048), ("complex128", 2048, 2048),
                                              ("complex64", 1000, 1000), ("complex128", 1000, 1000)])
def test_persistent_grid_boundaries(dev, oracle, dtype, n_in, n_out):
    """Batch sizes around the persistent grids (256 / 512 resident workgroups, pairs of rows per workgroup):
    one row fewer, exactly, one more, and an odd count a few grids long -- tail iterations, the prefetch guard
    and the duplicated odd row of the pair kernels."""
    import torch

    rd = torch.float32 if dtype == "complex64" else torch.float64
    for nb in (255, 256, 257, 511, 513, 1027):
        x = _rand((nb, n_in), dtype, seed=nb)
        spec = oracle.to_spectrum_values(np.pad(x.astype(np.complex128), [(0, 0), (0, n_out - n_in)]), 1)
        r = dev.pipeline_fused(dev.to_device(x), n_out, 0, want_argmax=True)
        tol = TIGHT[dtype] * (4 if n_out == 1000 else 1)
        assert _relerr(r.out.cpu().numpy(), spec) < tol, nb
        np.testing.assert_array_equal(r.argidx.cpu().numpy(), np.argmax(np.abs(spec), axis=1), err_msg=str(nb))


def test_clear_cache_between_launches(dev, oracle):
    """xm_clear_cache() frees the cached twiddle / chirp / half-rotation tables; the next launch rebuilds them.
    (Queued work must have finished first: the tables are read by in-flight kernels.)"""
    import torch

    from xmris_amd import _lib

    x = _rand((6, 1531), "complex64", seed=9)
    z = _rand((6, 2048), "complex64", seed=10)
    xd, zd = dev.to_device(x), dev.to_device(z)
    a1 = dev.fft(xd, 1, shift_out=True).cpu().numpy()
    b1 = dev.pipeline_fused(zd, 4096, 0).out.cpu().numpy()
    torch.cuda.synchronize()
    _lib.call("xm_clear_cache")
    assert np.array_equal(dev.fft(xd, 1, shift_out=True).cpu().numpy(), a1)
    assert np.array_equal(dev.pipeline_fused(zd, 4096, 0).out.cpu().numpy(), b1)
    assert _relerr(a1, oracle.to_spectrum_values(x.astype(np.complex128), 1)) < 4 * TIGHT["complex64"]


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
@pytest.mark.parametrize("nb,n_in,pad_left,n_used", [(9, 4096, 0, None), (9, 4096, 0, 2304), (7, 1531, 3, None),
                                                      (5, 1000, 0, 257), (1, 64, 10, None), (300, 512, 0, None)])
def test_row_l1_matches_numpy(dev, nb, n_in, pad_left, n_used, dtype):
    """xm_row_l1: windowed L1 norm of every row (the speculative schedule's guess), wide and scalar load paths,
    a window offset by the left pad, no window, and the truncated sum."""
    import torch

    x = _rand((nb, n_in), dtype, seed=n_in + nb)
    w = np.linspace(1.5, 0.1, n_in + 2 * pad_left + 5) * np.where(np.arange(n_in + 2 * pad_left + 5) % 7 == 0, -1.0, 1.0)
    xd = dev.to_device(x)
    rd = torch.float32 if dtype == "complex64" else torch.float64
    wd = torch.from_numpy(w).to("cuda", rd)
    m = n_in if n_used is None else n_used
    ref_w = (np.abs(x.astype(np.complex128))[:, :m] * np.abs(w[pad_left:pad_left + m])).sum(axis=1)
    ref_1 = np.abs(x.astype(np.complex128))[:, :m].sum(axis=1)
    tol = 2e-6 if dtype == "complex64" else 1e-13
    np.testing.assert_allclose(dev.row_l1(xd, wd, pad_left, n_used=n_used).cpu().numpy(), ref_w, rtol=tol)
    np.testing.assert_allclose(dev.row_l1(xd, None, 0, n_used=n_used).cpu().numpy(), ref_1, rtol=tol)
    # the sub-sampled ranking statistic: every 3rd 1-KiB block (128 complex64 / 64 complex128 samples) of the first m
    per_block = 128 if dtype == "complex64" else 64
    keep = ((np.arange(m) // per_block) % 3 == 0)
    ref_s = (np.abs(x.astype(np.complex128))[:, :m] * np.abs(w[pad_left:pad_left + m]) * keep).sum(axis=1)
    np.testing.assert_allclose(dev.row_l1(xd, wd, pad_left, n_used=n_used, sub_step=3).cpu().numpy(), ref_s, rtol=tol)


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
@pytest.mark.parametrize("nb,n_in,n_out", [(300, 4096, 8192), (70, 1024, 2048), (6, 400, 1024), (2000, 2048, 4096),
                                            (33, 8192, 16384)])
def test_guess_rows_and_refine(dev, nb, n_in, n_out, dtype):
    """xm_guess_rows: est[b] = max |X_c|^2 of the coarse spectrum (first <= 512 windowed samples, 1024 bins, the full
    transform's ortho scale), fp32 for either storage precision; xm_guess_refine: among the rows whose estimate is
    within the band of the largest one, the exact arg-max row of max |X|^2 (lowest row on ties), its value and its FID
    as complex128, and both keys left zero.  Rows are built so that the coarse estimate alone picks the WRONG row:
    row 3 carries a broad line that is fully inside the first 512 samples, row nb-2 a narrow one that is taller only
    in the full transform."""
    import torch

    rng = np.random.default_rng(n_in + nb)
    t = np.arange(n_in) * 2e-4
    x = 0.02 * (rng.standard_normal((nb, n_in)) + 1j * rng.standard_normal((nb, n_in)))
    x[3] += 1.00 * np.exp((-np.pi * 40.0 + 2j * np.pi * 500.0) * t)
    x[nb - 2] += 0.15 * np.exp((-np.pi * 0.5 + 2j * np.pi * -729.98) * t)  # half-way between two coarse bins
    x = x.astype(dtype)
    w = np.exp(-np.pi * 5.0 * np.arange(n_out) * 2e-4)
    xd = dev.to_device(x)
    w32 = torch.from_numpy(w).to("cuda", torch.float32)
    assert dev.guess_supported(xd, n_out)
    est = torch.full((nb,), -7.0, dtype=torch.float32, device="cuda")
    gkey, wkey = dev.new_argmax_key("cuda"), dev.new_argmax_key("cuda")
    dev.guess_rows(xd, n_out, w32, est, gkey)
    m = min(512, n_in)
    xc = x[:, :m].astype(np.complex128) * w[:m]
    ref_est = (np.abs(np.fft.fft(xc, n=1024, axis=1)) ** 2).max(axis=1) / n_out
    # rows of 512 samples and more take the matrix-core kernel (csrc/xm_coarse.h: fp16 operands behind a power-of-two
    # row scale, fp32 sums): an ESTIMATE, good to 2e-3; the FFT kernel (shorter rows) to 2e-5
    mfma = n_in >= 512 and not os.environ.get("XM_GUESS_FFT")
    np.testing.assert_allclose(est.cpu().numpy(), ref_est, rtol=2e-3 if mfma else 2e-5)
    assert dev.last_kernel().startswith("k_coarse_mfma" if mfma else "k_zf2p")
    full = np.abs(np.fft.fft(x.astype(np.complex128) * w[:n_in], n=n_out, axis=1)) ** 2 / n_out
    true_row = int(np.argmax(full.max(axis=1)))
    fooled = n_in > 512  # (rows no longer than the coarse stage reads: estimate and exact transform see the same samples)
    if fooled:
        assert true_row == nb - 2 and int(np.argmax(ref_est)) == 3, "the construction must fool the coarse estimate"
    gmax = torch.zeros(1, dtype=torch.float32, pin_memory=True)
    gflat = torch.zeros(1, dtype=torch.int64, pin_memory=True)
    row = torch.zeros((1, n_in), dtype=torch.complex128, device="cuda")
    dev.guess_refine(xd, n_out, w32, est, gkey, wkey, gmax, gflat, row, band=0.5)
    torch.cuda.synchronize()
    assert int(gflat.item()) == true_row * n_out
    np.testing.assert_allclose(float(gmax.item()), full[true_row].max(), rtol=2e-5)
    assert np.array_equal(row.cpu().numpy()[0], x[true_row].astype(np.complex128))
    assert int(gkey.abs().sum().item()) == 0 and int(wkey.abs().sum().item()) == 0
    # a band too narrow for the construction: the refine stage only sees row 3 and reports it (the verification of
    # the speculative schedule exists for this case)
    dev.guess_rows(xd, n_out, w32, est, gkey)
    dev.guess_refine(xd, n_out, w32, est, gkey, wkey, gmax, gflat, row, band=0.999)
    torch.cuda.synchronize()
    assert int(gflat.item()) == (3 if fooled else true_row) * n_out


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
def test_coarse_spectra_on_the_matrix_cores_scale_every_row(dev, dtype):
    """`k_coarse_mfma` feeds fp16 operands to the matrix cores behind a power-of-two scale PER ROW: rows of ADC counts
    (1e5), volts (1e-6), very large and very small numbers, a zero row and a NaN row in one launch must each come out
    within 2e-3 of the fp64 DFT of their first 512 windowed samples (phasing.py:229 is served by the exact check that
    follows; this is its ranking statistic), zero as zero, NaN as NaN, and the launch's key must name the NaN row."""
    import torch

    if os.environ.get("XM_GUESS_FFT"):
        pytest.skip("XM_GUESS_FFT: the matrix-core kernel is switched off")
    rng = np.random.default_rng(5)
    nb, n_in, n_out = 523, 1024, 2048
    t = np.arange(n_in) * 2e-4
    x = 0.05 * (rng.standard_normal((nb, n_in)) + 1j * rng.standard_normal((nb, n_in)))
    x += (rng.uniform(0.3, 1.0, nb)[:, None] * np.exp((-np.pi * rng.uniform(3, 40, nb)[:, None]
                                                       + 2j * np.pi * rng.uniform(-2400, 2400, nb)[:, None]) * t[None, :]))
    mag = 10.0 ** rng.uniform(-12, 12, nb)
    mag[:6] = [1e5, 1e-6, 3e15, 2e-15, 1.0, 1.0]
    x *= mag[:, None]
    x[4] = 0.0
    x = x.astype(dtype)
    w = np.exp(-np.pi * 5.0 * np.arange(n_out) * 2e-4)
    xc = x[:, :512].astype(np.complex128) * w[:512]
    ref = (np.abs(np.fft.fft(xc, n=1024, axis=1)) ** 2).max(axis=1) / n_out
    xd = dev.to_device(x)
    w32 = torch.from_numpy(w).to("cuda", torch.float32)
    est = torch.full((nb,), -7.0, dtype=torch.float32, device="cuda")
    gkey = dev.new_argmax_key("cuda")
    dev.guess_rows(xd, n_out, w32, est, gkey)
    assert dev.last_kernel().startswith("k_coarse_mfma")
    got = est.cpu().numpy().astype(np.float64)
    assert got[4] == 0.0
    live = ref > 0
    np.testing.assert_allclose(got[live], ref[live], rtol=2e-3)
    top = int(np.argmax(ref))
    key = int(gkey.cpu().numpy().view(np.uint64).max())
    assert 0xFFFFFFFF - (key & 0xFFFFFFFF) == top
    gkey.zero_()
    x[77, 300] = np.nan  # (inside the 512 samples the stage reads)
    dev.guess_rows(dev.to_device(x), n_out, w32, est, gkey)
    got = est.cpu().numpy()
    assert np.isnan(got[77]) and not np.isnan(np.delete(got, 77)).any()
    key = int(gkey.cpu().numpy().view(np.uint64).max())
    assert 0xFFFFFFFF - (key & 0xFFFFFFFF) == 77
    gkey.zero_()


def test_guess_stage_edge_cases(dev):
    """All-zero rows (np.argmax of zeros is 0: row 0 wins through the tie rule), a row of NaNs (always a candidate, and
    NaN outranks every number), exact ties (the lower row), one row, and geometries the stage refuses."""
    import torch

    n_in, n_out = 1024, 2048
    w32 = torch.ones(n_out, dtype=torch.float32, device="cuda")
    gmax = torch.zeros(1, dtype=torch.float32, pin_memory=True)
    gflat = torch.full((1,), -1, dtype=torch.int64, pin_memory=True)
    gkey, wkey = dev.new_argmax_key("cuda"), dev.new_argmax_key("cuda")

    def winner(x):
        xd = dev.to_device(x)
        est = torch.empty(x.shape[0], dtype=torch.float32, device="cuda")
        row = torch.zeros((1, n_in), dtype=torch.complex128, device="cuda")
        dev.guess_rows(xd, n_out, w32, est, gkey)
        dev.guess_refine(xd, n_out, w32, est, gkey, wkey, gmax, gflat, row)
        torch.cuda.synchronize()
        assert int(gkey.abs().sum().item()) == 0 and int(wkey.abs().sum().item()) == 0
        return int(gflat.item()) // n_out, float(gmax.item())

    assert winner(np.zeros((700, n_in), dtype=np.complex64)) == (0, 0.0)
    x = _rand((50, n_in), "complex64", seed=1)
    x[31] = x[12]  # an exact tie of the two largest rows
    x[12] *= 3
    x[31] *= 3
    assert winner(x)[0] == 12
    x[40, 5] = np.nan
    r, v = winner(x)
    assert r == 40 and np.isnan(v)
    assert winner(_rand((1, n_in), "complex64", seed=2))[0] == 0
    xd = dev.to_device(_rand((4, 1536), "complex64", seed=3))
    assert not dev.guess_supported(xd, 1536)          # no zero fill
    assert not dev.guess_supported(xd, 3072)          # half length without a power-of-two plan
    assert not dev.guess_supported(dev.to_device(_rand((4, 1001), "complex64", seed=3)), 2048)  # odd rows: no pair loads
    assert dev.guess_supported(dev.to_device(_rand((4, 1001), "complex128", seed=3)), 2048)


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
def test_argmax_returns_the_first_nan_like_numpy(dev, dtype):
    """phasing.py:229 `np.argmax(np.abs(values))` returns the FIRST NaN when there is one (NaN compares as the maximum
    in numpy's arg-max).  Staged arg-max on an array with NaNs, and the fused kernels on FIDs with a NaN sample (the
    whole spectrum of such a row is NaN, so the first NaN of the output is that row's first bin)."""
    x = _rand((6, 512), dtype, seed=3)
    x[4, 100] = np.nan
    x[2, 300] = complex(1.0, np.nan)
    x[1, 7] = 50.0  # a finite giant earlier in memory must not win
    amax, flat = dev.absmax_argmax(dev.to_device(x))
    assert flat == int(np.argmax(np.abs(x))) == 2 * 512 + 300 and np.isnan(amax)
    for n_in, n_out in ((512, 1024), (512, 512), (4096, 8192)):  # hot kernel / generic persistent kernels
        f = _rand((9, n_in), dtype, seed=n_in)
        f[5, n_in // 3] = np.nan
        f[7, 1] = np.nan
        for value_only in (False, True):
            res = dev.pipeline_fused(dev.to_device(f), n_out, want_out=False, want_argmax=True, argmax_value_only=value_only)
            amax, flat = dev.argmax_reduce(res.absmax2, res.argidx, n_out)
            assert flat // n_out == 5 and np.isnan(amax), (n_in, n_out, value_only, flat)
            if not value_only:
                assert flat == 5 * n_out  # all bins of row 5 are NaN: the first one


def test_arg_max_key_path_matches_the_per_row_path(dev):
    """XM_AMAX_GLOBAL_KEY / xm_row_l1's key / xm_argmax_key_take against the per-row outputs + xm_argmax_reduce:
    same winner (first row on equal values), key cleared for its next producer."""
    import torch

    nb, n_in, n_out = 3000, 1024, 2048
    x = _rand((nb, n_in), "complex64", seed=11)
    x[1717] *= 3.0
    x[2900] = x[1717]  # an exact tie: the lower row must win
    xd = dev.to_device(x)
    w = torch.linspace(1.0, 0.2, n_out, device="cuda")
    ref = dev.pipeline_fused(xd, n_out, window=w, want_argmax=True, argmax_value_only=True)
    amax_ref, flat_ref = dev.argmax_reduce(ref.absmax2, ref.argidx, n_out)
    key = dev.new_argmax_key("cuda")
    gmax, gflat = torch.empty(1, device="cuda"), torch.empty(1, dtype=torch.int64, device="cuda")
    for _ in range(2):  # twice: the key must come back cleared
        res = dev.pipeline_fused(xd, n_out, window=w, phase_ramp=(0.3, 0.001), global_key=key)
        dev.argmax_key_take(key, n_out, gmax, gflat)
        assert int(gflat.item()) // n_out == flat_ref // n_out == 1717
        assert abs(float(gmax.item()) ** 0.5 - amax_ref) <= 1e-6 * amax_ref
        assert int(key.abs().max().item()) == 0
    # ... or the kernel's last workgroup decodes into a pinned record itself
    rec = dev.new_key_result()
    for _ in range(2):
        dev.pipeline_fused(xd, n_out, window=w, phase_ramp=(0.3, 0.001), global_key=key, key_result=rec)
        torch.cuda.synchronize()
        m2, fl = dev.read_key_result(rec)
        assert fl == 1717 * n_out and abs(m2 ** 0.5 - amax_ref) <= 1e-6 * amax_ref and int(key.abs().max().item()) == 0
    # the guess kernel's key against its per-row norms, and the gather in the same launch
    norms = dev.row_l1(xd, w, 0, n_used=768, sub_step=2)
    dev.row_l1(xd, w, 0, out=None, n_used=768, sub_step=2, key=key)
    row = dev.argmax_key_take(key, n_out, gmax, gflat, xd)
    best = int(torch.argmax(norms).item())
    assert int(gflat.item()) == best * n_out and float(gmax.item()) == float(norms[best].item())
    np.testing.assert_array_equal(row.cpu().numpy()[0], x[best].astype(np.complex128))


@pytest.mark.parametrize("n_in,n_out,nb", [(4096, 8192, 1500), (8192, 16384, 300), (3000, 8192, 37)])
def test_complex128_arg_max_key(dev, n_in, n_out, nb):
    """XM_AMAX_GLOBAL_KEY for complex128 (k_zf2d: every wave leaves its (max |X|^2 bits, row) in a slot, the last
    workgroup out merges them into the result record; the next row arrives by global_load_lds): the record against the
    per-row maxima + xm_argmax_reduce -- the fp64 maximum to the bit, the lower row on an exact tie, a NaN row first --
    the spectra against the launch without the key, every mode that takes it (write + ramp, write, maxima only), and
    the predicate that says where it applies."""
    import torch

    x = _rand((nb, n_in), "complex128", seed=n_in + nb)
    x[nb // 2] *= 3.0
    x[nb - 3] = x[nb // 2]  # an exact tie: the lower row must win
    xd = dev.to_device(x)
    w = torch.linspace(1.0, 0.2, n_out, device="cuda", dtype=torch.float64)
    assert dev.key_native(xd, n_out) and not dev.key_native(dev.to_device(x[:, :1024]), 2048)  # (k_zf2<double>: no key)
    key, rec = dev.new_argmax_key("cuda"), dev.new_key_result()
    for kw in (dict(phase_ramp=(0.3, 0.001)), dict(), dict(want_out=False)):
        # (the per-row path of the SAME mode: a folded ramp factor rounds the last bits of |X|^2 differently)
        ref = dev.pipeline_fused(xd, n_out, window=w, want_argmax=True, argmax_value_only=True, **kw)
        m2_ref = float(ref.absmax2.max().item())
        for _ in range(2):  # (twice: the slots need no clearing)
            res = dev.pipeline_fused(xd, n_out, window=w, global_key=key, key_result=rec, **kw)
            torch.cuda.synchronize()
            m2, fl = dev.read_key_result(rec, complex128=True)
            assert fl == (nb // 2) * n_out and m2 == m2_ref, (kw, fl, m2, m2_ref)
        if kw.get("phase_ramp"):
            assert torch.equal(res.out, ref.out)
    x[5, 17] = np.nan
    dev.pipeline_fused(dev.to_device(x), n_out, window=w, phase_ramp=(0.3, 0.001), global_key=key, key_result=rec)
    torch.cuda.synchronize()
    m2, fl = dev.read_key_result(rec, complex128=True)
    assert fl == 5 * n_out and np.isnan(m2)
    with pytest.raises(Exception, match="result record"):  # complex128 keys are decoded by the launch itself
        dev.pipeline_fused(xd, n_out, window=w, phase_ramp=(0.3, 0.001), global_key=key)


def test_phase_ramp_equals_phase_table(dev):
    """xm_pipeline_fused_ramp against xm_pipeline_fused with the table of the same ramp: the native (factorised) path of
    the hot kernel on every half-length plan and of the two-spectra kernel (no zero fill: powers of two and 3 * 2^k),
    and the expand-to-scratch-table path elsewhere (5 * 2^k, odd lengths, complex128 without the zero fill)."""
    import torch

    for dtype, n_in, n_out, pad in (("complex64", 4096, 8192, 0), ("complex64", 2048, 4096, 0), ("complex64", 1000, 2048, 0),
                                    ("complex64", 400, 1024, 0), ("complex64", 1001, 2048, 0), ("complex64", 1536, 1536, 0),
                                    ("complex64", 4096, 4096, 0), ("complex64", 3000, 3072, 10), ("complex64", 1280, 1280, 0),
                                    ("complex64", 700, 768, 0), ("complex64", 6144, 6144, 0),
                                    ("complex64", 1000, 4096, 24), ("complex64", 8192, 16384, 0), ("complex64", 5000, 16384, 100),
                                    ("complex64", 5001, 16384, 0), ("complex128", 4096, 8192, 0), ("complex128", 1972, 1972, 0),
                                    ("complex128", 2048, 4096, 0), ("complex128", 1001, 2048, 3), ("complex128", 400, 1024, 0),
                                    ("complex128", 3000, 8192, 24), ("complex128", 2049, 4096, 0), ("complex128", 8192, 16384, 0)):
        # more rows than the persistent grid's first round
        nb = 2100 if (dtype == "complex128" and n_out <= 4096) else (600 if n_out == 16384 else 37)
        x = dev.to_device(_rand((nb, n_in), dtype, seed=n_in + n_out))
        rd = torch.float32 if dtype == "complex64" else torch.float64
        w = torch.linspace(1.0, 0.1, n_out, device="cuda", dtype=rd)
        a, b = 0.4321, -0.0123
        table = torch.from_numpy(np.exp(1j * (a + b * np.arange(n_out)))).to("cuda", x.dtype)
        ref = dev.pipeline_fused(x, n_out, pad, window=w, phase_table=table).out.cpu().numpy()
        got = dev.pipeline_fused(x, n_out, pad, window=w, phase_ramp=(a, b)).out.cpu().numpy()
        native = dev.ramp_native(x, n_out, pad)
        zf2 = 2 * (pad + n_in) <= n_out and n_out in (1024, 2048, 4096, 8192, 16384)
        # ... or, complex64 without the >= 2x zero fill, the two-spectra kernel's plans of at most 16 points per thread
        fft2 = dtype == "complex64" and not zf2 and n_out in (512, 1024, 2048, 4096, 8192, 768, 1536, 3072, 6144)
        assert native == ((zf2 and (dtype == "complex128" or (n_in % 2 == 0 and pad % 2 == 0))) or fft2), (dtype, n_in, n_out, pad)
        assert _relerr(got, ref) < (1e-6 if dtype == "complex64" else 1e-13), (dtype, n_in, n_out, pad, native)


@pytest.mark.parametrize("n_in,pad,n_out", [(4096, 0, 8192), (3000, 7, 8192), (4095, 1, 8192), (1, 0, 8192), (8192, 0, 16384),
                                            (5000, 3, 16384)])
def test_complex128_hot_kernel_modes(dev, oracle, n_in, pad, n_out):
    """k_zf2d (complex128, -> 8192: two workgroups per CU, halves one after the other, generated last-stage
    twiddles; -> 16384: one 1024-thread workgroup per CU, register twiddles) in each of its modes -- plain, ramp, ramp + per-row maxima, maxima without ramp -- against numpy in
    fp64, on more rows than one round of the persistent grid (the row queue hands out the rest), ragged / shifted
    zero fills and a row of NaNs."""
    import torch

    nb = 1100 if n_out == 8192 else 600  # (half length 8192: one 1024-thread workgroup per CU, 256 in the first round)
    x = _rand((nb, n_in), "complex128", seed=n_in + pad)
    x[5] *= 3.0
    x[17, 0] = np.nan
    w = oracle.exp_window(np.arange(n_out) * 2e-4, 3.0)
    xpad = np.zeros((nb, n_out), np.complex128)
    xpad[:, pad:pad + n_in] = x
    spec = oracle.to_spectrum_values(xpad * w, 1)
    a, b = -0.83, 0.00271
    ramp = np.exp(1j * (a + b * np.arange(n_out)))
    xd = dev.to_device(x)
    wd = torch.from_numpy(w).to("cuda")
    ok = np.ones(nb, bool)
    ok[17] = False

    def check(got, ref):
        assert np.isnan(got[17]).all()
        assert _relerr(got[ok], ref[ok]) < TIGHT["complex128"]

    check(dev.pipeline_fused(xd, n_out, pad, window=wd).out.cpu().numpy(), spec)
    check(dev.pipeline_fused(xd, n_out, pad, window=wd, phase_ramp=(a, b)).out.cpu().numpy(), spec * ramp)
    for kw in ({"phase_ramp": (a, b)}, {}):
        r = dev.pipeline_fused(xd, n_out, pad, window=wd, want_argmax=True, argmax_value_only=True, **kw)
        check(r.out.cpu().numpy(), spec * (ramp if kw else 1.0))
        m = r.absmax2.cpu().numpy()
        assert np.isnan(m[17])
        np.testing.assert_allclose(np.sqrt(m[ok]), np.abs(spec[ok]).max(axis=1), rtol=1e-12)
    # no window: the scale alone
    spec0 = oracle.to_spectrum_values(xpad, 1)
    check(dev.pipeline_fused(xd, n_out, pad, phase_ramp=(a, b)).out.cpu().numpy(), spec0 * ramp)


def test_tensor_on_a_non_current_device(dev):
    """The library's caches are per device: a tensor that lives on another GPU than the current one must be
    transformed there (tables allocated on ITS device), not through the current device's tables."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    x = _rand((8, 1024), "complex64", seed=1)
    with torch.cuda.device(0):
        y1 = dev.fft(dev.to_device(x, "cuda:1"), 1).cpu().numpy()
    assert _relerr(y1, np.fft.fft(x.astype(np.complex128), norm="ortho")) < TIGHT["complex64"]


def test_baseline_als_needs_four_points(dev):
    import torch

    from xmris_amd import _lib

    with pytest.raises(_lib.XmrisHipError):
        dev.baseline_als(torch.zeros((2, 3), dtype=torch.float64, device="cuda"), 1, 1e3, 0.01, 3)


def test_randomised_geometries_of_the_fused_entry_point(dev, oracle):
    """A seeded sweep over what the parametrised tests pin one at a time: storage precision x transform length
    (every kernel family) x zero-fill geometry (n_in, pad_left; ragged, odd, full) x batch size (below, at and far
    above the persistent grid, so static rounds, queue tickets and chunk tails all occur) x {window} x {no phase,
    table, ramp} x {no maxima, maxima + index, maxima value only} x {fftshift} -- each against numpy in fp64."""
    import torch

    import os

    rng = np.random.default_rng(int(os.environ.get("XM_SWEEP_SEED", "20240611")))  # other seeds: one-off hunts
    lengths = [64, 128, 512, 1024, 2048, 4096, 8192, 16384, 384, 768, 1536, 3072, 5120, 1000, 1531, 2000]
    batches = [1, 2, 3, 17, 255, 512, 513, 1025, 2100]
    for case in range(120):
        dtype = "complex64" if rng.random() < 0.5 else "complex128"
        n_out = int(rng.choice(lengths))
        kind = rng.integers(0, 4)
        if kind == 0:  # no zero fill
            n_in, pad = n_out, 0
        elif kind == 1:  # exactly 2x
            n_in, pad = n_out // 2, 0
        else:  # ragged, possibly shifted
            n_in = int(rng.integers(1, n_out + 1))
            pad = int(rng.integers(0, n_out - n_in + 1)) if rng.random() < 0.5 else 0
        nb = int(rng.choice(batches))
        if nb * n_out > (1 << 23):  # keep the case small enough for the numpy side
            nb = max(1, (1 << 23) // n_out)
        use_w = rng.random() < 0.7
        phase = ("none", "table", "ramp")[int(rng.integers(0, 3))]
        amax = ("none", "index", "value")[int(rng.integers(0, 3))]
        shift = rng.random() < 0.8
        tag = (case, dtype, nb, n_in, n_out, pad, use_w, phase, amax, shift)

        x = _rand((nb, n_in), dtype, seed=1000 + case)
        x[nb // 2] *= 2.5
        xpad = np.zeros((nb, n_out), np.complex128)
        xpad[:, pad:pad + n_in] = x
        w = np.exp(-np.linspace(0.0, 2.0, n_out)) if use_w else np.ones(n_out)
        spec = np.fft.fft(xpad * w, axis=1, norm="ortho")
        if shift:
            spec = np.roll(spec, n_out // 2, axis=1)
        a, b = float(rng.uniform(-3, 3)), float(rng.uniform(-0.01, 0.01))
        ph = np.exp(1j * (a + b * np.arange(n_out)))
        ref = spec * ph if phase != "none" else spec

        xd = dev.to_device(x)
        rd = torch.float32 if dtype == "complex64" else torch.float64
        kw = dict(shift_out=shift)
        if use_w:
            kw["window"] = torch.from_numpy(w).to("cuda", rd)
        if phase == "table":
            kw["phase_table"] = torch.from_numpy(ph).to("cuda", xd.dtype)
        elif phase == "ramp":
            kw["phase_ramp"] = (a, b)
        if amax != "none":
            kw.update(want_argmax=True, argmax_value_only=(amax == "value"))
        r = dev.pipeline_fused(xd, n_out, pad, **kw)
        tol = TIGHT[dtype] * (4 if n_out in (1000, 1531, 2000) else 1)
        assert _relerr(r.out.cpu().numpy(), ref) < tol, tag
        if amax != "none":  # maxima are taken BEFORE the phase (a unit factor)
            m = np.sqrt(r.absmax2.cpu().numpy().astype(np.float64))
            np.testing.assert_allclose(m, np.abs(spec).max(axis=1), rtol=30 * tol, err_msg=str(tag))
            if amax == "index":
                idx = r.argidx.cpu().numpy()
                mag = np.abs(spec)
                # the device compares in the storage precision: accept any index whose magnitude ties the maximum there
                assert np.all(mag[np.arange(nb), idx] >= mag.max(axis=1) * (1 - 30 * tol)), tag


@pytest.mark.parametrize("case", [
    # (n_batch, n_in, n_out, pad_left, in dtype, promote)
    (9, 1024, 2048, 0, "complex64", False), (9, 1024, 2048, 0, "complex64", True), (9, 1024, 2048, 0, "complex128", False),
    (5, 32, 128, 48, "complex64", True),                      # symmetric zero fill (fid.py:243-246)
    (3, 1023, 2048, 0, "complex64", False),                   # odd length: rows leave the 16-byte grid (one element per lane)
    (3, 1023, 2047, 511, "complex64", True), (4, 100, 101, 1, "complex128", False),
    (700, 4096, 8192, 0, "complex64", False),                 # more rows than resident workgroups: the row queue
    (1, 2, 4, 0, "complex64", True), (2, 4096, 4096, 0, "complex128", False),  # (no zero fill at all: a plain window)
])
def test_zero_fill_apodize_one_launch(dev, case):
    """`xm_zf_apod` (fid.py:251 followed by fid.py:136-139 in one pass): the padding is +0 bit for bit, the samples are
    numpy's products -- complex times real is two correctly rounded products in either implementation, so bit for bit
    as well -- for both positions, both precisions, numpy's complex64 -> complex128 promotion, rows on and off the
    16-byte grid, and more rows than the persistent grid holds."""
    import torch

    nb, n_in, n_out, pad_left, dtype, promote = case
    x = _rand((nb, n_in), dtype, seed=n_in + nb)
    w = np.exp(-np.pi * 5.0 * (np.arange(n_out) - pad_left) / 5000.0)
    got = dev.zf_apod(dev.to_device(x), n_out, pad_left, w, promote=promote)
    torch.cuda.synchronize()
    out_c128 = promote or dtype == "complex128"
    assert got.dtype == (torch.complex128 if out_c128 else torch.complex64) and tuple(got.shape) == (nb, n_out)
    wd = w if out_c128 else w.astype(np.float32)
    ref = np.zeros((nb, n_out), dtype=np.complex128 if out_c128 else np.complex64)
    ref[:, pad_left:pad_left + n_in] = x.astype(ref.dtype) * wd[pad_left:pad_left + n_in][None, :]
    g = got.cpu().numpy()
    assert np.array_equal(g.view(np.uint8), ref.view(np.uint8))  # signed zeros and all
    # a second launch on the same queue slot ring (the last workgroup out must have left the counters zero)
    again = dev.zf_apod(dev.to_device(x), n_out, pad_left, w, promote=promote).cpu().numpy()
    assert np.array_equal(again.view(np.uint8), ref.view(np.uint8))
