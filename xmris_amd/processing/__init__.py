"""GPU implementations of the reference's ``xmris.processing`` functions on the spectral hot path."""
from .baseline import baseline_als
from .fid import apodize_exp, apodize_lg, to_fid, to_spectrum, zero_fill
from .fourier import fft, fftc, fftshift, ifft, ifftc, ifftshift
from .phasing import autophase, phase

__all__ = ["baseline_als", "apodize_exp", "apodize_lg", "to_fid", "to_spectrum", "zero_fill", "fft", "fftc", "fftshift", "ifft",
           "ifftc", "ifftshift", "autophase", "phase"]
