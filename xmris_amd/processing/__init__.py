"""GPU implementations of the reference's ``xmris.processing`` functions on the spectral hot path."""
from .baseline import baseline_als
from .fid import apodize_exp, apodize_lg, to_fid, to_spectrum, zero_fill
from .fourier import fft, fftc, fftshift, ifft, ifftc, ifftshift
from .phasing import autophase, phase

__all__ = ["baseline_als", "apodize_exp", "apodize_lg", "to_fid", "to_spectrum", "zero_fill", "fft", "fftc", "fftshift", "ifft",
           "ifftc", "ifftshift", "autophase", "phase"]

# the lazy chain's end computes itself in one fused launch where it can (labeled.LabeledArray.data)
from .. import labeled as _labeled
from ._common import fused_materialise as _fused_materialise

_labeled._materialise_hook = _fused_materialise
