"""xmris_amd -- MI355X (gfx950) backend of the xmris `.xmr` spectral hot path.

zero_fill -> apodize_exp -> to_spectrum (ortho FFT + fftshift) -> autophase, behind the
reference's accessor method names.  Hand-written HIP kernels reached through a C ABI
(`include/xmris_hip.h`, `xmris_amd/libxmris_hip.so`); Python keeps dims / coords / attrs.
"""

__version__ = "0.4.0"

from . import _lib  # noqa: F401
from .accessor import XmrisAccessor, register_xarray_accessor
from .config import ATTRS, COORDS, DIMS
from .fused import spectral_pipeline
from .labeled import Coordinate, LabeledArray
from .vendor.bruker import remove_digital_filter
from .processing import (baseline_als, apodize_exp, apodize_lg, autophase, fft, fftc, fftshift, ifft, ifftc, ifftshift, phase,
                         to_fid, to_spectrum, zero_fill)

DataArray = LabeledArray  # convenience alias for code written against xarray's constructor signature
register_xarray_accessor()  # no-op when xarray is absent or the name `xmr` is already owned

__all__ = ["baseline_als", "ATTRS", "COORDS", "DIMS", "Coordinate", "DataArray", "LabeledArray", "XmrisAccessor",
           "apodize_exp", "apodize_lg", "autophase", "fft", "fftc", "fftshift", "ifft", "ifftc", "ifftshift",
           "phase", "register_xarray_accessor", "remove_digital_filter", "spectral_pipeline", "to_fid", "to_spectrum", "zero_fill"]
