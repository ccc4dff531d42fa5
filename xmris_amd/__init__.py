"""xmris_amd -- MI355X (gfx950) backend of the xmris `.xmr` spectral hot path.

zero_fill -> apodize_exp -> to_spectrum (ortho FFT + fftshift) -> autophase, behind the
reference's accessor method names.  Hand-written HIP kernels reached through a C ABI
(`include/xmris_hip.h`, `xmris_amd/libxmris_hip.so`); Python keeps dims / coords / attrs.
"""

__version__ = "0.1.0"

from . import _lib  # noqa: F401
