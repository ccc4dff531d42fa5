"""The ``.xmr`` accessor for the spectral hot path.

Same method names, keyword names and defaults as the reference's ``XmrisAccessor`` mixins
(``src/xmris/core/accessor.py``: Fourier 369-446, Processing 449-593, Phasing 596-683; defaults pinned
by ``tests/test_core.py:497-552``).  Works on ``xmris_amd.LabeledArray`` out of the box and on
``xarray.DataArray`` after ``register_xarray_accessor()`` (called at import when xarray is installed
and no other package owns the name).  Everything outside the hot path (plots, widgets, fitting,
ppm conversion, vendor I/O) is out of scope.
"""
from __future__ import annotations

from .config import DIMS
from .processing.fid import apodize_exp, apodize_lg, to_fid, to_spectrum, zero_fill
from .processing.fourier import fft, fftc, fftshift, ifft, ifftc, ifftshift
from .processing.phasing import autophase, phase
from .dims import _check_dims  # noqa: F401  (the reference re-exports it from its accessor module)


class XmrisFourierMixin:
    def fftshift(self, dim):
        return fftshift(self._obj, dim=dim)

    def ifftshift(self, dim):
        return ifftshift(self._obj, dim=dim)

    def fft(self, dim=DIMS.time, out_dim=None):
        return fft(self._obj, dim=dim, out_dim=out_dim)

    def ifft(self, dim=DIMS.frequency, out_dim=None):
        return ifft(self._obj, dim=dim, out_dim=out_dim)

    def fftc(self, dim=DIMS.time, out_dim=None):
        return fftc(self._obj, dim=dim, out_dim=out_dim)

    def ifftc(self, dim=DIMS.frequency, out_dim=None):
        return ifftc(self._obj, dim=dim, out_dim=out_dim)


class XmrisProcessingMixin:
    def apodize_exp(self, dim: str = DIMS.time, lb: float = 1.0):
        return apodize_exp(self._obj, dim=dim, lb=lb)

    def apodize_lg(self, dim: str = DIMS.time, lb: float = 1.0, gb: float = 1.0):
        return apodize_lg(self._obj, dim=dim, lb=lb, gb=gb)

    def to_spectrum(self, dim: str = DIMS.time, out_dim: str = DIMS.frequency):
        return to_spectrum(self._obj, dim=dim, out_dim=out_dim)

    def to_fid(self, dim: str = DIMS.frequency, out_dim: str = DIMS.time):
        return to_fid(self._obj, dim=dim, out_dim=out_dim)

    def zero_fill(self, dim: str = DIMS.time, target_points: int = 1024, position: str = "end"):
        return zero_fill(self._obj, dim=dim, target_points=target_points, position=position)

    def baseline_als(self, dim: str = DIMS.frequency, lam: float = 1e5, p: float = 0.001, n_iter: int = 10):
        from .processing.baseline import baseline_als

        return baseline_als(self._obj, dim=dim, lam=lam, p=p, n_iter=n_iter)


class XmrisPhasingMixin:
    def phase(self, dim: str = DIMS.frequency, p0: float = 0.0, p1: float = 0.0, pivot: float = None):
        return phase(self._obj, dim=dim, p0=p0, p1=p1, pivot=pivot)

    def autophase(self, dim: str = DIMS.frequency, method: str = "acme", peak_width: int = 100,
                  lb: float = 0.0, temp_time_dim: str = DIMS.time, **kwargs):
        # NB the accessor's peak_width default (100) differs from the function's (0.5), as in the
        # reference (accessor.py:634 vs phasing.py:166)
        return autophase(self._obj, dim=dim, method=method, peak_width=peak_width, lb=lb,
                         temp_time_dim=temp_time_dim, **kwargs)


class XmrisVendorMixin:
    def remove_digital_filter(self, group_delay: float, dim: str = "time", keep_length: bool = True):
        from .vendor.bruker import remove_digital_filter

        return remove_digital_filter(self._obj, group_delay=group_delay, dim=dim, keep_length=keep_length)


class XmrisFusedMixin:
    def spectral_pipeline(self, target_points: int = 1024, lb: float = 1.0, dim: str = DIMS.time,
                          out_dim: str = DIMS.frequency, method: str = "acme", peak_width: int = 100, **kwargs):
        """zero_fill -> apodize_exp -> to_spectrum -> autophase in two fused launches (an addition of
        this backend; result and metadata equal the four chained calls)."""
        from .fused import spectral_pipeline

        return spectral_pipeline(self._obj, target_points=target_points, lb=lb, dim=dim, out_dim=out_dim,
                                 method=method, peak_width=peak_width, **kwargs)


class XmrisAccessor(XmrisFourierMixin, XmrisProcessingMixin, XmrisPhasingMixin, XmrisVendorMixin, XmrisFusedMixin):
    """``obj.xmr.<method>`` for the hot-path methods."""

    def __init__(self, obj):
        self._obj = obj


def register_xarray_accessor(name: str = "xmr", force: bool = False) -> bool:
    """Register the accessor on ``xarray.DataArray``.  Returns False when xarray is missing or the
    name is already taken (e.g. by the reference package) and `force` is not set."""
    try:
        import xarray as xr
    except ImportError:
        return False
    if hasattr(xr.DataArray, name) and not force:
        return False
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xr.register_dataarray_accessor(name)(XmrisAccessor)
    return True
