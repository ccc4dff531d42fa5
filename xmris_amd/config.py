"""Vocabulary of dimension / coordinate / attribute names used on the spectral hot path.

Restates the NAMES and units of the reference's singletons (``src/xmris/core/config.py:9-44``
term class, ``:128-200`` attrs, ``:229-283`` dims/coords, ``:331-334`` singletons) so that the
drop-in emits identical dims, coordinate attrs (``long_name``/``units``) and lineage keys.
Descriptions are this project's own wording.
"""
from __future__ import annotations


class XmrisTerm(str):
    """A ``str`` that also carries ``description``, ``unit`` and a display ``long_name``."""

    def __new__(cls, value: str, description: str = "", unit: str = ""):
        obj = str.__new__(cls, value)
        obj.description = description
        obj.unit = unit
        return obj

    @property
    def long_name(self) -> str:
        return self.replace("_", " ").title()


class _Vocabulary:
    def _terms(self) -> dict:
        return {k: v for k, v in vars(type(self)).items() if isinstance(v, XmrisTerm)}

    def get_description(self, value: str) -> str:
        for term in self._terms().values():
            if term == value:
                return f"{term.description} [{term.unit}]" if term.unit else term.description
        return "No description available."

    def __iter__(self):
        return iter(self._terms().values())


class _Attrs(_Vocabulary):
    reference_frequency = XmrisTerm("reference_frequency", "Larmor frequency of the observed nucleus.", "MHz")
    carrier_ppm = XmrisTerm("carrier_ppm", "Chemical shift that sits at 0 Hz of the baseband signal.", "ppm")
    b0_field = XmrisTerm("b0_field", "Static field strength.", "Tesla")
    phase_p0 = XmrisTerm("phase_p0", "Zero-order phase applied to the whole spectrum.", "degrees")
    phase_p1 = XmrisTerm("phase_p1", "First-order phase: total twist over the full axis range.", "degrees")
    phase_pivot = XmrisTerm("phase_pivot", "Coordinate at which the first-order term vanishes.",
                            "dimension-dependent")
    phase_pivot_coord = XmrisTerm("phase_pivot_coord", "Name of the coordinate the pivot is expressed in.")
    apodization_lb = XmrisTerm("apodization_lb", "Exponential line broadening that was applied.", "Hz")
    apodization_gb = XmrisTerm("apodization_gb", "Gaussian broadening that was applied.", "Hz")
    zero_fill_target = XmrisTerm("zero_fill_target", "Number of points after zero filling.")
    zero_fill_position = XmrisTerm("zero_fill_position", "Where zeros were added: 'end' or 'symmetric'.")
    baseline_method = XmrisTerm("baseline_method", "Algorithm that estimated the removed baseline.")
    baseline_lam = XmrisTerm("baseline_lam", "Smoothness penalty of the AsLS baseline.")
    baseline_p = XmrisTerm("baseline_p", "Asymmetry parameter of the AsLS baseline.")
    baseline_iter = XmrisTerm("baseline_iter", "Number of AsLS re-weighting iterations.")


class _Dims(_Vocabulary):
    time = XmrisTerm("time", "Time axis of an FID.")
    frequency = XmrisTerm("frequency", "Relative frequency axis in Hz.")
    chemical_shift = XmrisTerm("chemical_shift", "Chemical-shift axis in ppm.")
    component = XmrisTerm("component", "Real / imaginary split axis.")
    average = XmrisTerm("average", "Signal averages.")
    coil = XmrisTerm("coil", "Receive coils.")
    echo = XmrisTerm("echo", "Echoes.")
    kx = XmrisTerm("kx", "k-space axis x.")
    ky = XmrisTerm("ky", "k-space axis y.")
    kz = XmrisTerm("kz", "k-space axis z.")
    x = XmrisTerm("x", "Image axis x.")
    y = XmrisTerm("y", "Image axis y.")
    z = XmrisTerm("z", "Image axis z.")


class _Coords(_Vocabulary):
    time = XmrisTerm("time", "Time coordinates.", "s")
    frequency = XmrisTerm("frequency", "Frequency coordinates.", "Hz")
    chemical_shift = XmrisTerm("chemical_shift", "Chemical-shift coordinates.", "ppm")
    kx = XmrisTerm("kx", "k-space coordinates along x.", "1/m")
    ky = XmrisTerm("ky", "k-space coordinates along y.", "1/m")
    kz = XmrisTerm("kz", "k-space coordinates along z.", "1/m")
    x = XmrisTerm("x", "Spatial coordinates along x.", "mm")
    y = XmrisTerm("y", "Spatial coordinates along y.", "mm")
    z = XmrisTerm("z", "Spatial coordinates along z.", "mm")


ATTRS = _Attrs()
DIMS = _Dims()
COORDS = _Coords()
