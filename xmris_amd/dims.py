"""Dimension validation and coordinate attrs of the host layer.

Behavioural contract (reference ``src/xmris/core/utils.py:8-33``, pinned by its ``tests/test_core.py:404-434``):
a method asked to work along a dimension the array does not have raises ``ValueError`` naming the method, the
missing dimension(s) and the available ones, and pointing at ``.rename`` as the fix; coordinates created for a
vocabulary term carry its ``long_name`` / ``units``.  The wording below is this package's own.
"""
from __future__ import annotations

from .config import XmrisTerm

_MISSING_DIM_HELP = (
    "Method '{method}' cannot run: missing dimension(s) {missing} on this array.\n"
    "Available dimensions: {have}.\n\n"
    "Either pass the axis you mean through the `dim` argument, or rename the axis first:\n"
    "    >>> obj = obj.rename({{{first!r}: 'correct_name'}})"
)


def _check_dims(da, dims, method_name: str) -> None:
    wanted = (dims,) if isinstance(dims, str) else tuple(dims)
    have = tuple(da.dims)
    missing = [d for d in wanted if d not in have]
    if not missing:
        return
    raise ValueError(_MISSING_DIM_HELP.format(method=method_name, missing=missing, have=list(have), first=missing[0]))


def term_attrs(term: XmrisTerm) -> dict:
    """Coordinate attrs for a vocabulary term: ``long_name`` always, ``units`` when the term has a unit."""
    out = {"long_name": term.long_name}
    if term.unit:
        out["units"] = term.unit
    return out
