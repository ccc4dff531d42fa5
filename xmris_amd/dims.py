"""Dimension validation and coordinate attrs of the host layer.

Behavioural contract (reference ``src/xmris/core/utils.py:8-33``, pinned by its ``tests/test_core.py:404-434``):
a method asked to work along a dimension the array does not have raises ``ValueError`` naming the method, the
missing dimension(s) and the available ones, and pointing at ``.rename`` as the fix; coordinates created for a
vocabulary term carry its ``long_name`` / ``units``.

The TEXTS of the user-visible errors and warnings of this path are part of the drop-in surface (callers match on them
with ``pytest.raises(match=...)`` / warning filters), so they are the reference's, kept in this one place:
``core/utils.py:14-21``, ``processing/fid.py:248``, ``processing/phasing.py:84-88, 220-224, 268``.
"""
from __future__ import annotations

from .config import XmrisTerm

_MISSING_DIM_HELP = (
    "Method '{method}' attempted to operate on missing dimension(s): {missing}.\n"
    "Available dimensions are: {have}.\n\n"
    "To fix this, either pass the correct `dim` string argument to the function,"
    " or rename your data's axes using xarray:\n"
    "    >>> obj = obj.rename({{{first!r}: 'correct_name'}})"
)
MSG_POSITION = "`position` must be either 'end' or 'symmetric'."
MSG_MODE = "Mode must be 'single' or 'all'."
MSG_MODE_ALL = "Applying autophase to each spectrum individually ('all') is not yet implemented."
MSG_METHOD = "Method must be 'acme', 'peak_minima', or 'positivity'"


def msg_phase_units(dim, old_coord, pivot) -> str:
    return (f"Applying phase in '{dim}', but previous phase operations were recorded in '{old_coord}'. "
            f"Ensure your pivot value ({pivot}) matches the current dimension's units.")


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
