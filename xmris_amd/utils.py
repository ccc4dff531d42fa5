"""Validation helpers (reference ``src/xmris/core/utils.py:8-33``)."""
from __future__ import annotations

from .config import XmrisTerm


def _check_dims(da, dims, method_name: str) -> None:
    """Raise the reference's ``ValueError`` (utils.py:13-21) when a requested dim is absent."""
    dims_to_check = [dims] if isinstance(dims, str) else list(dims)
    missing = [d for d in dims_to_check if d not in da.dims]
    if missing:
        raise ValueError(
            f"Method '{method_name}' attempted to operate on missing "
            f"dimension(s): {missing}.\n"
            f"Available dimensions are: {list(da.dims)}.\n\n"
            f"To fix this, either pass the correct `dim` string argument to the function,"
            f" or rename your data's axes using xarray:\n"
            f"    >>> obj = obj.rename({{{repr(missing[0])}: 'correct_name'}})"
        )


def term_attrs(term: XmrisTerm) -> dict:
    """Coordinate attrs injected for a vocabulary term (utils.py:29-33 ``as_variable``)."""
    attrs = {"long_name": term.long_name}
    if term.unit:
        attrs["units"] = term.unit
    return attrs
