"""Manual and automatic phase correction on the GPU.

Host-side mirror of the reference's ``src/xmris/processing/phasing.py``: ``phase`` applies one
(p0, p1, pivot) ramp to the whole N-D array (one broadcast-multiply launch); ``autophase``
(mode="single") finds the global |X| maximum on the device, optimises (p0, p1) on that ONE 1-D
slice on the host (same scipy differential evolution as the reference) and applies the result to
everything.
"""
from __future__ import annotations

import copy as _copy
import warnings

import numpy as np

from .. import autophase_solver as aps
from .. import device as dev
from ..config import ATTRS, DIMS
from ..dims import MSG_METHOD, MSG_MODE, MSG_MODE_ALL, _check_dims, msg_phase_units
from ._common import (Coordinate, LabeledArray, as_labeled, binary_op_name, device_data, like_input,
                      promote_for_float64_operand, to_host)


def _global_argmax(src: LabeledArray, x):
    """phasing.py:229-231 / 50-52: flat arg-max of |values| in C order, unravelled."""
    _, flat = dev.absmax_argmax(x)
    return flat, np.unravel_index(flat, src.shape)


def _phase_labeled(src: LabeledArray, x, dim, p0, p1, pivot) -> LabeledArray:
    coords = src.coords[dim].values
    table = aps.phase_table(coords, p0, p1, pivot)  # fp64 on the host (phasing.py:56-73)
    y = dev.phase_apply(promote_for_float64_operand(x), src.get_axis_num(dim), table)  # * complex128 (phasing.py:73)
    out = src.copy(data=y)
    out.name = binary_op_name(src, dim)
    out.attrs = _copy.copy(src.attrs)  # phasing.py:76
    if pivot is not None and ATTRS.phase_pivot_coord in out.attrs:  # phasing.py:79-88
        old = out.attrs[ATTRS.phase_pivot_coord]
        if old != dim:
            warnings.warn(msg_phase_units(dim, old, pivot))  # phasing.py:84-88
    out.attrs[ATTRS.phase_p0] = p0  # phasing.py:91-94
    out.attrs[ATTRS.phase_p1] = p1
    out.attrs[ATTRS.phase_pivot] = pivot
    out.attrs[ATTRS.phase_pivot_coord] = dim
    return out


def phase(da, dim: str = DIMS.frequency, p0: float = 0.0, p1: float = 0.0, pivot: float = None):
    """Zero- and first-order phase correction, phi = rad(p0) + rad(p1)*(c - pivot)/(max c - min c)
    (reference ``phasing.py:10-96``).  Default pivot: coordinate of the global |X| maximum."""
    src = as_labeled(da)
    _check_dims(src, dim, "phase")
    x, _ = device_data(src)
    if pivot is None:  # phasing.py:49-53
        _, idx = _global_argmax(src, x)
        pivot = float(src.coords[dim].values[idx[src.get_axis_num(dim)]])
    return like_input(_phase_labeled(src, x, dim, p0, p1, pivot), da)


def _fused_chain(src: LabeledArray, dim, method, peak_width, target_coord, p0_only, lb):
    """`autophase` on a spectrum that is still a RECORDED chain to_spectrum(apodize_exp([zero_fill](fid))) along one
    dim: run the fused kernels (`fused.spectral_pipeline`: two passes over the data instead of six, no intermediate
    is ever materialised) on the chain's root.  Returns None when the pattern does not apply (anything else was
    recorded, an intermediate was looked at, the FID axis is not the last one, autophase's own `lb` is set)."""
    if not src.is_deferred or lb > 0:
        return None
    steps, root = src.pending_chain()
    names = ["apodize" if s_[0] in ("apodize_exp", "apodize_lg") else s_[0] for s_ in steps]
    if names not in (["to_spectrum", "apodize", "zero_fill"], ["to_spectrum", "apodize"]):
        return None
    sp, ap = steps[0][1], steps[1][1]
    lg = steps[1][0] == "apodize_lg"
    zf = steps[2][1] if len(steps) == 3 else None
    d0 = sp["dim"]
    if sp["out_dim"] != dim or ap["dim"] != d0 or (zf is not None and zf["dim"] != d0):
        return None
    if d0 not in root.dims or root.get_axis_num(d0) != root.ndim - 1:
        return None
    if not np.issubdtype(root.dtype, np.complexfloating):
        return None
    from ..fused import spectral_pipeline

    # The fused path regenerates coordinates and attrs from the ROOT.  A caller may have edited an intermediate of the
    # recorded chain without looking at its data (`zf.coords["time"] = ...`, `zf.attrs[...] = ...`): the staged chain
    # honours such edits, so the fused path only runs when every intermediate still carries what the recording call
    # produced (the coordinate of the FID axis and the attrs are compared with a regenerated chain of metadata).
    if not _chain_metadata_untouched(src, root, steps):
        return None
    # the staged chain is complex128 from apodize_exp on (`_promote`); host-resident roots stay on the host here: the
    # fused front end uploads them chunk by chunk, overlapped with the passes
    base = root
    n = root.sizes[d0]
    # the weights as recorded (both windows), and for Lorentz-to-Gauss the lineage attrs that call stamps (fid.py:190-196)
    extra = dict(_window=ap["_weight"])
    if lg:
        from ..config import ATTRS

        extra["_apodization_attrs"] = {ATTRS.apodization_lb: ap["lb"], ATTRS.apodization_gb: ap["gb"]}
    return spectral_pipeline(base, target_points=zf["target_points"] if zf is not None else n, lb=ap["lb"], dim=d0,
                             out_dim=dim, position=zf["position"] if zf is not None else "end", method=method,
                             peak_width=peak_width, target_coord=target_coord, p0_only=p0_only, _promote=True, **extra)


def _chain_metadata_untouched(src: LabeledArray, root: LabeledArray, steps) -> bool:
    """Replay the recorded steps' METADATA (host arithmetic only, nothing is computed on the device) from the root and
    compare coordinates, attrs and names with what the chain's arrays carry now."""
    from . import fid as _fid

    nodes = []
    node = src
    while node.is_deferred:
        nodes.append(node)
        node = node._lazy.parent
    if node is not root or len(nodes) != len(steps):
        return False
    replay = root
    for (name, kw), have in zip(reversed(steps), reversed(nodes)):
        if name == "zero_fill":
            replay = _fid.zero_fill(replay, dim=kw["dim"], target_points=kw["target_points"], position=kw["position"])
        elif name == "apodize_exp":
            replay = _fid.apodize_exp(replay, dim=kw["dim"], lb=kw["lb"])
        elif name == "apodize_lg":
            replay = _fid.apodize_lg(replay, dim=kw["dim"], lb=kw["lb"], gb=kw["gb"])
        elif name == "to_spectrum":
            replay = _fid.to_spectrum(replay, dim=kw["dim"], out_dim=kw["out_dim"])
        else:
            return False
        if have.dims != replay.dims or have.attrs != replay.attrs or have.name != replay.name:
            return False
        if set(have.coords) != set(replay.coords):
            return False
        for k, c in replay.coords.items():
            hc = have.coords[k]
            if hc.dim != c.dim or hc.attrs != c.attrs or not np.array_equal(hc.values, c.values, equal_nan=True):
                return False
    return True


def autophase(da, dim: str = DIMS.frequency, method: str = "acme", mode: str = "single",
              peak_width: float = 0.5, target_coord: float | None = None, p0_only: bool = False,
              lb: float = 0.0, temp_time_dim: str = DIMS.time, **kwargs):
    """Automatic phase correction (reference ``phasing.py:161-290``)."""
    src = as_labeled(da)
    _check_dims(src, dim, "autophase")
    kwargs.setdefault("disp", False)
    if mode == "all":
        raise NotImplementedError(MSG_MODE_ALL)
    elif mode != "single":
        raise ValueError(MSG_MODE)
    if method not in aps.METHODS:
        raise ValueError(MSG_METHOD)
    fused = _fused_chain(src, dim, method, peak_width, target_coord, p0_only, lb)
    if fused is not None:
        return like_input(fused, da)

    coords = src.coords[dim].values
    x, _ = device_data(src)
    ax = src.get_axis_num(dim)
    _, idx = _global_argmax(src, x)  # device reduction, 16 bytes to the host
    if target_coord is not None:  # phasing.py:233-235
        target_idx = int(np.argmin(np.abs(coords - target_coord)))
        pivot = float(target_coord)
    else:  # phasing.py:237-238
        target_idx = int(idx[ax])
        pivot = float(coords[target_idx])

    # the ONE 1-D slice through the global maximum (phasing.py:241-242), upcast to complex128
    sel = tuple(slice(None) if i == ax else int(j) for i, j in enumerate(idx))
    sl = np.asarray(to_host(x[sel]), dtype=np.complex128)
    index_width = aps.index_width_of(coords, peak_width)  # phasing.py:245-247

    work, work_coords = sl, coords
    if lb > 0:  # phasing.py:250-253, on the 1-D slice only (three tiny launches)
        from .fid import apodize_exp, to_fid, to_spectrum

        one = LabeledArray(sl, (dim,), {dim: Coordinate(dim, coords)})
        tmp = to_spectrum(apodize_exp(to_fid(one, dim=dim, out_dim=temp_time_dim), dim=temp_time_dim, lb=lb),
                          dim=temp_time_dim, out_dim=dim)
        work, work_coords = tmp.values, tmp.coords[dim].values

    if method not in aps.METHODS:
        raise ValueError(MSG_METHOD)
    import os

    # the search's final polish follows the reference's route wherever it does anything (pipeline.run, polish="exact")
    p0_opt, p1_opt, _ = aps.solve(work, work_coords, pivot, target_idx, index_width, method=method,
                                  p0_only=p0_only, disp=kwargs.get("disp"), threads=aps.burst_threads(),
                                  polish=os.environ.get("XMRIS_AMD_POLISH", "exact"))
    return like_input(_phase_labeled(src, x, dim, p0_opt, p1_opt, pivot), da)  # phasing.py:290
