"""A small labelled-array container with the subset of the ``xarray.DataArray`` surface that the
spectral hot path touches (dims, coords with attrs, attrs, name, ``.values``, ``.copy``,
``get_axis_num``, ``isel``) and a ``.xmr`` accessor.

Why it exists: (1) ``xarray`` is an optional dependency of this backend (it is absent from the
build and GPU images), so the drop-in API must be usable and testable without it; (2) the data
may stay resident in HBM between chained ``.xmr`` calls (``data`` is then a torch tensor on the
GPU and ``.values`` copies to the host on demand), which a numpy-backed ``xarray.DataArray`` cannot
express.  ``from_xarray`` / ``to_xarray`` convert losslessly when xarray is installed.
"""
from __future__ import annotations

import copy as _copy

import numpy as np


class Coordinate:
    """One coordinate variable: the dimension it runs along, its values (host, fp64 as given) and
    attrs such as ``long_name`` / ``units``."""

    __slots__ = ("dim", "values", "attrs")

    def __init__(self, dim, values, attrs=None):
        self.dim = str(dim)
        self.values = np.asarray(values)
        self.attrs = dict(attrs or {})

    def copy(self):
        return Coordinate(self.dim, self.values.copy(), dict(self.attrs))

    def min(self):
        return self.values.min()

    def max(self):
        return self.values.max()

    def __len__(self):
        return len(self.values)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.values, dtype=dtype)

    def __repr__(self):
        return f"Coordinate(dim={self.dim!r}, n={len(self.values)}, attrs={self.attrs})"


def _is_tensor(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "is_cuda")


# set by xmris_amd.processing: a function that recognises a whole recorded chain when its END is asked for its data and
# computes it in one fused launch (returns None for anything else: the steps then run one by one)
_materialise_hook = None


class Deferred:
    """Array data that has not been computed yet (the lazy accessor chain): `thunk()` produces
    it on first use, `shape` / `dtype` are known up front (all metadata of the hot path is host arithmetic on
    coordinates), `step` = (operation name, its parameters) and `parent` = the LabeledArray it was applied to.  A
    consumer that recognises a whole recorded chain -- `autophase` on `to_spectrum(apodize_exp(zero_fill(fid)))` -- runs
    the fused kernels on the root instead and the intermediates are never materialised."""

    __slots__ = ("thunk", "shape", "dtype", "step", "parent")

    def __init__(self, thunk, shape, dtype, step, parent):
        self.thunk, self.shape, self.dtype, self.step, self.parent = thunk, tuple(shape), np.dtype(dtype), step, parent


class LabeledArray:
    def __init__(self, data, dims, coords=None, attrs=None, name=None):
        self._lazy = None
        if isinstance(data, Deferred):
            self._lazy, self._data = data, None
        else:
            self._data = data if _is_tensor(data) else np.asarray(data)
        self.dims = tuple(str(d) for d in dims)
        if len(self.dims) != len(self.shape):
            raise ValueError(f"{len(self.dims)} dims for data of shape {tuple(self.shape)}")
        self.coords = {}
        for k, c in (coords or {}).items():
            if isinstance(c, Coordinate):
                c = c.copy()
            elif isinstance(c, tuple) and len(c) in (2, 3):  # (dim, values[, attrs]) like xarray
                c = Coordinate(c[0], c[1], c[2] if len(c) == 3 else None)
            else:
                c = Coordinate(k, c)
            if c.dim in self.dims and len(c.values) != self.shape[self.dims.index(c.dim)]:
                raise ValueError(f"coordinate {k!r} has {len(c.values)} points, dim {c.dim!r} has "
                                 f"{self.shape[self.dims.index(c.dim)]}")
            self.coords[str(k)] = c
        self.attrs = dict(attrs or {})
        self.name = name

    # ---- data: computed on first use when deferred ----------------------------------------------
    @property
    def data(self):
        if self._data is None:
            fused = _materialise_hook(self) if _materialise_hook is not None else None
            self._data = fused if fused is not None else self._lazy.thunk()
            self._lazy = None  # the recorded chain (and its references to the parents) is no longer needed
        return self._data

    @data.setter
    def data(self, value):
        self._data, self._lazy = value, None

    @property
    def is_deferred(self) -> bool:
        return self._data is None

    def pending_chain(self):
        """([(operation, parameters), ...] newest first, root): the recorded, not yet computed operations that lead
        to this array and the LabeledArray they start from (itself when nothing is pending)."""
        steps, node = [], self
        while node._data is None:
            steps.append(node._lazy.step)
            node = node._lazy.parent
        return steps, node

    # ---- xarray-like surface ------------------------------------------------------------------
    @property
    def shape(self):
        return self._lazy.shape if self._data is None else tuple(self._data.shape)

    @property
    def ndim(self):
        return len(self.dims)

    @property
    def sizes(self):
        return dict(zip(self.dims, self.shape))

    @property
    def dtype(self):
        if self._data is None:
            return self._lazy.dtype
        if _is_tensor(self._data):
            return np.dtype(str(self._data.dtype).replace("torch.", ""))
        return self._data.dtype

    @property
    def is_device_resident(self) -> bool:
        return self._data is None or _is_tensor(self._data)  # deferred results are produced in HBM

    @property
    def values(self) -> np.ndarray:
        """Host ndarray (copies device-resident data over PCIe)."""
        if _is_tensor(self.data):
            from . import device as _dev

            return _dev.to_host(self.data)
        return self.data

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.values, dtype=dtype)

    def get_axis_num(self, dim) -> int:
        return self.dims.index(dim)

    def copy(self, deep=True, data=None):
        if data is None:
            data = self.data.clone() if _is_tensor(self.data) else self.data.copy()
        return LabeledArray(data, self.dims, self.coords, _copy.copy(self.attrs), self.name)

    def assign_attrs(self, *args, **kw):
        out = self.copy(data=self.data)
        for a in args:
            out.attrs.update(a)
        out.attrs.update(kw)
        return out

    def isel(self, indexers=None, **kw):
        """Integer / slice selection along named dims (integers drop the dim, like xarray)."""
        sel = dict(indexers or {}, **kw)
        idx, dims = [], []
        for d in self.dims:
            i = sel.get(d, slice(None))
            idx.append(i)
            if isinstance(i, slice):
                dims.append(d)
        coords = {}
        for k, c in self.coords.items():
            if c.dim in sel:
                i = sel[c.dim]
                if isinstance(i, slice):
                    coords[k] = Coordinate(c.dim, c.values[i], c.attrs)
            else:
                coords[k] = c
        return LabeledArray(self.data[tuple(idx)], dims, coords, _copy.copy(self.attrs), self.name)

    @property
    def real(self):
        return LabeledArray(self.values.real, self.dims, self.coords, _copy.copy(self.attrs), self.name)

    @property
    def imag(self):
        return LabeledArray(self.values.imag, self.dims, self.coords, _copy.copy(self.attrs), self.name)

    def to_host(self):
        return self.copy(data=self.values.copy()) if _is_tensor(self.data) else self

    @property
    def xmr(self):
        from .accessor import XmrisAccessor

        return XmrisAccessor(self)

    def __repr__(self):
        where = "deferred" if self._data is None else ("HBM" if _is_tensor(self._data) else "host")
        return (f"<xmris_amd.LabeledArray {self.sizes} {self.dtype} [{where}] coords={list(self.coords)} "
                f"attrs={list(self.attrs)}>")

    # ---- xarray bridge ------------------------------------------------------------------------
    @classmethod
    def from_xarray(cls, da):
        duck = getattr(da, "data", None)
        lazy_node = duck.node if isinstance(duck, LazyDuck) else None
        coords = {}
        for k, c in da.coords.items():
            if c.ndim == 1:
                coords[str(k)] = Coordinate(c.dims[0], np.asarray(c.values), dict(c.attrs))
            elif c.ndim > 1:
                # xarray carries multi-dimensional coordinates through every operation; this container cannot, and
                # dropping one silently would hand back a different object than the reference does
                raise NotImplementedError(f"coordinate {k!r} spans {c.ndim} dimensions {tuple(c.dims)}: xmris_amd carries "
                                          f"one-dimensional coordinates only (drop or reset it before the call)")
            # 0-d (scalar) coordinates carry no axis information for this path and are not kept
        if lazy_node is not None:
            # The DataArray wraps a step of a recorded chain (`to_xarray(lazy=True)`): the chain goes on -- with the
            # DataArray's CURRENT dims / coords / attrs / name, so an edit the caller made on it is seen by whoever
            # replays the chain's metadata (phasing._chain_metadata_untouched) -- and with the node's data once
            # somebody has computed them.
            data = lazy_node._lazy if lazy_node._data is None else lazy_node._data
            if tuple(da.dims) == lazy_node.dims:
                return cls(data, da.dims, coords, dict(da.attrs), da.name)
        return cls(np.asarray(da.values), da.dims, coords, dict(da.attrs), da.name)

    def to_xarray(self, lazy: bool = False):
        """An `xarray.DataArray` with this array's dims / coords / attrs / name.  Its data are host values (the
        reference's contract: numpy-backed, `accessor.py:452-550`) -- or, with `lazy` and while this array is still a
        recorded step, a `LazyDuck` around it: shape, dtype, coordinates and attrs are there at once, the next `.xmr`
        call continues the recorded chain, and `.values` / `np.asarray` / any arithmetic computes."""
        import xarray as xr

        coords = {k: xr.Variable(c.dim, c.values, attrs=dict(c.attrs)) for k, c in self.coords.items()
                  if c.dim in self.dims}
        if lazy and self.is_deferred:
            try:
                return xr.DataArray(LazyDuck(self), dims=self.dims, coords=coords, attrs=dict(self.attrs), name=self.name)
            except Exception:  # noqa: BLE001 -- an xarray that refuses the duck array gets host values
                pass
        return xr.DataArray(self.values, dims=self.dims, coords=coords, attrs=dict(self.attrs), name=self.name)


def is_xarray(obj) -> bool:
    return type(obj).__module__.startswith("xarray")


def as_labeled(obj) -> LabeledArray:
    if isinstance(obj, LabeledArray):
        return obj
    if is_xarray(obj):
        return LabeledArray.from_xarray(obj)
    raise TypeError(f"expected an xarray.DataArray or xmris_amd.LabeledArray, got {type(obj).__name__}")


def like_input(result: LabeledArray, original):
    """Return the result in the caller's container type: xarray in -> xarray out.  A result that is still a recorded
    step (zero_fill / apodize_* / to_spectrum with the lazy chain on) goes back as a DataArray around a `LazyDuck`, so
    the reference's four-call chain on DataArrays (README.md:66-73) fuses exactly like it does on LabeledArrays;
    everything else goes back with host values, as the reference's contract has it."""
    if not is_xarray(original):
        return result
    import os

    lazy = result.is_deferred and not os.environ.get("XMRIS_AMD_EAGER") and not os.environ.get("XMRIS_AMD_XARRAY_EAGER")
    return result.to_xarray(lazy=lazy)


class LazyDuck(np.lib.mixins.NDArrayOperatorsMixin):
    """A duck array in xarray's sense (`shape`, `dtype`, `ndim`, `__array__`, `__array_ufunc__`, `__array_function__`:
    xarray's documented requirements, "Working with numpy-like arrays") around ONE recorded step of the `.xmr` chain
    (`node`, a deferred `LabeledArray`).  Nothing is computed to make it or to carry it through `xr.DataArray(...)`;
    `np.asarray(duck)` -- what `DataArray.values` does -- any numpy function, any arithmetic, indexing or `astype`
    computes the node's data (on the GPU, fused where the chain allows, `processing/_common.fused_materialise`) and
    continues on the host ndarray, as the reference's numpy-backed DataArrays would (the arithmetic operators come
    from numpy's `NDArrayOperatorsMixin`, i.e. through `__array_ufunc__`)."""

    __slots__ = ("node",)

    def __init__(self, node: LabeledArray):
        self.node = node

    shape = property(lambda self: tuple(self.node.shape))
    dtype = property(lambda self: np.dtype(self.node.dtype))
    ndim = property(lambda self: len(self.node.shape))
    size = property(lambda self: int(np.prod(self.node.shape, dtype=np.int64)))
    nbytes = property(lambda self: self.size * np.dtype(self.node.dtype).itemsize)

    def __len__(self):
        if not self.node.shape:
            raise TypeError("len() of unsized object")
        return self.node.shape[0]

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.node.values, dtype=dtype)

    def __getitem__(self, key):
        return np.asarray(self)[key]

    def astype(self, dtype, **kw):
        return np.asarray(self).astype(dtype, **kw)

    def copy(self):
        return np.array(self)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        args = [np.asarray(x) if isinstance(x, LazyDuck) else x for x in inputs]
        if "out" in kwargs:
            kwargs["out"] = tuple(np.asarray(o) if isinstance(o, LazyDuck) else o for o in kwargs["out"])
        return getattr(ufunc, method)(*args, **kwargs)

    def __array_function__(self, func, types, args, kwargs):
        def plain(x):
            if isinstance(x, LazyDuck):
                return np.asarray(x)
            if isinstance(x, (list, tuple)):
                return type(x)(plain(v) for v in x)
            return x

        return func(*plain(args), **{k: plain(v) for k, v in kwargs.items()})

    def __repr__(self):
        return f"<xmris_amd.LazyDuck {self.shape} {self.dtype}: recorded {self.node._lazy.step[0] if self.node.is_deferred else 'step (computed)'}>"
