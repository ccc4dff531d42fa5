"""TEST DOUBLE (not product code): the slice of the `xarray` API that `xmris_amd`'s bridge touches
(`xmris_amd/labeled.py::from_xarray / to_xarray / like_input`, `accessor.py::register_xarray_accessor`), so that the
xarray face of the drop-in can be exercised on boxes without xarray (it is absent from the build and GPU images).
`install(monkeypatch)` puts a module named `xarray` into sys.modules: `DataArray` (values / dims / coords / attrs /
name), `Variable(dim, values, attrs)`, `register_dataarray_accessor` (a cached per-object accessor property, as
xarray's `_CachedAccessor` does).  Reference: `core/accessor.py:691-710` registers on `xr.DataArray` the same way."""
import sys
import types

import numpy as np


def is_duck_array(value) -> bool:
    """xarray's own test (`xarray.namedarray.utils.is_duck_array`, documented under "Working with numpy-like arrays"):
    an ndarray, or something with ndim / shape / dtype that implements `__array_function__` AND `__array_ufunc__`, or
    the array-API namespace."""
    if isinstance(value, np.ndarray):
        return True
    return (hasattr(value, "ndim") and hasattr(value, "shape") and hasattr(value, "dtype")
            and ((hasattr(value, "__array_function__") and hasattr(value, "__array_ufunc__"))
                 or hasattr(value, "__array_namespace__")))


class Variable:
    __module__ = "xarray.core.variable"

    def __init__(self, dims, data, attrs=None):
        self.dims = (dims,) if isinstance(dims, str) else tuple(dims)
        self.values = np.asarray(data)
        self.attrs = dict(attrs or {})

    @property
    def ndim(self):
        return self.values.ndim


class DataArray:
    __module__ = "xarray.core.dataarray"

    def __init__(self, data, dims=None, coords=None, attrs=None, name=None):
        # `as_compatible_data`: duck arrays are kept as they are (never coerced), anything else becomes an ndarray
        self._data = data if is_duck_array(data) else np.asarray(data)
        if len(self._data.shape) != self._data.ndim:
            raise ValueError("inconsistent duck array")
        self.dims = tuple(dims) if dims is not None else tuple(f"dim_{i}" for i in range(self._data.ndim))
        if len(self.dims) != self._data.ndim:
            raise ValueError(f"different number of dimensions on data and dims: {self._data.ndim} vs {len(self.dims)}")
        self.coords = {}
        for k, c in (coords or {}).items():
            if isinstance(c, Variable):
                v = c
            elif isinstance(c, tuple):
                v = Variable(c[0], c[1], c[2] if len(c) > 2 else None)
            else:
                v = Variable(k, c)
            for d, n in zip(v.dims, v.values.shape):
                if d in self.dims and self._data.shape[self.dims.index(d)] != n:
                    raise ValueError(f"conflicting sizes for dimension {d!r}")
            self.coords[k] = v
        self.attrs = dict(attrs or {})
        self.name = name

    @property
    def data(self):
        """The wrapped array itself (a duck array stays what it is)."""
        return self._data

    @property
    def values(self):
        """`np.asarray(self.data)`: always a host ndarray."""
        return np.asarray(self._data)

    @property
    def shape(self):
        return tuple(self._data.shape)

    @property
    def dtype(self):
        return self._data.dtype

    @property
    def ndim(self):
        return self._data.ndim


def _register_dataarray_accessor(name):
    def deco(cls):
        def getter(self):
            cache = self.__dict__.setdefault("_accessor_cache", {})
            if name not in cache:
                cache[name] = cls(self)
            return cache[name]

        setattr(DataArray, name, property(getter))
        return cls

    return deco


def install(monkeypatch):
    mod = types.ModuleType("xarray")
    mod.DataArray, mod.Variable = DataArray, Variable
    mod.is_duck_array = is_duck_array
    mod.register_dataarray_accessor = _register_dataarray_accessor
    monkeypatch.setitem(sys.modules, "xarray", mod)
    for attr in ("xmr", "xmr_amd"):  # accessors registered by an earlier test
        if attr in DataArray.__dict__:
            monkeypatch.delattr(DataArray, attr)
    return mod
