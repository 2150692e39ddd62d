"""NetCDF interchange of the path: `color(c,z,y,x)` f32 through libvrm.so (vrm_netcdf_read_color /
vrm_netcdf_write_color; host code, no GPU needed).  Read: the netCDF classic containers (CDF-1/2/5) and NetCDF-4 = HDF5, what
the reference's own netcdf crate and scripts/diffdvr.py:278 write (csrc/vrm_hdf5.cpp).  Written: classic, which netcdf-c reads,
or (`format="netcdf4"`) the HDF5 container `netcdf::create` itself makes, in its oldest dialect.  Reader and writer are pinned against libhdf5
itself (tests/test_hdf5_golden_cpu.py)."""
import ctypes
import os

import numpy as np

from . import _ffi


def read_color(path):
    """-> float32 array (4, Z, Y, X), the variable `color` as loaders/netcdf.rs:45 reads it."""
    x, y, z = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
    L = _ffi.lib()
    _ffi.check(L.vrm_netcdf_read_color(os.fsencode(path), ctypes.byref(x), ctypes.byref(y), ctypes.byref(z), None, 0))
    out = np.empty((4, z.value, y.value, x.value), dtype=np.float32)
    _ffi.check(L.vrm_netcdf_read_color(os.fsencode(path), ctypes.byref(x), ctypes.byref(y), ctypes.byref(z), out.ctypes.data, out.size))
    return out


def write_color(path, color_czyx, format="classic"):
    """float32 (4, Z, Y, X) -> NetCDF file with dimensions c, z, y, x and variable `color` (netcdf.rs:10-16); `format` is
    "classic" (CDF-1/2/5 by size) or "netcdf4" (HDF5)."""
    d = np.ascontiguousarray(color_czyx, dtype=np.float32)
    if d.ndim != 4 or d.shape[0] != 4:
        raise ValueError(f"expected (4, Z, Y, X), got {d.shape}")
    if format not in ("classic", "netcdf4"):
        raise ValueError(f"format is 'classic' or 'netcdf4', got {format!r}")
    fn = _ffi.lib().vrm_netcdf_write_color if format == "classic" else _ffi.lib().vrm_netcdf_write_color_nc4
    _ffi.check(fn(os.fsencode(path), d.shape[3], d.shape[2], d.shape[1], d.ctypes.data))
