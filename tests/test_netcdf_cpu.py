"""The NetCDF classic container of f-2 (host code of libvrm.so, no GPU): read what scipy writes, write what scipy reads
(CDF-1 and CDF-2), CDF-5 round trip, the header cases a real file can have, and the error behaviour."""
import os
import struct

import numpy as np
import pytest

from volume_renderer_amd import VrmError, netcdf

scipy_io = pytest.importorskip("scipy.io")


def _data(shape_zyx, seed=0):
    rng = np.random.default_rng(seed)
    return rng.random((4,) + shape_zyx, dtype=np.float32)


def _scipy_write(path, data, version, extra=True):
    f = scipy_io.netcdf_file(path, "w", version=version)
    if extra:   # a realistic header: other dimensions, attributes and variables before and after `color`
        f.history = "made by a test"
        f.createDimension("t", 3)
        v0 = f.createVariable("time", "d", ("t",))
        v0[:] = [0.0, 1.0, 2.0]
        v0.units = "s"
    for n, k in zip("czyx", data.shape):
        f.createDimension(n, k)
    col = f.createVariable("color", "f", ("c", "z", "y", "x"))
    col[:] = data
    col.long_name = "premultiplied colour and density"
    if extra:
        v1 = f.createVariable("flag", "b", ("c",))
        v1[:] = [1, 2, 3, 4]
    f.close()


@pytest.mark.parametrize("version", [1, 2])
@pytest.mark.parametrize("extra", [False, True])
def test_reads_scipy_files(tmp_path, version, extra):
    d = _data((5, 6, 7), version)
    p = os.path.join(tmp_path, "a.nc")
    _scipy_write(p, d, version, extra)
    got = netcdf.read_color(p)
    assert got.shape == d.shape and np.array_equal(got, d)


def test_scipy_reads_our_files(tmp_path):
    d = _data((3, 9, 4), 7)
    d[0, 0, 0, 0] = np.float32(1.0)
    d[3, 2, 8, 3] = np.float32(-0.0)
    p = os.path.join(tmp_path, "w.nc")
    netcdf.write_color(p, d)
    f = scipy_io.netcdf_file(p, "r", mmap=False)
    assert list(f.dimensions) == ["c", "z", "y", "x"] and [f.dimensions[k] for k in "czyx"] == [4, 3, 9, 4]   # netcdf.rs:11-14
    v = f.variables["color"]
    assert v.dimensions == ("c", "z", "y", "x") and v.data.dtype == np.dtype(">f4")
    assert np.array_equal(np.asarray(v.data, dtype=np.float32), d)
    f.close()
    assert np.array_equal(netcdf.read_color(p), d)
    assert open(p, "rb").read(4) == b"CDF\x01"


def _cdf5(path, data):
    """A CDF-5 file written by hand from the format specification (8-byte counts, dimids and offsets)."""
    def name(s):
        b = s.encode()
        return struct.pack(">q", len(b)) + b + b"\0" * (-len(b) % 4)
    h = b"CDF\x05" + struct.pack(">q", 0)
    h += struct.pack(">iq", 0x0A, 4) + b"".join(name(n) + struct.pack(">q", k) for n, k in zip("czyx", data.shape))
    h += struct.pack(">iq", 0x0C, 1) + name("title") + struct.pack(">iq", 2, 5) + b"hello\0\0\0"
    var = name("color") + struct.pack(">q", 4) + struct.pack(">4q", 0, 1, 2, 3) + struct.pack(">iq", 0, 0) + struct.pack(">iq", 5, data.size * 4)
    h += struct.pack(">iq", 0x0B, 1) + var
    begin = len(h) + 8
    open(path, "wb").write(h + struct.pack(">q", begin) + data.astype(">f4").tobytes())


def test_reads_cdf5(tmp_path):
    d = _data((4, 3, 5), 3)
    p = os.path.join(tmp_path, "five.nc")
    _cdf5(p, d)
    assert np.array_equal(netcdf.read_color(p), d)


@pytest.mark.parametrize("code", ["d", "i", "h", "b"])
def test_other_numeric_types_are_converted_like_netcdf_c(tmp_path, code):
    """loaders/netcdf.rs:45 reads with get::<f32>, i.e. nc_get_vara_float: netcdf-c hands any numeric external type over as
    (float)value.  Files by scipy.io.netcdf_file with `color` as double, int, short, byte."""
    rng = np.random.default_rng(7)
    if code == "d":
        v = rng.random((4, 3, 4, 5)) * 2.0 - 0.5             # 53-bit fractions: the cast rounds
    else:
        lim = {"i": 2 ** 31 - 1, "h": 2 ** 15 - 1, "b": 127}[code]
        v = rng.integers(-lim - 1, lim + 1, (4, 3, 4, 5))
    p = os.path.join(tmp_path, "typed.nc")
    f = scipy_io.netcdf_file(p, "w")
    for n, k in zip("czyx", v.shape):
        f.createDimension(n, k)
    var = f.createVariable("color", code, ("c", "z", "y", "x"))
    var[:] = v
    want = np.array(var[:]).astype(np.float32)
    f.close()
    got = netcdf.read_color(p)
    assert got.dtype == np.float32 and got.tobytes() == want.tobytes()
    if code == "d":
        assert (want.astype(np.float64) != v).any()


def test_double_beyond_float_range_is_an_error(tmp_path):
    """nc_get_vara_float returns NC_ERANGE for a double no float can hold, and the reference's `?` fails the load"""
    p = os.path.join(tmp_path, "huge.nc")
    f = scipy_io.netcdf_file(p, "w")
    for n, k in zip("czyx", (4, 2, 2, 2)):
        f.createDimension(n, k)
    v = np.zeros((4, 2, 2, 2))
    v[1, 1, 0, 1] = 1e300
    f.createVariable("color", "d", ("c", "z", "y", "x"))[:] = v
    f.close()
    with pytest.raises(VrmError, match="NC_ERANGE"):
        netcdf.read_color(p)


def test_errors(tmp_path):
    with pytest.raises(VrmError):
        netcdf.read_color(os.path.join(tmp_path, "missing.nc"))
    hdf = os.path.join(tmp_path, "four.nc")
    open(hdf, "wb").write(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)       # an HDF5 signature and nothing behind it (tests/test_hdf5_cpu.py
    with pytest.raises(VrmError) as e:                                 # covers the NetCDF-4 reader proper)
        netcdf.read_color(hdf)
    assert e.value.status == -4
    junk = os.path.join(tmp_path, "junk.nc")
    open(junk, "wb").write(b"hello world, definitely not netcdf")
    with pytest.raises(VrmError):
        netcdf.read_color(junk)
    # no `color` variable (netcdf.rs:37 expect) / wrong type / wrong dimension order / c != 4 / truncated data
    p = os.path.join(tmp_path, "nocolor.nc")
    f = scipy_io.netcdf_file(p, "w")
    f.createDimension("x", 2)
    f.createVariable("density", "f", ("x",))[:] = [1, 2]
    f.close()
    with pytest.raises(VrmError, match="color"):
        netcdf.read_color(p)
    p = os.path.join(tmp_path, "text.nc")
    f = scipy_io.netcdf_file(p, "w")
    for n, k in zip("czyx", (4, 2, 2, 2)):
        f.createDimension(n, k)
    f.createVariable("color", "c", ("c", "z", "y", "x"))[:] = np.full((4, 2, 2, 2), b"a")
    f.close()
    with pytest.raises(VrmError, match="not numeric"):       # nc_get_vara_float on text: NC_ECHAR
        netcdf.read_color(p)
    p = os.path.join(tmp_path, "order.nc")
    f = scipy_io.netcdf_file(p, "w")
    for n, k in zip("czyx", (4, 2, 2, 2)):
        f.createDimension(n, k)
    f.createVariable("color", "f", ("c", "x", "y", "z"))[:] = np.zeros((4, 2, 2, 2), np.float32)
    f.close()
    with pytest.raises(VrmError):
        netcdf.read_color(p)
    p = os.path.join(tmp_path, "c3.nc")
    f = scipy_io.netcdf_file(p, "w")
    for n, k in zip("czyx", (3, 2, 2, 2)):
        f.createDimension(n, k)
    f.createVariable("color", "f", ("c", "z", "y", "x"))[:] = np.zeros((3, 2, 2, 2), np.float32)
    f.close()
    with pytest.raises(VrmError):
        netcdf.read_color(p)
    good = os.path.join(tmp_path, "good.nc")
    netcdf.write_color(good, _data((4, 4, 4)))
    trunc = os.path.join(tmp_path, "trunc.nc")
    open(trunc, "wb").write(open(good, "rb").read()[:-16])
    with pytest.raises(VrmError, match="truncated"):
        netcdf.read_color(trunc)
    with pytest.raises(ValueError):
        netcdf.write_color(good, np.zeros((3, 2, 2, 2), np.float32))


def test_cdf5_dimensions_whose_product_wraps_are_refused(tmp_path):
    """three dimensions below 2^32 each can still wrap the 64-bit element count (2^22 * 2^21 * 2^21 * 4 = 2^66)"""
    class Shape:                                             # just enough of an array for _cdf5's header
        shape = (4, 1 << 22, 1 << 21, 1 << 21)
        size = 0
        def astype(self, _):
            return np.zeros(4, ">f4")
    p = os.path.join(tmp_path, "wrap5.nc")
    _cdf5(p, Shape())
    with pytest.raises(VrmError) as e:
        netcdf.read_color(p)
    assert e.value.status == -4
