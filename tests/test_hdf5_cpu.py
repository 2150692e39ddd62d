"""NetCDF-4 = HDF5 container of `color(c,z,y,x)` (f-2; host code of libvrm.so, no GPU): the reader of csrc/vrm_hdf5.cpp against
files assembled byte by byte from the HDF5 File Format Specification (tests/hdf5_builder.py) in the dialects a netCDF-4 file of
this shape can come in, with the damage and corner cases no library writes.  The pin against the HDF5 library itself -- files it
wrote, and its own reading of the builder's files and of the writer's -- is tests/test_hdf5_golden_cpu.py."""
import os

import numpy as np
import pytest

import hdf5_builder as hb
from volume_renderer_amd import VrmError, netcdf


def _data(shape_zyx, seed=0):
    rng = np.random.default_rng(seed)
    d = rng.random((4,) + shape_zyx, dtype=np.float32)
    d[0, 0, 0, 0] = np.float32(1.0)
    d[3, -1, -1, -1] = np.float32(-0.0)
    return d


def _write(tmp_path, name, blob):
    p = os.path.join(tmp_path, name)
    with open(p, "wb") as f:
        f.write(blob)
    return p


def test_old_style_contiguous(tmp_path):
    """superblock 0, version-1 object headers, symbol-table root group, contiguous little-endian data (HDF5 1.8 defaults)"""
    d = _data((5, 6, 7), 1)
    p = _write(tmp_path, "old.nc", hb.old_style(d))
    assert netcdf.read_color(p).tobytes() == d.tobytes()


@pytest.mark.parametrize("v1_pipeline", [False, True])
@pytest.mark.parametrize("superblock", [2, 3])
def test_new_style_chunked_filtered(tmp_path, v1_pipeline, superblock):
    """superblock 2/3, OHDR headers with continuation blocks, compact links with creation order, big-endian floats, chunks behind a
    two-level B-tree with shuffle + deflate; edge chunks, a chunk never written (fill value 0), a chunk with deflate masked out"""
    d = _data((7, 10, 13), 2)
    blob, expect = hb.new_style(d, chunk=(2, 3, 4, 5), skip_chunk=5, unfiltered_chunk=9, v1_pipeline=v1_pipeline, superblock=superblock)
    p = _write(tmp_path, "new.nc", blob)
    got = netcdf.read_color(p)
    assert got.shape == d.shape
    assert (expect != d).any()                               # the skipped chunk really is missing
    assert got.tobytes() == expect.tobytes()


def test_chunked_without_filters_and_with_fletcher32(tmp_path):
    d = _data((4, 4, 6), 3)
    blob, expect = hb.new_style(d, chunk=(4, 2, 2, 3), skip_chunk=None, unfiltered_chunk=None, filters=())
    assert netcdf.read_color(_write(tmp_path, "plain.nc", blob)).tobytes() == expect.tobytes() == d.tobytes()
    blob, expect = hb.new_style(d, chunk=(1, 4, 4, 6), skip_chunk=None, unfiltered_chunk=None, filters=(2, 1, 3))
    assert netcdf.read_color(_write(tmp_path, "f32.nc", blob)).tobytes() == d.tobytes()


@pytest.mark.parametrize("deflate", [False, True])
def test_layout_v4_single_chunk(tmp_path, deflate):
    d = _data((3, 5, 4), 4)
    assert netcdf.read_color(_write(tmp_path, "v4.nc", hb.single_chunk_v4(d, deflate))).tobytes() == d.tobytes()


def test_superblock_not_at_offset_zero(tmp_path):
    """a user block in front: the superblock is searched at 512, 1024, ...; addresses are relative to the base address"""
    d = _data((2, 3, 4), 5)
    blob = bytearray(hb.old_style(d))
    blob[24:32] = (512).to_bytes(8, "little")                # base address field of the version-0 superblock
    p = _write(tmp_path, "ub.nc", b"\0" * 512 + bytes(blob))
    assert netcdf.read_color(p).tobytes() == d.tobytes()


def test_errors(tmp_path):
    d = _data((3, 4, 5), 6)
    good = hb.old_style(d)
    with pytest.raises(VrmError) as e:                       # truncated in the data
        netcdf.read_color(_write(tmp_path, "trunc.nc", good[:200]))
    assert e.value.status == -4
    with pytest.raises(VrmError) as e:                       # dimension scale x disagrees with the variable's shape
        netcdf.read_color(_write(tmp_path, "dims.nc", hb.old_style(d, dims_extra={"x": 9})))
    assert "not color(c,z,y,x)" in str(e.value)
    with pytest.raises(VrmError) as e:                       # c != 4
        netcdf.read_color(_write(tmp_path, "c3.nc", hb.old_style(d[:3])))
    assert "expected 4" in str(e.value)
    blob = bytearray(good)
    i = blob.find(b"color\0")
    blob[i:i + 5] = b"colou"
    with pytest.raises(VrmError) as e:                       # netcdf.rs:37: expect("Could not find 'color' variable")
        netcdf.read_color(_write(tmp_path, "nocolor.nc", bytes(blob)))
    assert "could not find 'color'" in str(e.value)
    with pytest.raises(VrmError):                            # neither CDF nor HDF5
        netcdf.read_color(_write(tmp_path, "junk.nc", b"GIF89a" + b"\0" * 600))
    with pytest.raises(VrmError):                            # half the file is gone: metadata beyond the end
        netcdf.read_color(_write(tmp_path, "short.nc", good[: len(good) // 2]))


def test_lookup3_known_answers():
    """hashlittle() self-check against the values printed in lookup3.c's driver5() comments"""
    assert hb.lookup3(b"", 0) == 0xDEADBEEF
    assert hb.lookup3(b"", 0xDEADBEEF) == 0xBD5B7DDE
    assert hb.lookup3(b"Four score and seven years ago", 0) == 0x17770551
    assert hb.lookup3(b"Four score and seven years ago", 1) == 0xCD628161


def _raw_read(path, capacity_floats):
    """vrm_netcdf_read_color with a caller-sized buffer, as a C / Rust caller would call it"""
    import ctypes
    from volume_renderer_amd import _ffi
    L = _ffi.lib()
    x, y, z = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
    buf = (ctypes.c_float * max(capacity_floats, 1))()
    return L.vrm_netcdf_read_color(os.fsencode(path), ctypes.byref(x), ctypes.byref(y), ctypes.byref(z), buf, capacity_floats)


def test_dimensions_whose_product_wraps_are_refused(tmp_path):
    """4*z*y*x with z = 2^22, y = x = 2^21 is 0 modulo 2^64 -- the capacity check passed and a chunked file then
    wrote through the caller's (empty) buffer.  The element count is now formed with overflow checks, in the size query too."""
    import struct
    d = _data((7, 10, 13), 2)
    shape, huge = struct.pack("<QQQQ", 4, 7, 10, 13), struct.pack("<QQQQ", 4, 1 << 22, 1 << 21, 1 << 21)
    blob, _ = hb.new_style(d, chunk=(2, 3, 4, 5), skip_chunk=None, unfiltered_chunk=None)
    assert blob.count(shape) >= 1                           # the dataspace of `color` (current and maximum dimensions)
    p = _write(tmp_path, "wrap.nc", blob.replace(shape, huge))
    assert _raw_read(p, 16) == -4                           # VRM_ERR_IO, and nothing written (here the dimension scales disagree first)
    # the same with dimension scales that agree, so that the element count itself is what refuses
    blob = hb.old_style(d, dims_extra={"x": 1 << 21, "y": 1 << 21, "z": 1 << 22})
    assert blob.count(shape) == 1
    p = _write(tmp_path, "wrap2.nc", blob.replace(shape, huge))
    assert _raw_read(p, 16) == -4
    with pytest.raises(VrmError) as e:                      # the size query refuses as well
        netcdf.read_color(p)
    assert "does not fit" in str(e.value)


def test_chunk_larger_than_the_dataset_is_refused(tmp_path):
    """a tiny file must not make the reader stage a chunk of gigabytes: chunk dimensions are bounded by the dataset's"""
    import struct
    d = _data((4, 4, 6), 3)
    blob, _ = hb.new_style(d, chunk=(4, 2, 2, 3), skip_chunk=None, unfiltered_chunk=None, filters=())
    chunk = struct.pack("<IIII", 4, 2, 2, 3)
    assert blob.count(chunk) == 1                           # the version-3 layout message
    evil = blob.replace(chunk, struct.pack("<IIII", 4, 1 << 12, 1 << 12, 3))
    with pytest.raises(VrmError) as e:
        netcdf.read_color(_write(tmp_path, "bigchunk.nc", evil))
    assert "chunk dimension" in str(e.value)


@pytest.mark.parametrize("shape", [(5, 6, 7), (1, 1, 1), (3, 130, 2)])
def test_writer_round_trip_and_second_restatement(tmp_path, shape):
    """vrm_netcdf_write_color_nc4 (what `netcdf::create` + put_values of netcdf.rs:10-29 leave on disk, oldest HDF5 dialect):
    the reader returns the array bit for bit, and the file equals, byte for byte, the one tests/hdf5_builder.py assembles for the
    same array from the specification.  Read by the HDF5 library itself in test_hdf5_golden_cpu.py; netcdf-c is not available here."""
    d = _data(shape, 7)
    p = os.path.join(tmp_path, "w.nc")
    netcdf.write_color(p, d, format="netcdf4")
    assert netcdf.read_color(p).tobytes() == d.tobytes()
    blob = open(p, "rb").read()
    model = hb.netcdf4_classic_model(d)
    assert len(blob) == len(model)
    assert blob == model
    # the superblock's end-of-file address is the file's size; the signature is HDF5's
    assert blob[:8] == b"\x89HDF\r\n\x1a\n" and int.from_bytes(blob[40:48], "little") == len(blob)


def test_writer_rejects_bad_arguments(tmp_path):
    d = _data((2, 2, 2))
    with pytest.raises(ValueError):
        netcdf.write_color(os.path.join(tmp_path, "w.nc"), d, format="hdf4")
    with pytest.raises(VrmError):
        netcdf.write_color(os.path.join(tmp_path, "no", "such", "dir", "w.nc"), d, format="netcdf4")
