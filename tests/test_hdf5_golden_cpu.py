"""NetCDF-4 = HDF5 (f-2), PINNED against the real HDF5 library (libhdf5 1.10.6 -- the library netcdf-c, and through it the reference's
`netcdf` crate, sits on; this image carries it with its command-line tools under /opt/conda):

* tests/golden/hdf5/lib_*.nc were written BY libhdf5 (tools/make_hdf5_golden.c: `color(c,z,y,x)` f32 with netcdf-c's dimension scales and
  attributes, in every storage dialect the library offers for such a dataset); the reader of csrc/vrm_hdf5.cpp must return their values
  bit for bit.  The fixtures travel with the repository: these tests need no library at run time.
* where the library's tools are present (`h5dump` under /opt/conda/bin, otherwise skipped), files written by vrm_netcdf_write_color_nc4
  and the byte-built fixtures of tests/hdf5_builder.py are read BACK by the library and must give the values that went in.  (This is how
  the writer's one deviation from the library was found: an empty local-heap free list is stored as 1, not as the undefined address the
  published format names.)
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

import hdf5_builder as hb
from volume_renderer_amd import VrmError, netcdf

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hdf5")
NC_FILL_FLOAT = np.float32(9.9692099683868690e+36)

# file -> (z, y, x); what each one is: tools/make_hdf5_golden.c
LIB_FILES = {
    "lib_earliest_contiguous": (3, 5, 6), "lib_earliest_chunked": (5, 6, 7), "lib_v18_deflate_shuffle": (7, 10, 13),
    "lib_v18_fletcher_bigendian": (4, 4, 6), "lib_v18_compact": (2, 3, 4), "lib_latest_chunked": (6, 5, 9), "lib_latest_paged": (10, 10, 9),
    "lib_latest_implicit": (4, 5, 6), "lib_latest_single_chunk": (3, 4, 5), "lib_fill_value": (4, 6, 8), "lib_many_attributes": (3, 4, 5),
    "lib_many_variables": (3, 4, 5), "lib_hundred_variables": (2, 3, 4),
    "lib_f64_shuffle_deflate": (5, 4, 7), "lib_i16_bigendian": (3, 3, 5), "lib_u8_chunked": (4, 4, 4),
}


def golden_values(z, y, x):
    """element i (row-major over c, z, y, x) = ((i * 2654435761) mod 2^32 >> 8) / 2^24, exact in f32 (make_hdf5_golden.c: values())"""
    i = np.arange(4 * z * y * x, dtype=np.uint64)
    return ((((i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) >> np.uint64(8)).astype(np.float32) / np.float32(16777216.0)).reshape(4, z, y, x)


def golden_hash(z, y, x):
    return (np.arange(4 * z * y * x, dtype=np.uint64) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)


def expected(name):
    """what a reader must hand over for a golden file: f32, converted from the stored type as netcdf-c's nc_get_vara_float converts
    (the reference reads with get::<f32>, loaders/netcdf.rs:45)"""
    z, y, x = LIB_FILES[name]
    want = golden_values(z, y, x)
    if name == "lib_fill_value":
        want[:, 2:] = NC_FILL_FLOAT                      # only z < 2 was written: the other chunks do not exist
    elif name == "lib_f64_shuffle_deflate":
        want = (golden_hash(z, y, x).astype(np.float64) / 4294967296.0).astype(np.float32).reshape(4, z, y, x)   # (float)double
        assert (want.astype(np.float64) != golden_hash(z, y, x).reshape(4, z, y, x) / 4294967296.0).any()        # ... really rounds
    elif name == "lib_i16_bigendian":
        want = (golden_hash(z, y, x) >> np.uint64(16)).astype(np.uint16).view(np.int16).astype(np.float32).reshape(4, z, y, x)
        assert (want < 0).any()
    elif name == "lib_u8_chunked":
        want = (golden_hash(z, y, x) >> np.uint64(24)).astype(np.float32).reshape(4, z, y, x)
    return want


@pytest.mark.parametrize("name", sorted(LIB_FILES))
def test_reader_on_files_written_by_libhdf5(name):
    want = expected(name)
    got = netcdf.read_color(os.path.join(GOLDEN, name + ".nc"))
    assert got.shape == want.shape
    assert got.tobytes() == want.tobytes()


def test_every_golden_file_is_listed():
    assert sorted(f[:-3] for f in os.listdir(GOLDEN) if f.endswith(".nc")) == sorted(LIB_FILES)


# ---- the other direction: the library reads what this repository writes ---------------------------------------------------------------

H5DUMP = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
needs_h5dump = pytest.mark.skipif(not os.path.exists(H5DUMP), reason="no h5dump (libhdf5's own tool) on this machine")


def library_reads(path, tmp_path):
    """-> (`h5dump -H` text, the values of /color as the library decodes them, little-endian f32)"""
    head = subprocess.run([H5DUMP, "-H", path], capture_output=True, text=True, timeout=60)
    assert head.returncode == 0, head.stderr[-2000:]
    out = os.path.join(tmp_path, "color.bin")
    r = subprocess.run([H5DUMP, "-d", "/color", "-b", "LE", "-o", out, path], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr[-2000:]
    return head.stdout, np.fromfile(out, dtype="<f4")


@needs_h5dump
@pytest.mark.parametrize("shape", [(3, 5, 6), (1, 1, 1), (17, 4, 33)])
def test_library_reads_the_writer(tmp_path, shape):
    z, y, x = shape
    d = golden_values(z, y, x)
    p = os.path.join(tmp_path, "ours.nc")
    netcdf.write_color(p, d, format="netcdf4")
    head, got = library_reads(p, tmp_path)
    assert got.tobytes() == d.tobytes()
    assert 'DATASET "color"' in head and f"DATASPACE  SIMPLE {{ ( 4, {z}, {y}, {x} ) / ( 4, {z}, {y}, {x} ) }}" in head
    assert "H5T_IEEE_F32LE" in head
    for dim, n in zip("czyx", (4, z, y, x)):             # the dimension scales netcdf-c looks for
        assert f'DATASET "{dim}"' in head and f"DATASPACE  SIMPLE {{ ( {n} ) / ( {n} ) }}" in head
    for attribute in ("CLASS", "NAME", "_Netcdf4Dimid", "_Netcdf4Coordinates", "_NCProperties"):
        assert f'ATTRIBUTE "{attribute}"' in head
    # and the attribute VALUES as the library decodes them
    a = subprocess.run([H5DUMP, "-A", p], capture_output=True, text=True, timeout=60)
    assert a.returncode == 0
    assert '"DIMENSION_SCALE"' in a.stdout and "(0): 0, 1, 2, 3" in a.stdout
    assert "This is a netCDF dimension but not a netCDF variable." in a.stdout


@needs_h5dump
def test_library_reads_the_builders_fixtures(tmp_path):
    """the byte-built files of test_hdf5_cpu.py are valid HDF5 to the library itself, and hold what the tests say they hold
    (the fletcher32 fixture carries a dummy checksum, which the library would rightly refuse: not included)"""
    d = np.random.default_rng(11).random((4, 5, 6, 7), dtype=np.float32)
    cases = [("old", hb.old_style(d), d), ("v4", hb.single_chunk_v4(d, True), d), ("v4plain", hb.single_chunk_v4(d, False), d),
             ("classic_model", hb.netcdf4_classic_model(d), d)]
    for sb in (2, 3):
        for v1 in (False, True):
            blob, expect = hb.new_style(d, chunk=(2, 3, 4, 5), skip_chunk=5, unfiltered_chunk=9, v1_pipeline=v1, superblock=sb)
            cases.append((f"new{sb}{int(v1)}", blob, expect))
    for name, blob, expect in cases:
        p = os.path.join(tmp_path, name + ".nc")
        with open(p, "wb") as f:
            f.write(blob)
        _, got = library_reads(p, tmp_path)
        assert got.tobytes() == np.ascontiguousarray(expect, dtype="<f4").tobytes(), name
        assert netcdf.read_color(p).tobytes() == got.tobytes(), name


def test_unlimited_dimension_index_is_refused_cleanly(tmp_path):
    """version-4 layouts with an extensible-array or version-2 B-tree chunk index (datasets with unlimited dimensions) are not read:
    an I/O error that says so, not a wrong array"""
    d = golden_values(2, 2, 2)
    blob = bytearray(hb.single_chunk_v4(d, False))
    i = blob.index(bytes([4, 2, 0, 5, 4]))               # the layout message: version 4, chunked, flags, rank + 1, 4-byte dims
    assert blob[i + 5 + 5 * 4] == 1                      # the index type behind the five dimension sizes: 1 = single chunk
    blob[i + 5 + 5 * 4] = 4
    p = os.path.join(tmp_path, "ea.nc")
    with open(p, "wb") as f:
        f.write(bytes(blob))
    with pytest.raises(VrmError) as e:
        netcdf.read_color(p)
    assert e.value.status == -4 and "unlimited" in str(e.value)


def test_damaged_library_files_are_read_or_refused(tmp_path):
    """mutation fuzz over the library-written files (byte and bit flips, 8-byte fields overwritten, truncation): every outcome is an
    array or a VrmError, never a crash -- the paths added for these files (heaps, version-2 B-trees, fixed arrays, fill values) parse
    file-controlled sizes and addresses.  (12 000 mutations ran clean under the UndefinedBehaviorSanitizer build, tools/ubsan_host.sh.)"""
    rng = np.random.default_rng(5)
    files = sorted(LIB_FILES)
    p = os.path.join(tmp_path, "mut.nc")
    outcomes = {"read": 0, "refused": 0}
    for it in range(600):
        with open(os.path.join(GOLDEN, files[it % len(files)] + ".nc"), "rb") as f:
            b = bytearray(f.read())
        for _ in range(int(rng.integers(1, 6))):
            mode, i = int(rng.integers(0, 3)), int(rng.integers(0, len(b)))
            if mode == 0:
                b[i] = int(rng.integers(0, 256))
            elif mode == 1:
                b[i] ^= 1 << int(rng.integers(0, 8))
            else:
                j = int(rng.integers(0, len(b) - 8))
                b[j:j + 8] = bytes(rng.integers(0, 256, 8, dtype=np.uint8)) if rng.random() < 0.5 else b"\xff" * 8
        if rng.random() < 0.1:
            b = b[:int(rng.integers(8, len(b)))]
        with open(p, "wb") as f:
            f.write(bytes(b))
        try:
            netcdf.read_color(p)
            outcomes["read"] += 1
        except (VrmError, MemoryError):
            outcomes["refused"] += 1
    assert outcomes["read"] > 100 and outcomes["refused"] > 100, outcomes
