"""Behaviour of the C ABI on a real GPU: ingest paths, error codes, readback state machine, DAT loader,
the multi-view export ("Screenshot All") and its files."""
import ctypes
import json
import os

import numpy as np
import pytest

from conftest import have_gpu, image_metrics, sphere_cam
from volume_renderer_amd import synth

pytestmark = pytest.mark.gpu


def test_startup_grid_is_reference_default(rm, vr, orc):
    # RayMarcher::new: 3x3x3 of [0,0,0,255], attenuation 1, TF off (mod.rs:29, voxel.rs:13-16, grid.rs:197-198)
    u = rm.voxel_grid.uniform()
    assert list(u.dimensions) == [3, 3, 3, 3] and list(u.buffer) == [1.0] * 4
    assert list(u.box_size)[:3] == [1.0, 1.0, 1.0] and list(u.box_min)[:3] == [-0.5, -0.5, -0.5]
    vox = rm.voxel_grid.download_rgba8()
    assert vox.shape == (3, 3, 3, 4) and (vox == np.array([0, 0, 0, 255], np.uint8)).all()
    cam = sphere_cam(orc, 1, 3)
    img = vr.Screenshotter(rm, 48, 48)
    rm.draw(vr.uniform_from_array(cam), img)
    ref = orc.render(vox, cam, 48, 48)
    assert image_metrics(img.read(), ref["rgba8"])["max"] <= 1
    img.close()


def test_grid_uniform_matches_oracle(rm, orc):
    for dims in [(64, 32, 16), (5, 7, 3), (100, 100, 10)]:
        rm.voxel_grid.new(dims)
        rm.voxel_grid.attenuation = 2.5
        rm.voxel_grid.update_voxel_grid_buffer()
        u = rm.voxel_grid.uniform()
        d, bmin, bsz, buf = orc.grid_uniform(dims, 2.5)
        assert list(u.dimensions) == list(d)
        assert np.array_equal(np.array(u.box_min, np.float32), bmin) and np.array_equal(np.array(u.box_size, np.float32), bsz)
        assert np.array_equal(np.array(u.buffer, np.float32), buf)


def test_resize_resets_attenuation_and_transfer_function(rm, vr):
    from volume_renderer_amd import _ffi
    g = rm.voxel_grid
    g.attenuation = 9.0
    g.update_voxel_grid_buffer()
    g.set_transfer_function_active(True)
    g.update_transfer_function_buffer()
    _ffi.check(_ffi.lib().vrm_grid_resize(rm._h, 4, 4, 4))       # VoxelGrid::new as the loaders call it
    assert list(g.uniform().buffer) == [1.0] * 4                  # grid.rs:100,197
    vox = g.download_rgba8()
    assert (vox == np.array([0, 0, 0, 255], np.uint8)).all()


def test_upload_download_roundtrip_and_f32_quantisation(rm, orc):
    g = rm.voxel_grid
    vox = synth.random_rgba8((9, 6, 11), 4)
    g.upload_rgba8(vox)
    assert np.array_equal(g.download_rgba8(), vox)
    rng = np.random.default_rng(1)
    data = rng.uniform(-0.2, 1.3, size=(4, 7, 5, 6)).astype(np.float32)
    data[2, 3, 1, 2] = np.nan
    data[:, 0, 0, 0] = [0.999, 1.0, 254.5 / 255.0, 0.5]
    g.upload_f32(data)
    assert np.array_equal(g.download_rgba8(), orc.quantize_netcdf(data))          # netcdf.rs:54-57
    scal = rng.uniform(0, 1, size=(7, 5, 6)).astype(np.float32)
    g.upload_f32(scal)
    want = orc.quantize_netcdf(np.stack([scal] * 4))
    assert np.array_equal(g.download_rgba8(), want)


def test_upload_scalar_uchar_and_big_endian_ushort(rm, orc):
    g = rm.voxel_grid
    dims = (8, 4, 8)
    raw = np.arange(256, dtype=np.uint8)
    g.upload_scalar(raw, dims, 1)
    assert np.array_equal(g.download_rgba8().reshape(-1, 4), orc.quantize_dat(raw, 1))
    vals = np.random.default_rng(0).integers(0, 65536, size=256).astype(np.uint16)
    vals[:3] = [65535, 65534, 0]
    raw16 = vals.astype(">u2").view(np.uint8)
    g.upload_scalar(raw16, dims, 2)
    assert np.array_equal(g.download_rgba8().reshape(-1, 4), orc.quantize_dat(raw16, 2))   # dat.rs:64-74


def test_size_mismatch_and_bad_arguments_return_errors(rm, vr):
    from volume_renderer_amd import _ffi
    L = _ffi.lib()
    rm.voxel_grid.new((4, 4, 4))
    buf = np.zeros(10, np.uint8)
    assert L.vrm_grid_upload_rgba8(rm._h, buf.ctypes.data, buf.nbytes) == _ffi.VRM_ERR_INVALID
    assert b"expected 256 bytes" in L.vrm_last_error()
    assert L.vrm_grid_resize(rm._h, 0, 4, 4) == _ffi.VRM_ERR_INVALID
    f = np.zeros(64, np.float32)
    assert L.vrm_grid_upload_f32(rm._h, f.ctypes.data, 3) == _ffi.VRM_ERR_INVALID
    h = ctypes.c_void_p()
    assert L.vrm_image_create(rm._h, 0, 10, ctypes.byref(h)) == _ffi.VRM_ERR_INVALID
    cfg = _ffi.Config(99, 64, 64)
    assert L.vrm_create(ctypes.byref(cfg), ctypes.byref(h)) == _ffi.VRM_ERR_INVALID
    assert L.vrm_set_option(rm._h, 77, 1) == _ffi.VRM_ERR_INVALID
    with pytest.raises(vr.VrmError):
        rm.load_dat("/nonexistent/volume.dat")


def test_image_state_machine(rm, vr, orc):
    from volume_renderer_amd import _ffi
    L = _ffi.lib()
    img = vr.Screenshotter(rm, 40, 24)
    # never drawn: transparent clear colour (mod.rs:124)
    assert not img.read().any()
    p = ctypes.POINTER(ctypes.c_uint8)()
    pitch = ctypes.c_size_t()
    assert L.vrm_image_unmap(img._h) == _ffi.VRM_ERR_STATE
    assert L.vrm_image_map(img._h, ctypes.byref(p), ctypes.byref(pitch)) == 0
    assert pitch.value == 256 and pitch.value >= 40 * 4           # ceil(4w/256)*256 (screenshot.rs:94-96)
    assert L.vrm_image_map(img._h, ctypes.byref(p), ctypes.byref(pitch)) == _ffi.VRM_ERR_STATE
    cam = vr.uniform_from_array(sphere_cam(orc, 1, 3))
    assert L.vrm_draw(rm._h, ctypes.byref(cam), img._h) == _ffi.VRM_ERR_STATE   # mapped target
    assert L.vrm_image_unmap(img._h) == 0
    rm.draw(cam, img)
    img.screenshot()
    a = img.read()
    assert a[..., 3].any()
    assert np.array_equal(a, img.read())
    img.close()
    wide = vr.Screenshotter(rm, 100, 3)
    assert L.vrm_image_map(wide._h, ctypes.byref(p), ctypes.byref(pitch)) == 0 and pitch.value == 512
    L.vrm_image_unmap(wide._h)
    wide.close()


def test_two_contexts_are_independent(vr, orc, rm):
    other = vr.RayMarcher(0, 64, 64)
    rm.voxel_grid.upload_rgba8(synth.solid_rgba8((4, 4, 4), 255))
    other.voxel_grid.upload_rgba8(synth.solid_rgba8((4, 4, 4), 0))
    cam = vr.uniform_from_array(sphere_cam(orc, 1, 3))
    a, b = vr.Screenshotter(rm, 32, 32), vr.Screenshotter(other, 32, 32)
    rm.draw(cam, a)
    other.draw(cam, b)
    assert a.read()[..., 3].any() and not b.read().any()
    a.close(); b.close(); other.close()


def test_dat_loader(rm, vr, orc, tmp_path):
    dims = (12, 10, 8)
    raw = np.random.default_rng(2).integers(0, 256, size=dims[0] * dims[1] * dims[2]).astype(np.uint8)
    (tmp_path / "vol.raw").write_bytes(raw.tobytes())
    (tmp_path / "vol.dat").write_text("ObjectFileName: vol.raw\nResolution: 12 10 8\nSliceThickness: 1 1 1\nFormat: UCHAR\n")
    rm.load_dat(str(tmp_path / "vol.dat"))
    assert rm.voxel_grid.dimensions == dims
    got = rm.voxel_grid.download_rgba8()
    assert np.array_equal(got.reshape(-1, 4), orc.quantize_dat(raw, 1))
    # x fastest, then y, then z (dat.rs:58-60)
    assert got[3, 2, 5, 0] == raw[5 + 12 * (2 + 10 * 3)]
    vals = np.random.default_rng(3).integers(0, 65536, size=raw.size).astype(">u2")
    (tmp_path / "v16.raw").write_bytes(vals.tobytes())
    (tmp_path / "v16.dat").write_text("ObjectFileName: v16.raw\nResolution: 12 10 8\nFormat: USHORT\n")
    rm.load_dat(str(tmp_path / "v16.dat"))
    assert np.array_equal(rm.voxel_grid.download_rgba8().reshape(-1, 4), orc.quantize_dat(vals.view(np.uint8), 2))
    (tmp_path / "short.dat").write_text("ObjectFileName: vol.raw\nResolution: 64 64 64\nFormat: UCHAR\n")
    with pytest.raises(vr.VrmError):
        rm.load_dat(str(tmp_path / "short.dat"))


def test_export_sphere_files_and_cameras_json(vr, orc, tmp_path):
    from PIL import Image
    W, H = 72, 48
    rm = vr.RayMarcher(0, W, H)
    vox = synth.quantize_rgba8(synth.blobs_rgba_f32((24, 24, 24)))
    rm.voxel_grid.upload_rgba8(vox)
    rm.voxel_grid.attenuation = 6.0
    rm.voxel_grid.update_voxel_grid_buffer()
    out = tmp_path / "screenshots"
    rm.export_sphere(4, 5, 2.5, str(out))
    views = orc.sphere_views(4, 5)
    names = sorted(os.listdir(out))
    assert names == sorted([f"{ix}-{iy}.png" for ix, iy in views] + ["cameras.json"])       # state.rs:393
    js = json.loads((out / "cameras.json").read_text())
    assert js["look_at"] == [0.0, 0.0, 0.0] and len(js["positions"]) == len(views)
    for (ix, iy), entry in zip(views, js["positions"]):                                        # ssm.rs:56-64
        assert entry["img"] == f"{ix}-{iy}.png" and entry["fovy"] == 45.0
        pos = orc.sphere_position(ix, iy, 4, 5, 2.5)
        cam, basis = orc.camera_uniform(pos, aspect=W / H, with_basis=True)
        assert np.array_equal(np.array(entry["position"], np.float32), pos)                   # shortest round-trip floats
        assert np.array_equal(np.array(entry["right"], np.float32), basis[0])
        assert np.array_equal(np.array(entry["up"], np.float32), basis[1])
        assert np.array_equal(np.array(entry["front"], np.float32), basis[2])
        png = np.array(Image.open(out / entry["img"]))
        assert png.shape == (H, W, 4) and png.dtype == np.uint8
        ref = orc.render(vox, cam, W, H, attenuation=6.0)
        m = image_metrics(png, ref["rgba8"])
        assert m["max"] <= 1 and m["rmse_norm"] < 1e-3
    # view partitioning: two parts together write the same files, part 1 writes no cameras.json
    o0, o1 = tmp_path / "p0", tmp_path / "p1"
    rm.export_sphere(4, 5, 2.5, str(o0), part=0, n_parts=2)
    rm.export_sphere(4, 5, 2.5, str(o1), part=1, n_parts=2)
    n0, n1 = set(os.listdir(o0)), set(os.listdir(o1))
    assert "cameras.json" in n0 and "cameras.json" not in n1 and not (n0 & n1)
    assert n0 | n1 == set(names)
    for n in n1:
        assert np.array_equal(np.array(Image.open(o1 / n)), np.array(Image.open(out / n)))
    rm.close()


def test_save_png_decodes_to_the_mapped_pixels(rm, vr, orc, tmp_path):
    from PIL import Image
    rm.voxel_grid.upload_rgba8(synth.random_rgba8((6, 6, 6), 9))
    rm.voxel_grid.attenuation = 30.0
    rm.voxel_grid.update_voxel_grid_buffer()
    img = vr.Screenshotter(rm, 70, 33)          # width*4 = 280 -> pitch 512: padding must be stripped (screenshot.rs:71-72)
    rm.draw(vr.uniform_from_array(sphere_cam(orc, 5, 2, aspect=70 / 33)), img)
    img.save_screenshot_to_disk(str(tmp_path / "a.png"))
    assert np.array_equal(np.array(Image.open(tmp_path / "a.png")), img.read())
    img.close()


def test_reference_sample_count_matches_oracle(rm, vr, orc):
    dims = (20, 9, 33)
    vox = synth.random_rgba8(dims, 1)
    rm.voxel_grid.upload_rgba8(vox)
    views = [(0, 1), (3, 4), (5, 6)]
    cams = [sphere_cam(orc, ix, iy, aspect=50 / 37) for ix, iy in views]
    want = sum(orc.render(vox, c, 50, 37, want_f32=False)["samples"] for c in cams)
    got = rm.count_reference_samples([vr.uniform_from_array(c) for c in cams], 50, 37)
    assert got == want


def test_profiling_counts_launches(rm, vr, orc):
    rm.set_option(vr.OPT_PROFILE, 1)
    rm.profile_read()
    img = vr.Screenshotter(rm, 64, 64)
    cam = vr.uniform_from_array(sphere_cam(orc, 1, 3))
    for _ in range(3):
        rm.draw(cam, img)
    ms, n = rm.profile_read()
    assert n == 3 and ms > 0
    assert rm.profile_read() == (0.0, 0)
    img.close()


def test_netcdf_load_and_write(rm1, vr, orc, tmp_path):
    """open_voxel_grid / write_voxel_grid (loaders/netcdf.rs) on classic containers, against the oracle's quantisation."""
    scipy_io = pytest.importorskip("scipy.io")
    from volume_renderer_amd import netcdf
    rng = np.random.default_rng(4)
    data = rng.random((4, 7, 9, 11), dtype=np.float32) * np.float32(1.2) - np.float32(0.1)   # some values outside [0,1]: saturate
    data[2, 3, 4, 5] = np.nan                                                                   # Rust `as u8`: NaN -> 0
    p = os.path.join(tmp_path, "in.nc")
    f = scipy_io.netcdf_file(p, "w", version=2)
    for n, k in zip("czyx", data.shape):
        f.createDimension(n, k)
    f.createVariable("color", "f", ("c", "z", "y", "x"))[:] = data
    f.close()
    g = rm1.voxel_grid
    g.attenuation = 7.0
    g.update_voxel_grid_buffer()
    rm1.load_netcdf(p)
    assert g.dimensions == (11, 9, 7) and list(g.uniform().buffer) == [1.0] * 4      # VoxelGrid::new resets attenuation (netcdf.rs:47)
    vox = g.download_rgba8()
    assert np.array_equal(vox, orc.quantize_netcdf(data))
    out = os.path.join(tmp_path, "out.nc")
    rm1.write_netcdf(out)
    back = netcdf.read_color(out)
    want = (np.moveaxis(vox, -1, 0).astype(np.float32) / np.float32(255.0))         # voxel as f32 / 255.0 (netcdf.rs:23)
    assert np.array_equal(back, want)
    fr = scipy_io.netcdf_file(out, "r", mmap=False)
    assert np.array_equal(np.asarray(fr.variables["color"].data, dtype=np.float32), want)
    fr.close()
    rm1.load_netcdf(out)                                                            # write -> load is the identity on the voxels
    assert np.array_equal(g.download_rgba8(), vox)
    out4 = os.path.join(tmp_path, "out4.nc")                                         # the container netcdf::create itself makes (HDF5)
    rm1.write_netcdf(out4, format="netcdf4")
    assert open(out4, "rb").read(4) == b"\x89HDF" and np.array_equal(netcdf.read_color(out4), want)
    g.new((3, 3, 3))
    rm1.load_netcdf(out4)
    assert g.dimensions == (11, 9, 7) and np.array_equal(g.download_rgba8(), vox)
    with pytest.raises(vr.VrmError):
        rm1.load_netcdf(os.path.join(tmp_path, "missing.nc"))


def test_resize_that_cannot_be_allocated_leaves_a_usable_context(vr, orc):
    """A grid that does not fit the GPU: the call fails with the HIP error, the context falls back to the start-up
    3x3x3 grid (not to dangling or null buffers), draws keep working and a later resize succeeds."""
    from volume_renderer_amd import _ffi
    r = vr.RayMarcher(0, 32, 32)
    try:
        r.voxel_grid.new((20, 20, 20))
        L = _ffi.lib()
        rc = L.vrm_grid_resize(r._h, 8192, 8192, 8192)            # 2 TiB of RGBA8 alone
        assert rc == _ffi.VRM_ERR_HIP
        msg = L.vrm_last_error().decode()
        assert "8192 x 8192 x 8192" in msg and "3 x 3 x 3" in msg
        u = r.voxel_grid.uniform()
        assert list(u.dimensions)[:3] == [3, 3, 3]
        cam = sphere_cam(orc, 1, 3)
        img = vr.Screenshotter(r, 32, 32)
        r.draw(vr.uniform_from_array(cam), img)
        vox = np.zeros((3, 3, 3, 4), np.uint8)
        vox[..., 3] = 255
        assert image_metrics(img.read(), orc.render(vox, cam, 32, 32)["rgba8"])["max"] <= 1
        img.close()
        assert L.vrm_grid_resize(r._h, 0, 4, 4) == _ffi.VRM_ERR_INVALID
        _ffi.check(L.vrm_grid_resize(r._h, 12, 12, 12))
        assert list(r.voxel_grid.uniform().dimensions)[:3] == [12, 12, 12]
    finally:
        r.close()


def test_draw_waits_for_a_pending_readback_of_its_target(vr, orc):
    """draw A -> readback_async -> draw B into the same image -> map: the second draw must not overwrite the device image
    under the copy that is still reading it (write-after-read on img->dev).  The map after a re-draw returns B's pixels,
    and what an earlier map returned was A's -- never a mix of the two."""
    if not have_gpu():
        pytest.skip("no GPU")
    R = 2048                                                      # 16 MiB per image: the copy takes a while
    r = vr.RayMarcher(0, R, R)
    r.voxel_grid.upload_rgba8(synth.quantize_rgba8(synth.blobs_rgba_f32((64, 64, 64))))
    r.voxel_grid.attenuation = 30.0
    r.voxel_grid.update_voxel_grid_buffer()
    cam_a = vr.uniform_from_array(sphere_cam(orc, 1, 3))
    cam_b = vr.uniform_from_array(sphere_cam(orc, 5, 6))
    ref, img = vr.Screenshotter(r, R, R), vr.Screenshotter(r, R, R)
    r.draw(cam_a, ref)
    want_a = ref.read()
    r.draw(cam_b, ref)
    want_b = ref.read()
    assert not np.array_equal(want_a, want_b)
    for _ in range(4):
        r.draw(cam_a, img)
        img.screenshot()                                          # copy of A in flight on the copy stream
        r.draw(cam_b, img)                                        # must wait for that copy
        assert np.array_equal(img.read(), want_b)                 # latest draw wins, whole image
        r.draw(cam_a, img)
        img.screenshot()
        a_pixels = img.read()                                     # mapped before the next draw: A
        r.draw(cam_b, img)
        assert np.array_equal(a_pixels, want_a)
    img.close()
    ref.close()
    r.close()


def test_image_outlives_its_context(vr):
    """vrm_destroy before vrm_image_destroy (Python: RayMarcher.close() while a Screenshotter is still referenced): the
    image is orphaned, refuses every call but destroy, and destroy does not touch the freed context."""
    if not have_gpu():
        pytest.skip("no GPU")
    r = vr.RayMarcher(0, 64, 64)
    img = vr.Screenshotter(r, 64, 64)
    r.draw(vr.Camera(1.0).uniform(), img)
    r.close()
    with pytest.raises(vr.VrmError) as e:
        img.read()
    assert e.value.status == -5 and "context was destroyed" in str(e.value)
    with pytest.raises(vr.VrmError):
        img.screenshot()
    img.close()


def test_export_error_path_joins_and_reports(vr, tmp_path):
    """A failing PNG write in the middle of the export (the directory vanished) must come back as VRM_ERR_IO with every
    encoder thread joined and every image unmapped -- not terminate the process."""
    if not have_gpu():
        pytest.skip("no GPU")
    r = vr.RayMarcher(0, 64, 64)
    r.voxel_grid.upload_rgba8(synth.quantize_rgba8(synth.blobs_rgba_f32((16, 16, 16))))
    missing = tmp_path / "not_there"
    from volume_renderer_amd import _ffi
    org = np.zeros(3, dtype=np.float32)
    rc = _ffi.lib().vrm_export_sphere(r._h, 4, 4, ctypes.c_float(2.5), org.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                      os.fsencode(str(missing)), 1, 2)     # part 1: no cameras.json, straight to the PNGs
    assert rc == -4 and "cannot open" in _ffi.lib().vrm_last_error().decode()
    good = tmp_path / "out"
    r.export_sphere(4, 4, 2.5, str(good))                          # the context is still usable
    assert len(list(good.glob("*.png"))) == 12
    r.close()


def test_netcdf4_hdf5_load(rm1, vr, orc, tmp_path):
    """open_voxel_grid on a NetCDF-4 = HDF5 container (what the reference's netcdf crate and diffdvr.py write): both dialects of
    tests/hdf5_builder.py through vrm_load_netcdf, quantised on the GPU like the classic path.  (The container is pinned against libhdf5 in test_hdf5_golden_cpu.py.)"""
    import hdf5_builder as hb
    rng = np.random.default_rng(11)
    data = (rng.random((4, 6, 10, 9), dtype=np.float32) * 1.2 - 0.1).astype(np.float32)    # values below 0 and above 1 as well
    data[2, 3, 4, 5] = np.nan
    for name, blob in (("old.nc", hb.old_style(data)),
                       ("new.nc", hb.new_style(data, chunk=(2, 4, 4, 4), skip_chunk=None, unfiltered_chunk=3)[0])):
        p = os.path.join(tmp_path, name)
        open(p, "wb").write(blob)
        rm1.voxel_grid.attenuation = 7.0
        rm1.voxel_grid.update_voxel_grid_buffer()
        rm1.load_netcdf(p)
        g = rm1.voxel_grid
        assert g.dimensions == (9, 10, 6) and list(g.uniform().buffer) == [1.0] * 4
        assert np.array_equal(g.download_rgba8(), orc.quantize_netcdf(data))
    # ... and files the HDF5 library itself wrote (tests/golden/hdf5, tools/make_hdf5_golden.c): the dialect netcdf-c's defaults give,
    # a chunked + shuffled + deflated one, and one whose never-written chunks read as netCDF's fill value (9.97e36 -> 255 as u8)
    from test_hdf5_golden_cpu import GOLDEN, LIB_FILES, NC_FILL_FLOAT, golden_values
    for name in ("lib_earliest_contiguous", "lib_v18_deflate_shuffle", "lib_fill_value"):
        z, y, x = LIB_FILES[name]
        want = golden_values(z, y, x)
        if name == "lib_fill_value":
            want[:, 2:] = NC_FILL_FLOAT
        rm1.load_netcdf(os.path.join(GOLDEN, name + ".nc"))
        assert rm1.voxel_grid.dimensions == (x, y, z)
        assert np.array_equal(rm1.voxel_grid.download_rgba8(), orc.quantize_netcdf(want)), name


def test_degenerate_camera_uniforms_come_back(vr, orc):
    """Uniforms no host sends on purpose -- NaN, infinities, zeros, 1e30 -- give rays whose parameters are not numbers; the reference
    marches whatever it gets, and every march variant here must at least END (round 3: the LDS march hung on NaN rays; the whole
    matrix of variants, builds and volumes is tools/garbage_cameras.py).  Afterwards the context still renders a sane view."""
    if not have_gpu():
        pytest.skip("no GPU")
    W, H = 40, 24
    good = sphere_cam(orc, 2, 3, aspect=W / H)
    bad = {"nan": np.full(36, np.nan, np.float32), "inf": np.full(36, np.inf, np.float32), "zeros": np.zeros(36, np.float32),
           "huge": (good * np.float32(1e30)).astype(np.float32)}
    for scalar in (False, True):
        r = vr.RayMarcher(0, W, H)
        if scalar:
            r.voxel_grid.upload_scalar(synth.blobs_scalar_u8((40, 33, 21)).reshape(-1), (40, 33, 21), 1)
        else:
            r.voxel_grid.upload_rgba8(synth.random_rgba8((40, 33, 21), 3))
        img = vr.Screenshotter(r, W, H)
        r.draw(vr.uniform_from_array(good), img)
        want = img.read().copy()
        for c in bad.values():
            for k in (1, 4):
                for diag in (0, 1):
                    r.set_option(vr.OPT_KERNEL, k)
                    r.set_option(vr.OPT_DIAGNOSTICS, diag)
                    r.draw(vr.uniform_from_array(c), img)
                    img.read()
        r.set_option(vr.OPT_KERNEL, 1)
        r.set_option(vr.OPT_DIAGNOSTICS, 0)
        r.draw(vr.uniform_from_array(good), img)
        assert np.array_equal(img.read(), want)
        img.close()
        r.close()


def test_library_broadcast_single_rank(vr, orc):
    """vrm_comm_* / vrm_grid_broadcast with RCCL bound inside the library.  One rank is all a one-GPU box can run (RCCL
    refuses two ranks on one device): the id, the communicator, a broadcast in which this rank is the root, the
    state checks; the image is unchanged.  Between different GPUs the call is unmeasured."""
    if not have_gpu():
        pytest.skip("no GPU")
    r = vr.RayMarcher(0, 64, 64)
    vox = synth.quantize_rgba8(synth.blobs_rgba_f32((24, 20, 28)))
    r.voxel_grid.upload_rgba8(vox)
    cam = vr.uniform_from_array(sphere_cam(orc, 2, 3))
    img = vr.Screenshotter(r, 64, 64)
    r.draw(cam, img)
    before = img.read().copy()
    with pytest.raises(vr.VrmError, match="no communicator"):
        r.grid_broadcast(0)
    cid = vr.RayMarcher.comm_unique_id()
    assert len(cid) == 128 and any(cid)
    r.comm_init(cid, 0, 1)
    with pytest.raises(vr.VrmError, match="already has a communicator"):
        r.comm_init(cid, 0, 1)
    with pytest.raises(vr.VrmError, match="root"):
        r.grid_broadcast(1)
    r.grid_broadcast(0)
    r.draw(cam, img)
    assert np.array_equal(img.read(), before)
    assert np.array_equal(r.voxel_grid.download_rgba8(), vox)
    r.comm_destroy()
    r.comm_destroy()          # idempotent
    img.close()
    r.close()


def test_scalar_volume_is_held_compact_and_expands_on_demand(vr, orc):
    """a scalar volume keeps what the 1-channel march reads -- packed alpha bricks + pyramid, about 1 B/voxel --
    and drops the 8 B/voxel of RGBA copies; anything that needs them (download, RGBA-staging kernel options, the compare) gets
    them back exactly.  Dims off every brick boundary, so the edge-staging path (which used to read the RGBA copy) is on."""
    dims = (45, 38, 51)                                              # x, y, z
    n = dims[0] * dims[1] * dims[2]
    dens = synth.blobs_scalar_u8(dims)                              # (Z, Y, X)
    R = 96
    cams = [vr.uniform_from_array(sphere_cam(orc, ix, iy)) for ix, iy in ((1, 3), (5, 6), (0, 1))]

    def render(r):
        imgs = [vr.Screenshotter(r, R, R) for _ in cams]
        r.draw_batch(cams, imgs)
        out = [im.read() for im in imgs]
        for im in imgs:
            im.close()
        return out

    full = vr.RayMarcher(0, R, R)
    full.set_option(vr.OPT_COMPACT_SCALAR, 0)
    full.voxel_grid.new(dims)
    full.voxel_grid.upload_scalar(dens.reshape(-1), dims, 1)
    assert full.voxel_grid.resident_bytes() >= 9 * n
    want = render(full)
    assert any(w[..., 3].any() for w in want)

    r = vr.RayMarcher(0, R, R)
    r.voxel_grid.new(dims)
    # a grid that was resized and not yet filled holds its default voxels [0,0,0,255] as a promise: the 8 B/voxel of RGBA
    # storage come when something looks at it ...
    assert r.voxel_grid.resident_bytes() < 2.2 * n
    default = render(r)
    dvox = np.zeros(dims[::-1] + (4,), np.uint8)
    dvox[..., 3] = 255
    assert image_metrics(default[0], orc.render(dvox, sphere_cam(orc, 1, 3), R, R, want_f32=False)["rgba8"])["max"] <= 1
    assert default[0][..., 3].max() > 100 and r.voxel_grid.resident_bytes() >= 9 * n
    assert np.array_equal(r.voxel_grid.download_rgba8(), dvox)
    # ... and a scalar upload (direct ingest: packed bricks and pyramid straight from the raw samples) releases them again
    r.voxel_grid.upload_scalar(dens.reshape(-1), dims, 1)
    compact = r.voxel_grid.resident_bytes()
    assert compact < 2.2 * n, (compact, n)                           # packed bricks (padded to 16 x 4 x 2) + pyramid
    got = render(r)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)                                  # the same march from the same bytes
    assert r.voxel_grid.resident_bytes() == compact                  # the default draw needed nothing else
    ref = orc.render(np.repeat(dens[..., None], 4, axis=3), sphere_cam(orc, 1, 3), R, R, want_f32=False)
    assert image_metrics(got[0], ref["rgba8"])["max"] <= 1
    # the transfer function on the compact volume (still the packed march)
    r.voxel_grid.set_transfer_function_active(True)
    r.voxel_grid.update_transfer_function_buffer()
    full.voxel_grid.set_transfer_function_active(True)
    full.voxel_grid.update_transfer_function_buffer()
    for a, b in zip(render(r), render(full)):
        assert np.array_equal(a, b)
    assert r.voxel_grid.resident_bytes() == compact
    r.voxel_grid.set_transfer_function_active(False)
    r.voxel_grid.update_transfer_function_buffer()
    # the allocation-free draw may not expand: a clear error when the options need the RGBA copies
    imgs = [vr.Screenshotter(r, R, R) for _ in cams]
    r.draw_batch_static(cams, imgs)                                  # default options: fine
    assert np.array_equal(imgs[0].read(), want[0])
    r.set_option(vr.OPT_PACKED_ALPHA, 0)
    with pytest.raises(vr.VrmError) as e:
        r.draw_batch_static(cams, imgs)
    assert "compact" in str(e.value)
    for im in imgs:
        im.close()
    # RGBA-staged 1-channel march: expands, same image to rounding
    staged = render(r)
    assert r.voxel_grid.resident_bytes() >= 9 * n
    for a, b in zip(staged, want):
        assert image_metrics(a, b)["max"] <= 1
    r.set_option(vr.OPT_PACKED_ALPHA, 1)
    # a new upload compacts again; download gives [b,b,b,b] back exactly (and expands); the baseline kernel works after it
    r.voxel_grid.upload_scalar(dens.reshape(-1), dims, 1)
    assert r.voxel_grid.resident_bytes() == compact
    vox = r.voxel_grid.download_rgba8()
    assert np.array_equal(vox, np.repeat(dens[..., None], 4, axis=3))
    r.voxel_grid.upload_scalar(dens.reshape(-1), dims, 1)
    r.set_option(vr.OPT_KERNEL, 0)
    for a, b in zip(render(r), want):
        assert image_metrics(a, b)["max"] <= 1
    r.set_option(vr.OPT_KERNEL, 1)
    # an RGBA volume is never compacted
    r.voxel_grid.upload_rgba8(synth.quantize_rgba8(synth.blobs_rgba_f32(dims)))
    assert r.voxel_grid.resident_bytes() >= 9 * n
    full.close()
    r.close()


def test_comm_broadcast_two_ranks(tmp_path):
    """the receiver branch of vrm_grid_broadcast (ncclBroadcast into a fresh grid, then bricks / packed alpha /
    pyramid / scalar detection of the received voxels) between two DIFFERENT GPUs, one process per GPU.  Skipped on a one-GPU
    box (all this pool lends): there the multi-rank path stays unmeasured on hardware, as DESIGN.md section 5 says."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    idf = str(tmp_path / "comm.id")
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tools", "rccl_broadcast_rank.py"), str(r), "2", idf, str(tmp_path / f"out{r}")],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    h = [(tmp_path / f"out{r}").read_text() for r in range(2)]
    assert h[0] == h[1] and len(h[0]) == 64                       # same voxels downloaded, same image rendered on both ranks


def test_direct_scalar_ingest_equals_the_long_way_round(vr, orc):
    """vrm_grid_upload_scalar with the default options builds the packed alpha bricks and the occupancy pyramid straight from
    the raw samples (UCHAR and big-endian USHORT, dat.rs:62-74) and never allocates the RGBA form: the voxels that come back
    (vrm_grid_download_rgba8 expands the bricks) and the images must equal those of the upload that quantises to RGBA8 first
    (VRM_OPT_COMPACT_SCALAR = 0), and the oracle's quantisation.  Dims off every brick boundary; a fresh context whose grid was
    never materialised, and a context that held another volume before."""
    R = 80
    cam = sphere_cam(orc, 2, 5)
    for dims, bps in [((45, 38, 51), 1), ((33, 17, 9), 2), ((16, 4, 2), 2), ((1, 1, 1), 1)]:
        n = dims[0] * dims[1] * dims[2]
        rng = np.random.default_rng(n + bps)
        if bps == 1:
            raw = synth.blobs_scalar_u8(dims).reshape(-1) if min(dims) >= 8 else rng.integers(0, 256, n, dtype=np.uint8)
            want = raw.reshape(dims[::-1])
        else:
            v16 = rng.integers(0, 65536, n, dtype=np.uint16)
            v16[: n // 3] = 0                                         # empty space for the pyramid
            raw = v16.astype(">u2").view(np.uint8)
            want = ((v16.astype(np.float32) / np.float32(65535.0)) * np.float32(255.0)).astype(np.uint8).reshape(dims[::-1])
        out = {}
        for compact in (1, 0):
            r = vr.RayMarcher(0, R, R)
            r.set_option(vr.OPT_COMPACT_SCALAR, compact)
            if compact:                                               # a context that held something else before
                r.voxel_grid.new((7, 5, 3))
                r.voxel_grid.upload_rgba8(synth.random_rgba8((7, 5, 3), 1))
            r.voxel_grid.new(dims)
            r.voxel_grid.upload_scalar(raw, dims, bps)
            if compact:
                assert r.voxel_grid.resident_bytes() < 2.2 * n + 4096
            r.voxel_grid.attenuation = 6.0
            r.voxel_grid.update_voxel_grid_buffer()
            img = vr.Screenshotter(r, R, R)
            outs = []
            for skip in (1, 0):
                r.set_option(vr.OPT_SKIP_EMPTY, skip)
                r.draw(vr.uniform_from_array(cam), img)
                outs.append(img.read())
            assert np.array_equal(outs[0], outs[1])                   # the pyramid built from the raw samples is exact
            vox = r.voxel_grid.download_rgba8()
            img.close()
            r.close()
            out[compact] = (outs[0], vox)
        assert np.array_equal(out[1][1], out[0][1]) and np.array_equal(out[1][0], out[0][0])
        assert np.array_equal(out[1][1], np.repeat(want[..., None], 4, axis=3))
