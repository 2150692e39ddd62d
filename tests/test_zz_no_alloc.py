"""vrm_draw_batch_static (include/vrm.h): the allocation-free form of vrm_draw_batch -- the march kernel's descriptor
scratch lives in a file-scope device array instead of the ring the library allocated.  Checked through the C ABI against
vrm_draw_batch itself, bit for bit, on two calls in a row, the second with a smaller input (fewer views), so that stale
descriptors of the first call would show; then the size limit and the error behaviour."""
import ctypes

import numpy as np
import pytest

from conftest import have_gpu, sphere_cam
from volume_renderer_amd import _ffi, synth

pytestmark = pytest.mark.gpu


def _images(vr, r, n, R):
    return [vr.Screenshotter(r, R, R) for _ in range(n)]


def test_static_draw_matches_draw_batch_twice_in_a_row(vr, orc):
    if not have_gpu():
        pytest.skip("no GPU")
    R = 192
    r = vr.RayMarcher(0, R, R)
    r.voxel_grid.upload_rgba8(synth.quantize_rgba8(synth.blobs_rgba_f32((48, 40, 56))))
    r.voxel_grid.attenuation = 25.0
    r.voxel_grid.update_voxel_grid_buffer()
    views = orc.sphere_views(4, 4)
    cams = [vr.uniform_from_array(sphere_cam(orc, ix, iy, xdiv=4, ydiv=4)) for ix, iy in views]
    assert len(cams) == 12
    want = _images(vr, r, 12, R)
    r.draw_batch(cams, want)
    want_px = [im.read() for im in want]
    assert all(w[..., 3].any() for w in want_px)
    got = _images(vr, r, 12, R)
    # call 1: 7 views; call 2, straight after it, 3 OTHER views (a smaller input reusing the same scratch)
    r.draw_batch_static(cams[:7], got[:7])
    r.draw_batch_static(cams[9:12], got[9:12])
    for i in list(range(7)) + [9, 10, 11]:
        assert np.array_equal(got[i].read(), want_px[i]), i
    # and once more into the SAME targets with different cameras: descriptors of the earlier calls must not leak through
    r.draw_batch_static(cams[7:9], got[:2])
    assert np.array_equal(got[0].read(), want_px[7]) and np.array_equal(got[1].read(), want_px[8])
    # interleaved with the allocating entry point on the same context
    r.draw_batch(cams[:2], got[2:4])
    r.draw_batch_static(cams[2:4], got[:2])
    assert np.array_equal(got[2].read(), want_px[0]) and np.array_equal(got[0].read(), want_px[2])
    # every variant of the march takes its descriptors from the same place: scalar volume (packed 1-channel march) and TF
    r.voxel_grid.upload_scalar(synth.blobs_scalar_u8((48, 40, 56)).reshape(-1), (48, 40, 56), 1)
    r.draw_batch(cams[:3], want[:3])
    r.draw_batch_static(cams[:3], got[:3])
    for i in range(3):
        assert np.array_equal(got[i].read(), want[i].read())
    for im in want + got:
        im.close()
    r.close()


def test_static_draw_limits_and_errors(vr, orc):
    if not have_gpu():
        pytest.skip("no GPU")
    r = vr.RayMarcher(0, 16, 16)
    cam = vr.uniform_from_array(sphere_cam(orc, 1, 3))
    imgs = _images(vr, r, 257, 16)
    r.draw_batch_static([cam] * 256, imgs[:256])                   # the maximum
    with pytest.raises(vr.VrmError) as e:
        r.draw_batch_static([cam] * 257, imgs)                     # beyond it: a clear error, nothing launched
    assert e.value.status == _ffi.VRM_ERR_INVALID and "at most 256" in str(e.value)
    r.draw_batch([cam] * 257, imgs)                                # the allocating form takes it
    a, b = imgs[0].read(), imgs[256].read()
    assert np.array_equal(a, b)
    # diagnostics / profiling switches are ignored, not honoured by allocating
    r.set_option(vr.OPT_DIAGNOSTICS, 1)
    r.set_option(vr.OPT_PROFILE, 1)
    r.draw_batch_static([cam], imgs[:1])
    assert np.array_equal(imgs[0].read(), a)
    with pytest.raises(vr.VrmError):
        imgs[0].diagnostics()                                      # no diagnostics buffers were created
    assert r.profile_read() == (0.0, 0)
    rc = _ffi.lib().vrm_draw_batch_static(r._h, None, 1, None)
    assert rc == _ffi.VRM_ERR_INVALID
    # a grid that was resized and neither uploaded to nor drawn yet has no RGBA storage: the entry point that may not
    # allocate says so (VRM_ERR_STATE), the allocating draw materialises the default voxels, and then it works
    r.voxel_grid.new((9, 7, 5))
    with pytest.raises(vr.VrmError) as e:
        r.draw_batch_static([cam], imgs[:1])
    assert e.value.status == _ffi.VRM_ERR_STATE and "may not allocate" in str(e.value)
    r.draw_batch([cam], imgs[1:2])
    r.draw_batch_static([cam], imgs[:1])
    assert np.array_equal(imgs[0].read(), imgs[1].read()) and imgs[0].read()[..., 3].max() > 0
    # a scalar upload leaves the compact form, which the default options march without anything else
    r.voxel_grid.upload_scalar(np.full(9 * 7 * 5, 200, np.uint8), (9, 7, 5), 1)
    r.draw_batch_static([cam], imgs[:1])
    r.draw_batch([cam], imgs[1:2])
    assert np.array_equal(imgs[0].read(), imgs[1].read())
    for im in imgs:
        im.close()
    r.close()


def test_static_draw_across_a_wrap_of_the_descriptor_ring(vr, orc):
    """the pinned descriptor ring holds 4096 views; 18 maximum-size static calls wrap it once (the call then waits
    for the stream before reusing the slots).  The image of every call must still be its own cameras'."""
    if not have_gpu():
        pytest.skip("no GPU")
    R = 16
    r = vr.RayMarcher(0, R, R)
    r.voxel_grid.upload_rgba8(synth.quantize_rgba8(synth.blobs_rgba_f32((24, 20, 28))))
    r.voxel_grid.attenuation = 40.0
    r.voxel_grid.update_voxel_grid_buffer()
    views = orc.sphere_views(4, 4)
    cams = [vr.uniform_from_array(sphere_cam(orc, ix, iy, xdiv=4, ydiv=4)) for ix, iy in views]
    want = _images(vr, r, len(cams), R)
    r.draw_batch(cams, want)
    want_px = [im.read() for im in want]
    imgs = _images(vr, r, 256, R)
    for call in range(18):                                         # 18 * 256 = 4608 views > 4096 slots
        k = call % len(cams)
        r.draw_batch_static([cams[(k + i) % len(cams)] for i in range(256)], imgs)
        for i in (0, 100, 255):
            assert np.array_equal(imgs[i].read(), want_px[(k + i) % len(cams)]), (call, i)
    for im in want + imgs:
        im.close()
    r.close()
