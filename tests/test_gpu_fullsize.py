"""BASELINE.json's full sizes on the GPU: C3 (512^3 RGBA, 1024^2) and one C4 view (1024^3 scalar, 2048^2).
Whole views against the oracle (about a second of 16 CPU threads each at C3), plus the size-independent properties:
skipping on/off bit-identical, baseline kernel vs LDS kernel within 1 LSB, per-ray iteration counts summing to the
reference-equivalent sample count, sort-last over 4 slabs equal to the unpartitioned render."""
import numpy as np
import pytest
import torch

from conftest import image_metrics, sphere_cam
from volume_renderer_amd import sortlast, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c3_volume():
    return synth.quantize_rgba8(synth.blobs_rgba_f32((512, 512, 512)))


def _render(r, vr, cams36, R):
    imgs = [vr.Screenshotter(r, R, R) for _ in cams36]
    r.draw_batch([vr.uniform_from_array(c) for c in cams36], imgs)
    out = [im.read() for im in imgs]
    for im in imgs:
        im.close()
    return out


def test_c3_full_size(vr, orc, c3_volume):
    R = 1024
    views = [(3, 5), (9, 14), (0, 8)]                              # oblique, near a pole, axis-aligned (of the 16x17 sphere)
    cams = [sphere_cam(orc, ix, iy, xdiv=16, ydiv=17) for ix, iy in views]
    r = vr.RayMarcher(0, R, R)
    r.voxel_grid.upload_rgba8(c3_volume)
    got = _render(r, vr, cams, R)                                 # default: LDS kernel, skipping on
    for i in range(2):                                            # whole views against the oracle
        ref = orc.render(c3_volume, cams[i], R, R, want_f32=False)
        m = image_metrics(got[i], ref["rgba8"])
        assert m["max"] <= 1 and m["rmse_norm"] < 1e-3, m
        assert (ref["rgba8"][..., 3] > 0).mean() > 0.1           # the view is not trivially empty
    # the number of loop iterations per ray, summed, is the metric's unit
    r.set_option(vr.OPT_DIAGNOSTICS, 1)
    img = vr.Screenshotter(r, R, R)
    r.draw(vr.uniform_from_array(cams[0]), img)
    _, iters = img.diagnostics()
    assert int(np.asarray(iters, dtype=np.int64).sum()) == r.count_reference_samples([vr.uniform_from_array(cams[0])], R, R)
    assert np.array_equal(img.read(), got[0])
    img.close()
    r.set_option(vr.OPT_DIAGNOSTICS, 0)
    # skipping off: bit-identical; rendering twice: bit-identical
    r.set_option(vr.OPT_SKIP_EMPTY, 0)
    no_skip = _render(r, vr, cams, R)
    r.set_option(vr.OPT_SKIP_EMPTY, 1)
    again = _render(r, vr, cams, R)
    for a, b, c in zip(got, no_skip, again):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    # the global-gather baseline kernel: different arithmetic order in the filter only
    r.set_option(vr.OPT_KERNEL, 0)
    base = _render(r, vr, cams, R)
    r.set_option(vr.OPT_KERNEL, 1)
    for a, b in zip(got, base):
        m = image_metrics(a, b)
        assert m["max"] <= 1 and m["rmse_norm"] < 1e-3, m
    # sort-last over 4 slabs
    parts = []
    ucams = [vr.uniform_from_array(c) for c in cams]
    for z0, z1 in sortlast.slab_ranges(512, 4):
        s0, s1 = r.voxel_grid.new_slab((512, 512, 512), z0, z1)
        r.voxel_grid.upload_rgba8(c3_volume[s0:s1])
        t = torch.empty((len(cams), R, R, 4), dtype=torch.float32, device="cuda:0")
        r.draw_partial(ucams, R, R, t.data_ptr())
        r.synchronize()
        parts.append(t)
    stack = torch.stack(parts)
    for i, cam in enumerate(ucams):
        pv = stack[:, i].contiguous()
        out8 = torch.zeros((R, R, 4), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        r.composite(cam, pv.data_ptr(), 4, R, R, 0, R, out8.data_ptr(), None)
        r.synchronize()
        assert np.abs(out8.cpu().numpy().astype(int) - got[i].astype(int)).max() <= 1
    r.close()


def test_c4_one_view_against_oracle(vr, orc):
    D, R = 1024, 2048
    dens = synth.blobs_scalar_u8((D, D, D))
    r = vr.RayMarcher(0, R, R)
    r.voxel_grid.upload_scalar(dens.reshape(-1), (D, D, D), 1)
    cam = sphere_cam(orc, 5, 3, xdiv=16, ydiv=9)
    got = _render(r, vr, [cam], R)[0]
    r.set_option(vr.OPT_SKIP_EMPTY, 0)
    assert np.array_equal(_render(r, vr, [cam], R)[0], got)
    r.close()
    vox = np.repeat(dens[..., None], 4, axis=3)                   # [b,b,b,b] (dat.rs:74)
    del dens
    ref = orc.render(vox, cam, R, R, want_f32=False)
    m = image_metrics(got, ref["rgba8"])
    assert m["max"] <= 1 and m["rmse_norm"] < 1e-3, m
    assert (ref["rgba8"][..., 3] > 0).mean() > 0.1
