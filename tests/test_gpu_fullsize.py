"""BASELINE.json's full sizes on the GPU: C1 (128^3 DAT file, one 512^2 view), C2, C3 (512^3 RGBA, 1024^2), one C4 view (1024^3
scalar, 2048^2) and one rank's share of C5.
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


def test_c1_dat_volume_one_view(vr, orc, tmp_path):
    """BASELINE.json configs[0]: synthetic 128^3 uint8 DAT volume, ONE 512 x 512 view -- (ix, iy) = (1, 3) of the 8 x 8 sphere at
    radius 2.5, attenuation 1 (what every load resets it to, grid.rs:197), transfer function off -- read from DAT + RAW files on
    disk the way loaders/dat.rs:10-82 reads them (header keys, x fastest, UCHAR replicated to [b,b,b,b]), rendered by the
    HIP path and held to the oracle: <= 1 LSB, normalised RMSE < 1e-3, the oracle's iteration count per ray, skipping on/off
    bit-identical."""
    D, R = 128, 512
    dens = synth.blobs_scalar_u8((D, D, D))
    (tmp_path / "c1.raw").write_bytes(dens.tobytes())
    (tmp_path / "c1.dat").write_text(f"ObjectFileName: c1.raw\nResolution: {D} {D} {D}\nSliceThickness: 1 1 1\nFormat: UCHAR\n")
    r = vr.RayMarcher(0, R, R)
    r.voxel_grid.attenuation = 7.0                                # a load must reset this (grid.rs:100,197)
    r.voxel_grid.update_voxel_grid_buffer()
    r.load_dat(str(tmp_path / "c1.dat"))
    assert r.voxel_grid.dimensions == (D, D, D)
    vox = np.repeat(dens[..., None], 4, axis=3)                   # [b,b,b,b] (dat.rs:74)
    assert np.array_equal(r.voxel_grid.download_rgba8(), vox)
    cam = sphere_cam(orc, 1, 3, xdiv=8, ydiv=8)
    ref = orc.render(vox, cam, R, R, want_f32=False, want_steps=True)   # attenuation 1.0, TF off: the oracle's defaults
    assert (ref["rgba8"][..., 3] > 0).mean() > 0.1
    got = _render(r, vr, [cam], R)[0]
    m = image_metrics(got, ref["rgba8"])
    assert m["max"] <= 1 and m["rmse_norm"] < 1e-3, m
    r.set_option(vr.OPT_DIAGNOSTICS, 1)
    img = vr.Screenshotter(r, R, R)
    r.draw(vr.uniform_from_array(cam), img)
    _, iters = img.diagnostics()
    assert np.array_equal(iters, ref["steps"])                    # loop iterations per ray == the oracle's
    assert np.array_equal(img.read(), got)
    img.close()
    r.set_option(vr.OPT_DIAGNOSTICS, 0)
    r.set_option(vr.OPT_SKIP_EMPTY, 0)
    assert np.array_equal(_render(r, vr, [cam], R)[0], got)
    r.close()


def test_c2_full_size(vr, orc):
    """BASELINE.json configs[1]: 256^3 uint8 DAT volume, 64-view sphere export (8 x (9-1)) at 512 x 512, ingested through
    vrm_grid_upload_scalar (loaders/dat.rs:58-76).  Three views against the oracle, all 64 with skipping on and off."""
    D, R = 256, 512
    dens = synth.blobs_scalar_u8((D, D, D))
    r = vr.RayMarcher(0, R, R)
    r.voxel_grid.upload_scalar(dens.reshape(-1), (D, D, D), 1)
    assert r.voxel_grid.download_rgba8().shape == (D, D, D, 4)
    views = orc.sphere_views(8, 9)
    assert len(views) == 64
    cams = [sphere_cam(orc, ix, iy, xdiv=8, ydiv=9) for ix, iy in views]
    got = _render(r, vr, cams, R)                                 # default: packed 1-channel LDS march, skipping on
    vox = np.repeat(dens[..., None], 4, axis=3)                   # [b,b,b,b] (dat.rs:74)
    r.set_option(vr.OPT_DIAGNOSTICS, 1)
    for i in (3, 29, 60):                                         # near the top, the equator, near the bottom
        ref = orc.render(vox, cams[i], R, R, want_f32=False, want_steps=True)
        m = image_metrics(got[i], ref["rgba8"])
        assert m["max"] <= 1 and m["rmse_norm"] < 1e-3, (i, m)
        assert (ref["rgba8"][..., 3] > 0).mean() > 0.1
        img = vr.Screenshotter(r, R, R)
        r.draw(vr.uniform_from_array(cams[i]), img)
        _, iters = img.diagnostics()
        assert np.array_equal(iters, ref["steps"])                # loop iterations per ray == the oracle's
        assert np.array_equal(img.read(), got[i])                 # diagnostics build == production build
        img.close()
    r.set_option(vr.OPT_DIAGNOSTICS, 0)
    r.set_option(vr.OPT_SKIP_EMPTY, 0)
    no_skip = _render(r, vr, cams, R)
    for a, b in zip(got, no_skip):
        assert np.array_equal(a, b)
    r.set_option(vr.OPT_PACKED_ALPHA, 0)                          # RGBA-staged 1-channel march: same image to rounding
    rgba_staged = _render(r, vr, cams[:8], R)
    for a, b in zip(got, rgba_staged):
        assert image_metrics(a, b)["max"] <= 1
    r.close()


def test_c5_true_size_slab_of_one_rank(vr, orc):
    """BASELINE.json configs[4] at its true size for ONE rank: slab 3 of 8 of a 2048 x 2048 x 4096 scalar volume (the slices
    [1536, 2048) plus one ghost slice either side: 514 x 2048 x 2048 voxels, ~2.2 GB of input, ~20 GB resident) marched at
    2048 x 2048 on the whole volume's sample lattice.  Properties at full size; a 256 x 256 crop against the oracle, which
    renders the same partial from the stored slices alone (oracle.render_slab_stored)."""
    dims, R, G, k = (2048, 2048, 4096), 2048, 8, 3
    z0, z1 = sortlast.slab_ranges(dims[2], G)[k]
    r = vr.RayMarcher(0, R, R)
    s0, s1 = r.voxel_grid.new_slab(dims, z0, z1)
    assert (s0, s1) == (z0 - 1, z1 + 1)
    stored = synth.blobs_scalar_u8_slices(dims, s0, s1)
    assert stored.any()
    r.voxel_grid.upload_scalar(stored.reshape(-1), (dims[0], dims[1], s1 - s0), 1)
    u = r.voxel_grid.uniform()
    assert tuple(u.dimensions[:3]) == dims and tuple(u.box_size[:3]) == (1.0, 1.0, 2.0)   # the WHOLE volume's box
    cam36 = sphere_cam(orc, 3, 2, xdiv=8, ydiv=5)
    cam = vr.uniform_from_array(cam36)
    part = torch.empty((R, R, 4), dtype=torch.float32, device="cuda:0")
    r.draw_partial([cam], R, R, part.data_ptr())
    r.synchronize()
    got = part.cpu().numpy()
    assert got[..., 3].max() > 0.05 and np.isfinite(got).all()
    r.set_option(vr.OPT_SKIP_EMPTY, 0)                            # skipping off: bit-identical f32
    r.draw_partial([cam], R, R, part.data_ptr())
    r.synchronize()
    assert np.array_equal(part.cpu().numpy(), got)
    r.set_option(vr.OPT_SKIP_EMPTY, 1)
    # per ray: loop iterations over the whole box (summing to the reference-equivalent count) >= samples in this slab
    r.set_option(vr.OPT_DIAGNOSTICS, 1)
    img = vr.Screenshotter(r, R, R)
    r.draw(cam, img)
    f32_diag, iters = img.diagnostics()
    n_in = img.ray_diagnostics()[..., 5]
    r.set_option(vr.OPT_DIAGNOSTICS, 0)
    img.close()
    assert int(iters.astype(np.int64).sum()) == r.count_reference_samples([cam], R, R)
    assert (n_in <= iters).all() and 0 < n_in.sum() < 0.5 * iters.sum()
    assert np.abs(f32_diag - got).max() < 2e-5                   # diagnostics build (step by step) vs production (fast-forward)
    r.close()
    # the oracle on a crop where the slab is visible
    ys, xs = np.nonzero(got[..., 3] > 0.02)
    cy, cx = int(np.median(ys)), int(np.median(xs))
    x0, y0 = min(max(cx - 128, 0), R - 256), min(max(cy - 128, 0), R - 256)
    ref, steps = orc.render_slab_stored(stored, dims, s0, cam36, R, R, z0, z1, crop=(x0, y0, 256, 256))
    assert np.isfinite(ref).all()                                 # no texel outside the stored slices was touched
    assert np.array_equal(steps, iters[y0:y0 + 256, x0:x0 + 256])
    crop = got[y0:y0 + 256, x0:x0 + 256]
    assert ref[..., 3].max() > 0.02
    assert np.abs(crop - ref).max() < 1e-4, float(np.abs(crop - ref).max())
    assert np.abs(sortlast.unorm8(crop).astype(int) - sortlast.unorm8(ref).astype(int)).max() <= 1


def test_c5_resolution_fallback_chunks_skip_exactly(vr, orc):
    """Regression (round 2): at C5's resolution (2 texels per step and per pixel) the box of even ONE step of an oblique wave
    tile does not fit the LDS tile, the chunk falls back to 8 lockstep global-gather steps -- and the emptiness test used the
    one-step box for all 8, skipping samples whose texels it never looked at (a few hundred pixels of 4 M off by one sample).
    A thin slab of the true-size volume reproduces it in a second: skipping on/off must be bit-identical in f32."""
    dims, R = (2048, 2048, 4096), 2048
    z0, z1 = sortlast.slab_ranges(dims[2], 64)[24]
    r = vr.RayMarcher(0, R, R)
    s0, s1 = r.voxel_grid.new_slab(dims, z0, z1)
    stored = synth.blobs_scalar_u8_slices(dims, s0, s1)
    r.voxel_grid.upload_scalar(stored.reshape(-1), (dims[0], dims[1], s1 - s0), 1)
    cams = [vr.uniform_from_array(sphere_cam(orc, ix, iy, xdiv=8, ydiv=5)) for ix, iy in [(0, 1), (2, 2), (4, 3)]]
    part = torch.empty((len(cams), R, R, 4), dtype=torch.float32, device="cuda:0")
    out = {}
    for key, (kernel, skip, packed) in {"skip": (1, 1, 1), "noskip": (1, 0, 1), "rgba_skip": (1, 1, 0), "gather": (0, 0, 1)}.items():
        r.set_option(vr.OPT_KERNEL, kernel)
        r.set_option(vr.OPT_SKIP_EMPTY, skip)
        r.set_option(vr.OPT_PACKED_ALPHA, packed)
        r.draw_partial(cams, R, R, part.data_ptr())
        r.synchronize()
        out[key] = part.cpu().numpy().copy()
    r.close()
    assert out["skip"][..., 3].max() > 0.05
    assert np.array_equal(out["skip"], out["noskip"])
    # same samples through the global-gather baseline: its filter interpolates x as difference + product and it carries alpha,
    # the march two products and the transmittance (round 3) -- equal to a few f32 roundings of a 0..1 value
    assert np.abs(out["skip"] - out["gather"]).max() < 5e-6
    assert np.abs(out["rgba_skip"] - out["gather"]).max() < 5e-6
