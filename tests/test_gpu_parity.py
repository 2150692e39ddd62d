"""Parity of the HIP path (through the C ABI) with the CPU oracle on the same seeded inputs.

Gate (BASELINE.md section 2 / north_star): max |delta| <= 1/255 per RGBA channel on the 8-bit image and
normalised RMSE (scripts/similarity.py:59-62, /255) < 1e-3.  On top of that the f32 accumulators agree
to 1e-4 and the ray geometry (direction, t_min, t_max, iteration count, in-box sample count) is required
to be BIT-EXACT, because it drives the discontinuous decisions of the shader."""
import numpy as np
import pytest
import torch  # noqa: F401  (before libvrm.so initialises HIP: the library then shares torch's runtime, see bench.py)

from conftest import image_metrics, sphere_cam
from volume_renderer_amd import synth

pytestmark = pytest.mark.gpu

F32_TOL = 1e-4
F32_SEEN = []   # every f32 deviation check() has seen (tools/fuzz_parity.py prints their distribution)


def configure(rm, vox, att=1.0, use_tf=False):
    g = rm.voxel_grid
    g.upload_rgba8(vox)
    g.attenuation = att
    g.update_voxel_grid_buffer()
    g.set_transfer_function_active(use_tf)
    g.update_transfer_function_buffer()


def draw(rm, vr, cam36, W, H, diag=True):
    rm.set_option(vr.OPT_DIAGNOSTICS, 1 if diag else 0)
    img = vr.Screenshotter(rm, W, H)
    rm.draw(vr.uniform_from_array(cam36), img)
    out = {"rgba8": img.read()}
    if diag:
        out["f32"], out["iters"] = img.diagnostics()
        out["geo"] = img.ray_diagnostics()
    img.close()
    rm.set_option(vr.OPT_DIAGNOSTICS, 0)
    return out


def check(orc, got, vox, cam, W, H, att, use_tf, ert_free=True, f32_tol_scale=1.0):
    ref = orc.render(vox, cam, W, H, attenuation=att, tf=orc.tf_words(use_tf), want_steps=True)
    m = image_metrics(got["rgba8"], ref["rgba8"])
    assert m["max"] <= 1, m
    assert m["rmse_norm"] < 1e-3, m
    d32 = float(np.abs(got["f32"] - ref["f32"]).max())
    F32_SEEN.append(d32)
    assert d32 < F32_TOL * f32_tol_scale, f"f32 accumulators differ by {d32:.3e}"
    assert np.array_equal(got["iters"], ref["steps"]), f"{int((got['iters'] != ref['steps']).sum())} rays differ in iteration count"
    Z, Y, X, _ = vox.shape
    geo = orc.geometry(cam, (X, Y, Z), W, H)
    cols = [0, 1, 2, 3, 4, 6, 7] + ([5] if ert_free else [])
    for k in cols:
        assert np.array_equal(got["geo"][..., k].view(np.uint32), geo[..., k].view(np.uint32)), f"geometry column {k}"
    return m


VOLUMES = {
    "blobs_rgba": lambda: synth.quantize_rgba8(synth.blobs_rgba_f32((48, 40, 32))),
    "blobs_scalar": lambda: np.repeat(synth.blobs_scalar_u8((64, 64, 64))[..., None], 4, 3),
    "random_noncubic": lambda: synth.random_rgba8((17, 9, 33), 7),
    "solid_200": lambda: synth.solid_rgba8((8, 8, 8), 200),
}


@pytest.mark.parametrize("vol", sorted(VOLUMES))
@pytest.mark.parametrize("att,use_tf", [(1.0, False), (20.0, False), (1.0, True), (100.0, True)])
def test_parity_matrix(rm, vr, orc, vol, att, use_tf):
    vox = VOLUMES[vol]()
    configure(rm, vox, att, use_tf)
    for ix, iy in [(1, 3), (6, 6)]:
        cam = sphere_cam(orc, ix, iy)
        got = draw(rm, vr, cam, 96, 80)
        check(orc, got, vox, cam, 96, 80, att, use_tf)


def test_parity_all_views_of_default_sphere(rm, vr, orc):
    """the 8x8 export (56 views, similarity.py:51-55) at low resolution, averaged like similarity.py:75-78"""
    vox = synth.quantize_rgba8(synth.blobs_rgba_f32((32, 32, 32)))
    configure(rm, vox, 4.0, False)
    tot_rmse, worst = 0.0, 0
    views = orc.sphere_views(8, 8)
    for ix, iy in views:
        cam = sphere_cam(orc, ix, iy)
        got = draw(rm, vr, cam, 48, 48)
        m = check(orc, got, vox, cam, 48, 48, 4.0, False)
        tot_rmse += m["rmse_norm"]
        worst = max(worst, m["max"])
    assert worst <= 1 and tot_rmse / len(views) < 1e-3


def test_parity_ragged_image_sizes(rm, vr, orc):
    """sizes that are not multiples of the 16x16 workgroup tile or the 8x8 wave tile, incl. 1x1"""
    vox = synth.random_rgba8((12, 10, 14), 1)
    configure(rm, vox, 10.0, False)
    for W, H in [(1, 1), (7, 5), (50, 37), (65, 16), (16, 65), (130, 3)]:
        cam = sphere_cam(orc, 2, 5, aspect=W / H)
        got = draw(rm, vr, cam, W, H)
        check(orc, got, vox, cam, W, H, 10.0, False)


def test_parity_degenerate_volumes(rm, vr, orc):
    for dims in [(1, 1, 1), (2, 1, 1), (1, 5, 1), (3, 3, 3), (64, 2, 2), (2, 2, 64)]:
        vox = synth.random_rgba8(dims, 3)
        configure(rm, vox, 50.0, False)
        cam = sphere_cam(orc, 5, 3, radius=max(dims) / min(dims) * 1.2 + 1.5)
        got = draw(rm, vr, cam, 40, 40)
        check(orc, got, vox, cam, 40, 40, 50.0, False)


def test_parity_camera_inside_volume(rm, vr, orc):
    vox = synth.quantize_rgba8(synth.blobs_rgba_f32((24, 24, 24)))
    configure(rm, vox, 5.0, True)
    cam = orc.camera_uniform(np.array([0.1, 0.2, 0.3], np.float32), target=(1.0, 0.5, 0.0))
    got = draw(rm, vr, cam, 64, 64)
    check(orc, got, vox, cam, 64, 64, 5.0, True)
    assert (got["geo"][..., 3] < 0).all()  # negative t_min is kept (wgsl:137)


def test_parity_axis_aligned_rays_with_zero_direction_components(rm, vr, orc):
    vox = synth.random_rgba8((6, 6, 6), 5)
    configure(rm, vox, 20.0, False)
    cam = orc.camera_uniform(np.array([0.0, 0.0, 0.2], np.float32), target=(0.0, 0.0, -1.0))
    got = draw(rm, vr, cam, 15, 15)
    check(orc, got, vox, cam, 15, 15, 20.0, False)


def test_parity_reference_default_radius_100(rm, vr, orc):
    vox = synth.random_rgba8((8, 8, 8), 2)
    configure(rm, vox, 5.0, True)
    cam = sphere_cam(orc, 2, 5, radius=100.0)
    got = draw(rm, vr, cam, 512, 512)
    check(orc, got, vox, cam, 512, 512, 5.0, True)
    assert 10 < (got["rgba8"][..., 3] > 0).sum() < 2000


def test_early_termination_matches_oracle(rm, vr, orc):
    """alpha >= 1.0 (wgsl:176-178) only fires when one step closes the gap: attenuation 1000."""
    vox = synth.solid_rgba8((4, 4, 4), 255)
    configure(rm, vox, 1000.0, False)
    cam = sphere_cam(orc, 1, 3)
    got = draw(rm, vr, cam, 32, 32)
    ref = orc.render(vox, cam, 32, 32, attenuation=1000.0, want_steps=True)
    full = orc.geometry(cam, (4, 4, 4), 32, 32)[..., 7]
    assert (ref["steps"][full > 100] < 40).all()
    assert np.abs(got["iters"].astype(int) - ref["steps"].astype(int)).max() <= 1  # rounding of the last step to 1.0
    assert image_metrics(got["rgba8"], ref["rgba8"])["max"] <= 1


def test_closed_form_homogeneous_on_gpu(rm, vr, orc):
    for val, att in [(128, 1.0), (40, 7.5)]:
        vox = synth.solid_rgba8((8, 8, 8), val)
        configure(rm, vox, att, False)
        cam = sphere_cam(orc, 1, 3)
        got = draw(rm, vr, cam, 48, 48)
        n = got["geo"][..., 5].astype(np.float64)
        want = 1.0 - np.exp(-(val / 255.0) * att * 0.001 * n)
        assert np.abs(got["f32"][..., 3] - want).max() < 1e-4
        assert np.abs(got["f32"][..., 0] - (val / 255.0) * want).max() < 1e-4


def test_scalar_and_rgba_filters_agree(rm, vr, orc):
    """a DAT-style volume (r=g=b=a) takes the 1-channel filter; perturbing one voxel's red forces the 4-channel
    one; away from that voxel the two images must be identical."""
    s = synth.blobs_scalar_u8((32, 32, 32))
    vox = np.repeat(s[..., None], 4, 3)
    cam = sphere_cam(orc, 3, 4)
    configure(rm, vox, 3.0, False)
    a = draw(rm, vr, cam, 64, 64)
    vox2 = vox.copy()
    vox2[0, 0, 0, 0] ^= 1
    configure(rm, vox2, 3.0, False)
    b = draw(rm, vr, cam, 64, 64)
    assert (a["rgba8"] != b["rgba8"]).sum() <= 8
    assert np.abs(a["f32"] - b["f32"]).max() < 1e-5


def test_empty_space_skipping_is_bit_exact(vr, orc):
    """skipped samples contribute exactly 0 (a = 0 => alpha_src = 1 - exp(0) = 0), and the dt lattice is advanced
    with the same running f32 sum: image, f32 accumulators and iteration counts must be IDENTICAL with the option
    on and off, and the kernel variants must agree with each other as well."""
    if not __import__("os").path.exists("/dev/kfd"):
        pytest.skip("no GPU")
    rm = vr.RayMarcher(0, 256, 256)
    dens = synth.blobs_density((96, 80, 112))
    rgba = synth.quantize_rgba8(synth.blobs_rgba_f32((96, 80, 112)))
    island = np.zeros((64, 64, 64, 4), np.uint8)           # one small occupied island in a big empty box
    island[20:29, 40:44, 8:19] = (255, 10, 90, 200)
    assert (dens == 0).mean() > 0.5
    for vox, att, tf in [(rgba, 1.0, False), (rgba, 40.0, True), (island, 25.0, False)]:
        configure(rm, vox, att, tf)
        for ix, iy in [(1, 3), (4, 6), (7, 1)]:
            cam = sphere_cam(orc, ix, iy)
            res = {}
            for k, s in [(0, 0), (1, 0), (1, 1)]:
                rm.set_option(vr.OPT_KERNEL, k)
                rm.set_option(vr.OPT_SKIP_EMPTY, s)
                res[(k, s)] = draw(rm, vr, cam, 160, 120)
            base, skip, glob = res[(1, 0)], res[(1, 1)], res[(0, 0)]
            assert np.array_equal(skip["rgba8"], base["rgba8"])
            assert np.array_equal(skip["f32"].view(np.uint32), base["f32"].view(np.uint32))   # bit for bit
            for other in (skip, glob):
                assert np.array_equal(other["iters"], base["iters"])
                assert np.array_equal(other["geo"].view(np.uint32), base["geo"].view(np.uint32))
            # the global-gather variant forms the texel coordinate from the exact position, the LDS variant (in
            # interior chunks) straight from dt: continuous arithmetic, equal to rounding
            assert np.abs(glob["f32"] - base["f32"]).max() < F32_TOL
            assert np.abs(glob["rgba8"].astype(int) - base["rgba8"].astype(int)).max() <= 1
    rm.close()


def test_batch_equals_single_draws(rm, vr, orc):
    vox = synth.quantize_rgba8(synth.blobs_rgba_f32((32, 24, 40)))
    configure(rm, vox, 2.0, False)
    cams = [vr.uniform_from_array(sphere_cam(orc, ix, iy)) for ix, iy in [(0, 1), (3, 3), (7, 6), (4, 2), (1, 5)]]
    imgs = [vr.Screenshotter(rm, 80, 64) for _ in cams]
    rm.draw_batch(cams, imgs)
    batch = [im.read() for im in imgs]
    for cam, im, want in zip(cams, imgs, batch):
        rm.draw(cam, im)
        assert np.array_equal(im.read(), want)
    for im in imgs:
        im.close()


def test_far_cameras_filter_at_the_oracle_sequence(rm, vr, orc):
    """A camera hundreds of volume widths away: the camera position in texels (> 4096) makes the one-FMA texel coordinate
    too coarse for the f32 contract on a noise volume, so such views are marched by the FAR instance of the kernel (the
    oracle's own (ro + rd*dt)*scale + bias).  Same gates as everywhere else, and a batch that mixes near and far views
    gives every view the image it gets alone."""
    vox = synth.random_rgba8((130, 96, 65), 5)
    configure(rm, vox, 40.0, True)
    far = [orc.camera_uniform(orc.sphere_position(ix, iy, 8, 8, r)) for ix, iy, r in [(5, 2, 400.0), (1, 6, 100.0), (3, 3, 60.0)]]
    for cam in far:
        got = draw(rm, vr, cam, 96, 40)
        check(orc, got, vox, cam, 96, 40, 40.0, True)
    cams = [vr.uniform_from_array(c) for c in [sphere_cam(orc, 2, 3), far[0], sphere_cam(orc, 6, 5), far[1]]]
    imgs = [vr.Screenshotter(rm, 72, 56) for _ in cams]
    rm.draw_batch(cams, imgs)
    batch = [im.read() for im in imgs]
    for cam, im, want in zip(cams, imgs, batch):
        rm.draw(cam, im)
        assert np.array_equal(im.read(), want)
    for im in imgs:
        im.close()


def test_parity_step_cap_on_a_long_volume(rm, vr, orc):
    """box_size = dims / min(dims) (grid.rs:268): a 4x4x48 volume is 12 units long, rays along z would need 12000 steps and
    stop at the 5000th iteration (wgsl:96,139) -- per lane, wherever the chunking put that step."""
    vox = synth.random_rgba8((4, 4, 48), 12)
    configure(rm, vox, 0.5, False)
    for pos, tgt in [((0.05, 0.02, 9.0), (0.0, 0.0, 0.0)), ((3.0, 0.5, 7.0), (0.0, 0.0, -2.0))]:
        cam = orc.camera_uniform(np.array(pos, np.float32), target=tgt)
        got = draw(rm, vr, cam, 40, 24)
        check(orc, got, vox, cam, 40, 24, 0.5, False)
        if pos[2] == 9.0:
            assert got["iters"].max() == 5000


def test_rays_that_are_not_a_number_end_like_the_references(rm1, vr, orc):
    """A camera 1000 box widths away: f32 leaves the centre pixel's near-plane point ON the camera position, its direction is
    normalize(0) = NaN, and so are t_min and t_max (wgsl:247-299).  The reference marches such a ray through all 5000 iterations
    without a sample (every comparison of the in-box test is false).  The LDS march derived 0 steps per chunk from the NaN position
    and, once only such lanes were left in a wave, planned the same chunk for ever -- the default kernel hung (round 3, found by
    tools/far_sweep.py).  Every variant must end, with the reference's iteration count and its empty pixel; production build and
    diagnostics build, skipping on and off."""
    rm = rm1
    vox = synth.random_rgba8((64, 15, 64), 12345)
    configure(rm, vox, 100.0, False)
    W, H = 31, 83
    cam = orc.camera_uniform(orc.sphere_position(3, 5, 8, 8, 1000.0), target=(0.0, 0.0, 0.0), aspect=W / H)
    ref = orc.render(vox, cam, W, H, attenuation=100.0, tf=orc.tf_words(False), want_steps=True)
    assert ref["steps"].max() == 5000 and (ref["steps"] > 0).sum() == 1          # the one NaN ray "hits" (no comparison says miss)
    geo = orc.geometry(cam, (64, 15, 64), W, H)
    assert np.isnan(geo[ref["steps"] == 5000][0][:5]).all()
    try:
        for skip in (1, 0):
            rm.set_option(vr.OPT_SKIP_EMPTY, skip)
            for k in (1, 3, 4, 0):
                rm.set_option(vr.OPT_KERNEL, k)
                got = draw(rm, vr, cam, W, H)
                assert np.array_equal(got["iters"], ref["steps"]), (skip, k)
                assert np.array_equal(got["rgba8"], ref["rgba8"]), (skip, k)
                assert np.array_equal(draw(rm, vr, cam, W, H, diag=False)["rgba8"], ref["rgba8"]), (skip, k)
    finally:
        rm.set_option(vr.OPT_KERNEL, 1)
        rm.set_option(vr.OPT_SKIP_EMPTY, 1)


def test_parity_wide_field_of_view_per_tile(rm, vr, orc):
    """An 8x8 image: one wave tile spans the whole 45-degree field of view and, from inside the volume, rays of one wave
    leave through different faces -- the lanes do not share a major axis (lockstep fallback of the plane front)."""
    vox = synth.quantize_rgba8(synth.blobs_rgba_f32((40, 40, 40)))
    configure(rm, vox, 30.0, False)
    for pos, tgt in [((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)), ((0.3, -0.2, 0.1), (-1.0, 0.2, 0.4)), ((1.5, 1.4, 1.6), (0.0, 0.0, 0.0))]:
        cam = orc.camera_uniform(np.array(pos, np.float32), target=tgt)
        for W, H in [(8, 8), (3, 2)]:
            got = draw(rm, vr, cam, W, H)
            check(orc, got, vox, cam, W, H, 30.0, False)


def test_parity_volume_dims_off_the_brick_grid(rm, vr, orc):
    """dims that are not multiples of the 4x4x2 brick or the 4-texel staging quads, viewed so that boxes overhang every face"""
    for dims in [(13, 11, 7), (5, 6, 9), (31, 30, 29)]:
        vox = synth.random_rgba8(dims, sum(dims))
        configure(rm, vox, 25.0, False)
        for ix, iy in [(0, 4), (2, 4), (3, 1), (5, 7)]:
            cam = sphere_cam(orc, ix, iy, radius=2.2)
            got = draw(rm, vr, cam, 72, 56)
            check(orc, got, vox, cam, 72, 56, 25.0, False)


def test_production_kernel_equals_diagnostics_kernel(rm, vr, orc):
    """The image of the kernels without the diagnostics outputs (what bench.py and every user runs) is bit-identical to
    the diagnostics build that the parity tests above compare with the oracle."""
    cases = [(synth.quantize_rgba8(synth.blobs_rgba_f32((48, 40, 32))), 20.0, False, [(1, 3), (6, 6), (0, 4)]),
             (np.repeat(synth.blobs_scalar_u8((40, 40, 40))[..., None], 4, 3), 5.0, True, [(2, 2)]),
             (synth.random_rgba8((4, 4, 48), 12), 0.5, False, [(0, 4), (2, 4)])]
    for vox, att, tf, views in cases:
        configure(rm, vox, att, tf)
        for ix, iy in views:
            cam = sphere_cam(orc, ix, iy, radius=3.0 if vox.shape[0] == 48 else 2.5)
            a = draw(rm, vr, cam, 100, 84, diag=True)["rgba8"]
            b = draw(rm, vr, cam, 100, 84, diag=False)["rgba8"]
            assert np.array_equal(a, b)


@pytest.mark.parametrize("packed", [1, 0])
def test_parity_one_channel_march_both_staging_sources(rm, vr, orc, packed):
    """The 1-channel march (scalar volumes, transfer function) stages either the packed alpha bricks (default) or the
    RGBA8 bricks (VRM_OPT_PACKED_ALPHA = 0): both against the oracle, on dims off the 16-texel packed brick grid,
    thin volumes, ragged images and a camera inside the volume."""
    rm.set_option(vr.OPT_PACKED_ALPHA, packed)
    try:
        cases = [((37, 21, 19), 6.0, False, 72, 56, 2.5), ((16, 16, 16), 3.0, True, 64, 64, 2.5), ((65, 3, 3), 30.0, False, 50, 37, 30.0),
                 ((1, 1, 1), 80.0, False, 24, 24, 3.0), ((50, 7, 33), 12.0, True, 65, 16, 9.0), ((15, 17, 16), 9.0, False, 33, 70, 1.9)]
        for dims, att, use_tf, W, H, radius in cases:
            sc = synth.blobs_scalar_u8(dims) if min(dims) >= 8 else np.random.default_rng(sum(dims)).integers(0, 256, dims[::-1], dtype=np.uint8)
            vox = np.repeat(np.ascontiguousarray(sc)[..., None], 4, 3)
            configure(rm, vox, att, use_tf)
            for ix, iy in [(1, 3), (6, 5)]:
                cam = sphere_cam(orc, ix, iy, radius=radius, aspect=W / H)
                got = draw(rm, vr, cam, W, H)
                check(orc, got, vox, cam, W, H, att, use_tf)
        # the transfer function on a volume that is NOT scalar reads the alpha channel only (wgsl:145-147)
        vox = synth.random_rgba8((23, 18, 35), 11)
        configure(rm, vox, 15.0, True)
        cam = sphere_cam(orc, 3, 2)
        check(orc, draw(rm, vr, cam, 80, 64), vox, cam, 80, 64, 15.0, True)
        # camera inside the volume
        sc = synth.blobs_scalar_u8((40, 40, 40))
        vox = np.repeat(sc[..., None], 4, 3)
        configure(rm, vox, 5.0, False)
        cam = orc.camera_uniform((0.1, 0.05, -0.2), target=(0.4, 0.3, 0.5))
        check(orc, draw(rm, vr, cam, 64, 64), vox, cam, 64, 64, 5.0, False)
    finally:
        rm.set_option(vr.OPT_PACKED_ALPHA, 1)


def test_one_channel_staging_sources_agree(rm, vr, orc):
    """Packed and RGBA staging feed the same arithmetic the same values: equal iteration counts, f32 accumulators equal
    to rounding of the texel coordinate (the staged box has another origin)."""
    sc = synth.blobs_scalar_u8((72, 60, 52))
    vox = np.repeat(sc[..., None], 4, 3)
    configure(rm, vox, 8.0, False)
    cam = sphere_cam(orc, 5, 2)
    out = []
    for packed in (1, 0):
        rm.set_option(vr.OPT_PACKED_ALPHA, packed)
        out.append(draw(rm, vr, cam, 128, 96))
    rm.set_option(vr.OPT_PACKED_ALPHA, 1)
    assert np.array_equal(out[0]["iters"], out[1]["iters"])
    assert np.abs(out[0]["f32"] - out[1]["f32"]).max() < 1e-5
    assert np.abs(out[0]["rgba8"].astype(int) - out[1]["rgba8"].astype(int)).max() <= 1
    assert out[0]["rgba8"][..., 3].max() > 50


def test_production_f32_accumulators_equal_diagnostics_build(rm, vr, orc):
    """The production kernels cross empty chunks with advance_exact (n steps of the running f32 dt sum in O(1) per
    binade); the diagnostics build, which the oracle comparisons above read, takes them one by one.  The f32
    accumulators of the production kernel (vrm_draw_partial) must be BIT-identical: one step off the lattice anywhere
    moves every later sample.  Long empty runs in front of the data, t crossing the binades at 2, 4 and 64..128,
    negative t inside the volume, tiny t."""
    W, H = 88, 72
    far = np.zeros((48, 40, 56, 4), np.uint8)                      # (Z, Y, X, 4): data only in the last z slices and a corner
    far[40:, :, :, :] = synth.random_rgba8((56, 40, 8), 3)
    far[:6, :5, :7, :] = 200
    cases = [(synth.quantize_rgba8(synth.blobs_rgba_f32((64, 64, 64))), 8.0, False),
             (np.repeat(synth.blobs_scalar_u8((72, 48, 40))[..., None], 4, 3), 15.0, True),
             (far, 30.0, False)]
    cams = [sphere_cam(orc, 1, 3, radius=2.5, aspect=W / H), sphere_cam(orc, 6, 5, radius=1.2, aspect=W / H),
            sphere_cam(orc, 3, 2, radius=4.5, aspect=W / H), sphere_cam(orc, 0, 4, radius=100.0, aspect=W / H),
            sphere_cam(orc, 5, 6, radius=9.0, aspect=W / H),
            orc.camera_uniform((0.1, 0.05, -0.2), target=(0.4, 0.3, 0.5), aspect=W / H),
            orc.camera_uniform((0.0, 0.0, 0.0), target=(-0.3, 0.9, 0.2), aspect=W / H)]
    nonzero = 0
    for vox, att, tf in cases:
        configure(rm, vox, att, tf)
        for skip in (1, 0):
            rm.set_option(vr.OPT_SKIP_EMPTY, skip)
            part = torch.empty((len(cams), H, W, 4), dtype=torch.float32, device="cuda")
            rm.draw_partial([vr.uniform_from_array(c) for c in cams], W, H, part.data_ptr())
            rm.synchronize()
            prod = part.cpu().numpy()
            for i, c in enumerate(cams):
                ref = draw(rm, vr, c, W, H, diag=True)["f32"]
                assert np.array_equal(prod[i].view(np.uint32), ref.view(np.uint32)), (i, skip, float(np.abs(prod[i] - ref).max()))
                nonzero += int(ref[..., 3].max() > 0.05)
    rm.set_option(vr.OPT_SKIP_EMPTY, 1)
    assert nonzero >= 30


def test_executed_sample_counts(rm, vr, orc):
    """N_exec (vrm_image_read_sample_counts): with skipping off every in-box sample is filtered; with skipping on only
    those in chunks that were not passed over -- fewer on a sparse volume, never more, and the iteration lattice (N_ref)
    does not move."""
    vox = synth.quantize_rgba8(synth.blobs_rgba_f32((64, 64, 64)))
    configure(rm, vox, 3.0, False)
    cam = vr.uniform_from_array(sphere_cam(orc, 2, 3))
    rm.set_option(vr.OPT_DIAGNOSTICS, 1)
    img = vr.Screenshotter(rm, 96, 96)
    res = {}
    for skip in (0, 1):
        rm.set_option(vr.OPT_SKIP_EMPTY, skip)
        rm.draw(cam, img)
        res[skip] = (img.executed_samples().astype(np.int64), img.ray_diagnostics()[..., 5].astype(np.int64), img.diagnostics()[1].astype(np.int64))
    rm.set_option(vr.OPT_SKIP_EMPTY, 1)
    rm.set_option(vr.OPT_DIAGNOSTICS, 0)
    img.close()
    ex0, in0, it0 = res[0]
    ex1, in1, it1 = res[1]
    assert np.array_equal(ex0, in0) and np.array_equal(in0, in1) and np.array_equal(it0, it1)
    assert (ex1 <= in1).all() and it0.max() < 5001
    if rm.get_option(vr.OPT_KERNEL) != 0:          # the global-gather baseline does not skip
        assert ex1.sum() < 0.9 * in1.sum()


def test_two_rays_per_lane_against_oracle_and_one_ray_form(rm, vr, orc):
    """VRM_OPT_RAYS_PER_LANE = 2: wave tiles of 16x8 pixels, each lane marches two x-adjacent pixels through one staged box per
    chunk (the packed scalar march).  Against the oracle like every other form, and BIT-identical to the one-ray form in the
    8-bit image, the f32 accumulators, the iteration counts and the ray geometry: how the march is cut into chunks does not
    change what a sample contributes.  Image widths off the 16-pixel tile, a one-column image, dims off the brick grid, a
    camera inside the volume, early termination, skipping on and off, both dispatch modes."""
    cases = [((37, 21, 19), 6.0, 72, 56, 2.5), ((64, 64, 64), 3.0, 333, 90, 2.5), ((65, 3, 3), 30.0, 50, 37, 30.0), ((1, 1, 1), 80.0, 24, 24, 3.0),
             ((50, 7, 33), 12.0, 1, 40, 9.0), ((40, 40, 40), 900.0, 130, 100, 2.5), ((96, 80, 112), 2.0, 256, 200, 1.6)]
    try:
        for dims, att, W, H, radius in cases:
            sc = synth.blobs_scalar_u8(dims) if min(dims) >= 8 else np.random.default_rng(sum(dims)).integers(0, 256, dims[::-1], dtype=np.uint8)
            vox = np.repeat(np.ascontiguousarray(sc)[..., None], 4, 3)
            configure(rm, vox, att, False)
            cams = [sphere_cam(orc, 1, 3, radius=radius, aspect=W / H), sphere_cam(orc, 6, 5, radius=radius, aspect=W / H),
                    orc.camera_uniform((0.1, 0.05, -0.2), target=(0.4, 0.3, 0.5), aspect=W / H)]
            for cam in cams:
                for skip, kernel in [(1, 1), (0, 3), (1, 4)]:
                    rm.set_option(vr.OPT_SKIP_EMPTY, skip)
                    rm.set_option(vr.OPT_KERNEL, kernel)
                    out = {}
                    for rays in (1, 2):
                        rm.set_option(vr.OPT_RAYS_PER_LANE, rays)
                        out[rays] = draw(rm, vr, cam, W, H)
                        part = torch.empty((1, H, W, 4), dtype=torch.float32, device="cuda")
                        rm.draw_partial([vr.uniform_from_array(cam)], W, H, part.data_ptr())
                        rm.synchronize()
                        out[rays]["partial"] = part.cpu().numpy()[0]
                    for k in ("rgba8", "f32", "iters", "geo", "partial"):
                        a, b = np.ascontiguousarray(out[1][k]), np.ascontiguousarray(out[2][k])
                        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (dims, W, H, skip, kernel, k)
                    assert np.array_equal(out[2]["partial"].view(np.uint32), out[2]["f32"].view(np.uint32))
                if att < 600.0:
                    check(orc, out[2], vox, cam, W, H, att, False)
                else:   # (early termination: the last step may round to 1.0 one iteration apart, see test_early_termination_matches_oracle)
                    ref = orc.render(vox, cam, W, H, attenuation=att, want_steps=True)
                    assert image_metrics(out[2]["rgba8"], ref["rgba8"])["max"] <= 1
                    assert np.abs(out[2]["iters"].astype(int) - ref["steps"].astype(int)).max() <= 1
    finally:
        rm.set_option(vr.OPT_RAYS_PER_LANE, 0)
        rm.set_option(vr.OPT_SKIP_EMPTY, 1)
        rm.set_option(vr.OPT_KERNEL, 1)
