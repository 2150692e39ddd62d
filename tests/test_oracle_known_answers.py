"""Analytic known-answer tests that pin the CPU oracle (SURVEY.md section 4 list).

The reference has no tests, golden images or fixtures for the ray-marching path, so these closed-form
cases are what "the oracle follows raymarcher.wgsl" rests on.  CPU only.
"""
import numpy as np
import pytest

from conftest import sphere_cam


def solid(dims, rgba):
    X, Y, Z = dims
    v = np.zeros((Z, Y, X, 4), np.uint8)
    v[...] = rgba
    return v


def test_homogeneous_volume_closed_form(orc):
    # every voxel alpha = a: pixel alpha = 1 - exp(-a*att*0.001*n), colour = a * alpha (premultiplied, wgsl:150-159)
    for val, att in [(128, 1.0), (40, 7.5), (255, 1.0)]:
        vox = solid((8, 8, 8), (val, val, val, val))
        cam = sphere_cam(orc, 1, 3)
        r = orc.render(vox, cam, 48, 48, attenuation=att, want_steps=True)
        # n = samples that pass scene()'s in-box test: the first sample sits ON the entry face and rounding
        # can put it just outside, where scene() returns 0 (wgsl:218-223)
        geo = orc.geometry(cam, (8, 8, 8), 48, 48)
        n = geo[..., 5].astype(np.float64)
        assert np.array_equal(geo[..., 7].astype(np.uint32), r["steps"]) and (r["steps"] - n).max() <= 2
        a = val / 255.0
        want = 1.0 - np.exp(-a * att * 0.001 * n)
        got = r["f32"][..., 3].astype(np.float64)
        assert np.abs(got - want).max() < 1e-4
        for c in range(3):
            assert np.abs(r["f32"][..., c] - a * want).max() < 1e-4
        assert r["steps"].max() > 1000  # the cube diagonal region is crossed


def test_step_count_is_resolution_independent(orc):
    cam = sphere_cam(orc, 2, 5)
    a = orc.render(solid((4, 4, 4), (9, 9, 9, 9)), cam, 32, 32, want_steps=True)["steps"]
    b = orc.render(solid((32, 32, 32), (9, 9, 9, 9)), cam, 32, 32, want_steps=True)["steps"]
    assert np.array_equal(a, b)  # step is 0.001 world units, box edge is 1.0 (grid.rs:268-271)


def test_step_count_matches_chord_length(orc):
    cam = sphere_cam(orc, 0, 4)
    geo = orc.geometry(cam, (8, 8, 8), 40, 40)
    hit = geo[..., 6] > 0
    chord = (geo[..., 4] - geo[..., 3])[hit].astype(np.float64)
    iters = geo[..., 7][hit].astype(np.float64)
    assert np.all(np.abs(iters - np.ceil(chord / 0.001)) <= 2)  # running f32 sum drifts by at most a step
    assert not hit.all() and hit.any()


def test_empty_volume_and_misses_are_transparent(orc):
    vox = solid((6, 5, 4), (0, 0, 0, 0))
    r = orc.render(vox, sphere_cam(orc, 3, 2), 32, 32, want_steps=True)
    assert not r["rgba8"].any() and not r["f32"].any()
    # radius 100 (the reference default, state.rs:160): almost every ray misses the box -> (0,0,0,0) (wgsl:131-134)
    vox = solid((6, 5, 4), (255, 255, 255, 255))
    r = orc.render(vox, sphere_cam(orc, 3, 2, radius=100.0), 512, 512, want_steps=True)
    miss = r["steps"] == 0
    assert miss.mean() > 0.9 and (~miss).any()
    assert not r["f32"][miss].any()
    assert (r["f32"][~miss][:, 3] > 0).all()


def test_default_grid_is_black_and_opaque(orc):
    # RayMarcher::new: 3x3x3 of [0,0,0,255] (mod.rs:29, voxel.rs:13-16): colour stays 0, alpha accumulates
    vox = solid((3, 3, 3), (0, 0, 0, 255))
    r = orc.render(vox, sphere_cam(orc, 1, 3), 24, 24, want_steps=True)
    hit = r["steps"] > 0
    assert not r["f32"][..., :3].any()
    n = orc.geometry(sphere_cam(orc, 1, 3), (3, 3, 3), 24, 24)[..., 5][hit].astype(np.float64)
    assert np.abs(r["f32"][hit][:, 3] - (1 - np.exp(-0.001 * n))).max() < 1e-4


def test_impulse_axis_order_texel_centres_and_clamp(orc):
    dims = (5, 4, 3)
    X, Y, Z = dims
    vox = solid(dims, (0, 0, 0, 0))
    vox[2, 1, 3] = (10, 20, 30, 255)  # voxel x=3, y=1, z=2 -> idx = x + X*(y + Y*z) (grid.rs:211)
    _, bmin, bsz, _ = orc.grid_uniform(dims)
    bmin, bsz = bmin[:3].astype(np.float64), bsz[:3].astype(np.float64)

    def centre(ix, iy, iz):
        return bmin + (np.array([ix, iy, iz]) + 0.5) / np.array(dims) * bsz

    got = orc.scene(vox, centre(3, 1, 2))
    assert np.allclose(got, np.array([10, 20, 30, 255]) / 255.0, atol=2e-6)
    # half-way to the +x neighbour: linear filter gives half; same along y and z
    for off in [(0.5, 0, 0), (0, 0.5, 0), (0, 0, -0.5)]:
        got = orc.scene(vox, centre(3 + off[0], 1 + off[1], 2 + off[2]))
        assert np.allclose(got, 0.5 * np.array([10, 20, 30, 255]) / 255.0, atol=1e-5)
    # a different axis order would put the impulse elsewhere
    assert not orc.scene(vox, centre(2, 1, 3 % Z)).any()
    # clamp-to-edge: within half a texel of the face the edge texel is returned unattenuated
    edge = solid(dims, (0, 0, 0, 0))
    edge[:, :, 0] = (200, 0, 0, 200)  # the x = 0 face layer
    p = centre(0, 1, 1)
    p[0] = bmin[0] + 1e-4
    assert np.allclose(orc.scene(edge, p), np.array([200, 0, 0, 200]) / 255.0, atol=1e-5)
    # outside the half-open box -> zeros (wgsl:218-223): exactly box_max is out, exactly box_min is in
    full = solid(dims, (255, 255, 255, 255))
    bmax32 = (orc.grid_uniform(dims)[1][:3] + orc.grid_uniform(dims)[2][:3]).astype(np.float32)
    bmin32 = orc.grid_uniform(dims)[1][:3]
    mid = centre(2, 2, 1).astype(np.float32)
    for a in range(3):
        p = mid.copy(); p[a] = bmax32[a]
        assert not orc.scene(full, p).any()
        p = mid.copy(); p[a] = bmin32[a]
        assert orc.scene(full, p)[3] == 1.0
        p = mid.copy(); p[a] = np.nextafter(bmin32[a], np.float32(-10))
        assert not orc.scene(full, p).any()


def tf_expected(a):
    A = np.array([0.117, 0.188, 0.62]); B = np.array([0.7294] * 3); C = np.array([0.5725, 0.0, 0.039])
    if a >= 0.5:
        t = (a - 0.5) / 0.5
        return B * (1 - t) + C * t
    t = a / 0.5
    return A * (1 - t) + B * t


@pytest.mark.parametrize("val", [0, 64, 127, 128, 191, 255])
def test_transfer_function_nodes(orc, val):
    # wgsl:145-147: A->B on [0,0.5), B->C on [0.5,1]; colour = TF(a) * alpha for a homogeneous volume
    vox = solid((4, 4, 4), (7, 99, 200, val))  # rgb must be ignored when the TF is on (wgsl:153-154)
    r = orc.render(vox, sphere_cam(orc, 1, 3), 16, 16, attenuation=3.0, tf=orc.tf_words(True), want_steps=True)
    hit = r["steps"] > 0
    alpha = r["f32"][hit][:, 3].astype(np.float64)
    want = tf_expected(val / 255.0)[None, :] * alpha[:, None]
    assert np.abs(r["f32"][hit][:, :3] - want).max() < 1e-4
    # TF off: colour = sampled rgb
    r = orc.render(vox, sphere_cam(orc, 1, 3), 16, 16, attenuation=3.0, tf=orc.tf_words(False), want_steps=True)
    alpha = r["f32"][hit][:, 3].astype(np.float64)
    want = (np.array([7, 99, 200]) / 255.0)[None, :] * alpha[:, None]
    assert np.abs(r["f32"][hit][:, :3] - want).max() < 1e-4


def test_non_cubic_box(orc):
    dims, bmin, bsz, buf = orc.grid_uniform((64, 32, 16), attenuation=2.5)
    assert list(dims) == [64, 32, 16, 64]                  # dimensions.xyzx (grid.rs:274)
    assert np.allclose(bsz, [4, 2, 1, 0]) and np.allclose(bmin, [-2, -1, -0.5, 0])
    assert np.all(buf == np.float32(2.5))
    # a ray along +x through the centre crosses 4.0 world units -> 4000 steps; along z it would be 1000
    cam = orc.camera_uniform(np.array([-6.0, 0.0, 0.0], np.float32))
    hit, rd, t0, t1 = orc.ray(cam, (64, 32, 16), 32, 32, 64, 64)
    assert hit and abs((t1 - t0) - 4.0) < 2e-2 and rd[0] > 0.99


def test_camera_inside_box_keeps_negative_t_min(orc):
    cam = orc.camera_uniform(np.array([0.1, 0.05, -0.02], np.float32), target=(1.0, 0.3, 0.2))
    geo = orc.geometry(cam, (8, 8, 8), 16, 16)
    assert (geo[..., 6] == 1).all()           # every ray hits
    assert (geo[..., 3] < 0).all()            # t_min is not clamped (wgsl:137)
    assert (geo[..., 4] > 0).all()
    # samples behind the eye are taken: the iteration count covers t_min..t_max, not 0..t_max
    chord = (geo[..., 4] - geo[..., 3]).astype(np.float64)
    assert np.all(np.abs(geo[..., 7] - np.ceil(chord / 0.001)) <= 2)
    assert np.array_equal(geo[..., 5], geo[..., 7]) or (geo[..., 7] - geo[..., 5]).max() <= 2


def test_max_step_amount_caps_long_rays(orc):
    # box 1 x 1 x 8: a ray down the long axis would need 8000 steps, the loop stops at 5000 (wgsl:96,139)
    cam = orc.camera_uniform(np.array([0.0, 0.0, 9.0], np.float32))
    geo = orc.geometry(cam, (4, 4, 32), 9, 9)
    assert geo[4, 4, 6] == 1 and geo[4, 4, 7] == 5000
    vox = solid((4, 4, 32), (255, 255, 255, 255))
    r = orc.render(vox, cam, 9, 9, want_steps=True)
    assert r["steps"][4, 4] == 5000
    assert abs(r["f32"][4, 4, 3] - (1 - np.exp(-5.0))) < 2e-4


def test_early_termination_only_when_alpha_reaches_one(orc):
    vox = solid((4, 4, 4), (255, 255, 255, 255))
    cam = sphere_cam(orc, 1, 3)
    full = orc.geometry(cam, (4, 4, 4), 16, 16)[..., 7]
    # attenuation 1: alpha <= 1 - exp(-1.8) < 1 -> never terminates early (SURVEY a-7)
    r1 = orc.render(vox, cam, 16, 16, attenuation=1.0, want_steps=True)
    assert np.array_equal(r1["steps"], full.astype(np.uint32))
    # attenuation 100 (the GUI slider's maximum, state.rs:365): alpha_src <= 0.095, and alpha + (1-alpha)*alpha_src
    # stops changing in f32 once (1-alpha)*alpha_src drops under half an ulp, i.e. alpha stalls at 1 - 5*2^-24
    # and `alpha >= 1.0` (wgsl:176) never fires: inside the reference's own parameter range the early exit is dead.
    r2 = orc.render(vox, cam, 16, 16, attenuation=100.0, want_steps=True)
    assert np.array_equal(r2["steps"], full.astype(np.uint32))
    assert 0.999999 < r2["f32"][..., 3].max() < 1.0
    # it does fire once one step alone can close the gap (alpha_src >= 0.5 <=> a*att*0.001 >= ln 2)
    r3 = orc.render(vox, cam, 16, 16, attenuation=1000.0, want_steps=True)
    long_rays = full > 100
    assert (r3["steps"][long_rays] < 40).all()
    assert (r3["f32"][long_rays][:, 3] == 1.0).all()
    assert (r3["rgba8"][long_rays] == 255).all()


def test_unorm8_store_rounds_to_nearest(orc):
    vox = solid((4, 4, 4), (200, 100, 50, 180))
    r = orc.render(vox, sphere_cam(orc, 2, 3), 32, 32, attenuation=4.0)
    want = np.floor(np.clip(r["f32"], 0, 1) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
    assert np.array_equal(r["rgba8"], want)


def test_dat_quantisation(orc):
    # UCHAR is the identity for all 256 values, replicated to RGBA (dat.rs:72-74)
    q = orc.quantize_dat(np.arange(256, dtype=np.uint8), 1)
    assert np.array_equal(q, np.repeat(np.arange(256, dtype=np.uint8)[:, None], 4, 1))
    # USHORT: bytes combined big-endian (dat.rs:64-70), truncation: only 65535 maps to 255
    vals = np.array([0, 1, 256, 257, 65534, 65535, 0x1234, 0x8000], dtype=np.uint16)
    raw = vals.astype(">u2").view(np.uint8)
    q = orc.quantize_dat(raw, 2)[:, 0]
    want = np.floor(vals.astype(np.float64) / 65535.0 * 255.0).astype(np.uint8)
    assert np.array_equal(q, want) and q[4] == 254 and q[5] == 255
    # little-endian reading would give different values
    assert not np.array_equal(q, np.floor(vals.byteswap().astype(np.float64) / 65535.0 * 255.0).astype(np.uint8))


def test_netcdf_quantisation(orc):
    # (v*255.0) as u8: truncate toward zero, saturate, NaN -> 0 (netcdf.rs:54-57); axis order color(c,z,y,x)
    data = np.zeros((4, 2, 3, 5), np.float32)
    data[0, 1, 2, 4] = 0.5        # r of voxel x=4,y=2,z=1
    data[1, 0, 0, 0] = 0.999      # 254.745 -> 254
    data[2, 0, 0, 0] = 1.7        # saturates
    data[3, 0, 0, 0] = -0.3       # -> 0
    data[3, 1, 0, 0] = np.nan     # -> 0
    data[3, 1, 1, 1] = 1.0
    q = orc.quantize_netcdf(data)
    assert q.shape == (2, 3, 5, 4)
    assert q[1, 2, 4, 0] == 127 and q[0, 0, 0, 1] == 254 and q[0, 0, 0, 2] == 255
    assert q[0, 0, 0, 3] == 0 and q[1, 0, 0, 3] == 0 and q[1, 1, 1, 3] == 255


def test_sphere_positions_and_view_order(orc):
    views = orc.sphere_views(8, 8)
    assert len(views) == 56 and views[0] == (0, 1) and views[7] == (7, 1) and views[8] == (0, 2) and views[-1] == (7, 7)
    assert len(orc.sphere_views(16, 17)) == 256 and len(orc.sphere_views(8, 9)) == 64
    for ix, iy in [(0, 1), (3, 4), (7, 7)]:
        th, ph = ix / 8 * 2 * np.pi, iy / 8 * np.pi
        want = np.array([np.sin(ph) * np.cos(th), np.cos(ph), -np.sin(ph) * np.sin(th)]) * 100.0 + np.array([1, 2, 3])
        got = orc.sphere_position(ix, iy, 8, 8, 100.0, origin=(1, 2, 3))
        assert np.allclose(got, want, atol=2e-5 * 100)


def test_camera_uniform(orc):
    for ix, iy, rad in [(1, 3, 2.5), (6, 5, 100.0), (0, 1, 10.0)]:
        pos = orc.sphere_position(ix, iy, 8, 8, rad)
        cam, basis = orc.camera_uniform(pos, aspect=1.0, with_basis=True)
        right, up, front = basis
        assert np.allclose(cam[:4], [pos[0], pos[1], pos[2], 0])
        assert np.allclose(front, -pos / np.linalg.norm(pos), atol=1e-5)   # looks at the origin
        assert abs(right @ up) < 1e-5 and abs(right @ front) < 1e-5 and right[1] == pytest.approx(0, abs=1e-5)
        vp = cam[4:20].reshape(4, 4).T.astype(np.float64)
        ivp = cam[20:36].reshape(4, 4).T.astype(np.float64)
        assert np.abs(vp @ ivp - np.eye(4)).max() < 5e-3  # far = 1000 makes VP ill-conditioned in f32
        # perspective_rh: fovy 45 deg, near 0.1, far 1000, depth 0..1 (camera.rs:19-21,28)
        proj_h = 1 / np.tan(np.radians(45) / 2)
        view_rows = np.stack([right, up, -front])
        assert np.allclose(vp[:2, :3], proj_h * view_rows[:2], atol=1e-4)
        # the shader's ray through the image centre: inverse_view_proj * (0,0,1,1) has NO perspective divide
        # and no "- ro" (wgsl:101-107), so it is front + campos/zfar, not front (SURVEY a-1 quirk)
        w = ivp @ np.array([0, 0, 1, 1.0])
        want = front.astype(np.float64) + pos / 1000.0
        assert np.allclose(w[:3] / np.linalg.norm(w[:3]), want / np.linalg.norm(want), atol=2e-4)


def test_pixel_grid_orientation(orc):
    # framebuffer row 0 is the top of the image: uv.y = 1 - (py+.5)/H; +x to the right (mod.rs:8-12)
    cam, basis = orc.camera_uniform(np.array([0, 0, 3.0], np.float32), with_basis=True)
    right, up, _ = basis
    _, rd_tl, _, _ = orc.ray(cam, (4, 4, 4), 0, 0, 64, 64)
    _, rd_br, _, _ = orc.ray(cam, (4, 4, 4), 63, 63, 64, 64)
    assert rd_tl @ up > 0 and rd_tl @ right < 0
    assert rd_br @ up < 0 and rd_br @ right > 0
    # aspect: a wide image spreads rays further in x
    wide = orc.camera_uniform(np.array([0, 0, 3.0], np.float32), aspect=2.0)
    _, rd_w, _, _ = orc.ray(wide, (4, 4, 4), 0, 32, 128, 64)
    _, rd_s, _, _ = orc.ray(cam, (4, 4, 4), 0, 32, 64, 64)
    assert abs(rd_w[0]) > abs(rd_s[0])
