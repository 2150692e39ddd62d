"""GPU reductions of the evaluation loop through the C ABI, against the oracle / numpy."""
import io
import math
import os

import numpy as np
import pytest

from conftest import sphere_cam
from volume_renderer_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dims", [(8, 8, 8), (33, 17, 9), (64, 48, 40)])
@pytest.mark.parametrize("density_only", [True, False])
def test_compare_volume_matches_oracle(rm1, orc, dims, density_only):
    X, Y, Z = dims
    rng = np.random.default_rng(11)
    vox = synth.random_rgba8(dims, 7)                             # (Z, Y, X, 4)
    data = rng.random((4, Z, Y, X), dtype=np.float32)
    g = rm1.voxel_grid
    g.new(dims)
    g.upload_rgba8(vox)
    got = rm1.compare_to_netcdf_rmse(data, density_only)
    r32, r64 = orc.compare_volume(vox, data, density_only=density_only)
    # same f32 terms, f64 sum: equal to the oracle's f64 variant up to the final f32 rounding
    assert abs(got - r64) <= 2e-7 * r64
    # the reference's sequential f32 sum agrees to f32 summation error at these sizes
    assert abs(got - r32) <= 2e-4 * r32


def test_compare_volume_uses_context_transfer_function(rm1, orc, vr):
    dims = (16, 16, 16)
    vox = synth.random_rgba8(dims, 2)
    data = np.random.default_rng(3).random((4, 16, 16, 16), dtype=np.float32)
    g = rm1.voxel_grid
    g.new(dims)
    g.upload_rgba8(vox)
    cols = ((0.9, 0.1, 0.2, 1.0), (0.3, 0.8, 0.1, 1.0), (0.0, 0.5, 1.0, 1.0))
    g.transfer_function_colors.color_a[:] = cols[0]
    g.transfer_function_colors.color_b[:] = cols[1]
    g.transfer_function_colors.color_c[:] = cols[2]
    g.update_transfer_function_buffer()
    got = rm1.compare_to_netcdf_rmse(data, False)
    want = orc.compare_volume(vox, data, tf_colors=cols, density_only=False)[1]
    assert abs(got - want) <= 2e-7 * want
    with pytest.raises(ValueError):
        rm1.compare_to_netcdf_rmse(data[:, :8], False)


@pytest.mark.parametrize("n", [1, 3, 4, 5, 1023, 1000003])
def test_ssd_u8_exact(rm1, n):
    rng = np.random.default_rng(n)
    a = rng.integers(0, 256, n, dtype=np.uint8)
    b = rng.integers(0, 256, n, dtype=np.uint8)
    want = int(((a.astype(np.int64) - b.astype(np.int64)) ** 2).sum())
    assert rm1.ssd_u8(a, b) == want
    assert rm1.ssd_u8(a, a) == 0


def test_ssd_u8_worst_case_no_overflow(rm1):
    n = 1 << 22
    a = np.zeros(n, np.uint8)
    b = np.full(n, 255, np.uint8)
    assert rm1.ssd_u8(a, b) == n * 255 * 255


def test_image_ssd_matches_readback(rm1, vr, orc):
    dims = (32, 32, 32)
    g = rm1.voxel_grid
    g.new(dims)
    g.upload_rgba8(synth.quantize_rgba8(synth.blobs_rgba_f32(dims)))
    cam = vr.uniform_from_array(sphere_cam(orc, 1, 3))
    a, b = vr.Screenshotter(rm1, 100, 60), vr.Screenshotter(rm1, 100, 60)   # width*4 is not a multiple of the 256-byte pitch
    g.attenuation = 1.0
    g.update_voxel_grid_buffer()
    rm1.draw(cam, a)
    g.attenuation = 3.0
    g.update_voxel_grid_buffer()
    rm1.draw(cam, b)
    ssd, cnt = rm1.image_ssd(a, b)
    ia, ib = a.read().astype(np.int64), b.read().astype(np.int64)
    assert cnt == ia.size == 100 * 60 * 4
    assert ssd == int(((ia - ib) ** 2).sum()) and ssd > 0
    c = vr.Screenshotter(rm1, 64, 64)
    with pytest.raises(vr.VrmError):
        rm1.image_ssd(a, c)
    for im in (a, b, c):
        im.close()


def test_similarity_cli_table(rm1, vr, orc, tmp_path):
    """Two tiny exports differing in attenuation -> the table of scripts/similarity.py, checked against numpy on Pillow's decode."""
    Image = pytest.importorskip("PIL.Image")
    from volume_renderer_amd import similarity
    dims = (24, 24, 24)
    r = vr.RayMarcher(0, 48, 48)
    r.voxel_grid.new(dims)
    r.voxel_grid.upload_rgba8(synth.quantize_rgba8(synth.blobs_rgba_f32(dims)))
    d1, d2 = os.path.join(tmp_path, "gt"), os.path.join(tmp_path, "approx")
    r.voxel_grid.attenuation = 2.0
    r.voxel_grid.update_voxel_grid_buffer()
    r.export_sphere(4, 3, 2.5, d1)
    r.voxel_grid.attenuation = 4.0
    r.voxel_grid.update_voxel_grid_buffer()
    r.export_sphere(4, 3, 2.5, d2)
    r.close()
    out = io.StringIO()
    avg = similarity.compare_dirs(rm1, d1, d2, x_div=4, y_div=3, out=out)
    lines = out.getvalue().strip().split("\n")
    assert lines[0] == "filename, mse, rmse, psnr, ssim" and len(lines) == 1 + 4 * 2 + 1
    tot = np.zeros(4)
    k = 1
    for ix in range(4):                                           # itertools.product(range(x), range(1, y)) order
        for iy in range(1, 3):
            name = f"{ix}-{iy}.png"
            ga = np.array(Image.open(os.path.join(d1, name))).astype(np.float64)
            gb = np.array(Image.open(os.path.join(d2, name))).astype(np.float64)
            mse = float(np.mean((ga - gb) ** 2))                  # skimage.metrics.mean_squared_error
            f = lines[k].split(", ")
            assert f[0] == name and float(f[1]) == mse and float(f[2]) == math.sqrt(mse)
            assert abs(float(f[3]) - 10 * math.log10(255.0 ** 2 / mse)) < 1e-9
            ua, ub = ga.astype(np.uint8), gb.astype(np.uint8)
            ssim = orc.ssim_u8(ua, ub, similarity.data_range_u8(ua, ub))
            assert abs(float(f[4]) - ssim) < 1e-12 and 0.0 < ssim < 1.0
            tot += (mse, math.sqrt(mse), 10 * math.log10(255.0 ** 2 / mse), ssim)
            k += 1
    assert np.allclose(avg, tot / 8, rtol=1e-12, atol=1e-12)
    assert lines[-1].startswith("average, ")


@pytest.mark.parametrize("shape", [(7, 7), (7, 7, 4), (20, 31, 4), (33, 9, 3), (64, 48, 1), (257, 130, 2)])
def test_ssim_u8_matches_oracle(rm1, orc, shape):
    """vrm_ssim_u8 (exact integer window sums) against the scipy restatement (running f64 means): 1e-12."""
    rng = np.random.default_rng(sum(shape))
    a = rng.integers(0, 256, shape, dtype=np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-40, 41, shape), 0, 255).astype(np.uint8)
    for dr in (255, 180):
        got, want = rm1.ssim_u8(a, b, dr), orc.ssim_u8(a, b, dr)
        assert abs(got - want) < 1e-12, (got, want)
    assert rm1.ssim_u8(a, a, 255) == 1.0


def test_metrics_against_scikit_image_golden(rm1):
    """vrm_ssd_u8 / vrm_ssim_u8 against what scikit-image 0.18.3 itself returned for the calls of scripts/similarity.py:61-65
    (tests/golden/similarity, tools/make_similarity_golden.py): the pin of f-4's image metrics against the library, not a restatement."""
    from test_similarity_golden_cpu import PAIRS
    from volume_renderer_amd import similarity
    for p, gt, ap in PAIRS:
        mse, _, psnr = similarity.metrics_from_ssd(rm1.ssd_u8(gt, ap), gt.size)
        assert mse == p["mse"], p["name"]
        assert (math.isinf(psnr) and p["psnr"] == "inf") or abs(psnr - p["psnr"]) < 1e-12, p["name"]
        if p["ssim"] is not None:
            assert abs(rm1.ssim_u8(gt, ap, similarity.data_range_u8(gt, ap)) - p["ssim"]) < 1e-12, p["name"]


def test_ssim_u8_rendered_images_and_errors(rm1, vr, orc):
    dims = (32, 32, 32)
    g = rm1.voxel_grid
    g.new(dims)
    g.upload_rgba8(synth.quantize_rgba8(synth.blobs_rgba_f32(dims)))
    cam = vr.uniform_from_array(sphere_cam(orc, 1, 3))
    imgs = []
    for att in (1.0, 3.0):
        s = vr.Screenshotter(rm1, 96, 80)
        g.attenuation = att
        g.update_voxel_grid_buffer()
        rm1.draw(cam, s)
        imgs.append(s.read())
        s.close()
    from volume_renderer_amd import similarity
    dr = similarity.data_range_u8(imgs[0], imgs[1])
    got, want = rm1.ssim_u8(imgs[0], imgs[1], dr), orc.ssim_u8(imgs[0], imgs[1], dr)
    assert 0.0 < got < 1.0 and abs(got - want) < 1e-12
    with pytest.raises(vr.VrmError):
        rm1.ssim_u8(np.zeros((6, 20, 4), np.uint8), np.zeros((6, 20, 4), np.uint8), 255)      # smaller than the window
    with pytest.raises(ValueError):
        rm1.ssim_u8(np.zeros((8, 8), np.uint8), np.zeros((8, 9), np.uint8), 255)
