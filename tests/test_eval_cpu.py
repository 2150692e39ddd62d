"""CPU side of the evaluation helpers: the oracle's restatement of compare/netcdf.rs against known
answers, the PNG reader (host code of libvrm.so, no GPU needed) against Pillow, the metric formulas."""
import math
import os

import numpy as np
import pytest

from volume_renderer_amd import synth


def _tf_colour(orc, alpha):
    """grid.rs:285-287 colours through netcdf.rs:41-61 in f64 (independent of the C restatement)."""
    ca, cb, cc = (np.array(c[:3], dtype=np.float64) for c in orc.DEFAULT_TF_COLORS)
    if alpha < 0.5:
        t = alpha / 0.5
        return ca * (1 - t) + cb * t
    t = (alpha - 0.5) / 0.5
    return cb * (1 - t) + cc * t


def test_compare_volume_identical_is_zero(orc):
    vox = synth.random_rgba8((6, 5, 4), 3)                       # (Z, Y, X, 4)
    alpha = vox[..., 3].astype(np.float32) / np.float32(255.0)
    data = np.zeros((4,) + vox.shape[:3], np.float32)
    data[3] = alpha
    r32, r64 = orc.compare_volume(vox, data, density_only=True)
    assert r32 == 0.0 and r64 == 0.0
    # full mode: colours = the transfer function of alpha
    for idx in np.ndindex(*alpha.shape):
        data[(slice(0, 3),) + idx] = _tf_colour(orc, float(alpha[idx]))
    r32, r64 = orc.compare_volume(vox, data, density_only=False)
    assert r64 < 1e-7 and r32 < 1e-7


def test_compare_volume_constant_offset(orc):
    vox = np.zeros((8, 8, 8, 4), np.uint8)
    vox[..., 3] = 51                                              # alpha = 0.2
    data = np.zeros((4, 8, 8, 8), np.float32)
    data[3] = np.float32(0.2) + np.float32(0.25)
    r32, r64 = orc.compare_volume(vox, data, density_only=True)
    assert abs(r64 - 0.25) < 1e-6 and abs(r32 - 0.25) < 1e-5
    # all four channels: colour error 0 on r,g,b (data = TF colour), alpha error 0.25 -> sqrt(0.25^2 / 4)
    data[0:3] = _tf_colour(orc, 51 / 255.0).astype(np.float32)[:, None, None, None]
    r32, r64 = orc.compare_volume(vox, data, density_only=False)
    assert abs(r64 - 0.125) < 1e-6


def test_compare_volume_axis_order(orc):
    # data is indexed [c, z, y, x] against grid[(x, y, z)] (netcdf.rs:24-29): a single mismatching voxel
    vox = np.zeros((3, 4, 5, 4), np.uint8)                        # Z=3, Y=4, X=5
    data = np.zeros((4, 3, 4, 5), np.float32)
    vox[2, 1, 4, 3] = 255                                         # z=2, y=1, x=4
    data[3, 2, 1, 4] = 1.0
    assert orc.compare_volume(vox, data, density_only=True)[1] == 0.0
    data[3, 2, 1, 4] = 0.0
    data[3, 1, 2, 4] = 1.0                                        # y and z swapped: two voxels off by 1
    assert abs(orc.compare_volume(vox, data, density_only=True)[1] - math.sqrt(2.0 / 60.0)) < 1e-7


def test_png_reader_matches_pillow(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    from volume_renderer_amd import similarity
    rng = np.random.default_rng(5)
    smooth = (np.add.outer(np.arange(37), np.arange(53)) * 3 % 256).astype(np.uint8)   # makes Pillow pick non-zero filters
    cases = {
        "rgba.png": np.stack([smooth, smooth.T[:37, :53] if False else smooth[::-1], rng.integers(0, 256, smooth.shape, dtype=np.uint8),
                              np.full_like(smooth, 200)], axis=-1),
        "rgb.png": rng.integers(0, 256, (20, 31, 3), dtype=np.uint8),
        "gray.png": smooth,
        "la.png": np.stack([smooth, smooth[::-1]], axis=-1),
        "one.png": np.array([[[1, 2, 3, 4]]], np.uint8),
    }
    for name, arr in cases.items():
        path = os.path.join(tmp_path, name)
        Image.fromarray(arr).save(path, optimize=(name != "rgb.png"))
        got = similarity.png_read(path)
        want = np.array(Image.open(path))
        assert got.shape == want.shape, name
        assert np.array_equal(got, want), name
        assert np.array_equal(got, arr), name


def test_png_reader_errors(tmp_path):
    from volume_renderer_amd import VrmError, similarity
    with pytest.raises(VrmError):
        similarity.png_read(os.path.join(tmp_path, "missing.png"))
    bad = os.path.join(tmp_path, "bad.png")
    open(bad, "wb").write(b"not a png at all, but long enough to pass the size check.......")
    with pytest.raises(VrmError):
        similarity.png_read(bad)
    Image = pytest.importorskip("PIL.Image")
    p16 = os.path.join(tmp_path, "p16.png")
    Image.fromarray(np.zeros((4, 4), np.uint16)).save(p16)       # 16-bit: outside what this path writes or reads
    with pytest.raises(VrmError):
        similarity.png_read(p16)
    trunc = os.path.join(tmp_path, "trunc.png")
    ok = os.path.join(tmp_path, "ok.png")
    Image.fromarray(np.zeros((16, 16, 4), np.uint8)).save(ok)
    open(trunc, "wb").write(open(ok, "rb").read()[:-20])
    with pytest.raises(VrmError):
        similarity.png_read(trunc)


def test_metric_formulas():
    from volume_renderer_amd import similarity
    a = np.array([0, 10, 200, 255], np.uint8)
    b = np.array([3, 10, 190, 250], np.uint8)
    ssd = int(((a.astype(np.int64) - b.astype(np.int64)) ** 2).sum())
    mse, rmse, psnr = similarity.metrics_from_ssd(ssd, a.size)
    assert mse == ssd / 4 and rmse == math.sqrt(mse)
    assert abs(psnr - 10 * math.log10(255.0 ** 2 / mse)) < 1e-12   # skimage: data_range 255 for uint8 (similarity.py:63)
    assert similarity.metrics_from_ssd(0, 4)[2] == math.inf


def test_oracle_ssim_known_properties():
    """The SSIM restatement (oracle.ssim_u8; against scikit-image's own values in test_similarity_golden_cpu.py): identities that
    hold for skimage's structural_similarity by construction."""
    import oracle
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (20, 31, 4), dtype=np.uint8)
    b = rng.integers(0, 256, (20, 31, 4), dtype=np.uint8)
    assert oracle.ssim_u8(a, a, 255) == 1.0
    s = oracle.ssim_u8(a, b, 255)
    assert -1.0 <= s < 0.2                                           # independent noise
    assert s == oracle.ssim_u8(b, a, 255)                            # symmetric
    # the mean over channels of the per-channel values
    per = [oracle.ssim_u8(a[:, :, c], b[:, :, c], 255) for c in range(4)]
    assert abs(s - np.mean(per)) < 1e-15
    # two constant images: every variance is zero, S = (2 ux uy + C1) / (ux^2 + uy^2 + C1)
    x, y = np.full((9, 9), 100, np.uint8), np.full((9, 9), 140, np.uint8)
    c1 = (0.01 * 255) ** 2
    assert abs(oracle.ssim_u8(x, y, 255) - (2 * 100 * 140 + c1) / (100 ** 2 + 140 ** 2 + c1)) < 1e-12
    # a 7x7 image has exactly one whole window: the sample statistics of the image itself
    p, q = a[:7, :7, 0].astype(np.float64), b[:7, :7, 0].astype(np.float64)
    c2 = (0.03 * 255) ** 2
    cov = ((p - p.mean()) * (q - q.mean())).sum() / 48
    want = ((2 * p.mean() * q.mean() + c1) * (2 * cov + c2)) / ((p.mean() ** 2 + q.mean() ** 2 + c1) * (p.var(ddof=1) + q.var(ddof=1) + c2))
    assert abs(oracle.ssim_u8(a[:7, :7, 0], b[:7, :7, 0], 255) - want) < 1e-12


def test_similarity_data_range_is_uint8_arithmetic():
    from volume_renderer_amd import similarity
    gt = np.array([[3, 200]], np.uint8)
    approx = np.array([[5, 9]], np.uint8)
    assert similarity.data_range_u8(gt, approx) == 195
    assert similarity.data_range_u8(approx, np.array([[20, 30]], np.uint8)) == (9 - 20) % 256
