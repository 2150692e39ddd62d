"""Image-metric CLI of the evaluation loop: the same table scripts/similarity.py prints for two export
directories (reference scripts/similarity.py:37-80), computed by libvrm.so.

    python -m volume_renderer_amd.similarity <dir_gt> <dir_approx> [--x-div 8] [--y-div 8] [--device 0]

Per file `<ix>-<iy>.png` (ix in 0..x_div-1, iy in 1..y_div-1, the reference's itertools.product order):
mse = mean squared error over every channel value of the two uint8 images (skimage.metrics.mean_squared_error,
f64), rmse = sqrt(mse), psnr = 10*log10(255^2 / mse) (skimage's data_range for uint8 input); then the averages.
The PNGs are decoded by vrm_png_read and the sum of squared differences is an exact integer reduction on the
GPU (vrm_ssd_u8), so mse is bit-identical to skimage's.  ssim = structural_similarity(img_gt, img_approx,
data_range = img_gt.max() - img_approx.min(), channel_axis=-1) (similarity.py:64-65; the data range is the
script's own, formed in uint8 arithmetic as numpy forms it) through vrm_ssim_u8, which restates scikit-image's
published defaults (7x7 uniform window, K1 0.01, K2 0.03, sample covariance, f64, cropped border).  Pinned against
scikit-image 0.18.3 itself (tests/golden/similarity: what the library returns for these calls on 16 image pairs; mse and
psnr equal, ssim within 1e-12) and against an independent scipy restatement.  A 2-D (gray) image counts as one channel.
"""
import argparse
import ctypes
import math
import os
import sys

import numpy as np

from . import _ffi


def png_read(path):
    """skimage.io.imread for 8-bit non-interlaced PNGs -> (h, w, channels) uint8 (channels squeezed when 1)."""
    w, h, c = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
    L = _ffi.lib()
    _ffi.check(L.vrm_png_read(os.fsencode(path), ctypes.byref(w), ctypes.byref(h), ctypes.byref(c), None, 0))
    out = np.empty((h.value, w.value, c.value), dtype=np.uint8)
    _ffi.check(L.vrm_png_read(os.fsencode(path), ctypes.byref(w), ctypes.byref(h), ctypes.byref(c), out.ctypes.data, out.size))
    return out[:, :, 0] if c.value == 1 else out


def metrics_from_ssd(ssd, count):
    """(mse, rmse, psnr) as similarity.py:61-63 forms them."""
    mse = ssd / count
    rmse = math.sqrt(mse)
    psnr = 10.0 * math.log10(255.0 * 255.0 / mse) if mse > 0 else math.inf
    return mse, rmse, psnr


def data_range_u8(img_gt, img_approx):
    """similarity.py:64: `img_gt.max() - img_approx.min()` on uint8 arrays, i.e. modulo 256 as numpy computes it."""
    return (int(img_gt.max()) - int(img_approx.min())) & 0xFF


def compare_dirs(ray_marcher, dir_gt, dir_approx, x_div=8, y_div=8, out=sys.stdout):
    """similarity.py:47-80.  Returns the (mse, rmse, psnr, ssim) averages."""
    tot = [0.0, 0.0, 0.0, 0.0]
    n = 0
    print("filename, mse, rmse, psnr, ssim", file=out)
    for ix in range(x_div):
        for iy in range(1, y_div):
            name = f"{ix}-{iy}.png"
            a = png_read(os.path.join(dir_gt, name))
            b = png_read(os.path.join(dir_approx, name))
            if a.shape != b.shape:
                raise ValueError(f"{name}: shapes differ, {a.shape} vs {b.shape}")
            m = metrics_from_ssd(ray_marcher.ssd_u8(a, b), a.size) + (ray_marcher.ssim_u8(a, b, data_range_u8(a, b)),)
            for k in range(4):
                tot[k] += m[k]
            n += 1
            print(f"{name}, {m[0]}, {m[1]}, {m[2]}, {m[3]}", file=out)
    avg = [t / n for t in tot]
    print(f"average, {avg[0]}, {avg[1]}, {avg[2]}, {avg[3]}", file=out)
    return tuple(avg)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("dir_gt")
    ap.add_argument("dir_approx")
    ap.add_argument("--x-div", type=int, default=8, help="horizontal divisions of the export (reference: 8)")
    ap.add_argument("--y-div", type=int, default=8, help="vertical divisions; files iy = 1..y_div-1 (reference: 8)")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    from . import RayMarcher
    rm = RayMarcher(a.device, 64, 64)
    try:
        compare_dirs(rm, a.dir_gt, a.dir_approx, a.x_div, a.y_div)
    finally:
        rm.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
