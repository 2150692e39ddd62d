"""Golden image-metric values from scikit-image ITSELF, for tests/test_similarity_golden_cpu.py and tests/test_gpu_eval.py.

scripts/similarity.py:59-65 of the reference computes its table with skimage.metrics (mean_squared_error, peak_signal_noise_ratio,
structural_similarity); oracle.ssim_u8 and vrm_ssim_u8 restate those from scikit-image's published source.  The interpreter the tests
run under has no scikit-image, but this image carries an Anaconda Python 3.9 with scikit-image 0.18.3 under /opt/conda: this script --
own code; it calls the LIBRARY with the arguments the reference's script passes, it does not run the reference's script -- makes
deterministic uint8 image pairs and records what the library returns for each.

    /opt/conda/bin/python3.9 tools/make_similarity_golden.py      ->  tests/golden/similarity/pairs.npz, skimage_golden.json

(scikit-image 0.18.3 predates `channel_axis`; it takes `multichannel=True`, which the reference's call passes as well, and ignores the
other keyword: checked here by calling it both ways.)"""
import json
import os
import warnings

import numpy as np
import skimage
import skimage.metrics

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "similarity")


def pairs():
    rng = np.random.default_rng(20240607)
    out = []

    def smooth(h, w, c, k):
        yy, xx = np.mgrid[0:h, 0:w]
        ch = [127 + 100 * np.sin(xx * 0.3 + k) * np.cos(yy * 0.25 + 2 * k), 127 + 90 * np.cos(xx * 0.21 * (k + 1) + 0.5 * k),
              80 + 60 * np.sin((xx + yy) * 0.15 + k * k), 255.0 * (np.hypot(xx - w / 2, yy - h / 2) < 0.3 * min(h, w) + k % 3)]
        return np.stack(ch[:c] if c < 4 else ch, -1)

    for k, (h, w, c) in enumerate([(20, 24, 4), (7, 7, 4), (9, 31, 3), (33, 8, 4), (48, 40, 4), (16, 16, 3), (25, 25, 4), (12, 50, 4)]):
        base = smooth(h, w, c, k)
        gt = np.clip(base, 0, 255).astype(np.uint8)
        ap = np.clip(base + rng.normal(0, 1 + k, base.shape) * (rng.random(base.shape) < 0.3), 0, 255).astype(np.uint8)
        out.append((f"smooth_{h}x{w}x{c}", gt, ap))
    # rendering-like: mostly transparent black with a blob, approximations one LSB off here and there (what the parity contract allows)
    for k in range(3):
        h, w = 40, 36
        yy, xx = np.mgrid[0:h, 0:w]
        r = np.hypot(xx - 18 - k, yy - 20 + k)
        a = np.clip(255 * np.exp(-(r / (6 + k)) ** 2), 0, 255)
        gt = np.stack([a * 0.9, a * 0.5, a * 0.2, a], -1).astype(np.uint8)
        ap = gt.astype(np.int16) + (rng.random(gt.shape) < 0.02) * rng.choice([-1, 1], gt.shape)
        out.append((f"blob_lsb_{k}", gt, np.clip(ap, 0, 255).astype(np.uint8)))
    noise = rng.integers(0, 256, (15, 17, 4), dtype=np.uint8)
    out.append(("noise_vs_noise", noise, rng.integers(0, 256, (15, 17, 4), dtype=np.uint8)))
    out.append(("identical", noise, noise.copy()))
    out.append(("constant", np.full((10, 11, 4), 37, np.uint8), np.full((10, 11, 4), 37, np.uint8)))
    # the script's data range `img_gt.max() - img_approx.min()` is uint8 arithmetic: it wraps when the approximation is brighter throughout
    out.append(("data_range_wraps", rng.integers(0, 40, (14, 14, 4), dtype=np.uint8), rng.integers(60, 200, (14, 14, 4), dtype=np.uint8)))
    out.append(("data_range_small", rng.integers(100, 110, (14, 14, 3), dtype=np.uint8), rng.integers(98, 112, (14, 14, 3), dtype=np.uint8)))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    arrays, table = {}, []
    for name, gt, ap in pairs():
        arrays[name + "/gt"], arrays[name + "/approx"] = gt, ap
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with np.errstate(all="ignore"):
                data_range = gt.max() - ap.min()                                  # similarity.py:64, as numpy computes it on uint8
                mse = float(skimage.metrics.mean_squared_error(gt, ap))          # :61
                psnr = float(skimage.metrics.peak_signal_noise_ratio(gt, ap))    # :63
            if int(data_range) == 0:
                ssim = None   # the script's call divides by zero constants; the library's answer (nan) is not a contract
            else:
                ssim = float(skimage.metrics.structural_similarity(gt, ap, data_range=data_range, multichannel=True))       # :65
                both = float(skimage.metrics.structural_similarity(gt, ap, data_range=data_range, channel_axis=-1, multichannel=True))
                assert ssim == both
        table.append({"name": name, "shape": list(gt.shape), "data_range": int(data_range), "mse": mse,
                      "psnr": psnr if np.isfinite(psnr) else "inf", "ssim": ssim})
        print(table[-1])
    np.savez_compressed(os.path.join(OUT, "pairs.npz"), **arrays)
    with open(os.path.join(OUT, "skimage_golden.json"), "w") as f:
        json.dump({"generator": "tools/make_similarity_golden.py", "skimage": skimage.__version__, "numpy": np.__version__,
                   "calls": "mean_squared_error(gt, ap); peak_signal_noise_ratio(gt, ap); structural_similarity(gt, ap, "
                            "data_range=gt.max() - ap.min(), multichannel=True)   (scripts/similarity.py:61-65)",
                   "pairs": table}, f, indent=1)


if __name__ == "__main__":
    main()
