"""The image metrics of scripts/similarity.py (f-4), PINNED against scikit-image itself: tests/golden/similarity/ holds uint8 image
pairs and what scikit-image 0.18.3 returns for the calls the reference's script makes (tools/make_similarity_golden.py, run under the
Anaconda interpreter this image carries; the tests' own interpreter has no scikit-image).  Here: the oracle's restatement and the
host-side formulas against those values.  The product's GPU reductions meet the same file in tests/test_gpu_eval.py."""
import json
import math
import os

import numpy as np
import pytest

import oracle
from volume_renderer_amd import similarity

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "similarity")


def load_pairs():
    with open(os.path.join(GOLDEN, "skimage_golden.json")) as f:
        meta = json.load(f)
    z = np.load(os.path.join(GOLDEN, "pairs.npz"))
    return meta, [(p, z[p["name"] + "/gt"], z[p["name"] + "/approx"]) for p in meta["pairs"]]


META, PAIRS = load_pairs()


def test_provenance():
    assert META["skimage"] == "0.18.3" and len(PAIRS) == 16
    assert any(p["ssim"] is not None and p["ssim"] < 0 for p, _, _ in PAIRS) and any(p["ssim"] == 1.0 for p, _, _ in PAIRS)


@pytest.mark.parametrize("k", range(len(PAIRS)), ids=[p["name"] for p, _, _ in PAIRS])
def test_restatement_against_scikit_image(k):
    p, gt, ap = PAIRS[k]
    assert list(gt.shape) == p["shape"] and gt.dtype == np.uint8
    assert similarity.data_range_u8(gt, ap) == p["data_range"]             # similarity.py:64, uint8 wrap-around included
    ssd = int(((gt.astype(np.int64) - ap.astype(np.int64)) ** 2).sum())
    mse, rmse, psnr = similarity.metrics_from_ssd(ssd, gt.size)
    assert mse == p["mse"] and rmse == math.sqrt(p["mse"])
    if p["psnr"] == "inf":
        assert math.isinf(psnr)
    else:
        assert abs(psnr - p["psnr"]) < 1e-12
    if p["ssim"] is not None:
        assert abs(oracle.ssim_u8(gt, ap, float(p["data_range"])) - p["ssim"]) < 1e-12
