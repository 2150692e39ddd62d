"""Committed golden vectors (tests/golden/*.npz, written by tools/make_golden.py).

CPU: the oracle still reproduces them (pins the oracle against accidental edits).
GPU: the HIP path, called through the C ABI, reproduces them within the parity gate
(max |delta| <= 1 LSB per channel and normalised RMSE < 1e-3)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, image_metrics

FILES = sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))


def test_golden_files_exist():
    assert len(FILES) >= 6


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_oracle_reproduces_golden(orc, path):
    g = np.load(path)
    r = orc.render(g["voxels"], g["camera"], int(g["width"]), int(g["height"]), attenuation=float(g["attenuation"]),
                   tf=orc.tf_words(bool(g["use_tf"])), want_steps=True)
    assert np.array_equal(r["steps"], g["steps"])
    assert np.array_equal(r["rgba8"], g["rgba8"])
    assert np.array_equal(r["f32"], g["f32"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_hip_path_reproduces_golden(rm, vr, path):
    g = np.load(path)
    W, H = int(g["width"]), int(g["height"])
    grid = rm.voxel_grid
    grid.upload_rgba8(g["voxels"])
    grid.attenuation = float(g["attenuation"])
    grid.update_voxel_grid_buffer()
    grid.set_transfer_function_active(bool(g["use_tf"]))
    grid.update_transfer_function_buffer()
    rm.set_option(vr.OPT_DIAGNOSTICS, 1)
    img = vr.Screenshotter(rm, W, H)
    rm.draw(vr.uniform_from_array(g["camera"]), img)
    got = img.read()
    f32, iters = img.diagnostics()
    m = image_metrics(got, g["rgba8"])
    assert m["max"] <= 1 and m["rmse_norm"] < 1e-3, m
    assert np.array_equal(iters, g["steps"])
    assert np.abs(f32 - g["f32"]).max() < 1e-4
    img.close()
