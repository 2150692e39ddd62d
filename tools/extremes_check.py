#!/usr/bin/env python3
"""Extreme shapes, LDS-staged march (variant 1) against the global-gather baseline (variant 0), both production kernels:
very long / thin volumes, very wide / tall / large images, far and near cameras.  8-bit images must agree to 1 LSB."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import volume_renderer_amd as vr  # noqa: E402
from volume_renderer_amd import synth  # noqa: E402

CASES = [  # dims (X, Y, Z), image (W, H), radius, scalar
    ((5000, 1, 1), (300, 40), 4000.0, False), ((1, 1, 5000), (64, 300), 3000.0, True), ((1, 3000, 2), (200, 200), 2500.0, False),
    ((700, 3, 5), (1, 4000), 300.0, True), ((9, 9, 9), (4000, 1), 2.0, False), ((64, 64, 64), (4096, 4096), 2.5, True),
    ((300, 300, 300), (37, 23), 1.1, False), ((300, 300, 300), (16, 16), 60.0, False), ((2, 2, 2), (1024, 1024), 1.6, False),
    ((129, 65, 33), (640, 480), 0.2, True), ((1025, 7, 7), (512, 64), 200.0, True),
]
rm = vr.RayMarcher(0, 64, 64)
bad = 0
for dims, (W, H), radius, scalar in CASES:
    X, Y, Z = dims
    rng = np.random.default_rng(X + Y + Z)
    if scalar:
        rm.voxel_grid.upload_scalar(rng.integers(0, 256, (Z, Y, X), dtype=np.uint8), dims)
    else:
        rm.voxel_grid.upload_rgba8(rng.integers(0, 256, (Z, Y, X, 4), dtype=np.uint8))
    rm.voxel_grid.attenuation = 20.0 / max(1.0, max(dims) / min(dims) / 50.0)
    rm.voxel_grid.update_voxel_grid_buffer()
    cam = vr.Camera(W / H)
    ctl = vr.CameraSphereController(8, 8, radius=radius)
    worst = 0
    for ix, iy in [(1, 3), (6, 5), (0, 4), (2, 1)]:
        ctl.current_index_x, ctl.current_index_y = ix, iy
        ctl.update_position(cam)
        u = cam.uniform()
        imgs = []
        for k in (0, 1):
            rm.set_option(vr.OPT_KERNEL, k)
            s = vr.Screenshotter(rm, W, H)
            rm.draw(u, s)
            imgs.append(s.read().astype(np.int16))
            s.close()
        worst = max(worst, int(np.abs(imgs[0] - imgs[1]).max()))
    ok = worst <= 1
    bad += 0 if ok else 1
    print(f"dims {dims} image {W}x{H} radius {radius} {'scalar' if scalar else 'rgba'}: max |d| = {worst} LSB, max alpha {int(imgs[1][..., 3].max())} {'OK' if ok else 'FAIL'}", flush=True)
rm.close()
sys.exit(1 if bad else 0)
