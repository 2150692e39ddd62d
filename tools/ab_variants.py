#!/usr/bin/env python3
"""A/B check of march-kernel variants: renders the same views with VRM_OPT_KERNEL = a and = b and reports the
8-bit differences (max |d|, pixels differing).  python tools/ab_variants.py --a 1 --b 0 [--dim 64 --res 200]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import volume_renderer_amd as vr  # noqa: E402
from volume_renderer_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--a", type=int, default=1)
ap.add_argument("--b", type=int, default=0)
ap.add_argument("--dim", type=int, nargs="+", default=[64])
ap.add_argument("--res", type=int, nargs="+", default=[200])
ap.add_argument("--scalar", action="store_true")
ap.add_argument("--tf", action="store_true")
ap.add_argument("--radius", type=float, default=2.5)
ap.add_argument("--xdiv", type=int, default=8)
ap.add_argument("--ydiv", type=int, default=5)
ap.add_argument("--packed-a", type=int, default=1, help="VRM_OPT_PACKED_ALPHA for side a")
ap.add_argument("--packed-b", type=int, default=1, help="VRM_OPT_PACKED_ALPHA for side b")
a = ap.parse_args()
dims = tuple(a.dim * 3) if len(a.dim) == 1 else tuple(a.dim)
W, H = (a.res * 2)[:2]
rm = vr.RayMarcher(0, W, H)
g = rm.voxel_grid
g.new(dims)
if a.scalar:
    g.upload_scalar(synth.blobs_scalar_u8(dims), dims)
else:
    g.upload_rgba8(synth.quantize_rgba8(synth.blobs_rgba_f32(dims)))
if a.tf:
    g.set_transfer_function_active(True)
    g.update_voxel_grid_buffer()
worst = 0
ndiff = 0
ctl = vr.CameraSphereController(a.ydiv, a.xdiv, radius=a.radius)
camera = vr.Camera(W / H)
for (ix, iy) in vr.sphere_views(a.xdiv, a.ydiv):
    ctl.current_index_x, ctl.current_index_y = ix, iy
    ctl.update_position(camera)
    cam = camera.uniform()
    imgs = []
    for k, pk in ((a.a, a.packed_a), (a.b, a.packed_b)):
        rm.set_option(vr.OPT_KERNEL, k)
        rm.set_option(vr.OPT_PACKED_ALPHA, pk)
        s = vr.Screenshotter(rm, W, H)
        rm.draw(cam, s)
        imgs.append(s.read().astype(np.int32))
        s.close()
    d = np.abs(imgs[0] - imgs[1])
    worst = max(worst, int(d.max()))
    ndiff += int((d.max(axis=-1) > 0).sum())
    if imgs[0][..., 3].max() == 0:
        print("view", ix, iy, "is empty")
print(f"variants {a.a} vs {a.b}: dims {dims} image {W}x{H}: max |d| = {worst} LSB, {ndiff} pixels differ")
sys.exit(0 if worst <= 1 else 1)
