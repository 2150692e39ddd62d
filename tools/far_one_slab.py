"""The NaN-ray view (a camera 1000 box widths away, tools/far_one.py) through the sort-last path: every rank's slab partial of a
slab-partitioned volume must come back.  Run under a short timeout:    timeout 60 python tools/far_one_slab.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import oracle as orc  # noqa: E402
import volume_renderer_amd as vr  # noqa: E402
from volume_renderer_amd import sortlast, synth  # noqa: E402

dims = (64, 15, 64)
X, Y, Z = dims
W, H = 31, 83
rm = vr.RayMarcher(0, W, H)
pos = orc.sphere_position(3, 5, 8, 8, 1000.0)
cam36 = orc.camera_uniform(pos, target=(0.0, 0.0, 0.0), aspect=W / H)
cam = vr.uniform_from_array(cam36)
for scalar in (False, True):
    vox = synth.random_rgba8(dims, 12345)
    if scalar:
        vox = np.repeat(vox[..., 3:4], 4, 3)
    for skip in (1, 0):
        rm.set_option(vr.OPT_SKIP_EMPTY, skip)
        for z0, z1 in sortlast.slab_ranges(Z, 3):
            s0, s1 = rm.voxel_grid.new_slab(dims, z0, z1)
            if scalar:
                rm.voxel_grid.upload_scalar(np.ascontiguousarray(vox[s0:s1, :, :, 0]).reshape(-1), (X, Y, s1 - s0), 1)
            else:
                rm.voxel_grid.upload_rgba8(vox[s0:s1])
            rm.voxel_grid.attenuation = 100.0
            rm.voxel_grid.update_voxel_grid_buffer()
            part = torch.full((1, H, W, 4), -1.0, dtype=torch.float32, device="cuda:0")
            print(f"scalar {scalar} skip {skip} slab [{z0}, {z1}): drawing", flush=True)
            rm.draw_partial([cam], W, H, part.data_ptr())
            rm.synchronize()
            p = part.cpu().numpy()
            print("   done: max", float(np.nanmax(p)), "min", float(np.nanmin(p)), flush=True)
print("far_one_slab: ok")
