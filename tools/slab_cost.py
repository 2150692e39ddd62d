"""What a rank of a sort-last render pays for marching the whole lattice while owning 1/G of it: kernel time of one
slab's partial for G = 1, 2, 4, 8 (one GPU plays rank 0 and a middle rank).  usage: python tools/slab_cost.py [dim] [res]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import volume_renderer_amd as vr  # noqa: E402
from volume_renderer_amd import sortlast, synth  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dims = (D, D, 2 * D)
rm = vr.RayMarcher(0, R, R)
cam = vr.Camera(1.0)
csp = vr.CameraSphereController(5, 8, (0, 0, 0), 2.5)
cams = []
for ix, iy in vr.sphere_views(8, 5)[::4]:
    csp.current_index_x, csp.current_index_y = ix, iy
    csp.update_position(cam)
    cams.append(cam.uniform())
part = torch.empty((len(cams), R, R, 4), dtype=torch.float32, device="cuda:0")
for G in (1, 2, 4, 8):
    ranges = sortlast.slab_ranges(dims[2], G)
    for which in sorted({0, G // 2}):
        z0, z1 = ranges[which]
        s0, s1 = rm.voxel_grid.new_slab(dims, z0, z1)
        rm.voxel_grid.upload_scalar(synth.blobs_scalar_u8_slices(dims, s0, s1).reshape(-1), (D, D, s1 - s0), 1)
        rm.synchronize()
        for rep in range(2):
            t0 = time.perf_counter()
            rm.draw_partial(cams, R, R, part.data_ptr())
            rm.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
        print(f"G={G} slab {which} [{z0},{z1}): {ms / len(cams):.3f} ms/view for this rank's partial ({len(cams)} views {R}^2)", flush=True)
