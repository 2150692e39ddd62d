#!/usr/bin/env python3
"""End-to-end rehearsal of the sort-last path (bench.py --config C5 / SortLastRenderer) with several ranks on ONE GPU:
every rank marches its z slab with the HIP kernels on GPU 0, the row-band all-to-all runs over gloo on host tensors,
every rank composites its band on the GPU, rank 0 gathers the bands and compares the images with its own render of
the whole volume.  Launch:  python -m torch.distributed.run --nproc-per-node G --master-addr 127.0.0.1 tools/rehearse_sortlast.py
Exit code 0 iff every view agrees to 1 LSB (and the f32-free path found something to draw)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import volume_renderer_amd as vr  # noqa: E402
from volume_renderer_amd import sortlast, synth  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dims, W, H = (96, 80, 128), 200, 136
    scalar = os.environ.get("REHEARSE_SCALAR", "1") == "1"
    rm = vr.RayMarcher(0, W, H)
    slr = sortlast.SortLastRenderer(rm, dims)
    if scalar:
        full = synth.blobs_scalar_u8(dims)                      # (Z, Y, X)
        slr.upload_scalar_u8(full[slr.s0:slr.s1])
    else:
        full = synth.random_rgba8(dims, 5)                      # (Z, Y, X, 4)
        slr.upload_rgba8(full[slr.s0:slr.s1])
    rm.voxel_grid.attenuation = 6.0
    rm.voxel_grid.update_voxel_grid_buffer()
    cam = vr.Camera(W / H)
    csp = vr.CameraSphereController(5, 6, (0.0, 0.0, 0.0), 2.8)
    cams = []
    for ix, iy in vr.sphere_views(6, 5):
        csp.current_index_x, csp.current_index_y = ix, iy
        csp.update_position(cam)
        cams.append(cam.uniform())
    band, row0, rows = slr.render(cams, W, H, dev, host_exchange=True)
    mine = band.cpu()
    gathered = [None] * world
    dist.gather_object((row0, rows, mine.numpy()), gathered if rank == 0 else None, dst=0)
    ok = True
    if rank == 0:
        img = np.zeros((len(cams), H, W, 4), np.uint8)
        for r0, k, b in gathered:
            img[:, r0:r0 + k] = b
        ref = vr.RayMarcher(0, W, H)
        if scalar:
            ref.voxel_grid.upload_scalar(full, dims)
        else:
            ref.voxel_grid.new(dims)
            ref.voxel_grid.upload_rgba8(full)
        ref.voxel_grid.attenuation = 6.0
        ref.voxel_grid.update_voxel_grid_buffer()
        worst = 0
        for i, c in enumerate(cams):
            s = vr.Screenshotter(ref, W, H)
            ref.draw(c, s)
            want = s.read().astype(np.int32)
            s.close()
            worst = max(worst, int(np.abs(want - img[i].astype(np.int32)).max()))
        drawn = int(img[..., 3].max())
        ok = worst <= 1 and drawn > 20
        print(f"sort-last rehearsal: {world} ranks on one GPU, {'scalar' if scalar else 'RGBA'} {dims}, {len(cams)} views {W}x{H}: "
              f"max |d| = {worst} LSB vs the unpartitioned render, max alpha {drawn} -> {'OK' if ok else 'FAIL'}", flush=True)
        ref.close()
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, src=0)
    rm.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
