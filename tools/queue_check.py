"""Tile queues (VRM_OPT_KERNEL = 3) against the one-workgroup-per-tile dispatch (4): the same tiles marched by the same code, only
handed out differently -- f32 accumulators of vrm_draw_partial must be bit-identical.    python tools/queue_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import oracle
    import volume_renderer_amd as vr
    from volume_renderer_amd import synth

    bad = 0
    for name, dims, scalar, tf, att, W, H in [("rgba", (96, 80, 112), False, False, 1.0, 200, 176), ("rgba+tf", (64, 64, 64), False, True, 30.0, 333, 90),
                                              ("scalar", (103, 96, 91), True, False, 5.0, 64, 64), ("tiny", (5, 3, 9), False, False, 50.0, 7, 5),
                                              ("rgba ert", (40, 40, 40), False, False, 900.0, 130, 130)]:
        r = vr.RayMarcher(0, W, H)
        g = r.voxel_grid
        if scalar:
            g.upload_scalar(synth.blobs_scalar_u8(dims).reshape(-1), dims, 1)
        else:
            g.upload_rgba8(synth.quantize_rgba8(synth.blobs_rgba_f32(dims)))
        g.attenuation = att
        g.update_voxel_grid_buffer()
        g.set_transfer_function_active(tf)
        g.update_transfer_function_buffer()
        cams = [vr.uniform_from_array(oracle.camera_uniform(oracle.sphere_position(ix, iy, 8, 8, rad), aspect=W / H))
                for ix, iy, rad in [(1, 3, 2.5), (4, 6, 2.5), (7, 1, 1.2), (2, 2, 0.3), (5, 4, 400.0), (6, 6, 2.5), (0, 4, 3.0)]]
        part = torch.empty((len(cams), H, W, 4), dtype=torch.float32, device="cuda:0")
        imgs = [vr.Screenshotter(r, W, H) for _ in cams]
        for skip in (1, 0):
            r.set_option(vr.OPT_SKIP_EMPTY, skip)
            res = {}
            for k in (4, 3):
                r.set_option(vr.OPT_KERNEL, k)
                part.fill_(-1.0)
                r.draw_partial(cams, W, H, part.data_ptr())
                r.synchronize()
                r.draw_batch(cams, imgs)
                res[k] = (part.cpu().numpy().copy(), np.stack([im.read() for im in imgs]))
            same = np.array_equal(res[4][0].view(np.uint32), res[3][0].view(np.uint32)) and np.array_equal(res[4][1], res[3][1])
            print(f"{name:9s} dims {dims} {W}x{H} skip {skip}: {'bit-identical' if same else 'DIFFERENT'}", flush=True)
            bad += 0 if same else 1
        for im in imgs:
            im.close()
        r.close()
    print("queue_check:", "ok" if bad == 0 else f"{bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
