"""Two rays per lane (VRM_OPT_RAYS_PER_LANE = 2: 16x8-pixel wave tiles, one staged box per chunk for both rays of a lane) against
one (1): the same rays through differently cut chunks -- the f32 accumulators of vrm_draw_partial, the 8-bit images and the
diagnostics build's per-ray iteration counts must be bit-identical.    python tools/rays_check.py"""
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
    cases = [("scalar", (103, 96, 91), 5.0, 256, 192), ("odd size", (64, 64, 64), 1.0, 333, 90), ("narrow", (40, 50, 30), 30.0, 17, 130),
             ("tiny", (5, 3, 9), 50.0, 40, 24), ("ert", (40, 40, 40), 900.0, 130, 130), ("big", (256, 256, 256), 1.0, 512, 512)]
    for name, dims, att, W, H in cases:
        r = vr.RayMarcher(0, W, H)
        g = r.voxel_grid
        g.upload_scalar(synth.blobs_scalar_u8(dims).reshape(-1), dims, 1)
        g.attenuation = att
        g.update_voxel_grid_buffer()
        cams = [vr.uniform_from_array(oracle.camera_uniform(oracle.sphere_position(ix, iy, 8, 8, rad), aspect=W / H))
                for ix, iy, rad in [(1, 3, 2.5), (4, 6, 2.5), (7, 1, 1.2), (2, 2, 0.3), (6, 6, 2.5), (0, 4, 3.0)]]
        part = torch.empty((len(cams), H, W, 4), dtype=torch.float32, device="cuda:0")
        imgs = [vr.Screenshotter(r, W, H) for _ in cams]
        for skip in (1, 0):
            r.set_option(vr.OPT_SKIP_EMPTY, skip)
            for k in (4, 3):
                r.set_option(vr.OPT_KERNEL, k)
                res = {}
                for rays in (1, 2):
                    r.set_option(vr.OPT_RAYS_PER_LANE, rays)
                    r.set_option(vr.OPT_DIAGNOSTICS, 0)
                    part.fill_(-1.0)
                    r.draw_partial(cams, W, H, part.data_ptr())
                    r.synchronize()
                    r.draw_batch(cams, imgs)
                    px = np.stack([im.read() for im in imgs])
                    r.set_option(vr.OPT_DIAGNOSTICS, 1)
                    r.draw_batch(cams, imgs)
                    dg = [im.diagnostics() for im in imgs]
                    acc, it = np.stack([d[0] for d in dg]), np.stack([d[1] for d in dg])
                    ne = np.stack([im.executed_samples() for im in imgs])
                    geo = np.stack([im.ray_diagnostics() for im in imgs])
                    res[rays] = (part.cpu().numpy().copy(), px, it, acc, ne, geo)
                # (the executed-sample count is a statistic of the chunking, not a result: compared only with skipping off)
                same = all(np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))
                           for i, (a, b) in enumerate(zip(res[1], res[2])) if i != 4 or skip == 0)
                print(f"{name:9s} dims {dims} {W}x{H} skip {skip} kernel {k}: {'bit-identical' if same else 'DIFFERENT'}", flush=True)
                if not same:
                    for i, (a, b) in enumerate(zip(res[1], res[2])):
                        d = np.argwhere(np.ascontiguousarray(a) != np.ascontiguousarray(b))
                        if len(d):
                            print("   output", i, "differs at", len(d), "places, first", d[0], a[tuple(d[0])], b[tuple(d[0])])
                bad += 0 if same else 1
        for im in imgs:
            im.close()
        r.close()
    print("rays_check:", "ok" if bad == 0 else f"{bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
