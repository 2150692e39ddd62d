"""Fixed cost of one march launch on C3: kernel time (hipEvents around the launch) of launches of 8 .. 128 views, a linear fit,
and the diagnostics build's view of the same.    python tools/launch_overhead.py [volume edge] [image edge]"""
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
    R, D = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), (int(sys.argv[1]) if len(sys.argv) > 1 else 512)
    print(f"volume {D}^3, image {R}^2")
    rm = vr.RayMarcher(0, R, R)
    vol = torch.from_numpy(synth.quantize_rgba8(synth.blobs_rgba_f32((D, D, D)))).cuda()
    rm.voxel_grid.new((D, D, D))
    rm.voxel_grid.upload_rgba8_device(vol.data_ptr(), vol.numel())
    views = oracle.sphere_views(16, 17)
    cams = [vr.uniform_from_array(oracle.camera_uniform(oracle.sphere_position(ix, iy, 16, 17, 2.5))) for ix, iy in views]
    imgs = [vr.Screenshotter(rm, R, R) for _ in range(128)]
    xs, ys = [], []
    for kernel in (1,):
        rm.set_option(vr.OPT_KERNEL, kernel)
        for n in (8, 16, 32, 64, 128):
            tot, cnt = 0.0, 0
            for rep in range(3):
                rm.synchronize()
                rm.set_option(vr.OPT_PROFILE, 1)
                rm.profile_read()
                for c0 in range(0, 256, n):
                    rm.draw_batch(cams[c0:c0 + n], imgs[:n])
                rm.synchronize()
                k, l = rm.profile_read()
                if rep:
                    tot += k
                    cnt += l
            per = tot / cnt
            print(f"kernel option {kernel}: {n:3d} views per launch: {per:7.3f} ms per launch, {per / n:.4f} ms per view", flush=True)
            if kernel == 1:
                xs.append(n)
                ys.append(per)
    c, o = np.polyfit(xs, ys, 1)
    print(f"fit (tile queues): {o:.3f} ms per launch + {c:.4f} ms per view")


if __name__ == "__main__":
    main()
