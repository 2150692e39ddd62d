"""Balance of the view partition of the C3 export (256 sphere views, x inner) over N = 2, 4, 8 ranks, on ONE GPU: the plain
interleave i % N against the interleave rotated by one per row, (i + i // x_div) % N (volume_renderer_amd/distributed.py).
First from per-view kernel times (one view per launch), then as the ranks would run it: each rank's share of the export in
launches of up to 64 views, kernel time per rank.    python tools/partition_balance.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import oracle
    import volume_renderer_amd as vr
    from volume_renderer_amd import synth
    R, D = 1024, 512
    rm = vr.RayMarcher(0, R, R)
    vol = torch.from_numpy(synth.quantize_rgba8(synth.blobs_rgba_f32((D, D, D)))).cuda()
    rm.voxel_grid.new((D, D, D))
    rm.voxel_grid.upload_rgba8_device(vol.data_ptr(), vol.numel())
    views = oracle.sphere_views(16, 17)
    cams = [vr.uniform_from_array(oracle.camera_uniform(oracle.sphere_position(ix, iy, 16, 17, 2.5))) for ix, iy in views]
    imgs = [vr.Screenshotter(rm, R, R) for _ in range(4)]
    cost = np.zeros(len(cams))
    for rep in range(3):
        for i in range(len(cams)):                           # one view per launch (its kernel time by hipEvents)
            rm.synchronize()
            rm.set_option(vr.OPT_PROFILE, 1)
            rm.profile_read()
            rm.draw_batch(cams[i:i + 1], imgs[:1])
            rm.synchronize()
            k, _ = rm.profile_read()
            if rep:
                cost[i] += k / 2
    print("per-view kernel ms (one view per launch): min %.3f mean %.3f max %.3f" % (cost.min(), cost.mean(), cost.max()))
    for n in (2, 4, 8):
        for name, rule in (("i % N", lambda i, ix, iy: i % n), ("(ix + iy) % N", lambda i, ix, iy: (ix + iy) % n)):
            per = np.zeros(n)
            for i, (ix, iy) in enumerate(views):
                per[rule(i, ix, iy)] += cost[i]
            print(f"N={n} {name:14s}: per-rank ms {np.round(per, 2).tolist()}  max/mean {per.max() / per.mean():.3f}")
    # as the ranks run it: a rank's views in launches of up to 64
    from volume_renderer_amd import distributed as vd
    bank = [vr.Screenshotter(rm, R, R) for _ in range(64)]
    for n in (4, 8):
        for name, xd in (("i % N", None), ("(i + i // 16) % N", 16)):
            per = []
            for r in range(n):
                mine = [cams[i] for i in vd.partition_views(len(cams), r, n, xd)]
                t = 0.0
                for rep in range(3):
                    rm.synchronize()
                    rm.set_option(vr.OPT_PROFILE, 1)
                    rm.profile_read()
                    for c0 in range(0, len(mine), 64):
                        rm.draw_batch(mine[c0:c0 + 64], bank[:len(mine[c0:c0 + 64])])
                    rm.synchronize()
                    k, _ = rm.profile_read()
                    if rep:
                        t += k / 2
                per.append(t)
            per = np.asarray(per)
            print(f"N={n} {name:18s}: kernel ms per rank and export {np.round(per, 2).tolist()}  max/mean {per.max() / per.mean():.3f}")


if __name__ == "__main__":
    main()
