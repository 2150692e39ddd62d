"""Small fixed workload for rocprofv3 passes: N views of the C3 volume with one kernel variant.

usage: python tools/prof_workload.py [--kernel K] [--views N] [--dim D] [--res R] [--reps M] [--scalar] [--tf]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import volume_renderer_amd as vr  # noqa: E402
from volume_renderer_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", type=int, default=1)
    ap.add_argument("--skip", type=int, default=0)
    ap.add_argument("--views", type=int, default=16)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--scalar", action="store_true")
    ap.add_argument("--tf", action="store_true")
    ap.add_argument("--att", type=float, default=1.0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--packed", type=int, default=1, help="VRM_OPT_PACKED_ALPHA")
    ap.add_argument("--rays", type=int, default=0, help="VRM_OPT_RAYS_PER_LANE (0 = automatic)")
    ap.add_argument("--xdiv", type=int, default=16)
    ap.add_argument("--ydiv", type=int, default=17)
    ap.add_argument("--ix", type=int, default=-1, help="only the views of this azimuth index")
    ap.add_argument("--all", action="store_true", help="every view of the sphere, in launches of --views")
    ap.add_argument("--count-exec", action="store_true", help="also print the executed-sample and chunk counters of the launch "
                    "(one extra pass through the diagnostics build)")
    a = ap.parse_args()
    D, R = a.dim, a.res
    rm = vr.RayMarcher(0, R, R)
    rm.set_option(vr.OPT_KERNEL, a.kernel)
    rm.set_option(vr.OPT_SKIP_EMPTY, a.skip)
    rm.set_option(vr.OPT_CHUNK_STEPS, a.chunk)
    rm.set_option(vr.OPT_PACKED_ALPHA, a.packed)
    rm.set_option(vr.OPT_RAYS_PER_LANE, a.rays)
    if a.scalar:
        s = synth.blobs_scalar_u8((D, D, D))
        rm.voxel_grid.upload_scalar(s.reshape(-1), (D, D, D), 1)
    else:
        rm.voxel_grid.upload_rgba8(synth.quantize_rgba8(synth.blobs_rgba_f32((D, D, D))))
    rm.voxel_grid.attenuation = a.att
    rm.voxel_grid.update_voxel_grid_buffer()
    rm.voxel_grid.set_transfer_function_active(a.tf)
    rm.voxel_grid.update_transfer_function_buffer()
    allv = vr.sphere_views(a.xdiv, a.ydiv)
    if a.ix >= 0:
        views = [v for v in allv if v[0] == a.ix][: a.views]   # one azimuth, every polar step
    else:
        # a spread over polar AND azimuth steps (a plain stride of 8 or 16 would keep hitting the same azimuths,
        # i.e. only axis-aligned views, which are the cheap ones)
        views = [allv[(i * 37) % len(allv)] for i in range(min(a.views, len(allv)))]
    cam = vr.Camera(1.0)
    csp = vr.CameraSphereController(a.ydiv, a.xdiv, (0, 0, 0), 2.5)
    cams = []
    for ix, iy in views:
        csp.current_index_x, csp.current_index_y = ix, iy
        csp.update_position(cam)
        cams.append(cam.uniform())
    if a.all:
        # the whole export in launches of --views views, per-launch times (view direction changes the cost)
        cams = []
        for ix, iy in allv:
            csp.current_index_x, csp.current_index_y = ix, iy
            csp.update_position(cam)
            cams.append(cam.uniform())
        imgs = [vr.Screenshotter(rm, R, R) for _ in range(a.views)]
        n_ref = rm.count_reference_samples(cams, R, R)
        rm.set_option(vr.OPT_PROFILE, 1)
        for rep in range(a.reps):
            per = []
            for c0 in range(0, len(cams), a.views):
                part = cams[c0:c0 + a.views]
                rm.draw_batch(part, imgs[:len(part)])
                rm.synchronize()
                ms, n = rm.profile_read()
                per.append(ms / len(part))
            tot = sum(p * min(a.views, len(cams) - i * a.views) for i, p in enumerate(per))
            print(f"kernel {a.kernel} skip {a.skip} chunk {a.chunk} packed {a.packed}: all {len(cams)} views {R}^2 of {D}^3: {tot:.1f} ms => "
                  f"{tot / len(cams):.3f} ms/view, {n_ref / tot / 1e6:.1f} Gsamples/s; ms/view per launch: "
                  + " ".join(f"{p:.2f}" for p in per), flush=True)
        return
    imgs = [vr.Screenshotter(rm, R, R) for _ in cams]
    n_ref = rm.count_reference_samples(cams, R, R)
    rm.set_option(vr.OPT_PROFILE, 1)
    for rep in range(a.reps):
        rm.draw_batch(cams, imgs)
        rm.synchronize()
        ms, n = rm.profile_read()
        print(f"kernel {a.kernel} skip {a.skip}: {len(cams)} views {R}^2 of {D}^3: {ms:.2f} ms => {ms / len(cams):.3f} ms/view, "
              f"{n_ref / ms / 1e6:.1f} Gsamples/s (ref-equivalent)", flush=True)
    if a.count_exec:
        rm.set_option(vr.OPT_PROFILE, 0)
        rm.set_option(vr.OPT_DIAGNOSTICS, 1)
        rm.diag_counters()
        for c in cams:
            rm.draw(c, imgs[0])
        dc = rm.diag_counters()
        print(f"COUNTERS views={len(cams)} n_ref={n_ref} " + " ".join(f"{k}={v}" for k, v in dc.items()), flush=True)


if __name__ == "__main__":
    main()
