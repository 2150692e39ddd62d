"""Quick GPU-vs-oracle comparison + timing, for use under gpurun while developing kernels.

usage: python tools/gpu_check.py [--kernel K] [--skip S] [--big]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402  (checker only)
import volume_renderer_amd as vr  # noqa: E402
from volume_renderer_amd import synth, uniform_from_array  # noqa: E402


def compare(rm, vox, cams, W, H, att, use_tf, label):
    g = rm.voxel_grid
    g.upload_rgba8(vox)
    g.attenuation = att
    g.update_voxel_grid_buffer()
    g.set_transfer_function_active(use_tf)
    g.update_transfer_function_buffer()
    rm.set_option(vr.OPT_DIAGNOSTICS, 1)
    worst = 0
    for cam in cams:
        img = vr.Screenshotter(rm, W, H)
        rm.draw(uniform_from_array(cam), img)
        got = img.read()
        f32, iters = img.diagnostics()
        ref = oracle.render(vox, cam, W, H, attenuation=att, tf=oracle.tf_words(use_tf), want_steps=True)
        d8 = np.abs(got.astype(np.int32) - ref["rgba8"].astype(np.int32))
        df = np.abs(f32 - ref["f32"]).max()
        rmse = np.sqrt(np.mean((got.astype(np.float64) - ref["rgba8"].astype(np.float64)) ** 2))
        it_bad = int((iters != ref["steps"]).sum())
        print(f"[{label}] max|d8|={d8.max()} n(d8>0)={int((d8 > 0).sum())} rmse_lsb={rmse:.5f} rmse_norm={rmse / 255:.2e} "
              f"max|df32|={df:.3e} iter_mismatch={it_bad} samples={ref['samples']}", flush=True)
        worst = max(worst, int(d8.max()))
        geo = img.ray_diagnostics()
        Z, Y, X, _ = vox.shape
        rg = oracle.geometry(cam, (X, Y, Z), W, H)
        names = ["rdx", "rdy", "rdz", "tmin", "tmax", "n_in", "hit", "iters"]
        bad = {names[k]: int((geo[..., k].view(np.uint32) != rg[..., k].view(np.uint32)).sum()) for k in range(8)}
        print("    geometry bit mismatches:", bad, flush=True)
        if df > 1e-4:
            adf = np.abs(f32 - ref["f32"])
            for _ in range(3):
                y, x, c = np.unravel_index(np.argmax(adf), adf.shape)
                print(f"    worst px ({x},{y}) ch{c}: gpu={f32[y, x]} ref={ref['f32'][y, x]} iters gpu={iters[y, x]} ref={ref['steps'][y, x]}")
                adf[y, x, :] = 0
        img.close()
    rm.set_option(vr.OPT_DIAGNOSTICS, 0)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--skip", type=int, default=0)
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--views", type=int, default=32)
    a = ap.parse_args()

    rm = vr.RayMarcher(0, 256, 256)
    rm.set_option(vr.OPT_KERNEL, a.kernel)
    rm.set_option(vr.OPT_SKIP_EMPTY, a.skip)
    cams = [oracle.camera_uniform(oracle.sphere_position(ix, iy, 8, 8, 2.5)) for ix, iy in [(1, 3), (0, 4), (5, 1), (2, 6)]]
    worst = 0
    v1 = synth.quantize_rgba8(synth.blobs_rgba_f32((48, 40, 32)))
    worst = max(worst, compare(rm, v1, cams[:2], 96, 80, 1.0, False, "rgba 48x40x32"))
    worst = max(worst, compare(rm, v1, cams[2:3], 96, 80, 30.0, True, "rgba tf att30"))
    s = synth.blobs_scalar_u8((64, 64, 64))
    v2 = np.repeat(s[..., None], 4, axis=3)
    worst = max(worst, compare(rm, v2, cams[1:3], 128, 128, 1.0, False, "scalar 64^3"))
    v3 = synth.random_rgba8((17, 9, 33), 7)
    worst = max(worst, compare(rm, v3, cams[3:4], 64, 64, 100.0, False, "random att100"))
    inside = oracle.camera_uniform(np.array([0.1, 0.2, 0.3], np.float32), target=(1.0, 0.5, 0.0))
    worst = max(worst, compare(rm, v1, [inside], 64, 64, 5.0, False, "eye inside"))
    print("WORST", worst, flush=True)

    if a.big:
        D, R = a.dim, a.res
        t = time.time()
        vol = synth.quantize_rgba8(synth.blobs_rgba_f32((D, D, D)))
        print(f"generated {D}^3 in {time.time() - t:.1f}s", flush=True)
        rm.voxel_grid.upload_rgba8(vol)
        rm.voxel_grid.attenuation = 1.0
        rm.voxel_grid.update_voxel_grid_buffer()
        rm.voxel_grid.set_transfer_function_active(False)
        rm.voxel_grid.update_transfer_function_buffer()
        views = vr.sphere_views(16, 17)[:: max(1, 256 // a.views)][: a.views]
        cams = []
        cam = vr.Camera(1.0)
        csp = vr.CameraSphereController(17, 16, (0, 0, 0), 2.5)
        for ix, iy in views:
            csp.current_index_x, csp.current_index_y = ix, iy
            csp.update_position(cam)
            cams.append(cam.uniform())
        imgs = [vr.Screenshotter(rm, R, R) for _ in cams]
        rm.set_option(vr.OPT_PROFILE, 1)
        for rep in range(3):
            rm.synchronize()
            t = time.time()
            rm.draw_batch(cams, imgs)
            rm.synchronize()
            dt = time.time() - t
            ms, n = rm.profile_read()
            print(f"rep{rep}: {len(cams)} views {R}^2 of {D}^3: wall {dt * 1e3:.2f} ms, kernel {ms:.2f} ms/{n} launch "
                  f"=> {ms / len(cams):.3f} ms/view, {len(cams) / dt:.1f} views/s", flush=True)
        # per-view timing
        for i in range(0, len(cams), max(1, len(cams) // 8)):
            rm.draw(cams[i], imgs[i])
            ms, n = rm.profile_read()
            print(f"  view {views[i]}: {ms:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
