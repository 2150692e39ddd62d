#!/usr/bin/env python3
"""The WHOLE C5 volume (2048 x 2048 x 4096 scalar, 16 GiB of u8 -- 64 GiB as the f32 BASELINE.json names, beyond the reference's
4096-per-axis 3-D texture limit, state.rs:59) on ONE MI355X, unpartitioned: since round 3 a scalar volume is held compact
(packed alpha bricks + occupancy pyramid, about 1.1 B/voxel), so the 17.2 G voxels need 19 GB of the 288 GB once ingested
(the ingest builds that form straight from the raw samples: raw staging + 1 B/voxel, 35 GB; it passed through the 9 B/voxel
RGBA form, 155 GB, before the direct ingest).  Renders a few 2048^2 views, checks skipping on/off
bit-identical and a crop of each against the oracle (orc_render_slab_stored over all slices, from the u8 volume on the host).
    python tools/c5_whole.py [--dims X Y Z] [--res R] [--views N] [--json out.json]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs=3, default=[2048, 2048, 4096])
    ap.add_argument("--res", type=int, default=2048)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--crop", type=int, default=128)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    import oracle
    import volume_renderer_amd as vr
    from volume_renderer_amd import synth

    dims, R = tuple(a.dims), a.res
    t0 = time.time()
    vol = synth.blobs_scalar_u8_slices(dims, 0, dims[2])
    gen_s = time.time() - t0
    print(f"{dims}: {vol.nbytes / 1e9:.2f} GB of u8 generated in {gen_s:.1f} s, nonzero {np.count_nonzero(vol[::16]) / vol[::16].size:.3f}", flush=True)
    r = vr.RayMarcher(0, R, R)
    t0 = time.time()
    r.voxel_grid.new(dims)
    peak = r.voxel_grid.resident_bytes() + vol.nbytes   # the grid as the upload finds it + the raw samples staged on the device
    r.voxel_grid.upload_scalar(vol.reshape(-1), dims, 1)
    r.synchronize()
    ingest_s = time.time() - t0
    resident = r.voxel_grid.resident_bytes()
    print(f"ingested in {ingest_s:.1f} s; grid + staged raw samples were {peak / 1e9:.1f} GB during the ingest, the grid holds {resident / 1e9:.2f} GB now "
          f"({resident / vol.nbytes:.2f} B/voxel)", flush=True)
    views = oracle.sphere_views(8, 5)
    picks = [views[i] for i in np.linspace(0, len(views) - 1, a.views).astype(int)]
    cams36 = [oracle.camera_uniform(oracle.sphere_position(ix, iy, 8, 5, 2.5)) for ix, iy in picks]
    cams = [vr.uniform_from_array(c) for c in cams36]
    imgs = [vr.Screenshotter(r, R, R) for _ in cams]
    out = {"dims": dims, "res": R, "views": picks, "input_bytes": int(vol.nbytes), "gen_s": gen_s, "ingest_s": ingest_s,
           "grid_bytes_during_ingest": int(peak), "grid_bytes_resident": int(resident)}
    px = {}
    for skip in (1, 0):
        r.set_option(vr.OPT_SKIP_EMPTY, skip)
        r.draw_batch(cams, imgs)
        r.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            r.draw_batch(cams, imgs)
        r.synchronize()
        ms = (time.perf_counter() - t0) / 2 / len(cams) * 1e3
        px[skip] = [im.read().copy() for im in imgs]
        out[f"ms_per_view_skip{skip}"] = ms
        print(f"skip={skip}: {ms:.2f} ms per {R}^2 view", flush=True)
    same = all(np.array_equal(x, y) for x, y in zip(px[1], px[0]))
    print("skip on/off bit-identical:", same, flush=True)
    out["skip_on_off_identical"] = bool(same)
    r.set_option(vr.OPT_SKIP_EMPTY, 1)
    n_ref = r.count_reference_samples(cams, R, R)
    out["reference_samples_per_view"] = n_ref / len(cams)
    print(f"{n_ref / len(cams) / 1e9:.2f} G reference samples per view => {n_ref / len(cams) / (out['ms_per_view_skip1'] * 1e-3) / 1e9:.0f} Gsamples/s", flush=True)
    # a crop of every view against the oracle (f32 accumulators of the diagnostics build)
    r.set_option(vr.OPT_DIAGNOSTICS, 1)
    img = vr.Screenshotter(r, R, R)
    C = a.crop
    worst = 0.0
    for i, c36 in enumerate(cams36):
        r.draw(cams[i], img)
        f32, iters = img.diagnostics()
        assert np.array_equal(img.read(), px[1][i])
        ys, xs = np.nonzero(px[1][i][..., 3] > 5)
        if len(ys) == 0:
            continue
        cy, cx = int(np.median(ys)), int(np.median(xs))
        x0, y0 = min(max(cx - C // 2, 0), R - C), min(max(cy - C // 2, 0), R - C)
        t0 = time.time()
        ref, steps = oracle.render_slab_stored(vol, dims, 0, c36, R, R, 0, dims[2], crop=(x0, y0, C, C))
        e = float(np.abs(f32[y0:y0 + C, x0:x0 + C] - ref).max())
        worst = max(worst, e)
        ok_steps = bool(np.array_equal(steps, iters[y0:y0 + C, x0:x0 + C]))
        print(f"view {picks[i]} crop ({x0},{y0}) {C}^2 vs oracle: max |d| = {e:.3e}, iteration counts equal: {ok_steps} (oracle {time.time() - t0:.1f} s, "
              f"alpha max {ref[..., 3].max():.3f})", flush=True)
        assert ok_steps and e < 1e-4
    out["oracle_max_diff_f32"] = worst
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1, default=lambda o: list(o) if isinstance(o, tuple) else float(o))
    for im in imgs + [img]:
        im.close()
    r.close()


if __name__ == "__main__":
    main()
