"""One rank's share of C5 (2048 x 2048 x 4096 scalar volume in G z slabs) on one GPU: ingest of the stored slices, march
of the slab's f32 partial at RxR (skipping on / off / diagnostics build), a crop against the oracle, ms per view.

    python tools/c5_slab_rank.py [--slabs 8 --slab 3 --res 2048 --dims 2048 2048 4096 --views 4 --json out.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs=3, default=[2048, 2048, 4096])
    ap.add_argument("--slabs", type=int, default=8)
    ap.add_argument("--slab", type=int, default=3)
    ap.add_argument("--res", type=int, default=2048)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--crop", type=int, default=256)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    import torch
    import oracle
    import volume_renderer_amd as vr
    from volume_renderer_amd import sortlast, synth

    dims, R = tuple(a.dims), a.res
    z0, z1 = sortlast.slab_ranges(dims[2], a.slabs)[a.slab]
    torch.cuda.init()
    free0 = torch.cuda.mem_get_info(0)[0]
    r = vr.RayMarcher(0, R, R)
    s0, s1 = r.voxel_grid.new_slab(dims, z0, z1)
    t0 = time.time()
    stored = synth.blobs_scalar_u8_slices(dims, s0, s1)
    gen_s = time.time() - t0
    t0 = time.time()
    r.voxel_grid.upload_scalar(stored.reshape(-1), (dims[0], dims[1], s1 - s0), 1)
    r.synchronize()
    ingest_s = time.time() - t0
    resident = free0 - torch.cuda.mem_get_info(0)[0] # device memory the process took since before the context (the allocator may keep freed blocks)
    grid_resident = r.voxel_grid.resident_bytes()     # what the grid holds: packed alpha bricks + pyramid for a scalar volume (compact form)
    print(f"slab {a.slab}/{a.slabs}: slices [{z0},{z1}) stored [{s0},{s1}), {stored.nbytes / 1e9:.2f} GB input, "
          f"generated in {gen_s:.1f} s, ingested in {ingest_s:.2f} s, nonzero {np.count_nonzero(stored) / stored.size:.3f}, "
          f"grid holds {grid_resident / 1e9:.2f} GB ({grid_resident / stored.nbytes:.2f} x input), process took {resident / 1e9:.2f} GB of HBM", flush=True)
    views = oracle.sphere_views(8, 5)
    picks = [views[i] for i in np.linspace(0, len(views) - 1, a.views).astype(int)]
    cams36 = [oracle.camera_uniform(oracle.sphere_position(ix, iy, 8, 5, 2.5)) for ix, iy in picks]
    cams = [vr.uniform_from_array(c) for c in cams36]
    part = torch.empty((len(cams), R, R, 4), dtype=torch.float32, device="cuda:0")
    out = {"dims": dims, "slabs": a.slabs, "slab": a.slab, "res": R, "views": picks, "gen_s": gen_s, "ingest_s": ingest_s,
           "stored_slices": [s0, s1], "input_bytes": int(stored.nbytes), "grid_bytes_resident": int(grid_resident),
           "grid_resident_over_input": round(grid_resident / stored.nbytes, 3), "hbm_bytes_taken_by_process": int(resident)}
    res = {}
    for skip in (1, 0):
        r.set_option(vr.OPT_SKIP_EMPTY, skip)
        r.draw_partial(cams, R, R, part.data_ptr())
        r.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r.draw_partial(cams, R, R, part.data_ptr())
        r.synchronize()
        ms = (time.perf_counter() - t0) / 3 / len(cams) * 1e3
        res[skip] = part.cpu().numpy().copy()
        out[f"ms_per_view_skip{skip}"] = ms
        print(f"skip={skip}: {ms:.3f} ms/view", flush=True)
    r.set_option(vr.OPT_SKIP_EMPTY, 1)
    d = np.abs(res[1] - res[0])
    print(f"skip on vs off: max |d| = {d.max():.3e}, differing pixels {int((d.max(axis=-1) > 0).sum())} of {d.shape[0] * R * R}", flush=True)
    out["skip_on_off_max_diff"] = float(d.max())
    # the global-gather baseline kernel (no LDS staging, no skipping) and the oracle at the worst pixels
    r.set_option(vr.OPT_KERNEL, 0)
    r.draw_partial(cams, R, R, part.data_ptr())
    r.synchronize()
    res_k0 = part.cpu().numpy().copy()
    r.set_option(vr.OPT_KERNEL, 1)
    for skip in (1, 0):
        e = np.abs(res[skip] - res_k0)
        print(f"skip={skip} vs global-gather kernel: max |d| = {e.max():.3e}, pixels > 1e-5: {int((e.max(axis=-1) > 1e-5).sum())}", flush=True)
    if d.max() > 0 and not a.no_oracle:
        flat = d.max(axis=-1)
        for _ in range(3):
            vi, py, px = np.unravel_index(int(np.argmax(flat)), flat.shape)
            x0, y0 = min(max(px - 8, 0), R - 16), min(max(py - 8, 0), R - 16)
            ref, _ = oracle.render_slab_stored(stored, dims, s0, cams36[vi], R, R, z0, z1, crop=(x0, y0, 16, 16))
            o = ref[py - y0, px - x0]
            print(f"worst pixel view {vi} ({px},{py}): skip1 {res[1][vi, py, px]} skip0 {res[0][vi, py, px]} k0 {res_k0[vi, py, px]} oracle {o}", flush=True)
            print(f"   tile-local lane ({px % 8},{py % 8}); |skip1-oracle| {np.abs(res[1][vi, py, px] - o).max():.3e} |skip0-oracle| {np.abs(res[0][vi, py, px] - o).max():.3e}", flush=True)
            flat[vi, max(py - 8, 0):py + 8, max(px - 8, 0):px + 8] = 0
    # diagnostics build: every step walked
    r.set_option(vr.OPT_DIAGNOSTICS, 1)
    img = vr.Screenshotter(r, R, R)
    diag, iters_all = [], []
    for c in cams:
        r.draw(c, img)
        f, it = img.diagnostics()
        diag.append(f)
        iters_all.append(it)
    r.set_option(vr.OPT_DIAGNOSTICS, 0)
    diag = np.stack(diag)
    for skip in (1, 0):
        dd = np.abs(res[skip] - diag)
        print(f"skip={skip} vs diagnostics build: max |d| = {dd.max():.3e}, pixels > 1e-5: {int((dd.max(axis=-1) > 1e-5).sum())}", flush=True)
        out[f"diag_max_diff_skip{skip}"] = float(dd.max())
    n_ref = r.count_reference_samples(cams, R, R)
    assert n_ref == int(np.stack(iters_all).astype(np.int64).sum())
    out["reference_samples_per_view"] = n_ref / len(cams)
    if not a.no_oracle:
        C = a.crop
        for i, c36 in enumerate(cams36):
            ys, xs = np.nonzero(res[1][i][..., 3] > 0.02)
            if len(ys) == 0:
                print(f"view {picks[i]}: slab not visible")
                continue
            cy, cx = int(np.median(ys)), int(np.median(xs))
            x0, y0 = min(max(cx - C // 2, 0), R - C), min(max(cy - C // 2, 0), R - C)
            t0 = time.time()
            ref, steps = oracle.render_slab_stored(stored, dims, s0, c36, R, R, z0, z1, crop=(x0, y0, C, C))
            dt = time.time() - t0
            for name, arr in (("skip1", res[1][i]), ("skip0", res[0][i]), ("diag", diag[i])):
                e = np.abs(arr[y0:y0 + C, x0:x0 + C] - ref)
                print(f"view {picks[i]} crop ({x0},{y0}) {name} vs oracle: max |d| = {e.max():.3e}, > 1e-4: {int((e.max(axis=-1) > 1e-4).sum())}"
                      f" (oracle {dt:.1f} s, alpha max {ref[..., 3].max():.3f})", flush=True)
                out[f"oracle_max_diff_{name}_view{i}"] = float(e.max())
            assert np.array_equal(steps, iters_all[i][y0:y0 + C, x0:x0 + C])
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1, default=lambda o: list(o) if isinstance(o, tuple) else float(o))
    img.close()
    r.close()


if __name__ == "__main__":
    main()
