"""The workgroup-shared march (kernel variants 3-5) against the per-wave march (variant 1) on the production path: f32
accumulators (vrm_draw_partial) and 8-bit images must be bit-identical -- the sample arithmetic does not depend on the chunking.

    python tools/wg_check.py [--dim 96 --res 200]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=96)
    ap.add_argument("--res", type=int, default=200)
    a = ap.parse_args()
    import torch
    import oracle
    import volume_renderer_amd as vr
    from volume_renderer_amd import synth

    R = a.res
    bad = 0
    for name, dims, scalar, tf, att in [("rgba", (a.dim, a.dim - 16, a.dim + 16), False, False, 1.0), ("rgba+tf", (a.dim, a.dim, a.dim), False, True, 30.0),
                                        ("scalar", (a.dim + 7, a.dim, a.dim - 5), True, False, 5.0), ("tiny", (5, 3, 9), False, False, 50.0),
                                        ("rgba ert", (40, 40, 40), False, False, 900.0)]:
        r = vr.RayMarcher(0, R, R - 24)
        g = r.voxel_grid
        if scalar:
            g.upload_scalar(synth.blobs_scalar_u8(dims).reshape(-1), dims, 1)
        else:
            g.upload_rgba8(synth.quantize_rgba8(synth.blobs_rgba_f32(dims)))
        g.attenuation = att
        g.update_voxel_grid_buffer()
        g.set_transfer_function_active(tf)
        g.update_transfer_function_buffer()
        cams36 = [oracle.camera_uniform(oracle.sphere_position(ix, iy, 8, 8, rad)) for ix, iy, rad in
                  [(1, 3, 2.5), (4, 6, 2.5), (7, 1, 1.2), (2, 2, 0.3), (5, 4, 40.0)]]
        cams = [vr.uniform_from_array(c) for c in cams36]
        W, H = R, R - 24
        part = torch.empty((len(cams), H, W, 4), dtype=torch.float32, device="cuda:0")
        ref = {}
        for skip in (1, 0):
            r.set_option(vr.OPT_SKIP_EMPTY, skip)
            for k in (1, 3, 4, 5):
                r.set_option(vr.OPT_KERNEL, k)
                part.zero_()
                r.draw_partial(cams, W, H, part.data_ptr())
                r.synchronize()
                out = part.cpu().numpy().copy()
                if k == 1:
                    ref[skip] = out
                    continue
                same = np.array_equal(out.view(np.uint32), ref[skip].view(np.uint32))
                d = np.abs(out - ref[skip]).max()
                print(f"{name:9s} dims {dims} skip {skip} kernel {k}: {'bit-identical' if same else 'DIFFERENT'} (max |d| {d:.3e})", flush=True)
                bad += 0 if same else 1
        r.close()
    print("wg_check:", "ok" if bad == 0 else f"{bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
