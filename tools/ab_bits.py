"""Bit-identity of two library builds: renders the same RGBA and scalar volumes with the library named by VRM_LIB and prints one
digest per case over the f32 accumulators (vrm_draw_partial), the 8-bit images and the diagnostics build's iteration counts.
Run once per build and compare the lines:    VRM_LIB=volume_renderer_amd/variants/libvrm_X.so python tools/ab_bits.py"""
import hashlib
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

    cases = [("rgba blobs", (96, 80, 72), 1.0, 300, 200, False, False), ("rgba blobs att 40", (64, 64, 64), 40.0, 256, 256, False, False),
             ("rgba random", (33, 47, 29), 7.0, 130, 90, False, False), ("rgba tf", (64, 64, 64), 5.0, 200, 200, False, True),
             ("scalar", (103, 96, 91), 5.0, 256, 192, True, False), ("rgba big", (256, 256, 256), 1.0, 512, 512, False, False)]
    for name, dims, att, W, H, scalar, tf in cases:
        r = vr.RayMarcher(0, W, H)
        g = r.voxel_grid
        if scalar:
            g.upload_scalar(synth.blobs_scalar_u8(dims).reshape(-1), dims, 1)
        elif "random" in name:
            g.upload_rgba8(synth.random_rgba8(dims, 5))
        else:
            g.upload_rgba8(synth.quantize_rgba8(synth.blobs_rgba_f32(dims)))
        g.attenuation = att
        g.update_voxel_grid_buffer()
        g.set_transfer_function_active(tf)
        g.update_transfer_function_buffer()
        cams = [vr.uniform_from_array(oracle.camera_uniform(oracle.sphere_position(ix, iy, 8, 8, rad), aspect=W / H))
                for ix, iy, rad in [(1, 3, 2.5), (4, 6, 2.5), (7, 1, 1.2), (2, 2, 0.3), (6, 6, 2.5), (0, 4, 3.0)]]
        part = torch.empty((len(cams), H, W, 4), dtype=torch.float32, device="cuda:0")
        imgs = [vr.Screenshotter(r, W, H) for _ in cams]
        h = hashlib.sha256()
        for skip in (1, 0):
            r.set_option(vr.OPT_SKIP_EMPTY, skip)
            for packed in (1, 0):
                r.set_option(vr.OPT_PACKED_ALPHA, packed)
                r.set_option(vr.OPT_DIAGNOSTICS, 0)
                part.fill_(-1.0)
                r.draw_partial(cams, W, H, part.data_ptr())
                r.synchronize()
                h.update(part.cpu().numpy().tobytes())
                r.draw_batch(cams, imgs)
                for im in imgs:
                    h.update(np.ascontiguousarray(im.read()).tobytes())
                r.set_option(vr.OPT_DIAGNOSTICS, 1)
                r.draw_batch(cams, imgs)
                for im in imgs:
                    h.update(np.ascontiguousarray(im.diagnostics()[1]).tobytes())
        print(f"{name:18s} {h.hexdigest()[:24]}", flush=True)
        for im in imgs:
            im.close()
        r.close()


if __name__ == "__main__":
    main()
