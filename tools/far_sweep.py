"""f32 deviation of the march kernels from the oracle against camera distance, on a noise volume at attenuation 100 (the case family
of fuzz seed 9102 case 2174): kernel 0 (plain global gathers, the oracle's coordinate sequence) and kernel 1 (LDS march; FAR
instance beyond 4096 texels).    python tools/far_sweep.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch  # noqa: F401
    import oracle as orc
    import volume_renderer_amd as vr
    from test_gpu_parity import configure, draw
    from volume_renderer_amd import synth

    dims = (64, 15, 64)
    vox = synth.random_rgba8(dims, 12345)
    W, H, att = 31, 83, 100.0
    rm = vr.RayMarcher(0, 64, 64)
    rm.set_option(vr.OPT_SKIP_EMPTY, 0)
    configure(rm, vox, att, False)
    for radius in (3.0, 10.0, 30.0, 60.0, 70.0, 100.0, 200.0, 400.0, 1000.0):
        pos = orc.sphere_position(3, 5, 8, 8, radius)
        cam = orc.camera_uniform(pos, target=(0.0, 0.0, 0.0), aspect=W / H)
        ref = orc.render(vox, cam, W, H, attenuation=att, tf=orc.tf_words(False), want_steps=True)
        out = []
        for k in (0, 1):
            for chunk in (0, 1):
                rm.set_option(vr.OPT_KERNEL, k)
                rm.set_option(vr.OPT_CHUNK_STEPS, chunk)
                g = draw(rm, vr, cam, W, H)
                d = np.abs(g["f32"] - ref["f32"])
                out.append(f"kernel {k} chunk {chunk}: {d.max():.2e} (channels {' '.join('%.1e' % d[..., c].max() for c in range(4))}) iters equal {np.array_equal(g['iters'], ref['steps'])}")
        print(f"radius {radius:6.0f} ({max(abs(p) for p in pos) * 64:7.0f} texels): " + "; ".join(out), flush=True)


if __name__ == "__main__":
    main()
