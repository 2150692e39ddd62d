"""Camera uniforms no host would send on purpose -- NaN, infinities, zeros, huge values -- through every march variant: each draw must
come back (the reference marches whatever rays such a uniform gives; a kernel must at least end).  Run under a short timeout:
    timeout 120 python tools/garbage_cameras.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import oracle as orc  # noqa: E402
import volume_renderer_amd as vr  # noqa: E402
from volume_renderer_amd import synth  # noqa: E402

W, H = 40, 24
good = orc.camera_uniform(orc.sphere_position(2, 3, 8, 8, 2.5), target=(0.0, 0.0, 0.0), aspect=W / H)
cams = {"nan": np.full(36, np.nan, np.float32), "inf": np.full(36, np.inf, np.float32), "zeros": np.zeros(36, np.float32),
        "huge": (good * np.float32(1e30)).astype(np.float32), "position nan": good.copy(), "position 1e30": good.copy(),
        "matrix zero, position good": good.copy()}
# (the layout of the 36 floats: test_abi_and_host.py; whichever they are, poke the first and the last four)
cams["position nan"][-4:] = np.nan
cams["position 1e30"][-4:] = 1e30
cams["matrix zero, position good"][:32] = 0.0
for scalar in (False, True):
    rm = vr.RayMarcher(0, W, H)
    g = rm.voxel_grid
    if scalar:
        g.upload_scalar(synth.blobs_scalar_u8((40, 33, 21)).reshape(-1), (40, 33, 21), 1)
    else:
        g.upload_rgba8(synth.random_rgba8((40, 33, 21), 3))
    g.attenuation = 20.0
    g.update_voxel_grid_buffer()
    img = vr.Screenshotter(rm, W, H)
    for name, c in cams.items():
        for k in (1, 3, 4, 0):
            if scalar and k == 0:
                continue
            for skip in (1, 0):
                for diag in (0, 1):
                    rm.set_option(vr.OPT_KERNEL, k)
                    rm.set_option(vr.OPT_SKIP_EMPTY, skip)
                    rm.set_option(vr.OPT_DIAGNOSTICS, diag)
                    print(f"scalar {scalar} camera '{name}' kernel {k} skip {skip} diagnostics {diag}:", end=" ", flush=True)
                    rm.draw(vr.uniform_from_array(c), img)
                    px = img.read()
                    print("back, non-zero pixels", int((px != 0).any(axis=-1).sum()), flush=True)
    img.close()
    rm.close()
print("garbage_cameras: every draw came back")
