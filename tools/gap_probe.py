import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import oracle, volume_renderer_amd as vr
from volume_renderer_amd import synth
R, D, B = 1024, 512, 64
rm = vr.RayMarcher(0, R, R)
vol = torch.from_numpy(synth.quantize_rgba8(synth.blobs_rgba_f32((D, D, D)))).cuda()
rm.voxel_grid.new((D, D, D)); rm.voxel_grid.upload_rgba8_device(vol.data_ptr(), vol.numel())
cams = [vr.uniform_from_array(oracle.camera_uniform(oracle.sphere_position(ix, iy, 16, 17, 2.5))) for ix, iy in oracle.sphere_views(16, 17)]
banks = [[vr.Screenshotter(rm, R, R) for _ in range(B)] for _ in range(2)]
def step(readback):
    w = 0
    for c0 in range(0, len(cams), B):
        imgs = banks[w]
        if readback:
            for im in imgs: im.wait()
        rm.draw_batch(cams[c0:c0 + B], imgs)
        if readback:
            for im in imgs: im.screenshot()
        w ^= 1
for readback in (True, False, True, False):
    for _ in range(3): step(readback)
    rm.synchronize(); rm.set_option(vr.OPT_PROFILE, 1); rm.profile_read()
    t0 = time.perf_counter()
    for _ in range(10): step(readback)
    rm.synchronize()
    dt = (time.perf_counter() - t0) / 10 * 1e3
    k, n = rm.profile_read(); rm.set_option(vr.OPT_PROFILE, 0)
    print(f"readback {readback}: {dt:.2f} ms per step, kernels {k / 10:.2f} ms per step ({n} launches), gap {dt - k / 10:.2f} ms", flush=True)
