"""The C oracle against its independently written numpy twin (oracle/oracle_np.py).  CPU only.

Neither is the reference (which cannot run here); agreement of two separate restatements
of raymarcher.wgsl is the cross-check that a slip in one of them would not survive.
"""
import numpy as np
import pytest

from conftest import sphere_cam
from oracle import oracle_np
from volume_renderer_amd import synth

CASES = [
    # dims, view, image, attenuation, tf
    ((24, 20, 16), (1, 3), (40, 32), 1.0, False),
    ((24, 20, 16), (5, 1), (33, 27), 30.0, True),
    ((9, 17, 5), (2, 6), (32, 32), 100.0, False),
    ((16, 16, 16), (6, 4), (24, 40), 8.0, True),
]


@pytest.mark.parametrize("dims,view,size,att,tf", CASES)
def test_c_oracle_matches_numpy_twin(orc, dims, view, size, att, tf):
    vox = synth.random_rgba8(dims, seed=3) if dims[0] == 9 else synth.quantize_rgba8(synth.blobs_rgba_f32(dims))
    W, H = size
    cam = sphere_cam(orc, *view, aspect=W / H)
    a = oracle_np.render(vox, cam, W, H, att, tf)
    b = orc.render(vox, cam, W, H, attenuation=att, tf=orc.tf_words(tf), want_steps=True)
    assert np.array_equal(a["steps"], b["steps"])                      # same lattice, same loop exits
    assert np.abs(a["f32"] - b["f32"]).max() < 5e-5                    # exp() implementations differ in the last ulp
    assert np.abs(a["rgba8"].astype(int) - b["rgba8"].astype(int)).max() <= 1


def test_twin_geometry_is_bitwise(orc):
    cam = sphere_cam(orc, 3, 5)
    dims = (12, 7, 10)
    ro, rd = oracle_np.pixel_rays(cam, 37, 29)
    bmin, bsize = oracle_np.grid_box(dims)
    hit, t0, t1 = oracle_np.slab_test(ro, rd, bmin, bsize)
    geo = orc.geometry(cam, dims, 37, 29)
    assert np.array_equal(rd.view(np.uint32), geo[..., :3].view(np.uint32))
    assert np.array_equal(hit, geo[..., 6] > 0)
    assert np.array_equal(t0[hit].view(np.uint32), geo[..., 3][hit].view(np.uint32))
    assert np.array_equal(t1[hit].view(np.uint32), geo[..., 4][hit].view(np.uint32))


def test_twin_camera_inside_and_axis_aligned_rays(orc):
    # rd components of exactly 0 give +-inf / NaN slabs (wgsl:252-262): both restatements must agree
    vox = synth.random_rgba8((6, 6, 6), seed=5)
    cam = orc.camera_uniform(np.array([0.0, 0.0, 0.2], np.float32), target=(0.0, 0.0, -1.0))
    a = oracle_np.render(vox, cam, 15, 15, 20.0, False)
    b = orc.render(vox, cam, 15, 15, attenuation=20.0, want_steps=True)
    assert np.array_equal(a["steps"], b["steps"])
    assert np.abs(a["f32"] - b["f32"]).max() < 5e-5
