"""Sort-last on the GPU through the C ABI: slab partials against the oracle's, ownership of every sample
by exactly one slab (bit-exact counts), the compositing kernel against its numpy restatement, and slab render + composite
against the unpartitioned render.  One GPU plays every rank in turn; the exchange step itself is covered on CPU with gloo
(tests/test_sortlast_cpu.py) and runs over RCCL in bench.py --config C5."""
import numpy as np
import pytest
import torch

from conftest import sphere_cam
from volume_renderer_amd import sortlast, synth

pytestmark = pytest.mark.gpu

CAMS = [(1, 3), (3, 4), (6, 2), (0, 4), (4, 4)]


def _partial(rm, vr, cams36, W, H):
    t = torch.empty((len(cams36), H, W, 4), dtype=torch.float32, device="cuda:0")
    rm.draw_partial([vr.uniform_from_array(c) for c in cams36], W, H, t.data_ptr())
    rm.synchronize()
    return t


@pytest.mark.parametrize("n_slabs", [2, 3])
def test_slab_partials_match_oracle(rm, vr, orc, n_slabs):
    dims = (20, 18, 24)
    vox = synth.random_rgba8(dims, 5)
    W = H = 40
    cams = [sphere_cam(orc, ix, iy) for ix, iy in CAMS[:3]]
    for z0, z1 in sortlast.slab_ranges(dims[2], n_slabs):
        s0, s1 = rm.voxel_grid.new_slab(dims, z0, z1)
        assert (s0, s1) == sortlast.stored_range(dims[2], z0, z1)
        assert list(rm.voxel_grid.uniform().dimensions[:3]) == list(dims)     # the box stays the whole volume's
        rm.voxel_grid.upload_rgba8(vox[s0:s1])
        rm.voxel_grid.attenuation = 30.0
        rm.voxel_grid.update_voxel_grid_buffer()
        got = _partial(rm, vr, cams, W, H).cpu().numpy()
        for i, cam in enumerate(cams):
            want = orc.render_slab(vox, cam, W, H, z0, z1, attenuation=30.0)
            assert np.abs(got[i] - want).max() <= 1e-4, (z0, z1, i)


def test_every_sample_has_exactly_one_owner(rm, vr, orc):
    """Per ray: the in-slab sample counts of a partition add up to the in-box count of the unpartitioned march, bit for bit."""
    dims = (16, 16, 21)
    vox = synth.random_rgba8(dims, 6)
    W = H = 32
    rm.set_option(vr.OPT_DIAGNOSTICS, 1)
    img = vr.Screenshotter(rm, W, H)
    for ix, iy in CAMS:
        cam = vr.uniform_from_array(sphere_cam(orc, ix, iy))
        rm.voxel_grid.new(dims)
        rm.voxel_grid.upload_rgba8(vox)
        rm.draw(cam, img)
        full = np.asarray(img.ray_diagnostics()).reshape(H, W, 8).copy()
        for n_slabs in (2, 4):
            inside = np.zeros((H, W))
            for z0, z1 in sortlast.slab_ranges(dims[2], n_slabs):
                s0, s1 = rm.voxel_grid.new_slab(dims, z0, z1)
                rm.voxel_grid.upload_rgba8(vox[s0:s1])
                rm.draw(cam, img)
                g = np.asarray(img.ray_diagnostics()).reshape(H, W, 8)
                assert np.array_equal(g[..., :5], full[..., :5]) and np.array_equal(g[..., 7], full[..., 7])   # same rays, same lattice
                inside += g[..., 5]
            assert np.array_equal(inside, full[..., 5]), (ix, iy, n_slabs)
    img.close()


def test_composite_kernel_matches_numpy(rm1, vr, orc):
    rng = np.random.default_rng(8)
    W, H, G = 37, 29, 4
    parts = rng.random((G, H, W, 4), dtype=np.float32) * np.float32(0.6)
    for ix, iy in [(1, 3), (0, 4), (5, 5)]:
        cam36 = sphere_cam(orc, ix, iy)
        cam = vr.uniform_from_array(cam36)
        for row0, rows in [(0, H), (7, 11), (H - 1, 1)]:
            band = torch.from_numpy(np.ascontiguousarray(parts[:, row0:row0 + rows])).to("cuda:0")
            out8 = torch.zeros((rows, W, 4), dtype=torch.uint8, device="cuda:0")
            outf = torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda:0")
            torch.cuda.synchronize()
            rm1.composite(cam, band.data_ptr(), G, W, H, row0, rows, out8.data_ptr(), outf.data_ptr())
            rm1.synchronize()
            want = sortlast.composite_numpy(parts[:, row0:row0 + rows], sortlast.ascending_mask(cam36, W, H, row0, rows))
            assert np.allclose(outf.cpu().numpy(), want, rtol=3e-7, atol=1e-9)
            assert np.abs(out8.cpu().numpy().astype(int) - sortlast.unorm8(want).astype(int)).max() <= 1
    with pytest.raises(vr.VrmError):
        rm1.composite(cam, band.data_ptr(), G, W, H, H, 1, out8.data_ptr(), None)       # rows outside the image


@pytest.mark.parametrize("n_slabs", [2, 5])
def test_sort_last_equals_unpartitioned_render(rm1, vr, orc, n_slabs):
    dims = (48, 40, 64)
    vox = synth.quantize_rgba8(synth.blobs_rgba_f32(dims))
    W = H = 96
    cams36 = [sphere_cam(orc, ix, iy) for ix, iy in CAMS]
    cams = [vr.uniform_from_array(c) for c in cams36]
    g = rm1.voxel_grid
    g.new(dims)
    g.upload_rgba8(vox)
    g.attenuation = 40.0
    g.update_voxel_grid_buffer()
    full = _partial(rm1, vr, cams36, W, H).cpu().numpy()
    img = vr.Screenshotter(rm1, W, H)
    parts = []
    for z0, z1 in sortlast.slab_ranges(dims[2], n_slabs):
        s0, s1 = g.new_slab(dims, z0, z1)
        g.upload_rgba8(vox[s0:s1])
        g.attenuation = 40.0
        g.update_voxel_grid_buffer()
        parts.append(_partial(rm1, vr, cams36, W, H))
    stack = torch.stack(parts)                                         # (G, n, H, W, 4)
    g.new(dims)
    g.upload_rgba8(vox)
    g.attenuation = 40.0
    g.update_voxel_grid_buffer()
    for i, cam in enumerate(cams):
        per_view = stack[:, i].contiguous()
        out8 = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0")
        outf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        rm1.composite(cam, per_view.data_ptr(), n_slabs, W, H, 0, H, out8.data_ptr(), outf.data_ptr())
        rm1.draw(cam, img)
        ref8 = img.read()
        assert np.abs(outf.cpu().numpy() - full[i]).max() <= 2e-5          # f32 re-association of the blend only
        assert np.abs(out8.cpu().numpy().astype(int) - ref8.astype(int)).max() <= 1
        assert full[i].max() > 0.05
    img.close()


@pytest.mark.parametrize("ranks,scalar", [(2, "1"), (3, "0")])
def test_sortlast_end_to_end_ranks_on_one_gpu(ranks, scalar):
    """tools/rehearse_sortlast.py: `ranks` processes (one SortLastRenderer each) share GPU 0, march their slabs with the
    HIP kernels, exchange row bands over gloo, composite on the GPU; rank 0 compares with the unpartitioned render."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, REHEARSE_SCALAR=scalar, MASTER_ADDR="127.0.0.1")
    port = 29520 + ranks
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "tools", "rehearse_sortlast.py")],
                       env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "-> OK" in r.stdout
