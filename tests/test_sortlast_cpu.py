"""Sort-last on the CPU: the oracle's slab partials compose to the oracle's full image, the partition and
band helpers are exact covers, and the world-size-2 gloo run of the one exchange step (all-to-all of row bands + local
"over") reproduces the single-process composite."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, sphere_cam
from volume_renderer_amd import sortlast, synth


def test_slab_ranges_and_bands_are_exact_covers():
    for z, n in [(50, 1), (50, 2), (50, 7), (4096, 8), (3, 3)]:
        r = sortlast.slab_ranges(z, n)
        assert r[0][0] == 0 and r[-1][1] == z and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
        for a, b in r:
            s0, s1 = sortlast.stored_range(z, a, b)
            assert s0 == max(a - 1, 0) and s1 == min(b + 1, z)
    with pytest.raises(ValueError):
        sortlast.slab_ranges(3, 4)
    for h, n in [(1024, 8), (100, 3), (5, 8)]:
        b = sortlast.row_bands(h, n)
        assert sum(k for _, k in b) == h and all(b[i][0] + b[i][1] == b[i + 1][0] for i in range(n - 1))


# camera positions that make rd.z positive, negative and of mixed sign across the image
CAMS = [(1, 3), (3, 4), (6, 2), (0, 4), (4, 4)]


@pytest.mark.parametrize("n_slabs", [2, 3, 5])
def test_oracle_partials_compose_to_the_full_image(orc, n_slabs):
    dims = (14, 12, 20)
    vox = synth.random_rgba8(dims, 9)
    W = H = 24
    for ix, iy in CAMS:
        cam = sphere_cam(orc, ix, iy)
        full = orc.render(vox, cam, W, H, attenuation=30.0)
        parts = np.stack([orc.render_slab(vox, cam, W, H, a, b, attenuation=30.0) for a, b in sortlast.slab_ranges(dims[2], n_slabs)])
        asc = sortlast.ascending_mask(cam, W, H)
        comp = sortlast.composite_numpy(parts, asc)
        assert np.abs(comp - full["f32"]).max() <= 2e-5            # f32 re-association of a ~1000-term blend only
        assert np.abs(sortlast.unorm8(comp).astype(int) - full["rgba8"].astype(int)).max() <= 1
        # the wrong order is visibly wrong wherever more than one slab contributes (guards the order logic)
        if (ix, iy) == (1, 3):                                      # an oblique view: every ray crosses slab boundaries
            wrong = sortlast.composite_numpy(parts, ~asc)
            assert np.abs(wrong - full["f32"]).max() > 1e-4
    # mixed-sign view really is mixed (camera in the z = 0 plane region)
    assert len({bool(v) for v in sortlast.ascending_mask(sphere_cam(orc, 0, 4), W, H).ravel()}) == 2


def test_single_slab_partial_is_the_full_render(orc):
    dims = (10, 9, 8)
    vox = synth.random_rgba8(dims, 1)
    cam = sphere_cam(orc, 2, 3)
    full = orc.render(vox, cam, 20, 20, attenuation=8.0)
    part = orc.render_slab(vox, cam, 20, 20, 0, dims[2], attenuation=8.0)
    assert np.array_equal(part, full["f32"])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import oracle
    from volume_renderer_amd import distributed as vd
    from volume_renderer_amd import sortlast as sl
    vd.init_process_group("gloo")
    dims = (12, 10, 16)
    vox = synth.random_rgba8(dims, 9)                              # every rank could hold just its slab; the oracle takes the whole
    W, H = 20, 19                                                 # odd height: uneven bands
    z0, z1 = sl.slab_ranges(dims[2], world)[rank]
    cams = [oracle.camera_uniform(oracle.sphere_position(ix, iy, 8, 8, 2.5)) for ix, iy in [(1, 3), (0, 4), (5, 5)]]
    partial = np.stack([oracle.render_slab(vox, c, W, H, z0, z1, attenuation=30.0, threads=1) for c in cams])   # (n, H, W, 4)
    stack, row0, rows = sl.exchange_bands(torch.from_numpy(partial))
    assert stack.shape == (world, len(cams), rows, W, 4) and (row0, rows) == sl.row_bands(H, world)[rank]
    band = np.stack([sl.composite_numpy(stack[:, i].numpy(), sl.ascending_mask(cams[i], W, H, row0, rows)) for i in range(len(cams))])
    gathered = [None] * world
    dist.all_gather_object(gathered, (row0, rows, band))
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sort_last_exchange(orc):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    dims = (12, 10, 16)
    vox = synth.random_rgba8(dims, 9)
    W, H = 20, 19
    img = np.concatenate([g[2] for g in sorted(gathered, key=lambda g: g[0])], axis=1)     # bands back together: (n, H, W, 4)
    assert img.shape == (3, H, W, 4)
    for i, (ix, iy) in enumerate([(1, 3), (0, 4), (5, 5)]):
        cam = orc.camera_uniform(orc.sphere_position(ix, iy, 8, 8, 2.5))
        full = orc.render(vox, cam, W, H, attenuation=30.0)
        assert np.abs(img[i] - full["f32"]).max() <= 2e-5
        assert np.abs(sortlast.unorm8(img[i]).astype(int) - full["rgba8"].astype(int)).max() <= 1


def test_oracle_slab_from_stored_slices_only(orc):
    """oracle.render_slab_stored (the checker of the true-size C5 GPU test): the partial of a slab rendered from its stored
    slices alone -- RGBA8 or one byte per voxel, whole image or a crop -- is bit-identical to render_slab on the whole
    volume, and a texel outside the stored range would show as NaN."""
    dims = (20, 16, 24)
    vox = synth.quantize_rgba8(synth.blobs_rgba_f32(dims))
    cam = orc.camera_uniform(orc.sphere_position(1, 3, 8, 8, 2.5))
    W, H = 48, 40
    for z0, z1 in [(0, 9), (8, 16), (16, 24)]:
        s0, s1 = sortlast.stored_range(dims[2], z0, z1)
        full = orc.render_slab(vox, cam, W, H, z0, z1, attenuation=20.0)
        part, steps = orc.render_slab_stored(vox[s0:s1], dims, s0, cam, W, H, z0, z1, attenuation=20.0)
        assert np.array_equal(full, part)
        crop, csteps = orc.render_slab_stored(vox[s0:s1], dims, s0, cam, W, H, z0, z1, crop=(10, 5, 20, 30), attenuation=20.0)
        assert np.array_equal(full[5:35, 10:30], crop) and np.array_equal(steps[5:35, 10:30], csteps)
    # without the ghost slice the filter reaches outside the stored range: NaN, loudly
    bad, _ = orc.render_slab_stored(vox[8:16], dims, 8, cam, W, H, 8, 16, attenuation=20.0)
    assert np.isnan(bad).any()
    # scalar (DAT-style) storage == the replicated RGBA8 volume; whole-volume form == orc.render
    dens = synth.blobs_scalar_u8(dims)
    v4 = np.repeat(dens[..., None], 4, axis=3)
    full = orc.render(v4, cam, W, H, attenuation=20.0, want_steps=True)
    part, steps = orc.render_slab_stored(dens, dims, 0, cam, W, H, 0, dims[2], attenuation=20.0)
    assert np.array_equal(full["f32"], part) and np.array_equal(full["steps"], steps)
