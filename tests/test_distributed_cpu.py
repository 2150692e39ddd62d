"""N>1 path on CPU: world_size-2 gloo run of the view-partitioned export plumbing (volume broadcast,
interleaved view partition, barrier, max-over-ranks timing).  The renders inside the ranks use the CPU
oracle as a stand-in checker at a tiny size -- the product path itself only exists on the GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from volume_renderer_amd import distributed as vd


def test_partition_is_an_interleaved_exact_cover():
    for n, world in [(56, 1), (56, 2), (256, 8), (64, 3), (5, 8)]:
        parts = [vd.partition_views(n, r, world) for r in range(world)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
        assert all(p == list(range(r, n, world)) for r, p in enumerate(parts))
    # the export's form: the interleave rotated by one per row of x_div views -- still an exact cover, rows split evenly
    # when world divides x_div, and no rank holds the same column in two consecutive rows
    for n, world, x_div in [(256, 8, 16), (256, 4, 16), (56, 2, 8), (60, 3, 12), (20, 8, 5), (7, 2, 1)]:
        parts = [vd.partition_views(n, r, world, x_div) for r in range(world)]
        assert sorted(i for p in parts for i in p) == list(range(n))
        if x_div % world == 0 and n % x_div == 0:
            assert len({len(p) for p in parts}) == 1
        for r, p in enumerate(parts):
            assert all((i + i // x_div) % world == r for i in p)
            cols = {}
            for i in p:
                cols.setdefault(i // x_div, set()).add(i % x_div)
            if world > 1 and x_div % world == 0:
                assert all(not (cols[k] & cols[k + 1]) for k in cols if k + 1 in cols)
    with pytest.raises(ValueError):
        vd.partition_views(10, 2, 2)


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
    from volume_renderer_amd import synth
    d = vd.init_process_group("gloo")
    assert d is not None and dist.get_world_size() == world
    dims = (12, 10, 8)
    if rank == 0:
        vol = torch.from_numpy(synth.quantize_rgba8(synth.blobs_rgba_f32(dims)))
    else:
        vol = torch.zeros((dims[2], dims[1], dims[0], 4), dtype=torch.uint8)
    vd.broadcast_volume(vol, src=0)                       # the one collective of the path
    vox = vol.numpy()
    views = oracle.sphere_views(4, 4)
    mine = vd.partition_views(len(views), rank, world)
    out = {}
    for i in mine:
        ix, iy = views[i]
        cam = oracle.camera_uniform(oracle.sphere_position(ix, iy, 4, 4, 2.5))
        out[i] = oracle.render(vox, cam, 16, 16, attenuation=5.0, threads=1)["rgba8"]
    vd.barrier()
    t = vd.max_over_ranks(0.25 + rank)                    # max over ranks of a per-rank time
    n = vd.sum_over_ranks(len(mine))
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        q.put((t, n, gathered, vox.copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_export_partition(orc):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    t, n, gathered, vox = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    from volume_renderer_amd import synth
    assert np.array_equal(vox, synth.quantize_rgba8(synth.blobs_rgba_f32((12, 10, 8))))   # broadcast delivered rank 0's volume
    assert t == 1.25 and n == 12
    views = orc.sphere_views(4, 4)
    merged = {}
    for part in gathered:
        assert not (set(part) & set(merged))
        merged.update(part)
    assert sorted(merged) == list(range(len(views)))
    assert sorted(gathered[0]) == list(range(0, 12, 2)) and sorted(gathered[1]) == list(range(1, 12, 2))
    for i, (ix, iy) in enumerate(views):
        cam = orc.camera_uniform(orc.sphere_position(ix, iy, 4, 4, 2.5))
        assert np.array_equal(merged[i], orc.render(vox, cam, 16, 16, attenuation=5.0)["rgba8"])
