#!/usr/bin/env python3
"""One rank of the in-library volume broadcast over RCCL (vrm_comm_init / vrm_grid_broadcast) between DIFFERENT GPUs:
    python tools/rccl_broadcast_rank.py <rank> <n_ranks> <id_file> <out_file>
Rank 0 creates the communicator id (written to <id_file>), uploads a seeded RGBA volume and is the root; every other rank
waits for the id, receives into a freshly resized grid and writes the sha256 of what vrm_grid_download_rgba8 returns to
<out_file> -- as rank 0 does.  tests/test_gpu_api.py::test_comm_broadcast_two_ranks starts two of these when the box has two
GPUs (this pool's boxes have one: the test is skipped there and the path stays unmeasured on hardware)."""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import volume_renderer_amd as vr  # noqa: E402
from volume_renderer_amd import synth  # noqa: E402


def main():
    rank, n, id_file, out_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    dims = (40, 36, 44)
    r = vr.RayMarcher(rank, 32, 32)                       # device = rank
    r.voxel_grid.new(dims)
    if rank == 0:
        cid = vr.RayMarcher.comm_unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(bytes(cid))
        os.replace(id_file + ".tmp", id_file)
        r.voxel_grid.upload_rgba8(synth.quantize_rgba8(synth.blobs_rgba_f32(dims)))
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            if time.time() - t0 > 60:
                raise SystemExit("no communicator id from rank 0")
            time.sleep(0.05)
        cid = open(id_file, "rb").read()
    r.comm_init(cid, rank, n)
    r.grid_broadcast(0)
    vox = r.voxel_grid.download_rgba8()
    img = vr.Screenshotter(r, 32, 32)
    import oracle
    r.draw(vr.uniform_from_array(oracle.camera_uniform(oracle.sphere_position(1, 3, 8, 8, 2.5))), img)
    h = hashlib.sha256(np.ascontiguousarray(vox).tobytes() + img.read().tobytes()).hexdigest()
    open(out_file, "w").write(h)
    img.close()
    r.comm_destroy()
    r.close()


if __name__ == "__main__":
    main()
