"""bench.py -- the headline benchmark of the ray-marching path (BASELINE.json configs[2]):
512^3 float32 NetCDF-style volume (DiffDVR layout, quantised to RGBA8 on ingest exactly as
loaders/netcdf.rs does), 256-view sphere export at 1024^2, view-partitioned over N GPUs.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over the whole 256-view export (ray march of every view + the
asynchronous read-back of every image to pinned host memory).  Rank r renders views i with
i % N == r; the only collective is one RCCL broadcast of the volume before the timed region (reported
as volume_broadcast_ms).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def kernel_source_sha(root=ROOT):
    """sha256 (first 16 hex digits) over the sources the march kernel is built from: what a profiles/traffic.json entry was
    measured on, and what the in-tree library should have been built from."""
    import hashlib
    h = hashlib.sha256()
    for f in ("vrm_kernels.hip", "vrm_device.h"):
        with open(os.path.join(root, "volume_renderer_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def traffic_lookup(key, views_per_launch, root=ROOT):
    """(traffic bytes per launch, secondary ceilings, traffic_source) for the roofline object.  HBM-side bytes cannot be
    counted live (PMC needs rocprofv3 around the process): they come from the committed counter passes of the same workload
    (profiles/traffic.json, written by tools/install_profiles.py with the sha of the kernel sources they were measured on).
    An entry measured on OTHER kernel sources than the tree's is not reported: traffic and secondary are null and
    traffic_source says why."""
    src = {"file": "profiles/traffic.json", "key": key, "kernel_sha_built": None, "kernel_sha_measured": None, "measured": None,
           "match": False, "reason": None}
    try:
        src["kernel_sha_built"] = kernel_source_sha(root)
        tj = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    except (OSError, ValueError) as e:
        src["reason"] = f"profiles/traffic.json or the kernel sources could not be read ({e.__class__.__name__})"
        return None, None, src
    ent = tj.get(key)
    if not isinstance(ent, dict) or "hbm_bytes_per_view" not in ent:
        src["reason"] = "no counter pass committed for this workload"
        return None, None, src
    src["kernel_sha_measured"] = ent.get("kernel_sha")
    src["measured"] = ent.get("measured")
    src["passes"] = ent.get("source")
    if ent.get("kernel_sha") != src["kernel_sha_built"]:
        src["reason"] = "the committed counter pass was measured on other kernel sources than this tree's"
        return None, None, src
    src["match"] = True
    return int(ent["hbm_bytes_per_view"] * views_per_launch), ent.get("secondary"), src


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 outside torch.distributed.run: start the N ranks as a CHILD
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <same arguments>` (one process per GPU,
    RCCL between them), pass its output through and return its exit code.  Called before torch is imported or
    anything touches the GPU; the current process is never replaced."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this pool
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    json_lines = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if ln not in json_lines:
            print(ln, file=sys.stderr)
    if json_lines:
        print(json_lines[-1], flush=True)               # rank 0's ONE line
    return proc.returncode if proc.returncode != 0 or json_lines else 1


def launcher_selftest(a):
    """VRM_BENCH_LAUNCH_TEST=1 (tests/test_bench_launcher_cpu.py): the rank plumbing of this file without a GPU --
    gloo rendezvous, world-size check, barrier, max/sum over ranks, rank 0 prints one line."""
    from volume_renderer_amd import distributed as vd
    rank, _, world = vd.env_rank()
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    if world > 1:
        import torch.distributed as dist
        vd.init_process_group("gloo")
        assert dist.get_world_size() == a.gpus
    vd.barrier()
    t = vd.max_over_ranks(1.0 + rank)
    n = vd.sum_over_ranks(len(vd.partition_views(256, rank, world)))
    per_rank = vd.gather_over_ranks(1.0 + rank)
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "max_time": t, "views": int(n), "steps": a.steps,
                          "warmup": a.warmup, "per_rank": per_rank}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def run_c5(a):
    """C5: sort-last render of a dim x dim x 2*dim scalar volume cut into one z slab per rank.  Every rank
    generates and ingests ONLY its slab (+ ghost slices), marches the whole volume's sample lattice restricted to its
    slab into f32 partials, one RCCL all-to-all moves row bands, every rank composites its band.  One step = all views.
    The volume never exists on one GPU, so "scaling" is strong in the sense that the per-rank share shrinks with N."""
    import numpy as np
    import torch

    import volume_renderer_amd as vr
    from volume_renderer_amd import distributed as vd
    from volume_renderer_amd import sortlast, synth

    rank, local_rank, world = vd.env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ray-marching path has no CPU fallback")
    rehearse = os.environ.get("VRM_BENCH_REHEARSE", "0") == "1"   # all ranks on GPU 0, exchange over gloo (see main())
    gpu = 0 if rehearse else local_rank
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    coll_dev = "cpu" if rehearse else dev
    if world > 1:
        vd.init_process_group("gloo" if rehearse else "nccl")
    D, R = a.c5_dim, a.c5_res
    dims = (D, D, 2 * D)
    rm = vr.RayMarcher(gpu, R, R)
    rm.set_option(vr.OPT_KERNEL, a.kernel)
    rm.set_option(vr.OPT_SKIP_EMPTY, a.skip_empty)
    slr = sortlast.SortLastRenderer(rm, dims)
    t0 = time.time()
    slr.upload_scalar_u8(synth.blobs_scalar_u8_slices(dims, slr.s0, slr.s1))
    rm.voxel_grid.attenuation = a.attenuation
    rm.voxel_grid.update_voxel_grid_buffer()
    rm.synchronize()
    ingest_s = time.time() - t0

    cam = vr.Camera(1.0)
    xdiv, ydiv = 8, max(2, a.c5_views // 8 + 1)
    csp = vr.CameraSphereController(ydiv, xdiv, (0.0, 0.0, 0.0), a.radius)
    cams = []
    for ix, iy in vr.sphere_views(xdiv, ydiv):
        csp.current_index_x, csp.current_index_y = ix, iy
        csp.update_position(cam)
        cams.append(cam.uniform())
    B = max(1, min(4, len(cams)))       # views per exchange: 4 * 64 MiB of f32 partials per rank at 2048^2

    def step():
        last = None
        for c0 in range(0, len(cams), B):
            last = slr.render(cams[c0:c0 + B], R, R, dev, host_exchange=rehearse)
        return last

    n_ref = rm.count_reference_samples(cams, R, R)
    for _ in range(a.warmup):
        step()
    vd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    vd.barrier()
    elapsed = vd.max_over_ranks(time.perf_counter() - t0, coll_dev)

    # where the time goes on this rank: march of the partials vs exchange + composite (one more, instrumented, step)
    rm.set_option(vr.OPT_PROFILE, 1)
    rm.profile_read()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    part = torch.empty((B, R, R, 4), dtype=torch.float32, device=dev)
    rm.draw_partial(cams[:B], R, R, part.data_ptr())
    rm.synchronize()
    march_ms = (time.perf_counter() - t1) * 1e3
    stored = slr.s1 - slr.s0
    # this rank: its slab read once (scalar volume: the packed alpha copy, 1 B per voxel), its f32 partial written once
    b_alg_view = dims[0] * dims[1] * stored * 1 + R * R * 16
    achieved = b_alg_view * B / (march_ms * 1e-3) / 1e9
    out = {
        "metric": "views/s", "value": round(len(cams) * a.steps / elapsed, 3), "unit": "views/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"C5: {dims[0]}x{dims[1]}x{dims[2]} scalar volume (blobs seed {synth.DEFAULT_SEED}) in {world} z slab(s) "
                               f"with ghost slices, sort-last: f32 partials, direct-send all-to-all of row bands, over-composite; "
                               f"{len(cams)} views at {R}x{R}, radius {a.radius}, attenuation {a.attenuation}",
                   "volume": list(dims), "image": [R, R], "views": len(cams), "parallelism": f"z-slab sort-last x{world}",
                   "kernel_variant": a.kernel, "skip_empty": a.skip_empty},
        "gsamples_per_s": round(n_ref * a.steps / elapsed / 1e9, 3), "ms_per_view": round(elapsed / a.steps / len(cams) * 1e3, 4),
        "slab_ingest_s": round(ingest_s, 2),
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "kernel": "march kernel (variant %d), rank 0's slab" % a.kernel,
                     "avg_launch_ms": round(march_ms, 3), "algorithmic_bytes_per_view": b_alg_view, "views_per_launch": B},
        "cpu_baseline": None,
    }
    if rehearse:
        out["rehearsal"] = True
    if rank == 0:
        print(json.dumps(out), flush=True)
    rm.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def run_c1(a):
    """C1 (BASELINE.json configs[0]): synthetic 128^3 uint8 DAT volume, ONE 512 x 512 view -- (ix, iy) =
    (1, 3) of the 8 x 8 sphere at radius 2.5, attenuation 1, transfer function off -- on the CPU path with the core count
    stated, and the HIP path on the same view beside it.  The CPU path is oracle/vr_oracle.c (a restatement: the wgpu
    reference on a software adapter cannot be built here); one step = that one view."""
    import numpy as np
    import torch  # noqa: F401  (first: the HIP library then shares torch's libamdhip64)

    import oracle
    import volume_renderer_amd as vr
    from volume_renderer_amd import synth

    D, R = 128, 512
    dens = synth.blobs_scalar_u8((D, D, D))
    vox = np.repeat(dens[..., None], 4, axis=3)                      # [b,b,b,b] (loaders/dat.rs:74)
    cam36 = oracle.camera_uniform(oracle.sphere_position(1, 3, 8, 8, 2.5))
    threads = max(1, min(oracle.lib().orc_max_threads(), os.cpu_count() or 1, 64))
    t0 = time.perf_counter()
    reps, ref = 0, None
    while reps < 1 or (time.perf_counter() - t0 < 10.0 and reps < 50):
        ref = oracle.render(vox, cam36, R, R, want_f32=False, threads=threads)
        reps += 1
    cpu_s = (time.perf_counter() - t0) / reps
    rm = vr.RayMarcher(0, R, R)
    rm.set_option(vr.OPT_KERNEL, a.kernel)
    rm.set_option(vr.OPT_SKIP_EMPTY, a.skip_empty)
    rm.voxel_grid.upload_scalar(dens.reshape(-1), (D, D, D), 1)      # the DAT ingest path (vrm_grid_upload_scalar)
    cam = vr.uniform_from_array(cam36)
    img = vr.Screenshotter(rm, R, R)
    n_ref = rm.count_reference_samples([cam], R, R)
    for _ in range(max(1, a.warmup)):
        rm.draw(cam, img)
        img.screenshot()
        img.wait()
    rm.set_option(vr.OPT_PROFILE, 1)
    rm.profile_read()
    steps = max(a.steps, 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        rm.draw(cam, img)
        img.screenshot()
        img.wait()                                                    # one view per "frame", read back, like the reference's export
    rm.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = rm.profile_read()
    got = img.read()
    d = np.abs(got.astype(np.int32) - ref["rgba8"].astype(np.int32))
    b_alg = D * D * D * 1 + R * R * 4                                 # packed 1-channel march: 1 B per voxel + the RGBA8 image
    achieved = b_alg * launches / (kernel_ms * 1e-3) / 1e9
    out = {
        "metric": "views/s", "value": round(steps / elapsed, 3), "unit": "views/s", "n_gpus": 1, "steps": steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"C1: {D}^3 uint8 DAT-style volume (blobs seed {synth.DEFAULT_SEED}), single {R}x{R} view (1,3) of the 8x8 "
                               f"sphere, radius 2.5, attenuation 1.0, transfer function off; one draw + read-back per step",
                   "volume": [D, D, D], "image": [R, R], "views": 1, "parallelism": "single view", "kernel_variant": a.kernel,
                   "skip_empty": a.skip_empty},
        "gsamples_per_s": round(n_ref * steps / elapsed / 1e9, 3), "reference_samples_per_step": int(n_ref),
        "parity_vs_cpu_path": {"max_abs_lsb": int(d.max()), "rmse_normalised": float(np.sqrt(np.mean(d.astype(np.float64) ** 2)) / 255.0)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None, "kernel": "march kernel (variant %d)" % a.kernel,
                     "launches": int(launches), "avg_launch_ms": round(kernel_ms / max(1, launches), 4),
                     "algorithmic_bytes_per_view": b_alg, "views_per_launch": 1},
        "cpu_baseline": {"value": round(1.0 / cpu_s, 4), "unit": "views/s", "cores": threads, "kind": "port",
                         "sample": f"oracle/vr_oracle.c (CPU restatement of raymarcher.wgsl, NOT the wgpu reference on a software adapter, "
                                   f"which cannot be built here), the whole C1 view, mean of {reps} runs of {cpu_s:.3f} s, OpenMP over "
                                   f"image rows; {ref['samples']} samples per view",
                         "gsamples_per_s": round(ref["samples"] / cpu_s / 1e9, 5)},
    }
    print(json.dumps(out), flush=True)
    img.close()
    rm.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dim", type=int, default=512, help="volume edge (C3: 512)")
    ap.add_argument("--res", type=int, default=1024, help="image edge (C3: 1024)")
    ap.add_argument("--xdiv", type=int, default=16)
    ap.add_argument("--ydiv", type=int, default=17, help="views = xdiv*(ydiv-1) (C3: 16*16 = 256)")
    ap.add_argument("--radius", type=float, default=2.5)
    ap.add_argument("--attenuation", type=float, default=1.0)
    ap.add_argument("--tf", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0, help="views per launch; 0 = automatic: 64 for images up to 1024^2 and at least 256 views per rank, else 32 "
                    "(measured on C3, profiles/r3/batch_sweep.txt: 16 -> 3171, 32 -> 3315, 64 -> 3370, 128 -> 3306 views/s; "
                    "larger launches have shorter tails, and the two banks of pinned images stay below 1 GiB)")
    ap.add_argument("--kernel", type=int, default=1, help="march kernel variant (0: global gathers, 1: LDS-staged)")
    ap.add_argument("--skip-empty", type=int, default=1, help="exact empty-space skipping (bit-identical image); "
                    "the no-skip rate is measured too and reported as views_per_s_no_skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-views", type=int, default=16, help="views of the export rendered by the CPU baseline (at most)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="stop the CPU baseline after this much CPU-side wall time")
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4", "C5", "custom"],
                    help="BASELINE.json configs[0..4]; C3 is the headline and the default; 'custom' takes --dim/--res/... as given; "
                         "C1 = the single-view CPU-path config with the HIP path beside it, see run_c1(); "
                         "C5 = sort-last over a z-slab-partitioned volume (SURVEY 8e), see run_c5()")
    ap.add_argument("--c5-dim", type=int, default=1024, help="C5: volume is dim x dim x 2*dim scalar (SURVEY's 2048 needs 8 GPUs)")
    ap.add_argument("--c5-views", type=int, default=16)
    ap.add_argument("--c5-res", type=int, default=2048)
    ap.add_argument("--volume", default=None, choices=["rgba", "scalar"],
                    help="rgba: f32 color(c,z,y,x) NetCDF-style; scalar: DAT-style density replicated to r=g=b=a")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    if os.environ.get("VRM_BENCH_LAUNCH_TEST", "0") == "1":
        return launcher_selftest(a)
    presets = {"C2": dict(dim=256, res=512, xdiv=8, ydiv=9, volume="scalar"),     # 256^3 u8 DAT, 64 views, 512^2
               "C3": dict(dim=512, res=1024, xdiv=16, ydiv=17, volume="rgba"),    # 512^3 f32 RGBA, 256 views, 1024^2
               "C4": dict(dim=1024, res=2048, xdiv=16, ydiv=9, volume="scalar")}  # 1024^3 f32 scalar, 128 views, 2048^2
    if a.config == "C1":
        return run_c1(a)
    if a.config == "C5":
        return run_c5(a)
    if a.config in presets:
        for k, v in presets[a.config].items():
            if k == "volume":
                a.volume = a.volume or v
            else:
                setattr(a, k, v)
    a.volume = a.volume or "rgba"

    import numpy as np
    import torch  # first: the HIP library then shares torch's libamdhip64 instead of loading a second runtime

    import volume_renderer_amd as vr
    from volume_renderer_amd import distributed as vd
    from volume_renderer_amd import synth

    rank, local_rank, world = vd.env_rank()
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch as `python bench.py --gpus N` or as "
                         f"`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ray-marching path has no CPU fallback")
    # VRM_BENCH_REHEARSE=1: a logic rehearsal of the N-rank flow on a box with ONE GPU -- every rank renders on GPU 0 and
    # the collectives run over gloo on host tensors.  The line it prints carries "rehearsal": true and is not a measurement.
    rehearse = os.environ.get("VRM_BENCH_REHEARSE", "0") == "1"
    gpu = 0 if rehearse else local_rank
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    coll_dev = "cpu" if rehearse else dev
    if world > 1:
        vd.init_process_group("gloo" if rehearse else "nccl")
        import torch.distributed as dist
        assert dist.get_world_size() == a.gpus

    D, R = a.dim, a.res
    dims = (D, D, D)
    rm = vr.RayMarcher(gpu, R, R)
    rm.set_option(vr.OPT_KERNEL, a.kernel)
    rm.set_option(vr.OPT_SKIP_EMPTY, a.skip_empty)

    # ---- volume: rank 0 builds it, one RCCL broadcast, every rank ingests from device memory -----------
    t0 = time.time()
    host_q = None
    if rank == 0:
        if a.volume == "rgba":
            host_q = synth.quantize_rgba8(synth.blobs_rgba_f32(dims))  # (v*255.0) as u8, loaders/netcdf.rs:54-57
        else:
            host_q = np.repeat(synth.blobs_scalar_u8(dims)[..., None], 4, axis=3)  # [b,b,b,b], loaders/dat.rs:74
        vol = torch.from_numpy(host_q).to(dev)
    else:
        vol = torch.empty((D, D, D, 4), dtype=torch.uint8, device=dev)
    gen_s = time.time() - t0
    vd.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    if rehearse and world > 1:
        vol = vd.broadcast_volume(vol.cpu(), src=0).to(dev)
    else:
        vd.broadcast_volume(vol, src=0)
    torch.cuda.synchronize()
    bcast_ms = (time.time() - t0) * 1e3 if world > 1 else 0.0
    rm.voxel_grid.new(dims)
    rm.voxel_grid.upload_rgba8_device(vol.data_ptr(), vol.numel())
    rm.voxel_grid.attenuation = a.attenuation
    rm.voxel_grid.update_voxel_grid_buffer()
    rm.voxel_grid.set_transfer_function_active(bool(a.tf))
    rm.voxel_grid.update_transfer_function_buffer()
    del vol
    torch.cuda.empty_cache()

    # ---- view list (sphere_screenshot_manager.rs order), interleaved partition ------------------------
    cam = vr.Camera(1.0)
    csp = vr.CameraSphereController(a.ydiv, a.xdiv, (0.0, 0.0, 0.0), a.radius)
    cams = []
    for ix, iy in vr.sphere_views(a.xdiv, a.ydiv):
        csp.current_index_x, csp.current_index_y = ix, iy
        csp.update_position(cam)
        cams.append(cam.uniform())
    n_views = len(cams)
    mine = vd.partition_views(n_views, rank, world, a.xdiv)
    my_cams = [cams[i] for i in mine]
    # (64 only where that still leaves four launches to pipeline march and read-back: C2's 64 views in ONE launch of 64 ran at
    # 8.2 k views/s against 13.1 k in two of 32)
    B = max(1, min(a.batch if a.batch > 0 else (64 if R * R <= 1024 * 1024 and len(my_cams) >= 256 else 32), len(my_cams)))
    banks = [[vr.Screenshotter(rm, R, R) for _ in range(B)] for _ in range(2)]

    which = [0]                      # the banks alternate across steps too: with one launch per step (8 ranks x 32 views)
                                     # the read-back of step k overlaps the march of step k+1

    def step():
        for c0 in range(0, len(my_cams), B):
            chunk = my_cams[c0:c0 + B]
            imgs = banks[which[0]][:len(chunk)]
            for im in imgs:          # this bank's previous read-back must have landed before it is redrawn
                im.wait()
            rm.draw_batch(chunk, imgs)
            for im in imgs:
                im.screenshot()      # async device->host copy behind the draw
            which[0] ^= 1

    n_ref_mine = rm.count_reference_samples(my_cams, R, R) if my_cams else 0

    for _ in range(a.warmup):
        step()
    rm.synchronize()
    rm.set_option(vr.OPT_PROFILE, 1)
    rm.profile_read()
    vd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    rm.synchronize()             # compute stream and copy stream of the library
    torch.cuda.synchronize()
    vd.barrier()
    elapsed_mine = time.perf_counter() - t0
    elapsed = vd.max_over_ranks(elapsed_mine, coll_dev)
    kernel_ms, launches = rm.profile_read()
    rm.set_option(vr.OPT_PROFILE, 0)
    # per-rank spread, so that a bad N-GPU point can be attributed: every rank's own wall time and kernel time
    per_rank_ms = vd.gather_over_ranks(elapsed_mine / a.steps * 1e3, coll_dev)
    per_rank_kernel_ms = vd.gather_over_ranks(kernel_ms / a.steps, coll_dev)
    n_ref = vd.sum_over_ranks(n_ref_mine, coll_dev)

    # the same export with skipping off (every reference sample actually taken), one step, for transparency
    no_skip = None
    if a.skip_empty:
        rm.set_option(vr.OPT_SKIP_EMPTY, 0)
        step()
        rm.synchronize()
        vd.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        rm.synchronize()
        torch.cuda.synchronize()
        vd.barrier()
        no_skip = n_views / vd.max_over_ranks(time.perf_counter() - t0, coll_dev)
        rm.set_option(vr.OPT_SKIP_EMPTY, 1)

    # samples actually filtered (N_exec), how many of them came out with alpha exactly 0 (what finer skipping could
    # still save) and the chunk statistics: one pass of this rank's views through the diagnostics build
    rm.set_option(vr.OPT_DIAGNOSTICS, 1)
    rm.diag_counters()
    for cam_u in my_cams:
        rm.draw(cam_u, banks[0][0])
    dc = rm.diag_counters()
    rm.set_option(vr.OPT_DIAGNOSTICS, 0)
    n_exec = vd.sum_over_ranks(dc["executed"], coll_dev)
    n_exec_zero = vd.sum_over_ranks(dc["executed_zero_alpha"], coll_dev)
    chunk_stats = {k: int(vd.sum_over_ranks(dc[k], coll_dev)) for k in ("chunks", "chunks_staged", "chunks_empty", "probes_failed",
                                                                         "chunks_gathered", "box_retries")}

    views_per_s = n_views * a.steps / elapsed
    gsamples = n_ref * a.steps / elapsed / 1e9

    # ---- roofline of the march kernel on this rank: algorithmic bytes = every voxel read once (4 B, the
    # reference's own texel format) + every RGBA8 pixel written once, per view -------------
    # (the 1-channel march -- scalar volume or transfer function -- reads the packed alpha copy: 1 B per voxel)
    vox_bytes = 1 if (a.volume != "rgba" or a.tf) else 4
    b_alg_view = D * D * D * vox_bytes + R * R * 4
    views_timed = len(my_cams) * a.steps
    achieved = (b_alg_view * views_timed / (kernel_ms * 1e-3) / 1e9) if kernel_ms > 0 else 0.0
    key = f"{a.config}:kernel{a.kernel}:skip{a.skip_empty}" + (":tf" if a.tf else "")
    traffic, secondary, traffic_source = traffic_lookup(key, B)
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "kernel": "march kernel (variant %d)" % a.kernel, "launches": int(launches),
                "avg_launch_ms": round(kernel_ms / max(1, launches), 4),
                "algorithmic_bytes_per_view": b_alg_view, "views_per_launch": B, "secondary": secondary,
                "traffic_source": traffic_source}
    # SURVEY.md 8(d) prices the voxels in the INPUT file's format (f32 RGBA = 16 B, f32 scalar = 4 B); the kernel
    # reads the RGBA8 texels the reference's own texture holds (4 B; 1 B for the 1-channel march), so `achieved` and
    # `frac` above use those (the lower figures). The survey-priced pair is carried beside them so either convention
    # can be checked.
    in_bytes = 16 if a.volume == "rgba" else 4
    b_alg_survey = D * D * D * in_bytes + R * R * 4
    ach_survey = achieved * b_alg_survey / b_alg_view
    roofline["survey_8d"] = {"algorithmic_bytes_per_view": b_alg_survey, "achieved": round(ach_survey, 2),
                             "frac": round(ach_survey / HBM_PEAK_GBS, 5)}

    out = {
        "metric": "views/s", "value": round(views_per_s, 3), "unit": "views/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{a.config}: {D}^3 " + ("float32 RGBA volume" if a.volume == "rgba" else "scalar (DAT-style) volume") +
                               f" (blobs seed {synth.DEFAULT_SEED}, quantised to RGBA8 on ingest), "
                               f"{n_views}-view sphere export at {R}x{R}, radius {a.radius}, attenuation {a.attenuation}, "
                               f"transfer function {'on' if a.tf else 'off'}",
                   "volume": [D, D, D], "image": [R, R], "views": n_views, "parallelism": f"view-partitioned x{world}",
                   "kernel_variant": a.kernel, "skip_empty": a.skip_empty},
        "gsamples_per_s": round(gsamples, 3), "reference_samples_per_step": int(n_ref),
        "executed_samples_per_step": int(n_exec), "executed_empty_samples_per_step": int(n_exec_zero),
        "chunks_per_step": chunk_stats,
        "ms_per_view": round(elapsed / a.steps / n_views * 1e3, 4),
        "ms_per_step_min": round(min(per_rank_ms), 3), "ms_per_step_max": round(max(per_rank_ms), 3),
        "per_rank": {"ms_per_step": [round(v, 3) for v in per_rank_ms], "kernel_ms_per_step": [round(v, 3) for v in per_rank_kernel_ms],
                     "views": [len(vd.partition_views(n_views, r_, world, a.xdiv)) for r_ in range(world)]},
        "volume_broadcast_ms": round(bcast_ms, 2),
        "volume_broadcast_gbs": round(D * D * D * 4 / (bcast_ms * 1e-3) / 1e9, 2) if bcast_ms > 0 else None,
        "rccl_ranks": 0 if (rehearse or world == 1) else world, "roofline": roofline,
        "views_per_s_no_skip": None if no_skip is None else round(no_skip, 3),
    }

    # ---- CPU baseline: the oracle ("port"; the wgpu reference cannot run here) on a bounded sample -------
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        import oracle
        threads = max(1, min(16, os.cpu_count() or 1))
        # a bounded sample of the same workload: every (n_views / cpu_views)-th view of the export, full size,
        # until the list or the time budget runs out
        allv = vr.sphere_views(a.xdiv, a.ydiv)
        picks = allv[:: max(1, n_views // max(1, a.cpu_views))][: a.cpu_views]
        cpu_samples, cpu_s, done = 0, 0.0, 0
        for ix, iy in picks:
            cam36 = oracle.camera_uniform(oracle.sphere_position(ix, iy, a.xdiv, a.ydiv, a.radius))
            t0 = time.perf_counter()
            res = oracle.render(host_q, cam36, R, R, attenuation=a.attenuation, tf=oracle.tf_words(bool(a.tf)), want_f32=False,
                                threads=threads)
            cpu_s += time.perf_counter() - t0
            cpu_samples += res["samples"]
            done += 1
            if cpu_s > a.cpu_seconds:
                break
        out["cpu_baseline"] = {
            "value": round(done / cpu_s, 4), "unit": "views/s", "cores": threads, "kind": "port",
            "sample": f"oracle/vr_oracle.c (CPU restatement of raymarcher.wgsl, NOT the wgpu reference on a software adapter, "
                      f"which cannot be built here), {done} of the {n_views} views of the same export at {R}x{R} "
                      f"({cpu_samples} samples in {cpu_s:.2f} s, OpenMP over image rows)",
            "gsamples_per_s": round(cpu_samples / cpu_s / 1e9, 5),
        }
    if rehearse:
        out["rehearsal"] = True   # N ranks on one GPU over gloo: logic check only, never a measurement
    if rank == 0:
        print(json.dumps(out), flush=True)
    for bank in banks:
        for im in bank:
            im.close()
    rm.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
