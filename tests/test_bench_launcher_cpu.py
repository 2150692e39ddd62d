"""`python bench.py --gpus N` must start N ranks by itself: the launcher branch is run here on CPU
with the rank body replaced by its GPU-free self-test (VRM_BENCH_LAUNCH_TEST=1: gloo rendezvous, world-size check,
barrier, max/sum over ranks).  Exactly one JSON line must come back, from rank 0, with n_gpus == N."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env["VRM_BENCH_LAUNCH_TEST"] = "1"
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout)


@pytest.mark.parametrize("n", [2, 3])
def test_gpus_n_spawns_n_ranks_and_prints_one_line(n):
    p = _run(["--gpus", str(n), "--steps", "2", "--warmup", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["launcher_selftest"] is True
    assert out["max_time"] == float(n) and out["views"] == 256     # max over ranks of 1 + rank; the partition covers every view
    assert out["steps"] == 2 and out["warmup"] == 1                 # the arguments reached the ranks unchanged
    assert out["per_rank"] == [float(r + 1) for r in range(n)]       # the per-rank spread of the bench line, in rank order


def test_single_rank_does_not_spawn():
    p = _run(["--gpus", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip())
    assert out["n_gpus"] == 1


def test_world_size_mismatch_is_an_error():
    # under torch.distributed.run with a different N the ranks must refuse, not silently report WORLD_SIZE
    p = _run(["--gpus", "4"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0
    assert "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_traffic_counters_are_reported_only_for_the_kernel_they_were_measured_on(tmp_path):
    """roofline.traffic / roofline.secondary come from committed rocprofv3 passes; an entry measured on other
    kernel sources than the tree's must not be replayed as if it were this kernel's."""
    sys.path.insert(0, ROOT)
    import bench
    root = tmp_path
    os.makedirs(root / "volume_renderer_amd" / "csrc")
    os.makedirs(root / "profiles")
    (root / "volume_renderer_amd" / "csrc" / "vrm_kernels.hip").write_text("kernel v1")
    (root / "volume_renderer_amd" / "csrc" / "vrm_device.h").write_text("structs v1")
    sha = bench.kernel_source_sha(str(root))
    entry = {"hbm_bytes_per_view": 1000, "kernel_sha": sha, "measured": "2026-11-11", "source": "profiles/r3/x.txt", "secondary": {"valu_issue_frac": 0.9}}
    (root / "profiles" / "traffic.json").write_text(json.dumps({"C3:kernel1:skip1": entry}))
    traffic, secondary, src = bench.traffic_lookup("C3:kernel1:skip1", 32, str(root))
    assert traffic == 32000 and secondary == {"valu_issue_frac": 0.9}
    assert src["match"] is True and src["kernel_sha_built"] == src["kernel_sha_measured"] == sha and src["measured"] == "2026-11-11"
    (root / "volume_renderer_amd" / "csrc" / "vrm_kernels.hip").write_text("kernel v2")      # the kernel changed after the pass
    traffic, secondary, src = bench.traffic_lookup("C3:kernel1:skip1", 32, str(root))
    assert traffic is None and secondary is None and src["match"] is False
    assert src["kernel_sha_built"] != sha and "other kernel sources" in src["reason"]
    traffic, secondary, src = bench.traffic_lookup("C9:kernel1:skip1", 32, str(root))          # no pass for this workload
    assert traffic is None and "no counter pass" in src["reason"]
    # the committed file and the tree: whatever the state, the lookup answers with a reasoned source
    _, _, src = bench.traffic_lookup("C3:kernel1:skip1", 32)
    assert src["kernel_sha_built"] and (src["match"] or src["reason"])
