#!/usr/bin/env python3
"""Copies what tools/final_profiles.sh left under gpurun_out/ into profiles/<round>/ and refreshes profiles/traffic.json
(HBM-side bytes per view = (2*FETCH_SIZE + WRITE_SIZE) KiB / views per launch, from the PMC pass summaries).
usage: python tools/install_profiles.py [r1]"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r1"
O, P = os.path.join(ROOT, "gpurun_out", "final"), os.path.join(ROOT, "profiles", rnd)
os.makedirs(P, exist_ok=True)
for f in ["bench_final_c3.json", "bench_final_c3_under_rocprofv3.json", "bench_final_c3_kernel_stats.csv", "bench_final_c2.json",
          "bench_final_c4.json", "bench_final_c3_tf.json"]:
    shutil.copy(os.path.join(O, f), os.path.join(P, f))
passes = {"final_c3_skip1": ("pmc_final_skip1.txt", 32, "C3:kernel1:skip1"), "final_c3_skip0": ("pmc_final_skip0.txt", 32, "C3:kernel1:skip0"),
          "final_c4_skip1": ("pmc_final_c4_skip1.txt", 16, "C4:kernel1:skip1"), "final_c2_skip1": ("pmc_final_c2_skip1.txt", 32, "C2:kernel1:skip1")}
tj_path = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(tj_path))
for tag, (name, views, key) in passes.items():
    src = os.path.join(ROOT, "gpurun_out", "prof", tag, "summary.txt")
    shutil.copy(src, os.path.join(P, name))
    s = open(src).read()
    blk = s[s.index("== counters: void march_lds_kernel"):]
    f = float(re.search(r"FETCH_SIZE\s+per-dispatch mean = (\S+)", blk).group(1))
    w = float(re.search(r"WRITE_SIZE\s+per-dispatch mean = (\S+)", blk).group(1))
    ms = re.search(r"march_lds_kernel.*?avg_ms=(\S+)", s).group(1)
    t.setdefault(key, {})["hbm_bytes_per_view"] = int((2 * f + w) * 1024 / views)
    t[key]["source"] = f"profiles/{rnd}/{name} ({views} views per launch)"
    print(key, t[key]["hbm_bytes_per_view"], "B/view, kernel", ms, "ms per launch")
json.dump(t, open(tj_path, "w"), indent=1)
for f in ["bench_final_c3.json", "bench_final_c3_under_rocprofv3.json", "bench_final_c2.json", "bench_final_c4.json", "bench_final_c3_tf.json"]:
    d = json.load(open(os.path.join(P, f)))
    print(f, d["value"], "views/s", d["gsamples_per_s"], "Gsamples/s, launch", d["roofline"]["avg_launch_ms"], "ms, frac", d["roofline"]["frac"],
          "N_exec/N_ref", round(d.get("executed_samples_per_step", 0) / d["reference_samples_per_step"], 3))
