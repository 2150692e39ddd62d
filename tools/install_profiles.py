#!/usr/bin/env python3
"""Copies what tools/final_profiles.sh left under gpurun_out/ into profiles/<round>/ and refreshes profiles/traffic.json
(HBM-side bytes per view = (2*FETCH_SIZE + WRITE_SIZE) KiB / views per launch, from the PMC pass summaries).
usage: python tools/install_profiles.py [r1]"""
import json
import os
import re
import shutil
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r3"
N_SIMD, N_CU = 1024, 256   # MI355X: 256 CUs x 4 SIMDs
O, P = os.path.join(ROOT, "gpurun_out", "final"), os.path.join(ROOT, "profiles", rnd)
os.makedirs(P, exist_ok=True)
for f in ["bench_final_c3.json", "bench_final_c3_under_rocprofv3.json", "bench_final_c3_kernel_stats.csv", "bench_final_c2.json",
          "bench_final_c4.json", "bench_final_c3_tf.json", "bench_final_c1.json", "c5_slab_rank.json", "c5_slab_rank.log"]:
    if os.path.exists(os.path.join(O, f)):
        shutil.copy(os.path.join(O, f), os.path.join(P, f))
passes = {"final_c3_skip1": ("pmc_final_skip1.txt", 32, "C3:kernel1:skip1"), "final_c3_skip0": ("pmc_final_skip0.txt", 32, "C3:kernel1:skip0"),
          "final_c4_skip1": ("pmc_final_c4_skip1.txt", 16, "C4:kernel1:skip1"), "final_c2_skip1": ("pmc_final_c2_skip1.txt", 32, "C2:kernel1:skip1"),
          "final_c3tf_skip1": ("pmc_final_c3tf_skip1.txt", 32, "C3:kernel1:skip1:tf")}
tj_path = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(tj_path))
for tag, (name, views, key) in passes.items():
    src = os.path.join(ROOT, "gpurun_out", "prof", tag, "summary.txt")
    if not os.path.exists(src):
        continue
    shutil.copy(src, os.path.join(P, name))
    s = open(src).read()
    blk = s[s.index("== counters: void march_lds_kernel"):]
    f = float(re.search(r"FETCH_SIZE\s+per-dispatch mean = (\S+)", blk).group(1))
    w = float(re.search(r"WRITE_SIZE\s+per-dispatch mean = (\S+)", blk).group(1))
    # the production instance of the march kernel (third template argument DIAG = false; the trace pass also runs the
    # diagnostics instance once per view for the executed-sample counters)
    prod = [m for m in re.finditer(r"march_lds_kernel<(\w+), (\w+), (\w+),[^>]*>\s+calls=(\d+) total_ms=(\S+) avg_ms=(\S+)", s) if m.group(3) == "false"]
    ms = max(prod, key=lambda m: float(m.group(5))).group(6)
    t.setdefault(key, {})["hbm_bytes_per_view"] = int((2 * f + w) * 1024 / views)
    t[key]["source"] = f"profiles/{rnd}/{name} ({views} views per launch)"
    # what the pass was measured on: tools/final_profiles.sh writes the sha of the kernel sources of the tree it ran from
    t[key]["kernel_sha"] = open(os.path.join(ROOT, "gpurun_out", "final", "kernel_sha.txt")).read().strip()
    t[key]["measured"] = time.strftime("%Y-%m-%d", time.gmtime(os.path.getmtime(src)))
    print(key, t[key]["hbm_bytes_per_view"], "B/view, kernel", ms, "ms per launch")

    # SURVEY 8d's secondary ceilings (DESIGN.md 4.3 states the formulas); every input is a per-dispatch mean of that file
    def c(name_):
        m = re.search(r"^\s+" + name_ + r"\s+per-dispatch mean = (\S+)", blk, re.M)
        return float(m.group(1)) if m else None
    cyc = c("GRBM_GUI_ACTIVE") / 8.0                       # rocprofv3 sums the 8 XCDs
    sec = {"valu_issue_frac": round(c("SQ_ACTIVE_INST_VALU") * 4.0 / (N_SIMD * cyc), 4),   # quad-cycles of VALU issue / SIMD-cycles
           "lds_busy_frac": round(c("SQ_LDS_IDX_ACTIVE") / (N_CU * cyc), 4),
           "lds_bank_conflict_frac": round(c("SQ_LDS_BANK_CONFLICT") / c("SQ_LDS_IDX_ACTIVE"), 4),
           "l2_hit_rate": round(c("TCC_HIT_sum") / (c("TCC_HIT_sum") + c("TCC_MISS_sum")), 4),
           "mall_hit_rate": None,
           "mall_note": "rocprofv3 on gfx950 lists no Infinity-Cache hit/miss counter (rocprofv3 -L); TCC_EA0_RDREQ_DRAM classifies a "
                        "fabric read by destination, not by where it was served",
           "wait_frac": round(c("SQ_WAIT_INST_ANY") / c("SQ_WAVE_CYCLES"), 4),
           "valu_insts_per_launch": c("SQ_INSTS_VALU"), "salu_insts_per_launch": c("SQ_INSTS_SALU"),
           "kernel_clock_ghz": round(cyc / (float(ms) * 1e-3) / 1e9, 3)}
    if c("TCC_EA0_RDREQ_sum"):
        sec["ea_read_dram_dest_frac"] = round(c("TCC_EA0_RDREQ_DRAM_sum") / c("TCC_EA0_RDREQ_sum"), 4)
    m = re.search(r"COUNTERS views=(\d+) n_ref=(\d+) executed=(\d+) executed_zero_alpha=(\d+) chunks=(\d+) chunks_staged=(\d+) "
                  r"chunks_empty=(\d+) probes_failed=(\d+)", s)
    if m:
        n_exec = int(m.group(3))
        sec["valu_insts_per_wave_sample"] = round(c("SQ_INSTS_VALU") / (n_exec / 64.0), 2)
        sec["executed_over_reference_samples"] = round(n_exec / int(m.group(2)), 4)
        sec["executed_zero_alpha_frac"] = round(int(m.group(4)) / n_exec, 4)
        sec["chunk_prologues_per_view"] = int(m.group(5)) // int(m.group(1))
        sec["chunks_staged_per_view"] = int(m.group(6)) // int(m.group(1))
        mr = re.search(r"box_retries=(\d+)", s)
        if mr:
            sec["box_retries_per_view"] = int(mr.group(1)) // int(m.group(1))
    t[key]["secondary"] = sec
    print("   secondary:", {k: v for k, v in sec.items() if k != "mall_note"})
json.dump(t, open(tj_path, "w"), indent=1)
for f in ["bench_final_c3.json", "bench_final_c3_under_rocprofv3.json", "bench_final_c2.json", "bench_final_c4.json", "bench_final_c3_tf.json"]:
    if not os.path.exists(os.path.join(P, f)):
        continue
    d = json.load(open(os.path.join(P, f)))
    print(f, d["value"], "views/s", d["gsamples_per_s"], "Gsamples/s, launch", d["roofline"]["avg_launch_ms"], "ms, frac", d["roofline"]["frac"],
          "N_exec/N_ref", round(d.get("executed_samples_per_step", 0) / d["reference_samples_per_step"], 3))
