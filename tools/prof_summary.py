"""Summarise the CSVs of tools/prof_passes.sh: per-kernel time and per-kernel, per-dispatch counter means."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:90]


def main():
    root = sys.argv[1]
    tl = os.path.join(root, "trace.log")
    if os.path.exists(tl):
        for line in open(tl):
            if line.startswith("COUNTERS") or line.startswith("kernel "):
                print("== workload: " + line.strip())
    for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
        print(f"== kernel stats ({os.path.relpath(f, root)})")
        for row in csv.DictReader(open(f)):
            print(f"  {short(row.get('Name', '?')):90s} calls={row.get('Calls')} total_ms={float(row.get('TotalDurationNs', 0)) / 1e6:.3f} "
                  f"avg_ms={float(row.get('AverageNs', 0)) / 1e6:.4f} pct={row.get('Percentage')}")
    agg = defaultdict(lambda: defaultdict(float))
    ndisp = defaultdict(set)
    meta = {}
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            ndisp[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
            meta[k] = (row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"),
                       row.get("Workgroup_Size"))
    for k in sorted(agg):
        if "march" not in k and "quantize" not in k:
            continue
        print(f"== counters: {k}  (vgpr, agpr, sgpr, lds, wg) = {meta[k]}")
        for c in sorted(agg[k]):
            n = max(1, len(ndisp[(k, c)]))
            print(f"  {c:34s} per-dispatch mean = {agg[k][c] / n:.6g}   (dispatches {n})")


if __name__ == "__main__":
    main()
