#!/bin/bash
# One counter pass (instruction mix) over tools/prof_workload.py: tools/prof_sq1.sh <tag> <workload args...>  (under gpurun)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/prof/$TAG
mkdir -p "$OUT"
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU \
    --output-format csv -d "$OUT/sq1" -o sq1 -- python3 tools/prof_workload.py "$@" > "$OUT/sq1.log" 2>&1 || exit 1
python3 tools/prof_summary.py "$OUT" | grep -E "counters|SQ_"
