#!/bin/bash
# The two SQ counter passes + GRBM cycles of tools/prof_passes.sh only (quick look at what binds a kernel).
# usage: tools/prof_sq.sh <tag> <workload args...>   (under gpurun; VRM_LIB selects the build)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/prof/$TAG
mkdir -p "$OUT"
run() { local name=$1; shift; timeout -s KILL 170 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 tools/prof_workload.py "${WL[@]}" > "$OUT/$name.log" 2>&1; echo "pass $name rc=$?"; }
WL=("$@" --count-exec)
run trace --kernel-trace --stats
WL=("$@")
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU
run sq2 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
run mem1 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
grep -E "workload|SQ_|GRBM|march_lds" "$OUT/summary.txt" | grep -v "^== counters"
