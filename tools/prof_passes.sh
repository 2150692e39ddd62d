#!/bin/bash
# rocprofv3 passes over tools/prof_workload.py; one pass per counter group (rocprofv3 does not split).
# usage: tools/prof_passes.sh <tag> <workload args...>     (run under gpurun; output in gpurun_out/prof/<tag>)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/prof/$TAG
mkdir -p "$OUT"
run() { # name, rocprof args...
    local name=$1; shift
    timeout -s KILL 170 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 tools/prof_workload.py "${WL[@]}" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "pass $name rc=$rc"
    return $rc
}
WL=("$@")
WL=("$@" --count-exec)
run trace --kernel-trace --stats || exit 1
WL=("$@")
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU || exit 1
run sq2 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU || exit 1
run mem1 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE || exit 1
run mem2 --pmc FETCH_SIZE || exit 1
run mem3 --pmc WRITE_SIZE || exit 1
# (no Infinity-Cache hit/miss counter exists on gfx950; these classify fabric reads by destination, for the record)
run mem4 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum || exit 1
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
