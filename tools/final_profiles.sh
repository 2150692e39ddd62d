#!/bin/bash
# Refreshes the "final" artefacts of profiles/rN (run on the GPU box through gpurun; results land in gpurun_out/final):
# plain bench lines, the headline bench under rocprofv3 --kernel-trace --stats, and the PMC passes behind traffic.json.
# PASSES=0 tools/final_profiles.sh refreshes the bench lines only (after install_profiles.py has rewritten traffic.json, so
# that the lines carry the secondary ceilings of the same build).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
mkdir -p $O
# the sha of the kernel sources these measurements are taken on (profiles/traffic.json carries it; bench.py compares)
python3 -c "import bench; print(bench.kernel_source_sha())" > $O/kernel_sha.txt || exit 1
timeout -k 10 300 python3 bench.py > $O/bench_final_c3.json 2> $O/bench_c3.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o c3 -- python3 bench.py > $O/bench_final_c3_under_rocprofv3.json 2> $O/bench_c3_prof.err || exit 1
cp $O/trace/*/c3_kernel_stats.csv $O/bench_final_c3_kernel_stats.csv 2>/dev/null || cp $O/trace/c3_kernel_stats.csv $O/bench_final_c3_kernel_stats.csv || exit 1
timeout -k 10 300 python3 bench.py --config C2 > $O/bench_final_c2.json 2> $O/bench_c2.err || exit 1
timeout -k 10 300 python3 bench.py --config C4 > $O/bench_final_c4.json 2> $O/bench_c4.err || exit 1
timeout -k 10 300 python3 bench.py --tf 1 --no-cpu-baseline > $O/bench_final_c3_tf.json 2> $O/bench_c3_tf.err || exit 1
timeout -k 10 300 python3 bench.py --config C1 > $O/bench_final_c1.json 2> $O/bench_c1.err || exit 1
[ "${PASSES:-1}" = 0 ] && { echo done; exit 0; }
bash tools/prof_passes.sh final_c3_skip1 --dim 512 --res 1024 --xdiv 32 --ydiv 9 --skip 1 --views 32 > $O/passes_c3_skip1.log 2>&1 || exit 1
bash tools/prof_passes.sh final_c3_skip0 --dim 512 --res 1024 --xdiv 32 --ydiv 9 --skip 0 --views 32 > $O/passes_c3_skip0.log 2>&1 || exit 1
bash tools/prof_passes.sh final_c4_skip1 --dim 1024 --res 2048 --scalar --xdiv 16 --ydiv 9 --skip 1 --views 16 > $O/passes_c4_skip1.log 2>&1 || exit 1
bash tools/prof_passes.sh final_c2_skip1 --dim 256 --res 512 --scalar --xdiv 8 --ydiv 9 --skip 1 --views 32 > $O/passes_c2_skip1.log 2>&1 || exit 1
bash tools/prof_passes.sh final_c3tf_skip1 --dim 512 --res 1024 --xdiv 32 --ydiv 9 --skip 1 --views 32 --tf > $O/passes_c3tf_skip1.log 2>&1 || exit 1
echo done
