#!/bin/bash
# On the GPU box: HBM-side traffic and L2 hit rate of one launch of the C4 (or C3) workload for each named library build
# (volume_renderer_amd/variants/libvrm_<name>.so; "intree" = the in-tree build).  usage: tools/ab_traffic.sh c4|c3 name...
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WL=$1; shift
case $WL in
  c4) args="--dim 1024 --res 2048 --scalar --xdiv 16 --ydiv 9 --skip 1 --views 16";;
  c3) args="--dim 512 --res 1024 --xdiv 32 --ydiv 9 --skip 1 --views 32";;
esac
for name in "$@"; do
  lib=volume_renderer_amd/variants/libvrm_$name.so
  [ "$name" = intree ] && lib=volume_renderer_amd/libvrm.so
  O=gpurun_out/prof/traffic_${WL}_$name
  mkdir -p $O
  for pass in "fetch FETCH_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
    set -- $pass; pn=$1; shift
    VRM_LIB=$lib timeout -s KILL 170 rocprofv3 --pmc "$@" --output-format csv -d $O/$pn -o $pn -- python3 tools/prof_workload.py $args > $O/$pn.log 2>&1 || { echo "pass $pn failed for $name"; exit 1; }
  done
  echo "== $name ($WL): $(grep 'ms/view' $O/fetch.log | tail -1)"
  python3 tools/prof_summary.py $O | grep "FETCH_SIZE\|TCC_HIT\|TCC_MISS\|TCP_TCC" | grep -v "^==" | head -8
done
