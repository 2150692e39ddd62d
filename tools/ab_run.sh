#!/bin/bash
# On the GPU box (through gpurun): time the whole C3 / C4 / C2 exports with each library build named on the command line
# (volume_renderer_amd/variants/libvrm_<name>.so; "intree" = the in-tree build).  usage: tools/ab_run.sh "c3 c4 c2" name...
set -u
cd "$GRAFT_REPO_ROOT"
WLS=$1; shift
for name in "$@"; do
  lib=volume_renderer_amd/variants/libvrm_$name.so
  [ "$name" = intree ] && lib=volume_renderer_amd/libvrm.so
  for wl in $WLS; do
    case $wl in
      c3) args="--dim 512 --res 1024 --xdiv 16 --ydiv 17 --views 32";;
      c3tf) args="--dim 512 --res 1024 --xdiv 16 --ydiv 17 --views 32 --tf";;
      c3ns) args="--dim 512 --res 1024 --xdiv 16 --ydiv 17 --views 32 --skip 0";;
      c4) args="--dim 1024 --res 2048 --scalar --xdiv 16 --ydiv 9 --views 16";;
      c2) args="--dim 256 --res 512 --scalar --xdiv 8 --ydiv 9 --views 32";;
    esac
    skip="--skip 1"; [ "$wl" = c3ns ] && skip=""
    echo -n "$name $wl: "
    VRM_LIB=$lib timeout -k 10 170 python3 tools/prof_workload.py $args $skip ${EXTRA:-} --all --reps 3 2>&1 | tail -1 | sed -e 's/; ms\/view per launch.*//' -e 's/kernel 1 skip . chunk 0 packed 1: //' || exit 1
  done
done
