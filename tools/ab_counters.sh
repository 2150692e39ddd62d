#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for name in intree nogeo; do
  lib=volume_renderer_amd/variants/libvrm_$name.so
  [ "$name" = intree ] && lib=volume_renderer_amd/libvrm.so
  for wl in "--dim 512 --res 1024 --xdiv 16 --ydiv 17 --views 32 --tf" "--dim 512 --res 1024 --xdiv 16 --ydiv 17 --views 32" "--dim 1024 --res 2048 --scalar --xdiv 16 --ydiv 9 --views 16"; do
    echo "$name $wl"
    VRM_LIB=$lib timeout -k 10 170 python3 tools/prof_workload.py $wl --skip 1 --reps 2 --count-exec 2>&1 | tail -n 2 || exit 1
  done
done
