#!/bin/bash
# On the GPU box: the packed march (C3 volume, transfer function on) at several image sizes, tile queues (kernel 3) against one
# workgroup per tile (kernel 4): where the launcher's rule for the packed march should switch.  usage: tools/ab_res.sh "1024 1280 1536"
cd "$GRAFT_REPO_ROOT"
for res in $1; do
  for k in 3 4; do
    echo -n "res $res kernel $k: "
    timeout -k 10 170 python3 tools/prof_workload.py --dim 512 --res $res --xdiv 16 --ydiv 17 --views 16 --tf --skip 1 --kernel $k --all --reps 2 2>&1 | tail -n 1 | sed -e 's/; ms\/view per launch.*//' -e 's/.*views/views/' || exit 1
  done
done
