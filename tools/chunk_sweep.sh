#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { echo -n "$1 chunk $2: "; timeout -k 10 170 python3 tools/prof_workload.py $3 --skip 1 --all --reps 3 --chunk $2 2>&1 | tail -1 | sed -e 's/; ms\/view per launch.*//' -e 's/kernel 1 skip . chunk [0-9]* packed 1: //'; }
for c in 8 10 12 14 16 20; do run c3 $c "--dim 512 --res 1024 --xdiv 16 --ydiv 17 --views 32"; done
for c in 8 10 12 15 20 28; do run c4 $c "--dim 1024 --res 2048 --scalar --xdiv 16 --ydiv 9 --views 16"; done
for c in 12 16 20 27 36; do run c3tf $c "--dim 512 --res 1024 --xdiv 16 --ydiv 17 --views 32 --tf"; done
for c in 8 12 15 20 30; do run c2 $c "--dim 256 --res 512 --scalar --xdiv 8 --ydiv 9 --views 32"; done
