#!/bin/bash
# usage: tools/lane_sweep3.sh <tag>  -> gpurun_out/<tag>.txt : hardware queues x lanes at C2, alternating runs (tuning aid)
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1.txt; : > $out
run() { echo "== $*" >> $out; env "$@" python3 bench.py --no-cpu-baseline --no-c3 --no-latency --steps 20 --warmup 10 --windows 9 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['ms_per_step'],5), b['windows_ms_per_step'])" >> $out; }
for rep in 1 2 3; do
run GS4D_LANES=4
run GS4D_LANES=4 GPU_MAX_HW_QUEUES=8
run GS4D_LANES=5 GPU_MAX_HW_QUEUES=8
run GS4D_LANES=6 GPU_MAX_HW_QUEUES=8
done
run GS4D_LANES=4 GPU_MAX_HW_QUEUES=2
run GS4D_LANES=4 GPU_MAX_HW_QUEUES=16
cat $out
