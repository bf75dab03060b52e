#!/bin/bash
# usage: tools/lane_sweep.sh <tag>   (GPU box, repo root) -> gpurun_out/<tag>.txt : C2 throughput for a few lane counts and sort tile shapes (tuning aid)
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1.txt; : > $out
run() { echo "== $*" >> $out; env "$@" python3 bench.py --no-cpu-baseline --no-c3 --no-latency --steps 20 --warmup 10 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['ms_per_step'],5), b['windows_ms_per_step'])" >> $out; }
run GS4D_LANES=4
run GS4D_LANES=3
run GS4D_LANES=5
run GS4D_LANES=6
run GS4D_LANES=4 GS4D_SORT_SHAPE=3
run GS4D_LANES=4 GS4D_SORT_SHAPE=1
run GS4D_LANES=4 GS4D_SORT_SHAPE=4
run GS4D_LANES=4 GPU_MAX_HW_QUEUES=8
run GS4D_LANES=8 GPU_MAX_HW_QUEUES=8
cat $out
