#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1.txt; : > $out
run() { echo "== $*" >> $out; env "$@" python3 bench.py --no-cpu-baseline --no-c3 --no-latency --steps 20 --warmup 10 --windows 9 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['ms_per_step'],5), b['windows_ms_per_step'])" >> $out; }
run3() { echo "== C3 $*" >> $out; env "$@" python3 bench.py --splats 10000000 --no-cpu-baseline --no-c3 --no-latency --steps 10 --warmup 5 --windows 5 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['ms_per_step'],5), b['windows_ms_per_step'])" >> $out; }
run GS4D_SORT_SHAPE=2
run GS4D_SORT_SHAPE=3
run GS4D_SORT_SHAPE=5
run GS4D_SORT_SHAPE=2
run GS4D_SORT_SHAPE=3
run3 GS4D_SORT_SHAPE=5
run3 GS4D_SORT_SHAPE=3
cat $out
