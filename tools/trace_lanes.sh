#!/bin/bash
# usage: tools/trace_lanes.sh <tag> <splats> [lanes]   (GPU box, repo root) -> gpurun_out/<tag>/timeline.txt : kernel trace of a short bench run, per-queue gaps and concurrency
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; n=$2; lanes=${3:-4}
out=gpurun_out/$tag; mkdir -p $out
export GS4D_LANES=$lanes
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --splats $n --no-cpu-baseline --no-c3 --no-latency --steps 30 --warmup 5 --windows 1 --no-stage-events > $out/bench.json 2> $out/trace.log || { tail -5 $out/trace.log; exit 1; }
python3 tools/timeline.py $out/trace 20 8 > $out/timeline.txt 2>&1
rm -rf $out/trace
tail -8 $out/timeline.txt
