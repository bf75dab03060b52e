#!/bin/bash
# usage: tools/sort_prof.sh <tag> [env...]   -> prints per-kernel averages of the sort micro-benchmark under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
env "$@" true > /dev/null
( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp_$tag -- python3 tools/sort_bench.py ${SORT_N:-1000000} ${SORT_REPS:-40} depth > /dev/null 2>&1 )
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/sp_$tag/*/*kernel_stats.csv")[0]
print("== $tag")
for r in csv.DictReader(open(f)):
    if "k_os" in r["Name"]: print("  ", r["Name"][:34], r["Calls"], round(float(r["AverageNs"])/1e3,2))
PY
