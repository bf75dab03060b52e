#!/bin/bash
# usage: tools/kprof.sh <tag> <kernel-substring> <cmd...> : per-kernel average duration under rocprofv3 (kernels run alone when the command syncs per frame)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; pat=$2; shift; shift
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kp_$tag -- "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/kp_$tag/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "$pat" in r["Name"]: print("$tag".ljust(12), r["Name"][:40].ljust(40), r["Calls"].rjust(4), round(float(r["AverageNs"])/1e3,1))
PY
rm -rf gpurun_out/kp_$tag
