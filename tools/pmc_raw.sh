#!/bin/bash
# usage: tools/pmc_raw.sh <tag> <splats> <kernel-substring> <counter> [counter...]   (GPU box, repo root; one rocprofv3 --pmc pass per counter, kernels alone)
# prints the per-launch average of each raw counter for the kernels whose name contains the substring
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; n=$2; pat=$3; shift 3
export GS4D_LANES=1
for m in "$@"; do
  rocprofv3 --pmc $m --output-format csv -d gpurun_out/raw_${tag}_$m -- python3 bench.py --splats $n --steps 10 --warmup 3 --no-cpu-baseline --no-stage-events --no-c3 --no-latency --windows 1 > gpurun_out/raw_${tag}_$m.log 2>&1 || { echo "$m: failed"; tail -2 gpurun_out/raw_${tag}_$m.log; continue; }
  python3 - <<PY
import csv, glob, collections
fs = glob.glob("gpurun_out/raw_${tag}_$m/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(fs[0])) if fs else []:
    if r.get("Counter_Name") == "$m" and "$pat" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gs4d::", "")
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k, (s, c) in acc.items():
    if c >= 5: print("%-28s %-34s %16.0f per launch (%d launches)" % ("$m", k[:34], s / c, c))
PY
  rm -rf gpurun_out/raw_${tag}_$m
done
