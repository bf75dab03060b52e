#!/bin/bash
# usage: tools/pmc_sq.sh <tag> <splats> <steps>   (GPU box, repo root)  -> gpurun_out/<tag>_sq.txt
# Per-kernel averages of a few derived shader-core metrics with the kernels running ALONE (one lane), one rocprofv3 --pmc pass each.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; n=$2; steps=$3
export GS4D_LANES=1
for m in VALUBusy VALUUtilization LDSBankConflict MemUnitBusy OccupancyPercent SALUBusy; do
  rocprofv3 --pmc $m --output-format csv -d gpurun_out/sq_${tag}_$m -- python3 bench.py --splats $n --steps $steps --warmup 3 --no-cpu-baseline --no-stage-events --no-c3 --no-latency --windows 1 > gpurun_out/sq_${tag}_$m.log 2>&1
done
python3 - <<PY > gpurun_out/${tag}_sq.txt
import csv, glob, collections
out = collections.defaultdict(dict)
for m in "VALUBusy VALUUtilization LDSBankConflict MemUnitBusy OccupancyPercent SALUBusy".split():
    fs = glob.glob("gpurun_out/sq_${tag}_%s/**/*counter_collection.csv" % m, recursive=True)
    if not fs: continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(fs[0])):
        if r.get("Counter_Name") != m: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for k, (s, c) in acc.items(): out[k][m] = s / c
ms = "VALUBusy VALUUtilization LDSBankConflict MemUnitBusy OccupancyPercent SALUBusy".split()
print("%-40s" % "kernel (alone, GS4D_LANES=1)" + "".join("%18s" % m for m in ms))
for k, d in sorted(out.items()):
    if "gs4d" in k: print("%-40s" % k[:40] + "".join("%18.2f" % d.get(m, float("nan")) for m in ms))
PY
rm -rf gpurun_out/sq_${tag}_*
cat gpurun_out/${tag}_sq.txt
