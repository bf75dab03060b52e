#!/bin/bash
# usage: tools/pmc_raw.sh <tag> <splats> <steps> "<COUNTER ...>"   (GPU box, repo root)  -> gpurun_out/<tag>_raw.txt
# Per-kernel averages (per launch, summed over the device) of raw shader-core counters with the kernels running ALONE (one lane), one rocprofv3 --pmc pass per counter.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; n=$2; steps=$3; list="$4"
export GS4D_LANES=1
for m in $list; do
  timeout -k 10 120 rocprofv3 --pmc $m --output-format csv -d gpurun_out/raw_${tag}_$m -- python3 bench.py --splats $n --steps $steps --warmup 3 --no-cpu-baseline --no-stage-events --no-c3 --no-latency --windows 1 > gpurun_out/raw_${tag}_$m.log 2>&1
done
LIST="$list" python3 - <<PY > gpurun_out/${tag}_raw.txt
import csv, glob, collections, os
ms = os.environ["LIST"].split()
out = collections.defaultdict(dict)
for m in ms:
    fs = glob.glob("gpurun_out/raw_${tag}_%s/**/*counter_collection.csv" % m, recursive=True)
    if not fs: continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(fs[0])):
        if r.get("Counter_Name") != m: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for k, (s, c) in acc.items(): out[k][m] = s / c
print("%-44s" % "kernel (alone, GS4D_LANES=1)" + "".join("%22s" % m for m in ms))
for k, d in sorted(out.items()):
    if "gs4d" in k: print("%-44s" % k[:44] + "".join("%22.0f" % d.get(m, float("nan")) for m in ms))
PY
rm -rf gpurun_out/raw_${tag}_*
cat gpurun_out/${tag}_raw.txt
