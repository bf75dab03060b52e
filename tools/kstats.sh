#!/bin/bash
# usage (GPU box, repo root): tools/kstats.sh <tag> <splats> <steps> [lanes]
# rocprofv3 --kernel-trace --stats of bench.py: per-kernel average durations (lanes=1: every kernel alone) -> gpurun_out/<tag>_kstats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; n=$2; steps=$3; lanes=${4:-1}
export GS4D_LANES=$lanes
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kp_$tag -- python3 bench.py --splats $n --steps $steps --warmup 5 --no-cpu-baseline --no-stage-events --no-c3 --no-latency --windows 1 > gpurun_out/${tag}_trace.log 2>&1 || { tail -5 gpurun_out/${tag}_trace.log; exit 1; }
cp $(ls gpurun_out/kp_$tag/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_kstats.csv
rm -rf gpurun_out/kp_$tag
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${tag}_kstats.csv")))
frames = $steps + 5
print("== $tag  n=$n lanes=$lanes  (us per launch, launches per frame)")
tot = 0.0
for r in rows:
    name = r["Name"].split("(")[0].replace("void ", "").replace("gs4d::", "")
    calls = int(r["Calls"]); avg = float(r["AverageNs"]) / 1e3
    if calls >= frames // 2:
        print(f"  {name[:46]:46s} {avg:9.1f} us x {calls / frames:5.2f}")
        tot += avg * calls / frames
print(f"  sum per frame {tot:9.1f} us")
PY
grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/${tag}_trace.log | head -1 || true
