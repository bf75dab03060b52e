#!/bin/bash
# usage: tools/kstats.sh <tag> <splats> [lanes]   (on the GPU box, from the repo root; extra knobs through the environment)  -> gpurun_out/<tag>/
# Per-kernel average durations of the bench frame with ONE frame lane by default (every kernel alone): rocprofv3 --kernel-trace --stats of a short
# bench run, reduced to the gs4d kernels.  The quick look between two builds or two knob settings; tools/profile_round.sh is the full evidence set.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; n=$2; lanes=${3:-1}
out=gpurun_out/$tag; mkdir -p $out
export GS4D_LANES=$lanes
steps=30; [ "$n" -ge 5000000 ] && steps=12
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --splats $n --no-cpu-baseline --no-c3 --no-latency --steps $steps --warmup 5 --windows 1 --no-stage-events > $out/bench.json 2> $out/trace.log || { tail -5 $out/trace.log; exit 1; }
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
rm -rf $out/trace
python3 - $out/kernel_stats.csv $out/bench.json <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gs4d" in r["Name"]]
tot = 0.0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    name = r["Name"].split("(")[0][:70]
    print(f'{name:70s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"]) / 1e3:9.2f} us  total {float(r["TotalDurationNs"]) / 1e6:9.3f} ms')
try:
    b = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("ms_per_step", round(b["ms_per_step"], 5), "windows", b["windows_ms_per_step"])
except Exception as e:
    print("no bench line:", e)
PY
