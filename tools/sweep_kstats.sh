#!/bin/bash
# usage: tools/sweep_kstats.sh <tag> [extra gs4d_sweep args]   (on the GPU box, from the repo root)  -> gpurun_out/<tag>/
# Per-kernel durations of the C++ frame-sharded sweep (BASELINE.json configs[3]) with a communicator of one rank: what a frame of the sweep
# costs beyond a bench frame (RGBA8 pack, the gather's copies), kernel by kernel.  --gpus 1 does not fork, so the program runs under rocprofv3 as it is.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- 4dgaussiansplatrendering_amd/host/gs4d_sweep --gpus 1 --sweeps 2 --warmup 1 --no-verify "$@" > $out/sweep.json 2> $out/trace.log || { tail -5 $out/trace.log; exit 1; }
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
rm -rf $out/trace
python3 - $out/kernel_stats.csv $out/sweep.json <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frames = 256 * 3
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    name = r["Name"].split("(")[0][:70]
    print(f'{name:70s} calls {int(r["Calls"]):6d}  avg {float(r["AverageNs"]) / 1e3:9.2f} us  per frame {float(r["TotalDurationNs"]) / 1e3 / frames:8.2f} us')
print(open(sys.argv[2]).read().strip().splitlines()[-1][:600])
PY
