#!/bin/bash
# usage (GPU box, repo root): tools/fetch_calib.sh  ->  gpurun_out/fetch_calib.txt : FETCH_SIZE (KiB) per launch of every calibration kernel next to its known bytes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fc -- tools/fetch_calib > gpurun_out/fetch_calib_run.log 2>&1 || { tail -5 gpurun_out/fetch_calib_run.log; exit 1; }
python3 - <<'PY'
import csv, glob
from collections import defaultdict
f = glob.glob("gpurun_out/fc/*/*counter_collection.csv")[0]
acc, cnt = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        k = r["Kernel_Name"].split("(")[0]; acc[k] += float(r["Counter_Value"]); cnt[k] += 1
ng = 1 << 24
known = {"k_stream16": {"requested": 4 << 30}, "k_gather8": {"requested": ng * 8, "sectors64": ng * 64, "lines128": ng * 128},
         "k_gather48": {"requested": ng * 48, "sectors64": ng * 64, "lines128": ng * 128}, "k_gather64": {"requested": ng * 64, "lines128": ng * 128}}
out = open("gpurun_out/fetch_calib.txt", "w")
for k in sorted(acc):
    per = 1024.0 * acc[k] / cnt[k]
    line = f"{k:12s} FETCH_SIZE {per / 1e6:10.1f} MB/launch ({cnt[k]} launches)  " + "  ".join(f"{name} {b / 1e6:.1f} MB -> counter x {b / per:.3f}" for name, b in known.get(k, {}).items())
    print(line); out.write(line + "\n")
PY
rm -rf gpurun_out/fc
