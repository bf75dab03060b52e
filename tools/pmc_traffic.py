#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as MI355X_MICROARCH.md §HBM says:
counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so it is doubled (an upper bound for
narrow/uncoalesced reads, which are uncalibrated); WRITE_SIZE is taken as is.
usage: pmc_traffic.py <fetch_dir> <write_dir> <frames> <out.json>"""
import csv, glob, json, sys
from collections import defaultdict

def load(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    return acc, cnt

fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, nw = load(sys.argv[2], "WRITE_SIZE")
frames = int(sys.argv[3])
out = {}
for k in sorted(set(fetch) | set(write)):
    launches = max(nf.get(k, 0), nw.get(k, 0))
    rd = 2.0 * 1024.0 * fetch.get(k, 0.0) / max(1, nf.get(k, 1))
    wr = 1024.0 * write.get(k, 0.0) / max(1, nw.get(k, 1))
    out[k] = {"launches": launches, "launches_per_frame": launches / frames, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    print(f"{k[:60]:60s} launches/frame {launches / frames:5.2f}  read {rd / 1e6:9.2f} MB  write {wr / 1e6:9.2f} MB")
json.dump(out, open(sys.argv[4], "w"), indent=1)
