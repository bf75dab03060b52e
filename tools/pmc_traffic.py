#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters in KiB).

Correction of FETCH_SIZE on gfx950, calibrated with known-bytes micro-kernels in this library's access patterns (tools/fetch_calib.hip,
profiles/r02_fetch_calib.txt): the counter adds 64 B per memory-side read request.  A wide coalesced read (16 B per lane, 128-B
requests) is therefore reported at HALF its bytes (x2, as MI355X_MICROARCH.md says); a random gather of 8, 48 or 64 bytes inside one
64-B sector is reported at exactly one sector, 64 B (x1) — round 1 doubled those too and over-stated the gather kernels' traffic.
Kernels are classed by what dominates their reads: the compositing kernels and the round-1 binning kernel gather (x1); everything
else streams (x2).  WRITE_SIZE is taken as is.
usage: pmc_traffic.py <fetch_dir> <write_dir> <frames> <out.json>"""
import csv, glob, json, sys
from collections import defaultdict

GATHER = ("k_composite", "k_bin_emit", "k_bucket_tiles_staged")          # read mostly by 64-B-sector gathers (8-byte loads, one run per lane): FETCH_SIZE x 1


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
    factor = 1.0 if any(g in k for g in GATHER) else 2.0
    raw = 1024.0 * fetch.get(k, 0.0) / max(1, nf.get(k, 1))
    rd = factor * raw
    wr = 1024.0 * write.get(k, 0.0) / max(1, nw.get(k, 1))
    out[k] = {"launches": launches, "launches_per_frame": launches / frames, "fetch_size_bytes_per_launch": raw, "fetch_correction": factor,
              "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    print(f"{k[:60]:60s} launches/frame {launches / frames:5.2f}  read {rd / 1e6:9.2f} MB (FETCH_SIZE x{factor:.0f})  write {wr / 1e6:9.2f} MB")
json.dump(out, open(sys.argv[4], "w"), indent=1)
