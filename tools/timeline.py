#!/usr/bin/env python3
"""Steady-state timeline of the frame lanes' streams from a rocprofv3 kernel trace (csv).

usage: tools/timeline.py <dir with *_kernel_trace.csv> [first_frame] [frames]
Prints, for a window of frames, each kernel's start offset, duration and queue; then per-queue busy time and the fraction of the
window in which 0, 1, 2, 3+ kernels were running.  A "frame" is delimited by successive k_keygen launches.
"""
import csv, glob, sys, collections

d = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 3
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("gs4d::", "").replace("void ", "")
    name = name.split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2] in ("k_keygen", "k_project_count")]      # a frame begins with its key generation: a kernel of its own, or the projection that does it
if len(starts) < first + nfr + 1:
    first = max(0, len(starts) - nfr - 1)
t0, t1 = rows[starts[first]][0], rows[starts[first + nfr]][0]
print(f"window: frames {first}..{first+nfr-1}, {(t1 - t0) / 1e3 / nfr:.1f} us/frame")
win = [r for r in rows if r[1] > t0 and r[0] < t1]
for s, e, n, q in win:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{q:>3}  {n}")
busy = collections.defaultdict(int)
ev = []
for s, e, n, q in win:
    s2, e2 = max(s, t0), min(e, t1)
    busy[q] += e2 - s2
    ev.append((s2, 1)); ev.append((e2, -1))
ev.sort()
conc = collections.defaultdict(int)
cur, last = 0, t0
for t, dlt in ev:
    conc[min(cur, 3)] += t - last
    last = t
    cur += dlt
conc[min(cur, 3)] += t1 - last
print("busy per queue (us/frame):", {q: round(v / 1e3 / nfr, 1) for q, v in busy.items()})
print("concurrency (fraction of window):", {k: round(v / (t1 - t0), 3) for k, v in sorted(conc.items())})
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in win:
    agg[n][0] += 1; agg[n][1] += e - s
print("per kernel (launches/frame, us/frame):", {n: (round(c / nfr, 1), round(t / 1e3 / nfr, 1)) for n, (c, t) in agg.items()})
# idle time between consecutive kernels of one queue (a frame lane is one stream, normally one hardware queue): launch gaps
gaps = collections.defaultdict(list)
lastend = {}
for s, e, n, q in win:
    if q in lastend and s > lastend[q]:
        gaps[q].append((s - lastend[q]) / 1e3)
    lastend[q] = max(e, lastend.get(q, 0))
for q, g in gaps.items():
    g2 = sorted(g)
    print(f"queue {q}: {len(g)} gaps, total {sum(g) / nfr:.1f} us/frame, median {g2[len(g2) // 2]:.1f} us, max {g2[-1]:.1f} us")
