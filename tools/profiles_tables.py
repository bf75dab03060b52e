#!/usr/bin/env python3
"""Rewrites the kernel tables of profiles/README.md (between the `tables:` markers) from the committed evidence files of the current round (TAG)."""
import csv, json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
TAG = "r03"


def stats(name):
    out = {}
    for r in csv.DictReader(open(os.path.join(P, name))):
        out[r["Name"].split("(")[0].replace("void ", "").replace("gs4d::", "")] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    return out


def line(name):
    return json.loads(open(os.path.join(P, name)).read().strip().splitlines()[-1])


def table(cfg, frames, min_calls):
    a, o = stats(f"{TAG}_kernel_stats_alone_{cfg}.csv"), stats(f"{TAG}_kernel_stats_{cfg}.csv")
    t = f"| kernel, {cfg.upper()}, µs per launch (launches per frame) | alone (1 lane) | overlapped (4 lanes) |\n|---|---|---|\n"
    rows = [k for k in a if a[k][1] >= min_calls]
    for k in rows:
        t += f"| `{k}` ({a[k][1] / frames:.1f}) | {a[k][0]:.1f} | {o.get(k, (float('nan'), 0))[0]:.1f} |\n"
    t += f"| sum of kernel time per frame, alone (ms) | {sum(a[k][0] * a[k][1] / frames for k in rows) / 1e3:.3f} | |\n"
    return t


def traffic(cfg):
    pm = json.load(open(os.path.join(P, f"{TAG}_pmc_traffic_{cfg}.json")))
    return sum(v["hbm_bytes_per_launch"] * v["launches_per_frame"] for v in pm.values() if v["launches_per_frame"] >= 0.5) / 1e6


d, b2, b3 = line(f"{TAG}_bench_default_driver_settings.json"), line(f"{TAG}_bench_c2.json"), line(f"{TAG}_bench_c3.json")
txt = (f"Driver-style line (`{TAG}_bench_default_driver_settings.json`): C2 {d['ms_per_step']:.4f} ms/frame (windows {d['windows_ms_per_step']}), one lane "
       f"{d['latency_ms_one_lane']:.4f}; C3 {d['c3']['ms_per_step']:.3f} ms/frame, one lane {d['c3']['latency_ms_one_lane']:.3f}. The profiled commands' own lines: "
       f"C2 {b2['ms_per_step']:.4f}, C3 {b3['ms_per_step']:.3f} ms/frame.\n\n" + table("c2", 65, 30) + "\n" + table("c3", 29, 12) +
       f"\nHBM bytes per frame from the counters (kernels launched at least every other frame): C2 {traffic('c2'):.0f} MB = {traffic('c2') / 301.18:.2f} × the algorithmic "
       f"301 MB (round 2: 418 MB, 1.39 ×; round 1: 623 MB, 2.07 ×); C3 {traffic('c3') / 1e3:.2f} GB = {traffic('c3') / 2713.18:.2f} × 2.71 GB (round 2: 4.30 GB, 1.58 ×; round 1: 6.48 GB, 2.39 ×).\n")
path = os.path.join(P, "README.md")
s = open(path).read()
s = re.sub(r"<!-- tables:begin -->.*<!-- tables:end -->", "<!-- tables:begin -->\n" + txt + "<!-- tables:end -->", s, flags=re.S)
open(path, "w").write(s)
print(txt)
