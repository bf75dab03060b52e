#!/usr/bin/env python3
"""Rewrites the generated kernel tables — between the `tables:begin` / `tables:end` markers of profiles/README.md and DESIGN.md — from the committed
evidence files of the current round (TAG): per kernel the launches per frame, the algorithmic bytes per launch, the HBM bytes the
counters saw, the launch's duration alone (one frame lane) and beside the other lanes' kernels, and the fraction of the 8 TB/s roofline the kernel
reaches alone — on algorithmic bytes and on moved bytes.  One source for every per-kernel number quoted in the documents."""
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
TAG = "r04"
PEAK = 8000.0        # GB/s
W, H = 1920, 1080
CFG = {"c2": ("C2 — 10⁶ static 3D splats, 1080p (BASELINE.json configs[1])", 1_000_000, 65, 30),
       "c4": ("C4 on one GPU — 10⁶ true 4D splats at t = 25, 1080p (configs[3]'s set)", 1_000_000, 65, 30),
       "c3": ("C3 — 10⁷ static 3D splats, 1080p (configs[2])", 10_000_000, 29, 12)}


def stats(name):
    out = {}
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(open(path)):
        out[r["Name"].split("(")[0].replace("void ", "").replace("gs4d::", "")] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    return out


def line(name):
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return None
    return json.loads(open(path).read().strip().splitlines()[-1])


def algorithmic(kernel, n, launches_per_frame):
    """bytes per LAUNCH that SURVEY.md §8(d) credits to a kernel of the path (0: overhead of this design, not credited)"""
    if kernel.startswith("k_os_pass"):
        return 68.0 * n / max(launches_per_frame, 1e-9) if launches_per_frame <= 4.5 else 0.0      # the depth sort's passes share 68 B/splat (the tile sort's are overhead: the C3 row mixes both)
    if kernel.startswith("k_project_count") or kernel.startswith("k_preprocess"):
        return (96 + 48 + 8) * n                                                                    # record read, projected record written, key + index (the fused draw makes them)
    if kernel.startswith("k_composite"):
        return 48.0 * n + 16.0 * W * H
    if kernel.startswith("k_keygen"):
        return 8.0 * n
    return 0.0


def table(cfg):
    title, n, frames, min_calls = CFG[cfg]
    a, o = stats(f"{TAG}_kernel_stats_alone_{cfg}.csv"), stats(f"{TAG}_kernel_stats_{cfg}.csv")
    pmf = os.path.join(P, f"{TAG}_pmc_traffic_{cfg}.json")
    pm = json.load(open(pmf)) if os.path.exists(pmf) else {}
    if not a:
        return f"({cfg}: no evidence files)\n", 0.0, 0.0
    t = f"**{title}**\n\n| kernel | launches / frame | algorithmic MB / launch | counter MB / launch | µs alone | frac of 8 TB/s alone (algorithmic) | (moved) | µs beside the other lanes |\n|---|---|---|---|---|---|---|---|\n"
    rows = sorted((k for k in a if a[k][1] >= min_calls and not k.startswith(("k_lane_probe", "k_lds_order", "k_soa_repack", "k_fill", "k_pack"))), key=lambda k: -a[k][0] * a[k][1])
    tot_alone = 0.0
    for k in rows:
        lpf = a[k][1] / frames
        alg = algorithmic(k, n, lpf)
        hit = [v for kk, v in pm.items() if kk.replace("gs4d::", "").startswith(k.split("<")[0]) and abs(v["launches_per_frame"] - lpf) < 0.6]
        moved = hit[0]["hbm_bytes_per_launch"] if hit else None
        us = a[k][0]
        tot_alone += us * lpf
        fa = f"{alg / 1e3 / us / PEAK:.3f}" if alg else "—"
        fm = f"{moved / 1e3 / us / PEAK:.3f}" if moved else "—"
        t += f"| `{k}` | {lpf:.1f} | {alg / 1e6:.1f} | {moved / 1e6:.1f} | {us:.1f} | {fa} | {fm} | {o.get(k, (float('nan'), 0))[0]:.1f} |\n" if moved else \
             f"| `{k}` | {lpf:.1f} | {alg / 1e6:.1f} | — | {us:.1f} | {fa} | — | {o.get(k, (float('nan'), 0))[0]:.1f} |\n"
    traffic = sum(v["hbm_bytes_per_launch"] * v["launches_per_frame"] for v in pm.values() if v["launches_per_frame"] >= 0.5) / 1e6
    alg_frame = (268.0 * n + 16.0 * W * H) / 1e6
    b = line(f"{TAG}_bench_{cfg}.json")
    t += f"\nSum of kernel time per frame alone {tot_alone / 1e3:.3f} ms"
    if b:
        t += f"; the profiled command's own line: {b['ms_per_step']:.4f} ms/frame pipelined = {alg_frame / b['ms_per_step'] / 1e3 / PEAK * 1e3:.3f} of 8 TB/s on the {alg_frame:.0f} MB a frame moves algorithmically"
    if traffic:
        t += f"; counters: {traffic:.0f} MB per frame = {traffic / alg_frame:.2f} × algorithmic"
    return t + ".\n", tot_alone, traffic


def main():
    d = line(f"{TAG}_bench_default_driver_settings.json")
    txt = ""
    if d:
        txt += (f"Driver-style line (`{TAG}_bench_default_driver_settings.json`, `python bench.py --steps 20 --warmup 5`): C2 {d['ms_per_step']:.4f} ms/frame "
                f"(windows {d['windows_ms_per_step']}), one lane {d['latency_ms_one_lane']:.4f}, one key / index pair {d['one_key_index_pair']['ms_per_step']:.4f}; "
                f"C3 {d['c3']['ms_per_step']:.3f} ms/frame, one lane {d['c3']['latency_ms_one_lane']:.3f}; "
                f"C4 on one GPU {d['c4_n1']['ms_per_step']:.4f} ms/frame, one lane {d['c4_n1']['latency_ms_one_lane']:.4f}.\n\n")
    for cfg in ("c2", "c4", "c3"):
        t, _, _ = table(cfg)
        txt += t + "\n"
    for name in (os.path.join(P, "README.md"), os.path.join(ROOT, "DESIGN.md")):
        s = open(name).read()
        if "<!-- tables:begin -->" in s:
            s = re.sub(r"<!-- tables:begin -->.*<!-- tables:end -->", lambda m: "<!-- tables:begin -->\n" + txt + "<!-- tables:end -->", s, flags=re.S)
            open(name, "w").write(s)
    print(txt)


if __name__ == "__main__":
    main()
