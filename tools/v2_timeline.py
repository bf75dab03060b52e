#!/usr/bin/env python3
"""Reads the per-tile stamps a TUNING build of k_composite_v2 dumps (GS4D_V2_STAMP_FILE; csrc/composite2.hip) and says where the kernel's time goes:
how long tiles of each list length take, how many waves are resident over time, when the long lists start.  Stamps are 100 MHz ticks."""
import sys
import numpy as np

a = np.loadtxt(sys.argv[1], dtype=np.uint64)
wg = np.arange(len(a))
real = a[:, 2] > 0                       # workgroups that composited a tile (stamped at their end)
t0 = a[real, 0].min()
s, m, e, E, tile, xcc = [(a[real, k].astype(np.int64) - (t0 if k < 3 else 0)) for k in range(6)]
us = lambda x: x / 100.0
print(f"workgroups {len(a)}, with entries {real.sum()}, entries {E.sum()}, kernel span (first start .. last end) {us(e.max()):.1f} us")
print("list length   tiles   mean us  (ordering the list)   p90 us   max us   mean start us")
for lo, hi in ((1, 64), (65, 128), (129, 256), (257, 384), (385, 512), (513, 768), (769, 4096)):
    k = (E >= lo) & (E <= hi)
    if k.any():
        d = us(e[k] - s[k]); so = us(m[k] - s[k])
        print(f"{lo:4d}-{hi:<5d} {k.sum():7d} {d.mean():9.2f} {so.mean():12.2f} {np.percentile(d, 90):14.2f} {d.max():8.2f} {us(s[k]).mean():10.1f}")
ticks = np.arange(0, e.max() + 1, 100)   # every microsecond
res = [(int(((s <= t) & (e > t)).sum())) for t in ticks]
print("resident tile-waves per microsecond:", res)
late = np.argsort(e)[-10:]
print("last to finish: (start us, end us, entries, xcc)", [(round(us(s[i]), 1), round(us(e[i]), 1), int(E[i]), int(xcc[i])) for i in late])
starts_all = a[:, 0].astype(np.int64); starts_all = starts_all[starts_all > 0]
print(f"all workgroups (also the empty ones) start within {us(starts_all.max() - starts_all.min()):.1f} us" if len(starts_all) else "")
