#!/usr/bin/env python3
"""Stream overlap as the device saw it, without a profiler: start/end of every stage from the library's own event ring.

usage: tools/ev_timeline.py <splats> [first_frame] [frames] [mode]     mode: full | draw
(each timed stage costs two event records, so frames run ~25 us slower than untimed ones; the shape of the overlap is what to read)
"""
import sys, importlib, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), 'tests')]
import numpy as np, scenes
g = importlib.import_module('4dgaussiansplatrendering_amd')
n = int(sys.argv[1]); first = int(sys.argv[2]) if len(sys.argv) > 2 else 40; nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mode = sys.argv[4] if len(sys.argv) > 4 else 'full'
W, H = 1920, 1080
pos, q, scale, rgba = scenes.cube_params(n); rec = g.build_records_3d(pos, q, scale, rgba)
ctx = g.Context(W, H); data = ctx.buffer(rec)
kb = [(ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)) for _ in range(2)]
view = g.look_at(*scenes.CAM_CUBE); proj = g.perspective(60.0, W, H, 0.1, 5000.0)
ctx.set_clear_color(g.CLEAR_COLOR); ctx.set_mode(g.MODE_4D_SORTED); ctx.bind(2, data); ctx.set_uniforms(time=0.0, min_opacity=0.0, view=view, proj=proj)
def frame(k):
    keys, idx = kb[k & 1] if mode == 'full' or k < 2 else kb[0]
    ctx.clear()
    if mode == 'full' or k < 2:
        ctx.keygen(data, 0.0, scenes.CAM_CUBE[0], keys, idx, n); ctx.sort_pairs(keys, idx, n)
    ctx.bind(1, idx); ctx.draw_instanced(n)
for k in range(10): frame(k)
ctx.finish(); ctx.set_profiling(True)
for k in range(10, 10 + first + nfr + 2): frame(k)
tl = ctx.timeline() * 1e3
names = g.STAGES
t0 = np.min(tl[first][tl[first][:, 0] >= 0][:, 0])
rows = []
for f in range(first, first + nfr):
    for s, nm in enumerate(names):
        if tl[f, s, 0] >= 0: rows.append((tl[f, s, 0] - t0, tl[f, s, 1] - tl[f, s, 0], f, nm))
rows.sort()
for st, du, f, nm in rows: print(f"{st:9.1f} {du:7.1f}  frame {f}  {nm}")
per = (np.min(tl[first + nfr][tl[first + nfr][:, 0] >= 0][:, 0]) - t0) / nfr
print(f"period {per:.1f} us/frame (with stage events)")
