#!/usr/bin/env python3
"""Experiment: frames round-robin over L independent contexts (whole frames overlap instead of stages).
usage: [GS4D_STREAMS=1] tools/lanes_exp.py <splats> <steps> <lanes>"""
import sys, importlib, os, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), 'tests')]
import numpy as np, scenes
g = importlib.import_module('4dgaussiansplatrendering_amd')
n = int(sys.argv[1]); steps = int(sys.argv[2]); L = int(sys.argv[3])
W, H = 1920, 1080
pos, q, scale, rgba = scenes.cube_params(n); rec = g.build_records_3d(pos, q, scale, rgba)
view = g.look_at(*scenes.CAM_CUBE); proj = g.perspective(60.0, W, H, 0.1, 5000.0)
lanes = []
for l in range(L):
    ctx = g.Context(W, H); data = ctx.buffer(rec)
    kb = [(ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)) for _ in range(2)]
    ctx.set_clear_color(g.CLEAR_COLOR); ctx.set_mode(g.MODE_4D_SORTED); ctx.bind(2, data); ctx.set_uniforms(time=0.0, min_opacity=0.0, view=view, proj=proj)
    lanes.append((ctx, data, kb))
def frame(k):
    ctx, data, kb = lanes[k % L]
    keys, idx = kb[(k // L) & 1]
    ctx.clear(); ctx.keygen(data, 0.0, scenes.CAM_CUBE[0], keys, idx, n); ctx.sort_pairs(keys, idx, n); ctx.bind(1, idx); ctx.draw_instanced(n)
for k in range(20): frame(k)
for c, _, _ in lanes: c.finish()
t0 = time.perf_counter()
for k in range(steps): frame(k)
t1 = time.perf_counter()
for c, _, _ in lanes: c.finish()
t2 = time.perf_counter()
print(f"n {n} lanes {L} streams {os.environ.get('GS4D_STREAMS', '3')}: host {1e6 * (t1 - t0) / steps:.1f} us/frame, device {1e6 * (t2 - t0) / steps:.1f} us/frame")
