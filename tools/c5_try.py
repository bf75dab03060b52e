#!/usr/bin/env python3
"""BASELINE.json configs[4] at full size on one GPU: 10^7 4D splats of the NonLinearMotion generator (2745 time steps, four laps),
4K frame, t = 1372 — GPU frame vs the CPU checker.  usage: tools/c5_try.py [nrecords]"""
import sys, importlib, os, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), 'tests')]
import numpy as np, scenes, oracle_lib as oracle
g = importlib.import_module('4dgaussiansplatrendering_amd')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
W, H = 3840, 2160
tea = oracle.golden("teapot_vdata")
steps = 2745
t0 = time.time(); rec = g.scene_nonlinear(tea, steps=steps, angle_multiplier=360.0 / steps * 4.0, max_records=n); print('records', rec.shape, round(time.time() - t0, 1), 's')
cam = scenes.CAM_NONLINEAR; t = 1372.0
view = g.look_at(cam[0], cam[1]); proj = g.perspective(scenes.FOV, W, H, scenes.ZNEAR, scenes.ZFAR)
ctx = g.Context(W, H); data = ctx.buffer(rec); keys = ctx.buffer(nbytes=4 * n); idx = ctx.buffer(nbytes=4 * n)
ctx.set_clear_color(g.CLEAR_COLOR); ctx.set_mode(g.MODE_4D_SORTED); ctx.bind(2, data); ctx.set_uniforms(time=t, min_opacity=0.0, view=view, proj=proj)
for it in range(3):
    t0 = time.time()
    ctx.clear(); ctx.keygen(data, t, cam[0], keys, idx, n); ctx.sort_pairs(keys, idx, n); ctx.bind(1, idx); ctx.draw_instanced(n)
    img = ctx.read_pixels()
    print('gpu frame', round(1e3 * (time.time() - t0), 1), 'ms', ctx.stats())
t0 = time.time(); eimg, _, ms = oracle.render_4d(rec, True, t, 0.0, cam[0], view, proj, W, H, nthreads=min(os.cpu_count(), 64)); print('oracle', round(time.time() - t0, 1), 's', ms)
print('Linf', float(np.max(np.abs(img - eimg))), 'max dev from clear', float(np.abs(eimg - np.array(g.CLEAR_COLOR, np.float32)).max()))
perm = ctx.read(idx, np.uint32, n); eidx, ekeys = oracle.keygen(rec, t, cam[0]); _, eperm = oracle.sort_pairs(ekeys.view(np.uint32), eidx, "lsd")
print('permutation bit-exact', bool(np.array_equal(perm, eperm)))
