#!/usr/bin/env python3
"""Micro-benchmark of gs4d_sort_pairs (tuning aid): python tools/sort_bench.py [n] [reps] [bits]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gs4d = importlib.import_module("4dgaussiansplatrendering_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
kind = sys.argv[3] if len(sys.argv) > 3 else "depth"
rng = np.random.default_rng(0)
if kind == "depth":
    keys = (1.0 / rng.uniform(333.0, 1025.0, n)).astype(np.float32).view(np.uint32)
else:
    keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
ctx = gs4d.Context(64, 64)
kb0 = ctx.buffer(keys); vb0 = ctx.buffer(np.arange(n, dtype=np.uint32))
kb = ctx.buffer(nbytes=4 * n); vb = ctx.buffer(nbytes=4 * n)
import ctypes as C
lib = C.CDLL(None)
hip = C.CDLL("libamdhip64.so")
pk0, _ = ctx.device_ptr(kb0); pv0, _ = ctx.device_ptr(vb0); pk, _ = ctx.device_ptr(kb); pv, _ = ctx.device_ptr(vb)
def reset():
    hip.hipMemcpy(C.c_void_p(pk), C.c_void_p(pk0), C.c_size_t(4 * n), 3)
    hip.hipMemcpy(C.c_void_p(pv), C.c_void_p(pv0), C.c_size_t(4 * n), 3)
    hip.hipDeviceSynchronize()
for _ in range(5):
    reset(); ctx.sort_pairs(kb, vb, n); ctx.finish()      # the lanes' streams do not synchronise with the null stream reset() copies on
# sorted input is a different (friendlier) key distribution per pass, so re-randomise between timed sorts but time only the sorts
ts = []
for _ in range(reps):
    reset(); ctx.finish()
    t0 = time.perf_counter(); ctx.sort_pairs(kb, vb, n); ctx.finish(); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
print(f"n={n} kind={kind} sort wall us: median {np.median(ts):.1f} min {ts.min():.1f}")
out = ctx.read(kb, np.uint32, n)
assert np.all(out[1:] >= out[:-1])
