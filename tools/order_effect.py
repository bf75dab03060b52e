#!/usr/bin/env python3
"""usage: python tools/order_effect.py <mode> [count = 1] [splats = 1000000]   (GPU box)
Does what a process did BEFORE a context was created change how fast that context renders?
  mode ctx1 / ctx4 : create and close <count> dummy contexts with 1 / 4 frame lanes
  mode stream      : create and destroy <count> HIP streams (nothing else)
  mode keepstream  : create <count> HIP streams and keep them
  mode alloc       : hipMalloc + hipFree of <count> x 33 MB (nothing else)
  mode none        : nothing
then measures the bench frame (7 windows of 20 frames)."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, scenes
gs4d = importlib.import_module("4dgaussiansplatrendering_amd")
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
hip = C.CDLL("libamdhip64.so")
if mode in ("ctx1", "ctx4"):
    for _ in range(k):
        os.environ["GS4D_LANES"] = mode[-1]
        gs4d.Context(bench.W, bench.H).close()
    os.environ.pop("GS4D_LANES", None)
elif mode in ("stream", "keepstream"):
    hip.hipSetDevice(0)
    for _ in range(k):
        s = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
        if mode == "stream":
            assert hip.hipStreamDestroy(s) == 0
elif mode == "alloc":
    hip.hipSetDevice(0)
    for _ in range(k):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(33 << 20)) == 0
        assert hip.hipFree(p) == 0
res, _, _ = bench.measure_single(gs4d, scenes, n, 20, 5, 7, 0, stage_events=False)
print(f"{mode} x {k}: {res['ms_per_step']:.5f} ms/frame, windows {res['windows_ms_per_step']}, candidate streams rejected at context creation: {res['stats']['lane_streams_rejected']}")
