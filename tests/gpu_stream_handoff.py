"""The caller-stream hand-off case of tests/test_gpu_paths.py, as a program of its own: torch must initialise its HIP runtime BEFORE
libgs4d.so is loaded into the process (as bench.py does), which a pytest session that has already rendered frames cannot arrange.

The multi-GPU hand-off without the collective: the caller refills the record buffer through its device pointer on ITS stream
(gs4d_buffer_device_ptr + gs4d_buffer_invalidate), renders, reads the frame back on the device (gs4d_read_pixels_device,
gs4d_read_pixels_rgba8_device) and consumes it on its stream (gs4d_set_stream) — frame after frame, two frame lanes in flight.
Exit code 0 = every frame equals the CPU checker's (float: L-infinity <= 1e-4; RGBA8: <= 1 count)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE)]
torch.cuda.init()
import oracle_lib as oracle       # noqa: E402
import scenes                     # noqa: E402

TOL = 1e-4


def linf(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))))


def _rgba8(img):
    q = np.rint(np.clip(img.astype(np.float64), 0.0, 1.0) * 255.0).astype(np.uint32)
    return q[..., 0] | (q[..., 1] << 8) | (q[..., 2] << 16) | (q[..., 3] << 24)


def _max_count_diff(a, b):
    d = 0
    for s in (0, 8, 16, 24):
        d = max(d, int(np.abs(((a >> s) & 255).astype(np.int64) - ((b >> s) & 255).astype(np.int64)).max()))
    return d


def main():
    side = torch.cuda.Stream()
    gs4d = importlib.import_module("4dgaussiansplatrendering_amd")
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipMemcpyAsync.restype = C.c_int
    n, W, H = 50000, 640, 360
    cam = scenes.CAM_CUBE
    ctx = gs4d.Context(W, H)
    view = gs4d.look_at(cam[0], cam[1])
    proj = gs4d.perspective(scenes.FOV, W, H, scenes.ZNEAR, scenes.ZFAR)
    recs = []
    for seed in (71, 72, 73, 74):
        pos, q, sc, rgba = scenes.cube_params(n, seed=seed)
        recs.append(gs4d.build_records_3d(pos, q, sc * 4.0, rgba))
    ctx.set_stream(side.cuda_stream)
    data = ctx.buffer(nbytes=96 * n)
    dptr, nbytes = ctx.device_ptr(data)
    assert nbytes == 96 * n
    kb, ib = ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.set_mode(gs4d.MODE_4D_SORTED)
    ctx.bind(2, data)
    ctx.set_uniforms(time=0.0, min_opacity=0.0, view=view, proj=proj)
    outs8, outsf, keep = [], [], []
    with torch.cuda.stream(side):
        for rec in recs:
            src = torch.from_numpy(rec).to("cuda")
            keep.append(src)
            ctx.invalidate(data)                                   # the side stream now waits for the frames that still read `data`
            assert hip.hipMemcpyAsync(dptr, src.data_ptr(), 96 * n, 3, C.c_void_p(side.cuda_stream)) == 0
            ctx.clear()
            ctx.keygen(data, 0.0, cam[0], kb, ib, n)
            ctx.sort_pairs(kb, ib, n)
            ctx.bind(1, ib)
            ctx.draw_instanced(n)
            f8 = torch.empty(H * W, dtype=torch.int32, device="cuda")
            ff = torch.empty(H * W * 4, dtype=torch.float32, device="cuda")
            ctx.read_pixels_rgba8_device(f8.data_ptr(), f8.numel() * 4)
            ctx.read_pixels_device(ff.data_ptr(), ff.numel() * 4)
            outs8.append(f8.to("cpu", non_blocking=True))          # consumed on the caller's stream, no host synchronisation in between
            outsf.append(ff.to("cpu", non_blocking=True))
    side.synchronize()
    ctx.finish()
    for rec, f8, ff in zip(recs, outs8, outsf):
        eimg, _, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
        assert linf(ff.numpy().reshape(H, W, 4), eimg) <= TOL
        assert _max_count_diff(f8.numpy().view(np.uint32).reshape(H, W), _rgba8(eimg)) <= 1
    assert linf(outsf[0].numpy(), outsf[1].numpy()) > 0.05           # the frames do differ: stale records would be noticed
    assert ctx.stats()["unordered_draws"] == len(recs)
    ctx.set_stream(None)
    ctx.close()
    print("stream hand-off ok:", len(recs), "frames")


if __name__ == "__main__":
    main()
