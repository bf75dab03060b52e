#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X splat rasteriser (BASELINE.json metric: Gaussians/s + ms/frame @1080p).

A "step" is one frame of the hot path over splat records already resident in HBM:
    Clear -> key generation -> radix sort -> Draw (preprocess, tile binning, tile sort, composite)
exactly the call sequence of the reference's Scene::Render with sorting on (Scenes.h:312-339), through the C ABI.

  N = 1   workload = BASELINE.json configs[1]: 1,000,000 random 3D splats in a 400^3 cube, one 1080p frame per step.
  N > 1   independent frames shard over ranks (one process per GPU): every rank renders its own frame
          of the time sweep per step (4D splats, configs[3] shape) and the finished frames are gathered on rank 0 with
          one RCCL gather per step in the presentation format (RGBA8).  Weak scaling: per-GPU work is fixed.

Prints ONE JSON line on rank 0.  `value` = splats processed by all ranks / wall time of the timed region.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 1920, 1080
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6.3 TB/s


def algorithmic_bytes(n, w, h):
    """per frame 268 B/splat + 16 B/pixel, split by stage (tile-list traffic is overhead, not credited)."""
    return {
        "keygen": 8 * n,                    # key + index written
        "sort": 68 * n,                     # ideal 4-pass 8-bit LSD pair sort
        "preprocess": (96 + 48) * n,        # record read + projected record written
        "binning": 0, "pairsort": 0,
        "composite": 48 * n + 16 * w * h,   # projected record read in blend order + RGBA32F written once
        "frame": 268 * n + 16 * w * h,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--splats", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    multi = world > 1

    import torch
    import scenes
    gs4d = importlib.import_module("4dgaussiansplatrendering_amd")     # raises if libgs4d.so is missing: no fallback

    torch.cuda.set_device(local_rank)
    dist = None
    if multi:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n = args.splats
    cam = scenes.CAM_CUBE
    view = gs4d.look_at(cam[0], cam[1])
    proj = gs4d.perspective(scenes.FOV, W, H, scenes.ZNEAR, scenes.ZFAR)
    if multi:
        pos4, q, scale, life, fade, vel, rgba = scenes.cube_params_4d(n)
        rec = gs4d.build_records_4d(pos4, q, scale, life, fade, vel, rgba)
    else:
        pos, q, scale, rgba = scenes.cube_params(n)
        rec = gs4d.build_records_3d(pos, q, scale, rgba)

    ctx = gs4d.Context(W, H, device=local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)          # HIP stream shared with torch so that the gather orders after the draw
    data = ctx.buffer(rec)
    keys, idx = ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.set_mode(gs4d.MODE_4D_SORTED)
    ctx.bind(1, idx)
    ctx.bind(2, data)
    ctx.set_uniforms(time=0.0, min_opacity=0.0, view=view, proj=proj)

    frame8 = gathered = None
    if multi:
        frame8 = torch.empty(H * W, dtype=torch.int32, device="cuda")
        gathered = [torch.empty_like(frame8) for _ in range(world)] if rank == 0 else None

    total_frames = (args.warmup + args.steps) * world

    def step(k):
        # frame k of this rank in the time sweep (t_k = 50 * frame / (frames - 1)); static 3D records ignore t
        t = 50.0 * (k * world + rank) / max(1, total_frames - 1) if multi else 0.0
        ctx.clear()
        ctx.set_uniforms(time=t)
        ctx.keygen(data, t, cam[0], keys, idx, n)
        ctx.sort_pairs(keys, idx, n)
        ctx.draw_instanced(n)
        if multi:
            ctx.read_pixels_rgba8_device(frame8.data_ptr(), frame8.numel() * 4)
            dist.gather(frame8, gathered, dst=0)

    def fence():
        ctx.finish()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    fence()
    ctx.set_profiling(True)                     # HIP events on the launch stream, averaged over the timed frames
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    fence()
    elapsed = time.perf_counter() - t0
    stage_ms = ctx.timings()
    ctx.set_profiling(False)
    stats = ctx.stats()

    if multi:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    ms_per_step = 1e3 * elapsed / args.steps
    value = n * args.steps * world / elapsed

    if rank == 0:
        alg = algorithmic_bytes(n, W, H)
        timed = {k: v for k, v in stage_ms.items() if v > 0}
        dom = max(timed, key=timed.get) if timed else None
        # dominant kernel stage, priced with its share of the algorithmic bytes; whole frame beside it
        roofline = None
        if dom:
            credited = max((k for k in timed if alg[k] > 0), key=lambda k: timed[k])
            ach = alg[credited] / (timed[credited] * 1e-3) / 1e9
            frame_ach = alg["frame"] / (ms_per_step * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": credited, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                        "traffic": None, "kernel_ms": round(timed[credited], 5), "algorithmic_bytes_per_launch": alg[credited],
                        "slowest_stage": dom,
                        "frame": {"achieved": round(frame_ach, 2), "frac": round(frame_ach / HBM_PEAK_GBS, 5), "algorithmic_bytes": alg["frame"]},
                        "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()}}
        cpu = None
        if not args.no_cpu_baseline and not multi:
            cpu = cpu_baseline(rec, cam, view, proj)
        out = {
            "metric": "Gaussians/sec + ms/frame @1080p; sort permutation bit-exact",
            "value": value, "unit": "splats/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("1,000,000 random 3D splats in a 400^3 cube, single 1080p frame (BASELINE.json configs[1])" if not multi else
                                    "1,000,000 4D splats, time sweep, one 1080p frame per rank per step, RGBA8 frames gathered on rank 0 (BASELINE.json configs[3] shape)"),
                       "splats": n, "width": W, "height": H, "sort": "on", "frames_per_step": world,
                       "tile_list_entries": stats["entries"], "overflow_reruns": stats["reruns"]},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)

    ctx.close()
    if multi:
        dist.destroy_process_group()


def cpu_baseline(rec, cam, view, proj):
    """The CPU restatement (oracle/, kind 'port') timed on this host: the same frame, all stages, once single-threaded
    (the reference's only CPU stage, the key loop, is single-threaded) and once on all host cores."""
    import oracle_lib
    n = rec.shape[0]
    cores = os.cpu_count() or 1
    cores = min(cores, 64)
    t0 = time.perf_counter()
    _, _, ms1 = oracle_lib.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H, nthreads=1)
    t1 = time.perf_counter()
    _, _, msn = oracle_lib.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H, nthreads=cores)
    t2 = time.perf_counter()
    return {"value": n / (t2 - t1), "unit": "splats/s", "cores": cores, "kind": "port",
            "sample": f"one full frame of the same workload ({n} splats, 1080p), keygen+sort+preprocess+composite; all-core run {t2 - t1:.2f} s, single-thread run {t1 - t0:.2f} s",
            "single_thread_value": n / (t1 - t0),
            "stage_ms_single_thread": dict(zip(("keygen", "sort", "preprocess", "composite"), (round(x, 2) for x in ms1))),
            "stage_ms_all_cores": dict(zip(("keygen", "sort", "preprocess", "composite"), (round(x, 2) for x in msn)))}


if __name__ == "__main__":
    main()
