#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X splat rasteriser (BASELINE.json metric: Gaussians/s + ms/frame @1080p).

A "step" is one pass of the hot path over splat records already resident in HBM:
    Clear -> key generation -> radix sort -> Draw (projection, tile lists, ordered compositing)
exactly the call sequence of the reference's Scene::Render with sorting on (Scenes.h:312-339), through the C ABI.

  N = 1   workload = BASELINE.json configs[1]: 1,000,000 random 3D splats in a 400^3 cube, one 1080p frame per step.  The same
          invocation also times configs[2] (10^7 splats, the HBM-scale config) into the "c3" block of the line, the 10^6 TRUE 4D splats of
          configs[3] at mid-sweep on this one GPU into "c4_n1" (96-byte records, moving keys), and the frame time with a single frame lane
          (nothing overlaps: "latency_ms_one_lane").
  N > 1   workload = BASELINE.json configs[3], exactly: 1,000,000 4D splats, the 256-frame time sweep t_k = 50 k / 255, frame k on rank
          k mod N — or, with --rank0-frames-pct below 100, fewer frames on rank 0, which also receives everybody's (one process per GPU, no
          data-path collective); the finished frames travel to rank 0 in the presentation format
          (RGBA8) with one RCCL gather per --gather-every frames of every rank, from two batch buffers used alternately (a pack never waits
          for the gather that is in flight).  A step is one whole sweep (256 frames), so the total work per step is fixed as N grows:
          "scaling": "strong".  The line carries, per rank, how long the comm stream was busy per sweep and how long it ran on after the
          rank's last frame (so that a scaling curve explains itself).

Timing: W untimed warm-up steps, then windows of EXACTLY K steps, each bracketed by barrier + synchronize on both sides, maximum over
ranks per window; `ms_per_step` / `value` are the MEDIAN window (all windows are listed: a 20-step window lasts 3 ms at N = 1, short
enough for a clock ramp or a host hiccup to move it by tens of percent; --windows 1 gives the single-window contract reading).
Prints ONE JSON line on rank 0.  `value` = splats processed by all ranks / wall time.
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
SWEEP_FRAMES = 256          # BASELINE.json configs[3]
PROFILE_TAG = "r04"         # profiles/<tag>_pmc_traffic_c{2,3}.json hold the per-launch HBM traffic of this build's kernels

# stage -> (kernel the stage is made of, launches per frame); "sort" and "pairsort" launch counts come from the library's stats
KERNELS = {"keygen": "gs4d::k_keygen", "sort": "gs4d::k_os_pass", "preprocess": "gs4d::k_project_count", "binning": "gs4d::k_bucket_scatter",
           "pairsort": "gs4d::k_os_pass", "composite": "gs4d::k_composite_v2"}
KERNELS_ORDERED = dict(KERNELS, binning="gs4d::k_bin_emit", composite="gs4d::k_composite<")       # the instance-ordered path (long tile lists: configs[2])


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


class Scene:
    """One context with its resident buffers and the per-frame call sequence."""

    def __init__(self, gs4d, records, cam, view, proj, device, keybufs=4, lanes=None):
        old = os.environ.get("GS4D_LANES")
        if lanes is not None:
            os.environ["GS4D_LANES"] = str(lanes)
        try:
            self.ctx = gs4d.Context(W, H, device=device)
        finally:
            if lanes is not None:
                if old is None:
                    os.environ.pop("GS4D_LANES", None)
                else:
                    os.environ["GS4D_LANES"] = old
        self.gs4d, self.n, self.cam = gs4d, records.shape[0], cam
        ctx = self.ctx
        self.data = ctx.buffer(records)
        # per-frame key / sort-index buffers are double-buffered by the application (as any renderer does with per-frame resources):
        # frame f+1 can generate and sort its keys while frame f is still in flight
        self.keybufs = [(ctx.buffer(nbytes=4 * self.n), ctx.buffer(nbytes=4 * self.n)) for _ in range(keybufs)]
        ctx.set_clear_color(gs4d.CLEAR_COLOR)
        ctx.set_mode(gs4d.MODE_4D_SORTED)
        ctx.bind(2, self.data)
        ctx.set_uniforms(time=0.0, min_opacity=0.0, view=view, proj=proj)
        self.k = 0

    def frame(self, t=0.0):
        ctx = self.ctx
        keys, idx = self.keybufs[self.k % len(self.keybufs)]
        self.k += 1
        ctx.clear()
        ctx.set_uniforms(time=t)
        ctx.keygen(self.data, t, self.cam[0], keys, idx, self.n)
        ctx.sort_pairs(keys, idx, self.n)
        ctx.bind(1, idx)
        ctx.draw_instanced(self.n)

    def close(self):
        self.ctx.close()


def timed_windows(step, fence, steps, warmup, windows, reduce_max=None):
    """warm-up, then `windows` windows of exactly `steps` steps, each between two fences.  Returns the per-window seconds."""
    for k in range(warmup):
        step(k)
    fence()
    out = []
    k = warmup
    for _ in range(windows):
        t0 = time.perf_counter()
        for _ in range(steps):
            step(k)
            k += 1
        fence()
        el = time.perf_counter() - t0
        out.append(reduce_max(el) if reduce_max else el)
    return out


def roofline_block(gs4d_stats, stage_ms, warm_ms, n, ms_per_step, traffic_file, one=None):
    """The dominant credited kernel, priced per launch: achieved = algorithmic bytes per launch / average launch duration from HIP
    events on the stream the kernel is launched on (gs4d_set_profiling / gs4d_get_timings).

    one = the one-lane run of the same workload in this invocation (every stage timed, nothing overlapping): the block then names the kernel that
    takes the largest share of the frame's kernel time ALONE and prices it by that duration — `achieved` / `frac` say how well THAT KERNEL uses the
    memory system.  With four frame lanes in flight the same launch runs beside the other lanes' kernels and lasts 1.5-4x as long; that reading goes
    into `overlapped` (its frac is a statement about sharing the device, not about the kernel), and `frame` is the throughput evidence: all
    algorithmic bytes of a frame over the pipelined time per frame.  `moved`: the same kernel priced on the HBM bytes it actually moves (rocprofv3
    --pmc passes of this command, profiles/<tag>_pmc_traffic_*.json; tools/pmc_traffic.py) — below the algorithmic figure wherever a compact
    record shadow or a fused producer saves traffic, above it where sectors are fetched for less than they hold."""
    alg = algorithmic_bytes(n, W, H)
    ordered = gs4d_stats["tile_sort_passes"] > 0
    names = KERNELS_ORDERED if ordered else KERNELS
    launches_of = lambda st: {"sort": gs4d_stats["depth_sort_passes"], "pairsort": max(1, gs4d_stats["tile_sort_passes"])}.get(st, 1)
    over = {k: v for k, v in stage_ms.items() if v > 0 and alg[k] > 0}
    warm = {k: v for k, v in warm_ms.items() if v > 0}
    alone = {k: v for k, v in (one["steady_ms"] if one and one.get("steady_ms") else {}).items() if v > 0}
    alone_cred = {k: v for k, v in alone.items() if alg[k] > 0}
    if alone_cred:
        credited, basis = max(alone_cred, key=alone_cred.get), "alone"
        ms = alone_cred[credited]
    elif over:
        credited, basis = max(over, key=over.get), "overlapped"
        ms = over[credited]
    else:
        return None
    launches = launches_of(credited)
    kname = names[credited]
    ach = alg[credited] / (ms * 1e-3) / 1e9
    frame_ach = alg["frame"] / (ms_per_step * 1e-3) / 1e9
    traffic, tsrc = None, None
    if os.path.isfile(traffic_file):
        pm = json.load(open(traffic_file))
        hit = [v for k, v in pm.items() if k.startswith(kname)]
        if hit:
            traffic, tsrc = hit[0]["hbm_bytes_per_launch"], os.path.relpath(traffic_file, ROOT)
    out = {"bound": "hbm", "kernel": kname.rstrip("<"), "stage": credited, "launches_per_frame": launches, "basis": basis,
           "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
           "traffic": traffic, "traffic_source": tsrc,
           "kernel_ms": round(ms / launches, 5), "algorithmic_bytes_per_launch": alg[credited] // launches,
           "frame": {"achieved": round(frame_ach, 2), "frac": round(frame_ach / HBM_PEAK_GBS, 5), "algorithmic_bytes": alg["frame"], "ms_per_frame_pipelined": round(ms_per_step, 5)},
           "stage_ms_warmup_all_stages_timed": {k: round(v, 5) for k, v in warm_ms.items()}}
    if traffic:
        mv = traffic * launches / (ms * 1e-3) / 1e9
        out["moved"] = {"hbm_bytes_per_launch": traffic, "achieved": round(mv, 2), "frac": round(mv / HBM_PEAK_GBS, 5), "moved_over_algorithmic": round(traffic * launches / alg[credited], 3)}
    if basis == "alone":
        tot = sum(alone.values())
        out["share_of_frame_kernel_time_alone"] = round(ms / tot, 4) if tot > 0 else None
        out["stage_ms_one_lane"] = {k: round(v, 5) for k, v in (one["steady_ms"] or {}).items()}
        out["note"] = "kernel, duration and frac: the launch alone (one frame lane, HIP events around every stage, same invocation); `overlapped`: the same launch beside the other lanes' kernels in the pipelined timed windows; `frame`: all algorithmic bytes of a frame over the pipelined time per frame"
        o_ms = over.get(credited, warm.get(credited))
        if o_ms:
            o_ach = alg[credited] / (o_ms * 1e-3) / 1e9
            out["overlapped"] = {"kernel_ms": round(o_ms / launches, 5), "achieved": round(o_ach, 2), "frac": round(o_ach / HBM_PEAK_GBS, 5),
                                 "source": "timed windows" if credited in over else "warm-up frames", "slowest_stage": max(warm, key=warm.get) if warm else None}
    else:
        out["note"] = "no one-lane run in this invocation: duration and frac are those of a launch that runs beside the other frame lanes' kernels"
        out["slowest_stage"] = max(warm, key=warm.get) if warm else None
    return out


def morton_order(pos):
    """--spatial-order (an experiment, never the headline): the permutation that puts splats which are neighbours in space next to each other in
    the record buffer (30-bit Morton code of the position).  BASELINE.json's splats are in random order; what the order of the records is worth
    says where the gathers of the compositing kernels spend their time (profiles/r04_experiments.txt, item 13)."""
    g = np.clip((pos - pos.min(0)) / (pos.max(0) - pos.min(0)) * 1023.0, 0, 1023).astype(np.uint64)

    def spread(v):
        v = (v | (v << 16)) & np.uint64(0x030000FF)
        v = (v | (v << 8)) & np.uint64(0x0300F00F)
        v = (v | (v << 4)) & np.uint64(0x030C30C3)
        v = (v | (v << 2)) & np.uint64(0x09249249)
        return v
    return np.argsort(spread(g[:, 0]) | (spread(g[:, 1]) << np.uint64(1)) | (spread(g[:, 2]) << np.uint64(2)), kind="stable")


def measure_single(gs4d, scenes, n, steps, warmup, windows, device, stage_events=True, lanes=None, keybufs=4, steady_stages=0, four_d=False, t=0.0, settle_window=False, spatial_order=False):
    """One GPU: static 3D splats in the cube (configs[1] / configs[2]) or, four_d, the true 4D splats of configs[3] at time t.
    Returns (result dict, records, camera)."""
    cam = scenes.CAM_CUBE
    view = gs4d.look_at(cam[0], cam[1])
    proj = gs4d.perspective(scenes.FOV, W, H, scenes.ZNEAR, scenes.ZFAR)
    if four_d:
        pos4, q, scale, life, fade, vel, rgba = scenes.cube_params_4d(n)
        rec = gs4d.build_records_4d(pos4, q, scale, life, fade, vel, rgba)
        del pos4, q, scale, life, fade, vel, rgba
    else:
        pos, q, scale, rgba = scenes.cube_params(n)
        if spatial_order:
            o = morton_order(pos)
            pos, q, scale, rgba = pos[o], q[o], scale[o], rgba[o]
        rec = gs4d.build_records_3d(pos, q, scale, rgba)
        del pos, q, scale, rgba
    sc = Scene(gs4d, rec, cam, view, proj, device, lanes=lanes, keybufs=keybufs)
    ctx = sc.ctx
    frame = (lambda: sc.frame(t))

    def fence():
        ctx.finish()

    # warm-up with every stage timed (HIP events on the launch stream) to find the slowest credited one; in the timed windows only that
    # stage keeps its events, and only in every 8th frame (an event record costs ~2 us of back-to-back dispatch)
    # (the first frames of a context are not typical: the library learns the tile-list capacities of the scene in them — draws that are
    # aborted on the device and re-run — so the events start after the third warm-up frame where there are that many)
    learn = min(3, max(0, warmup - 1))
    for k in range(warmup):
        if k == learn:
            fence()
            ctx.set_profiling(True)
        frame()
    fence()
    warm_ms = ctx.timings()
    alg = algorithmic_bytes(n, W, H)
    cred = [k for k, v in warm_ms.items() if v > 0 and alg[k] > 0]
    dominant = max(cred, key=lambda k: warm_ms[k]) if cred else None
    ctx.set_profiling([dominant] if (dominant and stage_events) else False, every=8)
    # Frames that aborted on the device (tile-list capacity, list length) and were cleared away before anything observed them are never
    # re-run: a timed window that contains one has timed an incomplete render.  The library counts them; the bench refuses such windows —
    # once more from the start (the context has learned from the aborted frames by then), and if they still contain one, not at all.
    attempts = 0
    for attempt in range(2):
        attempts += 1
        aborted_before = ctx.stats()["aborted_discarded"]
        if settle_window:
            # a context that follows a much larger one in the same process (the side blocks) spends its first hundred frames beside the runtime still
            # giving that context's gigabytes back: one untimed window of the same length first (measured: 0.58-0.74 ms/frame in it against 0.125 after)
            timed_windows(lambda k: frame(), fence, steps, 0, 1)
        secs = timed_windows(lambda k: frame(), fence, steps, 0, windows)
        aborted = ctx.stats()["aborted_discarded"] - aborted_before
        if not aborted:
            break
        print(f"bench.py: {aborted} timed frame(s) aborted on the device and were never completed; timing the windows again", file=sys.stderr)
    if aborted:
        raise RuntimeError(f"{aborted} timed frame(s) aborted on the device and were never completed: the measurement is invalid")
    stage_ms = ctx.timings()
    ctx.set_profiling(False)
    steady_ms = None
    if steady_stages:
        # every stage timed over frames of the steady state (the warm-up frames above include the draws the library aborted and re-ran
        # while it learned the list capacities of this scene: their stage averages are not a kernel's duration)
        ctx.set_profiling(True)
        for k in range(steady_stages):
            frame()
        fence()
        steady_ms = ctx.timings()
        ctx.set_profiling(False)
    stats = ctx.stats()
    sc.close()
    ms = sorted(1e3 * s / steps for s in secs)
    med = ms[len(ms) // 2]
    res = {"ms_per_step": med, "value": n / (med * 1e-3), "windows_ms_per_step": [round(1e3 * s / steps, 5) for s in secs], "aborted_in_timed_windows": aborted, "timed_window_attempts": attempts,
           "stats": stats, "warm_ms": warm_ms, "stage_ms": stage_ms, "steady_ms": steady_ms}
    return res, rec, (cam, view, proj)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps each; the line reports the median window")
    ap.add_argument("--splats", type=int, default=1_000_000)
    ap.add_argument("--gather-every", type=int, default=8, help="N>1: frames of every rank per RCCL gather")
    ap.add_argument("--rank0-frames-pct", type=int, default=None, help="N>1: rank 0 renders this percentage of an equal share of the sweep (it also receives every other rank's frames); default: sharding.default_rank0_pct(N)")
    ap.add_argument("--lanes", type=int, default=None, help="frame lanes of the context (default: the library's, 4); experiments")
    ap.add_argument("--keybufs", type=int, default=None, help="key / sort-index buffer pairs the application cycles through (default: one per lane)")
    ap.add_argument("--four-d", action="store_true", help="N=1: the main workload is configs[3]'s set of true 4D splats at t = 25 instead of the static cube (profiling the c4_n1 block's kernels: tools/profile_round.sh)")
    ap.add_argument("--spatial-order", action="store_true", help="N=1, an experiment: the static splats are uploaded in Morton order of their positions instead of BASELINE.json's random order (the line says so in config.workload)")
    ap.add_argument("--no-c3", action="store_true", help="N=1: skip the configs[2] and c4_n1 blocks (10^7 splats; 10^6 4D splats)")
    ap.add_argument("--no-latency", action="store_true", help="N=1: skip the one-lane frame time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-events", action="store_true", help="do not record per-stage HIP events in the timed windows")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Libraries talk there too (RCCL prints a version banner at communicator creation on some boxes): keep
    # the real stdout for the line and point file descriptor 1 at stderr for everything else.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    multi = world > 1 or os.environ.get("GS4D_BENCH_FORCE_SWEEP") == "1"      # rehearsal on a one-GPU box: the N > 1 program (RCCL gathers, events, two batch buffers) with a communicator of ONE rank

    import torch
    import scenes
    gs4d = importlib.import_module("4dgaussiansplatrendering_amd")     # raises if libgs4d.so is missing: no fallback

    backend = os.environ.get("GS4D_BENCH_BACKEND", "nccl")      # "gloo" only to rehearse the N>1 code path on a one-GPU box
    if backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)

    if not multi:
        out = single_gpu(args, gs4d, scenes, local_rank)
    else:
        out = multi_gpu(args, gs4d, scenes, torch, rank, local_rank, world, backend)
    sys.stdout.flush()
    if rank == 0:
        os.write(line_fd, (json.dumps(out) + "\n").encode())
    os.close(line_fd)


def single_gpu(args, gs4d, scenes, device):
    n = args.splats
    kb = args.keybufs or args.lanes or 4          # one pair per frame lane (the library default is 4 lanes)
    fd = dict(four_d=True, t=25.0) if args.four_d else dict(spatial_order=True) if args.spatial_order else {}
    res, rec, (cam, view, proj) = measure_single(gs4d, scenes, n, args.steps, args.warmup, args.windows, device, stage_events=not args.no_stage_events, lanes=args.lanes, keybufs=kb, **fd)
    tag = "c4" if args.four_d else "c2" if n == 1_000_000 else "c3" if n == 10_000_000 else "x"
    tfile = lambda t: os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_pmc_traffic_{t}.json")
    # The four-lane side measurements come first, the one-lane ones last: a context created after a context with another number of frame
    # lanes runs ~10 % slower (tools/order_effect.py: 0.110 -> 0.122 ms/frame after one one-lane context has been created and closed; HIP maps
    # the lanes' streams onto its hardware queues differently then) — an artefact of this process's history, not of the workload.
    n3 = 10_000_000
    side = not args.no_c3 and n == 1_000_000 and not args.four_d
    r3 = r4 = None
    if side:
        r3, _, _ = measure_single(gs4d, scenes, n3, max(10, min(args.steps, 100) // 2), 8, 3, device, stage_events=not args.no_stage_events, lanes=args.lanes, keybufs=kb, spatial_order=args.spatial_order)
        # configs[3]'s workload on ONE GPU: 10^6 true 4D splats (96-byte records: nothing of sig is constant or symmetric-by-construction here), t mid-sweep
        r4, _, _ = measure_single(gs4d, scenes, n, min(args.steps, 100), 24, 5, device, stage_events=not args.no_stage_events, lanes=args.lanes, keybufs=kb, four_d=True, t=25.0, settle_window=True)
    one_pair = None
    if not args.no_latency:
        # the reference's own buffer layout: ONE key / index pair for every frame (Scenes.h m_key_buf / m_values_buf).  Frame f + 1 writes the
        # buffers frame f's sort is still filling: the lanes order themselves on the device (events), consecutive frames overlap less.
        r1, _, _ = measure_single(gs4d, scenes, n, min(args.steps, 50), min(args.warmup, 10), 3, device, stage_events=False, lanes=args.lanes, keybufs=1, **fd)
        one_pair = {"ms_per_step": round(r1["ms_per_step"], 5), "value": r1["value"], "unit": "splats/s",
                    "note": "same workload with one key / sort-index buffer pair instead of one per frame lane (the reference's layout, Scenes.h:241-247): the drop-in figure"}
    one = one3 = one4 = None
    if not args.no_latency:
        one, _, _ = measure_single(gs4d, scenes, n, min(args.steps, 50), min(args.warmup, 10), 3, device, stage_events=False, lanes=1, steady_stages=16, **fd)
        if side:
            one3, _, _ = measure_single(gs4d, scenes, n3, 10, 5, 3, device, stage_events=False, lanes=1, steady_stages=8, spatial_order=args.spatial_order)
            one4, _, _ = measure_single(gs4d, scenes, n, 30, 10, 3, device, stage_events=False, lanes=1, steady_stages=16, four_d=True, t=25.0)
    roofline = roofline_block(res["stats"], res["stage_ms"], res["warm_ms"], n, res["ms_per_step"], tfile(tag), one)

    def side_block(r, o, workload, nn, ptag):
        st_ = r["stats"]
        return {"workload": workload, "splats": nn, "ms_per_step": r["ms_per_step"], "value": r["value"], "unit": "splats/s", "windows_ms_per_step": r["windows_ms_per_step"],
                "tile_list_entries": st_["entries"], "longest_tile_list": st_["longest_list"], "unordered_draws": st_["unordered_draws"], "staged_list_draws": st_["staged_draws"],
                "depth_sort_passes": st_["depth_sort_passes"], "record_bytes_read_by_projection": st_["record_read_bytes"],
                "aborted_frames_in_timed_windows": r["aborted_in_timed_windows"], "timed_window_attempts": r["timed_window_attempts"],
                "latency_ms_one_lane": round(o["ms_per_step"], 5) if o else None,
                "roofline": roofline_block(st_, r["stage_ms"], r["warm_ms"], nn, r["ms_per_step"], tfile(ptag), o)}
    c3 = side_block(r3, one3, "10,000,000 random 3D splats in a 400^3 cube, single 1080p frame (BASELINE.json configs[2])", n3, "c3") if r3 else None
    c4 = side_block(r4, one4, "1,000,000 4D splats (BASELINE.json configs[3]'s set: mu_t ~ U[0,50], lifetime ~ U[0.5,2], velocity ~ U[-5,5]^3) at t = 25, single 1080p frame, ONE GPU", n, "c4") if r4 else None
    cpu = None if args.no_cpu_baseline else cpu_baseline(rec, cam, view, proj)
    st = res["stats"]
    return {
        "metric": "Gaussians/sec + ms/frame @1080p; sort permutation bit-exact",
        "value": res["value"], "unit": "splats/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "timing": f"median of {args.windows} windows of {args.steps} steps, each window between two synchronisations; a step is pipelined throughput ({st['lanes']} frame lanes in flight)",
        "windows_ms_per_step": res["windows_ms_per_step"],
        "timed_window_attempts": res["timed_window_attempts"],
        "latency_ms_one_lane": round(one["ms_per_step"], 5) if one else None,
        "config": {"workload": (f"{n:,} 4D splats of BASELINE.json configs[3]'s set at t = 25, single 1080p frame, one GPU" if args.four_d else
                                f"{n:,} random 3D splats in a 400^3 cube, single 1080p frame" + (" (BASELINE.json configs[1])" if n == 1_000_000 else " (BASELINE.json configs[2])" if n == 10_000_000 else "") +
                                (" — EXPERIMENT --spatial-order: records uploaded in Morton order of their positions, not the configuration's random order" if args.spatial_order else "")),
                   "splats": n, "width": W, "height": H, "sort": "on", "frames_per_step": 1, "frame_lanes": st["lanes"],
                   "tile_list_entries": st["entries"], "longest_tile_list": st["longest_list"], "unordered_draws": st["unordered_draws"], "staged_list_draws": st["staged_draws"], "overflow_reruns": st["reruns"],
                   "depth_sort_passes": st["depth_sort_passes"], "record_bytes_read_by_projection": st["record_read_bytes"], "tiles_composited": st["composited_tiles"], "tiles": st["tiles"],
                   "aborted_frames_in_timed_windows": res["aborted_in_timed_windows"], "key_index_buffer_pairs": kb, "lane_streams_rejected_at_create": st["lane_streams_rejected"], "lanes_sharing_a_hardware_queue": st["lanes_sharing_a_queue"]},
        "one_key_index_pair": one_pair,
        "roofline": roofline,
        "c3": c3,
        "c4_n1": c4,
        "cpu_baseline": cpu,
    }


def multi_gpu(args, gs4d, scenes, torch, rank, local_rank, world, backend):
    import torch.distributed as dist
    sharding = importlib.import_module("4dgaussiansplatrendering_amd.sharding")
    n = args.splats
    cam = scenes.CAM_CUBE
    view = gs4d.look_at(cam[0], cam[1])
    proj = gs4d.perspective(scenes.FOV, W, H, scenes.ZNEAR, scenes.ZFAR)
    pos4, q, scale, life, fade, vel, rgba = scenes.cube_params_4d(n)
    rec = gs4d.build_records_4d(pos4, q, scale, life, fade, vel, rgba)
    # the context first, the communicator after it: the frame lanes' streams are then the first streams this process creates (streams that are
    # alive when a context is created change how HIP maps its lanes onto hardware queues: tools/order_effect.py)
    sc = Scene(gs4d, rec, cam, view, proj, local_rank)
    ctx = sc.ctx
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    # A stream of our own becomes torch's current stream AND the context's caller stream: a packed frame is ordered before the gather that
    # sends it, and the next pack into the same slot after the gather that still reads it.  (Not torch's default stream: its handle is NULL,
    # which gs4d_set_stream reads as "no caller stream".)
    # Priority: with peers to send to, the stream is created in HIP's LOWEST priority class — by hipStreamCreateWithPriority itself and wrapped
    # as a torch ExternalStream: torch.cuda.Stream(priority=...) cannot do it on ROCm (c10/hip/HIPStream.h maps the least priority to 0, the
    # default class: the round-3 line claimed a class it did not have).  HIP keeps a pool of hardware queues per class and a queue executes in
    # order; in the default class this stream would share a queue with one frame lane (emulated on one GPU with gs4d_sweep --fake-comm-us 150:
    # 68.6 ms per sweep in the default class, 46.5 in the lowest).  What runs on it: the RGBA8 packs' hand-off events and whatever
    # the process group orders against torch's current stream.  The RCCL kernels themselves run on ProcessGroupNCCL's own internal stream (default
    # class, not ours to choose from Python) — `comm.rccl_kernels_on` in the line says so; host/gs4d_sweep.cpp, which calls ncclSend/ncclRecv
    # itself on a stream it created, is the host where the transfer really sits in the lowest class.
    prio_info = {"requested": "default", "actual": None, "range_least_greatest": None}
    comm_stream = None
    if world > 1 and backend == "nccl":
        try:
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            least, greatest = ctypes.c_int(0), ctypes.c_int(0)
            assert hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)) == 0
            raw = ctypes.c_void_p()
            assert hip.hipStreamCreateWithPriority(ctypes.byref(raw), ctypes.c_uint(1), ctypes.c_int(least.value)) == 0      # hipStreamNonBlocking
            comm_stream = torch.cuda.ExternalStream(raw.value, device=torch.device("cuda", local_rank))
            got = ctypes.c_int(0)
            hip.hipStreamGetPriority(raw, ctypes.byref(got))
            prio_info = {"requested": "lowest", "actual": got.value, "range_least_greatest": [least.value, greatest.value]}
        except Exception as e:      # noqa: BLE001
            prio_info = {"requested": "lowest", "actual": None, "error": repr(e)}
            comm_stream = None
    if comm_stream is None:
        comm_stream = torch.cuda.Stream()
    torch.cuda.set_stream(comm_stream)
    assert comm_stream.cuda_stream != 0
    ctx.set_stream(comm_stream.cuda_stream)
    gdev = "cuda" if backend == "nccl" else "cpu"
    G = max(1, args.gather_every)
    pct = args.rank0_frames_pct if args.rank0_frames_pct is not None else sharding.default_rank0_pct(world)
    mine = sharding.frames_for_rank(SWEEP_FRAMES, rank, world, pct)       # pct = 100: frame k -> rank k mod world
    most = sharding.most_frames(SWEEP_FRAMES, world, pct)                 # ranks with fewer frames pad their last batches
    # two batch buffers, used alternately: the packs of batch b + 1 fill one while the gather of batch b still reads the other.  A pack waits
    # only for the event recorded behind the gather that last read ITS buffer (gs4d_read_frame_rgba8_device_after), not for the comm stream.
    batch = [torch.zeros((G, H * W), dtype=torch.int32, device="cuda") for _ in range(2)]
    gathered = [[torch.empty((G, H * W), dtype=torch.int32, device=gdev) for _ in range(world)] for _ in range(2)] if rank == 0 else [None, None]
    free_ev = [torch.cuda.Event(), torch.cuda.Event()]
    free_valid = [False, False]
    pipelined = ctx.stats()["lanes"] >= 2                                 # with one frame lane there is no previous image to read
    comm = {"busy_ms": 0.0, "tail_ms": 0.0, "pairs": []}

    def render(j):
        sc.frame(sharding.sweep_time(mine[j], SWEEP_FRAMES))

    def pack(j, frames_back, x, slot):
        ctx.read_frame_rgba8_device_after(frames_back, batch[x][slot].data_ptr(), H * W * 4, free_ev[x].cuda_event if free_valid[x] else None)

    def gather(x, lo, hi, _first):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(comm_stream)
        part = batch[x][lo:hi]
        sharding.gather_frames(dist, part if backend == "nccl" else part.cpu(), [g[lo:hi] for g in gathered[x]] if rank == 0 else None, dst=0)
        e1.record(comm_stream)
        free_ev[x].record(comm_stream)
        free_valid[x] = True
        comm["pairs"].append((e0, e1))

    def sweep(_k):
        """One step: this rank's frames of the 256-frame sweep (sharding.run_sweep: the presentation loop the gloo tests drive too).  Presentation is
        software-pipelined as a swap chain is, every G presented frames the batch is gathered on rank 0 (the last batch of the sweep in quarters),
        every rank counts `most` presentations, so all ranks make the same collective calls whatever the deal."""
        sharding.run_sweep(len(mine), most, G, pipelined, render, pack, gather)

    def sweep_one_gpu(_k):
        """The same sweep with every frame on THIS GPU and nothing sent anywhere (frames packed to RGBA8 as above): the N = 1 point of the
        strong-scaling curve of this workload (the driver's N = 1 line is a different workload: configs[1])."""
        for k in range(SWEEP_FRAMES):
            sc.frame(sharding.sweep_time(k, SWEEP_FRAMES))
            if pipelined and k >= 1:
                ctx.read_frame_rgba8_device_after(1, batch[0][(k - 1) % G].data_ptr(), H * W * 4)
            elif not pipelined:
                ctx.read_frame_rgba8_device_after(0, batch[0][k % G].data_ptr(), H * W * 4)
        if pipelined:
            ctx.read_frame_rgba8_device_after(0, batch[0][(SWEEP_FRAMES - 1) % G].data_ptr(), H * W * 4)

    def fence_local():
        ctx.finish()
        torch.cuda.synchronize()

    one_gpu = None
    if rank == 0:
        so = timed_windows(sweep_one_gpu, fence_local, 1, 1, 3)
        so_ms = sorted(1e3 * x for x in so)[1]
        one_gpu = {"ms_per_step": round(so_ms, 4), "value": n * SWEEP_FRAMES / (so_ms * 1e-3), "unit": "splats/s",
                   "note": "the whole sweep on rank 0's GPU alone, measured by rank 0 before the distributed windows (median of 3): the N = 1 point of THIS workload"}

    def fence():
        ctx.finish()
        t0 = time.perf_counter()
        torch.cuda.synchronize()
        comm["tail_ms"] += 1e3 * (time.perf_counter() - t0)     # how long the comm stream ran on after this rank's last frame was rendered
        comm["busy_ms"] += sum(a.elapsed_time(b) for a, b in comm["pairs"])
        comm["pairs"].clear()
        dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(el):
        tt = torch.tensor([el], dtype=torch.float64, device=gdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def timed():
        for k in range(args.warmup):
            sweep(k)
        fence()
        comm["busy_ms"] = comm["tail_ms"] = 0.0
        return timed_windows(sweep, fence, args.steps, 0, args.windows, reduce_max)
    secs = timed()
    nsweeps = max(1, args.steps * args.windows)
    mine_comm = torch.tensor([comm["busy_ms"] / nsweeps, comm["tail_ms"] / max(1, args.windows)], dtype=torch.float64, device=gdev)
    all_comm = [torch.zeros(2, dtype=torch.float64, device=gdev) for _ in range(world)]
    dist.all_gather(all_comm, mine_comm)
    stats = ctx.stats()
    sc.close()
    dist.destroy_process_group()
    ms = sorted(1e3 * s / args.steps for s in secs)
    med = ms[len(ms) // 2]
    alg = algorithmic_bytes(n, W, H)
    frame_ach = alg["frame"] * SWEEP_FRAMES / (med * 1e-3) / 1e9
    return {
        "metric": "Gaussians/sec + ms/frame @1080p; sort permutation bit-exact",
        "value": n * SWEEP_FRAMES / (med * 1e-3), "unit": "splats/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": med, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "timing": f"median of {args.windows} windows of {args.steps} steps (a step = the whole {SWEEP_FRAMES}-frame sweep), barrier + synchronize around every window, maximum over ranks",
        "windows_ms_per_step": [round(1e3 * s / args.steps, 4) for s in secs],
        "config": {"workload": "1,000,000 4D splats, 256-frame time sweep t_k = 50 k / 255, frames dealt to the ranks, RGBA8 frames gathered on rank 0 (BASELINE.json configs[3])",
                   "splats": n, "width": W, "height": H, "sort": "on", "frames_per_step": SWEEP_FRAMES, "ms_per_frame": med / SWEEP_FRAMES,
                   "rank0_frames_pct_of_equal_share": pct, "frames_per_rank": [len(sharding.frames_for_rank(SWEEP_FRAMES, r, world, pct)) for r in range(world)],
                   "gather_format": "RGBA8 (presentation format, 4 B/pixel): the float image the <= 1e-4 parity bar is stated on stays on the rendering GPU",
                   "frames_per_gather_per_rank": G, "batch_buffers": 2, "frame_lanes": stats["lanes"], "tile_list_entries": stats["entries"], "overflow_reruns": stats["reruns"],
                   "aborted_frames": stats["aborted_discarded"]},
        "comm": {"stream_busy_ms_per_sweep_per_rank": [round(float(t[0]), 3) for t in all_comm], "stream_tail_ms_per_window_per_rank": [round(float(t[1]), 3) for t in all_comm],
                 "stream_priority": prio_info, "rccl_kernels_on": "ProcessGroupNCCL's internal stream (default priority class); the hand-off events and the packs' ordering are on the stream above",
                 "note": "busy: sum of the gathers' durations on the comm stream (for a sender that includes waiting for rank 0's matching receive); tail: time the comm stream ran on after the rank's last frame was rendered (exposed)"},
        "roofline": {"bound": "hbm", "kernel": None, "achieved": round(frame_ach, 2), "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": round(frame_ach / (HBM_PEAK_GBS * world), 5),
                     "traffic": None, "note": "whole-job algorithmic bytes over all ranks against N x 8 TB/s; the per-kernel figure is in the N = 1 line"},
        "one_gpu_same_workload": one_gpu,
        "cpu_baseline": None,
    }


def cpu_baseline(rec, cam, view, proj, budget_s=(12.0, 6.0)):
    """The CPU restatement (oracle/, kind 'port') timed on this host over a bounded sample of the same workload: whole frames
    (keygen + sort + preprocess + composite) repeated for ~12 s single-threaded — the reference's only CPU stage on this path,
    the key loop, is single-threaded — and for ~6 s on all host cores.  `value` is the all-core rate."""
    import oracle_lib
    n = rec.shape[0]
    cores = min(os.cpu_count() or 1, 64)

    def run(threads, budget):
        frames, t0, ms_acc = 0, time.perf_counter(), np.zeros(4)
        while True:
            _, _, ms = oracle_lib.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H, nthreads=threads)
            ms_acc += ms
            frames += 1
            el = time.perf_counter() - t0
            if el >= budget or frames >= 64:
                return frames, el, ms_acc / frames
    f1, t1, ms1 = run(1, budget_s[0])
    fn, tn, msn = run(cores, budget_s[1])
    names = ("keygen", "sort", "preprocess", "composite")
    return {"value": n * fn / tn, "unit": "splats/s", "cores": cores, "kind": "port",
            "sample": f"{fn} whole frames of the same workload ({n} splats, 1080p) in {tn:.1f} s on {cores} threads; {f1} frames in {t1:.1f} s single-threaded",
            "single_thread_value": n * f1 / t1,
            "stage_ms_single_thread": dict(zip(names, (round(float(x), 2) for x in ms1))),
            "stage_ms_all_cores": dict(zip(names, (round(float(x), 2) for x in msn)))}


if __name__ == "__main__":
    main()
