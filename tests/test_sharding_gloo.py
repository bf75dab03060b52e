"""CPU, world_size 2, gloo: the N>1 path of bench.py — frames dealt round-robin to ranks, one gather of finished frames to
rank 0 per step — produces exactly the frames a single process renders.  The frames themselves come from the CPU oracle
here (the HIP renderer needs a GPU); what is under test is the sharding + gather plumbing the GPU run uses."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
NFRAMES, N, W, H = 6, 400, 96, 64


def _render(frame):
    sys.path[:0] = [HERE, ROOT]
    import oracle_lib, scenes
    sh = importlib.import_module("4dgaussiansplatrendering_amd.sharding")
    gs4d = importlib.import_module("4dgaussiansplatrendering_amd")
    pos4, q, scale, life, fade, vel, rgba = scenes.cube_params_4d(N)
    rec = gs4d.build_records_4d(pos4, q, scale * 6.0, life * 20.0, fade, vel, rgba)
    view = oracle_lib.look_at(*scenes.CAM_CUBE)
    proj = oracle_lib.perspective(scenes.FOV, W, H, scenes.ZNEAR, scenes.ZFAR)
    t = sh.sweep_time(frame, NFRAMES)
    img, _, _ = oracle_lib.render_4d(rec, True, t, 0.0, scenes.CAM_CUBE[0], view, proj, W, H, nthreads=1)
    return img


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path[:0] = [HERE, ROOT]
    sh = importlib.import_module("4dgaussiansplatrendering_amd.sharding")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = {}
    for step in range(NFRAMES // world):
        f = sh.frame_of(step, rank, world)
        assert f in sh.frames_for_rank(NFRAMES, rank, world)
        mine = torch.from_numpy(_render(f))
        gathered = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
        sh.gather_frames(dist, mine, gathered, dst=0)
        if rank == 0:
            for r in range(world):
                frames[sh.frame_of(step, r, world)] = gathered[r].numpy().copy()
    dist.barrier()
    if rank == 0:
        np.save(out, np.stack([frames[k] for k in range(NFRAMES)]))
    dist.destroy_process_group()


def test_two_ranks_render_the_same_sweep(tmp_path):
    out = str(tmp_path / "frames.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    assert got.shape == (NFRAMES, H, W, 4)
    for k in range(NFRAMES):
        assert np.array_equal(got[k], _render(k)), f"frame {k}"
    assert not np.array_equal(got[0], got[NFRAMES - 1])        # the sweep actually changes the picture


def test_round_robin_partition():
    sh = importlib.import_module("4dgaussiansplatrendering_amd.sharding")
    for world in (1, 2, 4, 8):
        seen = sorted(f for r in range(world) for f in sh.frames_for_rank(256, r, world))
        assert seen == list(range(256))
        assert all(len(sh.frames_for_rank(256, r, world)) == 256 // world for r in range(world))
    assert sh.sweep_time(0, 256) == 0.0 and sh.sweep_time(255, 256) == 50.0
