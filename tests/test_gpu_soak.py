"""GPU: a long random walk through the states of the staged-list machinery (csrc/gs4d_api.hip run_draw / resolve_lane): 6 000 frames of a 4D set in
100 segments — the camera jumps (far, near, turned away so that the picture lands elsewhere, back), time runs, frames are cleared away unread for
most of a segment — and at the end of every segment the picture must equal, bit for bit, the one a context with exact lists (GS4D_STAGED=0) draws
for the same camera and time.  Guesses miss on the way (capacities, the launch box); no frame may come out wrong, hang or fail.

Reference path: Scenes.h:312-339; checker for eight of the segments: oracle/gs4d_oracle.cpp (image L-infinity <= 1e-4)."""
import numpy as np
import pytest

import scenes
from test_gpu_staged import frame, make_ctx, mats

pytestmark = pytest.mark.gpu
TOL = 1e-4


def test_random_walk_of_cameras_and_times(gs4d, oracle, monkeypatch):
    n, W, H = 50_000, 960, 540
    p4, q4, s4, life, fade, vel, col4 = scenes.cube_params_4d(n, seed=5)
    rec = gs4d.build_records_4d(p4, q4, s4 * 2.0, life * 8.0, fade, vel, col4)
    staged, bufs = make_ctx(gs4d, W, H, rec, monkeypatch, staged=True)
    exact, ebufs = make_ctx(gs4d, W, H, rec, monkeypatch, staged=False)
    monkeypatch.delenv("GS4D_STAGED", raising=False)
    rng = np.random.default_rng(2024)
    d0 = np.array(scenes.CAM_CUBE[1], np.float64)
    checked = 0
    for seg in range(100):
        dist = float(rng.choice([0.6, 1.0, 2.5, 4.0]))
        a = np.radians(float(rng.choice([0.0, 0.0, 7.0, -9.0, 15.0])))
        d = np.array([d0[0] * np.cos(a) - d0[2] * np.sin(a), d0[1], d0[0] * np.sin(a) + d0[2] * np.cos(a)])
        cam = (tuple(float(x) * dist for x in scenes.CAM_CUBE[0]), tuple(float(x) for x in d))
        t0 = float(rng.uniform(5.0, 45.0))
        for k in range(60):
            frame(staged, gs4d, bufs, n, cam, W, H, t=t0 + 0.01 * k)
        t = t0 + 0.01 * 59
        img = staged.read_pixels()
        perm = staged.read(bufs[2], np.uint32, n)
        frame(exact, gs4d, ebufs, n, cam, W, H, t=t)
        eimg = exact.read_pixels()
        assert np.array_equal(perm, exact.read(ebufs[2], np.uint32, n)), seg
        assert np.array_equal(img.view(np.uint32), eimg.view(np.uint32)), (seg, cam, t)
        if seg % 13 == 5:
            view, proj = mats(gs4d, cam, W, H)
            oimg, operm, _ = oracle.render_4d(rec, True, t, 0.0, cam[0], view, proj, W, H)
            assert np.array_equal(perm, operm)
            assert np.abs(img.astype(np.float64) - oimg).max() <= TOL
            checked += 1
    st, se = staged.stats(), exact.stats()
    staged.close()
    exact.close()
    assert checked == 8
    assert se["staged_draws"] == 0
    assert st["staged_draws"] > 5000 and 1 <= st["staged_misses"] <= 500, st          # most frames staged; the jumps did miss
