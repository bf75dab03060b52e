"""GPU: staged tile lists (csrc/tilelist.hip, k_bucket_tiles_staged; csrc/preprocess.hip, k_project_count<.., 2>).  Once a draw of a scene has
reported its fullest segment, its longest (bucket, segment) run and its fullest bucket, the following draws let the projection kernel count, scan
and place its segment's list entries in LDS and write them out as one dense block — no scan and no scatter kernel.  The capacities are guesses
(statistics plus a margin) that the device checks; a draw that does not fit is re-run exactly.  The picture must not depend on which way the lists
were built: the compositor orders every list by (key, record).  ("staged_draws" / "staged_misses" in gs4d_get_stats count the staged draws and
the guesses that failed; GS4D_STAGED=0 switches the staging off.)

Reference path: Scenes.h:312-339 (key loop -> sort -> Draw); checker: oracle/gs4d_oracle.cpp.  Bar: image L-infinity <= 1e-4 against the
checker, bit-identical between the two ways of building the lists, permutation bit-exact.
"""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu
TOL = 1e-4


def mats(gs4d, cam, W, H):
    return gs4d.look_at(cam[0], cam[1]), gs4d.perspective(scenes.FOV, W, H, scenes.ZNEAR, scenes.ZFAR)


def frame(ctx, gs4d, bufs, n, cam, W, H, t=0.0, sort=True):
    """one frame of the reference's loop: Clear -> key loop -> sort -> Draw"""
    db, kb, ib = bufs
    view, proj = mats(gs4d, cam, W, H)
    ctx.clear()
    ctx.set_uniforms(time=t, min_opacity=0.0, view=view, proj=proj)
    if sort:
        ctx.keygen(db, t, cam[0], kb, ib, n)
        ctx.sort_pairs(kb, ib, n)
        ctx.set_mode(gs4d.MODE_4D_SORTED)
        ctx.bind(1, ib)
        ctx.bind(2, db)
    else:
        ctx.set_mode(gs4d.MODE_4D_DIRECT)
        ctx.bind(1, db)
    ctx.draw_instanced(n)


def make_ctx(gs4d, W, H, rec, monkeypatch, staged=True, lanes=None):
    if staged:
        monkeypatch.delenv("GS4D_STAGED", raising=False)
    else:
        monkeypatch.setenv("GS4D_STAGED", "0")
    if lanes:
        monkeypatch.setenv("GS4D_LANES", str(lanes))
    ctx = gs4d.Context(W, H)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    n = rec.shape[0]
    return ctx, (ctx.buffer(rec), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n))


@pytest.mark.parametrize("sort", [True, False])
def test_staged_frames_equal_exact_frames(gs4d, oracle, monkeypatch, sort):
    n, W, H = 200_000, 960, 540
    pos, q, scale, rgba = scenes.cube_params(n)
    rec = gs4d.build_records_3d(pos, q, scale * 2.0, rgba)
    cam = scenes.CAM_CUBE
    ctx, bufs = make_ctx(gs4d, W, H, rec, monkeypatch, staged=True)
    for _ in range(10):
        frame(ctx, gs4d, bufs, n, cam, W, H, sort=sort)
    img = ctx.read_pixels()
    perm = ctx.read(bufs[2], np.uint32, n) if sort else None
    st = ctx.stats()
    ctx.close()
    assert st["staged_draws"] >= 4 and st["staged_misses"] == 0 and st["reruns"] == 0, st
    ctx2, bufs2 = make_ctx(gs4d, W, H, rec, monkeypatch, staged=False)
    for _ in range(6):
        frame(ctx2, gs4d, bufs2, n, cam, W, H, sort=sort)
    img2 = ctx2.read_pixels()
    st2 = ctx2.stats()
    ctx2.close()
    assert st2["staged_draws"] == 0
    assert np.array_equal(img.view(np.uint32), img2.view(np.uint32)), "the picture depends on how the tile lists were built"
    view, proj = mats(gs4d, cam, W, H)
    if sort:
        eimg, eperm, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
        assert np.array_equal(perm, eperm)
    else:
        p = oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H)
        eimg = oracle.composite(p, None, oracle.MODE_4D, W, H, oracle.clear_image(W, H))
    assert np.abs(img.astype(np.float64) - eimg).max() <= TOL
    assert np.abs(eimg - oracle.CLEAR).max() > 0.05


def test_a_guess_that_does_not_fit_is_rerun_exactly(gs4d, oracle, monkeypatch):
    """far camera (short runs) for a few frames, then a jump into the cube: runs and buckets grow several times over — the staged draw aborts on
    the device, is re-run with exact lists, and the frames after it are staged again with the new sizes"""
    n, W, H = 150_000, 800, 448
    pos, q, scale, rgba = scenes.cube_params(n, seed=7)
    rec = gs4d.build_records_3d(pos, q, scale * 2.0, rgba)
    far = ((1400.0, 900.0, -500.0), scenes.CAM_CUBE[1])
    near = ((330.0, 210.0, -110.0), scenes.CAM_CUBE[1])
    ctx, bufs = make_ctx(gs4d, W, H, rec, monkeypatch, staged=True)
    for _ in range(8):
        frame(ctx, gs4d, bufs, n, far, W, H)
    ctx.finish()
    s0 = ctx.stats()
    assert s0["staged_draws"] >= 3 and s0["staged_misses"] == 0, s0
    frame(ctx, gs4d, bufs, n, near, W, H)
    img = ctx.read_pixels()
    perm = ctx.read(bufs[2], np.uint32, n)
    s1 = ctx.stats()
    assert s1["staged_misses"] >= 1 and s1["reruns"] >= 1, s1
    view, proj = mats(gs4d, near, W, H)
    eimg, eperm, _ = oracle.render_4d(rec, True, 0.0, 0.0, near[0], view, proj, W, H)
    assert np.array_equal(perm, eperm)
    assert np.abs(img.astype(np.float64) - eimg).max() <= TOL
    # the next frames of the near camera: staged again, no further miss, same picture
    for _ in range(8):
        frame(ctx, gs4d, bufs, n, near, W, H)
    img2 = ctx.read_pixels()
    s2 = ctx.stats()
    ctx.close()
    assert s2["staged_draws"] > s1["staged_draws"], (s1, s2)
    assert s2["staged_misses"] <= s1["staged_misses"] + 3            # (the frames already in flight on the other lanes when the first miss was found)
    assert np.array_equal(img2.view(np.uint32), img.view(np.uint32))


def test_a_moving_camera_and_time_stay_correct(gs4d, oracle, monkeypatch):
    """a 4D set under a time and camera sweep: every frame read back and compared with the checker, staged or not"""
    n, W, H = 60_000, 640, 360
    p4, q4, s4, life, fade, vel, col4 = scenes.cube_params_4d(n)
    rec = gs4d.build_records_4d(p4, q4, s4 * 2.0, life * 8.0, fade, vel, col4)
    ctx, bufs = make_ctx(gs4d, W, H, rec, monkeypatch, staged=True)
    worst = 0.0
    for k in range(12):
        ang = 0.05 * k
        cam = ((551.58 * np.cos(ang) - 184.33 * np.sin(ang), 350.43, -184.33 * np.cos(ang) - 551.58 * np.sin(ang)), scenes.CAM_CUBE[1])
        t = 20.0 + 0.5 * k
        frame(ctx, gs4d, bufs, n, cam, W, H, t=t)
        if k % 3 == 2 or k >= 9:
            img = ctx.read_pixels()
            view, proj = mats(gs4d, cam, W, H)
            eimg, eperm, _ = oracle.render_4d(rec, True, t, 0.0, cam[0], view, proj, W, H)
            assert np.array_equal(ctx.read(bufs[2], np.uint32, n), eperm)
            worst = max(worst, float(np.abs(img.astype(np.float64) - eimg).max()))
    st = ctx.stats()
    ctx.close()
    assert worst <= TOL, worst
    assert st["staged_draws"] >= 3, st


def test_quads_and_2d_draws_take_the_staged_lists_too(gs4d, oracle, monkeypatch):
    W, H = 640, 360
    import splat_draw_cases as sd
    cam = sd.cameras(oracle)[0]
    verts = sd.verts72(oracle.golden("splat_draw_3d_in"))
    nq = verts.shape[0]
    proj = gs4d.perspective(scenes.FOV, W, H, scenes.ZNEAR, scenes.ZFAR)
    monkeypatch.delenv("GS4D_STAGED", raising=False)
    ctx = gs4d.Context(W, H)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    vb = ctx.buffer(verts)
    for _ in range(8):
        ctx.clear()
        ctx.set_uniforms(time=0.0, min_opacity=0.0, view=cam["view"], proj=proj)
        ctx.set_mode(gs4d.MODE_3D_FULL)
        ctx.draw_quads(vb, nq)
    img = ctx.read_pixels()
    st = ctx.stats()
    ctx.close()
    assert st["staged_draws"] >= 3 and st["staged_misses"] == 0, st
    p = oracle.preprocess(oracle.MODE_3D, verts, cam["view"], proj, W, H)
    eimg = oracle.composite(p, None, oracle.MODE_3D, W, H, oracle.clear_image(W, H))
    assert np.abs(img.astype(np.float64) - eimg).max() <= TOL


@pytest.mark.parametrize("n,W,H", [(80, 1280, 720), (37, 640, 360), (1000, 1280, 720)])
def test_large_footprints_and_tiny_sets_are_staged_too(gs4d, oracle, monkeypatch, n, W, H):
    """the reference's own teapot records at a close camera: footprints of tens to hundreds of tiles (the wave-cooperative branch of the placing pass,
    runs of dozens of entries), a set smaller than one wave, and 1000 records whose 45 000 entries do not fit a segment block (such a draw is never staged:
    its lists stay exact) — eight frames each, the later ones must equal the first (exact) one and the checker"""
    rec = np.ascontiguousarray(oracle.golden("linear_first1000")[:n], np.float32)
    cam = ((30.0, 45.0, 45.0), (0.0, -1.0, -1.0))
    ctx, bufs = make_ctx(gs4d, W, H, rec, monkeypatch, staged=True)
    imgs = []
    for k in range(8):
        frame(ctx, gs4d, bufs, n, cam, W, H, t=0.0)
        if k in (0, 7):
            imgs.append(ctx.read_pixels())
    st = ctx.stats()
    perm = ctx.read(bufs[2], np.uint32, n)
    ctx.close()
    assert st["staged_misses"] == 0, st
    if n <= 100 and st["unordered_draws"] >= 8:         # (lists short enough for the unordered path and a segment that fits its block: the later frames were staged)
        assert st["staged_draws"] >= 3, st
    if n == 1000:
        assert st["staged_draws"] == 0, st
    assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))
    view, proj = mats(gs4d, cam, W, H)
    eimg, eperm, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
    assert np.array_equal(perm, eperm)
    assert np.abs(imgs[1].astype(np.float64) - eimg).max() <= TOL
    assert np.abs(eimg - oracle.CLEAR).max() > 0.05


def test_the_compositor_is_launched_for_the_box_of_tiles_that_hold_entries(gs4d, oracle, monkeypatch):
    """A far camera: the cube covers the middle of the image.  Once a staged draw has reported which blocks of tiles held entries, the compositing
    kernel of the following staged draws is launched for that box (one block wider) only; the list kernel checks every non-empty tile against it.
    Same picture, bit for bit, as with GS4D_STAGED_BOX=0 (every tile launched) and as the exact lists give."""
    n, W, H = 150_000, 1280, 720
    pos, q, scale, rgba = scenes.cube_params(n, seed=11)
    rec = gs4d.build_records_3d(pos, q, scale * 2.0, rgba)
    far = ((1400.0, 900.0, -500.0), scenes.CAM_CUBE[1])
    imgs = {}
    for box in ("1", "0"):
        monkeypatch.setenv("GS4D_STAGED_BOX", box)
        ctx, bufs = make_ctx(gs4d, W, H, rec, monkeypatch, staged=True)
        for _ in range(12):
            frame(ctx, gs4d, bufs, n, far, W, H)
        imgs[box] = ctx.read_pixels()
        st = ctx.stats()
        ctx.close()
        assert st["staged_draws"] >= 6 and st["staged_misses"] == 0 and st["reruns"] == 0, st
        if box == "1":
            assert 0 < st["composited_tiles"] < st["tiles"] // 2, st
        else:
            assert st["composited_tiles"] == st["tiles"], st
    monkeypatch.delenv("GS4D_STAGED_BOX")
    assert np.array_equal(imgs["1"].view(np.uint32), imgs["0"].view(np.uint32))
    view, proj = mats(gs4d, far, W, H)
    eimg, _, _ = oracle.render_4d(rec, True, 0.0, 0.0, far[0], view, proj, W, H)
    assert np.abs(imgs["1"].astype(np.float64) - eimg).max() <= TOL
    assert np.abs(eimg - oracle.CLEAR).max() > 0.05


def test_entries_outside_the_launch_box_abort_the_draw_and_it_is_rerun(gs4d, oracle, monkeypatch):
    """The camera turns between two frames so that the cube lands in another part of the image: the staged draw's list kernel finds entries outside
    the box the compositor would be launched for, the draw aborts before a pixel is touched and is re-run exactly over the whole image; the next frames
    learn the new box.  A slow pan stays inside the margin and never misses."""
    n, W, H = 100_000, 1280, 720
    pos, q, scale, rgba = scenes.cube_params(n, seed=12)
    rec = gs4d.build_records_3d(pos, q, scale * 2.0, rgba)
    eye = (1400.0, 900.0, -500.0)
    d0 = np.array(scenes.CAM_CUBE[1], np.float64)

    def turned(deg):
        a = np.radians(deg)
        d = np.array([d0[0] * np.cos(a) - d0[2] * np.sin(a), d0[1], d0[0] * np.sin(a) + d0[2] * np.cos(a)])
        return (eye, tuple(float(x) for x in d))

    ctx, bufs = make_ctx(gs4d, W, H, rec, monkeypatch, staged=True)
    for _ in range(10):
        frame(ctx, gs4d, bufs, n, turned(0.0), W, H)
    ctx.finish()
    s0 = ctx.stats()
    assert s0["staged_misses"] == 0 and 0 < s0["composited_tiles"] < s0["tiles"], s0
    # a slow pan: 0.05 degrees a frame (about a pixel and a half)
    for k in range(1, 13):
        frame(ctx, gs4d, bufs, n, turned(0.05 * k), W, H)
    img = ctx.read_pixels()
    s1 = ctx.stats()
    assert s1["staged_misses"] == 0 and s1["reruns"] == 0, s1
    cam = turned(0.6)
    view, proj = mats(gs4d, cam, W, H)
    eimg, _, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
    assert np.abs(img.astype(np.float64) - eimg).max() <= TOL
    # a jump of 12 degrees: the cube moves by a quarter of the image
    cam = turned(12.0)
    frame(ctx, gs4d, bufs, n, cam, W, H)
    img = ctx.read_pixels()
    s2 = ctx.stats()
    assert s2["staged_misses"] >= 1 and s2["reruns"] >= 1, s2
    view, proj = mats(gs4d, cam, W, H)
    eimg, eperm, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
    assert np.array_equal(ctx.read(bufs[2], np.uint32, n), eperm)
    assert np.abs(img.astype(np.float64) - eimg).max() <= TOL
    assert np.abs(eimg - oracle.CLEAR).max() > 0.05
    for _ in range(10):
        frame(ctx, gs4d, bufs, n, cam, W, H)
    img2 = ctx.read_pixels()
    s3 = ctx.stats()
    ctx.close()
    assert s3["staged_draws"] > s2["staged_draws"] and 0 < s3["composited_tiles"] < s3["tiles"], (s2, s3)
    assert s3["staged_misses"] <= s2["staged_misses"] + 3
    assert np.array_equal(img2.view(np.uint32), img.view(np.uint32))


@pytest.mark.parametrize("slabs", [1, 4])
def test_staged_frames_of_a_tile_shard_and_of_sub_lists(gs4d, oracle, monkeypatch, slabs):
    """gs4d_set_tile_shard (a rank of BASELINE.json configs[4]'s tile-row deal) and GS4D_SLABS (sub-lists by key range) with staged lists and the
    launch box: the tenth frame is staged, composited over a box only, and equals the first (every tile launched) bit for bit; the rows of
    the other rank keep the clear colour."""
    import importlib
    sh = importlib.import_module("4dgaussiansplatrendering_amd.sharding")
    n, W, H = 80_000, 1024, 576
    pos, q, scale, rgba = scenes.cube_params(n, seed=21)
    rec = gs4d.build_records_3d(pos, q, scale * 3.0, rgba)
    far = ((1100.0, 700.0, -400.0), scenes.CAM_CUBE[1])
    monkeypatch.setenv("GS4D_SLABS", str(slabs))
    ctx, bufs = make_ctx(gs4d, W, H, rec, monkeypatch, staged=True)
    monkeypatch.delenv("GS4D_SLABS")
    ctx.set_tile_shard(1, 2)
    frame(ctx, gs4d, bufs, n, far, W, H)
    first = ctx.read_pixels()
    s0 = ctx.stats()
    for _ in range(10):
        frame(ctx, gs4d, bufs, n, far, W, H)
    last = ctx.read_pixels()
    s1 = ctx.stats()
    ctx.close()
    # (the first frame's draw may itself have been re-run staged: a first draw that outgrows the entry capacity leaves the statistics behind)
    assert s0["composited_tiles"] == s0["tiles"] and s1["staged_draws"] >= 5 and s1["staged_misses"] == 0, (s0, s1)
    assert 0 < s1["composited_tiles"] < s1["tiles"], s1
    assert np.array_equal(first.view(np.uint32), last.view(np.uint32))
    view, proj = mats(gs4d, far, W, H)
    eimg, _, _ = oracle.render_4d(rec, True, 0.0, 0.0, far[0], view, proj, W, H)
    mine = sh.band_pixel_rows(1, 2, H)
    others = sorted(set(range(H)) - set(mine))
    assert np.abs(last[mine].astype(np.float64) - eimg[mine]).max() <= TOL
    assert np.array_equal(last[others], np.broadcast_to(np.array(gs4d.CLEAR_COLOR, np.float32), (len(others), W, 4)))
    assert np.abs(eimg[mine] - oracle.CLEAR).max() > 0.05
