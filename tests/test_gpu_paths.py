"""GPU parity of the paths the round-1 suite did not reach, through the C ABI, against the CPU checker.

* the two ways a draw can build its tile lists — instance-ordered (binning.hip) and unordered + ordered in the compositor
  (tilelist.hip / composite2.hip): same image, including keys that tie, lists that outgrow the compositor's LDS and the fall-back;
* GS4D_KEY_VIEW_Z (the north star's "view-space depth keying"; the reference's key is the Euclidean one, Scenes.h:314-319);
* the caller-stream hand-off (gs4d_set_stream, gs4d_buffer_device_ptr + gs4d_buffer_invalidate, device read-backs): what bench.py's
  multi-GPU leg does with torch's stream, here with a side stream and no collective;
* every ranking variant and tile shape of the radix sort (GS4D_SORT_RANK, GS4D_SORT_SHAPE), and its pass scheduling with exactly one
  live digit (sort contract: radix_sort.hpp:258-392).
Bars: bit-exact for keys and permutations; per-pixel L-infinity <= 1e-4 for float images; <= 1 count for RGBA8.
"""
import os

import numpy as np
import pytest

import scenes
from test_gpu_render import cam_mats, linf, TOL

pytestmark = pytest.mark.gpu


def _ctx(gs4d, W, H, monkeypatch, **env):
    for k in ("GS4D_DRAW_PATH", "GS4D_SORT_RANK", "GS4D_SORT_SHAPE", "GS4D_SORT_RB"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    return gs4d.Context(W, H)


def _sorted_frame(ctx, gs4d, rec, cam, view, proj, t=0.0, key_mode=None):
    n = rec.shape[0]
    db, kb, ib = ctx.buffer(rec), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.clear()
    ctx.set_uniforms(time=t, min_opacity=0.0, view=view, proj=proj)
    ctx.keygen(db, t, cam[0], kb, ib, n, **({} if key_mode is None else {"key_mode": key_mode}))
    keys = ctx.read(kb, np.float32, n)
    ctx.sort_pairs(kb, ib, n)
    ctx.set_mode(gs4d.MODE_4D_SORTED)
    ctx.bind(1, ib)
    ctx.bind(2, db)
    ctx.draw_instanced(n)
    img = ctx.read_pixels()
    perm = ctx.read(ib, np.uint32, n)
    st = ctx.stats()
    for b in (db, kb, ib):
        ctx.delete(b)
    return img, perm, keys, st


@pytest.mark.parametrize("n,W,H,scale", [(200000, 1920, 1080, 1.0), (30000, 640, 360, 4.0)])
def test_ordered_and_unordered_lists_give_the_same_frame(gs4d, oracle, monkeypatch, n, W, H, scale):
    pos, q, sc, rgba = scenes.cube_params(n, seed=41)
    rec = gs4d.build_records_3d(pos, q, sc * scale, rgba)
    cam = scenes.CAM_CUBE
    view, proj = cam_mats(gs4d, cam, W, H)
    eimg, eperm, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
    imgs = {}
    for path in ("auto", "ordered"):
        ctx = _ctx(gs4d, W, H, monkeypatch, **({"GS4D_DRAW_PATH": "ordered"} if path == "ordered" else {}))
        img, perm, _, st = _sorted_frame(ctx, gs4d, rec, cam, view, proj)
        ctx.close()
        assert np.array_equal(perm, eperm)
        assert linf(img, eimg) <= TOL
        assert (st["unordered_draws"] > 0) == (path == "auto")
        assert (st["tile_sort_passes"] == 0) == (path == "auto")
        imgs[path] = img
    assert np.array_equal(imgs["auto"], imgs["ordered"])        # same records, same blend order, same chunking: the same bits


def test_equal_depth_keys_blend_in_index_order(gs4d, oracle, monkeypatch):
    """Coincident splats have equal keys: the stable sort keeps them in index order, and the unordered path must blend them so."""
    n0, rep, W, H = 1500, 4, 640, 360
    pos, q, sc, rgba = scenes.cube_params(n0, seed=43)
    pos, q, sc = np.repeat(pos * 0.3, rep, 0), np.repeat(q, rep, 0), np.repeat(sc * 5.0, rep, 0)
    _, _, _, rgba = scenes.cube_params(n0 * rep, seed=44)       # different colours on the coincident splats
    rgba[:, 3] = 0.9
    rec = gs4d.build_records_3d(pos, q, sc, rgba)
    cam = ((150.0, 100.0, -60.0), (-0.77, -0.57, 0.27))
    view, proj = cam_mats(gs4d, cam, W, H)
    ctx = _ctx(gs4d, W, H, monkeypatch)
    img, perm, keys, st = _sorted_frame(ctx, gs4d, rec, cam, view, proj)
    ctx.close()
    assert np.unique(keys.view(np.uint32)).size <= n0           # the keys do tie
    eimg, eperm, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
    assert np.array_equal(perm, eperm)
    assert linf(img, eimg) <= TOL
    assert st["unordered_draws"] == 1
    # blending the ties the other way round is a different picture: the test can tell
    flipped = np.lexsort((-np.arange(n0 * rep, dtype=np.int64), keys.view(np.uint32))).astype(np.uint32)       # ascending key, DEscending index
    eproj = oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H)
    other = oracle.composite(eproj, flipped, oracle.MODE_4D, W, H, oracle.clear_image(W, H))
    assert linf(other, eimg) > 1e-2


def test_long_lists_grow_the_compositor_then_fall_back(gs4d, oracle, monkeypatch):
    """Hundreds, then thousands, of splats on the same pixels.  The list-building kernels report the longest list; the draw is re-run with
    a larger LDS list capacity in the compositing wave (up to 1024 entries), and beyond that on the instance-ordered path — where the
    context then stays.  Every time the same picture."""
    W, H = 256, 256
    cam = ((0.0, 0.0, 60.0), (0.0, 0.0, -1.0))
    view, proj = cam_mats(gs4d, cam, W, H)
    ctx = _ctx(gs4d, W, H, monkeypatch)
    seen = []
    for n, spread in ((700, 0.1), (700, 0.1), (2500, 0.1), (3000, 0.0), (120, 0.1)):
        pos, q, sc, rgba = scenes.cube_params(n, seed=50 + n)
        rgba[:, 3] *= 0.05
        pos[:, 0:2] = 0.0                                                     # all on the view axis: one spot of the image; spread 0: one depth, equal keys
        rec = gs4d.build_records_3d(pos * spread, q, sc * 1.5, rgba)
        img, perm, _, st = _sorted_frame(ctx, gs4d, rec, cam, view, proj)
        eimg, eperm, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
        assert np.array_equal(perm, eperm)
        assert linf(img, eimg) <= TOL
        assert np.abs(eimg - np.array(gs4d.CLEAR_COLOR, np.float32)).max() > 0.05
        seen.append(st)
    # frame 1: unordered, re-run with a longer list capacity; frame 2: unordered, no re-run
    assert seen[0]["unordered_draws"] == 1 and seen[0]["reruns"] == 1 and 256 < seen[0]["longest_list"] <= 700 and seen[0]["tile_sort_passes"] == 0
    assert seen[1]["unordered_draws"] == 2 and seen[1]["reruns"] == 1 and seen[1]["tile_sort_passes"] == 0
    # frame 3: 2500 entries on a tile: more than the compositing wave holds -> one re-run, on the instance-ordered path; frames 4 and 5 stay there
    assert seen[2]["reruns"] == 2 and seen[2]["tile_sort_passes"] >= 2
    assert seen[3]["unordered_draws"] == seen[2]["unordered_draws"] and seen[3]["reruns"] == 2 and seen[3]["tile_sort_passes"] >= 2
    assert seen[4]["unordered_draws"] == seen[2]["unordered_draws"] and seen[4]["reruns"] == 2 and seen[4]["tile_sort_passes"] >= 2
    ctx.close()


@pytest.mark.parametrize("slabs", [4, 32])
def test_depth_slabs(gs4d, oracle, monkeypatch, slabs):
    """GS4D_SLABS: a tile's list kept as sub-lists by equal ranges of the blend key, each ordered by itself in the compositing wave, far
    sub-list first.  (The library itself sends long lists to the bucket sort; the mechanism is kept, and kept tested.)  Same frame as the
    instance-ordered path, bit for bit; sub-lists beyond 1024 entries send the draw to the instance-ordered path."""
    n, W, H = 60000, 640, 360
    pos, q, sc, rgba = scenes.cube_params(n, seed=61)
    rgba[:, 3] *= 0.3
    rec = gs4d.build_records_3d(pos * 0.25, q, sc * 4.0, rgba)
    cam = ((150.0, 100.0, -60.0), (-0.77, -0.57, 0.27))
    view, proj = cam_mats(gs4d, cam, W, H)
    eimg, eperm, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
    ctx = _ctx(gs4d, W, H, monkeypatch, GS4D_SLABS=slabs)
    img, perm, _, st = _sorted_frame(ctx, gs4d, rec, cam, view, proj)
    ctx.close()
    assert np.array_equal(perm, eperm) and linf(img, eimg) <= TOL
    assert st["unordered_draws"] >= 1 and st["tile_sort_passes"] == 0
    ctx = _ctx(gs4d, W, H, monkeypatch, GS4D_DRAW_PATH="ordered")
    img_o, _, _, _ = _sorted_frame(ctx, gs4d, rec, cam, view, proj)
    ctx.close()
    assert np.array_equal(img, img_o)
    # 5000 splats of one depth on one spot: no key range separates them
    pos2, q2, sc2, rgba2 = scenes.cube_params(5000, seed=62)
    rgba2[:, 3] *= 0.02
    pos2[:, :] = 0.0
    rec2 = gs4d.build_records_3d(pos2, q2, sc2 * 1.5, rgba2)
    cam2 = ((0.0, 0.0, 60.0), (0.0, 0.0, -1.0))
    view2, proj2 = cam_mats(gs4d, cam2, W, H)
    ctx = _ctx(gs4d, W, H, monkeypatch, GS4D_SLABS=slabs)
    img2, perm2, _, st2 = _sorted_frame(ctx, gs4d, rec2, cam2, view2, proj2)
    ctx.close()
    eimg2, eperm2, _ = oracle.render_4d(rec2, True, 0.0, 0.0, cam2[0], view2, proj2, W, H)
    assert np.array_equal(perm2, eperm2) and linf(img2, eimg2) <= TOL and st2["tile_sort_passes"] >= 2


def test_long_lists_in_a_dense_cloud(gs4d, oracle, monkeypatch):
    """A dense cloud (lists of a few thousand entries on the busiest tiles, keys that tie in pairs): the draw is issued on the unordered path,
    aborts on the device and is re-run on the instance-ordered path from a REGENERATED order (it never read the caller's index): same frame as
    with that path forced, bit for bit; and instance-index order (4D-direct) on the same context."""
    n, W, H = 300000, 640, 360
    pos, q, sc, rgba = scenes.cube_params(n, seed=63)
    rgba[:, 3] *= 0.2
    pos = np.repeat(pos[: n // 2] * 0.15, 2, 0)                 # every position twice: keys tie in pairs
    rec = gs4d.build_records_3d(pos, q, sc * 2.0, rgba)
    cam = ((150.0, 100.0, -60.0), (-0.77, -0.57, 0.27))
    view, proj = cam_mats(gs4d, cam, W, H)
    eimg, eperm, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H, nthreads=16)
    imgs = {}
    for path in ("auto", "ordered"):
        ctx = _ctx(gs4d, W, H, monkeypatch, **({"GS4D_DRAW_PATH": "ordered"} if path == "ordered" else {}))
        img, perm, _, st = _sorted_frame(ctx, gs4d, rec, cam, view, proj)
        assert np.array_equal(perm, eperm)
        assert linf(img, eimg) <= TOL
        if path == "auto":
            assert st["unordered_draws"] == 1 and st["reruns"] >= 1 and st["tile_sort_passes"] >= 2 and st["longest_list"] > 1024
            db = ctx.buffer(rec)
            ctx.clear()
            ctx.set_mode(gs4d.MODE_4D_DIRECT)
            ctx.bind(1, db)
            ctx.draw_instanced(n)
            direct = ctx.read_pixels()
            eproj = oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H)
            edirect = oracle.composite(eproj, None, oracle.MODE_4D, W, H, oracle.clear_image(W, H), nthreads=16)
            assert linf(direct, edirect) <= TOL
        ctx.close()
        imgs[path] = img
    assert np.array_equal(imgs["auto"], imgs["ordered"])


@pytest.mark.parametrize("path", ["auto", "ordered"])
def test_view_z_key_mode(gs4d, oracle, monkeypatch, path):
    n, W, H = 120000, 960, 540
    pos4, q, sc, life, fade, vel, rgba = scenes.cube_params_4d(n, seed=61)
    rec = gs4d.build_records_4d(pos4, q, sc * 3.0, life, fade, vel, rgba)
    cam, t = scenes.CAM_CUBE, 21.5
    view, proj = cam_mats(gs4d, cam, W, H)
    ctx = _ctx(gs4d, W, H, monkeypatch, **({"GS4D_DRAW_PATH": "ordered"} if path == "ordered" else {}))
    img, perm, keys, st = _sorted_frame(ctx, gs4d, rec, cam, view, proj, t=t, key_mode=gs4d.KEY_VIEW_Z)
    ctx.close()
    eidx, ekeys = oracle.keygen_viewz(rec, t, view)
    assert np.array_equal(keys.view(np.uint32), ekeys.view(np.uint32))
    _, eperm = oracle.sort_pairs(ekeys.view(np.uint32), eidx, "std")
    assert np.array_equal(perm, eperm)
    _, rkeys = oracle.keygen(rec, t, cam[0])
    _, rperm = oracle.sort_pairs(rkeys.view(np.uint32), eidx, "std")
    assert not np.array_equal(eperm, rperm)                      # it is a different order from the reference key's
    eproj = oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H, t, 0.0)
    eimg = oracle.composite(eproj, eperm, oracle.MODE_4D, W, H, oracle.clear_image(W, H))
    assert linf(img, eimg) <= TOL
    assert (st["unordered_draws"] > 0) == (path == "auto")


def test_tile_sort_of_a_4k_frame_takes_two_passes(gs4d, oracle, monkeypatch):
    """The instance-ordered path sorts its entries by tile id: 129 600 tiles at 3840 x 2160 are 17 bits — three passes of 8-bit digits, two of
    9-bit ones (k_bin_emit counts the digits the sort will use).  Same image either way, and the checker's."""
    n, W, H = 60000, 3840, 2160
    pos, q, sc, rgba = scenes.cube_params(n, seed=13)
    rec = gs4d.build_records_3d(pos, q, sc * 2.0, rgba)
    cam = scenes.CAM_CUBE
    view, proj = cam_mats(gs4d, cam, W, H)
    imgs = {}
    for rb, passes in ((0, 2), (8, 3)):
        ctx = _ctx(gs4d, W, H, monkeypatch, GS4D_DRAW_PATH="ordered", **({"GS4D_SORT_RB": rb} if rb else {}))
        img, perm, keys, st = _sorted_frame(ctx, gs4d, rec, cam, view, proj)
        assert st["tile_sort_passes"] == passes and st["unordered_draws"] == 0, st
        ctx.close()
        imgs[rb] = img
    assert np.array_equal(imgs[0], imgs[8])
    eimg, eperm, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
    assert np.array_equal(perm, eperm)
    assert linf(imgs[0], eimg) <= TOL


@pytest.mark.parametrize("fuse", [1, 0])
@pytest.mark.parametrize("path", ["auto", "ordered"])
def test_nine_bit_digits_for_wide_key_spans(gs4d, oracle, monkeypatch, path, fuse):
    """Moving (4D) splats: the host-proven span of the depth keys covers the motion and is wider than 2^24 bit patterns (25 bits here) — four
    passes of 8-bit digits, THREE of 9-bit ones (sort.hip sort_plan_rb).  Keys, permutation (bit-exact against the stable sort of the reference's keys) and image
    on both draw paths, with the keys generated by the draw (the projection kernel counts the 9-bit digits) and by k_keygen."""
    n, W, H = 150000, 960, 540
    pos4, q, sc, life, fade, vel, rgba = scenes.cube_params_4d(n, seed=71)
    rec = gs4d.build_records_4d(pos4, q, sc * 3.0, life, fade, vel * 0.3, rgba)      # (at full speed the extrapolated positions reach the camera: no upper bound for the keys at all)
    cam, t = scenes.CAM_CUBE, 21.5
    view, proj = cam_mats(gs4d, cam, W, H)
    env = {"GS4D_FUSE_KEYGEN": fuse}
    if path == "ordered":
        env["GS4D_DRAW_PATH"] = "ordered"
    ctx = _ctx(gs4d, W, H, monkeypatch, **env)
    db, kb, ib = ctx.buffer(rec), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    for _ in range(2):                                             # twice: the second frame reuses the sorter's alternating histogram slots
        ctx.clear()
        ctx.set_uniforms(time=t, min_opacity=0.0, view=view, proj=proj)
        ctx.keygen(db, t, cam[0], kb, ib, n)
        ctx.sort_pairs(kb, ib, n)
        ctx.set_mode(gs4d.MODE_4D_SORTED)
        ctx.bind(1, ib)
        ctx.bind(2, db)
        ctx.draw_instanced(n)
        img = ctx.read_pixels()
        st = ctx.stats()
        assert st["depth_sort_passes"] == 3, st
        eidx, ekeys = oracle.keygen(rec, t, cam[0])
        sk, perm = ctx.read(kb, np.uint32, n), ctx.read(ib, np.uint32, n)
        esk, eperm = oracle.sort_pairs(ekeys.view(np.uint32), eidx, "std")
        assert np.array_equal(sk, esk) and np.array_equal(perm, eperm)
        eproj = oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H, t, 0.0)
        eimg = oracle.composite(eproj, eperm, oracle.MODE_4D, W, H, oracle.clear_image(W, H))
        assert linf(img, eimg) <= TOL
    assert (st["unordered_draws"] > 0) == (path == "auto")
    ctx.close()
    # the same frame with 8-bit digits forced: four passes (the span is wider than 2^24), same result
    ctx = _ctx(gs4d, W, H, monkeypatch, GS4D_SORT_RB=8, **env)
    db, kb, ib = ctx.buffer(rec), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.clear()
    ctx.set_uniforms(time=t, min_opacity=0.0, view=view, proj=proj)
    ctx.keygen(db, t, cam[0], kb, ib, n)
    ctx.sort_pairs(kb, ib, n)
    ctx.set_mode(gs4d.MODE_4D_SORTED)
    ctx.bind(1, ib)
    ctx.bind(2, db)
    ctx.draw_instanced(n)
    img8 = ctx.read_pixels()
    assert ctx.stats()["depth_sort_passes"] == 4
    assert np.array_equal(ctx.read(ib, np.uint32, n), eperm) and np.array_equal(img8, img)
    ctx.close()
    monkeypatch.delenv("GS4D_FUSE_KEYGEN", raising=False)


@pytest.mark.parametrize("rank", [1, 2])
@pytest.mark.parametrize("shape,rb", [(1, 8), (2, 8), (3, 8), (4, 8), (5, 8), (6, 8), (7, 8), (2, 9), (3, 9), (5, 9), (6, 9), (7, 9)])
def test_sort_ranking_variants_and_tile_shapes(gs4d, oracle, monkeypatch, rank, shape, rb):
    """Every tile shape of a pass (threads x keys per thread), both rankings, and both digit widths: 8 bits (256 bins, four passes over
    32-bit keys) and 9 bits (512 bins, one thread per bin: the shapes of 512 threads and more; 4 x 9 bits cover the 32)."""
    ctx = _ctx(gs4d, 64, 64, monkeypatch, GS4D_SORT_RANK=rank, GS4D_SORT_SHAPE=shape, GS4D_SORT_RB=rb)
    rng = np.random.default_rng(100 * rank + shape)
    for n, distinct in ((70001, 0), (262144 + 5, 37)):
        keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        if distinct:
            keys = keys[:distinct][rng.integers(0, distinct, n)]
        vals = rng.permutation(n).astype(np.uint32)
        kb, vb = ctx.buffer(keys), ctx.buffer(vals)
        ctx.sort_pairs(kb, vb, n)
        ek, ev = oracle.sort_pairs(keys, vals, "std")
        assert np.array_equal(ctx.read(kb, np.uint32, n), ek)
        assert np.array_equal(ctx.read(vb, np.uint32, n), ev)
        ctx.delete(kb)
        ctx.delete(vb)
    # a frame through the tile sort as well (user-supplied order: the instance-ordered path)
    W = H = 64
    pos, q, sc, rgba = scenes.cube_params(4000, seed=7)
    rec = gs4d.build_records_3d(pos * 0.25, q, sc * 6.0, rgba)
    cam = ((120.0, 80.0, -40.0), (-0.77, -0.57, 0.27))
    view, proj = cam_mats(gs4d, cam, W, H)
    order = rng.permutation(4000).astype(np.uint32)
    db, ib = ctx.buffer(rec), ctx.buffer(order)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.clear()
    ctx.set_uniforms(time=0.0, min_opacity=0.0, view=view, proj=proj)
    ctx.set_mode(gs4d.MODE_4D_SORTED)
    ctx.bind(1, ib)
    ctx.bind(2, db)
    ctx.draw_instanced(4000)
    eimg = oracle.composite(oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H), order, oracle.MODE_4D, W, H, oracle.clear_image(W, H))
    assert linf(ctx.read_pixels(), eimg) <= TOL
    assert ctx.stats()["tile_sort_passes"] >= 2
    ctx.close()


@pytest.mark.parametrize("rb", [8, 9])
@pytest.mark.parametrize("n", [100003, 1 << 20])
@pytest.mark.parametrize("pattern", ["byte0", "byte1", "byte2", "byte3", "bytes0and3"])
def test_sort_with_one_live_digit(gs4d, oracle, monkeypatch, n, pattern, rb):
    """Keys that differ in ONE byte only: three of the four digit passes are identities, and the schedule has to add exactly one
    copying pass so that the result lands in the caller's buffers (the round-1 race was here: waves could disagree about it).  With 9-bit
    digits a byte of the key straddles two digits (two live passes) or sits inside one."""
    ctx = _ctx(gs4d, 64, 64, monkeypatch, GS4D_SORT_RB=rb)
    rng = np.random.default_rng(n % 1000)
    base = np.uint32(0x3A5C7E91)
    sh = {"byte0": (0,), "byte1": (8,), "byte2": (16,), "byte3": (24,), "bytes0and3": (0, 24)}[pattern]
    keys = np.full(n, base, np.uint32)
    for s in sh:
        keys = (keys & ~np.uint32(0xFF << s)) | (rng.integers(0, 256, n).astype(np.uint32) << np.uint32(s))
    vals = np.arange(n, dtype=np.uint32)
    for _ in range(3):                                             # repeated: a race shows as an occasional mis-sort or a time-out
        kb, vb = ctx.buffer(keys), ctx.buffer(vals)
        ctx.sort_pairs(kb, vb, n)
        ek, ev = oracle.sort_pairs(keys, vals, "std")
        assert np.array_equal(ctx.read(kb, np.uint32, n), ek)
        assert np.array_equal(ctx.read(vb, np.uint32, n), ev)
        ctx.delete(kb)
        ctx.delete(vb)
    ctx.close()


def test_caller_stream_handoff_with_refilled_device_buffer():
    """gs4d_set_stream + gs4d_buffer_device_ptr / gs4d_buffer_invalidate + device read-backs, consumed on a torch side stream: run as a
    program of its own (tests/gpu_stream_handoff.py) because torch has to initialise its HIP runtime before libgs4d.so is loaded."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_stream_handoff.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "stream hand-off ok" in r.stdout


def _grid_vertices(width, height, dx, dy):
    """The vertex array Renderer::DrawGrid builds (Renderer.cpp:113-135), including its zero-initialised first half."""
    total = (dx + 1) * 2 + (dy + 1) * 2
    v = [np.zeros((total, 3), np.float32)]
    sx, sy = -width / 2.0, -height / 2.0
    for i in range(dx + 1):
        x = np.float32(sx) + np.float32(width / dx) * np.float32(i)
        v.append(np.array([[x, 0, sy], [x, 0, -sy]], np.float32))
    for i in range(dy + 1):
        z = np.float32(sy) + np.float32(height / dy) * np.float32(i)
        v.append(np.array([[sx, 0, z], [-sx, 0, z]], np.float32))
    return np.concatenate(v)


def test_overlay_lines_then_splats(gs4d, oracle, monkeypatch):
    """What every 4D scene's Render() starts with (Scenes.h:303-310): grid, axes, a path — then the splats over them.  The line
    rasterisation rule is csrc/lines.hip's, held to the reference's line programs run by llvmpipe in test_gpu_gl.py::test_overlay_lines; here GPU == checker."""
    W, H = 1280, 720
    ctx = _ctx(gs4d, W, H, monkeypatch)
    cam = scenes.CAM_TEAPOT
    view, proj = cam_mats(gs4d, cam, W, H)
    vp = (proj.reshape(4, 4).T @ view.reshape(4, 4).T).T.reshape(-1).astype(np.float32)        # column-major P * V
    rec = oracle.golden("linear_first1000")
    grid = _grid_vertices(2000.0, 2000.0, 200, 200)
    axes = [(np.array([[0, 0, 0], [10, 0, 0]], np.float32), (1, 0, 0, 1)), (np.array([[0, 0, 0], [0, 10, 0]], np.float32), (0, 1, 0, 1)),
            (np.array([[0, 0, 0], [0, 0, 10]], np.float32), (0, 0, 1, 1))]
    path = np.stack([np.array([20 * np.cos(a), 5.0 + a, 20 * np.sin(a)], np.float32) for a in np.linspace(0, 6.0, 60)])
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.clear()
    eimg = oracle.clear_image(W, H)
    for verts, col, width, kw in [(grid, (1, 1, 1, 0.15), 1.0, {}), *[(v, c, 3.0, {}) for v, c in axes], (path, (1.0, 0.5, 0.1, 0.8), 5.0, {"strip": True})]:
        ctx.draw_lines(verts, col, width, viewproj=vp, **kw)
        oracle.draw_lines(eimg, verts, col, width, viewproj=vp, **kw)
    seg2d = np.array([[-0.9, -0.8], [0.7, 0.95]], np.float32)
    ctx.draw_lines(seg2d, (0.2, 0.9, 0.3, 0.5), 3.0)
    oracle.draw_lines(eimg, seg2d, (0.2, 0.9, 0.3, 0.5), 3.0)
    lines_only = ctx.read_pixels()
    assert np.abs(eimg - oracle.clear_image(W, H)).max() > 0.3          # lines are there
    diff = np.abs(lines_only.astype(np.float64) - eimg).max(axis=2)
    assert (diff > 1e-6).mean() < 1e-5 and diff.max() <= TOL
    # the splats of the frame blend over the overlays
    n = rec.shape[0]
    db, kb, ib = ctx.buffer(rec), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
    ctx.set_uniforms(time=0.0, min_opacity=0.0, view=view, proj=proj)
    ctx.keygen(db, 0.0, cam[0], kb, ib, n)
    ctx.sort_pairs(kb, ib, n)
    ctx.set_mode(gs4d.MODE_4D_SORTED)
    ctx.bind(1, ib)
    ctx.bind(2, db)
    ctx.draw_instanced(n)
    img = ctx.read_pixels()
    eidx, ekeys = oracle.keygen(rec, 0.0, cam[0])
    _, eperm = oracle.sort_pairs(ekeys.view(np.uint32), eidx, "std")
    oracle.composite(oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H, 0.0, 0.0), eperm, oracle.MODE_4D, W, H, eimg)
    diff = np.abs(img.astype(np.float64) - eimg).max(axis=2)
    assert (diff > 1e-5).mean() < 1e-5 and diff.max() <= TOL
    ctx.close()


@pytest.mark.parametrize("path", ["auto", "ordered"])
def test_fragment_colours_are_clamped_like_the_rop(gs4d, oracle, monkeypatch, path):
    """The reference blends into an RGBA8 window: the GL clamps every fragment's colour and alpha to [0, 1] first.  Colours above 1,
    negative ones, alpha above 1 and uMinOpacity above 1 must therefore give the clamped picture, in the 4D and in the 3D-Full pipeline."""
    n, W, H = 3000, 512, 384
    ctx = _ctx(gs4d, W, H, monkeypatch, **({"GS4D_DRAW_PATH": "ordered"} if path == "ordered" else {}))
    pos4, q, sc, life, fade, vel, rgba = scenes.cube_params_4d(n, seed=81)
    rgba = rgba * np.array([2.5, 1.0, 1.7, 1.6], np.float32) - np.array([0.3, 0.0, 0.2, 0.0], np.float32)
    rec = gs4d.build_records_4d(pos4 * np.array([0.25, 0.25, 0.25, 1.0], np.float32), q, sc * 5.0, life, fade, vel, rgba)
    cam = ((150.0, 100.0, -60.0), (-0.77, -0.57, 0.27))
    view, proj = cam_mats(gs4d, cam, W, H)
    db = ctx.buffer(rec)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.set_mode(gs4d.MODE_4D_DIRECT)
    ctx.bind(1, db)
    for t, mo in ((10.0, 0.0), (10.0, 1.3)):
        ctx.clear()
        ctx.set_uniforms(time=t, min_opacity=mo, view=view, proj=proj)
        ctx.draw_instanced(n)
        eimg = oracle.composite(oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H, t, mo), None, oracle.MODE_4D, W, H, oracle.clear_image(W, H))
        img = ctx.read_pixels()
        assert linf(img, eimg) <= TOL
        assert img.min() >= 0.0 and img.max() <= 1.0 + 1e-6
    # 3D-Full: colour * c is clamped per fragment
    m = 200
    pos, q, sc, rgba = scenes.cube_params(m, seed=82)
    rgba = rgba * np.array([3.0, 0.5, 2.0, 1.5], np.float32)
    verts = np.stack([gs4d.splat3d_mesh(pos[i] * 0.05, q[i], sc[i] * 2.0, rgba[i]) for i in range(m)])
    vb = ctx.buffer(verts)
    ctx.set_mode(gs4d.MODE_3D_FULL)
    ctx.clear()
    ctx.draw_quads(vb, m)
    eimg = oracle.composite(oracle.preprocess(oracle.MODE_3D, verts, view, proj, W, H), None, oracle.MODE_3D, W, H, oracle.clear_image(W, H))
    assert linf(ctx.read_pixels(), eimg) <= TOL
    assert np.abs(eimg - oracle.clear_image(W, H)).max() > 0.05
    ctx.close()


def test_projection_matrix_contract_and_near_plane_clip(gs4d, oracle, monkeypatch):
    """uProj must have glm::perspective's sparsity (anything else is refused, not drawn differently); and the quad's own clip-space z —
    ps.z + uProj[3][2] at w = 1 (Splat4DVertexShaderInstanced.GLSL:147) — is clipped to [-1, 1] by the GL: with a near plane of 30 and a
    far plane of 60 that removes every splat the shader's own NDC test lets through."""
    n, W, H = 20000, 400, 300
    ctx = _ctx(gs4d, W, H, monkeypatch)
    pos, q, sc, rgba = scenes.cube_params(n, seed=83)
    rec = gs4d.build_records_3d(pos * 0.2, q, sc * 3.0, rgba)
    cam = ((0.0, 0.0, 80.0), (0.0, 0.0, -1.0))
    view = gs4d.look_at(cam[0], cam[1])
    db = ctx.buffer(rec)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.set_mode(gs4d.MODE_4D_DIRECT)
    ctx.bind(1, db)
    for znear, zfar in ((0.1, 5000.0), (30.0, 60.0), (0.3, 500.0)):
        proj = gs4d.perspective(scenes.FOV, W, H, znear, zfar)
        ctx.clear()
        ctx.set_uniforms(time=0.0, min_opacity=0.0, view=view, proj=proj)
        ctx.draw_instanced(n)
        eproj = oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H)
        eimg = oracle.composite(eproj, None, oracle.MODE_4D, W, H, oracle.clear_image(W, H))
        assert linf(ctx.read_pixels(), eimg) <= TOL
        assert np.array_equal(ctx.debug_projected(n)[:, 14] != 0, eproj["valid"] != 0)
        if znear == 30.0:
            assert eproj["valid"].sum() == 0                    # ps.z + P[3][2] < -1 for everything in front of the far plane
        else:
            assert eproj["valid"].sum() > n // 2
    bad = gs4d.perspective(scenes.FOV, W, H, 0.1, 100.0).copy()
    bad[12] = 0.25                                              # an off-centre projection: not what the quad set-up assumes
    ctx.set_uniforms(proj=bad)
    with pytest.raises(gs4d.Gs4dError):
        ctx.draw_instanced(n)
    ctx.close()


# ---- key generation executed by the draw that consumes it (gs4d_keygen / gs4d_sort_pairs queued, k_project_count<., true>) ----
def _frames_with_late_reads(ctx, gs4d, rec, cams, W, H, key_mode=None):
    """The reference's frame (Scenes.h:305-345): keygen, sort, draw - nothing looks at the key buffers in between; read them afterwards."""
    n = rec.shape[0]
    db, kb, ib = ctx.buffer(rec), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    out = []
    for k, cam in enumerate(cams):
        view, proj = cam_mats(gs4d, cam, W, H)
        t = 0.25 * k
        ctx.clear()
        ctx.set_uniforms(time=t, min_opacity=0.0, view=view, proj=proj)
        ctx.keygen(db, t, cam[0], kb, ib, n, **({} if key_mode is None else {"key_mode": key_mode}))
        ctx.sort_pairs(kb, ib, n)
        ctx.set_mode(gs4d.MODE_4D_SORTED)
        ctx.bind(1, ib)
        ctx.bind(2, db)
        ctx.draw_instanced(n)
        out.append((ctx.read_pixels(), ctx.read(kb, np.uint32, n), ctx.read(ib, np.uint32, n)))
    st = ctx.stats()
    for b in (db, kb, ib):
        ctx.delete(b)
    return out, st


def _expected_frame(oracle, rec, t, cam, view, proj, W, H, viewz):
    eidx, ekeys = oracle.keygen_viewz(rec, t, view) if viewz else oracle.keygen(rec, t, cam[0])
    skeys, eperm = oracle.sort_pairs(ekeys.view(np.uint32), eidx, "std")
    eproj = oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H, t, 0.0)
    return ekeys.view(np.uint32), skeys, eperm, eproj


@pytest.mark.parametrize("path", ["auto", "ordered"])
@pytest.mark.parametrize("key_mode", ["euclid", "view_z"])
def test_draw_that_generates_its_own_depth_keys(gs4d, oracle, monkeypatch, key_mode, path):
    n, W, H = 120000, 1280, 720
    pos4, q, sc, life, fade, vel, rgba = scenes.cube_params_4d(n, seed=47)
    rec = gs4d.build_records_4d(pos4, q, sc * 2.0, life, fade, vel, rgba)
    c0 = scenes.CAM_CUBE
    far = (tuple(1.15 * np.asarray(c0[0], np.float32)),) + tuple(c0[1:])
    near = (tuple(0.85 * np.asarray(c0[0], np.float32)),) + tuple(c0[1:])
    cams = [c0, far, near, c0, far, near]                            # more frames than lanes: every lane is used again
    km = None if key_mode == "euclid" else gs4d.KEY_VIEW_Z
    res = {}
    for fuse in (1, 0):
        ctx = _ctx(gs4d, W, H, monkeypatch, GS4D_FUSE_KEYGEN=fuse, **({"GS4D_DRAW_PATH": "ordered"} if path == "ordered" else {}))
        res[fuse], st = _frames_with_late_reads(ctx, gs4d, rec, cams, W, H, km)
        ctx.close()
        assert st["fused_keygen_draws"] == (len(cams) if fuse else 0)
        assert st["unordered_draws"] == (len(cams) if path == "auto" else 0)
    for k, cam in enumerate(cams):
        view, proj = cam_mats(gs4d, cam, W, H)
        _, skeys, eperm, eproj = _expected_frame(oracle, rec, 0.25 * k, cam, view, proj, W, H, km is not None)
        eimg = oracle.composite(eproj, eperm, oracle.MODE_4D, W, H, oracle.clear_image(W, H))
        for fuse in (1, 0):
            img, keys_sorted, perm = res[fuse][k]
            assert np.array_equal(perm, eperm), (fuse, k)
            assert np.array_equal(keys_sorted, skeys), (fuse, k)
            assert linf(img, eimg) <= TOL, (fuse, k)
        assert np.array_equal(res[1][k][0], res[0][k][0])            # the same frame bit for bit either way


def test_queued_keygen_is_launched_when_something_else_follows(gs4d, oracle, monkeypatch):
    """The queue is only an optimisation: reads, a draw from another index, a second keygen, an upload all see finished buffers."""
    n, W, H = 20000, 640, 360
    pos4, q, sc, life, fade, vel, rgba = scenes.cube_params_4d(n, seed=48)
    rec = gs4d.build_records_4d(pos4, q, sc * 2.0, life, fade, vel, rgba)
    cam, t = scenes.CAM_CUBE, 3.0
    view, proj = cam_mats(gs4d, cam, W, H)
    ekeys, skeys, eperm, eproj = _expected_frame(oracle, rec, t, cam, view, proj, W, H, False)
    eimg = oracle.composite(eproj, eperm, oracle.MODE_4D, W, H, oracle.clear_image(W, H))
    ctx = _ctx(gs4d, W, H, monkeypatch)
    db, kb, ib, kb2, ib2 = ctx.buffer(rec), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.set_uniforms(time=t, min_opacity=0.0, view=view, proj=proj)
    # (a) keygen alone, then read: unsorted keys, identity index
    ctx.keygen(db, t, cam[0], kb, ib, n)
    assert np.array_equal(ctx.read(kb, np.uint32, n), ekeys)
    assert np.array_equal(ctx.read(ib, np.uint32, n), np.arange(n, dtype=np.uint32))
    # (b) keygen + sort, then a keygen into other buffers: the first pair is finished first
    ctx.keygen(db, t, cam[0], kb, ib, n)
    ctx.sort_pairs(kb, ib, n)
    ctx.keygen(db, t, cam[0], kb2, ib2, n)
    ctx.sort_pairs(kb2, ib2, n)
    # (c) ... and the draw takes its order from the FIRST index: the queued second pair is launched on its own, not by the draw
    ctx.clear()
    ctx.set_mode(gs4d.MODE_4D_SORTED)
    ctx.bind(1, ib)
    ctx.bind(2, db)
    ctx.draw_instanced(n)
    assert linf(ctx.read_pixels(), eimg) <= TOL
    assert np.array_equal(ctx.read(ib, np.uint32, n), eperm)
    assert np.array_equal(ctx.read(ib2, np.uint32, n), eperm)
    assert np.array_equal(ctx.read(kb2, np.uint32, n), skeys)
    assert ctx.stats()["fused_keygen_draws"] == 0
    # (d) keygen + sort, the index uploaded over, draw: the draw blends in the uploaded order (here the reverse: front to back)
    ctx.keygen(db, t, cam[0], kb, ib, n)
    ctx.sort_pairs(kb, ib, n)
    rev = np.ascontiguousarray(eperm[::-1])
    ctx.subdata(ib, rev)
    ctx.clear()
    ctx.draw_instanced(n)
    assert linf(ctx.read_pixels(), oracle.composite(eproj, rev, oracle.MODE_4D, W, H, oracle.clear_image(W, H))) <= TOL
    assert ctx.stats()["fused_keygen_draws"] == 0
    ctx.close()


# ---- glBlendFunc other than the default (Application.cpp:150 takes both factors from the blend menu, DebugMenus.h:41-59) ----
BLENDS = [("ONE", "ONE"), ("SRC_ALPHA", "ONE"), ("ONE", "ZERO"), ("ZERO", "ONE"), ("DST_COLOR", "ZERO"), ("ONE_MINUS_DST_ALPHA", "DST_ALPHA"),
          ("SRC_COLOR", "ONE_MINUS_SRC_COLOR"), ("ONE_MINUS_DST_COLOR", "ONE_MINUS_SRC_ALPHA"), ("ONE_MINUS_CONSTANT_ALPHA", "CONSTANT_COLOR"),
          ("SRC_ALPHA", "ONE_MINUS_SRC_ALPHA")]


@pytest.mark.parametrize("sf,df", BLENDS)
def test_blend_functions_of_the_menu(gs4d, oracle, monkeypatch, sf, df):
    """Lines, then 4D splats in sorted order, then 3D-Full quads (premultiplied colour), all with the selected function."""
    n, W, H = 6000, 480, 270
    blend = (getattr(gs4d, sf), getattr(gs4d, df))
    pos4, q, sc, life, fade, vel, rgba = scenes.cube_params_4d(n, seed=71)
    rec = gs4d.build_records_4d(pos4, q, sc * 6.0, life, fade, vel, rgba)
    cam, t = scenes.CAM_CUBE, 2.0
    view, proj = cam_mats(gs4d, cam, W, H)
    grid = np.array([[np.cos(a), np.sin(a)] for k in range(9) for a in (k * 0.35, k * 0.35 + np.pi)], np.float32) * 0.9     # 2D (NDC) segments crossing in one pixel
    col = np.array([0.9, 0.4, 0.2, 0.6], np.float32)
    ctx = _ctx(gs4d, W, H, monkeypatch)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.set_blend(*blend)
    ctx.clear()
    ctx.draw_lines(grid, col, width=2.0)
    db, kb, ib = ctx.buffer(rec), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
    ctx.set_uniforms(time=t, min_opacity=0.0, view=view, proj=proj)
    ctx.keygen(db, t, cam[0], kb, ib, n)
    ctx.sort_pairs(kb, ib, n)
    ctx.set_mode(gs4d.MODE_4D_SORTED)
    ctx.bind(1, ib)
    ctx.bind(2, db)
    ctx.draw_instanced(n)
    img1 = ctx.read_pixels()
    st = ctx.stats()
    # a second draw on top of the first, in record order (4D direct): the blend starts from what the first left
    ctx.set_mode(gs4d.MODE_4D_DIRECT)
    ctx.bind(1, db)
    ctx.draw_instanced(n // 2)
    img2 = ctx.read_pixels()
    ctx.close()
    over = blend == oracle.BLEND_OVER
    assert (st["unordered_draws"] > 0) == over                    # any other function is blended in draw order from instance-ordered lists
    _, skeys, eperm, eproj = _expected_frame(oracle, rec, t, cam, view, proj, W, H, False)
    e = oracle.clear_image(W, H)
    oracle.draw_lines(e, grid, col, width=2.0, blend=blend)
    oracle.composite(eproj, eperm, oracle.MODE_4D, W, H, e, blend=blend)
    assert linf(img1, e) <= TOL
    oracle.composite(eproj, np.arange(n // 2, dtype=np.uint32), oracle.MODE_4D, W, H, e, blend=blend)
    assert linf(img2, e) <= TOL
    if (sf, df) in (("ZERO", "ONE"), ("ONE_MINUS_DST_ALPHA", "DST_ALPHA")):
        assert linf(img2, oracle.clear_image(W, H)) == 0.0        # nothing drawn changes anything (the second pair: destination alpha is 1 from the clear on)
    else:
        assert linf(img1, oracle.clear_image(W, H)) > 0.05


def test_blend_function_with_premultiplied_quads(gs4d, oracle, monkeypatch):
    """3D-Full fragments carry colour * c (Splat3DFragShaderFull.GLSL:22): the premultiplied-alpha "over" (ONE, ONE_MINUS_SRC_ALPHA)."""
    W, H, n = 400, 400, 64
    cam = ((0.0, 0.0, 10.0), (0.0, 0.0, -1.0))
    view, proj = cam_mats(gs4d, cam, W, H)
    pos, q, scale, rgba = scenes.cube_params(n, seed=72)
    verts = np.zeros((n, 4, 18), np.float32)                      # {corner2, pos3, col4, sig9} x 4 vertices  (Geometry.h:37-42, Splat.h:433-447)
    corners = np.array([[0.5, 0.5], [0.5, -0.5], [-0.5, -0.5], [-0.5, 0.5]], np.float32)
    for i in range(n):
        verts[i, :, 0:2] = corners
        verts[i, :, 2:5] = pos[i] * 0.02
        verts[i, :, 5:9] = rgba[i]
        verts[i, :, 9:18] = gs4d.splat3d_cov(q[i], scale[i] * 0.6)
    blend = (gs4d.ONE, gs4d.ONE_MINUS_SRC_ALPHA)
    ctx = _ctx(gs4d, W, H, monkeypatch)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.set_blend(*blend)
    ctx.clear()
    ctx.set_uniforms(view=view, proj=proj)
    ctx.set_mode(gs4d.MODE_3D_FULL)
    vb = ctx.buffer(verts)
    ctx.draw_quads(vb, n)
    img = ctx.read_pixels()
    ctx.close()
    eproj = oracle.preprocess(oracle.MODE_3D, verts, view, proj, W, H)
    e = oracle.composite(eproj, None, oracle.MODE_3D, W, H, oracle.clear_image(W, H), blend=blend)
    assert linf(img, e) <= TOL
    assert linf(e, oracle.composite(eproj, None, oracle.MODE_3D, W, H, oracle.clear_image(W, H))) > 0.01       # and it is not the default function's image
