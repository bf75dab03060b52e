"""GPU parity: projection + ordered compositing, through the C ABI, against the CPU checker.

Bars (BASELINE.json north_star):
  * projected quad set-up (centre + the two affine rows that decide pixel coverage): bit-exact;
  * framebuffer: per-pixel L-infinity <= 1e-4 on every channel (TOL below).
Reference: Splat4DVertexShaderInstanced.GLSL:81-150, Splat4DFragShader.GLSL:16-31, Application.cpp:125,150-154.
"""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(autouse=True, params=["auto", "ordered"])
def draw_path(request, monkeypatch):
    """Every test of this module runs twice: with the library choosing how a draw builds its tile lists (unordered lists ordered in
    the compositor wherever the blend order is known without reading the sort index) and with instance-ordered lists forced."""
    if request.param == "ordered":
        monkeypatch.setenv("GS4D_DRAW_PATH", "ordered")
    else:
        monkeypatch.delenv("GS4D_DRAW_PATH", raising=False)
    return request.param


def cam_mats(gs4d, cam, W, H):
    view = gs4d.look_at(cam[0], cam[1])
    proj = gs4d.perspective(scenes.FOV, W, H, scenes.ZNEAR, scenes.ZFAR)
    return view, proj


def gpu_frame(ctx, gs4d, rec, cam, t=0.0, min_opacity=0.0, sort=True, order=None):
    """Replays Scenes.h:312-339 (key loop -> sort -> uniforms -> bind -> Draw) through the C ABI."""
    n = rec.shape[0]
    W, H = ctx.width, ctx.height
    view, proj = cam_mats(gs4d, cam, W, H)
    db = ctx.buffer(rec)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.clear()
    ctx.set_uniforms(time=t, min_opacity=min_opacity, view=view, proj=proj)
    bufs = [db]
    if sort or order is not None:
        kb, ib = ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
        bufs += [kb, ib]
        if order is not None:
            ctx.subdata(ib, np.ascontiguousarray(order, np.uint32))
        else:
            ctx.keygen(db, t, cam[0], kb, ib, n)
            ctx.sort_pairs(kb, ib, n)
        ctx.set_mode(gs4d.MODE_4D_SORTED)
        ctx.bind(1, ib)
        ctx.bind(2, db)
        ninst = n if order is None else len(order)
    else:
        ctx.set_mode(gs4d.MODE_4D_DIRECT)
        ctx.bind(1, db)
        ninst = n
    ctx.draw_instanced(ninst)
    img = ctx.read_pixels()
    projd = ctx.debug_projected(n)
    stats = ctx.stats()
    for b in bufs:
        ctx.delete(b)
    return img, projd, stats, (view, proj)


def check_projected(oracle, projd, eproj):
    """Everything that decides coverage is bit-exact; alpha carries an exp() and is held to 1e-6 relative."""
    valid = eproj["valid"] != 0
    assert np.array_equal(projd[:, 14] != 0, valid)
    for col, name in enumerate(["cx", "cy", "a0x", "a0y", "a1x", "a1y"]):
        assert np.array_equal(projd[valid, col].view(np.uint32), eproj[name][valid].view(np.uint32)), name
    np.testing.assert_allclose(projd[valid, 6], eproj["alpha"][valid], rtol=2e-6, atol=1e-12)
    for col, name in ((7, "r"), (8, "g"), (9, "b")):
        assert np.array_equal(projd[valid, col], eproj[name][valid])
    assert np.array_equal(projd[valid, 12].view(np.uint32), eproj["hx"][valid].view(np.uint32))


def linf(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))))


@pytest.fixture
def ctx1080(gs4d, draw_path):
    c = gs4d.Context(1920, 1080)
    yield c
    c.close()


def test_teapot_first1000_reference_records(ctx1080, gs4d, oracle):
    """Config 1': the first 1000 records of the reference-generated LinearMotion SSBO, 1080p, teapot camera, sort on."""
    rec = oracle.golden("linear_first1000")
    img, projd, stats, (view, proj) = gpu_frame(ctx1080, gs4d, rec, scenes.CAM_TEAPOT, t=0.0)
    eimg, eperm, _ = oracle.render_4d(rec, True, 0.0, 0.0, scenes.CAM_TEAPOT[0], view, proj, 1920, 1080)
    eproj = oracle.preprocess(oracle.MODE_4D, rec, view, proj, 1920, 1080, 0.0, 0.0)
    check_projected(oracle, projd, eproj)
    assert eproj["valid"].sum() > 900
    assert linf(img, eimg) <= TOL
    assert np.abs(eimg - np.array(gs4d.CLEAR_COLOR, np.float32)).max() > 0.2      # the teapot is actually visible
    assert stats["reruns"] == 0


@pytest.mark.parametrize("n", [1, 777, 20000, 1000000, 10000000])
def test_cube_1080p(ctx1080, gs4d, oracle, n):
    """Configs 2 and 3 (n = 1e6, 1e7) and smaller cuts: random 3D splats in the 400^3 cube, screenshot camera, 1080p."""
    pos, q, scale, rgba = scenes.cube_params(n)
    rec = gs4d.build_records_3d(pos, q, scale, rgba)
    img, projd, stats, (view, proj) = gpu_frame(ctx1080, gs4d, rec, scenes.CAM_CUBE)
    eimg, _, _ = oracle.render_4d(rec, True, 0.0, 0.0, scenes.CAM_CUBE[0], view, proj, 1920, 1080)
    eproj = oracle.preprocess(oracle.MODE_4D, rec, view, proj, 1920, 1080)
    check_projected(oracle, projd, eproj)
    assert linf(img, eimg) <= TOL
    if n >= 20000:
        assert eproj["valid"].mean() > 0.99          # the whole cube is inside the frustum of that camera


def test_time_sweep_4d(gs4d, oracle):
    """Config 4 in miniature: 4D splats, several times, uMinOpacity > 0 on one of them."""
    n, W, H = 60000, 960, 540
    ctx = gs4d.Context(W, H)
    pos4, q, scale, life, fade, vel, rgba = scenes.cube_params_4d(n)
    rec = gs4d.build_records_4d(pos4, q, scale * 4.0, life, fade, vel, rgba)
    for t, mo in ((0.0, 0.0), (12.5, 0.0), (25.0, 0.05), (50.0, 0.0)):
        img, projd, _, (view, proj) = gpu_frame(ctx, gs4d, rec, scenes.CAM_CUBE, t=t, min_opacity=mo)
        eimg, _, _ = oracle.render_4d(rec, True, t, mo, scenes.CAM_CUBE[0], view, proj, W, H)
        check_projected(oracle, projd, oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H, t, mo))
        assert linf(img, eimg) <= TOL
    ctx.close()


def test_nonlinear_teapot_4k(gs4d, oracle):
    """Config 5 in miniature: the NonLinearMotion scene (circular path, Scenes.h:517-545) at its reference size (335 248 splats)
    on a 3840x2160 frame, camera (0,60,60) -> (0,-1,-1), mid-sweep.  4K has 129 600 tiles: three tile-sort passes, long lists."""
    W, H = 3840, 2160
    ctx = gs4d.Context(W, H)
    rec = gs4d.scene_nonlinear(oracle.golden("teapot_vdata"))
    t = 46.0
    img, projd, stats, (view, proj) = gpu_frame(ctx, gs4d, rec, scenes.CAM_NONLINEAR, t=t)
    eimg, _, _ = oracle.render_4d(rec, True, t, 0.0, scenes.CAM_NONLINEAR[0], view, proj, W, H, nthreads=16)
    check_projected(oracle, projd, oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H, t, 0.0))
    assert linf(img, eimg) <= TOL
    assert np.abs(eimg - np.array(gs4d.CLEAR_COLOR, np.float32)).max() > 0.3
    assert stats["tiles"] == 480 * 270
    ctx.close()


@pytest.mark.parametrize("name,cam,t", [
    ("rotation", ((0.0, 60.0, 30.0), (0.0, -1.0, -0.5)), 20.25),      # cameras: Scenes.h:748-749, 1003-1004, 1941-1942, 2192-2193
    ("combined", ((50.0, 90.0, 90.0), (0.0, -1.0, -1.0)), 31.5),
    ("broken", ((0.0, 60.0, 60.0), (0.0, -1.0, -1.0)), 19.75),        # just before the jump back (y = fmod(1 + dt, 20))
    ("square", ((0.0, 60.0, 60.0), (0.0, -1.0, -1.0)), 23.0),         # at a corner of the square
])
def test_motion_scenes_render(gs4d, oracle, name, cam, t):
    """The other four teapot scenes of the reference at their full size and their own cameras, one mid-sweep frame each, 1280x720,
    sort on (the reference leaves m_DoSort off by default in some of them; the path is the same)."""
    W, H = 1280, 720
    ctx = gs4d.Context(W, H)
    rec = getattr(gs4d, "scene_" + name)(oracle.golden("teapot_vdata"))
    img, projd, _, (view, proj) = gpu_frame(ctx, gs4d, rec, cam, t=t)
    eimg, _, _ = oracle.render_4d(rec, True, t, 0.0, cam[0], view, proj, W, H, nthreads=16)
    check_projected(oracle, projd, oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H, t, 0.0))
    assert linf(img, eimg) <= TOL
    assert np.abs(eimg - np.array(gs4d.CLEAR_COLOR, np.float32)).max() > 0.3
    ctx.close()


def test_object_display_from_sd_file(gs4d, oracle):
    """ObjectDisplay (Scenes.h:2457-2500): records loaded from a .sd file, its camera (0,2,8) -> (0,-0.10,-1.4), sort on."""
    import os
    W, H = 1024, 576
    ctx = gs4d.Context(W, H)
    rec = gs4d.parse_sd(os.path.join(os.path.dirname(__file__), "golden", "synthetic.sd"), object_scale=1.0)
    cam = ((0.0, 2.0, 8.0), (0.0, -0.10, -1.4))
    img, projd, _, (view, proj) = gpu_frame(ctx, gs4d, rec, cam, t=0.0)
    eimg, _, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
    check_projected(oracle, projd, oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H, 0.0, 0.0))
    assert linf(img, eimg) <= TOL
    assert np.abs(eimg - np.array(gs4d.CLEAR_COLOR, np.float32)).max() > 0.05
    ctx.close()


def test_config5_full_size_10m_4d_splats_4k(gs4d, oracle):
    """BASELINE.json configs[4] at its full size on one GPU: 10^7 4D splats from the NonLinearMotion generator (2745 time steps = four
    laps of the circle), 3840x2160, t = 1372, sort on.  5.4e8 tile-list entries (the first frame overflows the
    speculative capacity and is re-run).  Per-pixel diff against the CPU checker (~15 s on 64 host threads) and the permutation."""
    import os
    n, W, H, steps = 10_000_000, 3840, 2160, 2745
    rec = gs4d.scene_nonlinear(oracle.golden("teapot_vdata"), steps=steps, angle_multiplier=360.0 / steps * 4.0, max_records=n)
    assert rec.shape == (n, 24)
    cam, t = scenes.CAM_NONLINEAR, 1372.0
    view, proj = cam_mats(gs4d, cam, W, H)
    ctx = gs4d.Context(W, H)
    data, keys, idx = ctx.buffer(rec), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.set_mode(gs4d.MODE_4D_SORTED)
    ctx.bind(2, data)
    ctx.set_uniforms(time=t, min_opacity=0.0, view=view, proj=proj)
    ctx.clear()
    ctx.keygen(data, t, cam[0], keys, idx, n)
    ctx.sort_pairs(keys, idx, n)
    ctx.bind(1, idx)
    ctx.draw_instanced(n)
    img = ctx.read_pixels()
    stats = ctx.stats()
    perm = ctx.read(idx, np.uint32, n)
    ctx.close()
    eidx, ekeys = oracle.keygen(rec, t, cam[0])
    _, eperm = oracle.sort_pairs(ekeys.view(np.uint32), eidx, "lsd")
    assert np.array_equal(perm, eperm)
    eimg, _, _ = oracle.render_4d(rec, True, t, 0.0, cam[0], view, proj, W, H, nthreads=min(os.cpu_count() or 8, 64))
    assert linf(img, eimg) <= TOL
    assert np.abs(eimg - np.array(gs4d.CLEAR_COLOR, np.float32)).max() > 0.3
    assert stats["entries"] > 100_000_000 and stats["reruns"] >= 1


@pytest.mark.parametrize("W,H", [(8, 8), (16, 8), (24, 40), (250, 130)])
def test_tiny_and_odd_framebuffers(gs4d, oracle, W, H):
    """One tile, two tiles, a few tiles: every entry of the tile lists carries the same (or nearly the same) tile id, so the tile sort's
    digits are constant and its passes are skipped on the device; the per-tile ranges must still come out."""
    ctx = gs4d.Context(W, H)
    pos, q, scale, rgba = scenes.cube_params(3000, seed=5)
    rec = gs4d.build_records_3d(pos * 0.05, q, scale * 6.0, rgba)
    cam = ((30.0, 20.0, -25.0), (-0.66, -0.44, 0.55))
    for _ in range(2):                     # twice: the ranges table must be back in its resting state (all zero) after a frame
        img, projd, _, (view, proj) = gpu_frame(ctx, gs4d, rec, cam)
        eimg, _, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
        assert linf(img, eimg) <= TOL
    assert np.abs(eimg - np.array(gs4d.CLEAR_COLOR, np.float32)).max() > 0.05
    ctx.close()


def test_time_sweep_1e6_4d_splats(ctx1080, gs4d, oracle):
    """Config 4, one frame of the sweep at full size: 1e6 4D splats (velocity, lifetime, mu_t in [0,50]) at t = 50*100/255."""
    n = 1000000
    pos4, q, scale, life, fade, vel, rgba = scenes.cube_params_4d(n)
    rec = gs4d.build_records_4d(pos4, q, scale, life, fade, vel, rgba)
    t = 50.0 * 100 / 255
    img, projd, _, (view, proj) = gpu_frame(ctx1080, gs4d, rec, scenes.CAM_CUBE, t=t)
    eimg, _, _ = oracle.render_4d(rec, True, t, 0.0, scenes.CAM_CUBE[0], view, proj, 1920, 1080, nthreads=16)
    check_projected(oracle, projd, oracle.preprocess(oracle.MODE_4D, rec, view, proj, 1920, 1080, t, 0.0))
    assert linf(img, eimg) <= TOL


def test_unsorted_and_arbitrary_order(gs4d, oracle):
    """m_do_sort == false (the reference's default): instances blend in index order; and any caller-provided permutation or
    subset is honoured (sortidx[] is just data, Splat4DVertexShaderInstanced.GLSL:9)."""
    n, W, H = 5000, 640, 360
    ctx = gs4d.Context(W, H)
    pos, q, scale, rgba = scenes.cube_params(n, seed=7)
    rec = gs4d.build_records_3d(pos * 0.25, q, scale * 6.0, rgba)     # denser + larger: order matters
    cam = ((120.0, 80.0, -40.0), (-0.77, -0.57, 0.27))
    view, proj = cam_mats(gs4d, cam, W, H)
    eproj = oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H)
    # identity order through the "Mod" pipeline
    img, _, _, _ = gpu_frame(ctx, gs4d, rec, cam, sort=False)
    eimg = oracle.composite(eproj, None, oracle.MODE_4D, W, H, oracle.clear_image(W, H))
    assert linf(img, eimg) <= TOL
    # a shuffled permutation, and a subset with repeats
    rng = np.random.default_rng(3)
    for order in (rng.permutation(n).astype(np.uint32), rng.integers(0, n, 1234).astype(np.uint32)):
        img, _, _, _ = gpu_frame(ctx, gs4d, rec, cam, order=order)
        eimg2 = oracle.composite(eproj, order, oracle.MODE_4D, W, H, oracle.clear_image(W, H))
        assert linf(img, eimg2) <= TOL
    assert linf(eimg, eimg2) > 1e-2        # the order does change the picture
    ctx.close()


def test_large_footprints_overflow_rerun_and_odd_size(gs4d, oracle):
    """Close-up splats covering thousands of tiles each: the tile lists outgrow their first capacity and the draw is re-run;
    the framebuffer is not a multiple of the 8-pixel tile."""
    n, W, H = 300, 1001, 701
    ctx = gs4d.Context(W, H)
    pos, q, scale, rgba = scenes.cube_params(n, seed=11)
    rgba[:, 3] *= 0.35
    rec = gs4d.build_records_3d(pos * 0.02, q, scale * 8.0, rgba)
    cam = ((0.0, 0.0, 30.0), (0.0, 0.0, -1.0))
    img, projd, stats, (view, proj) = gpu_frame(ctx, gs4d, rec, cam)
    eimg, _, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
    assert stats["entries"] > 2 * n + 65536 and stats["reruns"] >= 1
    assert linf(img, eimg) <= TOL
    # second frame: capacity is now sufficient, no re-run
    img2, _, stats2, _ = gpu_frame(ctx, gs4d, rec, cam)
    assert stats2["reruns"] == stats["reruns"]
    assert np.array_equal(img, img2)            # deterministic
    ctx.close()


def test_camera_inside_cloud_culls(gs4d, oracle):
    """Splats behind / beside the camera exercise the NDC cull (Splat4DVertexShaderInstanced.GLSL:108-115)."""
    n, W, H = 40000, 800, 600
    ctx = gs4d.Context(W, H)
    pos, q, scale, rgba = scenes.cube_params(n, seed=5)
    rec = gs4d.build_records_3d(pos, q, scale * 3.0, rgba)
    cam = ((10.0, -20.0, 5.0), (0.3, 0.1, -1.0))
    img, projd, _, (view, proj) = gpu_frame(ctx, gs4d, rec, cam)
    eproj = oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H)
    check_projected(oracle, projd, eproj)
    assert 0.02 < eproj["valid"].mean() < 0.6
    eimg, _, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
    assert linf(img, eimg) <= TOL
    ctx.close()


def test_two_draws_accumulate_and_clear_between_frames(gs4d, oracle):
    """Two Draw calls without a Clear blend into the same framebuffer (the reference draws overlays + splats per frame)."""
    nA, nB, W, H = 3000, 2000, 512, 512
    ctx = gs4d.Context(W, H)
    cam = ((150.0, 100.0, -60.0), (-0.77, -0.57, 0.27))
    view, proj = cam_mats(gs4d, cam, W, H)
    recs = []
    for n, seed in ((nA, 21), (nB, 22)):
        pos, q, scale, rgba = scenes.cube_params(n, seed=seed)
        recs.append(gs4d.build_records_3d(pos * 0.3, q, scale * 5.0, rgba))
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.clear()
    ctx.set_uniforms(time=0.0, min_opacity=0.0, view=view, proj=proj)
    ctx.set_mode(gs4d.MODE_4D_DIRECT)
    eimg = oracle.clear_image(W, H)
    bufs = []
    for rec in recs:
        b = ctx.buffer(rec)
        bufs.append(b)
        ctx.bind(1, b)
        ctx.draw_instanced(rec.shape[0])
        oracle.composite(oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H), None, oracle.MODE_4D, W, H, eimg)
    assert linf(ctx.read_pixels(), eimg) <= TOL
    # a Clear with nothing drawn afterwards reads back as the clear colour
    ctx.clear()
    assert np.array_equal(ctx.read_pixels(), oracle.clear_image(W, H))
    for b in bufs:
        ctx.delete(b)
    ctx.close()


def test_mode_3d_full_and_2d(gs4d, oracle):
    """Splat3DVertexShaderFull/FragShaderFull (colour premultiplied by c) and Splat2DVSI/2DFragShader."""
    W, H = 800, 800
    ctx = gs4d.Context(W, H)
    cam = ((0.0, 0.0, 10.0), (0.0, 0.0, -1.0))
    view, proj = cam_mats(gs4d, cam, W, H)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.set_uniforms(view=view, proj=proj)
    # --- 3D: 4 vertices x 18 floats per splat {corner2, pos3, col4, sig9}  (Geometry.h:37-42, Splat.h:433-447)
    n = 64
    pos, q, scale, rgba = scenes.cube_params(n, seed=31)
    verts = np.zeros((n, 4, 18), np.float32)
    corners = np.array([[0.5, 0.5], [0.5, -0.5], [-0.5, -0.5], [-0.5, 0.5]], np.float32)
    for i in range(n):
        cov = gs4d.splat3d_cov(q[i], scale[i] * 0.6)
        verts[i, :, 0:2] = corners
        verts[i, :, 2:5] = pos[i] * 0.02
        verts[i, :, 5:9] = rgba[i]
        verts[i, :, 9:18] = cov
    vb = ctx.buffer(verts)
    ctx.set_mode(gs4d.MODE_3D_FULL)
    ctx.clear()
    ctx.draw_quads(vb, n)
    img = ctx.read_pixels()
    eproj = oracle.preprocess(oracle.MODE_3D, verts, view, proj, W, H)
    eimg = oracle.composite(eproj, None, oracle.MODE_3D, W, H, oracle.clear_image(W, H))
    assert eproj["valid"].sum() == n
    assert linf(img, eimg) <= TOL
    with pytest.raises(gs4d.Gs4dError):
        ctx.draw_instanced(n)                      # wrong entry point for this pipeline
    # --- 2D: 12 floats per record {pos4, col4, mat2}  (Scenes.h:1447-1452)
    m = 20
    rng = np.random.default_rng(2)
    rec2 = np.zeros((m, 12), np.float32)
    rec2[:, 0:2] = rng.uniform(-2.0, 2.0, (m, 2))
    rec2[:, 4:8] = rng.uniform(0.2, 1.0, (m, 4))
    for i in range(m):
        ang, s0, s1 = rng.uniform(0, np.pi), rng.uniform(0.05, 0.4), rng.uniform(0.05, 0.4)
        R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        S = R @ np.diag([s0 * s0, s1 * s1]) @ R.T
        rec2[i, 8:12] = [S[0, 0], S[1, 0], S[0, 1], S[1, 1]]
    rec2[0, 8:12] = [0.09, 0.0, 0.0, 0.04]        # zero off-diagonal: the guarded branch (Splat2DVSI.GLSL:49-52)
    b2 = ctx.buffer(rec2)
    ctx.set_mode(gs4d.MODE_2D)
    ctx.bind(1, b2)
    ctx.clear()
    ctx.draw_instanced(m)
    img2 = ctx.read_pixels()
    eproj2 = oracle.preprocess(oracle.MODE_2D, rec2, view, proj, W, H)
    eimg2 = oracle.composite(eproj2, None, oracle.MODE_2D, W, H, oracle.clear_image(W, H))
    assert eproj2["valid"].sum() == m
    assert linf(img2, eimg2) <= TOL
    assert linf(eimg2, oracle.clear_image(W, H)) > 0.05
    ctx.delete(vb)
    ctx.delete(b2)
    ctx.close()


def test_errors_and_state(gs4d):
    ctx = gs4d.Context(64, 64)
    with pytest.raises(gs4d.Gs4dError):
        ctx.set_blend(gs4d.SRC_ALPHA, 0x0308)                    # GL_SRC_ALPHA_SATURATE: not a factor of the reference's menu -> GL_INVALID_ENUM
    ctx.set_blend(gs4d.SRC_ALPHA, gs4d.SRC_ALPHA)                # any pair of the menu's factors is accepted (test_gpu_paths covers the blending)
    ctx.set_blend(gs4d.SRC_ALPHA, gs4d.ONE_MINUS_SRC_ALPHA)
    with pytest.raises(gs4d.Gs4dError):
        ctx.draw_instanced(10)                                    # nothing bound
    b = ctx.buffer(np.zeros((4, 24), np.float32))
    with pytest.raises(gs4d.Gs4dError):
        ctx.subdata(b, np.zeros(100, np.float32), offset=96)      # past the end: GL_INVALID_VALUE
    with pytest.raises(gs4d.Gs4dError):
        ctx.bind(9, b)
    ctx.set_mode(gs4d.MODE_4D_SORTED)
    ctx.bind(2, b)
    with pytest.raises(gs4d.Gs4dError):
        ctx.draw_instanced(4)                                     # sortidx buffer missing
    ctx.resize(128, 32)
    assert ctx.read_pixels().shape == (32, 128, 4)
    ctx.close()
