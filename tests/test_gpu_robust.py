"""GPU: what happens when something goes wrong, or late — contract violations, draws that are validated after the fact, draws that are
never observed.  Through the C ABI, against the CPU checker (image L-infinity <= 1e-4).

* a key buffer rewritten under a queued sort without gs4d_buffer_invalidate: GS4D_E_DEVICE, not a memory fault (sort.hip's scatter is bounded);
* a draw whose tile lists overflow is re-run when somebody looks at the image — also when the context has moved on to another frame lane;
* a draw whose unordered lists cannot be ordered in the compositor is re-run on the ordered path WITHOUT the caller's sort index, which
  the application (one key / index pair, the reference's layout: Scenes.h m_key_buf / m_values_buf) has meanwhile overwritten;
* frames that abort on the device and are cleared away unobserved are counted (gs4d_get_stats) and still teach the context its capacities;
* glClearColor between a draw and its re-run does not repaint the earlier frame.
"""
import ctypes as C

import numpy as np
import pytest

import scenes
from test_gpu_render import cam_mats, linf, TOL

pytestmark = pytest.mark.gpu


def _hip():
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    hip.hipDeviceSynchronize.restype = C.c_int
    return hip


def test_key_buffer_rewritten_under_a_queued_sort_is_an_error_not_a_fault(gs4d, monkeypatch):
    """The caller keeps a device pointer to its key buffer, lets gs4d_keygen fill it (the kernel also accumulates the digit histograms the
    sort will use), overwrites the keys through the pointer WITHOUT gs4d_buffer_invalidate and then sorts: histograms and keys no longer
    describe the same array.  The sort must stay inside its buffers and the context must report the frame as failed."""
    monkeypatch.setenv("GS4D_FUSE_KEYGEN", "0")              # key generation is launched at once (not folded into a later draw)
    monkeypatch.delenv("GS4D_DRAW_PATH", raising=False)
    n, W, H = 300000, 640, 360
    ctx = gs4d.Context(W, H)
    pos, q, sc, rgba = scenes.cube_params(n, seed=91)
    rec = gs4d.build_records_3d(pos, q, sc, rgba)
    cam = scenes.CAM_CUBE
    data, kb, ib = ctx.buffer(rec), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
    kptr, nbytes = ctx.device_ptr(kb)                         # taken BEFORE the keys are generated; never announced again
    assert nbytes == 4 * n
    ctx.keygen(data, 0.0, cam[0], kb, ib, n)
    hip = _hip()
    assert hip.hipDeviceSynchronize() == 0
    junk = np.random.default_rng(5).integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)     # digits unlike the real keys' in every byte
    assert hip.hipMemcpy(C.c_void_p(kptr), junk.ctypes.data_as(C.c_void_p), 4 * n, 1) == 0
    ctx.sort_pairs(kb, ib, n)                                 # uses the histograms of the keys that are no longer there
    with pytest.raises(gs4d.Gs4dError, match="device-side check failed"):
        ctx.finish()
    ctx.close()
    # the device is fine: a fresh context renders
    ctx = gs4d.Context(W, H)
    ctx.clear()
    assert ctx.read_pixels().shape == (H, W, 4)
    ctx.close()


def _big_splats(gs4d, n=300, seed=11):
    pos, q, scale, rgba = scenes.cube_params(n, seed=seed)
    rgba[:, 3] *= 0.35
    return gs4d.build_records_3d(pos * 0.02, q, scale * 8.0, rgba)       # close-up: each covers thousands of tiles


@pytest.mark.parametrize("path", ["auto", "ordered"])
def test_overflow_rerun_seen_from_another_lane(gs4d, oracle, monkeypatch, path):
    """First frame of a context: the tile lists outgrow their speculative capacity.  Before anybody looks at the image the application
    queues the next frame's key generation — the context moves to the next lane.  read_pixels then re-runs the draw on ITS lane and
    must read the pixels behind that re-run, not behind the lane's old tail event."""
    if path == "ordered":
        monkeypatch.setenv("GS4D_DRAW_PATH", "ordered")
    else:
        monkeypatch.delenv("GS4D_DRAW_PATH", raising=False)
    W, H = 1001, 701
    cam = ((0.0, 0.0, 30.0), (0.0, 0.0, -1.0))
    view, proj = cam_mats(gs4d, cam, W, H)
    rec = _big_splats(gs4d)
    n = rec.shape[0]
    ctx = gs4d.Context(W, H)
    assert ctx.stats()["lanes"] >= 2
    data = ctx.buffer(rec)
    a = (ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n))
    b = (ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n))
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.set_mode(gs4d.MODE_4D_SORTED)
    ctx.bind(2, data)
    ctx.clear()
    ctx.set_uniforms(time=0.0, min_opacity=0.0, view=view, proj=proj)
    ctx.keygen(data, 0.0, cam[0], *a, n)
    ctx.sort_pairs(*a, n)
    ctx.bind(1, a[1])
    ctx.draw_instanced(n)
    ctx.keygen(data, 0.0, cam[0], *b, n)                       # the next frame starts: another lane is current from here on
    ctx.sort_pairs(*b, n)
    img = ctx.read_pixels()                                    # still the first image: validated, re-run, read
    eimg, _, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
    st = ctx.stats()
    assert st["reruns"] >= 1 and st["entries"] > 2 * n + 65536
    assert linf(img, eimg) <= TOL
    ctx.close()


@pytest.mark.parametrize("fuse", ["1", "0"])
def test_fallback_rerun_does_not_need_the_callers_sort_index(gs4d, oracle, monkeypatch, fuse):
    """ONE key / index buffer pair (the reference's layout).  Frame A: thousands of splats on one spot — a tile list longer than the
    compositing wave can order: the unordered draw aborts and has to be re-run on the instance-ordered path.  Before anybody observes it the
    application generates and sorts frame B's keys INTO THE SAME BUFFERS and draws B on top (no clear).  A's re-run must blend in A's
    order, not in whatever the index buffer holds by then."""
    monkeypatch.delenv("GS4D_DRAW_PATH", raising=False)
    monkeypatch.setenv("GS4D_FUSE_KEYGEN", fuse)             # "0": frame B's key generation and sort are LAUNCHED (they overwrite the buffers on the device) before frame A is validated
    W, H = 256, 256
    cam = ((0.0, 0.0, 60.0), (0.0, 0.0, -1.0))
    view, proj = cam_mats(gs4d, cam, W, H)
    n = 3000
    pos, q, sc, rgba = scenes.cube_params(n, seed=3050)
    rgba[:, 3] *= 0.05
    pos[:, 0:2] = 0.0
    recA = gs4d.build_records_3d(pos * 0.0, q, sc * 1.5, rgba)              # one spot, one depth: 3000 entries on a tile (more than the compositing wave holds), equal keys
    pos4, q4, sc4, life, fade, vel, rgba4 = scenes.cube_params_4d(n, seed=77)
    rgba4[:, 3] *= 0.3
    recB = gs4d.build_records_4d(pos4 * 0.1, q4, sc4 * 2.0, life, fade, vel, rgba4)
    ctx = gs4d.Context(W, H)
    dA, dB, kb, ib = ctx.buffer(recA), ctx.buffer(recB), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.set_mode(gs4d.MODE_4D_SORTED)
    ctx.clear()
    ctx.set_uniforms(time=0.0, min_opacity=0.0, view=view, proj=proj)
    ctx.bind(1, ib)
    ctx.bind(2, dA)
    ctx.keygen(dA, 0.0, cam[0], kb, ib, n)
    ctx.sort_pairs(kb, ib, n)
    ctx.draw_instanced(n)
    tB = 2.0
    ctx.set_uniforms(time=tB)
    ctx.bind(2, dB)
    ctx.keygen(dB, tB, cam[0], kb, ib, n)                      # same buffers: frame A's order is gone from them
    ctx.sort_pairs(kb, ib, n)
    ctx.draw_instanced(n)                                     # blends onto A's pixels (no clear in between)
    img = ctx.read_pixels()
    eimg = oracle.clear_image(W, H)
    for rec, t in ((recA, 0.0), (recB, tB)):
        _, ekeys = oracle.keygen(rec, t, cam[0])
        _, eperm = oracle.sort_pairs(ekeys.view(np.uint32), np.arange(n, dtype=np.uint32), "lsd")
        oracle.composite(oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H, t=t), eperm, oracle.MODE_4D, W, H, eimg)
    st = ctx.stats()
    assert st["reruns"] >= 1
    assert linf(img, eimg) <= TOL
    perm = ctx.read(ib, np.uint32, n)                         # and the caller's buffers hold frame B's order
    assert np.array_equal(perm, eperm)
    ctx.close()


def test_aborted_frames_nobody_looked_at_are_counted_and_learned_from(gs4d, oracle, monkeypatch):
    """A frame loop without read-backs (bench.py's timed window): Clear -> keygen -> sort -> Draw, again and again.  The first frames of
    this scene overflow the speculative tile-list capacity and are cleared away before anybody sees them: they are not re-run, but they
    are counted, and the capacity they asked for is there for the later frames — which then run complete."""
    monkeypatch.delenv("GS4D_DRAW_PATH", raising=False)
    W, H = 1001, 701
    cam = ((0.0, 0.0, 30.0), (0.0, 0.0, -1.0))
    view, proj = cam_mats(gs4d, cam, W, H)
    rec = _big_splats(gs4d, seed=12)
    n = rec.shape[0]
    ctx = gs4d.Context(W, H)
    lanes = ctx.stats()["lanes"]
    data = ctx.buffer(rec)
    pairs = [(ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)) for _ in range(lanes)]
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.set_mode(gs4d.MODE_4D_SORTED)
    ctx.bind(2, data)
    ctx.set_uniforms(time=0.0, min_opacity=0.0, view=view, proj=proj)

    def frame(f):
        kb, ib = pairs[f % lanes]
        ctx.clear()
        ctx.keygen(data, 0.0, cam[0], kb, ib, n)
        ctx.sort_pairs(kb, ib, n)
        ctx.bind(1, ib)
        ctx.draw_instanced(n)
    for f in range(3 * lanes):
        frame(f)
    ctx.finish()
    st = ctx.stats()
    assert st["aborted_discarded"] >= 1                       # the first round of lanes ran with too little capacity
    settled = st["aborted_discarded"]
    for f in range(3 * lanes, 5 * lanes):
        frame(f)
    ctx.finish()
    st2 = ctx.stats()
    assert st2["aborted_discarded"] == settled                # nothing aborts any more: every later frame was a complete render
    eimg, _, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
    assert linf(ctx.read_pixels(), eimg) <= TOL
    assert ctx.stats()["reruns"] == st2["reruns"]             # the frame that was read needed no re-run either
    ctx.close()


def test_clear_colour_set_after_a_draw_does_not_repaint_its_rerun(gs4d, oracle, monkeypatch):
    """glClearColor only affects later glClear calls.  A draw onto a (lazily) cleared image whose tile lists overflow is re-run after the
    application has already set the next frame's clear colour: the re-run composites over the colour the image was cleared with."""
    monkeypatch.delenv("GS4D_DRAW_PATH", raising=False)
    W, H = 1001, 701
    cam = ((0.0, 0.0, 30.0), (0.0, 0.0, -1.0))
    view, proj = cam_mats(gs4d, cam, W, H)
    rec = _big_splats(gs4d, seed=13)
    n = rec.shape[0]
    ctx = gs4d.Context(W, H)
    data, kb, ib = ctx.buffer(rec), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.set_mode(gs4d.MODE_4D_SORTED)
    ctx.clear()
    ctx.set_uniforms(time=0.0, min_opacity=0.0, view=view, proj=proj)
    ctx.keygen(data, 0.0, cam[0], kb, ib, n)
    ctx.sort_pairs(kb, ib, n)
    ctx.bind(1, ib)
    ctx.bind(2, data)
    ctx.draw_instanced(n)                                     # overflows: pending validation
    ctx.set_clear_color((1.0, 0.0, 1.0, 1.0))                 # for the NEXT clear
    img = ctx.read_pixels()
    assert ctx.stats()["reruns"] >= 1
    eimg, _, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
    assert linf(img, eimg) <= TOL
    ctx.clear()
    assert np.array_equal(ctx.read_pixels()[0, 0], np.array([1.0, 0.0, 1.0, 1.0], np.float32))
    ctx.close()


def test_compact_and_full_record_shadows(gs4d, oracle, monkeypatch):
    """The private SoA shadow of the records keeps a symmetric sig without its mirrored half (72 bytes a record instead of 96).  Symmetric
    records: both layouts give the same projected records and the same image, bit for bit.  A record whose sig is NOT symmetric must make
    the library fall back to the full layout — the shader reads iSig[c][r] where it reads it (…Instanced.GLSL:84-95), mirrored or not."""
    monkeypatch.delenv("GS4D_DRAW_PATH", raising=False)
    from test_gpu_render import gpu_frame, check_projected
    n, W, H = 20000, 640, 360
    pos4, q, sc, life, fade, vel, rgba = scenes.cube_params_4d(n, seed=21)
    rec = gs4d.build_records_4d(pos4, q, sc * 3.0, life * 20.0, fade, vel, rgba)
    # R S S R^T comes out of the constructor symmetric only up to rounding for a general rotation (the teapot scenes' records ARE symmetric
    # bit for bit: tests/test_oracle_golden.py); mirror the upper triangle so that this set qualifies for the compact shadow
    sig = rec[:, 8:].reshape(-1, 4, 4)
    iu = np.triu_indices(4, 1)
    sig[:, iu[1], iu[0]] = sig[:, iu[0], iu[1]]
    assert np.array_equal(sig, sig.transpose(0, 2, 1))
    cam = scenes.CAM_CUBE
    t = 20.0
    out = {}
    for full in (0, 1):
        monkeypatch.setenv("GS4D_SOA_FULL", str(full))           # read at every repack
        ctx = gs4d.Context(W, H)
        img, projd, _, (view, proj) = gpu_frame(ctx, gs4d, rec, cam, t=t)
        ctx.close()
        out[full] = (img, projd)
    eproj = oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H, t, 0.0)
    for full in (0, 1):
        check_projected(oracle, out[full][1], eproj)
    assert np.array_equal(out[0][0], out[1][0])
    eimg, _, _ = oracle.render_4d(rec, True, t, 0.0, cam[0], view, proj, W, H)
    assert linf(out[0][0], eimg) <= TOL
    # not symmetric: every 7th record gets another iSig[0][1] (and keeps iSig[1][0]), every 11th another iSig[2][3]
    rec2 = rec.copy()
    rec2[::7, 8 + 1] *= 1.25
    rec2[::11, 8 + 4 * 2 + 3] += 0.05
    monkeypatch.delenv("GS4D_SOA_FULL", raising=False)
    ctx = gs4d.Context(W, H)
    img2, projd2, _, _ = gpu_frame(ctx, gs4d, rec2, cam, t=t)
    check_projected(oracle, projd2, oracle.preprocess(oracle.MODE_4D, rec2, view, proj, W, H, t, 0.0))
    eimg2, _, _ = oracle.render_4d(rec2, True, t, 0.0, cam[0], view, proj, W, H)
    assert linf(img2, eimg2) <= TOL
    assert linf(eimg2, eimg) > 1e-3                              # the asymmetry is visible: a compact shadow would have lost it
    # and back: a symmetric buffer uploaded into the same context uses the compact layout again, same bits as before
    img3, _, _, _ = gpu_frame(ctx, gs4d, rec, cam, t=t)
    assert np.array_equal(img3, out[0][0])
    ctx.close()


@pytest.mark.parametrize("fuse", [1, 0])
def test_static_3d_record_shadow(gs4d, oracle, monkeypatch, fuse):
    """A set of static 3D splats in the reference's 4D record (Scenes.h ObjectDisplay: mu_t = 0, no space-time covariance, Sigma44 = 1 — the
    benchmark's records) has eight values that are the same in every record; the private shadow then keeps 64 bytes a record and passes
    the eight as constants.  Same projected records, same keys and same image as the 96-byte layout, bit for bit; any record that differs
    in one of the eight (even by a sign bit) sends the buffer to the next layout down and the result is still the reference's."""
    monkeypatch.delenv("GS4D_DRAW_PATH", raising=False)
    monkeypatch.setenv("GS4D_FUSE_KEYGEN", str(fuse))             # 0: the stand-alone key kernel reads the shadow too
    from test_gpu_render import gpu_frame, check_projected
    n, W, H = 20000, 640, 360
    pos, q, sc, rgba = scenes.cube_params(n, seed=5)
    rec = gs4d.build_records_3d(pos, q, sc * 3.0, rgba)
    assert np.all(rec[:, 3] == 0) and np.all(rec[:, 20:23] == 0) and np.all(rec[:, 23] == 1)
    cam = scenes.CAM_CUBE
    t = 3.0
    out = {}
    for full in (0, 1):
        monkeypatch.setenv("GS4D_SOA_FULL", str(full))
        ctx = gs4d.Context(W, H)
        img, projd, st, (view, proj) = gpu_frame(ctx, gs4d, rec, cam, t=t)
        assert st["record_read_bytes"] == (96 if full else 64)
        ctx.close()
        out[full] = (img, projd)
    monkeypatch.delenv("GS4D_SOA_FULL", raising=False)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
    check_projected(oracle, out[0][1], oracle.preprocess(oracle.MODE_4D, rec, view, proj, W, H, t, 0.0))
    eimg, _, _ = oracle.render_4d(rec, True, t, 0.0, cam[0], view, proj, W, H)
    assert linf(out[0][0], eimg) <= TOL
    # a symmetric set whose time components are NOT constant: 72 bytes; one record with mu_t = -0.0 (another bit pattern than 0.0): not 64;
    # an asymmetric spatial block alone does not matter to the static layout (it keeps all nine elements): still 64
    for edit, want in ((lambda r: r.__setitem__((slice(None, None, 9), 3), 0.5), 72), (lambda r: r.__setitem__((n // 2, 3), -0.0), 72),
                       (lambda r: r.__setitem__((7, 8 + 1), r[7, 8 + 1] * 1.5 + 0.01), 64), (lambda r: r.__setitem__((0, 23), 2.0), 72)):
        rec2 = rec.copy()
        sig = rec2[:, 8:].reshape(-1, 4, 4)
        iu = np.triu_indices(4, 1)
        sig[:, iu[1], iu[0]] = sig[:, iu[0], iu[1]]
        edit(rec2)
        ctx = gs4d.Context(W, H)
        img2, projd2, st2, _ = gpu_frame(ctx, gs4d, rec2, cam, t=t)
        assert st2["record_read_bytes"] == want, want
        ctx.close()
        check_projected(oracle, projd2, oracle.preprocess(oracle.MODE_4D, rec2, view, proj, W, H, t, 0.0))
        eimg2, _, _ = oracle.render_4d(rec2, True, t, 0.0, cam[0], view, proj, W, H)
        assert linf(img2, eimg2) <= TOL


def test_record_shadow_follows_buffer_updates(gs4d, oracle, monkeypatch):
    """The layout of the private shadow is decided per upload: a static set that gets one moving record through gs4d_buffer_subdata is
    repacked in a layout that can hold it (and back), and every frame is the reference's."""
    monkeypatch.delenv("GS4D_DRAW_PATH", raising=False)
    from test_gpu_render import cam_mats
    n, W, H = 12000, 640, 360
    pos, q, sc, rgba = scenes.cube_params(n, seed=9)
    rec = gs4d.build_records_3d(pos, q, sc * 3.0, rgba)
    cam, t = scenes.CAM_CUBE, 2.0
    view, proj = cam_mats(gs4d, cam, W, H)
    ctx = gs4d.Context(W, H)
    db, kb, ib = ctx.buffer(rec), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.set_mode(gs4d.MODE_4D_SORTED)

    def frame(r):
        ctx.clear()
        ctx.set_uniforms(time=t, min_opacity=0.0, view=view, proj=proj)
        ctx.keygen(db, t, cam[0], kb, ib, n)
        ctx.sort_pairs(kb, ib, n)
        ctx.bind(1, ib)
        ctx.bind(2, db)
        ctx.draw_instanced(n)
        img = ctx.read_pixels()
        eimg, eperm, _ = oracle.render_4d(r, True, t, 0.0, cam[0], view, proj, W, H)
        assert linf(img, eimg) <= TOL
        assert np.array_equal(ctx.read(ib, np.uint32, n), eperm)
        return ctx.stats()["record_read_bytes"]

    assert frame(rec) == 64
    rec2 = rec.copy()
    rec2[n // 3, 3] = 1.5                                          # one record gets a time of its own ...
    rec2[n // 3, 20:23] = (4.0, -3.0, 2.0)                         # ... and moves: sig[3].xyz (what the key loop extrapolates with, Scenes.h:30-33)
    rec2[n // 3, 8 + 3], rec2[n // 3, 12 + 3], rec2[n // 3, 16 + 3] = 4.0, -3.0, 2.0
    ctx.subdata(db, rec2)
    assert frame(rec2) in (72, 96)
    ctx.subdata(db, rec)
    assert frame(rec) == 64
    ctx.close()


@pytest.mark.parametrize("W,H,offset", [(640, 360, 0), (640, 360, 4), (62, 30, 0), (66, 41, 8)])
def test_rgba8_pack_any_width_and_alignment(gs4d, monkeypatch, W, H, offset):
    """gs4d_read_pixels_rgba8_device: four pixels per thread when the row length and the destination allow it, one otherwise; the bytes are
    the same — round(clamp(x) * 255) of the float image, clear colour where nothing was drawn (tiles that are not in memory)."""
    monkeypatch.delenv("GS4D_DRAW_PATH", raising=False)
    n = 3000
    pos, q, sc, rgba = scenes.cube_params(n, seed=17)
    rec = gs4d.build_records_3d(pos * 0.5, q, sc * 4.0, rgba)
    cam = scenes.CAM_CUBE
    view, proj = cam_mats(gs4d, cam, W, H)
    ctx = gs4d.Context(W, H)
    db, ob = ctx.buffer(rec), ctx.buffer(nbytes=W * H * 4 + 64)
    ctx.set_clear_color((0.25, 0.5, 0.75, 1.0))
    ctx.clear()
    ctx.set_uniforms(time=0.0, min_opacity=0.0, view=view, proj=proj)
    ctx.set_mode(gs4d.MODE_4D_DIRECT)
    ctx.bind(1, db)
    ctx.draw_instanced(n)
    ctx.read_pixels_rgba8_device(ctx.device_ptr(ob)[0] + offset, W * H * 4)
    ctx.finish()
    got = ctx.read(ob, np.uint8, W * H * 4, offset=offset).reshape(H, W, 4).astype(np.int32)
    img = ctx.read_pixels()
    ctx.close()
    want = np.rint(np.clip(img, 0.0, 1.0) * 255.0).astype(np.int32)
    assert np.array_equal(got, want)
    assert np.any(np.all(want == np.array([64, 128, 191, 255]), axis=2)) and np.any(np.any(want != np.array([64, 128, 191, 255]), axis=2))      # clear tiles and drawn ones


def test_lane_streams_are_chosen_beside_foreign_streams(gs4d, monkeypatch):
    """gs4d_create picks the frame lanes' streams by experiment (a candidate is kept only if a kernel on it runs while a spinning kernel
    occupies each lane chosen so far): with foreign streams alive in the process — which shift HIP's stream-to-hardware-queue mapping so
    that two lanes would share a queue — the context still gets its lanes, reports how many candidates it discarded, and renders the same
    frames as a context that took its streams as they came (GS4D_PROBE_QUEUES=0)."""
    monkeypatch.delenv("GS4D_DRAW_PATH", raising=False)
    hip = C.CDLL("libamdhip64.so")
    foreign = []
    for _ in range(2):
        s = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0      # hipStreamNonBlocking
        foreign.append(s)
    W, H, n = 640, 360, 20000
    pos, q, sc, rgba = scenes.cube_params(n, seed=23)
    rec = gs4d.build_records_3d(pos, q, sc * 3.0, rgba)
    cam = scenes.CAM_CUBE
    view, proj = cam_mats(gs4d, cam, W, H)
    imgs = []
    for probe in ("1", "0"):
        monkeypatch.setenv("GS4D_PROBE_QUEUES", probe)
        ctx = gs4d.Context(W, H)
        st = ctx.stats()
        assert st["lanes"] >= 1 and st["lane_streams_rejected"] >= 0
        if probe == "0":
            assert st["lane_streams_rejected"] == 0
        db, kb, ib = ctx.buffer(rec), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
        ctx.set_clear_color(gs4d.CLEAR_COLOR)
        ctx.set_mode(gs4d.MODE_4D_SORTED)
        frames = []
        for f in range(6):                                         # more frames than lanes: every lane renders
            ctx.clear()
            ctx.set_uniforms(time=float(f), min_opacity=0.0, view=view, proj=proj)
            ctx.keygen(db, float(f), cam[0], kb, ib, n)
            ctx.sort_pairs(kb, ib, n)
            ctx.bind(1, ib)
            ctx.bind(2, db)
            ctx.draw_instanced(n)
            frames.append(ctx.read_pixels())
        ctx.close()
        imgs.append(frames)
    monkeypatch.delenv("GS4D_PROBE_QUEUES", raising=False)
    for a, b in zip(*imgs):
        assert np.array_equal(a, b)
    for s in foreign:
        assert hip.hipStreamDestroy(s) == 0


@pytest.mark.parametrize("rename", [1, 0])
def test_one_key_pair_for_all_frames(gs4d, oracle, monkeypatch, rename):
    """The reference's buffer layout — ONE key buffer and ONE index buffer for every frame (Scenes.h m_key_buf / m_values_buf) — with frames
    queued back to back on several lanes: each gs4d_keygen overwrites both buffers entirely, so instead of waiting for the previous
    frame's sort (another lane) it takes fresh storage for them (GS4D_RENAME=0: it waits).  Every frame that is looked at, and the
    buffers' contents whenever they are read, are what a strictly sequential execution gives."""
    monkeypatch.delenv("GS4D_DRAW_PATH", raising=False)
    monkeypatch.setenv("GS4D_RENAME", str(rename))
    W, H = 640, 360
    cam = scenes.CAM_CUBE
    view, proj = cam_mats(gs4d, cam, W, H)
    n = 40000
    pos4, q, sc, life, fade, vel, rgba = scenes.cube_params_4d(n, seed=31)
    rec = gs4d.build_records_4d(pos4, q, sc * 4.0, life * 10.0, fade, vel, rgba)
    ctx = gs4d.Context(W, H)
    data, kb, ib = ctx.buffer(rec), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.set_mode(gs4d.MODE_4D_SORTED)
    ctx.bind(2, data)
    ctx.bind(1, ib)
    times = [3.0 * f for f in range(14)]
    look = {5, 9, 13}
    for f, t in enumerate(times):
        ctx.clear()
        ctx.set_uniforms(time=t, min_opacity=0.0, view=view, proj=proj)
        ctx.keygen(data, t, cam[0], kb, ib, n)
        ctx.sort_pairs(kb, ib, n)
        ctx.draw_instanced(n)
        if f in look:
            img = ctx.read_pixels()
            eimg, eperm, _ = oracle.render_4d(rec, True, t, 0.0, cam[0], view, proj, W, H)
            assert linf(img, eimg) <= TOL, f"frame {f}"
            if f == 9:
                assert np.array_equal(ctx.read(ib, np.uint32, n), eperm)       # the NAME holds this frame's order, whatever storage is behind it
                _, ekeys = oracle.keygen(rec, t, cam[0])
                assert np.array_equal(np.sort(ctx.read(kb, np.float32, n).view(np.uint32)), np.sort(ekeys.view(np.uint32)))
    ctx.finish()
    st = ctx.stats()
    assert (st["renamed_keygens"] > 0) == (rename == 1 and st["lanes"] > 1)
    # a buffer whose address the caller holds keeps its storage
    p0, _ = ctx.device_ptr(kb)
    for f in range(6):
        ctx.clear()
        ctx.keygen(data, 1.0 + f, cam[0], kb, ib, n)
        ctx.sort_pairs(kb, ib, n)
        ctx.draw_instanced(n)
    ctx.finish()
    assert ctx.device_ptr(kb)[0] == p0
    ctx.close()


@pytest.mark.parametrize("path", ["auto", "ordered"])
def test_tile_rectangles_beyond_the_packed_range(gs4d, oracle, monkeypatch, path):
    """The list-building kernels read a 4-byte packed tile rectangle per record (10 bits per tile coordinate, 6 per extent).  A frame wider
    than 8184 pixels has tiles beyond 1022, and a close-up splat spans more than 63 tiles: both take the escape route (the pixel
    rectangle of the projected record).  Same picture as the checker's on both list paths."""
    if path == "ordered":
        monkeypatch.setenv("GS4D_DRAW_PATH", "ordered")
    else:
        monkeypatch.delenv("GS4D_DRAW_PATH", raising=False)
    W, H = 8400, 96
    cam = ((0.0, 0.0, 40.0), (0.0, 0.0, -1.0))
    view, proj = cam_mats(gs4d, cam, W, H)
    n = 4000
    pos, q, sc, rgba = scenes.cube_params(n, seed=81)
    pos[:, 0] *= 10.0                                            # spread across the very wide frame (x in +-2000 at depth 40..: most of it visible)
    pos[:, 1] *= 0.02
    pos[:, 2] *= 0.05
    sc *= 6.0
    sc[:8] *= 40.0                                               # a few huge ones: hundreds of tiles across
    rgba[:, 3] *= 0.5
    rec = gs4d.build_records_3d(pos, q, sc, rgba)
    ctx = gs4d.Context(W, H)
    from test_gpu_render import gpu_frame
    img, projd, st, _ = gpu_frame(ctx, gs4d, rec, cam)
    ctx.close()
    eimg, _, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
    x0 = (projd[:, 10].copy().view(np.uint32) & 0xFFFF).astype(np.int64)
    x1 = (projd[:, 11].copy().view(np.uint32) & 0xFFFF).astype(np.int64)
    drawn = (projd[:, 14] != 0) & (x0 <= x1)
    assert (x0[drawn] // 8 >= 1023).any()                        # rectangles that start beyond the packed range
    assert ((x1[drawn] // 8 - x0[drawn] // 8) >= 63).any()       # ... and that are wider than it holds
    assert linf(img, eimg) <= TOL
    assert np.abs(eimg[:, 8184:] - np.array(gs4d.CLEAR_COLOR, np.float32)).max() > 0.05      # something is drawn out there


def test_two_contexts_created_and_used_on_two_threads(gs4d, oracle):
    """No process-wide state behind the C ABI: two threads each create a context of their own (gs4d_create may name any device) and run whole
    frames — key loop, sort, draw — at the same time.  (Round 3 cached the persistent sort's workgroup count in a function-local static: racy
    here, and wrong for contexts on two different devices.)  ctypes releases the GIL for the duration of a call, so the calls do overlap."""
    import threading
    W, H = 640, 360
    sizes = (150_000, 40_000)
    results, errors = {}, []

    def worker(k, n):
        try:
            pos, q, scale, rgba = scenes.cube_params(n, seed=100 + k)
            rec = gs4d.build_records_3d(pos, q, scale * 2.0, rgba)
            cam = scenes.CAM_CUBE
            view = gs4d.look_at(cam[0], cam[1]); proj = gs4d.perspective(scenes.FOV, W, H, scenes.ZNEAR, scenes.ZFAR)
            ctx = gs4d.Context(W, H)
            ctx.set_clear_color(gs4d.CLEAR_COLOR)
            db, kb, ib = ctx.buffer(rec), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
            for _ in range(6):
                ctx.clear()
                ctx.set_uniforms(time=0.0, min_opacity=0.0, view=view, proj=proj)
                ctx.keygen(db, 0.0, cam[0], kb, ib, n)
                ctx.sort_pairs(kb, ib, n)
                ctx.set_mode(gs4d.MODE_4D_SORTED)
                ctx.bind(1, ib); ctx.bind(2, db)
                ctx.draw_instanced(n)
            results[k] = (rec, ctx.read_pixels(), ctx.read(ib, np.uint32, n), view, proj, cam)
            ctx.close()
        except Exception as e:      # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k, n)) for k, n in enumerate(sizes)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert len(results) == len(sizes)
    for k in results:
        rec, img, perm, view, proj, cam = results[k]
        eimg, eperm, _ = oracle.render_4d(rec, True, 0.0, 0.0, cam[0], view, proj, W, H)
        assert np.array_equal(perm, eperm)
        assert np.abs(img.astype(np.float64) - eimg).max() <= 1e-4
