#!/usr/bin/env python3
"""Regenerates tests/golden/gl_*.npz: the reference's own GLSL programs executed here by oracle/_ref/refgl (Mesa llvmpipe, see
oracle/ref/refgl_main.cpp).  BUILD CONTAINER ONLY (needs /root/reference and the image's swrast_dri.so); the fixtures are committed.

Three families:
  gl_vs_*    vertex stage by transform feedback: gl_Position per quad corner, oSig, oColor, oFragPos per corner, oFaulty, oTimeOpacity
             (4D: Splat4DVertexShaderInstanced.GLSL; 3D: Splat3DVertexShaderFull.GLSL; 2D: Splat2DVSI.GLSL) — no rasteriser involved
  gl_img_*   images: clear (Application.cpp:125), blend state (:150-154), glDrawElementsInstanced (Renderer.cpp:33-39) into an RGBA32F
             attachment (+ once RGBA8), instances in the order the reference's compute sort (run by refgl too) leaves in the index buffer
  gl_sort_*  the three compute programs under radix_sort.hpp:258-392's dispatch sequence

Inputs are either committed refgen fixtures (tests/golden/*.bin: the reference's own records) or seeded synthetic sets (tests/scenes.py)
whose records are embedded in the fixture.  Camera matrices come from the checker (bit for bit GLM's: test_oracle_golden.py::test_camera).
Nothing under oracle/ is used by the product.
"""
import json
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import oracle_lib as ol          # noqa: E402
import scenes                    # noqa: E402
import splat_draw_cases as sd    # noqa: E402
import gl_cases                  # noqa: E402

REFGL = os.path.join(HERE, "_ref", "refgl")
OUT = os.path.join(ROOT, "tests", "golden")
TMP = tempfile.mkdtemp(prefix="refgl_")


def refgl(*args):
    subprocess.check_call([REFGL] + [str(a) for a in args], stderr=subprocess.DEVNULL)


def uniforms(t, min_opacity, view, proj):
    return np.concatenate([[t, min_opacity], view, proj]).astype(np.float32)


def gl_draw(kind, W, H, rec, sortidx, uni, want, blend=None):
    """-> dict with 'tf' (n, 6, K), 'img' (H, W, 4) f32, 'img8' (H, W, 4) u8 as requested"""
    rec = np.ascontiguousarray(rec, np.float32)
    n = rec.shape[0]
    rp, ip, up, op = (os.path.join(TMP, x) for x in ("rec.bin", "idx.bin", "uni.bin", "out"))
    rec.tofile(rp)
    uni.astype(np.float32).tofile(up)
    if sortidx is not None:
        np.ascontiguousarray(sortidx, np.uint32).tofile(ip)
    args = ["draw", kind, W, H, n, rp, ip if sortidx is not None else "-", up, op] + list(want)
    if blend is not None:
        args += ["blend", blend[0], blend[1]]
    refgl(*args)
    out = {}
    if "tf" in want:
        out["tf"] = np.fromfile(op + ".tf.f32", np.float32).reshape(n, 6, -1)
    if "img" in want:
        out["img"] = np.fromfile(op + ".img.f32", np.float32).reshape(H, W, 4)
    if "img8" in want:
        out["img8"] = np.fromfile(op + ".img.u8", np.uint8).reshape(H, W, 4)
    if "img16" in want:
        out["img16"] = np.fromfile(op + ".img.u16", np.uint16).reshape(H, W, 4)
    return out


def gl_sort(keys_u32, vals_u32):
    kp, vp, op = (os.path.join(TMP, x) for x in ("k.bin", "v.bin", "s"))
    np.ascontiguousarray(keys_u32, np.uint32).tofile(kp)
    np.ascontiguousarray(vals_u32, np.uint32).tofile(vp)
    refgl("sort", len(keys_u32), kp, vp, op)
    return np.fromfile(op + ".keys.u32", np.uint32), np.fromfile(op + ".vals.u32", np.uint32)


def corners_of(tf):
    """6 captured vertices per instance in index order 0,2,1,2,0,3 (Geometry.h:50) -> the four corners; the repeats must agree bit for bit"""
    b = tf.view(np.uint32)
    assert np.array_equal(b[:, 3], b[:, 1]) and np.array_equal(b[:, 4], b[:, 0]), "repeated vertices of an instance differ"
    return tf[:, [0, 2, 1, 5], :]


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    raw = open(path, "rb").read()
    print(f"  {name}.npz  {len(raw) / 1024:.0f} KiB")
    return {"bytes": len(raw), "crc32": zlib.crc32(raw)}


MAN = {}


def vs_fixture(name, kind, rec, t, min_opacity, view, proj, W, H, embed_records, source):
    uni = uniforms(t, min_opacity, view, proj)
    tf = gl_draw(kind, W, H, rec, None, uni, ["tf"])["tf"]
    c = corners_of(tf)
    arrays = {"uniforms": uni, "size": np.array([W, H], np.int32),
              "pos": c[:, :, 0:4], "sig": c[:, 0, 4:8], "color": c[:, 0, 8:12], "fragpos": c[:, :, 12:14]}
    if kind in ("4d", "4dmod"):
        arrays["faulty"] = (c[:, 0, 16] > 0).astype(np.uint8)
        arrays["topac"] = c[:, 0, 17]
    elif kind == "3d":
        arrays["faulty"] = (c[:, 0, 16] > 0).astype(np.uint8)
    else:
        arrays["sspos"] = c[:, 0, 16:18]
    # per-instance varyings are identical on the four corners of a visible instance
    vis = ~(arrays["faulty"] > 0) if "faulty" in arrays else np.ones(len(rec), bool)
    assert np.array_equal(c[vis][:, 1:, 4:12].view(np.uint32), np.broadcast_to(c[vis][:, :1, 4:12], c[vis][:, 1:, 4:12].shape).view(np.uint32))
    if embed_records:
        arrays["records"] = np.ascontiguousarray(rec, np.float32)
    MAN[name] = dict(save(name, **arrays), kind=kind, source=source, n=int(len(rec)), culled=int((~vis).sum()))


def crop_box(img, clear):
    """bounding box of the pixels that differ from the clear colour; outside it the image IS the clear colour, bit for bit"""
    touched = (img.view(np.uint32) != clear.view(np.uint32)[None, None, :]).any(axis=2)
    ys, xs = np.nonzero(touched)
    return int(xs.min()), int(ys.min()), int(xs.max()) + 1, int(ys.max()) + 1, int(touched.sum())


def img_fixture(name, kind, rec, t, min_opacity, cam_pos, view, proj, W, H, embed_records, source, sorted_draw=True, blend=None, want8=False, unorm16=False):
    """unorm16: the image comes from an RGBA16 attachment instead of RGBA32F — for blend pairs that leave [0, 1] (a float attachment does not
    clamp, the reference's fixed-point window does; the checker and the product clamp like the window)"""
    uni = uniforms(t, min_opacity, view, proj)
    arrays = {"uniforms": uni, "size": np.array([W, H], np.int32), "cam": np.asarray(cam_pos, np.float32)}
    order = None
    if kind == "4d" and sorted_draw:
        # keys as the reference's CPU loop makes them (Scenes.h:314-319; the checker's are bit for bit the reference's: test_oracle_golden.py),
        # sorted by the reference's compute programs: sort(m_values_buf = float keys, m_key_buf = indices), Scenes.h:327
        idx, key = ol.keygen(rec, t, cam_pos)
        ks, order = gl_sort(key.view(np.uint32), idx)
        assert np.array_equal(ks, np.sort(key.view(np.uint32)))
        arrays["order"] = order
    elif kind == "4d":
        order = np.arange(len(rec), dtype=np.uint32)
    # (kind "4dmod": Splat4DVertexShaderMod.GLSL has no sort index — instance k draws record k)
    want = [("img16" if unorm16 else "img")] + (["img8"] if want8 else [])
    res = gl_draw(kind, W, H, rec, order, uni, want, blend)
    if unorm16:
        full = res["img16"]
        clear = full[0, 0].copy()
        assert np.array_equal(clear, np.round(ol.CLEAR.astype(np.float64) * 65535.0).astype(np.uint16))
        x0, y0, x1, y1, touched = crop_box(full.astype(np.uint32), clear.astype(np.uint32))
        arrays["crop16"] = full[y0:y1, x0:x1].copy()
        res["img"] = full
    else:
        x0, y0, x1, y1, touched = crop_box(res["img"], ol.CLEAR)
        arrays["crop"] = res["img"][y0:y1, x0:x1].copy()
    arrays["box"] = np.array([x0, y0, x1, y1], np.int32)
    if want8:
        c8 = res["img8"]
        out8 = c8.copy(); out8[y0:y1, x0:x1] = c8[0, 0]
        assert (out8 == c8[0, 0]).all()
        arrays["crop8"] = c8[y0:y1, x0:x1].copy()
        arrays["clear8"] = c8[0, 0].copy()
    if blend is not None:
        arrays["blend"] = np.array(blend, np.int32)
    if embed_records:
        arrays["records"] = np.ascontiguousarray(rec, np.float32)
    MAN[name] = dict(save(name, **arrays), kind=kind, source=source, n=int(len(rec)), touched_pixels=touched,
                     full_crc32=zlib.crc32(res["img"].tobytes()))


def gl_lines(W, H, vp, sets):
    args = ["lines", W, H, os.path.join(TMP, "vp.bin"), os.path.join(TMP, "lines")]
    np.asarray(vp, np.float32).tofile(os.path.join(TMP, "vp.bin"))
    for k, (verts, col, width, strip) in enumerate(sets):
        cp, vp_ = os.path.join(TMP, f"col{k}.bin"), os.path.join(TMP, f"verts{k}.bin")
        np.asarray(col, np.float32).tofile(cp)
        np.ascontiguousarray(verts, np.float32).tofile(vp_)
        args += [cp, width, len(verts), vp_, 2 if np.asarray(verts).shape[-1] == 2 else int(strip)]
    refgl(*args)
    return np.fromfile(os.path.join(TMP, "lines.img.f32"), np.float32).reshape(H, W, 4)


def lines_fixture(name, cam, W, H, source):
    """the overlays every 4D scene's Render() starts with (Scenes.h:303-310): DrawGrid(2000, 2000, 200, 200, {1,1,1,0.15}, cam, 1) with its
    zero-initialised first half (Renderer.cpp:121), DrawAxis(cam, 500, 3) = three 10-unit lines, the unit line (width 5), the path line of
    LinearMotion (width 5) and a 60-point strip (width 2) — all into one frame"""
    view, proj = ol.look_at(cam[0], cam[1]), ol.perspective(scenes.FOV, W, H, scenes.ZNEAR, scenes.ZFAR)
    vp = (proj.reshape(4, 4).T @ view.reshape(4, 4).T).T.reshape(-1).astype(np.float32)                  # Camera::GetViewProjMatrix, column-major P * V
    sets = [(gl_cases.grid_vertices(2000.0, 2000.0, 200, 200), (1, 1, 1, 0.15), 1.0, 0)]
    sets += [(np.array([[0, 0, 0], p], np.float32), c, 3.0, 0) for p, c in (([10, 0, 0], (1, 0, 0, 1)), ([0, 10, 0], (0, 1, 0, 1)), ([0, 0, 10], (0, 0, 1, 1)))]
    sets += [(np.array([[0, 0, 0], [1, 0, 0]], np.float32), (1, 1, 1, 1), 5.0, 0), (np.array([[0, 0, 0], [50, 0, 0]], np.float32), (1, 0, 0, 1), 5.0, 0)]
    sets += [(np.stack([np.array([20 * np.cos(a), 5.0 + a, 20 * np.sin(a)], np.float32) for a in np.linspace(0, 6.0, 60)]), (1.0, 0.5, 0.1, 0.8), 2.0, 1)]
    sets += [(np.array([[-0.9, -0.8], [0.7, 0.95], [-0.5, 0.5], [0.5, 0.5]], np.float32), (0.2, 0.9, 0.3, 0.5), 3.0, 0)]       # Renderer::DrawLine(vec2, vec2, color): NDC, width 3 (Renderer.cpp:168-201)
    img = gl_lines(W, H, vp, sets)
    x0, y0, x1, y1, touched = crop_box(img, ol.CLEAR)
    arrays = {"size": np.array([W, H], np.int32), "vp": vp, "box": np.array([x0, y0, x1, y1], np.int32), "nsets": np.array([len(sets)], np.int32)}
    # the image is made of few distinct colours: keep it as a palette + 8-bit indices (lossless)
    crop = img[y0:y1, x0:x1]
    pal, inv = np.unique(crop.reshape(-1, 4).view(np.uint32), axis=0, return_inverse=True)
    assert len(pal) <= 65535
    arrays["palette"] = pal.view(np.float32)
    arrays["index"] = inv.reshape(crop.shape[:2]).astype(np.uint16 if len(pal) > 255 else np.uint8)
    for k, (verts, col, width, strip) in enumerate(sets):
        arrays[f"verts{k}"] = np.ascontiguousarray(verts, np.float32)
        arrays[f"style{k}"] = np.array(list(col) + [width, strip], np.float32)
    MAN[name] = dict(save(name, **arrays), kind="lines", source=source, touched_pixels=touched)


def main():
    info = json.loads(subprocess.check_output([REFGL, "info"], stderr=subprocess.DEVNULL))
    print("GL:", info)
    MAN["_gl"] = info

    cam_t = scenes.CAM_TEAPOT
    cam_n = scenes.CAM_NONLINEAR
    cam_c = scenes.CAM_CUBE
    FOV, ZN, ZF = scenes.FOV, scenes.ZNEAR, scenes.ZFAR

    def VP(cam, W, H):
        return ol.look_at(cam[0], cam[1]), ol.perspective(FOV, W, H, ZN, ZF)

    # ---- family (a): vertex stage ----------------------------------------------------------------------------------------
    print("vertex stage:")
    lin = ol.golden("linear_first1000")
    v, p = VP(cam_t, 1920, 1080)
    for k, t in enumerate((0.0, 12.5, 49.0)):                                 # the times of linear_keys_t*
        vs_fixture(f"gl_vs_linear_first1000_t{k}", "4d", lin, t, 0.0, v, p, 1920, 1080, False, "linear_first1000")
    vs_fixture("gl_vs_linear_first1000_minop", "4d", lin, 3.0, 0.3, v, p, 1920, 1080, False, "linear_first1000")
    # Splat4DVertexShaderMod.GLSL (Scenes.h:1765: the records at binding 1, no sort index) on the same records
    vs_fixture("gl_vs_mod_linear_first1000_t1", "4dmod", lin, 12.5, 0.0, v, p, 1920, 1080, False, "linear_first1000")
    for blk, fixture in sd.BLOCKS:
        rec = ol.golden(fixture)
        for c, cam in enumerate(sd.cameras(ol)):
            for k, t in enumerate(ol.golden(f"splat_draw_4d_b{blk}_times")):
                vs_fixture(f"gl_vs_nonlinear_b{blk}_cam{c}_t{k}", "4d", rec, float(t), 0.0, cam["view"], cam["proj"], cam["W"], cam["H"], False, fixture)
    # the other four teapot scenes of the reference (records from refgen), each from its own scene camera, at the block's own time and in its fade
    OTHER = (("rotation_block45_first200", ((0.0, 60.0, 30.0), (0.0, -1.0, -0.5)), "Scenes.h:748-749"),
             ("combined_block33_first200", ((50.0, 90.0, 90.0), (0.0, -1.0, -1.0)), "Scenes.h:1003-1004"),
             ("broken_block19_first200", ((0.0, 60.0, 60.0), (0.0, -1.0, -1.0)), "Scenes.h:1941-1942"),
             ("square_block91_first200", ((0.0, 60.0, 60.0), (0.0, -1.0, -1.0)), "Scenes.h:2192-2193"))
    for fixture, cam, where in OTHER:
        rec = ol.golden(fixture)
        v, p = VP(cam, 1280, 720)
        t0 = float(rec[0, 3])
        for k, t in enumerate((t0, t0 + 0.6)):
            vs_fixture(f"gl_vs_{fixture.split('_')[0]}_t{k}", "4d", rec, t, 0.0, v, p, 1280, 720, False, fixture)
    gs4d = __import__("4dgaussiansplatrendering_amd")
    pos, q, scale, rgba = scenes.cube_params(4096)
    cube = gs4d.build_records_3d(pos, q, scale, rgba)
    v, p = VP(cam_c, 1920, 1080)
    vs_fixture("gl_vs_cube4096", "4d", cube, 0.0, 0.0, v, p, 1920, 1080, True, "scenes.cube_params(4096) -> build_records_3d")
    p4, q4, s4, life, fade, vel, col4 = scenes.cube_params_4d(4096)
    cube4 = gs4d.build_records_4d(p4, q4, s4, life, fade, vel, col4)
    vs_fixture("gl_vs_cube4d4096", "4d", cube4, 25.0, 0.0, v, p, 1920, 1080, True, "scenes.cube_params_4d(4096) -> build_records_4d, t = 25")
    # BASELINE.json configs[4]'s frame size: the viewport transform at 3840 x 2160 (window coordinates up to 3840: float32 steps of 2.4e-4 px)
    v4k, p4k = VP(cam_c, 3840, 2160)
    vs_fixture("gl_vs_cube4d4096_4k", "4d", cube4, 25.0, 0.0, v4k, p4k, 3840, 2160, False, "gl_vs_cube4d4096 (its records), 3840 x 2160")
    vs_fixture("gl_vs_cube4096_4k", "4d", cube, 0.0, 0.0, v4k, p4k, 3840, 2160, False, "gl_vs_cube4096 (its records), 3840 x 2160")
    # a camera inside the cube: many records behind it or outside the 1.2 bound -> the cull branch and the quad z-clip
    cam_in = ((20.0, -35.0, 10.0), (0.3, 0.2, -1.0))
    v, p = VP(cam_in, 1280, 720)
    vs_fixture("gl_vs_cube4096_inside", "4d", cube, 0.0, 0.0, v, p, 1280, 720, True, "cube4096 seen from inside the cube")
    verts = sd.verts72(ol.golden("splat_draw_3d_in"))
    for c, cam in enumerate(sd.cameras(ol)):
        vs_fixture(f"gl_vs_3dfull_cam{c}", "3d", verts, 0.0, 0.0, cam["view"], cam["proj"], cam["W"], cam["H"], False, "splat_draw_3d_in -> verts72")
    g2 = ol.golden("gaussians2d_records")
    cam_2d = ((-10.0, 10.0, 0.0), (1.0, -1.0, 0.0))                           # Scenes.h Gaussians2D::init
    v, p = VP(cam_2d, 1920, 1080)
    vs_fixture("gl_vs_2d", "2d", g2, 0.0, 0.0, v, p, 1920, 1080, False, "gaussians2d_records")

    # ---- family (b): images ----------------------------------------------------------------------------------------------
    print("images:")
    v, p = VP(cam_t, 1920, 1080)
    img_fixture("gl_img_c1_1080p", "4d", lin, 0.0, 0.0, cam_t[0], v, p, 1920, 1080, False, "linear_first1000 (C1')", want8=True)
    nl = ol.golden("nonlinear_block45_first200")
    tm = float(ol.golden("splat_draw_4d_b45_times")[1])
    v, p = VP(cam_n, 640, 360)
    img_fixture("gl_img_nonlinear_b45_640", "4d", nl, tm, 0.0, cam_n[0], v, p, 640, 360, False, "nonlinear_block45_first200")
    img_fixture("gl_img_nonlinear_b45_640_minop", "4d", nl, tm + 0.8, 0.25, cam_n[0], v, p, 640, 360, False, "nonlinear_block45_first200, uMinOpacity 0.25")
    img_fixture("gl_img_mod_nonlinear_b45_640", "4dmod", nl, tm, 0.0, cam_n[0], v, p, 640, 360, False, "nonlinear_block45_first200, Splat4DVertexShaderMod.GLSL: record order", sorted_draw=False)
    for s, d, tag in ((1, 0x0303, "one_oneminus"), (0x0302, 1, "srcalpha_one")):      # two more pairs of DebugMenus.h:41-59's menu
        img_fixture(f"gl_img_nonlinear_b45_640_{tag}", "4d", nl, tm, 0.0, cam_n[0], v, p, 640, 360, False, "nonlinear_block45_first200", blend=(s, d), unorm16=True)
    for fixture, cam, where in OTHER:
        rec = ol.golden(fixture)
        v, p = VP(cam, 640, 360)
        img_fixture(f"gl_img_{fixture.split('_')[0]}_640", "4d", rec, float(rec[0, 3]) + 0.4, 0.0, cam[0], v, p, 640, 360, False, fixture)
    # a dense cut of the cube: long blend chains (scale x3 as in __graft_entry__.smoke)
    cube3 = gs4d.build_records_3d(pos, q, scale * 3.0, rgba)
    cam_near = ((330.0, 210.0, -110.0), cam_c[1])
    v, p = VP(cam_near, 640, 360)
    img_fixture("gl_img_cube4096_640", "4d", cube3, 0.0, 0.0, cam_near[0], v, p, 640, 360, True, "scenes.cube_params(4096), scale x3")
    cube43 = gs4d.build_records_4d(p4, q4, s4 * 3.0, life * 8.0, fade, vel, col4)
    img_fixture("gl_img_cube4d4096_640", "4d", cube43, 25.0, 0.0, cam_near[0], v, p, 640, 360, True, "scenes.cube_params_4d(4096), scale x3, lifetime x8, t = 25")
    cam0 = sd.cameras(ol)[0]
    W3, H3 = 640, 360
    v3 = cam0["view"]; p3 = ol.perspective(FOV, W3, H3, ZN, ZF)
    img_fixture("gl_img_3dfull_640", "3d", verts, 0.0, 0.0, cam0["pos"], v3, p3, W3, H3, False, "splat_draw_3d_in -> verts72 (buffer order)")
    v, p = VP(cam_2d, 320, 180)
    img_fixture("gl_img_2d_320", "2d", g2, 0.0, 0.0, cam_2d[0], v, p, 320, 180, False, "gaussians2d_records (buffer order)")

    # ---- overlay lines (row f3) ------------------------------------------------------------------------------------------------
    print("lines:")
    lines_fixture("gl_lines_teapot_1080p", cam_t, 1920, 1080, "LinearMotion's overlays from its camera")
    lines_fixture("gl_lines_nonlinear_720p", cam_n, 1280, 720, "the same overlays from NonLinearMotion's camera")
    lines_fixture("gl_lines_2dcam_640", cam_2d, 640, 360, "the same overlays from Gaussians2D/3D's camera")

    # ---- family (c): the sort --------------------------------------------------------------------------------------------
    print("sort:")
    sort_arrays = {}
    for n in (5, 257, 2049, 100003):
        if n == 2049:
            keys = (scenes.uniform(n, 40) * 2.0 ** 32).astype(np.uint64).astype(np.uint32)                  # every digit live
        else:
            keys = ((scenes.uniform(n, 41) * 37).astype(np.uint32) * np.uint32(0x01010101)) ^ np.uint32(n)     # heavy duplicates
        ks, vs = gl_sort(keys, np.arange(n, dtype=np.uint32))
        assert np.array_equal(ks, np.sort(keys))
        if n <= 2049:
            sort_arrays[f"keys_{n}"] = keys
            sort_arrays[f"perm_{n}"] = vs
        else:
            sort_arrays[f"keys_{n}"] = keys
            sort_arrays[f"permcrc_{n}"] = np.array([zlib.crc32(vs.tobytes())], np.uint32)
    for k in range(3):
        key = ol.golden(f"linear_keys_t{k}_first4000").astype(np.float32)
        ks, vs = gl_sort(key.view(np.uint32), np.arange(key.size, dtype=np.uint32))
        sort_arrays[f"perm_linear_keys_t{k}_first4000"] = vs
    MAN["gl_sort"] = dict(save("gl_sort", **sort_arrays), source="seeded duplicates (scenes.uniform streams 40/41), linear_keys_t*_first4000")

    with open(os.path.join(OUT, "manifest_gl.json"), "w") as f:
        json.dump(MAN, f, indent=1, sort_keys=True)
    print("wrote manifest_gl.json")


if __name__ == "__main__":
    main()
