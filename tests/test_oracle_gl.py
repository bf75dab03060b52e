"""CPU: the checker (oracle/gs4d_oracle.cpp) against the reference's GLSL programs EXECUTED — tests/golden/gl_*.npz, written in the build
container by oracle/make_golden_gl.py through oracle/_ref/refgl (the reference's shader files, unmodified, on Mesa llvmpipe).  This is what
pins the GPU half of the oracle; bars and what llvmpipe rounds differently: gl_cases.py."""
import os
import zlib

import numpy as np
import pytest

import gl_cases as gl

MODES = {"4d": 0, "4dmod": 0, "3d": 2, "2d": 3}          # 4dmod: Splat4DVertexShaderMod.GLSL — the same arithmetic, instance k draws record k


def test_fixture_files_match_manifest():
    man = gl.manifest()
    assert man["_gl"]["renderer"].startswith("llvmpipe") and man["_gl"]["subpixel_bits"] == 8
    n = 0
    for name, ent in man.items():
        if name.startswith("gl_"):
            raw = open(os.path.join(gl.GOLDEN, name + ".npz"), "rb").read()
            assert len(raw) == ent["bytes"] and zlib.crc32(raw) == ent["crc32"], name
            n += 1
    assert n == len(gl.names("gl_")) and n >= 33


@pytest.mark.parametrize("name", gl.names("gl_vs_"))
def test_vertex_stage(oracle, name):
    fix = gl.load(name)
    kind = gl.manifest()[name]["kind"]
    rec = gl.records(fix, name, oracle.golden)
    t, mo, view, proj = gl.split_uniforms(fix)
    W, H = (int(x) for x in fix["size"])
    p = oracle.preprocess(MODES[kind], rec, view, proj, W, H, t=t, min_opacity=mo)
    conic = np.stack([p["q00"], p["q01"], p["q10"], p["q11"]], 1)
    m = gl.check_vertex_stage(fix, kind, gl.got_from_oracle(p), name, alpha_rtol=2e-6, conic=conic)
    print(name, m)
    assert m["n"] == rec.shape[0]


def test_vertex_vectors_exercise_the_branches():
    man = gl.manifest()
    assert man["gl_vs_cube4096_inside"]["culled"] > 1000                                    # the cull of …Instanced.GLSL:108-115
    fix = gl.load("gl_vs_cube4d4096")
    a = fix["topac"][fix["faulty"] == 0]
    assert ((a > 1e-3) & (a < 0.9)).sum() > 50                                             # a live p(t), :48-51
    fix = gl.load("gl_vs_linear_first1000_minop")
    assert (fix["topac"] == np.float32(0.3)).sum() > 100                                   # maxf(p(t), uMinOpacity), :83


def test_vertex_comparison_has_teeth(oracle):
    """one factor 1/Sigma44 too many in the conditioning, or a transposed Jacobian, fails the bars"""
    name = "gl_vs_cube4d4096"
    fix = gl.load(name)
    rec = fix["records"].copy()
    t, mo, view, proj = gl.split_uniforms(fix)
    W, H = (int(x) for x in fix["size"])
    rec[:, 8 + 3] /= rec[:, 8 + 15]
    p = oracle.preprocess(0, rec, view, proj, W, H, t=t, min_opacity=mo)
    with pytest.raises(AssertionError):
        gl.check_vertex_stage(fix, "4d", gl.got_from_oracle(p), "perturbed")


@pytest.mark.parametrize("name", gl.names("gl_img_"))
def test_image(oracle, name):
    fix = gl.load(name)
    kind = gl.manifest()[name]["kind"]
    rec = gl.records(fix, name, oracle.golden)
    t, mo, view, proj = gl.split_uniforms(fix)
    W, H = (int(x) for x in fix["size"])
    p = oracle.preprocess(MODES[kind], rec, view, proj, W, H, t=t, min_opacity=mo)
    order = None
    if "order" in fix:
        # the order the reference's compute sort left in the index buffer == the checker's sort of the checker's keys
        idx, key = oracle.keygen(rec, t, fix["cam"])
        _, order = oracle.sort_pairs(key.view(np.uint32), idx, "std")
        assert np.array_equal(order, fix["order"]), f"{name}: draw order differs from the reference's compute sort"
    blend = tuple(int(x) for x in fix["blend"]) if "blend" in fix else oracle.BLEND_OVER
    img = oracle.composite(p, order, MODES[kind], W, H, oracle.clear_image(W, H), blend=blend)
    m = gl.check_image(fix, img, gl.got_from_oracle(p), name)
    print(name, m)


def test_rgba8_window_image_is_the_float_image_quantised_per_blend(oracle):
    """for the record: the reference's window is RGBA8 and every blend rounds to 8 bits; the float image stays within what that accumulates"""
    fix = gl.load("gl_img_c1_1080p")
    x0, y0, x1, y1 = (int(x) for x in fix["box"])
    f = fix["crop"]; q = fix["crop8"].astype(np.float32) / 255.0
    assert np.array_equal(fix["clear8"], np.round(gl.CLEAR * 255.0).astype(np.uint8))
    d = np.abs(f - q).max()
    assert d <= 12 / 255.0          # up to ~20 blended layers per pixel, half an 8-bit step each in the worst case
    assert np.abs(f - q).mean() <= 1.0 / 255.0


def test_sort_permutations(oracle):
    fix = gl.load("gl_sort")
    for n in (5, 257, 2049, 100003):
        keys = fix[f"keys_{n}"]
        for which in ("std", "lsd", "glsl"):
            if which == "glsl" and n > 3000:
                continue
            ks, perm = oracle.sort_pairs(keys, np.arange(n, dtype=np.uint32), which)
            if f"perm_{n}" in fix:
                assert np.array_equal(perm, fix[f"perm_{n}"]), (n, which)
            else:
                assert zlib.crc32(perm.tobytes()) == int(fix[f"permcrc_{n}"][0]), (n, which)
    for k in range(3):
        key = oracle.golden(f"linear_keys_t{k}_first4000").astype(np.float32)
        _, perm = oracle.sort_pairs(key.view(np.uint32), np.arange(key.size, dtype=np.uint32), "std")
        assert np.array_equal(perm, fix[f"perm_linear_keys_t{k}_first4000"])


@pytest.mark.parametrize("name", gl.names("gl_lines_"))
def test_overlay_lines(oracle, name):
    """row f3: Renderer::DrawGrid / DrawAxis / DrawLine (Renderer.cpp:41-215) through Shader/Lines/LineVert.GLSL + LineFrag.GLSL"""
    fix = gl.load(name)
    W, H = (int(x) for x in fix["size"])
    img = oracle.clear_image(W, H)
    for verts, col, width, strip in gl.line_sets(fix):
        oracle.draw_lines(img, verts, col, width, viewproj=fix["vp"], strip=strip)
    m = gl.check_lines(fix, img, name)
    print(name, m)
