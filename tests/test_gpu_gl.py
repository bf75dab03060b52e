"""GPU: the HIP kernels, through the C ABI, against the reference's GLSL programs EXECUTED (tests/golden/gl_*.npz: the reference's shader
files run by Mesa llvmpipe in the build container, oracle/ref/refgl_main.cpp + oracle/make_golden_gl.py).  Nothing of the CPU checker's
arithmetic is involved in the comparison: the projected records read back through gs4d_debug_read_projected, the framebuffer and the sort's
permutation are judged against what the reference's own shaders produced (gl_cases.py states the bars and what llvmpipe rounds differently).

  vertex stage   Splat4DVertexShaderInstanced.GLSL:81-150, Splat3DVertexShaderFull.GLSL:43-98, Splat2DVSI.GLSL:59-94   (transform feedback)
  images         + Splat4DFragShader.GLSL:16-31 / 3D / 2D, the rasteriser, the blend of Application.cpp:150-154            (RGBA32F attachment)
  lines          Shader/Lines/LineVert.GLSL + LineFrag.GLSL, Renderer.cpp:41-215                                          (RGBA32F attachment)
  sort           radix_sort_{count,local_offsets,reorder}.comp.glsl under radix_sort.hpp:258-392                          (permutation)
"""
import zlib

import numpy as np
import pytest

import gl_cases as gl

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "ordered"])
def draw_path(request, monkeypatch):
    if request.param == "ordered":
        monkeypatch.setenv("GS4D_DRAW_PATH", "ordered")
    else:
        monkeypatch.delenv("GS4D_DRAW_PATH", raising=False)
    return request.param


def issue(ctx, gs4d, kind, rec, n, order=None):
    """the draw of one fixture; returns the buffers to delete"""
    bufs = [ctx.buffer(rec)]
    if kind == "3d":
        ctx.set_mode(gs4d.MODE_3D_FULL)
        ctx.draw_quads(bufs[0], n)
    elif kind == "2d":
        ctx.set_mode(gs4d.MODE_2D)
        ctx.bind(1, bufs[0])
        ctx.draw_instanced(n)
    elif order is None:
        ctx.set_mode(gs4d.MODE_4D_DIRECT)
        ctx.bind(1, bufs[0])
        ctx.draw_instanced(n)
    else:
        ib = ctx.buffer(np.ascontiguousarray(order, np.uint32))
        bufs.append(ib)
        ctx.set_mode(gs4d.MODE_4D_SORTED)
        ctx.bind(1, ib)
        ctx.bind(2, bufs[0])
        ctx.draw_instanced(n)
    return bufs


@pytest.mark.parametrize("name", gl.names("gl_vs_"))
def test_vertex_stage(gs4d, oracle, name):
    fix = gl.load(name)
    kind = gl.manifest()[name]["kind"]
    rec = np.ascontiguousarray(gl.records(fix, name, oracle.golden), np.float32)
    n = rec.shape[0]
    t, mo, view, proj = gl.split_uniforms(fix)
    W, H = (int(x) for x in fix["size"])
    ctx = gs4d.Context(W, H)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.clear()
    ctx.set_uniforms(time=t, min_opacity=mo, view=view, proj=proj)
    bufs = issue(ctx, gs4d, kind, rec, n)
    p16 = ctx.debug_projected(n)
    for b in bufs:
        ctx.delete(b)
    ctx.close()
    m = gl.check_vertex_stage(fix, kind, gl.got_from_device(p16), name, alpha_rtol=5e-6)
    print(name, m)


@pytest.mark.parametrize("name", gl.names("gl_img_"))
def test_image(gs4d, oracle, draw_path, name):
    fix = gl.load(name)
    kind = gl.manifest()[name]["kind"]
    rec = np.ascontiguousarray(gl.records(fix, name, oracle.golden), np.float32)
    n = rec.shape[0]
    t, mo, view, proj = gl.split_uniforms(fix)
    W, H = (int(x) for x in fix["size"])
    ctx = gs4d.Context(W, H)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    if "blend" in fix:
        ctx.set_blend(int(fix["blend"][0]), int(fix["blend"][1]))
    ctx.clear()
    ctx.set_uniforms(time=t, min_opacity=mo, view=view, proj=proj)
    bufs = []
    if "order" in fix:
        # Scenes.h:312-339 through the C ABI: key loop -> sort -> bind -> Draw; the permutation must be the reference's compute sort's
        db, kb, ib = ctx.buffer(rec), ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
        bufs = [db, kb, ib]
        ctx.keygen(db, t, tuple(float(x) for x in fix["cam"]), kb, ib, n)
        ctx.sort_pairs(kb, ib, n)
        ctx.set_mode(gs4d.MODE_4D_SORTED)
        ctx.bind(1, ib)
        ctx.bind(2, db)
        ctx.draw_instanced(n)
        perm = ctx.read(ib, np.uint32, n)
        assert np.array_equal(perm, fix["order"]), f"{name}: permutation differs from the reference's compute sort"
    else:
        bufs = issue(ctx, gs4d, kind, rec, n)
    img = ctx.read_pixels()
    p16 = ctx.debug_projected(n)
    for b in bufs:
        ctx.delete(b)
    ctx.close()
    m = gl.check_image(fix, img, gl.got_from_device(p16), f"{name} [{draw_path}]")
    print(name, draw_path, m)


@pytest.mark.parametrize("name", gl.names("gl_lines_"))
def test_overlay_lines(gs4d, name):
    """row f3: csrc/lines.hip against Shader/Lines/LineVert.GLSL + LineFrag.GLSL run by the GL (grid, axes, unit line, path, strip)"""
    fix = gl.load(name)
    W, H = (int(x) for x in fix["size"])
    ctx = gs4d.Context(W, H)
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.clear()
    for verts, col, width, strip in gl.line_sets(fix):
        ctx.draw_lines(verts, col, width, viewproj=fix["vp"], strip=strip)
    img = ctx.read_pixels()
    ctx.close()
    m = gl.check_lines(fix, img, name)
    print(name, m)


def test_sort_permutations(gs4d):
    fix = gl.load("gl_sort")
    ctx = gs4d.Context(64, 64)

    def gpu_sort(keys_u32):
        n = keys_u32.size
        kb, ib = ctx.buffer(np.ascontiguousarray(keys_u32, np.uint32)), ctx.buffer(np.arange(n, dtype=np.uint32))
        ctx.sort_pairs(kb, ib, n)
        ks, perm = ctx.read(kb, np.uint32, n), ctx.read(ib, np.uint32, n)
        ctx.delete(kb); ctx.delete(ib)
        return ks, perm

    for n in (5, 257, 2049, 100003):
        keys = fix[f"keys_{n}"]
        ks, perm = gpu_sort(keys)
        assert np.array_equal(ks, np.sort(keys))
        if f"perm_{n}" in fix:
            assert np.array_equal(perm, fix[f"perm_{n}"]), n
        else:
            assert zlib.crc32(perm.tobytes()) == int(fix[f"permcrc_{n}"][0]), n
    import oracle_lib
    for k in range(3):
        key = oracle_lib.golden(f"linear_keys_t{k}_first4000").astype(np.float32)
        _, perm = gpu_sort(key.view(np.uint32))
        assert np.array_equal(perm, fix[f"perm_linear_keys_t{k}_first4000"])
    ctx.close()
