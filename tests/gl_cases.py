"""The reference's GLSL programs EXECUTED (tests/golden/gl_*.npz, written by oracle/make_golden_gl.py through oracle/_ref/refgl: the
reference's shader files, unmodified, compiled and run by Mesa 23.2.1's llvmpipe in the build container) — and the bars a projected
record set, an image or a permutation has to meet against them.  Used for the CPU checker (test_oracle_gl.py) and for the HIP kernels
(test_gpu_gl.py) alike: the same function judges both.

What llvmpipe does differently from an exact float evaluation of the shader text, and what that costs here (measured):
  * exp(x) is exp2(x * log2 e) in float32: the product's rounding costs 1.2e-7 |x| relative (measured 2.4e-7 for |x| <= 1, 1.1e-5 at
    x = -100) -> alpha bar 2e-6 (CPU) / 5e-6 (device) + 2e-7 |ln p(t)|;
  * normalize()/inversesqrt(): rsqrt with one Newton step in some places -> the conic oSig agrees to ~2e-7 relative (bar 1e-5);
  * the rasteriser snaps window coordinates to 1/256 pixel (GL_SUBPIXEL_BITS = 8, as the reference's desktop GPUs do) before the edge
    test, this build tests |u|,|v| <= 0.5 in float at the pixel centre: a pixel whose centre lies within 1/128 pixel of a quad edge may
    be covered by one and not by the other.  Such a pixel differs by up to alpha * exp(-8) = 3.4e-4 * alpha per disagreeing splat.
    check_image() counts them, requires each to have such an edge, bounds their number and bounds everything else by 1e-4.
"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
QUAD = np.array([[0.5, 0.5], [0.5, -0.5], [-0.5, -0.5], [-0.5, 0.5]], np.float64)     # Geometry.h:44-49
CLEAR = np.array([0.18431373, 0.20784314, 0.25882353, 1.0], np.float32)              # Application.cpp:125
EDGE_PX = 1.0 / 128.0


def manifest():
    with open(os.path.join(GOLDEN, "manifest_gl.json")) as f:
        return json.load(f)


def load(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def names(prefix, kind=None):
    return sorted(k for k, v in manifest().items() if k.startswith(prefix) and (kind is None or v.get("kind") == kind))


def records(fix, name, golden):
    """the input records of a fixture: embedded (synthetic sets) or one of refgen's committed record files"""
    if "records" in fix:
        return fix["records"]
    src = manifest()[name]["source"].split(" ")[0].rstrip(",")
    if src.startswith("gl_"):                                   # the records embedded in another fixture
        return load(src)["records"]
    if src == "splat_draw_3d_in":
        import splat_draw_cases as sd
        return sd.verts72(golden(src))
    return golden(src)


def split_uniforms(fix):
    u = fix["uniforms"]
    return float(u[0]), float(u[1]), u[2:18].copy(), u[18:34].copy()


def as_got(cx, cy, a0x, a0y, a1x, a1y, alpha, r, g, b, valid):
    f = lambda a: np.asarray(a, np.float64)
    return {"cx": f(cx), "cy": f(cy), "a0x": f(a0x), "a0y": f(a0y), "a1x": f(a1x), "a1y": f(a1y), "alpha": np.asarray(alpha, np.float32),
            "rgb": np.stack([np.asarray(r, np.float32), np.asarray(g, np.float32), np.asarray(b, np.float32)], 1), "valid": np.asarray(valid) != 0}


def got_from_oracle(p):
    return as_got(p["cx"], p["cy"], p["a0x"], p["a0y"], p["a1x"], p["a1y"], p["alpha"], p["r"], p["g"], p["b"], p["valid"])


def got_from_device(p16):
    """rows of gs4d_debug_read_projected: cx,cy,a0x,a0y,a1x,a1y,alpha,r,g,b,rect0,rect1,hx,hy,valid,0 (include/gs4d.h)"""
    return as_got(p16[:, 0], p16[:, 1], p16[:, 2], p16[:, 3], p16[:, 4], p16[:, 5], p16[:, 6], p16[:, 7], p16[:, 8], p16[:, 9], p16[:, 14].view(np.uint32))


def check_vertex_stage(fix, kind, got, what, alpha_rtol=2e-6, conic=None):
    """got: as_got(...) in RECORD order.  conic: optional (n, 4) q00,q01,q10,q11 of the candidate (the CPU checker keeps it).
    Returns the measured maxima."""
    W, H = (int(x) for x in fix["size"])
    pos = fix["pos"].astype(np.float64)                         # (n, 4 corners, xyzw) clip coordinates
    n = pos.shape[0]
    faulty = fix["faulty"] > 0 if "faulty" in fix else np.zeros(n, bool)
    with np.errstate(invalid="ignore", divide="ignore"):
        w = pos[:, :, 3]
        ndc = pos[:, :, :3] / w[:, :, None]
        # the fixed-function clip keeps -w <= z <= w; every corner of a quad shares z and w (…Instanced.GLSL:145-147), so it keeps all or nothing
        inside = (np.abs(pos[:, :, 2]) <= w).all(axis=1) & (w > 0).all(axis=1) & np.isfinite(pos).all(axis=(1, 2))
    gl_valid = ~faulty & inside
    assert np.array_equal(gl_valid, got["valid"]), f"{what}: visible set differs from the GL's for records {np.flatnonzero(gl_valid != got['valid'])[:8]}"
    m = {"n": int(n), "visible": int(gl_valid.sum())}
    if not gl_valid.any():
        return m
    v = gl_valid
    win = np.stack([(ndc[v, :, 0] + 1.0) * (W / 2.0), (ndc[v, :, 1] + 1.0) * (H / 2.0)], 2)          # (nv, 4, 2) window coordinates
    ctr = win.mean(axis=1)
    off = win - ctr[:, None, :]
    # (1) centre: float32 carries ~6e-5 px at x ~ 1000; a few roundings apart
    m["centre_px"] = float(max(np.abs(ctr[:, 0] - got["cx"][v]).max(), np.abs(ctr[:, 1] - got["cy"][v]).max()))
    assert m["centre_px"] <= 5e-4, f"{what}: quad centre off by {m['centre_px']} px"
    # (2) the quad: the candidate's affine rows map window offsets to quad-local (u, v); its inverse gives where the corners must be
    A = np.stack([np.stack([got["a0x"][v], got["a0y"][v]], 1), np.stack([got["a1x"][v], got["a1y"][v]], 1)], 1)     # (nv, 2, 2): rows a0, a1
    Ainv = np.linalg.inv(A)
    want_off = np.einsum("nij,kj->nki", Ainv, QUAD)
    extent = np.abs(off).max(axis=(1, 2))
    d_corner = np.abs(want_off - off).max(axis=(1, 2))
    m["corner_px"] = float(d_corner.max()); m["corner_rel"] = float((d_corner / np.maximum(extent, 1e-30)).max())
    assert (d_corner <= 5e-4 + 2e-5 * extent).all(), f"{what}: quad corners off by {m['corner_px']} px ({m['corner_rel']} of the extent)"
    # (3) the fragment function: with the GL's own oSig and oFragPos, x.oSig.x / 2 at a corner must be 32 (u^2 + v^2) = 16, which is what
    #     the candidate evaluates from (u, v) alone (exp2(-32 log2e (u^2+v^2)), composite_common.h) — (2) pinned its (u, v) to the GL's corners
    x = fix["fragpos"].astype(np.float64)[v]
    S = fix["sig"].astype(np.float64)[v].reshape(-1, 2, 2)       # column-major mat2: S[c][r]
    half_d = 0.5 * np.einsum("nki,nij,nkj->nk", x, S, x)
    m["fragfn_rel"] = float(np.abs(half_d / 16.0 - 1.0).max())
    assert m["fragfn_rel"] <= 2e-5, f"{what}: x.oSig.x/2 at the quad corners is not 16 to {m['fragfn_rel']}"
    if conic is not None:
        q = np.asarray(conic, np.float64)[v]
        g = fix["sig"].astype(np.float64)[v]
        m["conic_rel"] = float((np.abs(q - g).max(axis=1) / np.abs(g).max(axis=1)).max())
        assert m["conic_rel"] <= 1e-5, f"{what}: conic off by {m['conic_rel']} relative"
    # (4) colour and opacity
    col = fix["color"][v]
    assert np.array_equal(got["rgb"][v].view(np.uint32), col[:, :3].view(np.uint32)), f"{what}: colour is not passed through bit for bit"
    a_gl = (fix["topac"][v].astype(np.float64) * col[:, 3]) if "topac" in fix else col[:, 3].astype(np.float64)
    a_got = got["alpha"][v].astype(np.float64)
    big = a_gl > 1e-30
    # exp(x) evaluated as exp2(x * log2(e)) in float32 (llvmpipe; the device's v_exp_f32 path alike) rounds the product: ~1.2e-7 |x| relative
    arg = np.abs(np.log(np.maximum(fix["topac"][v].astype(np.float64), 1e-300))) if "topac" in fix else np.zeros(int(v.sum()))
    rel = np.abs(a_got - a_gl)[big] / a_gl[big]
    m["alpha_rel"] = float(rel.max()) if big.any() else 0.0
    m["alpha_rel_near_one"] = float(rel[arg[big] <= 1.0].max()) if (arg[big] <= 1.0).any() else 0.0
    assert (rel <= alpha_rtol + 2e-7 * arg[big]).all(), f"{what}: alpha off by {m['alpha_rel']} relative"
    assert np.abs(a_got[~big]).max(initial=0.0) <= 1e-30
    return m


def gl_image(fix):
    W, H = (int(x) for x in fix["size"])
    img = np.empty((H, W, 4), np.float32)
    img[:] = CLEAR
    x0, y0, x1, y1 = (int(x) for x in fix["box"])
    if "crop16" in fix:         # an RGBA16 (unsigned normalised) attachment: see check_image
        img[:] = np.round(CLEAR.astype(np.float64) * 65535.0) / 65535.0
        img[y0:y1, x0:x1] = fix["crop16"].astype(np.float64) / 65535.0
    else:
        img[y0:y1, x0:x1] = fix["crop"]
    return img


def edge_pixels(xs, ys, got):
    """for each pixel (xs[k], ys[k]): (does a visible quad's edge pass within EDGE_PX of the pixel centre?,
    G = sum over the covering quads of alpha * |grad c| in 1/pixel: what a shift of the window coordinates by one pixel would change)"""
    v = got["valid"]
    cx, cy = got["cx"][v], got["cy"][v]
    a0x, a0y, a1x, a1y = got["a0x"][v], got["a0y"][v], got["a1x"][v], got["a1y"][v]
    alpha = got["alpha"][v].astype(np.float64)
    g0, g1 = np.hypot(a0x, a0y) * EDGE_PX, np.hypot(a1x, a1y) * EDGE_PX
    edge = np.zeros(len(xs), bool)
    grad = np.zeros(len(xs))
    for k, (x, y) in enumerate(zip(xs, ys)):
        dx, dy = (x + 0.5) - cx, (y + 0.5) - cy
        u, w = a0x * dx + a0y * dy, a1x * dx + a1y * dy
        au, aw = np.abs(u), np.abs(w)
        near = (au <= 0.5 + g0) & (aw <= 0.5 + g1) & ((np.abs(au - 0.5) <= g0) | (np.abs(aw - 0.5) <= g1))
        edge[k] = bool(near.any())
        cov = (au <= 0.5) & (aw <= 0.5)
        c = 64.0 * np.exp(-32.0 * (u[cov] ** 2 + w[cov] ** 2)) * alpha[cov]
        grad[k] = float((c * np.hypot(u[cov] * a0x[cov] + w[cov] * a1x[cov], u[cov] * a0y[cov] + w[cov] * a1y[cov])).sum())
    return edge, grad


def check_image(fix, img, got, what, tol=1e-4):
    """img (H, W, 4) float32, row 0 = bottom; got: as_got of the drawn records (any order).  Returns the measurements.
    A pixel may differ from the GL's by more than `tol` only
      (a) where a quad edge passes within 1/128 px of its centre (sub-pixel snapping, module docstring), or
      (b) by what 3 ulp of a float32 window coordinate move it: both sides are float32 evaluations of the same expressions and place a
          quad's centre a few roundings apart (check_vertex_stage measures <= 2.5e-4 px at 1080p); a splat one or two pixels wide turns
          that into G * shift, G = sum alpha |grad c| (measured: d / G <= 7e-5 px at 640 px, where the ulp is 3e-5)."""
    ref = gl_image(fix)
    assert img.shape == ref.shape, (img.shape, ref.shape)
    W = ref.shape[1]
    if "crop16" in fix:
        # blend pairs whose result leaves [0, 1] are drawn into a fixed-point RGBA16 attachment (it clamps like the reference's window; a float
        # one does not): every blend rounds to 1/65535, half a step (7.6e-6) per layer — the bar widens by 20 layers' worth
        tol = tol + 20 * 0.5 / 65535.0
    d = np.abs(img.astype(np.float64) - ref).max(axis=2)
    touched = int((np.abs(ref - CLEAR).max(axis=2) > 1e-5).sum())
    ys, xs = np.nonzero(d > tol)
    on_edge, grad = edge_pixels(xs, ys, got)
    shift = 3.0 * float(np.spacing(np.float32(W / 2.0)))
    steep = ~on_edge & (d[ys, xs] <= tol + grad * shift)
    bad = ~on_edge & ~steep
    m = {"linf": float(d.max()), "touched": touched, "beyond_tol": int(len(xs)), "beyond_tol_on_edge": int(on_edge.sum()), "beyond_tol_steep": int(steep.sum())}
    d_off = d.copy(); d_off[ys[on_edge], xs[on_edge]] = 0.0
    m["linf_off_edge"] = float(d_off.max())
    assert not bad.any(), f"{what}: {int(bad.sum())} pixels differ by more than {tol} with no quad edge and no gradient to explain it, e.g. {list(zip(xs[bad][:4], ys[bad][:4]))} ({d[ys, xs][bad][:4]}, G {grad[bad][:4]})"
    assert len(xs) <= max(8, touched // 500), f"{what}: {len(xs)} pixels beyond {tol} of {touched} touched"
    assert m["linf"] <= 2e-3, f"{what}: an edge pixel differs by {m['linf']}"
    assert m["linf_off_edge"] <= 5e-4, f"{what}: off-edge difference {m['linf_off_edge']}"
    assert touched > 100, f"{what}: empty image"
    return m


def grid_vertices(width, height, dx, dy):
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


def line_sets(fix):
    for k in range(int(fix["nsets"][0])):
        st = fix[f"style{k}"]
        yield fix[f"verts{k}"], tuple(float(x) for x in st[:4]), float(st[4]), bool(st[5])


def gl_lines_image(fix):
    W, H = (int(x) for x in fix["size"])
    img = np.empty((H, W, 4), np.float32)
    img[:] = CLEAR
    x0, y0, x1, y1 = (int(x) for x in fix["box"])
    img[y0:y1, x0:x1] = fix["palette"][fix["index"].astype(np.int64)]
    return img


def check_lines(fix, img, what):
    """Overlay lines against the reference's line programs run by llvmpipe.  The rule of csrc/lines.hip reproduces the GL's pixels except
    where a line passes within a sub-pixel step of a pixel boundary at the pixel-centre crossing that decides its row/column (the GL snaps
    end points to 1/256 px): there the fragment lands in the neighbouring pixel.  Bars: at most 0.2 % of the line pixels differ, every
    differing pixel has a differing 8-neighbour (a swapped pair) or lies at a segment end, and the summed colour over the frame agrees
    (fragments move, none appear or vanish beyond the segment ends)."""
    ref = gl_lines_image(fix)
    d = np.abs(img.astype(np.float64) - ref).max(axis=2)
    line_px = int((np.abs(ref - CLEAR).max(axis=2) > 0).sum())
    bad = d > 1e-6
    m = {"line_pixels": line_px, "differing": int(bad.sum()), "linf_elsewhere": float(d[~bad].max())}
    assert line_px > 1000, f"{what}: no lines in the fixture"
    assert bad.sum() <= max(8, line_px // 500), f"{what}: {int(bad.sum())} of {line_px} line pixels differ"
    ys, xs = np.nonzero(bad)
    H, W = bad.shape
    lonely = 0
    for x, y in zip(xs, ys):
        nb = bad[max(0, y - 1):min(H, y + 2), max(0, x - 1):min(W, x + 2)].sum() - 1
        lonely += int(nb == 0)
    m["lonely"] = lonely
    assert lonely <= 8, f"{what}: {lonely} differing pixels without a differing neighbour"
    return m
