"""ctypes wrapper around oracle/libgs4d_oracle.so — the CPU checker.

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package never imports this module.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
_LIB = None

PROJ_DTYPE = np.dtype([
    ("cx", "f4"), ("cy", "f4"), ("a0x", "f4"), ("a0y", "f4"), ("a1x", "f4"), ("a1y", "f4"),
    ("r", "f4"), ("g", "f4"), ("b", "f4"), ("alpha", "f4"), ("hx", "f4"), ("hy", "f4"), ("valid", "u4"),
    ("e0x", "f4"), ("e0y", "f4"), ("e1x", "f4"), ("e1y", "f4"), ("s0", "f4"), ("s1", "f4"),
    ("q00", "f4"), ("q01", "f4"), ("q10", "f4"), ("q11", "f4"),
])

MODE_4D, MODE_4D_DIRECT, MODE_3D, MODE_2D = 0, 1, 2, 3


def build():
    if os.environ.get("GS4D_ORACLE_SO"):                    # tests/san_driver.py: the AddressSanitizer / UBSan build of the checker
        return os.environ["GS4D_ORACLE_SO"]
    so = os.path.join(ORACLE_DIR, "libgs4d_oracle.so")
    src = os.path.join(ORACLE_DIR, "gs4d_oracle.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        assert _LIB.gs4do_proj_size() == PROJ_DTYPE.itemsize, (_LIB.gs4do_proj_size(), PROJ_DTYPE.itemsize)
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def golden(name):
    man = {}
    for mf in ("manifest.json", "manifest_draw.json"):      # refgen's fixtures; refdraw's (oracle/ref/refdraw_main.cpp)
        with open(os.path.join(GOLDEN_DIR, mf)) as f:
            man.update(json.load(f))
    ent = man[name]
    if not isinstance(ent, dict) or "dtype" not in ent:
        return ent
    dt = {"f32": np.float32, "u32": np.uint32}[ent["dtype"]]
    a = np.fromfile(os.path.join(GOLDEN_DIR, name + ".bin"), dtype=dt)
    assert a.size == ent["count"]
    return a.reshape(-1, ent["cols"]) if ent["cols"] > 1 else a


# ---- camera -------------------------------------------------------------------------------------
def look_at(eye, orientation, up=(0, 1, 0)):
    out = np.zeros(16, np.float32)
    lib().gs4do_look_at(_p(_f32(eye)), _p(_f32(orientation)), _p(_f32(up)), _p(out))
    return out


def perspective(fov_deg, w, h, znear, zfar):
    out = np.zeros(16, np.float32)
    lib().gs4do_perspective(C.c_float(fov_deg), C.c_int(w), C.c_int(h), C.c_float(znear), C.c_float(zfar), _p(out))
    return out


# ---- parameterisation ---------------------------------------------------------------------------
def quat_look_at(n, up=(0, 1, 0)):
    out = np.zeros(4, np.float32)
    lib().gs4do_quat_look_at(_p(_f32(n)), _p(_f32(up)), _p(out))
    return out


def splat3d_cov(q, s):
    out = np.zeros(9, np.float32)
    lib().gs4do_splat3d_cov(_p(_f32(q)), _p(_f32(s)), _p(out))
    return out


def splat4d_cov(q, s, life, fade, d):
    out = np.zeros(16, np.float32)
    lib().gs4do_splat4d_cov(_p(_f32(q)), _p(_f32(s)), C.c_float(life), C.c_float(fade), _p(_f32(d)), _p(out))
    return out


def splat4d_cov2q(q0, q1, s4):
    out = np.zeros(16, np.float32)
    lib().gs4do_splat4d_cov2q(_p(_f32(q0)), _p(_f32(q1)), _p(_f32(s4)), _p(out))
    return out


# ---- key / sort -----------------------------------------------------------------------------------
def keygen(records, t, cam):
    rec = _f32(records).reshape(-1, 24)
    n = rec.shape[0]
    idx = np.zeros(n, np.uint32)
    key = np.zeros(n, np.float32)
    lib().gs4do_keygen(_p(rec), C.c_size_t(n), C.c_float(t), _p(_f32(cam)), _p(idx), _p(key))
    return idx, key


def keygen_viewz(records, t, view):
    """The build's extra key mode GS4D_KEY_VIEW_Z (no reference counterpart): 1 / view-space depth of the conditioned mean."""
    rec = _f32(records).reshape(-1, 24)
    n = rec.shape[0]
    idx = np.zeros(n, np.uint32)
    key = np.zeros(n, np.float32)
    lib().gs4do_keygen_viewz(_p(rec), C.c_size_t(n), C.c_float(t), _p(_f32(view)), _p(idx), _p(key))
    return idx, key


def sort_pairs(keys_u32, vals_u32, which="lsd"):
    k = np.array(keys_u32, dtype=np.uint32, copy=True)
    v = np.array(vals_u32, dtype=np.uint32, copy=True)
    fn = {"lsd": lib().gs4do_sort_pairs, "std": lib().gs4do_sort_pairs_std, "glsl": lib().gs4do_glsl_radix_sort}[which]
    fn(_p(k), _p(v), C.c_size_t(k.size))
    return k, v


# ---- preprocess / composite ---------------------------------------------------------------------
def preprocess(mode, data, view, proj, W, H, t=0.0, min_opacity=0.0):
    view, proj = _f32(view), _f32(proj)
    if mode in (MODE_4D, MODE_4D_DIRECT):
        rec = _f32(data).reshape(-1, 24)
        out = np.zeros(rec.shape[0], PROJ_DTYPE)
        lib().gs4do_preprocess_4d(_p(rec), C.c_size_t(rec.shape[0]), C.c_float(t), C.c_float(min_opacity), _p(view), _p(proj), C.c_int(W), C.c_int(H), _p(out))
    elif mode == MODE_3D:
        rec = _f32(data).reshape(-1, 72)
        out = np.zeros(rec.shape[0], PROJ_DTYPE)
        lib().gs4do_preprocess_3d(_p(rec), C.c_size_t(rec.shape[0]), _p(view), _p(proj), C.c_int(W), C.c_int(H), _p(out))
    elif mode == MODE_2D:
        rec = _f32(data).reshape(-1, 12)
        out = np.zeros(rec.shape[0], PROJ_DTYPE)
        lib().gs4do_preprocess_2d(_p(rec), C.c_size_t(rec.shape[0]), _p(view), _p(proj), C.c_int(W), C.c_int(H), _p(out))
    else:
        raise ValueError(mode)
    return out


BLEND_OVER = (0x0302, 0x0303)          # glBlendFunc(GL_SRC_ALPHA, GL_ONE_MINUS_SRC_ALPHA), Application.cpp:137-138


def composite(proj_arr, order, frag_mode, W, H, rgba, nthreads=8, blend=BLEND_OVER):
    assert rgba.dtype == np.float32 and rgba.size == W * H * 4 and rgba.flags.c_contiguous
    if order is not None:
        order = np.ascontiguousarray(order, dtype=np.uint32)
        n = order.size
    else:
        n = proj_arr.size
    fm = MODE_4D if frag_mode == MODE_4D_DIRECT else frag_mode
    lib().gs4do_composite_blend(_p(proj_arr), _p(order) if order is not None else None, C.c_size_t(n), C.c_int(fm), C.c_int(W), C.c_int(H), _p(rgba), C.c_int(nthreads),
                                C.c_int(blend[0]), C.c_int(blend[1]))
    return rgba


def draw_lines(rgba, verts, color, width=1.0, viewproj=None, strip=False, blend=BLEND_OVER):
    """Overlay lines blended into `rgba` (H, W, 4) in place; verts (n, 3) with viewproj or (n, 2) NDC."""
    H, W = rgba.shape[:2]
    v = _f32(verts)
    dims = v.shape[-1]
    vp = _f32(viewproj) if viewproj is not None else np.eye(4, dtype=np.float32).reshape(-1)
    lib().gs4do_draw_lines_blend(_p(rgba), C.c_int(W), C.c_int(H), _p(v), C.c_size_t(v.size // dims), C.c_int(dims), C.c_int(1 if strip else 0), _p(vp), _p(_f32(color)), C.c_float(width),
                                 C.c_int(blend[0]), C.c_int(blend[1]))
    return rgba


CLEAR = np.array([0.18431373, 0.20784314, 0.25882353, 1.0], np.float32)  # Application.cpp:125


def clear_image(W, H, clear=CLEAR):
    img = np.empty((H, W, 4), np.float32)
    img[:] = _f32(clear)
    return img


def render_4d(records, do_sort, t, min_opacity, cam, view, proj, W, H, clear=CLEAR, nthreads=8):
    rec = _f32(records).reshape(-1, 24)
    n = rec.shape[0]
    img = np.empty((H, W, 4), np.float32)
    perm = np.zeros(n, np.uint32)
    ms = (C.c_double * 4)()
    lib().gs4do_render_4d(_p(rec), C.c_size_t(n), C.c_int(1 if do_sort else 0), C.c_float(t), C.c_float(min_opacity), _p(_f32(cam)), _p(_f32(view)), _p(_f32(proj)),
                          C.c_int(W), C.c_int(H), _p(_f32(clear)), _p(img), _p(perm), C.c_int(nthreads), ms)
    return img, perm, list(ms)
