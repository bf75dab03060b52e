"""4dgaussiansplatrendering_amd — ctypes binding of libgs4d.so (include/gs4d.h).

The product is the HIP library; this module is plumbing for the tests and bench.py.  It has no CPU
fallback: importing it without the built library, or creating a Context without a GPU, raises.

(The directory name starts with a digit, so import it with
``importlib.import_module("4dgaussiansplatrendering_amd")``.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgs4d.so")

MODE_4D_SORTED, MODE_4D_DIRECT, MODE_3D_FULL, MODE_2D = 0, 1, 2, 3
U_TIME, U_MIN_OPACITY = 0, 1
U_VIEW, U_PROJ = 0, 1
# glBlendFunc factors (GL enum values): the reference's blend menu, DebugMenus.h:41-59
ZERO, ONE, SRC_COLOR, ONE_MINUS_SRC_COLOR, SRC_ALPHA, ONE_MINUS_SRC_ALPHA = 0, 1, 0x0300, 0x0301, 0x0302, 0x0303
DST_ALPHA, ONE_MINUS_DST_ALPHA, DST_COLOR, ONE_MINUS_DST_COLOR = 0x0304, 0x0305, 0x0306, 0x0307
CONSTANT_COLOR, ONE_MINUS_CONSTANT_COLOR, CONSTANT_ALPHA, ONE_MINUS_CONSTANT_ALPHA = 0x8001, 0x8002, 0x8003, 0x8004
KEY_REF_INV_EUCLID, KEY_VIEW_Z = 0, 1
STAGES = ("keygen", "sort", "preprocess", "binning", "pairsort", "composite")
CLEAR_COLOR = (0.18431373, 0.20784314, 0.25882353, 1.0)   # Application.cpp:125


class Gs4dError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make lib` (or __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    vp, sz, u32, i32, f32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_float
    sig = {
        "gs4d_create": (i32, [i32, i32, i32, C.POINTER(vp)]),
        "gs4d_destroy": (None, [vp]),
        "gs4d_resize": (i32, [vp, i32, i32]),
        "gs4d_last_error": (C.c_char_p, [vp]),
        "gs4d_buffer_create": (i32, [vp, vp, sz, C.POINTER(u32)]),
        "gs4d_buffer_subdata": (i32, [vp, u32, sz, vp, sz]),
        "gs4d_buffer_read": (i32, [vp, u32, sz, vp, sz]),
        "gs4d_buffer_destroy": (i32, [vp, u32]),
        "gs4d_buffer_device_ptr": (i32, [vp, u32, C.POINTER(vp), C.POINTER(sz)]),
        "gs4d_buffer_invalidate": (i32, [vp, u32]),
        "gs4d_bind_storage": (i32, [vp, i32, u32]),
        "gs4d_set_mode": (i32, [vp, i32]),
        "gs4d_set_uniform_1f": (i32, [vp, i32, f32]),
        "gs4d_set_uniform_mat4": (i32, [vp, i32, vp]),
        "gs4d_set_clear_color": (i32, [vp, vp]),
        "gs4d_set_blend": (i32, [vp, i32, i32]),
        "gs4d_clear": (i32, [vp]),
        "gs4d_sort_pairs": (i32, [vp, u32, u32, sz]),
        "gs4d_keygen": (i32, [vp, u32, f32, vp, u32, u32, sz, i32]),
        "gs4d_draw_instanced": (i32, [vp, sz]),
        "gs4d_draw_quads": (i32, [vp, u32, sz]),
        "gs4d_draw_lines": (i32, [vp, vp, sz, i32, i32, vp, vp, f32]),
        "gs4d_host_camera_input": (None, [vp, vp, vp, vp]),
        "gs4d_host_camera_rotate": (None, [vp, C.c_double, C.c_double]),
        "gs4d_host_camera_look_at_point": (None, [vp, vp]),
        "gs4d_host_camera_viewport": (None, [i32, i32, vp]),
        "gs4d_host_camera_focal": (None, [f32, i32, i32, vp]),
        "gs4d_read_pixels": (i32, [vp, vp, sz]),
        "gs4d_read_pixels_device": (i32, [vp, vp, sz]),
        "gs4d_read_pixels_rgba8_device": (i32, [vp, vp, sz]),
        "gs4d_read_frame_rgba8_device": (i32, [vp, i32, vp, sz]),
        "gs4d_read_frame_rgba8_device_after": (i32, [vp, i32, vp, sz, vp]),
        "gs4d_set_tile_shard": (i32, [vp, i32, i32]),
        "gs4d_band_rows": (i32, [vp, vp]),
        "gs4d_read_band_rgba8_device": (i32, [vp, vp, sz]),
        "gs4d_set_stream": (i32, [vp, vp]),
        "gs4d_finish": (i32, [vp]),
        "gs4d_set_profiling": (i32, [vp, i32]),
        "gs4d_get_timings": (i32, [vp, vp]),
        "gs4d_get_timeline": (i32, [vp, vp, i32, vp]),
        "gs4d_get_stats": (i32, [vp, vp]),
        "gs4d_debug_read_projected": (i32, [vp, vp, sz]),
        "gs4d_host_look_at": (None, [vp, vp, vp, vp]),
        "gs4d_host_perspective": (None, [f32, i32, i32, f32, f32, vp]),
        "gs4d_host_quat_look_at": (None, [vp, vp, vp]),
        "gs4d_host_splat3d_cov": (None, [vp, vp, vp]),
        "gs4d_host_splat3d_mesh": (None, [vp, vp, vp, vp, vp]),
        "gs4d_host_splat2d_sigma_inv": (None, [vp, f32, f32, vp]),
        "gs4d_host_gaussians2d_record": (None, [f32, f32, f32, f32, f32, vp, vp]),
        "gs4d_host_splat4d_cov": (None, [vp, vp, f32, f32, vp, vp]),
        "gs4d_host_splat4d_cov2q": (None, [vp, vp, vp, vp]),
        "gs4d_host_build_records_3d": (None, [sz, vp, vp, vp, vp, vp]),
        "gs4d_host_build_records_4d": (None, [sz, vp, vp, vp, vp, vp, vp, vp, vp]),
        "gs4d_host_scene_linear": (None, [sz, vp, i32, f32, f32, vp, f32, f32, f32, vp]),
        "gs4d_host_scene_nonlinear": (None, [sz, vp, i32, f32, f32, f32, vp, f32, f32, f32, sz, vp]),
        "gs4d_host_scene_rotation": (None, [sz, vp, i32, f32, f32, vp, f32, f32, f32, sz, vp]),
        "gs4d_host_scene_combined": (None, [sz, vp, i32, f32, f32, f32, f32, f32, vp, f32, f32, f32, sz, vp]),
        "gs4d_host_scene_broken": (None, [sz, vp, i32, f32, vp, f32, f32, f32, sz, vp]),
        "gs4d_host_scene_square": (None, [sz, vp, i32, f32, f32, vp, f32, f32, f32, sz, vp]),
        "gs4d_host_parse_vdata": (C.c_long, [C.c_char_p, vp, sz]),
        "gs4d_host_parse_sd": (C.c_long, [C.c_char_p, f32, vp, sz]),
        "gs4d_host_write_png": (i32, [C.c_char_p, vp, i32, i32]),
        "gs4d_version": (C.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export what gs4d.h declares
        fn.restype, fn.argtypes = res, args
    return lib, tuple(sig)


_lib, EXPORTS = _load()


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- host-side parameterisation (CPU code in libgs4d.so; Splat.h / Camera.cpp mirror) ---------------
def look_at(eye, orientation, up=(0.0, 1.0, 0.0)):
    out = np.zeros(16, np.float32)
    _lib.gs4d_host_look_at(_ptr(_f32(eye)), _ptr(_f32(orientation)), _ptr(_f32(up)), _ptr(out))
    return out


def perspective(fov_deg, width, height, znear, zfar):
    out = np.zeros(16, np.float32)
    _lib.gs4d_host_perspective(fov_deg, width, height, znear, zfar, _ptr(out))
    return out


def quat_look_at(direction, up=(0.0, 1.0, 0.0)):
    out = np.zeros(4, np.float32)
    _lib.gs4d_host_quat_look_at(_ptr(_f32(direction)), _ptr(_f32(up)), _ptr(out))
    return out


def splat3d_cov(q_wxyz, scale3):
    out = np.zeros(9, np.float32)
    _lib.gs4d_host_splat3d_cov(_ptr(_f32(q_wxyz)), _ptr(_f32(scale3)), _ptr(out))
    return out


def splat4d_cov(q_wxyz, scale3, lifetime, fade, dir3):
    out = np.zeros(16, np.float32)
    _lib.gs4d_host_splat4d_cov(_ptr(_f32(q_wxyz)), _ptr(_f32(scale3)), lifetime, fade, _ptr(_f32(dir3)), _ptr(out))
    return out


def splat4d_cov2q(q0, q1, scale4):
    out = np.zeros(16, np.float32)
    _lib.gs4d_host_splat4d_cov2q(_ptr(_f32(q0)), _ptr(_f32(q1)), _ptr(_f32(scale4)), _ptr(out))
    return out


def build_records_3d(pos3, q_wxyz, scale3, rgba):
    pos3, q, s, col = _f32(pos3).reshape(-1, 3), _f32(q_wxyz).reshape(-1, 4), _f32(scale3).reshape(-1, 3), _f32(rgba).reshape(-1, 4)
    n = pos3.shape[0]
    assert q.shape[0] == n and s.shape[0] == n and col.shape[0] == n
    rec = np.empty((n, 24), np.float32)
    _lib.gs4d_host_build_records_3d(n, _ptr(pos3), _ptr(q), _ptr(s), _ptr(col), _ptr(rec))
    return rec


def build_records_4d(pos4, q_wxyz, scale3, lifetime, fade, dir3, rgba):
    pos4, q, s = _f32(pos4).reshape(-1, 4), _f32(q_wxyz).reshape(-1, 4), _f32(scale3).reshape(-1, 3)
    life, fd, d, col = _f32(lifetime).reshape(-1), _f32(fade).reshape(-1), _f32(dir3).reshape(-1, 3), _f32(rgba).reshape(-1, 4)
    n = pos4.shape[0]
    assert all(x.shape[0] == n for x in (q, s, life, fd, d, col))
    rec = np.empty((n, 24), np.float32)
    _lib.gs4d_host_build_records_4d(n, _ptr(pos4), _ptr(q), _ptr(s), _ptr(life), _ptr(fd), _ptr(d), _ptr(col), _ptr(rec))
    return rec


def scene_linear(verts6, steps=50, time_multiplier=1.0, object_scale=5.0, splat_scale=(4.0, 4.0, 1.0), lifetime=1.0, fade=0.5, speed=1.0):
    """LinearMotion::init records (Scenes.h:258-279) with the class defaults (Scenes.h:186-201)."""
    v = _f32(verts6).reshape(-1, 6)
    rec = np.empty((v.shape[0] * steps, 24), np.float32)
    _lib.gs4d_host_scene_linear(v.shape[0], _ptr(v), steps, time_multiplier, object_scale, _ptr(_f32(splat_scale)), lifetime, fade, speed, _ptr(rec))
    return rec


def scene_nonlinear(verts6, steps=92, angle_multiplier=4.0, radius=20.0, object_scale=5.0, splat_scale=(4.0, 4.0, 1.0), lifetime=1.0, fade=0.5, speed=20.0,
                    max_records=None):
    """NonLinearMotion::init records (Scenes.h:517-545) with the class defaults (Scenes.h:451-467)."""
    v = _f32(verts6).reshape(-1, 6)
    n = v.shape[0] * steps if max_records is None else min(max_records, v.shape[0] * steps)
    rec = np.empty((n, 24), np.float32)
    _lib.gs4d_host_scene_nonlinear(v.shape[0], _ptr(v), steps, angle_multiplier, radius, object_scale, _ptr(_f32(splat_scale)), lifetime, fade, speed, n, _ptr(rec))
    return rec


def _scene_out(verts6, steps, max_records):
    v = _f32(verts6).reshape(-1, 6)
    n = v.shape[0] * steps if max_records is None else min(max_records, v.shape[0] * steps)
    return v, n, np.empty((n, 24), np.float32)


def scene_rotation(verts6, steps=92, angle_multiplier=4.0, object_scale=5.0, splat_scale=(4.0, 4.0, 1.0), lifetime=0.6, fade=0.5, speed=5.0, max_records=None):
    """RotationMotion::init records (Scenes.h:775-803) with the class defaults (Scenes.h:711-727).  Camera (0,60,30) / (0,-1,-0.5)."""
    v, n, rec = _scene_out(verts6, steps, max_records)
    _lib.gs4d_host_scene_rotation(v.shape[0], _ptr(v), steps, angle_multiplier, object_scale, _ptr(_f32(splat_scale)), lifetime, fade, speed, n, _ptr(rec))
    return rec


def scene_combined(verts6, steps=65, angle_multiplier=8.0, lin_multiplier=8.0, amplitude=1.0, frequency=0.15, object_scale=5.0, splat_scale=(4.0, 4.0, 0.0),
                   lifetime=1.0, fade=0.5, speed=1.0, max_records=None):
    """CombinedMotion::init records (Scenes.h:1035-1068) with the class defaults (Scenes.h:959-976).  Camera (50,90,90) / (0,-1,-1)."""
    v, n, rec = _scene_out(verts6, steps, max_records)
    _lib.gs4d_host_scene_combined(v.shape[0], _ptr(v), steps, angle_multiplier, lin_multiplier, amplitude, frequency, object_scale, _ptr(_f32(splat_scale)),
                                  lifetime, fade, speed, n, _ptr(rec))
    return rec


def scene_broken(verts6, steps=92, object_scale=5.0, splat_scale=(4.0, 4.0, 1.0), lifetime=1.0, fade=0.5, speed=1.0, max_records=None):
    """BrokenMotion::init records (Scenes.h:1965-1989) with the class defaults (Scenes.h:1899-1912).  Camera (0,60,60) / (0,-1,-1)."""
    v, n, rec = _scene_out(verts6, steps, max_records)
    _lib.gs4d_host_scene_broken(v.shape[0], _ptr(v), steps, object_scale, _ptr(_f32(splat_scale)), lifetime, fade, speed, n, _ptr(rec))
    return rec


def scene_square(verts6, steps=92, square_size=40.0, object_scale=5.0, splat_scale=(4.0, 4.0, 1.0), lifetime=1.0, fade=0.5, speed=1.0, max_records=None):
    """SquareMotion::init records (Scenes.h:2216-2259) with the class defaults (Scenes.h:2151-2165).  Camera (0,60,60) / (0,-1,-1)."""
    v, n, rec = _scene_out(verts6, steps, max_records)
    _lib.gs4d_host_scene_square(v.shape[0], _ptr(v), steps, square_size, object_scale, _ptr(_f32(splat_scale)), lifetime, fade, speed, n, _ptr(rec))
    return rec


def parse_vdata(path, cap_vertices=1 << 20):
    buf = np.empty((cap_vertices, 6), np.float32)
    n = _lib.gs4d_host_parse_vdata(os.fsencode(path), _ptr(buf), cap_vertices)
    if n < 0:
        raise FileNotFoundError(path)
    return buf[:min(n, cap_vertices)].copy()


def splat3d_mesh(pos3, q_wxyz, scale3, color4):
    """Splat3D::GetSplatMesh (Splat.h:433-447): (4, 18) float32 = four 72-byte vertices {corner, position, colour, Sigma3}."""
    out = np.empty((4, 18), np.float32)
    _lib.gs4d_host_splat3d_mesh(_ptr(_f32(pos3)), _ptr(_f32(q_wxyz)), _ptr(_f32(scale3)), _ptr(_f32(color4)), _ptr(out))
    return out


def splat2d_sigma_inv(v0, l0, l1):
    """Splat2D::CalcAndSetSigma (Splat.h:576-582): inverse 2x2 covariance, column-major."""
    out = np.empty(4, np.float32)
    _lib.gs4d_host_splat2d_sigma_inv(_ptr(_f32(v0)), l0, l1, _ptr(out))
    return out


def gaussians2d_record(angle, s0, s1, px, py, rgb):
    """One 48-byte record of the Gaussians2D scene (Scenes.h:1490-1496) for GS4D_MODE_2D."""
    out = np.empty(12, np.float32)
    _lib.gs4d_host_gaussians2d_record(angle, s0, s1, px, py, _ptr(_f32(rgb)), _ptr(out))
    return out


def parse_sd(path, object_scale=1.0, cap_records=1 << 22):
    """`.sd` splat file -> (n, 24) records (VDataParser.h:60-123 + ObjectDisplay::init, Scenes.h:2483-2491; its object scale defaults to 1)."""
    buf = np.empty((cap_records, 24), np.float32)
    n = _lib.gs4d_host_parse_sd(os.fsencode(path), object_scale, _ptr(buf), cap_records)
    if n < 0:
        raise FileNotFoundError(path)
    return buf[:min(n, cap_records)].copy()


class CameraState(C.Structure):
    """gs4d_camera_state (include/gs4d.h): the Camera of Camera.h:16-85 as plain data."""
    _fields_ = [("position", C.c_float * 3), ("orientation", C.c_float * 3), ("up", C.c_float * 3), ("width", C.c_int), ("height", C.c_int),
                ("sensitivity", C.c_float), ("speed", C.c_float), ("fast_speed", C.c_float),
                ("capture_mouse", C.c_int), ("first_capture", C.c_int), ("fix_view", C.c_int), ("fix_position", C.c_int), ("lock_x", C.c_int), ("lock_y", C.c_int)]

    @classmethod
    def make(cls, width, height, position, orientation, up=(0.0, 1.0, 0.0)):
        st = cls()
        st.position[:], st.orientation[:], st.up[:] = position, orientation, up
        st.width, st.height = width, height
        st.sensitivity, st.speed, st.fast_speed = 100.0, 0.5, 2.0          # Camera.h:78-80
        st.first_capture = 1
        return st


class CameraInput(C.Structure):
    _fields_ = [("keys", C.c_uint), ("mouse_x", C.c_double), ("mouse_y", C.c_double), ("imgui_active", C.c_int)]


CAMKEY = {"W": 1, "S": 2, "A": 4, "D": 8, "E": 16, "Q": 32, "SPACE": 64, "LCTRL": 128, "LSHIFT": 256, "C": 512, "ESC": 1024}


def camera_input(state, keys, mouse_x, mouse_y, imgui_active=False):
    """One Camera::HandleInput call (Camera.cpp:116-183) on a CameraState; returns (recenter_cursor, hide_cursor)."""
    inp = CameraInput(int(keys), float(mouse_x), float(mouse_y), 1 if imgui_active else 0)
    rc, hide = C.c_int(0), C.c_int(0)
    _lib.gs4d_host_camera_input(C.byref(state), C.byref(inp), C.byref(rc), C.byref(hide))
    return bool(rc.value), bool(hide.value)


def camera_look_at_point(state, point):
    _lib.gs4d_host_camera_look_at_point(C.byref(state), _ptr(_f32(point)))


def camera_viewport(width, height):
    out = np.zeros(2, np.float32)
    _lib.gs4d_host_camera_viewport(width, height, _ptr(out))
    return out


def camera_focal(fov, width, height):
    out = np.zeros(2, np.float32)
    _lib.gs4d_host_camera_focal(fov, width, height, _ptr(out))
    return out


def write_png(path, rgba8):
    """(H, W, 4) uint8 frame, bottom row first (the framebuffer's orientation) -> PNG file."""
    a = np.ascontiguousarray(rgba8, np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("write_png: expected an (H, W, 4) uint8 array")
    if _lib.gs4d_host_write_png(os.fsencode(path), _ptr(a), a.shape[1], a.shape[0]) != 0:
        raise OSError(f"cannot write {path}")


# ---- device context -----------------------------------------------------------------------------
class Context:
    """One GPU context (gs4d_ctx): buffers, pipeline state, a small swap chain of RGBA32F framebuffers (one per frame lane)."""

    def __init__(self, width, height, device=0):
        h = C.c_void_p()
        rc = _lib.gs4d_create(device, width, height, C.byref(h))
        if rc != 0:
            raise Gs4dError(f"gs4d_create failed ({rc}): {_lib.gs4d_last_error(None).decode()}")
        self._h = h
        self.width, self.height = width, height

    def close(self):
        if getattr(self, "_h", None):
            _lib.gs4d_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise Gs4dError(f"gs4d error {rc}: {_lib.gs4d_last_error(self._h).decode()}")

    # buffers
    def buffer(self, data=None, nbytes=None):
        name = C.c_uint32()
        if data is not None:
            a = np.ascontiguousarray(data)
            self._chk(_lib.gs4d_buffer_create(self._h, _ptr(a), a.nbytes, C.byref(name)))
        else:
            self._chk(_lib.gs4d_buffer_create(self._h, None, nbytes, C.byref(name)))
        return name.value

    def subdata(self, buf, data, offset=0):
        a = np.ascontiguousarray(data)
        self._chk(_lib.gs4d_buffer_subdata(self._h, buf, offset, _ptr(a), a.nbytes))

    def read(self, buf, dtype, count, offset=0):
        out = np.empty(count, dtype)
        self._chk(_lib.gs4d_buffer_read(self._h, buf, offset, _ptr(out), out.nbytes))
        return out

    def delete(self, buf):
        self._chk(_lib.gs4d_buffer_destroy(self._h, buf))

    def device_ptr(self, buf):
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(_lib.gs4d_buffer_device_ptr(self._h, buf, C.byref(p), C.byref(n)))
        return p.value, n.value

    def invalidate(self, buf):
        """Before overwriting a buffer through its device pointer on the caller's stream (gs4d_buffer_invalidate)."""
        self._chk(_lib.gs4d_buffer_invalidate(self._h, buf))

    def bind(self, slot, buf):
        self._chk(_lib.gs4d_bind_storage(self._h, slot, buf))

    # state
    def set_mode(self, mode):
        self._chk(_lib.gs4d_set_mode(self._h, mode))

    def set_uniforms(self, time=None, min_opacity=None, view=None, proj=None):
        if time is not None:
            self._chk(_lib.gs4d_set_uniform_1f(self._h, U_TIME, time))
        if min_opacity is not None:
            self._chk(_lib.gs4d_set_uniform_1f(self._h, U_MIN_OPACITY, min_opacity))
        if view is not None:
            self._chk(_lib.gs4d_set_uniform_mat4(self._h, U_VIEW, _ptr(_f32(view))))
        if proj is not None:
            self._chk(_lib.gs4d_set_uniform_mat4(self._h, U_PROJ, _ptr(_f32(proj))))

    def set_clear_color(self, rgba):
        self._chk(_lib.gs4d_set_clear_color(self._h, _ptr(_f32(rgba))))

    def set_blend(self, src, dst):
        self._chk(_lib.gs4d_set_blend(self._h, src, dst))

    def clear(self):
        self._chk(_lib.gs4d_clear(self._h))

    def resize(self, width, height):
        self._chk(_lib.gs4d_resize(self._h, width, height))
        self.width, self.height = width, height

    # ordering
    def sort_pairs(self, keys, vals, n):
        self._chk(_lib.gs4d_sort_pairs(self._h, keys, vals, n))

    def keygen(self, data, t, cam_pos, keys, idx, n, key_mode=KEY_REF_INV_EUCLID):
        self._chk(_lib.gs4d_keygen(self._h, data, t, _ptr(_f32(cam_pos)), keys, idx, n, key_mode))

    # draw / read-back
    def draw_instanced(self, instances):
        self._chk(_lib.gs4d_draw_instanced(self._h, instances))

    def draw_quads(self, vertices, nquads):
        self._chk(_lib.gs4d_draw_quads(self._h, vertices, nquads))

    def draw_lines(self, verts, rgba, width=1.0, viewproj=None, strip=False):
        """Renderer::DrawLine/DrawGrid/DrawAxis: verts (n, 3) with viewproj, or (n, 2) NDC positions; GL_LINES pairs or a GL_LINE_STRIP."""
        v = _f32(verts)
        dims = v.shape[-1]
        vp = _f32(viewproj) if viewproj is not None else None
        self._chk(_lib.gs4d_draw_lines(self._h, _ptr(v), v.size // dims, dims, 1 if strip else 0, _ptr(vp) if vp is not None else None, _ptr(_f32(rgba)), width))

    def read_pixels(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._chk(_lib.gs4d_read_pixels(self._h, _ptr(out), out.nbytes))
        return out

    def read_pixels_device(self, dptr, nbytes):
        self._chk(_lib.gs4d_read_pixels_device(self._h, C.c_void_p(dptr), nbytes))

    def read_pixels_rgba8_device(self, dptr, nbytes):
        self._chk(_lib.gs4d_read_pixels_rgba8_device(self._h, C.c_void_p(dptr), nbytes))

    def read_frame_rgba8_device(self, frames_back, dptr, nbytes):
        """Pack the current (0) or the previous (1) image of the swap chain to RGBA8 at device pointer `dptr`, asynchronously."""
        self._chk(_lib.gs4d_read_frame_rgba8_device(self._h, frames_back, C.c_void_p(dptr), nbytes))

    def read_frame_rgba8_device_after(self, frames_back, dptr, nbytes, hip_event=None):
        """As read_frame_rgba8_device, but the pack waits only for `hip_event` (a hipEvent_t handle; None: for nothing) instead of for
        everything queued on the caller's stream: for callers that alternate between destination buffers."""
        self._chk(_lib.gs4d_read_frame_rgba8_device_after(self._h, frames_back, C.c_void_p(dptr), nbytes, C.c_void_p(hip_event or 0)))

    def set_tile_shard(self, rank, world):
        """Single-frame sharding: this context bins and composites the tile rows ty % world == rank only."""
        self._chk(_lib.gs4d_set_tile_shard(self._h, rank, world))

    def band_rows(self):
        n = C.c_int(0)
        self._chk(_lib.gs4d_band_rows(self._h, C.byref(n)))
        return n.value

    def read_band_rgba8_device(self, dptr, nbytes):
        self._chk(_lib.gs4d_read_band_rgba8_device(self._h, C.c_void_p(dptr), nbytes))

    def set_stream(self, hip_stream):
        self._chk(_lib.gs4d_set_stream(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    def finish(self):
        self._chk(_lib.gs4d_finish(self._h))

    # measurement
    def set_profiling(self, on, every=1):
        """False: off; True: every stage; an iterable of stage names: only those.  every=k: time only every k-th frame (each timed stage
        costs two event records in a timed frame)."""
        if isinstance(on, (list, tuple, set)):
            mask = 0
            for name in on:
                mask |= 1 << STAGES.index(name)
        else:
            mask = 0x3F if on else 0
        self._chk(_lib.gs4d_set_profiling(self._h, mask | ((int(every) & 0xFF) << 8 if mask and every > 1 else 0)))

    def timings(self):
        ms = np.zeros(len(STAGES), np.float32)
        self._chk(_lib.gs4d_get_timings(self._h, _ptr(ms)))
        return dict(zip(STAGES, (float(x) for x in ms)))

    def timeline(self, max_frames=128):
        """[frames, stages, 2] start/end (ms since the first timed stage of frame 0) of the frames recorded so far; -1 where not run."""
        ms = np.full((max_frames, len(STAGES), 2), -1.0, np.float32)
        n = C.c_int(0)
        self._chk(_lib.gs4d_get_timeline(self._h, _ptr(ms), max_frames, C.byref(n)))
        return ms[:n.value]

    def stats(self):
        st = np.zeros(8, np.uint64)
        self._chk(_lib.gs4d_get_stats(self._h, _ptr(st)))
        return {"entries": int(st[0]) & 0xFFFFFFFF, "staged_draws": int(st[0]) >> 32, "capacity": int(st[1]) & 0xFFFFFFFFFF, "staged_misses": int(st[1]) >> 40, "reruns": int(st[2]) & 0xFFFFFFFF, "aborted_discarded": int(st[2]) >> 32, "tiles": int(st[3]) & 0xFFFFFFFF, "record_read_bytes": (int(st[3]) >> 32) & 0xFF, "composited_tiles": int(st[3]) >> 40,
                "depth_sort_passes": int(st[4]) & 0xFFFFFFFF, "lane_streams_rejected": int(st[4]) >> 32, "tile_sort_passes": int(st[5]) & 0xFFFFFFFF, "renamed_keygens": int(st[5]) >> 32, "lanes": int(st[6]) & 0xFFFF, "lanes_sharing_a_queue": (int(st[6]) >> 16) & 0xFFFF, "fused_keygen_draws": int(st[6]) >> 32,
                "unordered_draws": int(st[7]) & 0xFFFFFFFF, "longest_list": int(st[7]) >> 32}

    def debug_projected(self, n):
        out = np.empty((n, 16), np.float32)
        self._chk(_lib.gs4d_debug_read_projected(self._h, _ptr(out), n))
        return out


def version():
    return _lib.gs4d_version().decode()
