// gs4d_host.cpp — host-side (CPU) half of the path: the splat parameterisation and camera matrices the
// reference computes with GLM before anything reaches the GPU.  Part of libgs4d.so so that callers without
// GLM (the Python binding, the C++ scene driver) can build SSBO contents that are bit-identical to the
// reference's.  Pinned by tests/golden/* (generated from the reference's own code).
//
//   Splat4D::Splat4D (rot,scale,lifetime,fade,dir)   4DSplatRendering/Splat.h:132-159
//   Splat4D::Splat4D (rot0,rot1,scalar)              4DSplatRendering/Splat.h:91-130
//   Splat3D::Splat3D                                 4DSplatRendering/Splat.h:334-344
//   Camera::GetViewMatrix / GetProjMatrix            4DSplatRendering/Camera.cpp:50-58
//   orientation of teapot splats                     4DSplatRendering/Scenes.h:268
//   SplatData record layout                          4DSplatRendering/Scenes.h:22-37
// GLM 0.9.9.9 evaluation order is reproduced (float32, no contraction: build with -ffp-contract=off).
#include "../../include/gs4d.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace {

// column-major fixed-size matrices, element (col c, row r) = a[c*N + r]
template <int N> struct Mat { float a[N * N]; float& at(int c, int r) { return a[c * N + r]; } float at(int c, int r) const { return a[c * N + r]; } };
using Mat3 = Mat<3>;
using Mat4 = Mat<4>;
struct Vec3 { float x, y, z; };
struct Q { float w, x, y, z; };

inline Vec3 sub(Vec3 a, Vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline float dot(Vec3 a, Vec3 b) { const float p0 = a.x * b.x, p1 = a.y * b.y, p2 = a.z * b.z; return p0 + p1 + p2; }
inline Vec3 cross(Vec3 a, Vec3 b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
inline Vec3 scale(Vec3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline Vec3 unit(Vec3 a) { return scale(a, 1.0f / std::sqrt(dot(a, a))); }

// mat3 product, element = ((a0*b0) + (a1*b1)) + (a2*b2)
Mat3 mm(const Mat3& A, const Mat3& B) {
    Mat3 R;
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) R.at(c, r) = A.at(0, r) * B.at(c, 0) + A.at(1, r) * B.at(c, 1) + A.at(2, r) * B.at(c, 2);
    return R;
}
Mat3 tr(const Mat3& A) { Mat3 R; for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) R.at(c, r) = A.at(r, c); return R; }
Mat4 mm(const Mat4& A, const Mat4& B) {
    Mat4 R;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r)
        R.at(c, r) = ((A.at(0, r) * B.at(c, 0) + A.at(1, r) * B.at(c, 1)) + A.at(2, r) * B.at(c, 2)) + A.at(3, r) * B.at(c, 3);
    return R;
}
Mat4 tr(const Mat4& A) { Mat4 R; for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) R.at(c, r) = A.at(r, c); return R; }

Mat3 rot_of(Q q) {   // glm::mat3_cast
    const float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z, xz = q.x * q.z, xy = q.x * q.y, yz = q.y * q.z, wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
    Mat3 R;
    R.at(0, 0) = 1.0f - 2.0f * (yy + zz); R.at(0, 1) = 2.0f * (xy + wz);        R.at(0, 2) = 2.0f * (xz - wy);
    R.at(1, 0) = 2.0f * (xy - wz);        R.at(1, 1) = 1.0f - 2.0f * (xx + zz); R.at(1, 2) = 2.0f * (yz + wx);
    R.at(2, 0) = 2.0f * (xz + wy);        R.at(2, 1) = 2.0f * (yz - wx);        R.at(2, 2) = 1.0f - 2.0f * (xx + yy);
    return R;
}

Q unit(Q q) {        // glm::normalize(quat)
    const float a = q.w * q.w, b = q.x * q.x, c = q.y * q.y, d = q.z * q.z;
    const float len = std::sqrt((a + b) + (c + d));
    if (len <= 0.0f) return { 1.0f, 0.0f, 0.0f, 0.0f };
    const float inv = 1.0f / len;
    return { q.w * inv, q.x * inv, q.y * inv, q.z * inv };
}

Q quat_of(const Mat3& m) {   // glm::quat_cast
    const float tx = m.at(0, 0) - m.at(1, 1) - m.at(2, 2);
    const float ty = m.at(1, 1) - m.at(0, 0) - m.at(2, 2);
    const float tz = m.at(2, 2) - m.at(0, 0) - m.at(1, 1);
    const float tw = m.at(0, 0) + m.at(1, 1) + m.at(2, 2);
    int which = 0; float best = tw;
    if (tx > best) { best = tx; which = 1; }
    if (ty > best) { best = ty; which = 2; }
    if (tz > best) { best = tz; which = 3; }
    const float big = std::sqrt(best + 1.0f) * 0.5f;
    const float k = 0.25f / big;
    if (which == 0) return { big, (m.at(1, 2) - m.at(2, 1)) * k, (m.at(2, 0) - m.at(0, 2)) * k, (m.at(0, 1) - m.at(1, 0)) * k };
    if (which == 1) return { (m.at(1, 2) - m.at(2, 1)) * k, big, (m.at(0, 1) + m.at(1, 0)) * k, (m.at(2, 0) + m.at(0, 2)) * k };
    if (which == 2) return { (m.at(2, 0) - m.at(0, 2)) * k, (m.at(0, 1) + m.at(1, 0)) * k, big, (m.at(1, 2) + m.at(2, 1)) * k };
    return { (m.at(0, 1) - m.at(1, 0)) * k, (m.at(2, 0) + m.at(0, 2)) * k, (m.at(1, 2) + m.at(2, 1)) * k, big };
}

Mat3 sigma3(Q q, const float s[3]) {     // R * S * S * transpose(R), left to right
    Mat3 S; std::memset(&S, 0, sizeof S); S.at(0, 0) = s[0]; S.at(1, 1) = s[1]; S.at(2, 2) = s[2];
    const Mat3 R = rot_of(q);
    return mm(mm(mm(R, S), S), tr(R));
}

} // namespace

extern "C" {

void gs4d_host_look_at(const float eye[3], const float orientation[3], const float up[3], float view[16]) {
    const Vec3 e = { eye[0], eye[1], eye[2] };
    const Vec3 centre = { eye[0] + orientation[0], eye[1] + orientation[1], eye[2] + orientation[2] };
    const Vec3 f = unit(sub(centre, e));
    const Vec3 s = unit(cross(f, { up[0], up[1], up[2] }));
    const Vec3 u = cross(s, f);
    const float m[16] = { s.x, u.x, -f.x, 0.0f, s.y, u.y, -f.y, 0.0f, s.z, u.z, -f.z, 0.0f, -dot(s, e), -dot(u, e), dot(f, e), 1.0f };
    std::memcpy(view, m, sizeof m);
}

void gs4d_host_perspective(float fov_deg, int width, int height, float znear, float zfar, float proj[16]) {
    const float fovy = fov_deg * 0.01745329251994329576923690768489f;
    const float aspect = (float)width / (float)height;
    const float th = std::tan(fovy / 2.0f);
    float m[16] = { 0 };
    m[0] = 1.0f / (aspect * th);
    m[5] = 1.0f / th;
    m[10] = -(zfar + znear) / (zfar - znear);
    m[11] = -1.0f;
    m[14] = -(2.0f * zfar * znear) / (zfar - znear);
    std::memcpy(proj, m, sizeof m);
}

// ---- Camera input model (Camera.cpp:90-99, 116-220) ----------------------------------------------------------------------
static Vec3 v3(const float* p) { return { p[0], p[1], p[2] }; }
static void put(float* p, Vec3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
static Vec3 add(Vec3 a, Vec3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
// glm::rotate(vec3 v, float angle, vec3 normal) = mat3(glm::rotate(mat4(1), angle, normal)) * v   (gtx/rotate_vector.inl:44-52,
// ext/matrix_transform.inl:18-46; with m = identity the 4x4 product leaves Rotate itself)
static Vec3 rotate_about(Vec3 v, float angle, Vec3 normal) {
    const float c = std::cos(angle), s = std::sin(angle);
    const Vec3 axis = unit(normal);
    const Vec3 temp = scale(axis, 1.0f - c);
    float R[3][3];
    R[0][0] = c + temp.x * axis.x;          R[0][1] = temp.x * axis.y + s * axis.z; R[0][2] = temp.x * axis.z - s * axis.y;
    R[1][0] = temp.y * axis.x - s * axis.z; R[1][1] = c + temp.y * axis.y;          R[1][2] = temp.y * axis.z + s * axis.x;
    R[2][0] = temp.z * axis.x + s * axis.y; R[2][1] = temp.z * axis.y - s * axis.x; R[2][2] = c + temp.z * axis.z;
    // Result[k] = m[0]*R[k][0] + m[1]*R[k][1] + m[2]*R[k][2] with m = identity: the products by 0 and 1 are exact, the sums add zeros
    float M[3][3];
    for (int k = 0; k < 3; ++k) for (int r = 0; r < 3; ++r) {
        const float e0 = (r == 0 ? 1.0f : 0.0f) * R[k][0], e1 = (r == 1 ? 1.0f : 0.0f) * R[k][1], e2 = (r == 2 ? 1.0f : 0.0f) * R[k][2];
        M[k][r] = e0 + e1 + e2;
    }
    return { M[0][0] * v.x + M[1][0] * v.y + M[2][0] * v.z, M[0][1] * v.x + M[1][1] * v.y + M[2][1] * v.z, M[0][2] * v.x + M[1][2] * v.y + M[2][2] * v.z };
}

void gs4d_host_camera_rotate(gs4d_camera_state* st, double mouseX, double mouseY) {
    if (st->fix_view) return;
    const double rotX = -(double)st->sensitivity * (mouseY - int((double)st->height / 2.0)) / (double)(st->height);
    const double rotY = -(double)st->sensitivity * (mouseX - int((double)st->width / 2.0)) / (double)(st->width);
    Vec3 o = v3(st->orientation), u = v3(st->up);
    if (!st->lock_x) o = rotate_about(o, (float)(rotX * 0.01745329251994329576923690768489), unit(cross(o, u)));
    const Vec3 side = unit(cross(u, o));
    u = unit(cross(o, side));
    if (!st->lock_y) o = rotate_about(o, (float)(rotY * 0.01745329251994329576923690768489), u);
    put(st->orientation, o); put(st->up, u);
}

void gs4d_host_camera_input(gs4d_camera_state* st, const gs4d_camera_input* in, int* recenter_cursor, int* hide_cursor) {
    if (recenter_cursor) *recenter_cursor = 0;
    if (hide_cursor) *hide_cursor = 0;
    if (in->imgui_active) return;
    const float currentSpeed = (in->keys & GS4D_CAMKEY_LSHIFT) ? st->fast_speed : st->speed;
    Vec3 p = v3(st->position), o = v3(st->orientation), u = v3(st->up);
    if (!st->fix_position) {
        if (in->keys & GS4D_CAMKEY_W) p = add(p, scale(o, currentSpeed));
        if (in->keys & GS4D_CAMKEY_S) p = add(p, scale(o, -currentSpeed));
        if (in->keys & GS4D_CAMKEY_A) { const Vec3 r = unit(cross(o, u)); const float k = -currentSpeed; p = add(p, { k * r.x, k * r.y, k * r.z }); }
        if (in->keys & GS4D_CAMKEY_D) { const Vec3 r = unit(cross(o, u)); p = add(p, { currentSpeed * r.x, currentSpeed * r.y, currentSpeed * r.z }); }
    }
    if (in->keys & GS4D_CAMKEY_E) u = rotate_about(u, 1.0f * 0.01745329251994329576923690768489f, o);
    if (in->keys & GS4D_CAMKEY_Q) u = rotate_about(u, -1.0f * 0.01745329251994329576923690768489f, o);
    if (in->keys & GS4D_CAMKEY_SPACE) p = add(p, { currentSpeed * u.x, currentSpeed * u.y, currentSpeed * u.z });
    if (in->keys & GS4D_CAMKEY_LCTRL) { const float k = -currentSpeed; p = add(p, { k * u.x, k * u.y, k * u.z }); }
    put(st->position, p); put(st->up, u);
    double mx = in->mouse_x, my = in->mouse_y;
    if ((in->keys & GS4D_CAMKEY_C) && !st->capture_mouse) {
        if (hide_cursor) *hide_cursor = 1;
        if (recenter_cursor) *recenter_cursor = 1;
        mx = (double)st->width / 2.0; my = (double)st->height / 2.0;
        st->first_capture = 0; st->capture_mouse = 1;
    }
    if (in->keys & GS4D_CAMKEY_ESC) st->capture_mouse = 0;
    if (st->capture_mouse) {
        if (!st->fix_view && recenter_cursor) *recenter_cursor = 1;
        gs4d_host_camera_rotate(st, mx, my);
    }
}

void gs4d_host_camera_look_at_point(gs4d_camera_state* st, const float point[3]) {
    const Vec3 o = unit(sub(v3(point), v3(st->position)));
    const Vec3 side = unit(cross(v3(st->up), o));
    put(st->orientation, o); put(st->up, unit(cross(o, side)));
}

void gs4d_host_camera_viewport(int width, int height, float out2[2]) {        // glm::normalize(glm::vec2(w, h)) = v * inversesqrt(dot(v, v))
    const float x = (float)width, y = (float)height;
    const float inv = 1.0f / std::sqrt(x * x + y * y);
    out2[0] = x * inv; out2[1] = y * inv;
}

void gs4d_host_camera_focal(float fov, int width, int height, float out2[2]) { // the reference passes DEGREES to tanf (Camera.cpp:97): reproduced
    const float d = (2.0f * tanf(fov * 0.5f));
    out2[0] = width / d; out2[1] = height / d;
}

void gs4d_host_quat_look_at(const float dir[3], const float up[3], float q_wxyz[4]) {
    const Vec3 d = unit(Vec3{ dir[0], dir[1], dir[2] });
    const Vec3 back = { -d.x, -d.y, -d.z };
    const Vec3 right = cross({ up[0], up[1], up[2] }, back);
    const Vec3 c0 = scale(right, 1.0f / std::sqrt(std::fmax(0.00001f, dot(right, right))));
    const Vec3 c1 = cross(back, c0);
    Mat3 B;
    B.at(0, 0) = c0.x; B.at(0, 1) = c0.y; B.at(0, 2) = c0.z;
    B.at(1, 0) = c1.x; B.at(1, 1) = c1.y; B.at(1, 2) = c1.z;
    B.at(2, 0) = back.x; B.at(2, 1) = back.y; B.at(2, 2) = back.z;
    const Q q = unit(quat_of(B));
    q_wxyz[0] = q.w; q_wxyz[1] = q.x; q_wxyz[2] = q.y; q_wxyz[3] = q.z;
}

void gs4d_host_splat3d_cov(const float q_wxyz[4], const float scale3[3], float cov9[9]) {
    const Mat3 g = sigma3({ q_wxyz[0], q_wxyz[1], q_wxyz[2], q_wxyz[3] }, scale3);
    std::memcpy(cov9, g.a, sizeof g.a);
}

// Splat3D::GetSplatMesh / MakeMesh (Splat.h:433-473; vertex layout Geometry.h:37-42): the four 72-byte vertices of one splat's quad,
// {corner(2), position(3), colour(4), Sigma3(9, column-major)}, corners in the order the index buffer 0,2,1 / 2,0,3 expects (Geometry.h:44-50).
void gs4d_host_splat3d_mesh(const float pos3[3], const float q_wxyz[4], const float scale3[3], const float color4[4], float verts72[72]) {
    float cov[9];
    gs4d_host_splat3d_cov(q_wxyz, scale3, cov);
    static const float corner[4][2] = { { 0.5f, 0.5f }, { 0.5f, -0.5f }, { -0.5f, -0.5f }, { -0.5f, 0.5f } };
    for (int v = 0; v < 4; ++v) {
        float* o = verts72 + 18 * v;
        o[0] = corner[v][0]; o[1] = corner[v][1];
        std::memcpy(o + 2, pos3, 12);
        std::memcpy(o + 5, color4, 16);
        std::memcpy(o + 9, cov, 36);
    }
}

// Splat2D ctor + CalcAndSetSigma (Splat.h:551-582): Sigma^-1 of R S S^T R^T with R = (normalize(v0), normalize(v0.y, -v0.x)), S = diag(sqrt l0, sqrt l1)
void gs4d_host_splat2d_sigma_inv(const float v0[2], float l0, float l1, float sigma_inv4[4]) {
    const float s0 = sqrtf(l0), s1 = sqrtf(l1);
    auto norm2 = [](float x, float y, float* o) { const float inv = 1.0f / std::sqrt(x * x + y * y); o[0] = x * inv; o[1] = y * inv; };   // glm::normalize = v * inversesqrt(dot(v, v))
    float r0[2], r1[2];
    norm2(v0[0], v0[1], r0);
    norm2(v0[1], -v0[0], r1);
    // column-major 2x2: m[c][r]; glm mat2 * mat2 (type_mat2x2.inl:453-460)
    struct M2 { float m[2][2]; };
    auto mul = [](const M2& a, const M2& b) { M2 c;
        c.m[0][0] = a.m[0][0] * b.m[0][0] + a.m[1][0] * b.m[0][1]; c.m[0][1] = a.m[0][1] * b.m[0][0] + a.m[1][1] * b.m[0][1];
        c.m[1][0] = a.m[0][0] * b.m[1][0] + a.m[1][0] * b.m[1][1]; c.m[1][1] = a.m[0][1] * b.m[1][0] + a.m[1][1] * b.m[1][1]; return c; };
    auto tr = [](const M2& a) { M2 c; c.m[0][0] = a.m[0][0]; c.m[0][1] = a.m[1][0]; c.m[1][0] = a.m[0][1]; c.m[1][1] = a.m[1][1]; return c; };
    const M2 S = { { { s0, 0.0f }, { 0.0f, s1 } } }, R = { { { r0[0], r0[1] }, { r1[0], r1[1] } } };
    const M2 sig = mul(mul(mul(R, S), tr(S)), tr(R));
    const float ood = 1.0f / (sig.m[0][0] * sig.m[1][1] - sig.m[1][0] * sig.m[0][1]);        // glm::inverse (func_matrix.inl:303-317)
    sigma_inv4[0] = sig.m[1][1] * ood; sigma_inv4[1] = -sig.m[0][1] * ood; sigma_inv4[2] = -sig.m[1][0] * ood; sigma_inv4[3] = sig.m[0][0] * ood;
}

// One 48-byte record of the Gaussians2D scene (Scenes.h:1490-1496, struct :1447-1452): {x, y, 0, 0 | r, g, b, 1 | R S S R^T} with
// R = {cosf a, -sinf a, sinf a, cos a} (`cos` on a float picks the float overload, here as under MSVC: the fixtures confirm it).
void gs4d_host_gaussians2d_record(float angle, float s0, float s1, float px, float py, const float rgb[3], float rec12[12]) {
    struct M2 { float m[2][2]; };
    auto mul = [](const M2& a, const M2& b) { M2 c;
        c.m[0][0] = a.m[0][0] * b.m[0][0] + a.m[1][0] * b.m[0][1]; c.m[0][1] = a.m[0][1] * b.m[0][0] + a.m[1][1] * b.m[0][1];
        c.m[1][0] = a.m[0][0] * b.m[1][0] + a.m[1][0] * b.m[1][1]; c.m[1][1] = a.m[0][1] * b.m[1][0] + a.m[1][1] * b.m[1][1]; return c; };
    auto tr = [](const M2& a) { M2 c; c.m[0][0] = a.m[0][0]; c.m[0][1] = a.m[1][0]; c.m[1][0] = a.m[0][1]; c.m[1][1] = a.m[1][1]; return c; };
    const M2 R = { { { cosf(angle), -sinf(angle) }, { sinf(angle), std::cos(angle) } } }, S = { { { s0, 0.0f }, { 0.0f, s1 } } };
    const M2 g = mul(mul(mul(R, S), S), tr(R));
    rec12[0] = px; rec12[1] = py; rec12[2] = 0.0f; rec12[3] = 0.0f;
    rec12[4] = rgb[0]; rec12[5] = rgb[1]; rec12[6] = rgb[2]; rec12[7] = 1.0f;
    rec12[8] = g.m[0][0]; rec12[9] = g.m[0][1]; rec12[10] = g.m[1][0]; rec12[11] = g.m[1][1];
}

void gs4d_host_splat4d_cov(const float q_wxyz[4], const float scale3[3], float lifetime, float fade, const float dir[3], float cov16[16]) {
    // Splat.h:139: `log(fadeof)` on a float is the float overload; the -2.0 factor promotes the quotient to double
    const double denom = (fade == 0.5f) ? (double)1.3862943611198906f : -2.0 * (double)std::log(fade);
    const float sd = (float)((double)(lifetime * lifetime) / denom);
    const float td[3] = { dir[0] * sd, dir[1] * sd, dir[2] * sd };
    const Mat3 sig = sigma3({ q_wxyz[0], q_wxyz[1], q_wxyz[2], q_wxyz[3] }, scale3);
    const float inv = 1.0f / sd;
    Mat4 C;
    for (int c = 0; c < 3; ++c) {
        for (int r = 0; r < 3; ++r) C.at(c, r) = sig.at(c, r) + (td[r] * td[c]) * inv;     // sig + (1/s) * outerProduct(td, td)
        C.at(c, 3) = td[c];
        C.at(3, c) = td[c];
    }
    C.at(3, 3) = sd;
    std::memcpy(cov16, C.a, sizeof C.a);
}

void gs4d_host_splat4d_cov2q(const float q0_wxyz[4], const float q1_wxyz[4], const float scale4[4], float cov16[16]) {
    const Q l = unit(Q{ q0_wxyz[0], q0_wxyz[1], q0_wxyz[2], q0_wxyz[3] });
    const Q r = unit(Q{ q1_wxyz[0], q1_wxyz[1], q1_wxyz[2], q1_wxyz[3] });
    const Mat4 L = { { l.w, -l.x, -l.y, -l.z,  l.x, l.w, -l.z, l.y,  l.y, l.z, l.w, -l.x,  l.z, -l.y, l.x, l.w } };
    const Mat4 R = { { r.w, -r.x, -r.y, -r.z,  r.x, r.w, r.z, -r.y,  r.y, -r.z, r.w, r.x,  r.z, r.y, -r.x, r.w } };
    Mat4 S; std::memset(&S, 0, sizeof S);
    for (int i = 0; i < 4; ++i) S.at(i, i) = scale4[i];
    const Mat4 rot = mm(L, R);
    const Mat4 g = mm(mm(mm(rot, S), tr(S)), tr(rot));
    std::memcpy(cov16, g.a, sizeof g.a);
}

void gs4d_host_build_records_3d(size_t n, const float* pos3, const float* q_wxyz, const float* scale3, const float* rgba, float* rec) {
    for (size_t i = 0; i < n; ++i) {
        float* o = rec + 24 * i;
        float g[9];
        gs4d_host_splat3d_cov(q_wxyz + 4 * i, scale3 + 3 * i, g);
        o[0] = pos3[3 * i]; o[1] = pos3[3 * i + 1]; o[2] = pos3[3 * i + 2]; o[3] = 0.0f;         // mu_t = 0
        std::memcpy(o + 4, rgba + 4 * i, 16);
        for (int c = 0; c < 3; ++c) { o[8 + 4 * c] = g[3 * c]; o[9 + 4 * c] = g[3 * c + 1]; o[10 + 4 * c] = g[3 * c + 2]; o[11 + 4 * c] = 0.0f; }
        o[20] = 0.0f; o[21] = 0.0f; o[22] = 0.0f; o[23] = 1.0f;                                 // Sigma44 = 1
    }
}

void gs4d_host_build_records_4d(size_t n, const float* pos4, const float* q_wxyz, const float* scale3, const float* lifetime, const float* fade,
                                const float* dir3, const float* rgba, float* rec) {
    for (size_t i = 0; i < n; ++i) {
        float* o = rec + 24 * i;
        std::memcpy(o, pos4 + 4 * i, 16);
        std::memcpy(o + 4, rgba + 4 * i, 16);
        gs4d_host_splat4d_cov(q_wxyz + 4 * i, scale3 + 3 * i, lifetime[i], fade[i], dir3 + 3 * i, o + 8);
    }
}


// ---- scene generators: the loops of LinearMotion::init (Scenes.h:258-279) and NonLinearMotion::init
//      (Scenes.h:517-545) with GetModelExtrema (:75-91) and GetColor (:58-68; Utils.cpp lerp/mapf, note mapf ignores `a` in the
//      numerator, Utils.cpp:130-133).  Pinned by the CRCs of the reference-generated SSBOs in tests/golden/manifest.json. ----
} // extern "C"

namespace {

inline float fminu(float a, float b) { return a < b ? a : b; }          // Utils::minf
inline float fmaxu(float a, float b) { return a > b ? a : b; }          // Utils::maxf
inline float lerpu(float a, float b, float t) { return a + ((b - a) * t); }
inline float mapfu(float x, float a, float b, float c, float d) { return c + ((x / (b - a)) * (d - c)); }
inline float clamp01(float x) { const float lo = (x < 0.0f) ? 0.0f : x; return (1.0f < lo) ? 1.0f : lo; }     // glm::clamp = min(max(x,0),1)

struct Extrema { float mn[3], mx[3]; };
Extrema extrema(const float* v6, size_t n) {
    Extrema e; for (int k = 0; k < 3; ++k) { e.mx[k] = -INFINITY; e.mn[k] = INFINITY; }
    for (size_t i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { const float p = v6[6 * i + k]; e.mx[k] = fmaxu(p, e.mx[k]); e.mn[k] = fminu(p, e.mn[k]); }
    return e;
}
void get_color(const float pos[3], const Extrema& e, const float nrm[3], float out[4], float mina = 0.65f, float maxa = 1.0f, float lower = 0.0f) {
    const float t0 = 0.0f * nrm[0], t1 = -1.0f * nrm[1], t2 = 0.0f * nrm[2];
    const float d = (t0 + t1) + t2;                                       // glm::dot({0,-1,0}, normal)
    const float max_bright = mapfu(-d, -1.0f, 1.0f, mina, maxa);
    for (int k = 0; k < 3; ++k) out[k] = clamp01(lerpu(lower, max_bright, ((pos[k] - e.mn[k]) / (e.mx[k] - e.mn[k]))));
    out[3] = clamp01(1.0f);
}
// vec3(glm::rotate(vec4(v, 0), angle, vec3(0,1,0)))  (gtx/rotate_vector.inl:66-74 -> gtx/transform -> ext/matrix_transform.inl:18-46)
void rotate_about_y(const float v3[3], float angle, float out[3]) {
    const float c = std::cos(angle), s = std::sin(angle);
    const float inv = 1.0f / std::sqrt((0.0f * 0.0f + 1.0f * 1.0f) + 0.0f * 0.0f);
    const float ax[3] = { 0.0f * inv, 1.0f * inv, 0.0f * inv };
    const float tmp[3] = { (1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2] };
    float Rm[3][3];
    Rm[0][0] = c + tmp[0] * ax[0];          Rm[0][1] = tmp[0] * ax[1] + s * ax[2]; Rm[0][2] = tmp[0] * ax[2] - s * ax[1];
    Rm[1][0] = tmp[1] * ax[0] - s * ax[2];  Rm[1][1] = c + tmp[1] * ax[1];         Rm[1][2] = tmp[1] * ax[2] + s * ax[0];
    Rm[2][0] = tmp[2] * ax[0] + s * ax[1];  Rm[2][1] = tmp[2] * ax[1] - s * ax[0]; Rm[2][2] = c + tmp[2] * ax[2];
    // Result[c] = I[0]*R[c][0] + I[1]*R[c][1] + I[2]*R[c][2]; Result[3] = I[3]
    float M[4][4];
    for (int cc = 0; cc < 3; ++cc) for (int r = 0; r < 4; ++r) {
        const float i0 = r == 0 ? 1.0f : 0.0f, i1 = r == 1 ? 1.0f : 0.0f, i2 = r == 2 ? 1.0f : 0.0f;
        M[cc][r] = (i0 * Rm[cc][0] + i1 * Rm[cc][1]) + i2 * Rm[cc][2];
    }
    for (int r = 0; r < 4; ++r) M[3][r] = r == 3 ? 1.0f : 0.0f;
    const float v[4] = { v3[0], v3[1], v3[2], 0.0f };
    for (int r = 0; r < 3; ++r) out[r] = (M[0][r] * v[0] + M[1][r] * v[1]) + (M[2][r] * v[2] + M[3][r] * v[3]);     // mat4*vec4: (Mul0+Mul1)+(Mul2+Mul3)
}
void rotate_forward_about_y(float angle, float out[3]) { const float fwd[3] = { 1.0f, 0.0f, 0.0f }; rotate_about_y(fwd, angle, out); }

constexpr float RAD = 0.01745329251994329576923690768489f;      // glm::radians

// One record from (position, time, facing normal, velocity, colour inputs): the tail every scene loop shares
// (Splat4D ctor Splat.h:132-159; quatLookAt + normalize; GetColor Scenes.h:58-68)
void make_record(const float pos[3], float t, const float face_normal[3], const float splat_scale[3], float lifetime, float fade, const float vel[3],
                 const float col_pos[3], const Extrema& e, const float col_normal[3], float* rec) {
    const float up[3] = { 0.0f, 1.0f, 0.0f };
    rec[0] = pos[0]; rec[1] = pos[1]; rec[2] = pos[2]; rec[3] = t;
    get_color(col_pos, e, col_normal, rec + 4);
    float q[4]; gs4d_host_quat_look_at(face_normal, up, q);
    gs4d_host_splat4d_cov(q, splat_scale, lifetime, fade, vel, rec + 8);
}

} // namespace

extern "C" {

void gs4d_host_scene_linear(size_t nverts, const float* verts6, int steps, float time_multiplier, float object_scale, const float splat_scale[3],
                            float lifetime, float fade, float speed, float* records24) {
    const Extrema e = extrema(verts6, nverts);
    const float up[3] = { 0.0f, 1.0f, 0.0f };
    size_t o = 0;
    for (int dt = 0; dt < steps; ++dt) for (size_t i = 0; i < nverts; ++i, ++o) {
        const float* pos = verts6 + 6 * i; const float* nrm = pos + 3;
        const float off = float(dt * time_multiplier);                      // dir * float(dt * m_lin_time_multiplyer), dir = (1,0,0)
        float* rec = records24 + 24 * o;
        rec[0] = (object_scale * pos[0]) + 1.0f * off; rec[1] = (object_scale * pos[1]) + 0.0f * off; rec[2] = (object_scale * pos[2]) + 0.0f * off; rec[3] = float(dt);
        get_color(pos, e, nrm, rec + 4);
        float q[4]; gs4d_host_quat_look_at(nrm, up, q);
        const float ninv = 1.0f / std::sqrt((1.0f * 1.0f + 0.0f * 0.0f) + 0.0f * 0.0f);   // glm::normalize(dir)
        const float dir[3] = { (1.0f * ninv) * speed, (0.0f * ninv) * speed, (0.0f * ninv) * speed };
        gs4d_host_splat4d_cov(q, splat_scale, lifetime, fade, dir, rec + 8);
    }
}

void gs4d_host_scene_nonlinear(size_t nverts, const float* verts6, int steps, float angle_multiplier, float radius, float object_scale,
                               const float splat_scale[3], float lifetime, float fade, float speed, size_t max_records, float* records24) {
    const Extrema e = extrema(verts6, nverts);
    const float up[3] = { 0.0f, 1.0f, 0.0f };
    size_t o = 0;
    for (int dt = 0; dt < steps && o < max_records; ++dt) {
        float cur[3], nxt[3];
        rotate_forward_about_y(float(dt * angle_multiplier) * RAD, cur);
        rotate_forward_about_y(float((dt + 1) * angle_multiplier) * RAD, nxt);
        for (size_t i = 0; i < nverts && o < max_records; ++i, ++o) {
            const float* pos = verts6 + 6 * i; const float* nrm = pos + 3;
            float* rec = records24 + 24 * o;
            for (int k = 0; k < 3; ++k) rec[k] = (object_scale * pos[k]) + (cur[k] * radius);
            rec[3] = float(dt);
            get_color(pos, e, nrm, rec + 4);
            float q[4]; gs4d_host_quat_look_at(nrm, up, q);
            const float dir[3] = { (nxt[0] - cur[0]) * speed, (nxt[1] - cur[1]) * speed, (nxt[2] - cur[2]) * speed };
            gs4d_host_splat4d_cov(q, splat_scale, lifetime, fade, dir, rec + 8);
        }
    }
}

// RotationMotion::init (Scenes.h:775-803): every vertex and its normal turn about the Y axis, 'angle_multiplier' degrees per time step.
void gs4d_host_scene_rotation(size_t nverts, const float* verts6, int steps, float angle_multiplier, float object_scale, const float splat_scale[3],
                              float lifetime, float fade, float speed, size_t max_records, float* records24) {
    const Extrema e = extrema(verts6, nverts);
    size_t o = 0;
    for (int dt = 0; dt < steps && o < max_records; ++dt) {
        const float a0 = float(dt * angle_multiplier) * RAD, a1 = float((dt + 1) * angle_multiplier) * RAD;
        for (size_t i = 0; i < nverts && o < max_records; ++i, ++o) {
            const float* pos = verts6 + 6 * i; const float* nrm = pos + 3;
            float cur[3], nxt[3], nr[3];
            rotate_about_y(pos, a0, cur); rotate_about_y(pos, a1, nxt); rotate_about_y(nrm, a0, nr);
            const float p[3] = { object_scale * cur[0], object_scale * cur[1], object_scale * cur[2] };
            const float vel[3] = { (nxt[0] - cur[0]) * speed, (nxt[1] - cur[1]) * speed, (nxt[2] - cur[2]) * speed };
            make_record(p, float(dt), nr, splat_scale, lifetime, fade, vel, pos, e, nrm, records24 + 24 * o);
        }
    }
}

// CombinedMotion::init (Scenes.h:1035-1068): rotation about Y plus a sine-wave translation along X.
void gs4d_host_scene_combined(size_t nverts, const float* verts6, int steps, float angle_multiplier, float lin_multiplier, float amplitude, float frequency,
                              float object_scale, const float splat_scale[3], float lifetime, float fade, float speed, size_t max_records, float* records24) {
    const Extrema e = extrema(verts6, nverts);
    size_t o = 0;
    for (int dt = 0; dt < steps && o < max_records; ++dt) {
        const float a0 = float(dt * angle_multiplier) * RAD, a1 = float((dt + 1) * angle_multiplier) * RAD;
        const float w0[3] = { lin_multiplier * (frequency * float(dt)), lin_multiplier * (amplitude * sinf(frequency * float(dt))), lin_multiplier * 0.0f };
        const float w1[3] = { lin_multiplier * (frequency * float(dt + 1)), lin_multiplier * (amplitude * sinf(frequency * float(dt + 1))), lin_multiplier * 0.0f };
        for (size_t i = 0; i < nverts && o < max_records; ++i, ++o) {
            const float* pos = verts6 + 6 * i; const float* nrm = pos + 3;
            const float md[3] = { object_scale * pos[0], object_scale * pos[1], object_scale * pos[2] };
            float r0[3], r1[3], nr[3];
            rotate_about_y(md, a0, r0); rotate_about_y(md, a1, r1); rotate_about_y(nrm, a0, nr);
            const float p[3] = { r0[0] + w0[0], r0[1] + w0[1], r0[2] + w0[2] };
            const float pn[3] = { r1[0] + w1[0], r1[1] + w1[1], r1[2] + w1[2] };
            const float vel[3] = { (pn[0] - p[0]) * speed, (pn[1] - p[1]) * speed, (pn[2] - p[2]) * speed };
            make_record(p, float(dt), nr, splat_scale, lifetime, fade, vel, pos, e, nrm, records24 + 24 * o);
        }
    }
}

// BrokenMotion::init (Scenes.h:1965-1989): the object jumps back every 20 steps (y = fmod(1 + dt, 20)).
void gs4d_host_scene_broken(size_t nverts, const float* verts6, int steps, float object_scale, const float splat_scale[3],
                            float lifetime, float fade, float speed, size_t max_records, float* records24) {
    const Extrema e = extrema(verts6, nverts);
    size_t o = 0;
    for (int dt = 0; dt < steps && o < max_records; ++dt) {
        const float pd[3] = { 1.0f + dt, std::fmod((1.0f + dt), 20.0f), 0.0f };
        const float pn[3] = { 1.0f + (dt + 1.0f), std::fmod((1.0f + (dt + 1.0f)), 20.0f), 0.0f };
        const float vel[3] = { (pn[0] - pd[0]) * speed, (pn[1] - pd[1]) * speed, (pn[2] - pd[2]) * speed };
        for (size_t i = 0; i < nverts && o < max_records; ++i, ++o) {
            const float* pos = verts6 + 6 * i; const float* nrm = pos + 3;
            const float p[3] = { (object_scale * pos[0]) + pd[0], (object_scale * pos[1]) + pd[1], (object_scale * pos[2]) + pd[2] };
            make_record(p, float(dt), nrm, splat_scale, lifetime, fade, vel, pos, e, nrm, records24 + 24 * o);
        }
    }
}

// SquareMotion::init (Scenes.h:2216-2259): the object walks the sides of a square in the XZ plane, steps/4 steps per side.
void gs4d_host_scene_square(size_t nverts, const float* verts6, int steps, float square_size, float object_scale, const float splat_scale[3],
                            float lifetime, float fade, float speed, size_t max_records, float* records24) {
    const Extrema e = extrema(verts6, nverts);
    size_t o = 0;
    int side = 0;
    const int per_side = steps / 4;
    if (per_side <= 0) return;                               // the reference divides by zero here (Scenes.h:2219-2220)
    const float delta = square_size / float(per_side);
    float pd[3] = { square_size / 2.0f, 0.0f, square_size / 2.0f };
    float pn[3] = { square_size / 2.0f + (delta * -1.0f), 0.0f + (delta * 0.0f), square_size / 2.0f + (delta * 0.0f) };
    static const float DIRS[4][3] = { { -1.0f, 0.0f, 0.0f }, { 0.0f, 0.0f, -1.0f }, { 1.0f, 0.0f, 0.0f }, { 0.0f, 0.0f, 1.0f } };
    for (int dt = 0; dt < steps && o < max_records; ++dt) {
        float dir[3] = { 0.0f, 0.0f, 0.0f };
        if (dt > 0 && dt % per_side == 0) side += 1;
        if (side >= 0 && side < 4) for (int k = 0; k < 3; ++k) dir[k] = DIRS[side][k];
        for (int k = 0; k < 3; ++k) pd[k] = pd[k] + (delta * dir[k]);
        int s2 = side;
        if ((dt + 1) > 0 && (dt + 1) % per_side == 0) s2 += 1;
        if (s2 >= 0 && s2 < 4) for (int k = 0; k < 3; ++k) dir[k] = DIRS[s2][k];       // s2 == 4 keeps the direction of `side`
        for (int k = 0; k < 3; ++k) pn[k] = pn[k] + (delta * dir[k]);
        const float vel[3] = { (pn[0] - pd[0]) * speed, (pn[1] - pd[1]) * speed, (pn[2] - pd[2]) * speed };
        for (size_t i = 0; i < nverts && o < max_records; ++i, ++o) {
            const float* pos = verts6 + 6 * i; const float* nrm = pos + 3;
            const float p[3] = { (object_scale * pos[0]) + pd[0], (object_scale * pos[1]) + pd[1], (object_scale * pos[2]) + pd[2] };
            make_record(p, float(dt), nrm, splat_scale, lifetime, fade, vel, pos, e, nrm, records24 + 24 * o);
        }
    }
}

// VData::parse (VDataParser.h:25-58): whitespace-separated std::stof tokens, 6 per vertex (position, normal).
// Returns the number of vertices in the file (which may exceed cap_vertices; only cap_vertices are written), or -1 if it cannot be opened.
long gs4d_host_parse_vdata(const char* path, float* verts6, size_t cap_vertices) {
    std::ifstream file(path);
    if (!file.is_open()) return -1;
    std::vector<float> vals; std::string word;
    while (file >> word) vals.push_back(std::stof(word));
    const size_t n = vals.size() / 6;
    for (size_t i = 0; i < n && i < cap_vertices; ++i) std::memcpy(verts6 + 6 * i, vals.data() + 6 * i, 24);
    return (long)n;
}

// VData::parse_splat_data (VDataParser.h:60-123) followed by the ObjectDisplay record loop (Scenes.h:2483-2491): whitespace-separated
// std::stof tokens, 23 per splat — position(3), colour(4), 4x4 covariance column by column(16) — become 96-byte records
// {object_scale * position, 0 | colour | covariance}.  Returns the number of splats in the file (which may exceed cap_records; only
// cap_records are written), or -1 if it cannot be opened.  A trailing partial splat is ignored (the reference reads past the end there).
long gs4d_host_parse_sd(const char* path, float object_scale, float* records24, size_t cap_records) {
    std::ifstream file(path);
    if (!file.is_open()) return -1;
    std::vector<float> vals; std::string word;
    while (file >> word) vals.push_back(std::stof(word));
    const size_t n = vals.size() / 23;
    for (size_t i = 0; i < n && i < cap_records; ++i) {
        const float* w = vals.data() + 23 * i;
        float* rec = records24 + 24 * i;
        rec[0] = object_scale * w[0]; rec[1] = object_scale * w[1]; rec[2] = object_scale * w[2]; rec[3] = 0.0f;
        std::memcpy(rec + 4, w + 3, 16);
        std::memcpy(rec + 8, w + 7, 64);
    }
    return (long)n;
}

// Presentation: the packed RGBA8 frame (gs4d_read_pixels_rgba8_device, bottom row first like the GL window
// framebuffer the reference presents, Application.cpp:89, 186) as a PNG file, top row first.  Self-contained: stored (uncompressed)
// deflate blocks, CRC-32 and Adler-32 computed here.  Returns 0, or -1 if the file cannot be written.
int gs4d_host_write_png(const char* path, const uint8_t* rgba8, int width, int height) {
    if (!path || !rgba8 || width <= 0 || height <= 0) return -1;
    static uint32_t table[256]; static bool have = false;
    if (!have) { for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; } have = true; }
    auto crc = [&](uint32_t c, const uint8_t* p, size_t n) { for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 255u] ^ (c >> 8); return c; };
    auto be32 = [](uint8_t* o, uint32_t v) { o[0] = (uint8_t)(v >> 24); o[1] = (uint8_t)(v >> 16); o[2] = (uint8_t)(v >> 8); o[3] = (uint8_t)v; };
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    bool ok = true;
    auto chunk = [&](const char type[4], const std::vector<uint8_t>& data) {
        uint8_t hdr[8]; be32(hdr, (uint32_t)data.size()); std::memcpy(hdr + 4, type, 4);
        uint32_t c = crc(0xFFFFFFFFu, hdr + 4, 4); c = crc(c, data.data(), data.size()) ^ 0xFFFFFFFFu;
        uint8_t tail[4]; be32(tail, c);
        ok = ok && fwrite(hdr, 1, 8, f) == 8 && (data.empty() || fwrite(data.data(), 1, data.size(), f) == data.size()) && fwrite(tail, 1, 4, f) == 4;
    };
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    ok = fwrite(sig, 1, 8, f) == 8;
    std::vector<uint8_t> ihdr(13); be32(ihdr.data(), (uint32_t)width); be32(ihdr.data() + 4, (uint32_t)height);
    ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;            // 8 bits, RGBA, deflate, no filter method, no interlace
    chunk("IHDR", ihdr);
    // raw scanlines: filter byte 0 + row, top row first (the framebuffer's row 0 is the bottom row)
    const size_t stride = (size_t)width * 4, raw_n = (stride + 1) * (size_t)height;
    std::vector<uint8_t> raw(raw_n);
    for (int y = 0; y < height; ++y) { uint8_t* o = raw.data() + (stride + 1) * (size_t)y; o[0] = 0; std::memcpy(o + 1, rgba8 + stride * (size_t)(height - 1 - y), stride); }
    std::vector<uint8_t> z; z.reserve(raw_n + raw_n / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    for (size_t off = 0; off < raw_n;) {
        const size_t n = std::min<size_t>(65535, raw_n - off);
        z.push_back(off + n == raw_n ? 1 : 0);
        z.push_back((uint8_t)(n & 255)); z.push_back((uint8_t)(n >> 8)); z.push_back((uint8_t)(~n & 255)); z.push_back((uint8_t)((~n >> 8) & 255));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        off += n;
    }
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < raw_n; ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
    uint8_t ad[4]; be32(ad, (b << 16) | a); z.insert(z.end(), ad, ad + 4);
    chunk("IDAT", z);
    chunk("IEND", {});
    ok = (fclose(f) == 0) && ok;
    return ok ? 0 : -1;
}

} // extern "C"

