// gs4d_oracle.cpp — CPU restatement of the reference's forward splat path.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load it.  The product (libgs4d.so) never links,
// loads or calls anything in oracle/.
//
// What it restates (file:line are relative to the reference tree):
//   * host parameterisation          4DSplatRendering/Splat.h:91-159, 334-344 (GLM 0.9.9.9 op order)
//   * camera                         4DSplatRendering/Camera.cpp:50-63 (glm::lookAt / glm::perspective)
//   * sort key                       4DSplatRendering/Scenes.h:28-36, 314-319
//   * radix sort (3 GLSL kernels)    resources/radix_sort_{count,local_offsets,reorder}.comp.glsl,
//                                    driver Dependencies/GPU_RADIX_SORT/radix_sort.hpp:258-392
//   * 4D/3D/2D vertex shaders        Shader/Splats4D/Splat4DVertexShaderInstanced.GLSL:48-150,
//                                    Shader/Splats3D/Splat3DVertexShaderFull.GLSL:43-98,
//                                    Shader/Splats2D/Splat2DVSI.GLSL:59-94
//   * fragment shaders               Shader/Splats4D/Splat4DFragShader.GLSL:16-31,
//                                    Shader/Splats3D/Splat3DFragShaderFull.GLSL:16-24,
//                                    Shader/Splats2D/Splat2DFragShader.GLSL:10-25
//   * blend / clear state            4DSplatRendering/Application.cpp:125, 150-154
//   * quad                           4DSplatRendering/Geometry.h:44-50
//
// Pinning status (see DESIGN.md "Oracle"):
//   * host parameterisation, camera, sort key, .vdata parse, scene generators:
//       PINNED bit-for-bit by tests/golden/*.bin, generated here from the reference's own
//       C++ compiled where it lies (oracle/ref/refgen.cpp -> oracle/_ref/refgen).
//   * radix sort, GLSL vertex/fragment math, the fixed-function rasteriser/blender, the overlay lines:
//       PINNED against the reference's own GLSL programs EXECUTED in the build container: oracle/ref/refgl_main.cpp brings up a
//       headless OpenGL 4.5 core context on the image's Mesa 23.2.1 llvmpipe (swrast_dri.so + libglapi, no X, no EGL) and runs the
//       reference's shader files unmodified — transform-feedback captures of the three vertex shaders, RGBA32F images of the three
//       splat programs and the line program under the reference's blend/clear state, the three compute programs under
//       radix_sort.hpp:258-392's dispatch sequence.  oracle/make_golden_gl.py wrote tests/golden/gl_*.npz (manifest_gl.json);
//       tests/test_oracle_gl.py holds this file to them: cull identical record for record, quad centre <= 1.5e-4 px, conic 1.2e-7
//       relative, p(t) bit-equal up to the exp2 lowering of llvmpipe's exp, permutations identical, images within 1e-4 except at
//       pixels whose centre lies within 1/128 px of a quad edge (the GL snaps vertices to 1/256 px; gs4do_covered() tests in float):
//       7 of 23 056 touched pixels at 1080p (tests/gl_cases.py states every bar).
//
// float32 throughout, operations in the written order; build with -ffp-contract=off so the only
// fused operations are the explicit fmaf() calls.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <thread>
#include <algorithm>

#define GS4DO_API extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------------------------------------
// small GLM-faithful helpers (column-major, m[c][r])
// ---------------------------------------------------------------------------------------------
namespace {

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
struct M2 { float m[2][2]; };
struct M3 { float m[3][3]; };
struct M4 { float m[4][4]; };
struct Quat { float w, x, y, z; };  // GLM 0.9.9.9 memory / ctor order w,x,y,z (glm/detail/type_quat.hpp:42-60)

inline float dot3(V3 a, V3 b) { float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return tx + ty + tz; }  // glm/detail/func_geometric.inl compute_dot<vec3>
inline V3 cross3(V3 x, V3 y) { return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; } // func_geometric.inl compute_cross
inline float inversesqrtf_(float x) { return 1.0f / sqrtf(x); }                                            // func_exponential.inl:136-139
inline V3 normalize3(V3 v) { float s = inversesqrtf_(dot3(v, v)); return { v.x * s, v.y * s, v.z * s }; }      // func_geometric.inl:82-90
inline V2 normalize2(V2 v) { float tx = v.x * v.x, ty = v.y * v.y; float s = inversesqrtf_(tx + ty); return { v.x * s, v.y * s }; }

// glm/detail/type_mat3x3.inl:486-520
inline M3 mul33(const M3& a, const M3& b) {
    M3 r;
    for (int c = 0; c < 3; ++c)
        for (int rr = 0; rr < 3; ++rr)
            r.m[c][rr] = a.m[0][rr] * b.m[c][0] + a.m[1][rr] * b.m[c][1] + a.m[2][rr] * b.m[c][2];
    return r;
}
inline M3 transpose3(const M3& a) { M3 r; for (int c = 0; c < 3; ++c) for (int rr = 0; rr < 3; ++rr) r.m[c][rr] = a.m[rr][c]; return r; }
inline M3 diag3(V3 s) { M3 r; memset(&r, 0, sizeof r); r.m[0][0] = s.x; r.m[1][1] = s.y; r.m[2][2] = s.z; return r; }
// glm/matrix.inl outerProduct(c, r): result[i] = c * r[i]
inline M3 outer3(V3 c, V3 r) { M3 o; float rv[3] = { r.x, r.y, r.z }; for (int i = 0; i < 3; ++i) { o.m[i][0] = c.x * rv[i]; o.m[i][1] = c.y * rv[i]; o.m[i][2] = c.z * rv[i]; } return o; }

// glm/gtc/quaternion.inl:47-74
inline M3 mat3_cast(Quat q) {
    M3 R;
    float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z, qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z, qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
    R.m[0][0] = 1.0f - 2.0f * (qyy + qzz); R.m[0][1] = 2.0f * (qxy + qwz);        R.m[0][2] = 2.0f * (qxz - qwy);
    R.m[1][0] = 2.0f * (qxy - qwz);        R.m[1][1] = 1.0f - 2.0f * (qxx + qzz); R.m[1][2] = 2.0f * (qyz + qwx);
    R.m[2][0] = 2.0f * (qxz + qwy);        R.m[2][1] = 2.0f * (qyz - qwx);        R.m[2][2] = 1.0f - 2.0f * (qxx + qyy);
    return R;
}
// glm/gtc/quaternion.inl:81-126
inline Quat quat_cast(const M3& m) {
    float fx = m.m[0][0] - m.m[1][1] - m.m[2][2];
    float fy = m.m[1][1] - m.m[0][0] - m.m[2][2];
    float fz = m.m[2][2] - m.m[0][0] - m.m[1][1];
    float fw = m.m[0][0] + m.m[1][1] + m.m[2][2];
    int big = 0; float fb = fw;
    if (fx > fb) { fb = fx; big = 1; }
    if (fy > fb) { fb = fy; big = 2; }
    if (fz > fb) { fb = fz; big = 3; }
    float bv = sqrtf(fb + 1.0f) * 0.5f;
    float mult = 0.25f / bv;
    switch (big) {
    case 0: return { bv, (m.m[1][2] - m.m[2][1]) * mult, (m.m[2][0] - m.m[0][2]) * mult, (m.m[0][1] - m.m[1][0]) * mult };
    case 1: return { (m.m[1][2] - m.m[2][1]) * mult, bv, (m.m[0][1] + m.m[1][0]) * mult, (m.m[2][0] + m.m[0][2]) * mult };
    case 2: return { (m.m[2][0] - m.m[0][2]) * mult, (m.m[0][1] + m.m[1][0]) * mult, bv, (m.m[1][2] + m.m[2][1]) * mult };
    default: return { (m.m[0][1] - m.m[1][0]) * mult, (m.m[2][0] + m.m[0][2]) * mult, (m.m[1][2] + m.m[2][1]) * mult, bv };
    }
}
// glm/ext/quaternion_geometric.inl: dot(q,q) = (w*w + x*x) + (y*y + z*z) via compute_dot<qua> on the w,x,y,z layout
inline Quat normalize_q(Quat q) {
    float tw = q.w * q.w, tx = q.x * q.x, ty = q.y * q.y, tz = q.z * q.z;
    float len = sqrtf((tw + tx) + (ty + tz));
    if (len <= 0.0f) return { 1, 0, 0, 0 };
    float o = 1.0f / len;
    return { q.w * o, q.x * o, q.y * o, q.z * o };
}
// glm/gtc/quaternion.inl:179-190 (RH)
inline Quat quatLookAtRH(V3 direction, V3 up) {
    M3 R;
    V3 c2 = { -direction.x, -direction.y, -direction.z };
    V3 right = cross3(up, c2);
    float s = inversesqrtf_(fmaxf(0.00001f, dot3(right, right)));
    V3 c0 = { right.x * s, right.y * s, right.z * s };
    V3 c1 = cross3(c2, c0);
    R.m[0][0] = c0.x; R.m[0][1] = c0.y; R.m[0][2] = c0.z;
    R.m[1][0] = c1.x; R.m[1][1] = c1.y; R.m[1][2] = c1.z;
    R.m[2][0] = c2.x; R.m[2][1] = c2.y; R.m[2][2] = c2.z;
    return quat_cast(R);
}

// glm/detail/type_mat4x4.inl:630-648 : Result[c] = ((A0*B[c][0] + A1*B[c][1]) + A2*B[c][2]) + A3*B[c][3]
inline M4 mul44(const M4& a, const M4& b) {
    M4 r;
    for (int c = 0; c < 4; ++c)
        for (int rr = 0; rr < 4; ++rr)
            r.m[c][rr] = ((a.m[0][rr] * b.m[c][0] + a.m[1][rr] * b.m[c][1]) + a.m[2][rr] * b.m[c][2]) + a.m[3][rr] * b.m[c][3];
    return r;
}
inline M4 transpose4(const M4& a) { M4 r; for (int c = 0; c < 4; ++c) for (int rr = 0; rr < 4; ++rr) r.m[c][rr] = a.m[rr][c]; return r; }

} // namespace

// ---------------------------------------------------------------------------------------------
// Camera  (Camera.cpp:50-63; glm/ext/matrix_transform.inl:153-174; glm/ext/matrix_clip_space.inl:249-262)
// ---------------------------------------------------------------------------------------------
GS4DO_API void gs4do_look_at(const float eye[3], const float orientation[3], const float up[3], float view[16]) {
    V3 e = { eye[0], eye[1], eye[2] };
    V3 center = { eye[0] + orientation[0], eye[1] + orientation[1], eye[2] + orientation[2] };  // Camera.cpp:52
    V3 upv = { up[0], up[1], up[2] };
    V3 f = normalize3({ center.x - e.x, center.y - e.y, center.z - e.z });
    V3 s = normalize3(cross3(f, upv));
    V3 u = cross3(s, f);
    M4 R; memset(&R, 0, sizeof R); R.m[0][0] = R.m[1][1] = R.m[2][2] = R.m[3][3] = 1.0f;
    R.m[0][0] = s.x; R.m[1][0] = s.y; R.m[2][0] = s.z;
    R.m[0][1] = u.x; R.m[1][1] = u.y; R.m[2][1] = u.z;
    R.m[0][2] = -f.x; R.m[1][2] = -f.y; R.m[2][2] = -f.z;
    R.m[3][0] = -dot3(s, e); R.m[3][1] = -dot3(u, e); R.m[3][2] = dot3(f, e);
    memcpy(view, &R, sizeof R);
}

GS4DO_API void gs4do_perspective(float fov_deg, int width, int height, float znear, float zfar, float proj[16]) {
    float fovy = fov_deg * 0.01745329251994329576923690768489f;     // glm::radians
    float aspect = (float)width / (float)height;                     // Camera.cpp:57
    float t = tanf(fovy / 2.0f);
    M4 P; memset(&P, 0, sizeof P);
    P.m[0][0] = 1.0f / (aspect * t);
    P.m[1][1] = 1.0f / t;
    P.m[2][2] = -(zfar + znear) / (zfar - znear);
    P.m[2][3] = -1.0f;
    P.m[3][2] = -(2.0f * zfar * znear) / (zfar - znear);
    memcpy(proj, &P, sizeof P);
}

// ---------------------------------------------------------------------------------------------
// Host parameterisation (Splat.h)
// ---------------------------------------------------------------------------------------------
// Scenes.h:268 : glm::normalize(glm::quatLookAt(glm::normalize(n), vec3(0,1,0)))
GS4DO_API void gs4do_quat_look_at(const float dir[3], const float up[3], float q_wxyz[4]) {
    V3 d = normalize3({ dir[0], dir[1], dir[2] });
    Quat q = normalize_q(quatLookAtRH(d, { up[0], up[1], up[2] }));
    q_wxyz[0] = q.w; q_wxyz[1] = q.x; q_wxyz[2] = q.y; q_wxyz[3] = q.z;
}

// Splat.h:334-344 : Sigma3 = toMat3(q) * diag(s) * diag(s) * transpose(toMat3(q))   (no normalisation of q in the ctor)
GS4DO_API void gs4do_splat3d_cov(const float q_wxyz[4], const float scale[3], float cov[9]) {
    M3 S = diag3({ scale[0], scale[1], scale[2] });
    M3 R = mat3_cast({ q_wxyz[0], q_wxyz[1], q_wxyz[2], q_wxyz[3] });
    M3 g = mul33(mul33(mul33(R, S), S), transpose3(R));
    memcpy(cov, &g, sizeof g);
}

// Splat.h:132-159
GS4DO_API void gs4do_splat4d_cov(const float q_wxyz[4], const float scale[3], float lifetime, float fade, const float dir[3], float cov[16]) {
    const float STD_LOWER = 1.3862943611198906f;                                       // Splat.h:29
    // Splat.h:139 — log(fadeof) with a float argument resolves to the float overload (logf); the product with -2.0
    // makes the else-branch, hence the whole ?:, a double; the division is double and narrows to float.
    // (pinned by tests/golden/splat4d_ctor2_*: the all-double log variant mismatches 6 of 64 vectors)
    double den = (fade == 0.5f) ? (double)STD_LOWER : (-2.0 * (double)logf(fade));   // log(float) resolves to the float overload
    float sd = (float)((double)(lifetime * lifetime) / den);
    V3 tdir = { dir[0] * sd, dir[1] * sd, dir[2] * sd };
    M3 R = mat3_cast({ q_wxyz[0], q_wxyz[1], q_wxyz[2], q_wxyz[3] });
    M3 S = diag3({ scale[0], scale[1], scale[2] });
    M3 sig = mul33(mul33(mul33(R, S), S), transpose3(R));
    M3 op = outer3(tdir, tdir);
    float inv = 1.0f / sd;
    M3 up;
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) up.m[c][r] = sig.m[c][r] + op.m[c][r] * inv;    // scalar*mat = m[c]*scalar, then mat+mat
    M4 C;
    float td[3] = { tdir.x, tdir.y, tdir.z };
    for (int c = 0; c < 3; ++c) { for (int r = 0; r < 3; ++r) C.m[c][r] = up.m[c][r]; C.m[c][3] = td[c]; }
    C.m[3][0] = tdir.x; C.m[3][1] = tdir.y; C.m[3][2] = tdir.z; C.m[3][3] = sd;
    memcpy(cov, &C, sizeof C);
}

// Splat.h:91-130 (two-quaternion 4D rotation)
GS4DO_API void gs4do_splat4d_cov2q(const float q0_wxyz[4], const float q1_wxyz[4], const float scalar[4], float cov[16]) {
    Quat n0 = normalize_q({ q0_wxyz[0], q0_wxyz[1], q0_wxyz[2], q0_wxyz[3] });
    Quat n1 = normalize_q({ q1_wxyz[0], q1_wxyz[1], q1_wxyz[2], q1_wxyz[3] });
    float a = n0.w, b = n0.x, c = n0.y, d = n0.z;
    float p = n1.w, q = n1.x, r = n1.y, s = n1.z;
    // glm::mat4{...16 scalars...} fills column by column
    M4 Rl = { { { a, -b, -c, -d }, { b, a, -d, c }, { c, d, a, -b }, { d, -c, b, a } } };
    M4 Rr = { { { p, -q, -r, -s }, { q, p, s, -r }, { r, -s, p, q }, { s, r, -q, p } } };
    M4 Sc; memset(&Sc, 0, sizeof Sc);
    Sc.m[0][0] = scalar[0]; Sc.m[1][1] = scalar[1]; Sc.m[2][2] = scalar[2]; Sc.m[3][3] = scalar[3];
    M4 rot = mul44(Rl, Rr);
    M4 g = mul44(mul44(mul44(rot, Sc), transpose4(Sc)), transpose4(rot));
    memcpy(cov, &g, sizeof g);
}

// ---------------------------------------------------------------------------------------------
// Sort key (Scenes.h:28-36 GetMeanInTime, Scenes.h:314-319 key loop)
// record = 24 floats: pos[4], col[4], sig[16] (sig[c][r] at 8+4c+r)
// ---------------------------------------------------------------------------------------------
GS4DO_API void gs4do_keygen(const float* records, size_t n, float t, const float cam[3], uint32_t* idx_out, float* key_out) {
    for (size_t i = 0; i < n; ++i) {
        const float* rec = records + 24 * i;
        float ct = t - rec[3];
        float x = rec[0] + rec[8 + 12 + 0] * ct;      // sig[3].x
        float y = rec[1] + rec[8 + 12 + 1] * ct;
        float z = rec[2] + rec[8 + 12 + 2] * ct;
        float dx = x - cam[0], dy = y - cam[1], dz = z - cam[2];   // vec4 - vec4(camPos,1)
        idx_out[i] = (uint32_t)i;
        key_out[i] = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    }
}

// GS4D_KEY_VIEW_Z — the build's own extra key mode (the north star's "view-space depth keying"; the reference has no such key):
// key = 1 / max(-z_view, 1e-20) of the shader's time-conditioned mean  mu + (t - mu_t)/Sigma44 * Sigma[3].xyz
// (Splat4DVertexShaderInstanced.GLSL:86), z_view = row 2 of uView applied to it, summed left to right.  Nearer => larger key,
// so ascending order is far -> near like the reference key.  Restated here only so that the GPU mode has a checker.
GS4DO_API void gs4do_keygen_viewz(const float* records, size_t n, float t, const float V[16], uint32_t* idx_out, float* key_out) {
    for (size_t i = 0; i < n; ++i) {
        const float* rec = records + 24 * i;
        float k = (1.0f / rec[8 + 15]) * (t - rec[3]);
        float x = rec[0] + k * rec[8 + 12 + 0], y = rec[1] + k * rec[8 + 12 + 1], z = rec[2] + k * rec[8 + 12 + 2];
        float zv = ((V[2] * x + V[6] * y) + V[10] * z) + V[14];
        idx_out[i] = (uint32_t)i;
        key_out[i] = 1.0f / fmaxf(-zv, 1e-20f);
    }
}

// Sort contract: stable ascending by the uint32 bit pattern of the key, payload follows.
// This is the CPU "port" used as the checker and as the cpu_baseline sort stage: 4-pass 8-bit LSD.
GS4DO_API void gs4do_sort_pairs(uint32_t* keys, uint32_t* vals, size_t n) {
    if (n <= 1) return;                                 // radix_sort.hpp:260
    std::vector<uint32_t> k2(n), v2(n);
    uint32_t *ka = keys, *va = vals, *kb = k2.data(), *vb = v2.data();
    for (int pass = 0; pass < 4; ++pass) {
        size_t cnt[257] = { 0 };
        int sh = 8 * pass;
        for (size_t i = 0; i < n; ++i) cnt[((ka[i] >> sh) & 255u) + 1]++;
        for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
        for (size_t i = 0; i < n; ++i) { size_t p = cnt[(ka[i] >> sh) & 255u]++; kb[p] = ka[i]; vb[p] = va[i]; }
        std::swap(ka, kb); std::swap(va, vb);
    }
    // 4 passes: result is back in the caller's buffers
}

// Independent checker for the checker: std::stable_sort on (key, original position)
GS4DO_API void gs4do_sort_pairs_std(uint32_t* keys, uint32_t* vals, size_t n) {
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
    std::vector<uint32_t> k2(n), v2(n);
    for (size_t i = 0; i < n; ++i) { k2[i] = keys[order[i]]; v2[i] = vals[order[i]]; }
    memcpy(keys, k2.data(), 4 * n); memcpy(vals, v2.data(), 4 * n);
}

// CPU simulation of the reference's three GLSL compute kernels under the radix_sort.hpp driver
// (64 threads x 4 items = 256 keys per block, 4-bit digits, 8 passes).  Small n only.
GS4DO_API void gs4do_glsl_radix_sort(uint32_t* keys, uint32_t* vals, size_t n) {
    if (n <= 1) return;
    const uint32_t BLK = 256, RAD = 16;
    uint32_t blocks = (uint32_t)ceilf((float)n / (float)BLK);                         // radix_sort.hpp:181-184
    uint32_t p2 = (uint32_t)exp2(ceil(log2((double)blocks)));                         // round_to_power_of_2
    std::vector<uint32_t> kbuf[2] = { std::vector<uint32_t>(keys, keys + n), std::vector<uint32_t>(n) };
    std::vector<uint32_t> vbuf[2] = { std::vector<uint32_t>(vals, vals + n), std::vector<uint32_t>(n) };
    std::vector<uint32_t> local(p2 * RAD), glob(RAD);
    for (uint32_t pass = 0; pass < 8; ++pass) {
        std::fill(local.begin(), local.end(), 0u); std::fill(glob.begin(), glob.end(), 0u);
        const std::vector<uint32_t>& kin = kbuf[pass % 2]; const std::vector<uint32_t>& vin = vbuf[pass % 2];
        std::vector<uint32_t>& kout = kbuf[(pass + 1) % 2]; std::vector<uint32_t>& vout = vbuf[(pass + 1) % 2];
        // count (radix_sort_count.comp.glsl:37-52)
        for (size_t i = 0; i < n; ++i) { uint32_t rad = (kin[i] >> (4 * pass)) & 15u; local[rad * p2 + i / BLK]++; glob[rad]++; }
        // local offsets: Blelloch exclusive scan over blocks, per radix row (radix_sort_local_offsets.comp.glsl:89-165)
        uint32_t lg = (uint32_t)log2((double)p2);
        for (uint32_t d = 0; d < lg; ++d) { uint32_t step = 1u << d; for (uint32_t k = 0; k < p2; k += 2 * step) { uint32_t from = k + step - 1, to = from + step; if (to < p2) for (uint32_t r = 0; r < RAD; ++r) local[r * p2 + to] += local[r * p2 + from]; } }
        for (uint32_t r = 0; r < RAD; ++r) local[r * p2 + p2 - 1] = 0;
        for (int d = (int)lg - 1; d >= 0; --d) { uint32_t step = 1u << d; for (uint32_t k = 0; k < p2; k += 2 * step) { uint32_t from = k + step - 1, to = from + step; if (to < p2) for (uint32_t r = 0; r < RAD; ++r) { uint32_t t = local[r * p2 + to]; local[r * p2 + to] = local[r * p2 + from] + t; local[r * p2 + from] = t; } } }
        // reorder (radix_sort_reorder.comp.glsl:70-351)
        uint32_t goff[RAD]; { uint32_t s = 0; for (uint32_t i = 0; i < RAD; ++i) { goff[i] = s; s += glob[i]; } }
        for (uint32_t b = 0; b < blocks; ++b) {
            uint32_t skey[2][BLK], sidx[2][BLK];
            for (uint32_t l = 0; l < BLK; ++l) { size_t gi = (size_t)b * BLK + l; skey[0][l] = gi < n ? kin[gi] : 0xFFFFFFFFu; skey[1][l] = 0xFFFFFFFFu; sidx[0][l] = l; sidx[1][l] = 0xFFFFFFFFu; }
            uint32_t goffl[RAD] = { 0 };
            uint32_t bi;
            for (bi = 0; bi <= pass; ++bi) {
                uint32_t cur = bi % 2, nxt = (bi + 1) % 2;
                // exclusive prefix of the per-radix predicate == running count of equal radix before loc
                uint32_t run[RAD] = { 0 }; uint32_t pre[BLK];
                for (uint32_t l = 0; l < BLK; ++l) { uint32_t r = (skey[cur][l] >> (4 * bi)) & 15u; pre[l] = run[r]++; }
                uint32_t last = (b == blocks - 1) ? (uint32_t)(n - (size_t)(blocks - 1) * BLK - 1) : BLK - 1;
                // in_partition_group_off: counts up to and including `last` only (reorder:236-244)
                uint32_t cnt_upto[RAD] = { 0 };
                for (uint32_t l = 0; l <= last; ++l) cnt_upto[(skey[cur][l] >> (4 * bi)) & 15u]++;
                { uint32_t s = 0; for (uint32_t i = 0; i < RAD; ++i) { goffl[i] = s; s += cnt_upto[i]; } }
                for (uint32_t l = 0; l < BLK; ++l) {
                    uint32_t r = (skey[cur][l] >> (4 * bi)) & 15u;
                    uint32_t dest = goffl[r] + pre[l];
                    if (dest < BLK) { skey[nxt][dest] = skey[cur][l]; sidx[nxt][dest] = sidx[cur][l]; }
                }
            }
            uint32_t cur = bi % 2;
            for (uint32_t l = 0; l < BLK; ++l) {
                size_t gi = (size_t)b * BLK + l;
                if (gi < n) {
                    uint32_t k = skey[cur][l]; uint32_t r = (k >> (4 * pass)) & 15u;
                    uint32_t dest = goff[r] + local[r * p2 + b] + (l - goffl[r]);
                    kout[dest] = k; vout[dest] = vin[(size_t)b * BLK + sidx[cur][l]];
                }
            }
        }
    }
    memcpy(keys, kbuf[0].data(), 4 * n); memcpy(vals, vbuf[0].data(), 4 * n);
}

// ---------------------------------------------------------------------------------------------
// Vertex-stage restatement -> projected splat
// ---------------------------------------------------------------------------------------------
// Compact projected record: exactly the fields the HIP preprocess kernel writes (compared bit-for-bit
// except `alpha`, which carries an exp()).
struct gs4do_proj {
    float cx, cy;         // quad centre in window pixels (x right, y up, pixel centres at +0.5)
    float a0x, a0y;       // quad-local u = fmaf(a0x, dx, a0y*dy)
    float a1x, a1y;       // quad-local v = fmaf(a1x, dx, a1y*dy)
    float r, g, b;        // oColor.rgb
    float alpha;          // per-splat alpha factor: 4D oTimeOpacity*oColor.a ; 3D/2D oColor.a
    float hx, hy;         // conservative half extents of the quad's pixel bounding box
    uint32_t valid;       // 0: culled / produces no fragments
    // GLSL-literal extras used only by the oracle's fragment stage
    float e0x, e0y, e1x, e1y, s0, s1; // R columns, S diagonal as used in "R*S" (2D path: s0=l1, s1=l0)
    float q00, q01, q10, q11;          // oSig = inverse(R*S*S*transpose(R)), q[c][r]
};

namespace {

inline float maxf_glsl(float a, float b) { return a >= b ? a : b; }   // Splat4DVertexShaderInstanced.GLSL:53-56

// GetEigenValues2x2 + GetEigenVectors2x2 + main():132-143 ; returns false if anything is non-finite
struct Eig { float e0x, e0y, e1x, e1y, l0, l1; };
inline Eig eigen2(float u00, float u01, float u10, float u11, bool guard_zero_offdiag) {
    Eig E;
    float m = (u00 + u11) * 0.5f;
    float p = (u00 * u11) - (u01 * u10);
    float d = sqrtf((m * m) - p);
    float lx = maxf_glsl(m - d, 0.000001f), ly = maxf_glsl(m + d, 0.000001f);
    V2 ev0, ev1;
    if (guard_zero_offdiag && u01 == 0.0f) { ev0 = { 1, 0 }; ev1 = { 0, 1 }; }       // Splat2DVSI.GLSL:49-52
    else {
        ev0 = normalize2({ u01, lx - u00 });
        ev1 = { ev0.y, -ev0.x };
        ev0 = normalize2(ev0); ev1 = normalize2(ev1);                               // :77
    }
    ev0 = normalize2(ev0); ev1 = normalize2(ev1);                                   // :137-138
    E.e0x = ev0.x; E.e0y = ev0.y; E.e1x = ev1.x; E.e1y = ev1.y; E.l0 = lx; E.l1 = ly;
    return E;
}

// inverse(R*S*S*transpose(R)) with GLSL/GLM mat2 semantics (column-major)
inline void conic(const Eig& E, float s0, float s1, float q[2][2]) {
    // R = mat2(v0, v1): R[0]=v0, R[1]=v1.  S = mat2(s0,0,0,s1).
    // (R*S)[c][r] = R[0][r]*S[c][0] + R[1][r]*S[c][1]
    float R[2][2] = { { E.e0x, E.e0y }, { E.e1x, E.e1y } };
    float S[2][2] = { { s0, 0.0f }, { 0.0f, s1 } };
    float A[2][2], B[2][2], C[2][2], Rt[2][2] = { { R[0][0], R[1][0] }, { R[0][1], R[1][1] } };
    auto mul = [](float X[2][2], float Y[2][2], float Z[2][2]) { for (int c = 0; c < 2; ++c) for (int r = 0; r < 2; ++r) Z[c][r] = X[0][r] * Y[c][0] + X[1][r] * Y[c][1]; };
    mul(R, S, A); mul(A, S, B); mul(B, Rt, C);
    float det = C[0][0] * C[1][1] - C[1][0] * C[0][1];
    float od = 1.0f / det;                                                         // glm/GLSL inverse(mat2)
    q[0][0] = C[1][1] * od; q[0][1] = -C[0][1] * od; q[1][0] = -C[1][0] * od; q[1][1] = C[0][0] * od;
}

inline bool finite6(float a, float b, float c, float d, float e, float f) { return std::isfinite(a) && std::isfinite(b) && std::isfinite(c) && std::isfinite(d) && std::isfinite(e) && std::isfinite(f); }

// Window-space set-up shared by all modes.  THIS is the build's statement of the fixed-function
// viewport transform + quad set-up (R1 in SURVEY.md §8a); the HIP preprocess kernel performs the same
// operations in the same order.
//   ndc centre (ncx, ncy); per-axis NDC scale of the quad offset (kx, ky):  ndc = nc + k * (R*S*v)
inline void window_setup(gs4do_proj& o, float ncx, float ncy, float kx, float ky, int W, int H) {
    float hw = (float)W * 0.5f, hh = (float)H * 0.5f;
    float sx = kx * hw, sy = ky * hh;
    o.cx = fmaf(ncx, hw, hw);
    o.cy = fmaf(ncy, hh, hh);
    float r0 = 1.0f / o.s0, r1 = 1.0f / o.s1;
    o.a0x = (o.e0x * r0) / sx; o.a0y = (o.e0y * r0) / sy;
    o.a1x = (o.e1x * r1) / sx; o.a1y = (o.e1y * r1) / sy;
    o.hx = 0.5f * (fabsf(o.e0x) * o.s0 + fabsf(o.e1x) * o.s1) * fabsf(sx);
    o.hy = 0.5f * (fabsf(o.e0y) * o.s0 + fabsf(o.e1y) * o.s1) * fabsf(sy);
    if (!finite6(o.cx, o.cy, o.a0x, o.a0y, o.a1x, o.a1y) || !std::isfinite(o.hx) || !std::isfinite(o.hy)) o.valid = 0;
}

// mat4 * vec4 as written in the shader; evaluation order ((m0*x + m1*y) + m2*z) + m3*w
inline V4 mulM4V4(const float M[16], V4 v) {
    V4 r;
    r.x = ((M[0] * v.x + M[4] * v.y) + M[8] * v.z) + M[12] * v.w;
    r.y = ((M[1] * v.x + M[5] * v.y) + M[9] * v.z) + M[13] * v.w;
    r.z = ((M[2] * v.x + M[6] * v.y) + M[10] * v.z) + M[14] * v.w;
    r.w = ((M[3] * v.x + M[7] * v.y) + M[11] * v.z) + M[15] * v.w;
    return r;
}

// Shared 3D tail: main():97-149 of Splat4DVertexShaderInstanced.GLSL == main():45-97 of Splat3DVertexShaderFull.GLSL
// mean = world position, C = 3x3 covariance C[c][r]
inline void project3d(gs4do_proj& o, V3 mean, const float C[3][3], const float V[16], const float P[16], int W, int H) {
    V4 pc = mulM4V4(V, { mean.x, mean.y, mean.z, 1.0f });
    V4 ps = mulM4V4(P, pc);
    float rw = 1.0f / ps.w;
    ps = { rw * ps.x, rw * ps.y, rw * ps.z, rw * ps.w };
    float z = ps.z / ps.w;
    float bound = 1.2f * ps.w;
    // negated form so that NaNs cull (GLSL comparisons with NaN are false => the reference would not cull, but a
    // NaN position produces no fragments anyway)
    if (z < 0.0f || z > 1.0f || ps.x < -bound || ps.x > bound || ps.y < -bound || ps.y > bound) { o.valid = 0; return; }
    if (!(std::isfinite(ps.x) && std::isfinite(ps.y) && std::isfinite(z))) { o.valid = 0; return; }
    // gl_Position = uProj * vec4(R*S*v, 0, 1) + ps (Splat4DVertexShaderInstanced.GLSL:147): every corner has z = ps.z + uProj[3][2] at
    // w = 1, and the fixed-function clipper keeps -w <= z <= w (OpenGL 4.4, 13.5) — the whole quad or nothing
    { const float zq = ps.z + P[14]; if (zq < -1.0f || zq > 1.0f) { o.valid = 0; return; } }
    float z2 = pc.z * pc.z;
    // J columns (1/z, 0, -x/z2), (0, 1/z, -y/z2), (0,0,0)
    float J[3][3] = { { 1.0f / pc.z, 0.0f, -pc.x / z2 }, { 0.0f, 1.0f / pc.z, -pc.y / z2 }, { 0.0f, 0.0f, 0.0f } };
    // W = mat3(uView); T = transpose(W) * J ; transpose(W)[c][r] = W[r][c] = V[4r + c]
    float Wt[3][3]; for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) Wt[c][r] = V[4 * r + c];
    float T[3][3], Tt[3][3], A[3][3], cov3[3][3];
    auto mul = [](float X[3][3], const float Y[3][3], float Z[3][3]) { for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) Z[c][r] = X[0][r] * Y[c][0] + X[1][r] * Y[c][1] + X[2][r] * Y[c][2]; };
    mul(Wt, J, T);
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) Tt[c][r] = T[r][c];
    mul(Tt, C, A);         // transpose(T) * sig3x3
    mul(A, T, cov3);       // ... * T
    Eig E = eigen2(cov3[0][0], cov3[0][1], cov3[1][0], cov3[1][1], false);
    float l0 = sqrtf(E.l0), l1 = sqrtf(E.l1);
    o.e0x = E.e0x; o.e0y = E.e0y; o.e1x = E.e1x; o.e1y = E.e1y; o.s0 = l0; o.s1 = l1;
    float q[2][2]; conic(E, l0, l1, q);
    o.q00 = q[0][0]; o.q01 = q[0][1]; o.q10 = q[1][0]; o.q11 = q[1][1];
    // gl_Position = uProj * vec4(R*S*v, 0, 1) + ps : x = P00*g.x (+P[1][0]*g.y + P[3][0], zero for glm::perspective), w = ps.w (~1)
    window_setup(o, ps.x, ps.y, P[0], P[5], W, H);
}

} // namespace

// mode ids shared with include/gs4d.h
enum { GS4DO_MODE_4D = 0, GS4DO_MODE_3D = 2, GS4DO_MODE_2D = 3 };

// 4D record (24 floats).  Splat4DVertexShaderInstanced.GLSL:81-150
GS4DO_API void gs4do_preprocess_4d(const float* records, size_t n, float t, float min_opacity, const float V[16], const float P[16], int W, int H, gs4do_proj* out) {
    for (size_t i = 0; i < n; ++i) {
        const float* rec = records + 24 * i; const float* S = rec + 8;             // S[4c + r]
        gs4do_proj o; memset(&o, 0, sizeof o); o.valid = 1;
        float s44 = S[15];
        float dt = t - rec[3];
        float ot = maxf_glsl(expf(-0.5f * dt * (1.0f / s44) * dt), min_opacity);   // :48-51, 83  (((-0.5*dt)*(1/s44))*dt)
        V3 a = { S[3], S[7], S[11] };                                              // iSig[0][3], [1][3], [2][3]
        V3 b = { S[12], S[13], S[14] };                                            // iSig[3][0..2]
        float k = (1.0f / s44) * dt;                                               // (1/S44) * (uTime - mu_t)
        V3 mean = { rec[0] + k * a.x, rec[1] + k * a.y, rec[2] + k * a.z };         // :86
        float inv = 1.0f / s44;
        V3 tv = { inv * b.x, inv * b.y, inv * b.z };                               // :87
        float C[3][3]; float av[3] = { a.x, a.y, a.z }, tvv[3] = { tv.x, tv.y, tv.z };
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) C[c][r] = S[4 * c + r] - av[r] * tvv[c];   // :89-95 outerProduct(a,tv)[c] = a*tv[c]
        o.r = rec[4]; o.g = rec[5]; o.b = rec[6];
        o.alpha = ot * rec[7];
        project3d(o, mean, C, V, P, W, H);
        if (!std::isfinite(o.alpha)) o.valid = 0;
        out[i] = o;
    }
}

// 3D "Full" path: 4 vertices x 18 floats per splat {vpos2, spos3, col4, sig9}; the four vertices of a quad carry
// identical splat attributes (Splat.h:433-447), vertex 0 is read.  Splat3DVertexShaderFull.GLSL:43-98
GS4DO_API void gs4do_preprocess_3d(const float* verts, size_t n, const float V[16], const float P[16], int W, int H, gs4do_proj* out) {
    for (size_t i = 0; i < n; ++i) {
        const float* v = verts + 72 * i;
        gs4do_proj o; memset(&o, 0, sizeof o); o.valid = 1;
        float C[3][3]; for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) C[c][r] = v[9 + 3 * c + r];
        o.r = v[5]; o.g = v[6]; o.b = v[7]; o.alpha = v[8];
        project3d(o, { v[2], v[3], v[4] }, C, V, P, W, H);
        out[i] = o;
    }
}

// 2D path: 12 floats per record {pos4, col4, sig(mat2 as vec4)}.  Splat2DVSI.GLSL:59-94 (SCREEN_SPACE_POS)
GS4DO_API void gs4do_preprocess_2d(const float* records, size_t n, const float V[16], const float P[16], int W, int H, gs4do_proj* out) {
    (void)V;
    for (size_t i = 0; i < n; ++i) {
        const float* rec = records + 12 * i;
        gs4do_proj o; memset(&o, 0, sizeof o); o.valid = 1;
        V4 ps = mulM4V4(P, { rec[0], rec[1], -1.0f, 1.0f });                        // :64
        float rw = 1.0f / ps.w; ps = { rw * ps.x, rw * ps.y, rw * ps.z, rw * ps.w };
        Eig E = eigen2(rec[8], rec[9], rec[10], rec[11], true);                     // geoinf = mat2(sig.xy, sig.zw)
        float l0 = sqrtf(E.l0 * 2.0f), l1 = sqrtf(E.l1 * 2.0f);                     // :68-69
        o.e0x = E.e0x; o.e0y = E.e0y; o.e1x = E.e1x; o.e1y = E.e1y; o.s0 = l1; o.s1 = l0;   // S = mat2(l1,0,0,l0) :76
        float q[2][2]; conic(E, l1, l0, q);
        o.q00 = q[0][0]; o.q01 = q[0][1]; o.q10 = q[1][0]; o.q11 = q[1][1];
        o.r = rec[4]; o.g = rec[5]; o.b = rec[6]; o.alpha = rec[7];
        // gl_Position = uProj * (vec4(R*S*v, -5, 1) + ps): clip.x = P00*(g.x+ps.x), clip.z = P22*(ps.z-5) + P32*(1+ps.w), clip.w = -(ps.z-5)
        float zc = -5.0f + ps.z, wc4 = 1.0f + ps.w;
        float clipw = P[11] * zc + P[15] * wc4;
        float clipz = P[10] * zc + P[14] * wc4;
        if (!(clipw > 0.0f) || clipz < -clipw || clipz > clipw) { o.valid = 0; out[i] = o; continue; }   // whole quad shares z,w
        float kx = P[0] / clipw, ky = P[5] / clipw;
        window_setup(o, kx * ps.x, ky * ps.y, kx, ky, W, H);
        out[i] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// Rasteriser coverage rule (the build's statement of R1) + fragment stage + blend
// ---------------------------------------------------------------------------------------------
// Pixel (i,j) (j counted from the bottom row, GL window origin) is covered iff, with
//   dx = (i + 0.5f) - cx,  dy = (j + 0.5f) - cy,
//   u  = fmaf(a0x, dx, a0y*dy),  v = fmaf(a1x, dx, a1y*dy),
// |u| <= 0.5 and |v| <= 0.5 (u,v are the interpolated quad-local iVPos).
static inline bool gs4do_covered(const gs4do_proj& p, int i, int j, float& u, float& v) {
    float dx = ((float)i + 0.5f) - p.cx, dy = ((float)j + 0.5f) - p.cy;
    u = fmaf(p.a0x, dx, p.a0y * dy);
    v = fmaf(p.a1x, dx, p.a1y * dy);
    return fabsf(u) <= 0.5f && fabsf(v) <= 0.5f;
}

// frag_mode: 0 = 4D (Splat4DFragShader), 2 = 3D-Full (colour premultiplied by c), 3 = 2D (oSig*x column form)
static inline bool gs4do_fragment(const gs4do_proj& p, int frag_mode, float u, float v, float src[4]) {
    // oFragPos = ((R*8)*S)*iVPos, interpolated linearly (w == 1 for every corner)
    float r00 = p.e0x * 8.0f, r01 = p.e0y * 8.0f, r10 = p.e1x * 8.0f, r11 = p.e1y * 8.0f;     // R*8
    float m00 = r00 * p.s0, m01 = r01 * p.s0, m10 = r10 * p.s1, m11 = r11 * p.s1;             // (R*8)*S (S diagonal; zero terms dropped)
    float x = m00 * u + m10 * v, y = m01 * u + m11 * v;
    float sx, sy;
    if (frag_mode == GS4DO_MODE_2D) { sx = p.q00 * x + p.q10 * y; sy = p.q01 * x + p.q11 * y; }      // oSig * x
    else                            { sx = x * p.q00 + y * p.q01; sy = x * p.q10 + y * p.q11; }      // x * oSig
    float c = expf(-0.5f * (sx * x + sy * y));
    if (!(c >= 0.0001f)) return false;                                                        // discard (NaN discards too)
    if (frag_mode == GS4DO_MODE_3D) { src[0] = c * p.r; src[1] = c * p.g; src[2] = c * p.b; src[3] = c * p.alpha; }
    else { src[0] = p.r; src[1] = p.g; src[2] = p.b; src[3] = p.alpha * c; }                   // 4D: oTimeOpacity*c*oColor.w
    return true;
}

// glBlendFunc(sfactor, dfactor) with the default equation FUNC_ADD (OpenGL 4.4, 17.3.8, tables 17.1/17.2), the factors the reference's
// menu offers (DebugMenus.h:41-59), applied to all four channels (glBlendFunc sets the RGB and the alpha factors alike).  Fixed-point
// framebuffer: the source is clamped before (done by the callers), the result after.  The blend colour is never set by the reference
// (no glBlendColor call): it stays (0, 0, 0, 0), so CONSTANT_* = 0 and ONE_MINUS_CONSTANT_* = 1.
static inline float gs4do_blend_factor(int f, int ch, const float src[4], const float dst[4]) {
    switch (f) {
    case 0:      return 0.0f;                 // GL_ZERO
    case 1:      return 1.0f;                 // GL_ONE
    case 0x0300: return src[ch];              // GL_SRC_COLOR
    case 0x0301: return 1.0f - src[ch];       // GL_ONE_MINUS_SRC_COLOR
    case 0x0302: return src[3];               // GL_SRC_ALPHA
    case 0x0303: return 1.0f - src[3];        // GL_ONE_MINUS_SRC_ALPHA
    case 0x0304: return dst[3];               // GL_DST_ALPHA
    case 0x0305: return 1.0f - dst[3];        // GL_ONE_MINUS_DST_ALPHA
    case 0x0306: return dst[ch];              // GL_DST_COLOR
    case 0x0307: return 1.0f - dst[ch];       // GL_ONE_MINUS_DST_COLOR
    case 0x8001: case 0x8003: return 0.0f;    // GL_CONSTANT_COLOR / GL_CONSTANT_ALPHA
    case 0x8002: case 0x8004: return 1.0f;    // GL_ONE_MINUS_CONSTANT_*
    default:     return 0.0f;
    }
}
static inline void gs4do_blend(int sf, int df, const float src[4], float* d) {
    float out[4];
    for (int q = 0; q < 4; ++q) {
        const float S = gs4do_blend_factor(sf, q, src, d), D = gs4do_blend_factor(df, q, src, d);
        out[q] = fminf(fmaxf(src[q] * S + d[q] * D, 0.0f), 1.0f);
    }
    d[0] = out[0]; d[1] = out[1]; d[2] = out[2]; d[3] = out[3];
}

// Ordered "over" blend into an RGBA32F image (row 0 = bottom), instance order = order[k] (or k if order == NULL).
// Application.cpp:150-154 : dst = src*src.a + dst*(1-src.a) on all four channels, depth test off.
GS4DO_API void gs4do_composite_blend(const gs4do_proj* proj, const uint32_t* order, size_t ninst, int frag_mode, int W, int H, float* rgba, int nthreads, int sf, int df);
GS4DO_API void gs4do_composite(const gs4do_proj* proj, const uint32_t* order, size_t ninst, int frag_mode, int W, int H, float* rgba, int nthreads) {
    gs4do_composite_blend(proj, order, ninst, frag_mode, W, H, rgba, nthreads, 0x0302, 0x0303);
}
// The same with any blend function of the menu (Application.cpp:150 glBlendFunc(blendOpt.selected0, blendOpt.selected1)).
GS4DO_API void gs4do_composite_blend(const gs4do_proj* proj, const uint32_t* order, size_t ninst, int frag_mode, int W, int H, float* rgba, int nthreads, int sf, int df) {
    const bool over = sf == 0x0302 && df == 0x0303;
    if (nthreads < 1) nthreads = 1;
    auto band = [&](int j0, int j1) {
        for (size_t k = 0; k < ninst; ++k) {
            const gs4do_proj& p = proj[order ? order[k] : k];
            if (!p.valid) continue;
            // generous candidate box (+1.5 px); the predicate decides
            float fx0 = p.cx - p.hx - 1.5f, fx1 = p.cx + p.hx + 1.5f, fy0 = p.cy - p.hy - 1.5f, fy1 = p.cy + p.hy + 1.5f;
            if (!(fx1 >= 0.0f && fy1 >= (float)j0 && fx0 <= (float)W && fy0 <= (float)j1)) continue;
            int i0 = (int)fmaxf(0.0f, floorf(fx0)), i1 = (int)fminf((float)(W - 1), ceilf(fx1));
            int jj0 = (int)fmaxf((float)j0, floorf(fy0)), jj1 = (int)fminf((float)(j1 - 1), ceilf(fy1));
            for (int j = jj0; j <= jj1; ++j)
                for (int i = i0; i <= i1; ++i) {
                    float u, v, src[4];
                    if (!gs4do_covered(p, i, j, u, v)) continue;
                    if (!gs4do_fragment(p, frag_mode, u, v, src)) continue;
                    // the reference's framebuffer is fixed-point (RGBA8): fragment colour and alpha are clamped to [0, 1] before the blend (OpenGL 4.4, 17.3.8)
                    for (int q = 0; q < 4; ++q) src[q] = fminf(fmaxf(src[q], 0.0f), 1.0f);
                    float* d = rgba + 4 * ((size_t)j * W + i);
                    if (!over) { gs4do_blend(sf, df, src, d); continue; }
                    float a = src[3], ia = 1.0f - a;
                    d[0] = src[0] * a + d[0] * ia; d[1] = src[1] * a + d[1] * ia; d[2] = src[2] * a + d[2] * ia; d[3] = src[3] * a + d[3] * ia;
                }
        }
    };
    if (nthreads == 1) { band(0, H); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) { int j0 = (int)((int64_t)H * t / nthreads), j1 = (int)((int64_t)H * (t + 1) / nthreads); th.emplace_back(band, j0, j1); }
    for (auto& x : th) x.join();
}

// ---- overlay lines: Renderer::DrawLine / DrawGrid / DrawAxis (Renderer.cpp:41-215), Shader/Lines/LineVert.GLSL:11, Line2DVert.GLSL:11 ----
// The vertex stage is the shader's (gl_Position = uViewProj * vec4(aPos, 1) resp. vec4(aPos, 0, 1)); clipping, line rasterisation and
// the blend are fixed-function in the reference: restated from the OpenGL 4.4 core specification (13.5 clipping, 14.5.2.1 / 14.5.2.2
// line segments and wide lines, 17.3.8 blending) in the form csrc/lines.hip documents.  Pinned against the reference's line programs run by llvmpipe (tests/golden/gl_lines_*.npz, test_oracle_gl.py): the reference holds no line
// images.  Fragments are blended one after another, segment by segment — the order the GL would produce them in.
static bool clip_t(float num, float den, float& t0, float& t1) {
    if (den == 0.0f) return num >= 0.0f;
    const float t = -num / den;
    if (den > 0.0f) { if (t > t1) return false; if (t > t0) t0 = t; }
    else { if (t < t0) return false; if (t < t1) t1 = t; }
    return true;
}
GS4DO_API void gs4do_draw_lines_blend(float* rgba, int W, int H, const float* verts, size_t nverts, int dims, int strip, const float* M, const float color[4], float width, int sf, int df);
GS4DO_API void gs4do_draw_lines(float* rgba, int W, int H, const float* verts, size_t nverts, int dims, int strip, const float* M, const float color[4], float width) {
    gs4do_draw_lines_blend(rgba, W, H, verts, nverts, dims, strip, M, color, width, 0x0302, 0x0303);
}
GS4DO_API void gs4do_draw_lines_blend(float* rgba, int W, int H, const float* verts, size_t nverts, int dims, int strip, const float* M, const float color[4], float width, int sf, int df) {
    const bool over = sf == 0x0302 && df == 0x0303;
    const size_t nseg = strip ? (nverts >= 2 ? nverts - 1 : 0) : nverts / 2;
    int wpx = (int)floorf(width + 0.5f); if (!(wpx >= 1)) wpx = 1; if (wpx > 64) wpx = 64;
    float col[4]; for (int q = 0; q < 4; ++q) col[q] = fminf(fmaxf(color[q], 0.0f), 1.0f);      // clamped like every fragment colour
    const float a = col[3], om = 1.0f - a;
    const float sr = col[0] * a, sg = col[1] * a, sb = col[2] * a, sa = a * a;
    for (size_t s = 0; s < nseg; ++s) {
        const size_t i0 = strip ? s : 2 * s, i1 = i0 + 1;
        float c0[4], c1[4];
        if (dims == 3) {
            const float* p = verts + 3 * i0; const float* q = verts + 3 * i1;
            for (int r = 0; r < 4; ++r) {
                c0[r] = ((M[r] * p[0] + M[4 + r] * p[1]) + M[8 + r] * p[2]) + M[12 + r] * 1.0f;
                c1[r] = ((M[r] * q[0] + M[4 + r] * q[1]) + M[8 + r] * q[2]) + M[12 + r] * 1.0f;
            }
        } else {
            c0[0] = verts[2 * i0]; c0[1] = verts[2 * i0 + 1]; c0[2] = 0.0f; c0[3] = 1.0f;
            c1[0] = verts[2 * i1]; c1[1] = verts[2 * i1 + 1]; c1[2] = 0.0f; c1[3] = 1.0f;
        }
        float t0 = 0.0f, t1 = 1.0f;
        bool vis = true;
        for (int ax = 0; ax < 3; ++ax) {
            vis = vis && clip_t(c0[3] + c0[ax], (c1[3] - c0[3]) + (c1[ax] - c0[ax]), t0, t1);
            vis = vis && clip_t(c0[3] - c0[ax], (c1[3] - c0[3]) - (c1[ax] - c0[ax]), t0, t1);
        }
        if (!vis || !(t0 <= t1)) continue;
        float e0[4], e1[4];
        for (int r = 0; r < 4; ++r) { const float d = c1[r] - c0[r]; e0[r] = c0[r] + t0 * d; e1[r] = c0[r] + t1 * d; }
        if (!(e0[3] > 0.0f) || !(e1[3] > 0.0f)) continue;
        const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
        const float gax = (e0[0] / e0[3] + 1.0f) * hw, gay = (e0[1] / e0[3] + 1.0f) * hh;
        const float gbx = (e1[0] / e1[3] + 1.0f) * hw, gby = (e1[1] / e1[3] + 1.0f) * hh;
        if (!(std::isfinite(gax) && std::isfinite(gay) && std::isfinite(gbx) && std::isfinite(gby))) continue;
        const bool xmajor = fabsf(gbx - gax) >= fabsf(gby - gay);
        float ma = xmajor ? gax : gay, mb = xmajor ? gbx : gby, na = xmajor ? gay : gax, nb = xmajor ? gby : gbx;
        if (ma > mb) { std::swap(ma, mb); std::swap(na, nb); }
        if (!(mb > ma)) continue;
        const float first = ceilf(ma - 0.5f), last = ceilf(mb - 0.5f);
        for (float i = first; i < last; i += 1.0f) {
            const float t = ((i + 0.5f) - ma) / (mb - ma);
            const float minor = (na + t * (nb - na)) - 0.5f * (float)(wpx - 1);
            for (int k = 0; k < wpx; ++k) {
                const float j = (ceilf(minor) - 1.0f) + (float)k;       // a line ON a pixel boundary belongs to the pixel below/left: GL perturbs by (-eps, -eps^2) (14.5.2.1)
                const float x = xmajor ? i : j, y = xmajor ? j : i;
                if (!(x >= 0.0f && y >= 0.0f && x < (float)W && y < (float)H)) continue;
                float* d = rgba + 4 * ((size_t)(int)y * W + (int)x);
                if (!over) { gs4do_blend(sf, df, col, d); continue; }
                d[0] = sr + d[0] * om; d[1] = sg + d[1] * om; d[2] = sb + d[2] * om; d[3] = sa + d[3] * om;
            }
        }
    }
}

GS4DO_API void gs4do_clear(float* rgba, int W, int H, const float clear[4]) {
    for (size_t i = 0; i < (size_t)W * H; ++i) { rgba[4 * i] = clear[0]; rgba[4 * i + 1] = clear[1]; rgba[4 * i + 2] = clear[2]; rgba[4 * i + 3] = clear[3]; }
}

GS4DO_API size_t gs4do_proj_size(void) { return sizeof(gs4do_proj); }

// Copy the compact fields out as flat float/uint arrays for the tests: out13[n][13] = cx,cy,a0x,a0y,a1x,a1y,r,g,b,alpha,hx,hy,valid(as float)
GS4DO_API void gs4do_proj_compact(const gs4do_proj* p, size_t n, float* out13) {
    for (size_t i = 0; i < n; ++i) { const gs4do_proj& o = p[i]; float* d = out13 + 13 * i;
        d[0] = o.cx; d[1] = o.cy; d[2] = o.a0x; d[3] = o.a0y; d[4] = o.a1x; d[5] = o.a1y; d[6] = o.r; d[7] = o.g; d[8] = o.b; d[9] = o.alpha; d[10] = o.hx; d[11] = o.hy; d[12] = o.valid ? 1.0f : 0.0f; }
}

// ---------------------------------------------------------------------------------------------
// Whole frame (Scenes.h:301-340 Render, with the key loop + sort of :312-328 when do_sort != 0)
// returns per-stage wall times in ms: [keygen, sort, preprocess, composite]
// ---------------------------------------------------------------------------------------------
#include <chrono>
GS4DO_API void gs4do_render_4d(const float* records, size_t n, int do_sort, float t, float min_opacity, const float cam[3], const float V[16], const float P[16],
                               int W, int H, const float clear[4], float* rgba, uint32_t* perm_out, int nthreads, double stage_ms[4]) {
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::vector<uint32_t> idx(n); std::vector<float> key(n);
    auto t0 = clk::now();
    if (do_sort) gs4do_keygen(records, n, t, cam, idx.data(), key.data()); else for (size_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
    auto t1 = clk::now();
    if (do_sort) gs4do_sort_pairs(reinterpret_cast<uint32_t*>(key.data()), idx.data(), n);
    auto t2 = clk::now();
    std::vector<gs4do_proj> proj(n);
    if (nthreads <= 1) gs4do_preprocess_4d(records, n, t, min_opacity, V, P, W, H, proj.data());
    else {
        std::vector<std::thread> th;
        for (int k = 0; k < nthreads; ++k) { size_t a = n * k / nthreads, b = n * (k + 1) / nthreads; th.emplace_back([=, &proj] { gs4do_preprocess_4d(records + 24 * a, b - a, t, min_opacity, V, P, W, H, proj.data() + a); }); }
        for (auto& x : th) x.join();
    }
    auto t3 = clk::now();
    gs4do_clear(rgba, W, H, clear);
    gs4do_composite(proj.data(), idx.data(), n, GS4DO_MODE_4D, W, H, rgba, nthreads);
    auto t4 = clk::now();
    if (perm_out) memcpy(perm_out, idx.data(), 4 * n);
    if (stage_ms) { stage_ms[0] = ms(t0, t1); stage_ms[1] = ms(t1, t2); stage_ms[2] = ms(t2, t3); stage_ms[3] = ms(t3, t4); }
}
