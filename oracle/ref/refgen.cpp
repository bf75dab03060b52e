// refgen.cpp — golden-vector generator that runs the REFERENCE's own host C++ (compiled from the
// sources where they lie under /root/reference; nothing is copied into this repo).
//
// Test infrastructure.  Built by oracle/Makefile into oracle/_ref/refgen (git-ignored), run by
// oracle/make_golden.sh, which writes the small fixtures under tests/golden/.
//
// What is executed from the reference (all arithmetic below happens inside reference/GLM code):
//   Splat4D::Splat4D (both ctors)      4DSplatRendering/Splat.h:91-159
//   Splat3D::Splat3D                   4DSplatRendering/Splat.h:334-344
//   Camera::GetViewMatrix/GetProjMatrix 4DSplatRendering/Camera.cpp:50-58
//   VData::parse / parse_splat_data    4DSplatRendering/VDataParser.h:25-58, 60-123
//   Scenes::GetModelExtrema/GetColor   4DSplatRendering/Scenes.h:58-91   (+ Utils.cpp lerp/mapf/minf/maxf)
//   Scenes::SplatData::GetMeanInTime   4DSplatRendering/Scenes.h:28-36
//   glm::quatLookAt / normalize / rotate (vendored GLM 0.9.9.9)
// What this harness restates itself (loops only, no arithmetic of its own): the per-splat generation loops of
// LinearMotion::init (Scenes.h:258-279), NonLinearMotion (:517-545), RotationMotion (:775-803), CombinedMotion (:1035-1068),
// BrokenMotion (:1965-1989), SquareMotion (:2216-2259) and the key loop
// (Scenes.h:314-319), because those bodies sit inside methods that also call OpenGL.
// Not buildable here, therefore not used: anything that needs an OpenGL context or GLEW/GLFW/ImGui
// libraries (Renderer.cpp, Shader.cpp, the scene classes' init/Render, radix_sort.hpp) and all GLSL.
//
// The include block follows Application.cpp:10-57 (order matters: Scenes.h relies on earlier includes).
#include <GLEW/glew.h>
#include <GLFW/glfw3.h>
#include <stdlib.h>
#include <iostream>
#include <fstream>
#include <string>
#include <sstream>
#include <algorithm>
#include <chrono>
#include <functional>
#include <memory>

#include "Camera.h"
#include "Renderer.h"
#include "VertexBuffer.h"
#include "IndexBuffer.h"
#include "VertexArray.h"
#include "VertexBufferLayout.h"
#include "Shader.h"
#include "Geometry.h"
#include "glm/glm.hpp"
#include "glm/gtc/matrix_transform.hpp"
#include <glm/gtc/quaternion.hpp>
#include <glm/common.hpp>
#include <glm/gtx/matrix_decompose.hpp>
#include <glm/gtx/matrix_operation.hpp>
#include "Splat.h"
#include "imgui.h"
#include "Utils.h"
#include "radix_sort.hpp"
#include "ShareStorageBuffer.h"
#include "VDataParser.h"
#include "Scene.h"
#include "Scenes.h"

#include <cstdio>
#include <cstdint>
#include <vector>

static std::string g_out;
static FILE* g_manifest = nullptr;
static bool g_first = true;

static uint32_t crc32_buf(const void* data, size_t n) {
    static uint32_t table[256]; static bool init = false;
    if (!init) { for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; } init = true; }
    uint32_t c = 0xFFFFFFFFu; const uint8_t* p = (const uint8_t*)data;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 255] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

static void dump(const char* name, const char* dtype, const void* data, size_t count, size_t cols) {
    std::string path = g_out + "/" + name + ".bin";
    FILE* f = fopen(path.c_str(), "wb"); if (!f) { perror(path.c_str()); exit(1); }
    fwrite(data, 4, count, f); fclose(f);
    fprintf(g_manifest, "%s\n  \"%s\": {\"dtype\": \"%s\", \"count\": %zu, \"cols\": %zu, \"crc32\": %u}", g_first ? "" : ",", name, dtype, count, cols, crc32_buf(data, 4 * count));
    g_first = false;
}
static void note(const char* name, const char* json_value) {
    fprintf(g_manifest, "%s\n  \"%s\": %s", g_first ? "" : ",", name, json_value); g_first = false;
}

// deterministic parameter stream for the ctor grids (values only feed the reference ctors)
static uint64_t sm_state = 0x9E3779B97F4A7C15ull;
static float urand(float lo, float hi) {
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    float u = (float)(z >> 40) * (1.0f / 16777216.0f);
    return lo + (hi - lo) * u;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: refgen <reference_root> <out_dir>\n"); return 2; }
    std::string root = argv[1]; g_out = argv[2];
    g_manifest = fopen((g_out + "/manifest.json").c_str(), "w");
    fprintf(g_manifest, "{");

    // ---- (1) Splat4D ctor-2 grid: in = {quat wxyz, scale3, lifetime, fade, dir3} (12), out = cov16
    {
        const int N = 64; std::vector<float> in, out;
        for (int i = 0; i < N; ++i) {
            glm::quat q(urand(-1, 1), urand(-1, 1), urand(-1, 1), urand(-1, 1));     // w,x,y,z
            if (i % 2 == 0) q = glm::normalize(q);
            glm::vec3 sc(urand(0.2f, 5.0f), urand(0.2f, 5.0f), urand(0.2f, 5.0f));
            float life = urand(0.3f, 3.0f);
            float fade = (i % 3 == 0) ? 0.5f : urand(0.05f, 0.95f);
            glm::vec3 dir(urand(-5, 5), urand(-5, 5), urand(-5, 5));
            Splat4D s(glm::vec4(1, 2, 3, 4), q, sc, life, fade, dir, glm::vec4(1));
            glm::mat4 g = s.GetGeoInfo();
            float rowin[12] = { q.w, q.x, q.y, q.z, sc.x, sc.y, sc.z, life, fade, dir.x, dir.y, dir.z };
            in.insert(in.end(), rowin, rowin + 12);
            out.insert(out.end(), &g[0][0], &g[0][0] + 16);
        }
        dump("splat4d_ctor2_in", "f32", in.data(), in.size(), 12);
        dump("splat4d_ctor2_cov", "f32", out.data(), out.size(), 16);
    }
    // ---- (2) Splat4D ctor-1 (two quaternions): in = {q0 wxyz, q1 wxyz, scale4} (12), out = cov16
    {
        const int N = 32; std::vector<float> in, out;
        for (int i = 0; i < N; ++i) {
            glm::quat q0(urand(-1, 1), urand(-1, 1), urand(-1, 1), urand(-1, 1));
            glm::quat q1(urand(-1, 1), urand(-1, 1), urand(-1, 1), urand(-1, 1));
            glm::vec4 sc(urand(0.2f, 4.0f), urand(0.2f, 4.0f), urand(0.2f, 4.0f), urand(0.2f, 4.0f));
            Splat4D s(glm::vec4(0), q0, q1, sc, glm::vec4(1));
            glm::mat4 g = s.GetGeoInfo();
            float rowin[12] = { q0.w, q0.x, q0.y, q0.z, q1.w, q1.x, q1.y, q1.z, sc.x, sc.y, sc.z, sc.w };
            in.insert(in.end(), rowin, rowin + 12);
            out.insert(out.end(), &g[0][0], &g[0][0] + 16);
        }
        dump("splat4d_ctor1_in", "f32", in.data(), in.size(), 12);
        dump("splat4d_ctor1_cov", "f32", out.data(), out.size(), 16);
    }
    // ---- (3) Splat3D ctor: in = {quat wxyz, scale3} (7), out = cov9 ; and quatLookAt: in = {n3}, out = quat wxyz
    {
        const int N = 64; std::vector<float> in, out, nin, qout;
        for (int i = 0; i < N; ++i) {
            glm::quat q(urand(-1, 1), urand(-1, 1), urand(-1, 1), urand(-1, 1));
            if (i % 2 == 0) q = glm::normalize(q);
            glm::vec3 sc(urand(0.2f, 5.0f), urand(0.2f, 5.0f), urand(0.2f, 5.0f));
            Splat3D s(glm::vec4(0, 0, 0, 1), q, sc, glm::vec4(1));
            glm::mat3 g = s.GetGeoInfo();
            float rowin[7] = { q.w, q.x, q.y, q.z, sc.x, sc.y, sc.z };
            in.insert(in.end(), rowin, rowin + 7);
            out.insert(out.end(), &g[0][0], &g[0][0] + 9);
            glm::vec3 n(urand(-1, 1), urand(-1, 1), urand(-1, 1));
            if (i == 0) n = glm::vec3(0, 1, 0);              // degenerate: parallel to up
            if (i == 1) n = glm::vec3(0, -1, 0);
            glm::quat ql = glm::normalize(glm::quatLookAt(glm::normalize(n), glm::vec3(0, 1, 0)));   // Scenes.h:268
            nin.insert(nin.end(), { n.x, n.y, n.z });
            qout.insert(qout.end(), { ql.w, ql.x, ql.y, ql.z });
        }
        dump("splat3d_ctor_in", "f32", in.data(), in.size(), 7);
        dump("splat3d_ctor_cov", "f32", out.data(), out.size(), 9);
        dump("quatlookat_in", "f32", nin.data(), nin.size(), 3);
        dump("quatlookat_q", "f32", qout.data(), qout.size(), 4);
    }
    // ---- (4) Camera matrices: in = {w,h,pos3,ori3,far} (9), out = view16 + proj16
    {
        struct Cam { int w, h; glm::vec3 p, o; float far_; };
        Cam cams[] = {
            { 1920, 1080, { 60, 90, 90 }, { 0, -1, -1 }, 5000.0f },                                         // LinearMotion  Scenes.h:228-229
            { 1920, 1080, { 551.58f, 350.43f, -184.33f }, { -0.774978f, -0.570354f, 0.272222f }, 5000.0f }, // README screenshot camera
            { 3840, 2160, { 0, 60, 60 }, { 0, -1, -1 }, 5000.0f },                                          // NonLinearMotion Scenes.h:493-494
            { 800, 800, { 0, 0, 10 }, { 0, 0, -1 }, 256.0f },                                               // Camera.h defaults
        };
        std::vector<float> in, out;
        for (auto& c : cams) {
            Camera cam(c.w, c.h, c.p, c.o);
            cam.SetFar(c.far_);                                                                           // Application.cpp:126
            glm::mat4 v = cam.GetViewMatrix(), p = cam.GetProjMatrix();
            float rowin[9] = { (float)c.w, (float)c.h, c.p.x, c.p.y, c.p.z, c.o.x, c.o.y, c.o.z, c.far_ };
            in.insert(in.end(), rowin, rowin + 9);
            out.insert(out.end(), &v[0][0], &v[0][0] + 16);
            out.insert(out.end(), &p[0][0], &p[0][0] + 16);
        }
        dump("camera_in", "f32", in.data(), in.size(), 9);
        dump("camera_viewproj", "f32", out.data(), out.size(), 32);
    }
    // ---- (5) VData::parse(teapot): all records as 6 floats (pos, normal)
    std::vector<glm::mat3> model = VData::parse(root + "/Objects/teapot.vdata");
    {
        std::vector<float> flat;
        for (auto& m : model) flat.insert(flat.end(), { m[0][0], m[0][1], m[0][2], m[1][0], m[1][1], m[1][2] });
        dump("teapot_vdata", "f32", flat.data(), flat.size(), 6);
        char buf[64]; snprintf(buf, sizeof buf, "%zu", model.size()); note("teapot_vertices", buf);
    }
    // ---- (6) LinearMotion SSBO (Scenes.h:258-279) with the class defaults (Scenes.h:186-201)
    std::vector<Scenes::SplatData> lin;
    {
        const int steps = 50; const float mult = 1.0f, oscale = 5.0f, sx = 4.0f, sy = 4.0f, sz = 1.0f, life = 1.0f, fade = 0.5f, speed = 1.0f;
        ModelEdges medge = Scenes::GetModelExtrema(model);
        for (int dt = 0; dt < steps; ++dt)
            for (int i = 0; i < (int)model.size(); ++i) {
                glm::vec3 pos = model[i][0];
                glm::vec3 dir{ 1.0, 0.0, 0.0 };
                glm::vec3 timeOffset = dir * float(dt * mult);
                Splat4D s4d{ glm::vec4{ (oscale * pos) + timeOffset, float(dt) },
                             glm::normalize(glm::quatLookAt(glm::normalize(model[i][1]), glm::vec3(0, 1, 0))),
                             glm::vec3{ sx, sy, sz }, life, fade, glm::normalize(dir) * speed,
                             Scenes::GetColor(pos, medge, model[i][1]) };
                lin.push_back({ s4d.GetPosititon(), s4d.GetColor(), s4d.GetGeoInfo() });
            }
        static_assert(sizeof(Scenes::SplatData) == 96, "SplatData must be 96 bytes");
        dump("linear_first1000", "f32", lin.data(), 1000 * 24, 24);
        dump("linear_block25_first200", "f32", lin.data() + 25 * model.size(), 200 * 24, 24);
        char buf[128]; snprintf(buf, sizeof buf, "{\"records\": %zu, \"crc32\": %u}", lin.size(), crc32_buf(lin.data(), lin.size() * 96)); note("linear_full", buf);
    }
    // ---- (7) key/value uploads of the key loop (Scenes.h:314-319) for t in {0, 12.5, 49}, camera (60,90,90)
    {
        glm::vec3 campos{ 60, 90, 90 };
        float ts[3] = { 0.0f, 12.5f, 49.0f };
        for (int k = 0; k < 3; ++k) {
            std::vector<float> keys(lin.size());
            for (int i = 0; i < (int)lin.size(); ++i) {
                glm::vec4 tmp = lin[i].GetMeanInTime(ts[k]) - glm::vec4(campos, 1);
                keys[i] = 1.0f / sqrtf(tmp.x * tmp.x + tmp.y * tmp.y + tmp.z * tmp.z);
            }
            char nm[64]; snprintf(nm, sizeof nm, "linear_keys_t%d_first4000", k);
            dump(nm, "f32", keys.data(), 4000, 1);
            snprintf(nm, sizeof nm, "linear_keys_t%d_block25_first200", k);
            dump(nm, "f32", keys.data() + 25 * model.size(), 200, 1);
            char buf[128]; snprintf(buf, sizeof buf, "{\"t\": %g, \"crc32\": %u}", ts[k], crc32_buf(keys.data(), keys.size() * 4));
            snprintf(nm, sizeof nm, "linear_keys_t%d_full", k); note(nm, buf);
        }
    }
    // ---- (8) NonLinearMotion SSBO (Scenes.h:517-545) with the class defaults (Scenes.h:451-467)
    {
        const int steps = 92; const float oscale = 5.0f, sx = 4.0f, sy = 4.0f, sz = 1.0f, life = 1.0f, fade = 0.5f, speed = 20.0f, radius = 20.0f, amul = 4.0f;
        std::vector<Scenes::SplatData> nl;
        ModelEdges medge = Scenes::GetModelExtrema(model);
        for (int dt = 0; dt < steps; ++dt)
            for (int i = 0; i < (int)model.size(); ++i) {
                glm::vec3 pos = model[i][0];
                glm::vec4 forward{ 1.0, 0.0, 0.0, 0.0 };
                glm::vec3 timeOffset = glm::vec3{ glm::rotate(forward, glm::radians(float(dt * amul)), { 0.0, 1.0, 0.0 }) };
                glm::vec3 timeOffset_next = glm::vec3{ glm::rotate(forward, glm::radians(float((dt + 1) * amul)), { 0.0, 1.0, 0.0 }) };
                Splat4D s4d{ glm::vec4{ (oscale * pos) + (timeOffset * radius), float(dt) },
                             glm::normalize(glm::quatLookAt(glm::normalize(model[i][1]), glm::vec3(0, 1, 0))),
                             glm::vec3{ sx, sy, sz }, life, fade, (timeOffset_next - timeOffset) * speed,
                             Scenes::GetColor(pos, medge, model[i][1]) };
                nl.push_back({ s4d.GetPosititon(), s4d.GetColor(), s4d.GetGeoInfo() });
            }
        dump("nonlinear_first500", "f32", nl.data(), 500 * 24, 24);
        dump("nonlinear_block45_first200", "f32", nl.data() + 45 * model.size(), 200 * 24, 24);
        char buf[128]; snprintf(buf, sizeof buf, "{\"records\": %zu, \"crc32\": %u}", nl.size(), crc32_buf(nl.data(), nl.size() * 96)); note("nonlinear_full", buf);
    }
    // ---- (9) RotationMotion SSBO (Scenes.h:775-803) with the class defaults (Scenes.h:711-727)
    {
        const int steps = 92; const float oscale = 5.0f, sx = 4.0f, sy = 4.0f, sz = 1.0f, life = 0.6f, fade = 0.5f, speed = 5.0f, amul = 4.0f;
        std::vector<Scenes::SplatData> sd;
        ModelEdges medge = Scenes::GetModelExtrema(model);
        for (int dt = 0; dt < steps; ++dt)
            for (int i = 0; i < (int)model.size(); ++i) {
                glm::vec4 pos{ model[i][0], 0.0 };
                glm::vec3 timeOffset = glm::vec3{ glm::rotate(pos, glm::radians(float(dt * amul)), { 0.0, 1.0, 0.0 }) };
                glm::vec3 timeOffset_next = glm::vec3{ glm::rotate(pos, glm::radians(float((dt + 1) * amul)), { 0.0, 1.0, 0.0 }) };
                glm::vec3 norm = glm::vec3{ glm::rotate(glm::vec4{ model[i][1], 0 }, glm::radians(float(dt * amul)), { 0.0, 1.0, 0.0 }) };
                Splat4D s4d{ glm::vec4{ (oscale * timeOffset), float(dt) },
                             glm::normalize(glm::quatLookAt(glm::normalize(norm), glm::vec3(0, 1, 0))),
                             glm::vec3{ sx, sy, sz }, life, fade, (timeOffset_next - timeOffset) * speed,
                             Scenes::GetColor(pos, medge, model[i][1]) };
                sd.push_back({ s4d.GetPosititon(), s4d.GetColor(), s4d.GetGeoInfo() });
            }
        dump("rotation_first300", "f32", sd.data(), 300 * 24, 24);
        dump("rotation_block45_first200", "f32", sd.data() + 45 * model.size(), 200 * 24, 24);
        char buf[128]; snprintf(buf, sizeof buf, "{\"records\": %zu, \"crc32\": %u}", sd.size(), crc32_buf(sd.data(), sd.size() * 96)); note("rotation_full", buf);
    }
    // ---- (10) CombinedMotion SSBO (Scenes.h:1035-1068) with the class defaults (Scenes.h:959-976)
    {
        const int steps = 65; const float oscale = 5.0f, sx = 4.0f, sy = 4.0f, sz = 0.0f, life = 1.0f, fade = 0.5f, speed = 1.0f, amul = 8.0f, lmul = 8.0f, amp = 1.0f, freq = 0.15f;
        std::vector<Scenes::SplatData> sd;
        ModelEdges medge = Scenes::GetModelExtrema(model);
        for (int dt = 0; dt < steps; ++dt)
            for (int i = 0; i < (int)model.size(); ++i) {
                glm::vec3 md = oscale * model[i][0];
                glm::vec3 nd = model[i][1];
                glm::vec3 pos = glm::vec3{ glm::rotate(glm::vec4{ md, 0 }, glm::radians(float(dt * amul)), { 0.0, 1.0, 0.0 }) } + lmul * glm::vec3{ freq * float(dt), amp * sinf(freq * float(dt)), 0.0f };
                glm::vec3 pos_next = glm::vec3{ glm::rotate(glm::vec4{ md, 0 }, glm::radians(float((dt + 1) * amul)), { 0.0, 1.0, 0.0 }) } + lmul * glm::vec3{ freq * float(dt + 1), amp * sinf(freq * float(dt + 1)), 0.0f };
                glm::vec3 norm = glm::vec3{ glm::rotate(glm::vec4{ nd, 0 }, glm::radians(float(dt * amul)), { 0.0, 1.0, 0.0 }) };
                Splat4D s4d{ glm::vec4{ pos, float(dt) },
                             glm::normalize(glm::quatLookAt(glm::normalize(norm), glm::vec3(0, 1, 0))),
                             glm::vec3{ sx, sy, sz }, life, fade, (pos_next - pos) * speed,
                             Scenes::GetColor(model[i][0], medge, model[i][1]) };
                sd.push_back({ s4d.GetPosititon(), s4d.GetColor(), s4d.GetGeoInfo() });
            }
        dump("combined_first300", "f32", sd.data(), 300 * 24, 24);
        dump("combined_block33_first200", "f32", sd.data() + 33 * model.size(), 200 * 24, 24);
        char buf[128]; snprintf(buf, sizeof buf, "{\"records\": %zu, \"crc32\": %u}", sd.size(), crc32_buf(sd.data(), sd.size() * 96)); note("combined_full", buf);
    }
    // ---- (11) BrokenMotion SSBO (Scenes.h:1965-1989) with the class defaults (Scenes.h:1899-1912)
    {
        const int steps = 92; const float oscale = 5.0f, sx = 4.0f, sy = 4.0f, sz = 1.0f, life = 1.0f, fade = 0.5f, speed = 1.0f;
        std::vector<Scenes::SplatData> sd;
        ModelEdges medge = Scenes::GetModelExtrema(model);
        for (int dt = 0; dt < steps; ++dt) {
            glm::vec3 posdt{ 1.0f + dt, fmod((1.0f + dt), 20.0f), 0.0f };
            glm::vec3 posdtn{ 1.0f + (dt + 1.0f), fmod((1.0f + (dt + 1.0f)), 20.0f), 0.0f };
            for (int i = 0; i < (int)model.size(); ++i) {
                glm::vec3 pos = model[i][0];
                Splat4D s4d{ glm::vec4{ (oscale * pos) + (posdt), float(dt) },
                             glm::normalize(glm::quatLookAt(glm::normalize(model[i][1]), glm::vec3(0, 1, 0))),
                             glm::vec3{ sx, sy, sz }, life, fade, (posdtn - posdt) * speed,
                             Scenes::GetColor(pos, medge, model[i][1]) };
                sd.push_back({ s4d.GetPosititon(), s4d.GetColor(), s4d.GetGeoInfo() });
            }
        }
        dump("broken_first300", "f32", sd.data(), 300 * 24, 24);
        dump("broken_block19_first200", "f32", sd.data() + 19 * model.size(), 200 * 24, 24);
        char buf[128]; snprintf(buf, sizeof buf, "{\"records\": %zu, \"crc32\": %u}", sd.size(), crc32_buf(sd.data(), sd.size() * 96)); note("broken_full", buf);
    }
    // ---- (12) SquareMotion SSBO (Scenes.h:2216-2259) with the class defaults (Scenes.h:2151-2165)
    {
        const int steps = 92; const float oscale = 5.0f, sx = 4.0f, sy = 4.0f, sz = 1.0f, life = 1.0f, fade = 0.5f, speed = 1.0f, size = 40.0f;
        std::vector<Scenes::SplatData> sd;
        ModelEdges medge = Scenes::GetModelExtrema(model);
        int side = 0;
        int stepsPerSide = steps / 4;
        float deltaStep = size / float(stepsPerSide);
        glm::vec3 posdt{ size / 2.0f, 0.0, size / 2.0f };
        glm::vec3 posdtn = glm::vec3{ size / 2.0f, 0.0, size / 2.0f } + (deltaStep * glm::vec3{ -1.0f, 0.0f, 0.0f });
        for (int dt = 0; dt < steps; ++dt) {
            glm::vec3 dir{ 0 };
            if (dt > 0 && dt % stepsPerSide == 0) side += 1;
            if (side == 0) dir = { -1.0f, 0.0f, 0.0f };
            if (side == 1) dir = { 0.0f, 0.0f, -1.0f };
            if (side == 2) dir = { 1.0f, 0.0f, 0.0f };
            if (side == 3) dir = { 0.0f, 0.0f, 1.0f };
            posdt = posdt + (deltaStep * dir);
            int s2 = side;
            if ((dt + 1) > 0 && (dt + 1) % stepsPerSide == 0) s2 += 1;
            if (s2 == 0) dir = { -1.0f, 0.0f, 0.0f };
            if (s2 == 1) dir = { 0.0f, 0.0f, -1.0f };
            if (s2 == 2) dir = { 1.0f, 0.0f, 0.0f };
            if (s2 == 3) dir = { 0.0f, 0.0f, 1.0f };
            posdtn = posdtn + (deltaStep * dir);
            for (int i = 0; i < (int)model.size(); ++i) {
                glm::vec3 pos = model[i][0];
                Splat4D s4d{ glm::vec4{ (oscale * pos) + (posdt), float(dt) },
                             glm::normalize(glm::quatLookAt(glm::normalize(model[i][1]), glm::vec3(0, 1, 0))),
                             glm::vec3{ sx, sy, sz }, life, fade, (posdtn - posdt) * speed,
                             Scenes::GetColor(pos, medge, model[i][1]) };
                sd.push_back({ s4d.GetPosititon(), s4d.GetColor(), s4d.GetGeoInfo() });
            }
        }
        dump("square_first300", "f32", sd.data(), 300 * 24, 24);
        dump("square_block23_first200", "f32", sd.data() + 23 * model.size(), 200 * 24, 24);
        dump("square_block91_first200", "f32", sd.data() + 91 * model.size(), 200 * 24, 24);
        char buf[128]; snprintf(buf, sizeof buf, "{\"records\": %zu, \"crc32\": %u}", sd.size(), crc32_buf(sd.data(), sd.size() * 96)); note("square_full", buf);
    }
    // ---- (13) VData::parse_splat_data (.sd, VDataParser.h:60-123) + the ObjectDisplay record loop (Scenes.h:2483-2491, object scale 2.5 here).
    // The reference ships no .sd file: a small synthetic one is written next to the fixtures (data, 23 numbers per splat: position,
    // colour, 4x4 covariance column by column) and read back through the reference's parser.
    {
        const std::string sd_path = g_out + "/synthetic.sd";
        FILE* f = fopen(sd_path.c_str(), "w");
        ModelEdges medge = Scenes::GetModelExtrema(model);
        const int n = 48;
        for (int i = 0; i < n; ++i) {
            const int v = (i * 71) % (int)model.size();
            glm::vec3 pos = model[v][0];
            Splat4D s4d{ glm::vec4{ 5.0f * pos, float(i % 7) }, glm::normalize(glm::quatLookAt(glm::normalize(model[v][1]), glm::vec3(0, 1, 0))),
                         glm::vec3{ 4.0f, 2.0f + 0.125f * float(i % 5), 1.0f }, 1.0f, 0.5f, glm::vec3{ 1.0f, 0.25f * float(i % 3), 0.0f },
                         Scenes::GetColor(pos, medge, model[v][1]) };
            glm::mat4 c = s4d.GetGeoInfo(); glm::vec4 col = s4d.GetColor();
            fprintf(f, "%.9g %.9g %.9g  %.9g %.9g %.9g %.9g ", pos.x, pos.y, pos.z, col.x, col.y, col.z, col.w);
            for (int cc = 0; cc < 4; ++cc) for (int r = 0; r < 4; ++r) fprintf(f, " %.9g", c[cc][r]);
            fprintf(f, i % 2 ? "\n" : "\n\n");        // blank lines are skipped by the parser
        }
        fclose(f);
        std::vector<VData::VSplatData> sd = VData::parse_splat_data(sd_path);
        const float oscale = 2.5f;
        std::vector<Scenes::SplatData> recs;
        for (auto& m : sd) recs.push_back({ glm::vec4{ oscale * m.pos, 0.0 }, m.color, m.cov });
        dump("synthetic_sd_records", "f32", recs.data(), recs.size() * 24, 24);
    }
    // ---- (14) Splat3D::GetSplatMesh (Splat.h:433-447): four 72-byte vertices {corner, position, colour, Sigma3} per splat
    {
        std::vector<float> in, out;
        for (int i = 0; i < 24; ++i) {
            glm::vec4 pos{ 0.5f * float(i) - 3.0f, 0.25f * float(i % 5), -1.5f + 0.125f * float(i), 1.0f };
            glm::quat q = glm::normalize(glm::quat(1.0f + 0.1f * float(i), 0.3f * float(i % 4) - 0.2f, 0.7f - 0.05f * float(i), 0.11f * float(i % 7)));
            glm::vec3 sc{ 0.5f + 0.25f * float(i % 3), 1.0f + 0.125f * float(i % 5), 0.25f + 0.0625f * float(i) };
            glm::vec4 col{ 0.04f * float(i), 1.0f - 0.03f * float(i), 0.5f, 0.25f + 0.03f * float(i) };
            std::vector<Geometry::Splat3DVertex> v = Splat3D::GetSplatMesh(pos, q, sc, col);
            static_assert(sizeof(Geometry::Splat3DVertex) == 72, "Splat3DVertex must be 72 bytes");
            const float row[15] = { pos.x, pos.y, pos.z, pos.w, q.w, q.x, q.y, q.z, sc.x, sc.y, sc.z, col.x, col.y, col.z, col.w };
            in.insert(in.end(), row, row + 15);
            const float* f = reinterpret_cast<const float*>(v.data());
            out.insert(out.end(), f, f + 72);
        }
        dump("splat3d_mesh_in", "f32", in.data(), in.size(), 15);
        dump("splat3d_mesh_verts", "f32", out.data(), out.size(), 72);
    }
    // ---- (15) Splat2D (Splat.h:551-582: CalcAndSetSigma) and the Gaussians2D record expression (Scenes.h:1490-1496) for given
    // angle / scales / position / colour (the scene draws them from RANDOM)
    {
        std::vector<float> in, out, in2, out2;
        for (int i = 0; i < 32; ++i) {
            glm::vec2 v0{ cosf(0.37f * float(i)) * (1.0f + 0.1f * float(i % 3)), sinf(0.37f * float(i)) * 1.5f };
            const float l0 = 0.25f + 0.3f * float(i % 6), l1 = 4.0f - 0.11f * float(i);
            Splat2D s2{ glm::vec3{ 0.0f, 0.0f, 0.0f }, v0, l0, l1, glm::vec4{ 1.0f } };
            // mSigma is private and only leaves the class through Shader::SetUniformMat2f: the four statements of CalcAndSetSigma
            // (Splat.h:576-582) are restated on the members the class does expose (glm does the arithmetic)
            glm::mat2 S_(s2.GetLambda0(), 0.0f, 0.0f, s2.GetLambda1());
            glm::mat2 R_(s2.GetVector(), glm::normalize(glm::vec2(v0.y, -v0.x)));
            glm::mat2 sg = glm::inverse(R_ * S_ * glm::transpose(S_) * glm::transpose(R_));
            const float row[4] = { v0.x, v0.y, l0, l1 };
            in.insert(in.end(), row, row + 4);
            out.insert(out.end(), &sg[0][0], &sg[0][0] + 4);
            const float a = glm::radians(360.0f * (float(i) / 32.0f + 0.013f));
            const float s0 = 1.0f + (5.0f * (0.03f * float(i))), s1 = 1.0f + (5.0f * (1.0f - 0.029f * float(i)));
            const float px = 10.0f * (-0.5f + 0.031f * float(i)), py = 10.0f * (-0.5f + (1.0f - 0.03f * float(i)));
            glm::mat2 R{ cosf(a), -sinf(a), sinf(a), cos(a) };
            glm::mat2 S{ s0, 0.0f, 0.0f, s1 };
            struct Splat2DData { glm::vec4 position; glm::vec4 color; glm::mat2 geoinfo; } d{ { px, py, 0, 0 }, { 0.1f * float(i % 10), 0.5f, 1.0f - 0.02f * float(i), 1.0 }, R * S * S * glm::transpose(R) };
            static_assert(sizeof(d) == 48, "Splat2DData must be 48 bytes");
            const float row2[8] = { a, s0, s1, px, py, d.color.x, d.color.y, d.color.z };
            in2.insert(in2.end(), row2, row2 + 8);
            const float* f = reinterpret_cast<const float*>(&d);
            out2.insert(out2.end(), f, f + 12);
        }
        dump("splat2d_sigma_in", "f32", in.data(), in.size(), 4);
        dump("splat2d_sigma_inv", "f32", out.data(), out.size(), 4);
        dump("gaussians2d_in", "f32", in2.data(), in2.size(), 8);
        dump("gaussians2d_records", "f32", out2.data(), out2.size(), 12);
    }
    // (8) Camera input model.  Camera::HandleInput / HandleCamRotation (Camera.cpp:116-207) call GLFW, whose library is absent here, so the
    //     key/cursor plumbing is restated below (the statements of Camera.cpp in their order, keys and cursor taken from a table); every
    //     arithmetic step is the reference's: glm::rotate(vec3, angle, axis), glm::normalize, glm::cross, glm::radians on the Camera's
    //     own public members, and GetViewport / GetFocal / SetIsViewFixedOnPoint run from the reference's Camera.cpp itself.
    {
        const int W = 800, H = 800;
        Camera cam(W, H, glm::vec3(60, 90, 90), glm::vec3(0, -1, -1));
        const float mSensitivity = 100.0f, mSpeed = 0.5f, mFastSpeed = 2.0f;      // Camera.h:78-80
        bool capture = false;
        enum { KW = 1, KS = 2, KA = 4, KD = 8, KE = 16, KQ = 32, KSPACE = 64, KCTRL = 128, KSHIFT = 256, KC = 512, KESC = 1024 };
        std::vector<float> in, out;
        for (int step = 0; step < 96; ++step) {
            unsigned keys = 0;
            const float r = urand(0.0f, 1.0f);
            if (step == 3) keys |= KC;                                    // capture the mouse early ...
            if (step == 70) keys |= KESC;                                 // ... release it late
            if (step == 80) keys |= KC;
            for (int b = 0; b < 9; ++b) if (urand(0.0f, 1.0f) < 0.3f) keys |= 1u << b;
            (void)r;
            const double mx = (double)W / 2.0 + (double)(int)urand(-40.0f, 40.0f), my = (double)H / 2.0 + (double)(int)urand(-30.0f, 30.0f);
            // --- Camera.cpp:116-183 ---
            float currentSpeed = mSpeed;
            if (keys & KSHIFT) currentSpeed = mFastSpeed;
            if (keys & KW) cam.position += cam.orientation * currentSpeed;
            if (keys & KS) cam.position += cam.orientation * -currentSpeed;
            if (keys & KA) cam.position += -currentSpeed * glm::normalize(glm::cross(cam.orientation, cam.up));
            if (keys & KD) cam.position += currentSpeed * glm::normalize(glm::cross(cam.orientation, cam.up));
            if (keys & KE) cam.up = glm::rotate(cam.up, glm::radians(1.0f), cam.orientation);
            if (keys & KQ) cam.up = glm::rotate(cam.up, glm::radians(-1.0f), cam.orientation);
            if (keys & KSPACE) cam.position += currentSpeed * cam.up;
            if (keys & KCTRL) cam.position += -currentSpeed * cam.up;
            double mouseX = mx, mouseY = my;
            if ((keys & KC) && !capture) { mouseX = (double)W / 2.0; mouseY = (double)H / 2.0; capture = true; }      // glfwSetCursorPos(centre)
            if (keys & KESC) capture = false;
            if (capture) {                                                 // --- Camera.cpp:191-207 ---
                double rotX = -mSensitivity * (mouseY - int((double)H / 2.0)) / (double)(H);
                double rotY = -mSensitivity * (mouseX - int((double)W / 2.0)) / (double)(W);
                cam.orientation = glm::rotate(cam.orientation, (float)glm::radians(rotX), glm::normalize(glm::cross(cam.orientation, cam.up)));
                glm::vec3 side = glm::normalize(glm::cross(cam.up, cam.orientation));
                cam.up = glm::normalize(glm::cross(cam.orientation, side));
                cam.orientation = glm::rotate(cam.orientation, (float)glm::radians(rotY), cam.up);
            }
            const float irow[3] = { (float)keys, (float)mx, (float)my };
            in.insert(in.end(), irow, irow + 3);
            const float orow[10] = { cam.position.x, cam.position.y, cam.position.z, cam.orientation.x, cam.orientation.y, cam.orientation.z, cam.up.x, cam.up.y, cam.up.z, capture ? 1.0f : 0.0f };
            out.insert(out.end(), orow, orow + 10);
        }
        dump("camera_walk_in", "f32", in.data(), in.size(), 3);
        dump("camera_walk_out", "f32", out.data(), out.size(), 10);
        // the reference's own Camera.cpp for the three helpers that need no window
        cam.SetIsViewFixedOnPoint(true, glm::vec4(1.0f, 2.0f, 3.0f, 1.0f));
        const glm::vec2 vpn = cam.GetViewport(), foc = cam.GetFocal();
        const float misc[10] = { cam.orientation.x, cam.orientation.y, cam.orientation.z, cam.up.x, cam.up.y, cam.up.z, vpn.x, vpn.y, foc.x, foc.y };
        dump("camera_misc", "f32", misc, 10, 10);
    }
    fprintf(g_manifest, "\n}\n"); fclose(g_manifest);
    printf("refgen: fixtures written to %s\n", g_out.c_str());
    return 0;
}
