// refdraw_main.cpp — golden-vector generator for the reference's CPU restatement of its own vertex stage: Splat4D::Draw
// (4DSplatRendering/Splat.h:163-247) and Splat3D::Draw (Splat.h:355-431), the only executable code in the reference that overlaps the
// GLSL half of the path (SURVEY.md §8(c) family 8).  Test infrastructure.
//
// Built like refscene (oracle/Makefile target `refdraw` -> oracle/_ref/refdraw): the reference's Splat.h, Scenes.h, Utils.cpp,
// VDataParser.h compiled unmodified where they lie, GLM from the reference tree, against the PRODUCT's drop-in headers (host/shadow/).
// What that means for the vectors it writes — stated so that nobody reads more into them than they hold:
//   * every arithmetic statement between the splat's members and the SetUniform* calls is the reference's (Splat.h) and GLM's:
//     time conditioning, view/projection, Jacobian, T = W J, cov3 = T^t Sigma T, eigenvalues, eigenvectors, the cull, p(t);
//   * the SINK is this build's own header: the shadow Shader keeps the last value of every uniform (gs4d_compat.h, LastUniform) and the
//     harness reads uScreenPos / uScale / uVec1 / uVec2 / uSigma / uColor back from it;
//   * the camera matrices come from the shadow Camera (gs4d_host_look_at / gs4d_host_perspective / gs4d_host_camera_viewport), which is
//     pinned bit-for-bit to the reference's Camera.cpp + GLM by tests/golden/camera_* (tests/test_host_math.py);
//   * the per-splat parameter loops are restated from NonLinearMotion::init (Scenes.h:517-545) as in refgen.cpp, and checked here against
//     the records refgen wrote (tests/golden/nonlinear_*.bin) — bit for bit, or this program fails.
// So this is an INDEPENDENT CROSS-CHECK of V1-V5 of SURVEY.md §8(a) at float precision, not "the reference's renderer run here".
// No GPU and no gs4d context is needed: nothing below creates one (Splat3D::Draw ends with three Renderer::DrawLine calls, which need a
// context; the harness catches the exception they raise — the uniforms have been set by then).
//
//   refdraw <reference_root> <golden_dir>
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
#include "BSPTree.h"
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
static void dump(const char* name, const void* data, size_t count, size_t cols) {
    const std::string path = g_out + "/" + name + ".bin";
    FILE* f = fopen(path.c_str(), "wb"); if (!f) { perror(path.c_str()); exit(1); }
    fwrite(data, 4, count, f); fclose(f);
    fprintf(g_manifest, "%s\n  \"%s\": {\"dtype\": \"f32\", \"count\": %zu, \"cols\": %zu, \"crc32\": %u}", g_first ? "" : ",", name, count, cols, crc32_buf(data, 4 * count));
    g_first = false;
}
static std::vector<float> slurp(const std::string& path) {
    std::vector<float> v; FILE* f = fopen(path.c_str(), "rb"); if (!f) { perror(path.c_str()); exit(1); }
    float buf[4096]; size_t k; while ((k = fread(buf, 4, 4096, f)) > 0) v.insert(v.end(), buf, buf + k);
    fclose(f); return v;
}

// One output row per Draw call: visible (1/0), uScreenPos.xy, uScale.xy, uVec1.xy, uVec2.xy, uSigma (4, column-major), uColor.rgba = 17 floats
constexpr int ROW = 17;
static void harvest(Shader& sh, std::vector<float>& out) {
    float row[ROW] = { 0 };
    float v[16];
    if (sh.LastUniform("uScreenPos", v, 2) == 2) {      // the cull (Splat.h:230-236, 411-415) returns before any uniform is set
        row[0] = 1.0f; row[1] = v[0]; row[2] = v[1];
        if (sh.LastUniform("uScale", v, 2) != 2) { fprintf(stderr, "refdraw: uScale missing\n"); exit(1); } row[3] = v[0]; row[4] = v[1];
        if (sh.LastUniform("uVec1", v, 2) != 2) { fprintf(stderr, "refdraw: uVec1 missing\n"); exit(1); } row[5] = v[0]; row[6] = v[1];
        if (sh.LastUniform("uVec2", v, 2) != 2) { fprintf(stderr, "refdraw: uVec2 missing\n"); exit(1); } row[7] = v[0]; row[8] = v[1];
        if (sh.LastUniform("uSigma", v, 4) != 4) { fprintf(stderr, "refdraw: uSigma missing\n"); exit(1); } for (int k = 0; k < 4; ++k) row[9 + k] = v[k];
        if (sh.LastUniform("uColor", v, 4) != 4) { fprintf(stderr, "refdraw: uColor missing\n"); exit(1); } for (int k = 0; k < 4; ++k) row[13 + k] = v[k];
    }
    out.insert(out.end(), row, row + ROW);
    sh.ForgetUniforms();
}

struct CamSpec { int w, h; glm::vec3 p, o; };

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: refdraw <reference_root> <golden_dir>\n"); return 2; }
    const std::string root = argv[1]; g_out = argv[2];
    g_manifest = fopen((g_out + "/manifest_draw.json").c_str(), "w");
    if (!g_manifest) { perror("manifest_draw.json"); return 1; }
    fprintf(g_manifest, "{");

    std::vector<glm::mat3> model = VData::parse(root + "/Objects/teapot.vdata");
    ModelEdges medge = Scenes::GetModelExtrema(model);
    // the two blocks of NonLinearMotion (Scenes.h:517-545, class defaults :451-467) refgen.cpp wrote as records: time step 0 (first 500) and 45 (first 200)
    const float oscale = 5.0f, sx = 4.0f, sy = 4.0f, sz = 1.0f, life = 1.0f, fade = 0.5f, speed = 20.0f, radius = 20.0f, amul = 4.0f;
    auto make4d = [&](int dt, int i) {
        glm::vec3 pos = model[i][0];
        glm::vec4 forward{ 1.0, 0.0, 0.0, 0.0 };
        glm::vec3 timeOffset = glm::vec3{ glm::rotate(forward, glm::radians(float(dt * amul)), { 0.0, 1.0, 0.0 }) };
        glm::vec3 timeOffset_next = glm::vec3{ glm::rotate(forward, glm::radians(float((dt + 1) * amul)), { 0.0, 1.0, 0.0 }) };
        return Splat4D{ glm::vec4{ (oscale * pos) + (timeOffset * radius), float(dt) },
                        glm::normalize(glm::quatLookAt(glm::normalize(model[i][1]), glm::vec3(0, 1, 0))),
                        glm::vec3{ sx, sy, sz }, life, fade, (timeOffset_next - timeOffset) * speed,
                        Scenes::GetColor(pos, medge, model[i][1]) };
    };
    struct Block { int dt, count; const char* fixture; float times[3]; };
    const Block blocks[2] = { { 0, 500, "nonlinear_first500", { 0.0f, 0.75f, 14.25f } }, { 45, 200, "nonlinear_block45_first200", { 41.75f, 44.5f, 45.0f } } };
    // cameras: NonLinearMotion's own (Scenes.h:493-494, far plane Application.cpp:126) and a close, oblique one that culls part of the model
    const CamSpec cams[2] = { { 1920, 1080, { 0, 60, 60 }, { 0.0f, -1.0f, -1.0f } }, { 1280, 720, { 9.0f, 9.0f, 8.0f }, { -0.45f, -0.2f, -1.0f } } };
    {
        std::vector<float> camrows;
        for (const CamSpec& cs : cams) {
            Camera cam(cs.w, cs.h, cs.p, cs.o); cam.SetFar(5000.0f);
            glm::mat4 v = cam.GetViewMatrix(), p = cam.GetProjMatrix();
            const float head[8] = { (float)cs.w, (float)cs.h, cs.p.x, cs.p.y, cs.p.z, cs.o.x, cs.o.y, cs.o.z };
            camrows.insert(camrows.end(), head, head + 8);
            camrows.insert(camrows.end(), &v[0][0], &v[0][0] + 16);
            camrows.insert(camrows.end(), &p[0][0], &p[0][0] + 16);
        }
        dump("splat_draw_cameras", camrows.data(), camrows.size(), 40);
    }
    Renderer renderer;                 // constructing it touches no context (two line programs are named, nothing is compiled)
    GLFWwindow window;
    for (int b = 0; b < 2; ++b) {
        const Block& B = blocks[b];
        std::vector<Splat4D> splats;
        std::vector<float> recs;
        for (int i = 0; i < B.count; ++i) {
            splats.push_back(make4d(B.dt, i));
            Splat4D& s = splats.back();
            const Scenes::SplatData d{ s.GetPosititon(), s.GetColor(), s.GetGeoInfo() };
            const float* f = reinterpret_cast<const float*>(&d);
            recs.insert(recs.end(), f, f + 24);
        }
        // the same records refgen wrote, bit for bit — the tests feed THOSE to the checker
        const std::vector<float> have = slurp(g_out + "/" + B.fixture + ".bin");
        if (have.size() != recs.size() || memcmp(have.data(), recs.data(), recs.size() * 4) != 0) { fprintf(stderr, "refdraw: %s.bin does not hold the records generated here\n", B.fixture); return 1; }
        for (int c = 0; c < 2; ++c) {
            Camera cam(cams[c].w, cams[c].h, cams[c].p, cams[c].o); cam.SetFar(5000.0f);
            for (int k = 0; k < 3; ++k) {
                std::vector<float> rows;
                Shader sh;             // no sources: no pipeline is selected, Bind() touches no context
                for (Splat4D& s : splats) { s.SetTime(B.times[k]); s.Draw(&window, renderer, sh, cam); harvest(sh, rows); }
                char nm[96]; snprintf(nm, sizeof nm, "splat_draw_4d_b%d_cam%d_t%d", B.dt, c, k);
                dump(nm, rows.data(), rows.size(), ROW);
            }
        }
        char buf[96]; snprintf(buf, sizeof buf, "[%g, %g, %g]", B.times[0], B.times[1], B.times[2]);
        char nm[64]; snprintf(nm, sizeof nm, "splat_draw_4d_b%d_times", B.dt);
        fprintf(g_manifest, ",\n  \"%s\": %s", nm, buf);
    }
    // static 3D splats on the teapot's first 500 vertices, parameterised as the 4D scenes parameterise theirs (position 5 x vertex, rotation
    // from the normal, scale (4, 4, 1), colour gradient) — Splat3D::Splat3D (Splat.h:334-344) + Splat3D::Draw (Splat.h:355-431)
    {
        const int N = 500;
        std::vector<float> in;
        std::vector<Splat3D> s3;
        for (int i = 0; i < N; ++i) {
            glm::vec3 pos = model[i][0];
            glm::vec4 p4{ oscale * pos, 1.0f };
            glm::quat q = glm::normalize(glm::quatLookAt(glm::normalize(model[i][1]), glm::vec3(0, 1, 0)));
            glm::vec3 sc{ sx, sy, sz };
            glm::vec4 col = Scenes::GetColor(pos, medge, model[i][1]);
            col.w = 0.35f + 0.0013f * float(i);                         // the gradient's alpha is 1 everywhere: make the column carry information
            s3.emplace_back(p4, q, sc, col);
            glm::mat3 g = s3.back().GetGeoInfo();
            in.insert(in.end(), { p4.x, p4.y, p4.z });
            in.insert(in.end(), { col.x, col.y, col.z, col.w });
            in.insert(in.end(), &g[0][0], &g[0][0] + 9);
        }
        dump("splat_draw_3d_in", in.data(), in.size(), 16);
        for (int c = 0; c < 2; ++c) {
            Camera cam(cams[c].w, cams[c].h, cams[c].p, cams[c].o); cam.SetFar(5000.0f);
            std::vector<float> rows;
            Shader sh;
            int threw = 0;
            for (Splat3D& s : s3) {
                try { s.Draw(renderer, sh, cam); } catch (const std::runtime_error&) { ++threw; }      // Renderer::DrawLine without a context (Splat.h:427-429), after the uniforms
                harvest(sh, rows);
            }
            char nm[64]; snprintf(nm, sizeof nm, "splat_draw_3d_cam%d", c);
            dump(nm, rows.data(), rows.size(), ROW);
            fprintf(stderr, "refdraw: 3D camera %d: %d of %d draws reached the helper axes\n", c, threw, N);
        }
    }
    fprintf(g_manifest, "\n}\n"); fclose(g_manifest);
    printf("refdraw: fixtures written to %s\n", g_out.c_str());
    return 0;
}
