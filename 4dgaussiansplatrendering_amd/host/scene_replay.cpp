// scene_replay.cpp — the reference's LinearMotion scene (Scenes.h:174-340) driven through the mirrored host classes
// (gs4d_compat.h) exactly as Scenes.h writes it: init() builds the SSBO and the key/value buffers, Render() runs the key
// loop -> uploads -> sorter.sort -> uniforms -> binds -> Renderer::Draw.  Two things differ from the reference text and are
// marked NEW: the splat records come from gs4d_host_scene_linear (the reference builds them inline with GLM), and the CPU key
// loop can be replaced by the GPU key generation (--gpu-keys).  Writes the RGBA32F framebuffer to a file for the parity test.
//
//   scene_replay <teapot_vdata.bin> <out.rgba32f> <width> <height> <time> [--gpu-keys] [--no-sort]
//   scene_replay <teapot_vdata.bin> <out.rgba32f> <width> <height> <time> --frames N --png <prefix> [--keys WASDEQ...] [--dt step]
// The second form is the frame loop of Application.cpp:145-190 without a window: per iteration Clear, Camera::HandleInput (from a
// scripted key table), blend state, Update (time += dt), Render; frame k-1 is presented — packed to RGBA8 on the device and written as
// <prefix>_%04d.png — after frame k has been queued, as a swap chain does.  <out.rgba32f> receives the last frame.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "gs4d_compat.h"

struct SplatData { float pos[4]; float col[4]; float sig[16];                     // Scenes.h:22-37
    void GetMeanInTime(float ctime, float out[4]) { float c = ctime - pos[3]; out[0] = pos[0] + sig[12] * c; out[1] = pos[1] + sig[13] * c; out[2] = pos[2] + sig[14] * c; out[3] = 1; } };

class LinearMotion {
    Renderer& m_renderer; Camera& m_camera;
    std::vector<float> m_vModelData;               // 6 floats per vertex
    std::vector<SplatData> m_sdata;
    GLuint m_key_buf = 0, m_values_buf = 0;
    std::unique_ptr<ShareStorageBuffer> m_ssbo_splat_data;
    std::unique_ptr<radix_sort::sorter> m_sorter;
    std::vector<GLuint> m_key_buffer_data_pre; std::vector<GLfloat> m_val_buffer_data_pre;
    Shader m_S4DShaderInstanced;
    unsigned int m_numOf4DSpltas = 0;
    const Geometry::Quad quad;
    int m_steps_in_time = 50;
public:
    float m_time = 0.0f, m_min_opacity = 0.0f;
    bool m_do_sort = true, m_gpu_keys = false;
    LinearMotion(Renderer& r, Camera& c, std::vector<float> model) : m_renderer(r), m_camera(c), m_vModelData(std::move(model)) {
        m_S4DShaderInstanced.AddShaderSource("../Shader/Splats4D/Splat4DFragShader.GLSL", GL_FRAGMENT_SHADER);          // Scenes.h:214-218
        m_S4DShaderInstanced.AddShaderSource("../Shader/Splats4D/Splat4DVertexShaderInstanced.GLSL", GL_VERTEX_SHADER);
        m_S4DShaderInstanced.BuildShader();
    }
    ~LinearMotion() { if (m_key_buf) GLCall(glDeleteBuffers(1, &m_key_buf)); if (m_values_buf) GLCall(glDeleteBuffers(1, &m_values_buf)); }
    void init() {                                                                                                         // Scenes.h:226-289
        m_camera.SetPosition({ 60, 90, 90 }); m_camera.SetOrientation({ 0, -1.0f, -1.0f });
        const size_t nverts = m_vModelData.size() / 6;
        m_numOf4DSpltas = (unsigned)(nverts * m_steps_in_time);
        m_sdata.resize(m_numOf4DSpltas);
        glGenBuffers(1, &m_key_buf); glBindBuffer(GL_SHADER_STORAGE_BUFFER, m_key_buf);
        glBufferStorage(GL_SHADER_STORAGE_BUFFER, m_numOf4DSpltas * sizeof(unsigned int), nullptr, GL_DYNAMIC_STORAGE_BIT);
        glGenBuffers(1, &m_values_buf); glBindBuffer(GL_SHADER_STORAGE_BUFFER, m_values_buf);
        glBufferStorage(GL_SHADER_STORAGE_BUFFER, m_numOf4DSpltas * sizeof(float), nullptr, GL_DYNAMIC_STORAGE_BIT);
        const float sc[3] = { 4.0f, 4.0f, 1.0f };
        gs4d_host_scene_linear(nverts, m_vModelData.data(), m_steps_in_time, 1.0f, 5.0f, sc, 1.0f, 0.5f, 1.0f, &m_sdata[0].pos[0]);   // NEW: Scenes.h:258-279 loop
        m_key_buffer_data_pre.resize(m_numOf4DSpltas); m_val_buffer_data_pre.assign(m_numOf4DSpltas, 0.0f);
        for (unsigned i = 0; i < m_numOf4DSpltas; ++i) m_key_buffer_data_pre[i] = i;
        glBindBuffer(GL_SHADER_STORAGE_BUFFER, m_key_buf);
        glBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, m_numOf4DSpltas * sizeof(GLuint), m_key_buffer_data_pre.data());
        m_sorter = std::make_unique<radix_sort::sorter>(m_numOf4DSpltas);
        m_ssbo_splat_data = std::make_unique<ShareStorageBuffer>(m_sdata.data(), (unsigned)(m_numOf4DSpltas * sizeof(SplatData)));
    }
    void unload() { m_ssbo_splat_data.reset(); m_sorter.reset(); if (m_key_buf) GLCall(glDeleteBuffers(1, &m_key_buf)); if (m_values_buf) GLCall(glDeleteBuffers(1, &m_values_buf)); }   // Scenes.h:291-299 (names not zeroed: the dtor deletes again)
    void Render() {                                                                                                       // Scenes.h:301-340
        if (m_do_sort) {
            if (m_gpu_keys) {                                                                                             // NEW: replaces :314-325
                auto cp = m_camera.GetPosition();
                const float cam[3] = { cp.x, cp.y, cp.z };
                gs4d::compat::Check(gs4d_keygen(gs4d::compat::Current(), gs4d::compat::gl().of(m_ssbo_splat_data->Name()), m_time, cam, gs4d::compat::gl().of(m_values_buf),
                                                gs4d::compat::gl().of(m_key_buf), m_numOf4DSpltas, GS4D_KEY_REF_INV_EUCLID), "gs4d_keygen");
            } else {
                auto cp = m_camera.GetPosition();
                for (unsigned i = 0; i < m_numOf4DSpltas; ++i) {
                    m_key_buffer_data_pre[i] = i;
                    float m[4]; m_sdata[i].GetMeanInTime(m_time, m);
                    float tx = m[0] - cp.x, ty = m[1] - cp.y, tz = m[2] - cp.z;
                    m_val_buffer_data_pre[i] = 1.0f / sqrtf(tx * tx + ty * ty + tz * tz);
                }
                glBindBuffer(GL_SHADER_STORAGE_BUFFER, m_key_buf);
                glBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, m_numOf4DSpltas * sizeof(GLuint), m_key_buffer_data_pre.data());
                glBindBuffer(GL_SHADER_STORAGE_BUFFER, m_values_buf);
                glBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, m_numOf4DSpltas * sizeof(GLfloat), m_val_buffer_data_pre.data());
            }
            m_sorter->sort(m_values_buf, m_key_buf, m_numOf4DSpltas);
        }
        m_S4DShaderInstanced.Bind();
        m_S4DShaderInstanced.SetUniform1f("uTime", m_time);
        m_S4DShaderInstanced.SetUniform1f("uMinOpacity", m_min_opacity);
        m_S4DShaderInstanced.SetUniformMat4f("uView", m_camera.GetViewMatrix());
        m_S4DShaderInstanced.SetUniformMat4f("uProj", m_camera.GetProjMatrix());
        GLCall(glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 1, m_key_buf));
        m_ssbo_splat_data->Bind(2);
        m_renderer.Draw(quad.QuadVA, quad.QuadIdxBuffer, (int)m_numOf4DSpltas);
    }
    unsigned count() const { return m_numOf4DSpltas; }
};

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: scene_replay <teapot_vdata.bin> <out.rgba32f> <width> <height> <time> [--gpu-keys] [--no-sort]\n"); return 2; }
    const int W = std::atoi(argv[3]), H = std::atoi(argv[4]);
    const float t = (float)std::atof(argv[5]);
    bool gpu_keys = false, do_sort = true;
    int frames = 1; const char* png = nullptr; const char* keys = ""; float dt = 0.25f;      // Scenes.h:187 m_time_speed
    for (int i = 6; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--gpu-keys")) gpu_keys = true;
        if (!std::strcmp(argv[i], "--no-sort")) do_sort = false;
        if (!std::strcmp(argv[i], "--frames") && i + 1 < argc) frames = std::atoi(argv[++i]);
        if (!std::strcmp(argv[i], "--png") && i + 1 < argc) png = argv[++i];
        if (!std::strcmp(argv[i], "--keys") && i + 1 < argc) keys = argv[++i];
        if (!std::strcmp(argv[i], "--dt") && i + 1 < argc) dt = (float)std::atof(argv[++i]);
    }
    std::vector<float> model;
    { FILE* f = std::fopen(argv[1], "rb"); if (!f) { std::perror(argv[1]); return 1; } float v; while (std::fread(&v, 4, 1, f) == 1) model.push_back(v); std::fclose(f); }
    gs4d_ctx* ctx = nullptr;
    if (gs4d_create(0, W, H, &ctx) != GS4D_OK) { std::fprintf(stderr, "gs4d_create: %s\n", gs4d_last_error(nullptr)); return 1; }
    gs4d::compat::MakeCurrent(ctx);
    try {
        // Application.cpp:69,125-126,137-154: viewport, clear colour, far plane, blend state, then the frame loop body
        Camera cam(W, H); cam.SetFar(5000.0f);
        glClearColor(0.18431373f, 0.20784314f, 0.25882353f, 1.0f);
        Renderer renderer;
        {
            LinearMotion scene(renderer, cam, model);
            scene.init();
            scene.m_time = t; scene.m_do_sort = do_sort; scene.m_gpu_keys = gpu_keys;
            GLFWwindow window;                                                     // input state only
            for (const char* k = keys; *k; ++k) { const int code = *k == '_' ? GLFW_KEY_SPACE : (*k >= 'a' && *k <= 'z') ? *k - 32 : *k; if (code >= 0 && code < 512) window.keys[code] = 1; }
            // the presentation buffer is a library buffer: its device pointer is what the pack kernel writes, gs4d_buffer_read brings it back
            void* dev8 = nullptr; std::vector<uint8_t> host8; gs4d_buf pbuf = 0;
            if (png) {
                gs4d::compat::Check(gs4d_buffer_create(ctx, nullptr, (size_t)W * H * 4, &pbuf), "gs4d_buffer_create");
                size_t nb = 0; gs4d::compat::Check(gs4d_buffer_device_ptr(ctx, pbuf, &dev8, &nb), "gs4d_buffer_device_ptr");
                host8.resize((size_t)W * H * 4);
            }
            auto present = [&](int frames_back, int index) {
                // RGBA8 pack on the device behind the frame's own compositing kernel, then a copy the host waits for: the next frame is already queued
                gs4d::compat::Check(gs4d_read_frame_rgba8_device(ctx, frames_back, dev8, (size_t)W * H * 4), "gs4d_read_frame_rgba8_device");
                gs4d::compat::Check(gs4d_finish(ctx), "gs4d_finish");
                gs4d::compat::Check(gs4d_buffer_read(ctx, pbuf, 0, host8.data(), host8.size()), "gs4d_buffer_read");
                char name[512]; std::snprintf(name, sizeof name, "%s_%04d.png", png, index);
                if (gs4d_host_write_png(name, host8.data(), W, H) != 0) throw std::runtime_error(std::string("cannot write ") + name);
            };
            uint64_t st[8] = { 0 };
            gs4d::compat::Check(gs4d_get_stats(ctx, st), "gs4d_get_stats");
            const bool swap_chain = (st[6] & 0xFFFFu) >= 2;                    // one frame lane (GS4D_LANES=1): no previous image is kept, present the current one
            for (int f = 0; f < frames; ++f) {                                     // Application.cpp:145-190
                if (png && !swap_chain && f > 0) present(0, f - 1);                // no swap chain: the finished image goes out before Clear() starts the next one
                renderer.Clear();
                cam.HandleInput(&window);
                glBlendFunc(GL_SRC_ALPHA, GL_ONE_MINUS_SRC_ALPHA);
                glEnable(GL_BLEND); glDisable(GL_DEPTH_TEST);
                if (f > 0) scene.m_time += dt;                                     // Update(): m_time += m_time_speed (Scenes.h:346)
                scene.Render();
                if (png && swap_chain && f > 0) present(1, f - 1);                 // the previous image of the swap chain
            }
            if (png) present(0, frames - 1);
            if (pbuf) gs4d_buffer_destroy(ctx, pbuf);
            std::vector<float> img((size_t)W * H * 4);
            gs4d::compat::Check(gs4d_read_pixels(ctx, img.data(), img.size() * 4), "gs4d_read_pixels");
            FILE* f = std::fopen(argv[2], "wb"); if (!f) { std::perror(argv[2]); return 1; }
            std::fwrite(img.data(), 4, img.size(), f); std::fclose(f);
            std::printf("scene_replay: %u splats, %dx%d, t=%g, keys=%s, sort=%d\n", scene.count(), W, H, t, gpu_keys ? "gpu" : "cpu", (int)do_sort);
            scene.unload();
        }   // ~LinearMotion deletes the key/value names a second time: tolerated
    } catch (const std::exception& e) { std::fprintf(stderr, "scene_replay failed: %s\n", e.what()); gs4d_destroy(ctx); return 1; }
    gs4d_destroy(ctx);
    return 0;
}
