// gs4d_compat.h — C++ mirror of the reference's host classes on this path, implemented over the C ABI (include/gs4d.h).
//
// Same class names, method names, argument meaning and error behaviour as the reference, so that a scene written against
// the reference (Scenes.h:226-340) drives the HIP renderer unchanged:
//   ShareStorageBuffer            4DSplatRendering/ShareStorageBuffer.h:13-21, .cpp:3-40
//   Shader                        4DSplatRendering/Shader.h:28-72  (AddShaderSource/BuildShader/Bind/SetUniform*)
//   Renderer                      4DSplatRendering/Renderer.h:37-39, .cpp:20-39 (Clear, Draw, Draw instanced)
//   radix_sort::sorter            Dependencies/GPU_RADIX_SORT/radix_sort.hpp:219, 258
//   VertexBuffer/IndexBuffer/VertexArray/VertexBufferLayout, Geometry::Quad   (API shape only: the unit quad is implicit)
//   Camera                        4DSplatRendering/Camera.h:29-54, .cpp:50-63
//   raw GL used by the scenes     glGenBuffers/glBindBuffer/glBufferStorage/glBufferSubData/glBindBufferBase/glDeleteBuffers
//                                 (Scenes.h:241-247, 282-283, 321-325, 336, 220-224) + glClearColor/glBlendFunc/glViewport
// No OpenGL behind it: buffer "names" are gs4d_buf handles of the current gs4d context (gs4d::compat::MakeCurrent).
// Matrix/vector arguments are templates: anything laid out like glm::mat4 / glm::vec3 (column-major floats) works, so a
// maintainer keeps passing GLM types; gs4d::compat::mat4/vec3 are provided for builds without GLM.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/gs4d.h"

namespace gs4d { namespace compat {

struct vec3 { float x, y, z; vec3(float a = 0, float b = 0, float c = 0) : x(a), y(b), z(c) {} float& operator[](int i) { return (&x)[i]; } };
struct vec4 { float x, y, z, w; };
struct mat4 { float m[4][4]; float* operator[](int c) { return m[c]; } const float* operator[](int c) const { return m[c]; } };

// ---- the "GL context": one gs4d context is current per thread (Application.cpp:97 glfwMakeContextCurrent) ----
inline gs4d_ctx*& current_slot() { static thread_local gs4d_ctx* c = nullptr; return c; }
inline void MakeCurrent(gs4d_ctx* c) { current_slot() = c; }
inline gs4d_ctx* Current() {
    gs4d_ctx* c = current_slot();
    if (!c) throw std::runtime_error("gs4d::compat: no current context (call gs4d::compat::MakeCurrent)");
    return c;
}
// GLCall(x) in the reference prints the GL error and breaks (Renderer.h:16-19, Renderer.cpp:5-18); here a failing ABI call
// prints the library's message and throws.
inline void Check(int rc, const char* what) {
    if (rc != GS4D_OK) {
        std::string msg = std::string("[gs4d Error]: ( ") + std::to_string(rc) + " ) " + what + " : " + gs4d_last_error(current_slot());
        std::fprintf(stderr, "%s\n", msg.c_str());
        throw std::runtime_error(msg);
    }
}

// ---- GL buffer names: glGenBuffers hands out names before storage exists, so names map to gs4d buffers through a table ----
struct GLState {
    GLState() : handle(1, 0) {}
    std::vector<gs4d_buf> handle;            // GL name -> gs4d buffer (0 = no storage yet / deleted); name 0 is "none"
    unsigned int bound_ssbo = 0;             // glBindBuffer(GL_SHADER_STORAGE_BUFFER, name)
    gs4d_buf of(unsigned int name) const { return name < handle.size() ? handle[name] : 0; }
};
inline GLState& gl() { static thread_local GLState s; return s; }

} } // namespace gs4d::compat

// ---- the GL constants and the raw calls scenes make themselves -------------------------------------------------
typedef unsigned int GLuint; typedef int GLint; typedef int GLsizei; typedef unsigned int GLenum; typedef float GLfloat; typedef std::ptrdiff_t GLsizeiptr; typedef std::ptrdiff_t GLintptr; typedef unsigned int GLbitfield;
#ifndef GL_SHADER_STORAGE_BUFFER
#define GL_SHADER_STORAGE_BUFFER 0x90D2
#define GL_DYNAMIC_STORAGE_BIT 0x0100
#define GL_DYNAMIC_DRAW 0x88E8
#define GL_FRAGMENT_SHADER 0x8B30
#define GL_VERTEX_SHADER 0x8B31
#define GL_SRC_ALPHA 0x0302
#define GL_ONE_MINUS_SRC_ALPHA 0x0303
#define GL_COLOR_BUFFER_BIT 0x4000
#define GL_DEPTH_BUFFER_BIT 0x0100
#endif
#ifndef GLCall
#define GLCall(x) x
#endif

inline void glGenBuffers(GLsizei n, GLuint* names) { auto& g = gs4d::compat::gl(); for (GLsizei i = 0; i < n; ++i) { g.handle.push_back(0); names[i] = (GLuint)(g.handle.size() - 1); } }
inline void glBindBuffer(GLenum, GLuint name) { gs4d::compat::gl().bound_ssbo = name; }
inline void gs4d_gl_alloc(GLsizeiptr size, const void* data, const char* what) {       // (re)creates the storage of the bound name
    using namespace gs4d::compat;
    auto& g = gl();
    if (g.bound_ssbo == 0 || g.bound_ssbo >= g.handle.size()) Check(GS4D_E_INVALID, what);
    if (g.handle[g.bound_ssbo]) Check(gs4d_buffer_destroy(Current(), g.handle[g.bound_ssbo]), what);
    gs4d_buf b = 0;
    Check(gs4d_buffer_create(Current(), data, (size_t)size, &b), what);
    g.handle[g.bound_ssbo] = b;
}
inline void glBufferStorage(GLenum, GLsizeiptr size, const void* data, GLbitfield) { gs4d_gl_alloc(size, data, "glBufferStorage"); }
inline void glBufferData(GLenum, GLsizeiptr size, const void* data, GLenum) { gs4d_gl_alloc(size, data, "glBufferData"); }
inline void glBufferSubData(GLenum, GLintptr offset, GLsizeiptr size, const void* data) {
    using namespace gs4d::compat;
    Check(gs4d_buffer_subdata(Current(), gl().of(gl().bound_ssbo), (size_t)offset, data, (size_t)size), "glBufferSubData");
}
inline void glBindBufferBase(GLenum, GLuint slot, GLuint name) { gs4d::compat::Check(gs4d_bind_storage(gs4d::compat::Current(), (int)slot, gs4d::compat::gl().of(name)), "glBindBufferBase"); }
inline void glDeleteBuffers(GLsizei n, const GLuint* names) {   // deleting 0 / already-deleted names is silently ignored, as in GL (Scenes.h:220-224 + 291-299 delete twice)
    auto& g = gs4d::compat::gl();
    for (GLsizei i = 0; i < n; ++i) {
        if (names[i] == 0 || names[i] >= g.handle.size() || !g.handle[names[i]]) continue;
        if (gs4d::compat::current_slot()) gs4d_buffer_destroy(gs4d::compat::current_slot(), g.handle[names[i]]);
        g.handle[names[i]] = 0;
    }
}
inline void glClearColor(float r, float g, float b, float a) { const float c[4] = { r, g, b, a }; gs4d::compat::Check(gs4d_set_clear_color(gs4d::compat::Current(), c), "glClearColor"); }
inline void glBlendFunc(GLenum s, GLenum d) { gs4d::compat::Check(gs4d_set_blend(gs4d::compat::Current(), (int)s, (int)d), "glBlendFunc"); }
inline void glViewport(GLint, GLint, GLsizei w, GLsizei h) { gs4d::compat::Check(gs4d_resize(gs4d::compat::Current(), w, h), "glViewport"); }

// ---- ShareStorageBuffer (ShareStorageBuffer.h:13-21) -------------------------------------------------------
class ShareStorageBuffer {
public:
    ShareStorageBuffer(const void* data, unsigned int size) {                     // ShareStorageBuffer.cpp:3-8
        glGenBuffers(1, &m_RendererID);
        glBindBuffer(GL_SHADER_STORAGE_BUFFER, m_RendererID);
        glBufferData(GL_SHADER_STORAGE_BUFFER, size, data, GL_DYNAMIC_DRAW);
    }
    ~ShareStorageBuffer() { glDeleteBuffers(1, &m_RendererID); }
    ShareStorageBuffer(const ShareStorageBuffer&) = delete;
    void Bind() const { glBindBuffer(GL_SHADER_STORAGE_BUFFER, m_RendererID); }
    void Bind(int position) const { glBindBufferBase(GL_SHADER_STORAGE_BUFFER, (GLuint)position, m_RendererID); }
    void Unbind() const { glBindBuffer(GL_SHADER_STORAGE_BUFFER, 0); }
    void SubData(unsigned int offset, const void* data, unsigned int size) const { Bind(); glBufferSubData(GL_SHADER_STORAGE_BUFFER, offset, size, data); }
    void SubData(const void* data, unsigned int size) const { SubData(0, data, size); }
    bool isDynamic() { return mIsDynamic; }
    unsigned int Name() const { return m_RendererID; }     // GL-style name (not in the reference: used by the GPU key generation call)
private:
    unsigned int m_RendererID = 0;
    bool mIsDynamic = false;
};

// ---- Shader (Shader.h:28-72): the shader *paths* select the pipeline; uniforms go to the context -----------------
typedef unsigned int ShaderType;
class Shader {
public:
    Shader() {}
    Shader(Shader&) = delete;
    void AddShaderSource(const std::string& path, ShaderType) {
        // the pair of GLSL files a scene names identifies which fixed pipeline it wants
        if (path.find("Splat4DVertexShaderInstanced") != std::string::npos) m_mode = GS4D_MODE_4D_SORTED;
        else if (path.find("Splat4DVertexShaderMod") != std::string::npos) m_mode = GS4D_MODE_4D_DIRECT;
        else if (path.find("Splat3DVertexShaderFull") != std::string::npos) m_mode = GS4D_MODE_3D_FULL;
        else if (path.find("Splat2DVSI") != std::string::npos) m_mode = GS4D_MODE_2D;
        m_paths.push_back(path);
    }
    void BuildShader() { if (m_mode < 0) std::fprintf(stderr, "[gs4d] Shader: no splat pipeline matches the given sources; draws with it are ignored\n"); }   // Shader.cpp:47-75 logs and continues
    void RebuildShader() { BuildShader(); }
    void Bind() const { if (m_mode >= 0) gs4d::compat::Check(gs4d_set_mode(gs4d::compat::Current(), m_mode), "Shader::Bind"); }
    void Unbind() const {}
    void SetUniform1f(const std::string& name, float v) {
        if (name == "uTime") gs4d::compat::Check(gs4d_set_uniform_1f(gs4d::compat::Current(), GS4D_U_TIME, v), "SetUniform1f(uTime)");
        else if (name == "uMinOpacity") gs4d::compat::Check(gs4d_set_uniform_1f(gs4d::compat::Current(), GS4D_U_MIN_OPACITY, v), "SetUniform1f(uMinOpacity)");
    }
    template <class M> void SetUniformMat4f(const std::string& name, const M& m) {
        static_assert(sizeof(M) == 64, "SetUniformMat4f expects 16 column-major floats (glm::mat4)");
        const float* p = reinterpret_cast<const float*>(&m);
        if (name == "uView") gs4d::compat::Check(gs4d_set_uniform_mat4(gs4d::compat::Current(), GS4D_U_VIEW, p), "SetUniformMat4f(uView)");
        else if (name == "uProj") gs4d::compat::Check(gs4d_set_uniform_mat4(gs4d::compat::Current(), GS4D_U_PROJ, p), "SetUniformMat4f(uProj)");
    }
    // uniforms of other arities exist in the reference for the legacy per-splat shaders; accepted and ignored
    void SetUniform1i(const std::string&, int) {}
    void SetUniform2f(const std::string&, float, float) {}
    void SetUniform3f(const std::string&, float, float, float) {}
    void SetUniform4f(const std::string&, float, float, float, float) {}
    template <class V> void SetUniform2f(const std::string&, const V&) {}         // glm::vec2 / vec3 / vec4 overloads (Shader.h:47-59)
    template <class V> void SetUniform3f(const std::string&, const V&) {}
    template <class V> void SetUniform4f(const std::string&, const V&) {}
    template <class M> void SetUniformMat3f(const std::string&, const M&) {}     // Shader.h:64-68
    template <class M> void SetUniformMat2f(const std::string&, const M&) {}
    static bool TryCompile(const std::string&, ShaderType) { return true; }      // no GLSL is compiled on this path
    int Mode() const { return m_mode; }
private:
    int m_mode = -1;
    std::vector<std::string> m_paths;
};

// ---- vertex-side wrappers: API shape only (the unit quad of Geometry.h:44-50 is built into the rasteriser) -----
class VertexBuffer {
public:
    VertexBuffer(const void* data, unsigned int size) { glGenBuffers(1, &m_id); glBindBuffer(GL_SHADER_STORAGE_BUFFER, m_id); glBufferData(GL_SHADER_STORAGE_BUFFER, size, data, GL_DYNAMIC_DRAW); }
    ~VertexBuffer() { glDeleteBuffers(1, &m_id); }
    VertexBuffer(const VertexBuffer&) = delete;
    void Bind() const {} void Unbind() const {}
    void SubData(unsigned int offset, const void* data, unsigned int size) const { glBindBuffer(GL_SHADER_STORAGE_BUFFER, m_id); glBufferSubData(GL_SHADER_STORAGE_BUFFER, offset, size, data); }
    unsigned int Name() const { return m_id; }
private:
    unsigned int m_id = 0;
};
class IndexBuffer {
public:
    IndexBuffer(const unsigned int*, unsigned int count) : m_Count(count) {}
    void Bind() const {} void Unbind() const {}
    unsigned int GetCount() const { return m_Count; }
private:
    unsigned int m_Count;
};
class VertexBufferLayout { public: template <class T> void Push(unsigned int = 1) {} };
class VertexArray {
public:
    void AddBuffer(const VertexBuffer& vb, const VertexBufferLayout&) { m_vb = vb.Name(); }
    void Bind() const {} void Unbind() const {}
    unsigned int BoundVertexBuffer() const { return m_vb; }
private:
    unsigned int m_vb = 0;
};
namespace Geometry {
struct Vertex2D { float Position[2]; };
struct Quad {
    const Vertex2D QuadVerteices[4] = { { { 0.5f, 0.5f } }, { { 0.5f, -0.5f } }, { { -0.5f, -0.5f } }, { { -0.5f, 0.5f } } };
    const unsigned int QuadIdxBufferData[6] = { 0, 2, 1, 2, 0, 3 };
    VertexBuffer QuadVB{ QuadVerteices, sizeof QuadVerteices };
    IndexBuffer QuadIdxBuffer{ QuadIdxBufferData, 6 };
    VertexArray QuadVA{};
    VertexBufferLayout QuadVBLayout{};
    Quad() { QuadVA.AddBuffer(QuadVB, QuadVBLayout); }
};
}

// ---- Renderer (Renderer.h:37-39) -----------------------------------------------------------------------------
class Renderer {
public:
    void Clear() const { gs4d::compat::Check(gs4d_clear(gs4d::compat::Current()), "Renderer::Clear"); }
    // non-instanced: the 3D-Full path, 4 vertices x 72 B per splat, 6 indices per splat (Scenes.h:1690-1692)
    void Draw(const VertexArray& va, const IndexBuffer& ib) const { gs4d::compat::Check(gs4d_draw_quads(gs4d::compat::Current(), gs4d::compat::gl().of(va.BoundVertexBuffer()), ib.GetCount() / 6), "Renderer::Draw"); }
    void Draw(const VertexArray&, const IndexBuffer&, int instances) const { gs4d::compat::Check(gs4d_draw_instanced(gs4d::compat::Current(), (size_t)instances), "Renderer::Draw(instanced)"); }
};

// ---- radix_sort::sorter (radix_sort.hpp:219, 258) ------------------------------------------------------------
namespace radix_sort {
struct sorter {
    explicit sorter(size_t /*init_arr_len*/) {}          // scratch grows on demand inside the library
    void sort(GLuint key_buf, GLuint val_buf, size_t arr_len) { gs4d::compat::Check(gs4d_sort_pairs(gs4d::compat::Current(), gs4d::compat::gl().of(key_buf), gs4d::compat::gl().of(val_buf), arr_len), "radix_sort::sorter::sort"); }
};
}

// ---- Camera (Camera.h:29-54): the three getters the path uses + the setters scenes call --------------------------
class Camera {
public:
    gs4d::compat::vec3 position, orientation{ 1.0f, 0.0f, 0.0f }, up{ 0.0f, 1.0f, 0.0f };
    Camera(int width, int height) : position(0, 0, 0), orientation(0, 0, -1), mWidth(width), mHeight(height) {}
    template <class V> Camera(int width, int height, const V& p) : position(p[0], p[1], p[2]), orientation(0, 0, -1), mWidth(width), mHeight(height) {}
    template <class V> Camera(int width, int height, const V& p, const V& o) : position(p[0], p[1], p[2]), orientation(o[0], o[1], o[2]), mWidth(width), mHeight(height) {}
    gs4d::compat::mat4 GetViewMatrix() { gs4d::compat::mat4 m; gs4d_host_look_at(&position.x, &orientation.x, &up.x, &m.m[0][0]); return m; }
    gs4d::compat::mat4 GetProjMatrix() { gs4d::compat::mat4 m; gs4d_host_perspective(mFOV, mWidth, mHeight, mNear, mFar, &m.m[0][0]); return m; }
    // glm::mat4 operator*: Result[c] = sum_k A[k] * B[c][k], summed left to right (type_mat4x4.inl), A = proj, B = view (Camera.cpp:45-48)
    gs4d::compat::mat4 GetViewProjMatrix() {
        const gs4d::compat::mat4 P = GetProjMatrix(), V = GetViewMatrix(); gs4d::compat::mat4 R;
        for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) R.m[c][r] = ((P.m[0][r] * V.m[c][0] + P.m[1][r] * V.m[c][1]) + P.m[2][r] * V.m[c][2]) + P.m[3][r] * V.m[c][3];
        return R;
    }
    gs4d::compat::vec3 GetPosition() { return position; }
    float GetFar() { return mFar; } float GetNear() { return mNear; } float GetFOV() { return mFOV; }
    float GetScreenWidth() { return (float)mWidth; } float GetScreenHeight() { return (float)mHeight; }
    void SetNear(float v) { mNear = v; } void SetFar(float v) { mFar = v; } void SetFOV(float v) { mFOV = v; }
    void SetWidth(int w) { mWidth = w; } void SetHeight(int h) { mHeight = h; }
    void Resize(int w, int h) { mWidth = w; mHeight = h; }
    template <class V> void SetPosition(const V& p) { position = gs4d::compat::vec3(p[0], p[1], p[2]); }
    template <class V> void SetOrientation(const V& o) { orientation = gs4d::compat::vec3(o[0], o[1], o[2]); }
    template <class V> void SetUp(const V& u) { up = gs4d::compat::vec3(u[0], u[1], u[2]); }
private:
    float mFOV = 60.0f, mNear = 0.1f, mFar = 256.0f;      // Camera.h:71-73
    int mWidth, mHeight;
};
