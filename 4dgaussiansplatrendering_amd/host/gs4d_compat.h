// gs4d_compat.h — C++ mirror of the reference's host classes on this path, implemented over the C ABI (include/gs4d.h).
//
// Same class names, method names, argument meaning and error behaviour as the reference, so that the reference's own scene code
// (4DSplatRendering/Splat.h, Scene.h, Scenes.h — unmodified) drives the HIP renderer:
//   ShareStorageBuffer            4DSplatRendering/ShareStorageBuffer.h:13-21, .cpp:3-40
//   Shader                        4DSplatRendering/Shader.h:28-72  (AddShaderSource/BuildShader/Bind/SetUniform*)
//   Renderer                      4DSplatRendering/Renderer.h:27-50, .cpp:20-215 (Clear, Draw, Draw instanced, DrawLine/DrawGrid/DrawAxis)
//   radix_sort::sorter            Dependencies/GPU_RADIX_SORT/radix_sort.hpp:219, 258
//   VertexBuffer/IndexBuffer/VertexArray/VertexBufferLayout, Geometry::*   VertexBuffer.h, IndexBuffer.h, VertexArray.h, VertexBufferLayout.h, Geometry.h
//   Camera                        4DSplatRendering/Camera.h:16-85, .cpp:5-239 (getters, setters, HandleInput as a state machine)
//   raw GL used by the scenes     glGenBuffers/glBindBuffer/glBufferStorage/glBufferSubData/glBindBufferBase/glDeleteBuffers
//                                 (Scenes.h:241-247, 282-283, 321-325, 336, 220-224) + glClearColor/glBlendFunc/glViewport
// No OpenGL behind it: buffer "names" are gs4d_buf handles of the current gs4d context (gs4d::compat::MakeCurrent).
//
// Two ways to use it.  (1) With the reference tree: host/shadow/ holds one forwarding header per reference header this file
// replaces (Shader.h, Renderer.h, Camera.h, Geometry.h, VertexArray.h, VertexBuffer.h, IndexBuffer.h, VertexBufferLayout.h,
// ShareStorageBuffer.h, radix_sort.hpp, GLEW/glew.h, GLFW/glfw3.h, imgui.h); copied over the reference's files of the same names
//  the rest of the tree compiles unchanged, and with GLM on the include path every vector / matrix in this file IS the
// glm type.  (2) Without GLM (host/scene_replay.cpp, the GPU box): small stand-in vec/mat structs with the same layout.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/gs4d.h"

#if !defined(GS4D_COMPAT_NO_GLM) && defined(__has_include)
#if __has_include(<glm/glm.hpp>)
#define GS4D_COMPAT_GLM 1
#include <glm/glm.hpp>
#endif
#endif

namespace gs4d { namespace compat {

#ifdef GS4D_COMPAT_GLM
using vec2 = glm::vec2; using vec3 = glm::vec3; using vec4 = glm::vec4; using mat2 = glm::mat2; using mat3 = glm::mat3; using mat4 = glm::mat4;
#else
struct vec2 { float x, y; vec2(float a = 0, float b = 0) : x(a), y(b) {} float& operator[](int i) { return (&x)[i]; } const float& operator[](int i) const { return (&x)[i]; } };
struct vec3 { float x, y, z; vec3(float a = 0, float b = 0, float c = 0) : x(a), y(b), z(c) {} float& operator[](int i) { return (&x)[i]; } const float& operator[](int i) const { return (&x)[i]; } };
struct vec4 { float x, y, z, w; vec4(float a = 0, float b = 0, float c = 0, float d = 0) : x(a), y(b), z(c), w(d) {} float& operator[](int i) { return (&x)[i]; } const float& operator[](int i) const { return (&x)[i]; } };
struct mat2 { float m[2][2]; float* operator[](int c) { return m[c]; } const float* operator[](int c) const { return m[c]; } };
struct mat3 { float m[3][3]; float* operator[](int c) { return m[c]; } const float* operator[](int c) const { return m[c]; } };
struct mat4 { float m[4][4]; float* operator[](int c) { return m[c]; } const float* operator[](int c) const { return m[c]; } };
#endif
static_assert(sizeof(vec3) == 12 && sizeof(vec4) == 16 && sizeof(mat4) == 64, "vectors and matrices are packed column-major floats");

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
    unsigned int bound_ssbo = 0;             // glBindBuffer(target, name): one binding point is enough for what the scenes do
    gs4d_buf of(unsigned int name) const { return name < handle.size() ? handle[name] : 0; }
};
inline GLState& gl() { static thread_local GLState s; return s; }

} } // namespace gs4d::compat

// ---- the GL types, constants and raw calls scenes make themselves ----------------------------------------------
typedef unsigned int GLuint; typedef int GLint; typedef int GLsizei; typedef unsigned int GLenum; typedef float GLfloat; typedef std::ptrdiff_t GLsizeiptr; typedef std::ptrdiff_t GLintptr;
typedef unsigned int GLbitfield; typedef unsigned char GLboolean; typedef void GLvoid;
#ifndef GL_SHADER_STORAGE_BUFFER
#define GL_SHADER_STORAGE_BUFFER 0x90D2
#define GL_ARRAY_BUFFER 0x8892
#define GL_ELEMENT_ARRAY_BUFFER 0x8893
#define GL_DYNAMIC_STORAGE_BIT 0x0100
#define GL_DYNAMIC_DRAW 0x88E8
#define GL_STATIC_DRAW 0x88E4
#define GL_FRAGMENT_SHADER 0x8B30
#define GL_VERTEX_SHADER 0x8B31
#define GL_COMPUTE_SHADER 0x91B9
#define GL_ZERO 0
#define GL_ONE 1
#define GL_SRC_COLOR 0x0300
#define GL_ONE_MINUS_SRC_COLOR 0x0301
#define GL_SRC_ALPHA 0x0302
#define GL_ONE_MINUS_SRC_ALPHA 0x0303
#define GL_DST_ALPHA 0x0304
#define GL_ONE_MINUS_DST_ALPHA 0x0305
#define GL_DST_COLOR 0x0306
#define GL_ONE_MINUS_DST_COLOR 0x0307
#define GL_CONSTANT_COLOR 0x8001
#define GL_ONE_MINUS_CONSTANT_COLOR 0x8002
#define GL_CONSTANT_ALPHA 0x8003
#define GL_ONE_MINUS_CONSTANT_ALPHA 0x8004
#define GL_COLOR_BUFFER_BIT 0x4000
#define GL_DEPTH_BUFFER_BIT 0x0100
#define GL_BLEND 0x0BE2
#define GL_DEPTH_TEST 0x0B71
#define GL_FLOAT 0x1406
#define GL_UNSIGNED_INT 0x1405
#define GL_UNSIGNED_BYTE 0x1401
#define GL_FALSE 0
#define GL_TRUE 1
#define GL_NO_ERROR 0
#define GL_TRIANGLES 0x0004
#define GL_LINES 0x0001
#define GL_LINE_STRIP 0x0003
#endif
// Renderer.h:16-22: the error-check wrapper.  A failing call has already thrown (Check), so the wrapper only evaluates its argument.
#ifndef ASSERT
#define ASSERT(x) if (!(x)) throw std::runtime_error("ASSERT(" #x ")")
#endif
#ifndef GLCall
#define GLCall(x) x
#endif
inline void GLClearError() {}
inline bool GLLogCall(const char*, const char*, int) { return true; }
inline GLenum glGetError() { return GL_NO_ERROR; }

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
inline void glClear(GLbitfield) { gs4d::compat::Check(gs4d_clear(gs4d::compat::Current()), "glClear"); }
inline void glEnable(GLenum) {}              // Application.cpp:153-163: blending is always on and the depth test always off on this path
inline void glDisable(GLenum) {}
inline void glLineWidth(GLfloat) {}          // the width travels with gs4d_draw_lines

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
struct ShaderSource { unsigned int type; std::string path; std::string source; };
class Shader {
public:
    Shader() {}
    Shader(Shader&) = delete;
    void AddShaderSource(const std::string& path, ShaderType type) {
        // the pair of GLSL files a scene names identifies which fixed pipeline it wants
        if (path.find("Splat4DVertexShaderInstanced") != std::string::npos) m_mode = GS4D_MODE_4D_SORTED;
        else if (path.find("Splat4DVertexShaderMod") != std::string::npos) m_mode = GS4D_MODE_4D_DIRECT;
        else if (path.find("Splat3DVertexShaderFull") != std::string::npos) m_mode = GS4D_MODE_3D_FULL;
        else if (path.find("Splat2DVSI") != std::string::npos) m_mode = GS4D_MODE_2D;
        else if (path.find("Shader/Lines/") != std::string::npos) m_lines = true;         // Renderer.h:29-34: drawn by gs4d_draw_lines
        m_sources.push_back({ type, path, std::string() });
    }
    void BuildShader() { if (m_mode < 0 && !m_lines && !m_sources.empty()) std::fprintf(stderr, "[gs4d] Shader: no pipeline of this path matches %s; draws with it are ignored\n", m_sources.back().path.c_str()); }   // Shader.cpp:47-75 logs and continues
    void RebuildShader() { BuildShader(); }
    void Bind() const { if (m_mode >= 0) gs4d::compat::Check(gs4d_set_mode(gs4d::compat::Current(), m_mode), "Shader::Bind"); }
    void Unbind() const {}
    void SetUniform1f(const std::string& name, float v) {
        if (m_mode < 0) return;
        if (name == "uTime") gs4d::compat::Check(gs4d_set_uniform_1f(gs4d::compat::Current(), GS4D_U_TIME, v), "SetUniform1f(uTime)");
        else if (name == "uMinOpacity") gs4d::compat::Check(gs4d_set_uniform_1f(gs4d::compat::Current(), GS4D_U_MIN_OPACITY, v), "SetUniform1f(uMinOpacity)");
    }
    template <class M> void SetUniformMat4f(const std::string& name, const M& m) {
        static_assert(sizeof(M) == 64, "SetUniformMat4f expects 16 column-major floats (glm::mat4)");
        const float* p = reinterpret_cast<const float*>(&m);
        Keep(name, p, 16);
        if (m_mode < 0) return;
        if (name == "uView") gs4d::compat::Check(gs4d_set_uniform_mat4(gs4d::compat::Current(), GS4D_U_VIEW, p), "SetUniformMat4f(uView)");
        else if (name == "uProj") gs4d::compat::Check(gs4d_set_uniform_mat4(gs4d::compat::Current(), GS4D_U_PROJ, p), "SetUniformMat4f(uProj)");
    }
    // Uniforms of other arities exist in the reference for the legacy per-splat shaders (Splat.h:163-247, 355-431, 584-600).  No pipeline
    // of this path consumes them; the last value given to each name is kept and can be read back (LastUniform) — that is how the
    // test tree's golden-vector harness observes what the reference's Splat4D::Draw / Splat3D::Draw compute on the CPU.
    void SetUniform1i(const std::string& n, int v) { const float f[1] = { (float)v }; Keep(n, f, 1); }
    void SetUniform2f(const std::string& n, float a, float b) { const float f[2] = { a, b }; Keep(n, f, 2); }
    void SetUniform3f(const std::string& n, float a, float b, float c) { const float f[3] = { a, b, c }; Keep(n, f, 3); }
    void SetUniform4f(const std::string& n, float a, float b, float c, float d) { const float f[4] = { a, b, c, d }; Keep(n, f, 4); }
    void SetUniform2f(const std::string& n, gs4d::compat::vec2 v) { Keep(n, &v[0], 2); }                  // Shader.h:47-59
    void SetUniform3f(const std::string& n, gs4d::compat::vec3 v) { Keep(n, &v[0], 3); }
    void SetUniform4f(const std::string& n, gs4d::compat::vec4 v) { Keep(n, &v[0], 4); }
    void SetUniformMat3f(const std::string& n, gs4d::compat::mat3 m) { Keep(n, &m[0][0], 9); }            // Shader.h:64-68
    void SetUniformMat2f(const std::string& n, gs4d::compat::mat2 m) { Keep(n, &m[0][0], 4); }
    // not in the reference: the last value set for `name` (up to 16 floats); returns the number of floats it had, 0 if it was never set
    int LastUniform(const std::string& name, float* out, int max_floats) const {
        const auto it = m_last.find(name);
        if (it == m_last.end()) return 0;
        for (int i = 0; i < it->second.n && i < max_floats; ++i) out[i] = it->second.v[i];
        return it->second.n;
    }
    void ForgetUniforms() { m_last.clear(); }
    static bool TryCompile(std::string, ShaderType) { return true; }               // no GLSL is compiled on this path
    void DispatchCompute(unsigned int, unsigned int, unsigned int) {}
    void DispatchCompute(unsigned int, unsigned int) {}
    unsigned int GetRenderID() { return 0; }
    size_t GetShaderID() { return 0; }
    std::vector<ShaderSource> GetShaderSources() { return m_sources; }
    ShaderSource* GetShourceByIdx(int idx) { return idx >= 0 && (size_t)idx < m_sources.size() ? &m_sources[idx] : nullptr; }
    size_t GetShaderSourceSize() { return m_sources.size(); }
    int Mode() const { return m_mode; }
private:
    struct Kept { int n = 0; float v[16] = { 0 }; };
    void Keep(const std::string& name, const float* f, int n) { Kept k; k.n = n; for (int i = 0; i < n && i < 16; ++i) k.v[i] = f[i]; m_last[name] = k; }
    int m_mode = -1;
    bool m_lines = false;
    std::vector<ShaderSource> m_sources;
    std::unordered_map<std::string, Kept> m_last;
};

// ---- vertex-side wrappers: API shape only (the unit quad of Geometry.h:44-50 is built into the rasteriser) -----
class VertexBuffer {
public:
    VertexBuffer(const void* data, unsigned int size) { glGenBuffers(1, &m_id); glBindBuffer(GL_ARRAY_BUFFER, m_id); glBufferData(GL_ARRAY_BUFFER, size, data, GL_DYNAMIC_DRAW); }
    ~VertexBuffer() { glDeleteBuffers(1, &m_id); }
    VertexBuffer(const VertexBuffer&) = delete;
    void Bind() const {} void Unbind() const {}
    void SubData(unsigned int offset, const void* data, unsigned int size) const { glBindBuffer(GL_ARRAY_BUFFER, m_id); glBufferSubData(GL_ARRAY_BUFFER, offset, size, data); }
    void SubData(const void* data, unsigned int size) const { SubData(0, data, size); }
    bool isDynamic() { return true; }
    unsigned int Name() const { return m_id; }
private:
    unsigned int m_id = 0;
};
class IndexBuffer {
public:
    IndexBuffer(const unsigned int*, unsigned int count) : m_Count(count) {}
    void Bind() const {} void Unbind() const {}
    unsigned int GetCount() const { return m_Count; }
    void SubData(unsigned int, const void*, unsigned int) const {}
private:
    unsigned int m_Count;
};
struct VertexBufferElement { unsigned int type, count, normalized; unsigned int customOffset = 0; bool useCustomOffset = false;
    static unsigned int GetSizeOfType(unsigned int type) { return type == GL_UNSIGNED_BYTE ? 1u : 4u; } };
class VertexBufferLayout {                                                         // VertexBufferLayout.h:39-128: the layout is fixed by the pipeline mode here
public:
    template <class T> void Push(unsigned int) {}
    template <class T> void Push() {}
    unsigned int GetStride() const { return 0; }
};
class VertexArray {
public:
    void AddBuffer(const VertexBuffer& vb, const VertexBufferLayout&) { m_vb = vb.Name(); }
    void Bind() const {} void Unbind() const {}
    unsigned int BoundVertexBuffer() const { return m_vb; }
private:
    unsigned int m_vb = 0;
};
namespace Geometry {                                                              // Geometry.h:18-68
struct Vertex { gs4d::compat::vec2 Position; gs4d::compat::vec4 Color; };
struct Vertex2D { gs4d::compat::vec2 Position; };
struct Splat4DVertex { gs4d::compat::vec2 VPosition; gs4d::compat::vec4 SPosition; gs4d::compat::vec4 Color; gs4d::compat::mat4 GeoInfo; };
struct Splat3DVertex { gs4d::compat::vec2 VPosition; gs4d::compat::vec3 SPosition; gs4d::compat::vec4 Color; gs4d::compat::mat3 GeoInfo; };
static_assert(sizeof(Splat3DVertex) == 72, "the 72-byte vertex gs4d_draw_quads takes (Geometry.h:37-42)");
const Vertex2D QuadVerteices[] = { { { 0.5f, 0.5f } }, { { 0.5f, -0.5f } }, { { -0.5f, -0.5f } }, { { -0.5f, 0.5f } } };
const unsigned int QuadIdxBufferData[] = { 0, 2, 1, 2, 0, 3 };
struct Quad {
    VertexBuffer QuadVB{ QuadVerteices, 4 * sizeof(Vertex2D) };
    IndexBuffer QuadIdxBuffer{ QuadIdxBufferData, 6 };
    VertexArray QuadVA{};
    VertexBufferLayout QuadVBLayout{};
    Quad() { QuadVBLayout.Push<gs4d::compat::vec2>(); QuadVA.AddBuffer(QuadVB, QuadVBLayout); }
    ~Quad() {}
};
}

// ---- input model behind the GLFW calls the reference makes (host/shadow/GLFW/glfw3.h): a window is just its input state -------
struct GLFWwindow {
    unsigned char keys[512] = { 0 };         // indexed by GLFW key code: 1 = pressed
    double cursor_x = 0.0, cursor_y = 0.0;
    int cursor_mode = 0;
    bool should_close = false;
};
#ifndef GLFW_PRESS
#define GLFW_RELEASE 0
#define GLFW_PRESS 1
#define GLFW_KEY_SPACE 32
#define GLFW_KEY_A 65
#define GLFW_KEY_C 67
#define GLFW_KEY_D 68
#define GLFW_KEY_E 69
#define GLFW_KEY_M 77
#define GLFW_KEY_Q 81
#define GLFW_KEY_S 83
#define GLFW_KEY_W 87
#define GLFW_KEY_ESCAPE 256
#define GLFW_KEY_LEFT_SHIFT 340
#define GLFW_KEY_LEFT_CONTROL 341
#define GLFW_CURSOR 0x00033001
#define GLFW_CURSOR_NORMAL 0x00034001
#define GLFW_CURSOR_HIDDEN 0x00034002
inline int glfwGetKey(GLFWwindow* w, int key) { return (w && key >= 0 && key < 512 && w->keys[key]) ? GLFW_PRESS : GLFW_RELEASE; }
inline void glfwSetInputMode(GLFWwindow* w, int mode, int value) { if (w && mode == GLFW_CURSOR) w->cursor_mode = value; }
inline void glfwSetCursorPos(GLFWwindow* w, double x, double y) { if (w) { w->cursor_x = x; w->cursor_y = y; } }
inline void glfwGetCursorPos(GLFWwindow* w, double* x, double* y) { if (x) *x = w ? w->cursor_x : 0.0; if (y) *y = w ? w->cursor_y : 0.0; }
inline int glfwWindowShouldClose(GLFWwindow* w) { return w ? (int)w->should_close : 1; }
inline void glfwPollEvents() {}
#endif

// ---- Camera (Camera.h:16-85, Camera.cpp) ---------------------------------------------------------------------------
class Camera {
public:
    gs4d::compat::vec3 position, orientation{ 1.0f, 0.0f, 0.0f }, up{ 0.0f, 1.0f, 0.0f };
    Camera(int width, int height) : position(0.0f, 0.0f, 0.0f), orientation(0.0f, 0.0f, -1.0f), mWidth(width), mHeight(height) {}
    Camera(int width, int height, gs4d::compat::vec3 startPos) : position(startPos), orientation(0.0f, 0.0f, -1.0f), mWidth(width), mHeight(height) {}
    Camera(int width, int height, gs4d::compat::vec3 startPos, gs4d::compat::vec3 o) : position(startPos), orientation(o), mWidth(width), mHeight(height) {}
    Camera(int width, int height, gs4d::compat::vec3 startPos, gs4d::compat::vec3 o, gs4d::compat::vec3 u) : position(startPos), orientation(o), up(u), mWidth(width), mHeight(height) {}
    gs4d::compat::mat4 GetViewMatrix() { gs4d::compat::mat4 m; gs4d_host_look_at(&position[0], &orientation[0], &up[0], reinterpret_cast<float*>(&m)); return m; }      // Camera.cpp:50-53
    gs4d::compat::mat4 GetProjMatrix() { gs4d::compat::mat4 m; gs4d_host_perspective(mFOV, mWidth, mHeight, mNear, mFar, reinterpret_cast<float*>(&m)); return m; }   // Camera.cpp:55-58
    // glm::mat4 operator*: Result[c] = sum_k A[k] * B[c][k], summed left to right (type_mat4x4.inl), A = proj, B = view (Camera.cpp:44-48)
    gs4d::compat::mat4 GetViewProjMatrix() {
        gs4d::compat::mat4 P = GetProjMatrix(), V = GetViewMatrix(), R;
        for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) R[c][r] = ((P[0][r] * V[c][0] + P[1][r] * V[c][1]) + P[2][r] * V[c][2]) + P[3][r] * V[c][3];
        return R;
    }
    gs4d::compat::vec3 GetPosition() { return position; }
    float GetFar() { return mFar; } float GetNear() { return mNear; } float GetFOV() { return mFOV; }
    float GetScreenWidth() { return (float)mWidth; } float GetScreenHeight() { return (float)mHeight; }
    gs4d::compat::vec2 GetViewport() { float v[2]; gs4d_host_camera_viewport(mWidth, mHeight, v); return gs4d::compat::vec2(v[0], v[1]); }             // Camera.cpp:90-93
    gs4d::compat::vec2 GetFocal() { float v[2]; gs4d_host_camera_focal(mFOV, mWidth, mHeight, v); return gs4d::compat::vec2(v[0], v[1]); }            // Camera.cpp:95-99
    float GetSpeed() { return mSpeed; }
    bool IsViewFixed() { return mFixViewPoint; }
    bool IsPositionFixed() { return mFixPostion; }
    // Camera.cpp:116-207: WASD / E,Q roll / space, ctrl / C captures the mouse, Esc releases it; the arithmetic is gs4d_host_camera_input
    void HandleInput(GLFWwindow* window, bool imguiActive = false) {
        gs4d_camera_state st = State();
        gs4d_camera_input in; std::memset(&in, 0, sizeof in);
        static const struct { int key; unsigned bit; } map[] = { { GLFW_KEY_W, GS4D_CAMKEY_W }, { GLFW_KEY_S, GS4D_CAMKEY_S }, { GLFW_KEY_A, GS4D_CAMKEY_A }, { GLFW_KEY_D, GS4D_CAMKEY_D },
            { GLFW_KEY_E, GS4D_CAMKEY_E }, { GLFW_KEY_Q, GS4D_CAMKEY_Q }, { GLFW_KEY_SPACE, GS4D_CAMKEY_SPACE }, { GLFW_KEY_LEFT_CONTROL, GS4D_CAMKEY_LCTRL }, { GLFW_KEY_LEFT_SHIFT, GS4D_CAMKEY_LSHIFT },
            { GLFW_KEY_C, GS4D_CAMKEY_C }, { GLFW_KEY_ESCAPE, GS4D_CAMKEY_ESC } };
        for (const auto& m : map) if (glfwGetKey(window, m.key) == GLFW_PRESS) in.keys |= m.bit;
        glfwGetCursorPos(window, &in.mouse_x, &in.mouse_y);
        in.imgui_active = imguiActive ? 1 : 0;
        int recenter = 0, hide = 0;
        gs4d_host_camera_input(&st, &in, &recenter, &hide);
        if (hide) glfwSetInputMode(window, GLFW_CURSOR, GLFW_CURSOR_HIDDEN);
        if (recenter) glfwSetCursorPos(window, (double)mWidth / 2.0, (double)mHeight / 2.0);
        Load(st);
    }
    void HandleCamRotation(GLFWwindow* window) {                                   // Camera.cpp:191-207
        gs4d_camera_state st = State();
        double mx, my; glfwGetCursorPos(window, &mx, &my);
        glfwSetCursorPos(window, (double)mWidth / 2.0, (double)mHeight / 2.0);
        gs4d_host_camera_rotate(&st, mx, my);
        Load(st);
    }
    void SetWidth(int width) { mWidth = width; } void SetHeight(int height) { mHeight = height; }
    void SetNear(float v) { mNear = v; } void SetFar(float v) { mFar = v; } void SetFOV(float v) { mFOV = v; }
    void Resize(int w, int h) { mWidth = w; mHeight = h; }
    void SetPosition(gs4d::compat::vec3 p) { position = p; }
    void SetOrientation(gs4d::compat::vec3 o) { orientation = o; }
    void SetUp(gs4d::compat::vec3 u) { up = u; }
    void SetIsViewFixedOnPoint(bool fixed, gs4d::compat::vec4 point) {            // Camera.cpp:209-220
        SetIsViewFixedOnPoint(fixed);
        if (fixed) { gs4d_camera_state st = State(); const float p[3] = { point[0], point[1], point[2] }; gs4d_host_camera_look_at_point(&st, p); Load(st); }
    }
    void SetIsViewFixedOnPoint(bool fixed) { mFixViewPoint = fixed; if (fixed) mCaptureMouse = false; }
    void SetIsPositionFixed(bool fixed) { mFixPostion = fixed; }
    void SetSpeed(float speed) { mSpeed = speed; }
    bool IsLockX() { return mLockX; } bool IsLockY() { return mLockY; }
    void SetLockX(bool lock) { mLockX = lock; } void SetLockY(bool lock) { mLockY = lock; }
private:
    gs4d_camera_state State() const {
        gs4d_camera_state st;
        for (int i = 0; i < 3; ++i) { st.position[i] = position[i]; st.orientation[i] = orientation[i]; st.up[i] = up[i]; }
        st.width = mWidth; st.height = mHeight; st.sensitivity = mSensitivity; st.speed = mSpeed; st.fast_speed = mFastSpeed;
        st.capture_mouse = mCaptureMouse; st.first_capture = mFirstCapture; st.fix_view = mFixViewPoint; st.fix_position = mFixPostion; st.lock_x = mLockX; st.lock_y = mLockY;
        return st;
    }
    void Load(const gs4d_camera_state& st) {
        for (int i = 0; i < 3; ++i) { position[i] = st.position[i]; orientation[i] = st.orientation[i]; up[i] = st.up[i]; }
        mCaptureMouse = st.capture_mouse != 0; mFirstCapture = st.first_capture != 0;
    }
    bool mCaptureMouse = false, mFirstCapture = true, mFixViewPoint = false, mFixPostion = false;
    float mFOV = 60.0f, mNear = 0.1f, mFar = 256.0f;      // Camera.h:71-73
    int mWidth, mHeight;
    float mSensitivity = 100.0f, mSpeed = 0.5f, mFastSpeed = 2.0f;
    bool mLockY = false, mLockX = false;
};

// ---- Renderer (Renderer.h:24-50) -----------------------------------------------------------------------------
class Renderer {
public:
    Renderer() {                                                                   // Renderer.h:28-35: the two line programs
        mLine2D.AddShaderSource("../Shader/Lines/Line2DFrag.GLSL", GL_FRAGMENT_SHADER); mLine2D.AddShaderSource("../Shader/Lines/Line2DVert.GLSL", GL_VERTEX_SHADER); mLine2D.BuildShader();
        mLine3D.AddShaderSource("../Shader/Lines/LineFrag.GLSL", GL_FRAGMENT_SHADER); mLine3D.AddShaderSource("../Shader/Lines/LineVert.GLSL", GL_VERTEX_SHADER); mLine3D.BuildShader();
    }
    void Clear() const { gs4d::compat::Check(gs4d_clear(gs4d::compat::Current()), "Renderer::Clear"); }
    // non-instanced: the 3D-Full path, 4 vertices x 72 B per splat, 6 indices per splat (Scenes.h:1690-1692)
    void Draw(const VertexArray& va, const IndexBuffer& ib) const { gs4d::compat::Check(gs4d_draw_quads(gs4d::compat::Current(), gs4d::compat::gl().of(va.BoundVertexBuffer()), ib.GetCount() / 6), "Renderer::Draw"); }
    void Draw(const VertexArray&, const IndexBuffer&, int instances) const { gs4d::compat::Check(gs4d_draw_instanced(gs4d::compat::Current(), (size_t)instances), "Renderer::Draw(instanced)"); }
    // overlays (Renderer.cpp:41-215): GL_LINES / GL_LINE_STRIP with the flat-colour programs of Shader/Lines, blended like everything else
    void DrawLine(gs4d::compat::vec3 v0, gs4d::compat::vec3 v1, gs4d::compat::vec4 color, Camera& cam, float thickness) {
        const float v[6] = { v0[0], v0[1], v0[2], v1[0], v1[1], v1[2] };
        Lines(v, 2, 3, 0, cam, color, thickness, "Renderer::DrawLine");
    }
    void DrawLine(std::vector<gs4d::compat::vec3>& points, gs4d::compat::vec4 color, Camera& cam, float thickness) {
        if (points.size() < 2) return;
        Lines(reinterpret_cast<const float*>(points.data()), points.size(), 3, 1, cam, color, thickness, "Renderer::DrawLine(strip)");
    }
    void DrawLine(gs4d::compat::vec3 v0, gs4d::compat::vec3 v1, gs4d::compat::vec4 color, Camera& cam) { DrawLine(v0, v1, color, cam, 1.0f); }
    void DrawGrid(float width, float height, const unsigned int divisionsX, const unsigned int divisionsY, gs4d::compat::vec4 color, Camera& cam, float thickness) {
        // Renderer.cpp:113-135 sizes the vector to the line count and then push_back()s: the first `totalLines` vertices are (0,0,0) —
        // degenerate segments that rasterise to nothing.  They are kept (the vertex count is part of the call) and cost nothing.
        const int totalLines = (int)(((divisionsX + 1) * 2) + ((divisionsY + 1) * 2));
        const float distX = width / divisionsX, distY = height / divisionsY;
        std::vector<float> verts((size_t)totalLines * 3, 0.0f);
        const float startX = -width / 2.0f, startY = -height / 2.0f;
        for (int i = 0; i < (int)(divisionsX + 1); ++i) { const float x = startX + (distX * float(i)); verts.insert(verts.end(), { x, 0.0f, startY, x, 0.0f, -startY }); }
        for (int i = 0; i < (int)(divisionsY + 1); ++i) { const float z = startY + (distY * float(i)); verts.insert(verts.end(), { startX, 0.0f, z, -startX, 0.0f, z }); }
        Lines(verts.data(), verts.size() / 3, 3, 0, cam, color, thickness, "Renderer::DrawGrid");
    }
    void DrawLine(gs4d::compat::vec2 v0, gs4d::compat::vec2 v1, gs4d::compat::vec4 color) {                      // Renderer.cpp:170-203: NDC positions, width 3
        const float v[4] = { v0[0], v0[1], v1[0], v1[1] };
        const float c[4] = { color[0], color[1], color[2], color[3] };
        gs4d::compat::Check(gs4d_draw_lines(gs4d::compat::Current(), v, 2, 2, 0, nullptr, c, 3.0f), "Renderer::DrawLine(2D)");
    }
    void DrawAxis(Camera& cam, float length = 1.0f, float thickness = 1.0f) {      // Renderer.cpp:205-215 (`length` is ignored there too: the axes are 10 long)
        (void)length;
        DrawLine(gs4d::compat::vec3(0.0f, 0.0f, 0.0f), gs4d::compat::vec3(10.0f, 0.0f, 0.0f), gs4d::compat::vec4(1.0f, 0.0f, 0.0f, 1.0f), cam, thickness);
        DrawLine(gs4d::compat::vec3(0.0f, 0.0f, 0.0f), gs4d::compat::vec3(0.0f, 10.0f, 0.0f), gs4d::compat::vec4(0.0f, 1.0f, 0.0f, 1.0f), cam, thickness);
        DrawLine(gs4d::compat::vec3(0.0f, 0.0f, 0.0f), gs4d::compat::vec3(0.0f, 0.0f, 10.0f), gs4d::compat::vec4(0.0f, 0.0f, 1.0f, 1.0f), cam, thickness);
    }
private:
    void Lines(const float* v, size_t nverts, int dims, int strip, Camera& cam, const gs4d::compat::vec4& color, float thickness, const char* what) {
        const gs4d::compat::mat4 vp = cam.GetViewProjMatrix();
        const float c[4] = { color[0], color[1], color[2], color[3] };
        gs4d::compat::Check(gs4d_draw_lines(gs4d::compat::Current(), v, nverts, dims, strip, reinterpret_cast<const float*>(&vp), c, thickness), what);
    }
    Shader mLine3D, mLine2D;
};

// ---- radix_sort::sorter (radix_sort.hpp:219, 258) ------------------------------------------------------------
namespace radix_sort {
struct sorter {
    explicit sorter(size_t /*init_arr_len*/) {}          // scratch grows on demand inside the library
    void sort(GLuint key_buf, GLuint val_buf, size_t arr_len) { gs4d::compat::Check(gs4d_sort_pairs(gs4d::compat::Current(), gs4d::compat::gl().of(key_buf), gs4d::compat::gl().of(val_buf), arr_len), "radix_sort::sorter::sort"); }
};
}
