// refgl — the reference's own GLSL programs EXECUTED, in this build container, by Mesa's software OpenGL.
//
// TEST INFRASTRUCTURE, BUILD CONTAINER ONLY.  Nothing here ships, nothing here runs on the GPU box; only the small
// fixtures it writes (tests/golden/gl_*) travel.  The product (libgs4d.so) never links, loads or calls it.
//
// What runs: the reference's shader TEXT, read unmodified at run time from $GS4D_REF (default /root/reference):
//     Shader/Splats4D/Splat4DVertexShaderInstanced.GLSL + Splat4DFragShader.GLSL   (and Splat4DVertexShaderMod.GLSL: kind 4dmod)
//     Shader/Splats3D/Splat3DVertexShaderFull.GLSL      + Splat3DFragShaderFull.GLSL
//     Shader/Splats2D/Splat2DVSI.GLSL                   + Splat2DFragShader.GLSL
//     Shader/Lines/LineVert.GLSL                        + LineFrag.GLSL  (overlay lines)
//     resources/radix_sort_{count,local_offsets,reorder}.comp.glsl
// compiled by the GLSL compiler of the image's Mesa 23.2.1 and executed by its llvmpipe rasteriser (OpenGL 4.5 core).  No
// source of the reference is copied: this file holds the GL *host calls* that feed those programs, each citing the reference
// call it repeats.
//
// How the context comes up with no X server, no EGL, no package installed: the image holds /usr/lib/x86_64-linux-gnu/dri/
// swrast_dri.so, libglapi.so.0 and GL/internal/dri_interface.h.  This program is its own DRI loader, as libGL/libEGL/gbm are:
// __driDriverGetExtensions_swrast() -> DRI_SWRast::createNewScreen2 (with the DRI_SWRastLoader callbacks a loader must offer;
// they describe the *window* a loader owns — there is none, every draw goes to a framebuffer object, so they report a 16x16
// drawable and move no pixels) -> createContextAttribs(OPENGL_CORE 4.4) -> createNewDrawable -> DRI_Core::bindContext.  GL entry
// points come from _glapi_get_proc_address.
//
// Commands (all files raw little-endian):
//   refgl info
//   refgl draw <4d|3d|2d> <W> <H> <n> <records.bin> <sortidx.bin|-> <uniforms.bin> <out_prefix> [tf] [img] [img8] [img16] [blend S D]
//        records : 4d 96 B/record (SplatData, Scenes.h:22-37); 3d 4 x 72 B vertices per splat (Geometry.h:36-41); 2d 48 B/record
//        sortidx : uint32[n] bound at SSBO slot 1 (4d only; "-" = identity)
//        uniforms: float32 {uTime, uMinOpacity, uView[16], uProj[16]}
//        tf   -> <out_prefix>.tf.f32   : per instance 6 vertices (index order 0,2,1,2,0,3) x captured varyings (the tfv lists in cmd_draw)
//        img  -> <out_prefix>.img.f32  : H x W x RGBA float, row 0 = bottom (glReadPixels order) from an RGBA32F colour attachment
//        img8 -> <out_prefix>.img.u8   : the same draw into an RGBA8 attachment (what the reference's window holds)
//        img16-> <out_prefix>.img.u16  : the same draw into an RGBA16 (unsigned normalised) attachment: a fixed-point framebuffer like the
//                                        reference's window — source and result of every blend clamped to [0, 1] (OpenGL 4.4, 17.3.8), which a
//                                        float attachment does not do — with 1.5e-5 steps instead of 1/255
//   refgl lines <W> <H> <viewproj.bin> <out_prefix> (<color.bin> <width> <nverts> <verts.bin> <strip:0|1|2>)+    all sets into one frame; strip 2 = 2D lines (vec2 NDC positions, Line2DVert/Frag)
//   refgl sort <n> <keys.bin> <vals.bin> <out_prefix>      -> <out_prefix>.keys.u32 / .vals.u32
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <dlfcn.h>
#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>

// ---------------------------------------------------------------------------------------------------------------------------
// the DRI loader side
// ---------------------------------------------------------------------------------------------------------------------------
static void ld_getDrawableInfo(__DRIdrawable*, int* x, int* y, int* w, int* h, void*) { *x = *y = 0; *w = *h = 16; }
static void ld_putImage(__DRIdrawable*, int, int, int, int, int, char*, void*) {}
static void ld_getImage(__DRIdrawable*, int, int, int, int, char*, void*) {}
static void ld_putImage2(__DRIdrawable*, int, int, int, int, int, int, char*, void*) {}
static void ld_getImage2(__DRIdrawable*, int, int, int, int, int, char*, void*) {}

#define GLFUNCS(X) \
    X(PFNGLGETSTRINGPROC, glGetString) X(PFNGLGETERRORPROC, glGetError) X(PFNGLGETINTEGERVPROC, glGetIntegerv) X(PFNGLGETINTEGERI_VPROC, glGetIntegeri_v) \
    X(PFNGLCREATESHADERPROC, glCreateShader) X(PFNGLSHADERSOURCEPROC, glShaderSource) X(PFNGLCOMPILESHADERPROC, glCompileShader) \
    X(PFNGLGETSHADERIVPROC, glGetShaderiv) X(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog) X(PFNGLCREATEPROGRAMPROC, glCreateProgram) \
    X(PFNGLATTACHSHADERPROC, glAttachShader) X(PFNGLLINKPROGRAMPROC, glLinkProgram) X(PFNGLGETPROGRAMIVPROC, glGetProgramiv) \
    X(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog) X(PFNGLUSEPROGRAMPROC, glUseProgram) X(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation) \
    X(PFNGLUNIFORM1FPROC, glUniform1f) X(PFNGLUNIFORM1UIPROC, glUniform1ui) X(PFNGLUNIFORM4FPROC, glUniform4f) X(PFNGLUNIFORMMATRIX4FVPROC, glUniformMatrix4fv) \
    X(PFNGLTRANSFORMFEEDBACKVARYINGSPROC, glTransformFeedbackVaryings) X(PFNGLGENBUFFERSPROC, glGenBuffers) X(PFNGLBINDBUFFERPROC, glBindBuffer) \
    X(PFNGLBUFFERDATAPROC, glBufferData) X(PFNGLBUFFERSTORAGEPROC, glBufferStorage) X(PFNGLBUFFERSUBDATAPROC, glBufferSubData) \
    X(PFNGLGETBUFFERSUBDATAPROC, glGetBufferSubData) X(PFNGLBINDBUFFERBASEPROC, glBindBufferBase) X(PFNGLBINDBUFFERRANGEPROC, glBindBufferRange) \
    X(PFNGLCLEARBUFFERDATAPROC, glClearBufferData) X(PFNGLGENVERTEXARRAYSPROC, glGenVertexArrays) X(PFNGLBINDVERTEXARRAYPROC, glBindVertexArray) \
    X(PFNGLENABLEVERTEXATTRIBARRAYPROC, glEnableVertexAttribArray) X(PFNGLVERTEXATTRIBPOINTERPROC, glVertexAttribPointer) \
    X(PFNGLGENTEXTURESPROC, glGenTextures) X(PFNGLBINDTEXTUREPROC, glBindTexture) X(PFNGLTEXSTORAGE2DPROC, glTexStorage2D) \
    X(PFNGLGENFRAMEBUFFERSPROC, glGenFramebuffers) X(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer) X(PFNGLFRAMEBUFFERTEXTURE2DPROC, glFramebufferTexture2D) \
    X(PFNGLCHECKFRAMEBUFFERSTATUSPROC, glCheckFramebufferStatus) X(PFNGLVIEWPORTPROC, glViewport) X(PFNGLCLEARCOLORPROC, glClearColor) X(PFNGLCLEARPROC, glClear) \
    X(PFNGLENABLEPROC, glEnable) X(PFNGLDISABLEPROC, glDisable) X(PFNGLBLENDFUNCPROC, glBlendFunc) X(PFNGLDRAWELEMENTSINSTANCEDPROC, glDrawElementsInstanced) \
    X(PFNGLDRAWELEMENTSPROC, glDrawElements) X(PFNGLDRAWARRAYSPROC, glDrawArrays) X(PFNGLBEGINTRANSFORMFEEDBACKPROC, glBeginTransformFeedback) \
    X(PFNGLENDTRANSFORMFEEDBACKPROC, glEndTransformFeedback) X(PFNGLREADPIXELSPROC, glReadPixels) X(PFNGLFINISHPROC, glFinish) \
    X(PFNGLDISPATCHCOMPUTEPROC, glDispatchCompute) X(PFNGLMEMORYBARRIERPROC, glMemoryBarrier) X(PFNGLPIXELSTOREIPROC, glPixelStorei) \
    X(PFNGLDELETEBUFFERSPROC, glDeleteBuffers) X(PFNGLLINEWIDTHPROC, glLineWidth) X(PFNGLGETFLOATVPROC, glGetFloatv)
#define X(T, n) static T n;
GLFUNCS(X)
#undef X

static void die(const char* fmt, ...) __attribute__((format(printf, 1, 2), noreturn));
#include <cstdarg>
static void die(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); fprintf(stderr, "refgl: "); vfprintf(stderr, fmt, ap); fprintf(stderr, "\n"); va_end(ap);
    exit(2);
}
#define GLCHK(where) do { GLenum e_ = glGetError(); if (e_ != GL_NO_ERROR) die("GL error 0x%x at %s (line %d)", e_, where, __LINE__); } while (0)

static void gl_up() {
    void* glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!glapi) die("libglapi.so.0: %s", dlerror());
    const char* drvpath = getenv("GS4D_SWRAST_DRI");
    void* drv = dlopen(drvpath ? drvpath : "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!drv) die("swrast_dri.so: %s", dlerror());
    auto getext = (const __DRIextension** (*)(void))dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!getext) die("no __driDriverGetExtensions_swrast");
    const __DRIextension** exts = getext();
    const __DRIcoreExtension* core = nullptr; const __DRIswrastExtension* sw = nullptr;
    for (int i = 0; exts[i]; ++i) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension*)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) sw = (const __DRIswrastExtension*)exts[i];
    }
    if (!core || !sw || sw->base.version < 4) die("driver lacks DRI_Core / DRI_SWRast v4");
    static __DRIswrastLoaderExtension loader;
    memset(&loader, 0, sizeof loader);
    loader.base.name = __DRI_SWRAST_LOADER; loader.base.version = 3;
    loader.getDrawableInfo = ld_getDrawableInfo; loader.putImage = ld_putImage; loader.getImage = ld_getImage;
    loader.putImage2 = ld_putImage2; loader.getImage2 = ld_getImage2;
    static const __DRIextension* loader_exts[] = { &loader.base, nullptr };
    const __DRIconfig** configs = nullptr;
    __DRIscreen* scr = sw->createNewScreen2(0, loader_exts, exts, &configs, nullptr);
    if (!scr || !configs || !configs[0]) die("createNewScreen2 failed");
    uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 4 };   // "#version 440 core"
    unsigned err = 0;
    __DRIcontext* ctx = sw->createContextAttribs(scr, __DRI_API_OPENGL_CORE, configs[0], nullptr, 2, attribs, &err, nullptr);
    if (!ctx) die("createContextAttribs failed (%u)", err);
    __DRIdrawable* dr = sw->createNewDrawable(scr, configs[0], nullptr);
    if (!dr || !core->bindContext(ctx, dr, dr)) die("bindContext failed");
    auto gp = (void* (*)(const char*))dlsym(glapi, "_glapi_get_proc_address");
    if (!gp) die("no _glapi_get_proc_address");
#define X(T, n) n = (T)gp(#n); if (!n) die("no GL entry point %s", #n);
    GLFUNCS(X)
#undef X
}

// ---------------------------------------------------------------------------------------------------------------------------
// files, programs
// ---------------------------------------------------------------------------------------------------------------------------
static std::string ref_root() { const char* r = getenv("GS4D_REF"); return r ? r : "/root/reference"; }

static std::vector<uint8_t> slurp(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb"); if (!f) die("cannot open %s", path.c_str());
    std::vector<uint8_t> b; uint8_t tmp[1 << 16]; size_t k;
    while ((k = fread(tmp, 1, sizeof tmp, f)) > 0) b.insert(b.end(), tmp, tmp + k);
    fclose(f); return b;
}
static void spill(const std::string& path, const void* p, size_t bytes) {
    FILE* f = fopen(path.c_str(), "wb"); if (!f) die("cannot write %s", path.c_str());
    if (fwrite(p, 1, bytes, f) != bytes) die("short write %s", path.c_str());
    fclose(f);
}

// the reference's Shader::BuildShader / radix_sort::shader::compile: glShaderSource of the whole file, glCompileShader (Shader.cpp; radix_sort.hpp:84-100)
static GLuint compile_file(GLenum type, const std::string& rel) {
    std::vector<uint8_t> src = slurp(ref_root() + "/" + rel);
    src.push_back(0);
    const char* s = (const char*)src.data();
    GLuint sh = glCreateShader(type);
    glShaderSource(sh, 1, &s, nullptr);
    glCompileShader(sh);
    GLint ok = 0; glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
    if (!ok) { char log[8192]; glGetShaderInfoLog(sh, sizeof log, nullptr, log); die("compile %s:\n%s", rel.c_str(), log); }
    return sh;
}
static GLuint link_program(std::initializer_list<GLuint> shaders, const std::vector<const char*>& tf = {}) {
    GLuint p = glCreateProgram();
    for (GLuint s : shaders) glAttachShader(p, s);
    if (!tf.empty()) glTransformFeedbackVaryings(p, (GLsizei)tf.size(), tf.data(), GL_INTERLEAVED_ATTRIBS);
    glLinkProgram(p);
    GLint ok = 0; glGetProgramiv(p, GL_LINK_STATUS, &ok);
    if (!ok) { char log[8192]; glGetProgramInfoLog(p, sizeof log, nullptr, log); die("link:\n%s", log); }
    return p;
}
static GLint uni(GLuint prog, const char* name) {
    GLint l = glGetUniformLocation(prog, name);
    if (l < 0) die("uniform %s not found", name);
    return l;
}

struct Target { GLuint fbo = 0, tex = 0; };
static Target make_target(int W, int H, GLenum internal) {
    Target t;
    glGenTextures(1, &t.tex); glBindTexture(GL_TEXTURE_2D, t.tex); glTexStorage2D(GL_TEXTURE_2D, 1, internal, W, H);
    glGenFramebuffers(1, &t.fbo); glBindFramebuffer(GL_FRAMEBUFFER, t.fbo);
    glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, t.tex, 0);
    if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) die("framebuffer incomplete");
    glViewport(0, 0, W, H);                                        // the window's viewport (GLFW default: the whole framebuffer)
    return t;
}
// per-frame state of the reference's main loop
static void frame_state(GLenum sfac, GLenum dfac) {
    glClearColor(0.1843137254901961, 0.20784313725490197, 0.25882352941176473, 1.0f);   // Application.cpp:125
    glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT);                                  // Renderer::Clear, Renderer.cpp:22-25
    glBlendFunc(sfac, dfac);                                                             // Application.cpp:150 (menu default :137-138)
    glEnable(GL_BLEND);                                                                  // :153
    glDisable(GL_DEPTH_TEST);                                                            // :154
}

// ---------------------------------------------------------------------------------------------------------------------------
// draw
// ---------------------------------------------------------------------------------------------------------------------------
static int cmd_draw(int argc, char** argv) {
    if (argc < 10) die("draw: too few arguments");
    std::string kind = argv[2];
    int W = atoi(argv[3]), H = atoi(argv[4]); long n = atol(argv[5]);
    std::vector<uint8_t> rec = slurp(argv[6]);
    std::string sortpath = argv[7];
    std::vector<uint8_t> ub = slurp(argv[8]);
    std::string out = argv[9];
    bool want_tf = false, want_img = false, want_img8 = false, want_img16 = false; GLenum sfac = GL_SRC_ALPHA, dfac = GL_ONE_MINUS_SRC_ALPHA;
    for (int i = 10; i < argc; ++i) {
        if (!strcmp(argv[i], "tf")) want_tf = true; else if (!strcmp(argv[i], "img")) want_img = true; else if (!strcmp(argv[i], "img8")) want_img8 = true;
        else if (!strcmp(argv[i], "img16")) want_img16 = true;
        else if (!strcmp(argv[i], "blend") && i + 2 < argc) { sfac = (GLenum)strtoul(argv[i + 1], nullptr, 0); dfac = (GLenum)strtoul(argv[i + 2], nullptr, 0); i += 2; }
        else die("draw: unknown flag %s", argv[i]);
    }
    if (ub.size() != 34 * 4) die("uniforms: expected 34 floats");
    const float* u = (const float*)ub.data();
    const bool mod = kind == "4dmod";      // Splat4DVertexShaderMod.GLSL: the records at binding 1, no sort index (Scenes.h:1765, 1800-1808)
    if (mod) kind = "4d";
    const size_t recsz = kind == "4d" ? 96 : kind == "3d" ? 4 * 72 : kind == "2d" ? 48 : 0;
    if (!recsz) die("draw: kind must be 4d, 4dmod, 3d or 2d");
    if (rec.size() != recsz * (size_t)n) die("records: %zu bytes for %ld records of %zu", rec.size(), n, recsz);

    const char *vs, *fs; std::vector<const char*> tfv;
    if (kind == "4d")      { vs = mod ? "Shader/Splats4D/Splat4DVertexShaderMod.GLSL" : "Shader/Splats4D/Splat4DVertexShaderInstanced.GLSL"; fs = "Shader/Splats4D/Splat4DFragShader.GLSL";
                             tfv = { "gl_Position", "oSig", "oColor", "oFragPos", "oFaulty", "oTimeOpacity" }; }          // 18 floats
    else if (kind == "3d") { vs = "Shader/Splats3D/Splat3DVertexShaderFull.GLSL"; fs = "Shader/Splats3D/Splat3DFragShaderFull.GLSL";
                             tfv = { "gl_Position", "oSig", "oColor", "oFragPos", "oFaulty" }; }                          // 17 floats
    else                   { vs = "Shader/Splats2D/Splat2DVSI.GLSL"; fs = "Shader/Splats2D/Splat2DFragShader.GLSL";
                             tfv = { "gl_Position", "oSig", "oColor", "oFragPos", "oSSPos" }; }                           // 18 floats
    const int tf_floats = kind == "3d" ? 17 : 18;

    GLuint vao; glGenVertexArrays(1, &vao); glBindVertexArray(vao);                     // Application.cpp:117-119
    // quad + its index list: Geometry.h:44-50 (QuadVerteices, QuadIdxBufferData), layout Geometry.h:62-66 / VertexArray.cpp:15-31
    static const float quad[8] = { 0.5f, 0.5f, 0.5f, -0.5f, -0.5f, -0.5f, -0.5f, 0.5f };
    static const unsigned quad_idx[6] = { 0, 2, 1, 2, 0, 3 };
    GLuint vbo, ibo; glGenBuffers(1, &vbo); glGenBuffers(1, &ibo);
    std::vector<unsigned> idx3d;
    glBindBuffer(GL_ARRAY_BUFFER, vbo);
    if (kind == "3d") {
        // Gaussians3D: vbo of Splat3DVertex {vec2, vec3, vec4, mat3}, one 4-vertex mesh per splat (Splat.h:461-473), index list
        // offset + {0,2,1,2,0,3} per splat (Splat.h:475-, Scenes.h:1630-1650)
        glBufferData(GL_ARRAY_BUFFER, (GLsizeiptr)rec.size(), rec.data(), GL_DYNAMIC_DRAW);
        const int counts[6] = { 2, 3, 4, 3, 3, 3 }; size_t off = 0;
        for (int i = 0; i < 6; ++i) { glEnableVertexAttribArray(i); glVertexAttribPointer(i, counts[i], GL_FLOAT, GL_FALSE, 72, (const void*)off); off += 4 * counts[i]; }
        idx3d.resize(6 * (size_t)n);
        for (long k = 0; k < n; ++k) for (int j = 0; j < 6; ++j) idx3d[6 * k + j] = 4 * (unsigned)k + quad_idx[j];
        glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, ibo);
        glBufferData(GL_ELEMENT_ARRAY_BUFFER, (GLsizeiptr)(idx3d.size() * 4), idx3d.data(), GL_STATIC_DRAW);
    } else {
        glBufferData(GL_ARRAY_BUFFER, sizeof quad, quad, GL_STATIC_DRAW);
        glEnableVertexAttribArray(0); glVertexAttribPointer(0, 2, GL_FLOAT, GL_FALSE, 8, nullptr);
        glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, ibo);
        glBufferData(GL_ELEMENT_ARRAY_BUFFER, sizeof quad_idx, quad_idx, GL_STATIC_DRAW);
    }
    GLCHK("geometry");

    GLuint ssbo_data = 0, ssbo_idx = 0;
    if (kind != "3d") {
        glGenBuffers(1, &ssbo_data); glBindBuffer(GL_SHADER_STORAGE_BUFFER, ssbo_data);
        glBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)rec.size(), rec.data(), GL_DYNAMIC_DRAW);        // ShareStorageBuffer.cpp:3-8
    }
    if (kind == "4d" && !mod) {
        std::vector<uint32_t> si((size_t)n);
        if (sortpath == "-") for (long i = 0; i < n; ++i) si[i] = (uint32_t)i;
        else { std::vector<uint8_t> b = slurp(sortpath); if (b.size() != 4 * (size_t)n) die("sortidx size"); memcpy(si.data(), b.data(), b.size()); }
        glGenBuffers(1, &ssbo_idx); glBindBuffer(GL_SHADER_STORAGE_BUFFER, ssbo_idx);
        glBufferStorage(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)(4 * n), nullptr, GL_DYNAMIC_STORAGE_BIT);     // Scenes.h:241-243
        glBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, (GLsizeiptr)(4 * n), si.data());                        // Scenes.h:281-282
    }
    GLCHK("buffers");

    GLuint vsh = compile_file(GL_VERTEX_SHADER, vs), fsh = compile_file(GL_FRAGMENT_SHADER, fs);

    auto bind_and_uniforms = [&](GLuint prog) {
        glUseProgram(prog);                                                                                  // Scenes.h:330
        if (kind == "4d") { glUniform1f(uni(prog, "uTime"), u[0]); glUniform1f(uni(prog, "uMinOpacity"), u[1]); }   // :331-332
        if (kind != "2d") glUniformMatrix4fv(uni(prog, "uView"), 1, GL_FALSE, u + 2);                        // :333 (2d: uView is unused => optimised out)
        glUniformMatrix4fv(uni(prog, "uProj"), 1, GL_FALSE, u + 18);                                         // :334
        if (kind == "4d" && !mod) { glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 1, ssbo_idx); glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 2, ssbo_data); }   // :336-337
        if (mod) glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 1, ssbo_data);                                   // Scenes.h:1806
        if (kind == "2d") glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 1, ssbo_data);                          // Scenes.h Gaussians2D::Render
    };
    auto issue = [&]() {
        glBindVertexArray(vao); glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, ibo);
        if (kind == "3d") glDrawElements(GL_TRIANGLES, (GLsizei)(6 * n), GL_UNSIGNED_INT, nullptr);          // Renderer.cpp:27-31
        else glDrawElementsInstanced(GL_TRIANGLES, 6, GL_UNSIGNED_INT, nullptr, (GLsizei)n);                 // Renderer.cpp:33-39
    };

    if (want_tf) {
        GLuint prog = link_program({ vsh, fsh }, tfv);
        bind_and_uniforms(prog);
        size_t bytes = (size_t)n * 6 * tf_floats * 4;
        GLuint tfb; glGenBuffers(1, &tfb); glBindBuffer(GL_TRANSFORM_FEEDBACK_BUFFER, tfb);
        std::vector<float> nanfill((size_t)n * 6 * tf_floats, NAN);                   // what a culled vertex leaves unwritten stays recognisable
        glBufferData(GL_TRANSFORM_FEEDBACK_BUFFER, (GLsizeiptr)bytes, nanfill.data(), GL_DYNAMIC_READ);
        glBindBufferBase(GL_TRANSFORM_FEEDBACK_BUFFER, 0, tfb);
        glEnable(GL_RASTERIZER_DISCARD);
        glBeginTransformFeedback(GL_TRIANGLES);
        issue();
        glEndTransformFeedback();
        glDisable(GL_RASTERIZER_DISCARD);
        glFinish(); GLCHK("transform feedback");
        std::vector<float> cap((size_t)n * 6 * tf_floats);
        glGetBufferSubData(GL_TRANSFORM_FEEDBACK_BUFFER, 0, (GLsizeiptr)bytes, cap.data());
        spill(out + ".tf.f32", cap.data(), bytes);
    }
    if (want_img || want_img8 || want_img16) {
        GLuint prog = link_program({ vsh, fsh });
        for (int pass = 0; pass < 3; ++pass) {
            if (pass == 0 && !want_img) continue;
            if (pass == 1 && !want_img8) continue;
            if (pass == 2 && !want_img16) continue;
            make_target(W, H, pass == 0 ? GL_RGBA32F : pass == 1 ? GL_RGBA8 : GL_RGBA16);
            frame_state(sfac, dfac);
            bind_and_uniforms(prog);
            issue();
            glFinish(); GLCHK("draw");
            glPixelStorei(GL_PACK_ALIGNMENT, 1);
            if (pass == 0) { std::vector<float> px((size_t)W * H * 4); glReadPixels(0, 0, W, H, GL_RGBA, GL_FLOAT, px.data()); GLCHK("read"); spill(out + ".img.f32", px.data(), px.size() * 4); }
            else if (pass == 1) { std::vector<uint8_t> px((size_t)W * H * 4); glReadPixels(0, 0, W, H, GL_RGBA, GL_UNSIGNED_BYTE, px.data()); GLCHK("read"); spill(out + ".img.u8", px.data(), px.size()); }
            else { std::vector<uint16_t> px((size_t)W * H * 4); glReadPixels(0, 0, W, H, GL_RGBA, GL_UNSIGNED_SHORT, px.data()); GLCHK("read"); spill(out + ".img.u16", px.data(), px.size() * 2); }
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// overlay lines: Renderer::DrawLine / DrawGrid / DrawAxis (Renderer.cpp:41-215): GL_LINES or GL_LINE_STRIP of vec3 vertices,
// uniforms uViewProj / uColor, glLineWidth(thickness)
// ---------------------------------------------------------------------------------------------------------------------------
static int cmd_lines(int argc, char** argv) {
    if (argc < 11 || (argc - 6) % 5) die("lines: W H viewproj.bin out_prefix (color.bin width nverts verts.bin strip)+");
    int W = atoi(argv[2]), H = atoi(argv[3]);
    std::vector<uint8_t> vp = slurp(argv[4]); std::string out = argv[5];
    if (vp.size() != 64) die("lines: viewproj must be 16 floats");
    GLuint prog = link_program({ compile_file(GL_VERTEX_SHADER, "Shader/Lines/LineVert.GLSL"), compile_file(GL_FRAGMENT_SHADER, "Shader/Lines/LineFrag.GLSL") });   // Renderer.h:32-34
    GLuint prog2d = link_program({ compile_file(GL_VERTEX_SHADER, "Shader/Lines/Line2DVert.GLSL"), compile_file(GL_FRAGMENT_SHADER, "Shader/Lines/Line2DFrag.GLSL") });   // Renderer.h:29-31
    make_target(W, H, GL_RGBA32F);
    frame_state(GL_SRC_ALPHA, GL_ONE_MINUS_SRC_ALPHA);
    for (int a = 6; a + 4 < argc; a += 5) {
        std::vector<uint8_t> col = slurp(argv[a]);
        float width = (float)atof(argv[a + 1]); long nv = atol(argv[a + 2]);
        std::vector<uint8_t> verts = slurp(argv[a + 3]); int strip = atoi(argv[a + 4]);
        const bool two_d = strip == 2;                    // strip 2: Renderer::DrawLine(vec2, vec2, color) — NDC positions, the 2D program (Renderer.cpp:168-201)
        if (col.size() != 16 || verts.size() != (size_t)nv * (two_d ? 8 : 12)) die("lines: bad input sizes");
        // Renderer::DrawLine / DrawGrid body, Renderer.cpp:41-73 / 113-160: program, two uniforms, a fresh VBO + VAO, attribute 0 = vec3, width, draw
        const float* c = (const float*)col.data();
        if (two_d) { glUseProgram(prog2d); glUniform4f(uni(prog2d, "uColor"), c[0], c[1], c[2], c[3]); }
        else {
            glUseProgram(prog);
            glUniformMatrix4fv(uni(prog, "uViewProj"), 1, GL_FALSE, (const float*)vp.data());
            glUniform4f(uni(prog, "uColor"), c[0], c[1], c[2], c[3]);
        }
        GLuint vbo; glGenBuffers(1, &vbo); glBindBuffer(GL_ARRAY_BUFFER, vbo);
        glBufferData(GL_ARRAY_BUFFER, (GLsizeiptr)verts.size(), verts.data(), GL_STATIC_DRAW);
        GLuint vao; glGenVertexArrays(1, &vao); glBindVertexArray(vao);
        glEnableVertexAttribArray(0); glVertexAttribPointer(0, two_d ? 2 : 3, GL_FLOAT, GL_FALSE, 0, nullptr);
        glLineWidth(width);
        GLenum e = glGetError();
        if (e) fprintf(stderr, "refgl lines: glLineWidth(%g) -> 0x%x\n", width, e);
        glDrawArrays(strip == 1 ? GL_LINE_STRIP : GL_LINES, 0, (GLsizei)nv);
        glDeleteBuffers(1, &vbo);
    }
    glFinish(); GLCHK("lines");
    std::vector<float> px((size_t)W * H * 4); glPixelStorei(GL_PACK_ALIGNMENT, 1);
    glReadPixels(0, 0, W, H, GL_RGBA, GL_FLOAT, px.data()); GLCHK("read");
    spill(out + ".img.f32", px.data(), px.size() * 4);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// sort: radix_sort::sorter::sort(key_buf, val_buf, arr_len), radix_sort.hpp:258-392 — the host sequence, call for call
// ---------------------------------------------------------------------------------------------------------------------------
static const GLuint THREADS_PER_BLOCK = 64, ITEMS_PER_THREAD = 4, BITSET_NUM = 4;          // radix_sort.hpp:10-13
static const GLuint BITSET_COUNT = 32 / BITSET_NUM, BITSET_SIZE = 16;                     // :13-14

static GLuint calc_thread_blocks_num(size_t arr_len) { return GLuint(ceil(float(arr_len) / float(THREADS_PER_BLOCK * ITEMS_PER_THREAD))); }   // :180-183
static GLuint round_to_power_of_2(GLuint dim) { return (GLuint)exp2(ceil(log2(dim))); }                                                     // :185-189

static int cmd_sort(int argc, char** argv) {
    if (argc != 6) die("sort: n keys.bin vals.bin out_prefix");
    size_t arr_len = (size_t)atol(argv[2]);
    std::vector<uint8_t> kb = slurp(argv[3]), vb = slurp(argv[4]); std::string out = argv[5];
    if (kb.size() != 4 * arr_len || vb.size() != 4 * arr_len) die("sort: input sizes");
    GLuint count_p = link_program({ compile_file(GL_COMPUTE_SHADER, "resources/radix_sort_count.comp.glsl") });                   // :218-254
    GLuint offs_p = link_program({ compile_file(GL_COMPUTE_SHADER, "resources/radix_sort_local_offsets.comp.glsl") });
    GLuint reorder_p = link_program({ compile_file(GL_COMPUTE_SHADER, "resources/radix_sort_reorder.comp.glsl") });
    const GLuint k_zero = 0;

    // the caller's buffers, created as the scenes create theirs (Scenes.h:241-247), then uploaded (Scenes.h:321-325)
    GLuint key_buf, val_buf;
    glGenBuffers(1, &key_buf); glBindBuffer(GL_SHADER_STORAGE_BUFFER, key_buf);
    glBufferStorage(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)(4 * arr_len), nullptr, GL_DYNAMIC_STORAGE_BIT);
    glBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, (GLsizeiptr)(4 * arr_len), kb.data());
    glGenBuffers(1, &val_buf); glBindBuffer(GL_SHADER_STORAGE_BUFFER, val_buf);
    glBufferStorage(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)(4 * arr_len), nullptr, GL_DYNAMIC_STORAGE_BIT);
    glBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, (GLsizeiptr)(4 * arr_len), vb.data());

    // resize_internal_buf, :191-216
    GLuint local_offsets_buf, glob_counts_buf, keys_scratch_buf, values_scratch_buf;
    glGenBuffers(1, &local_offsets_buf); glBindBuffer(GL_SHADER_STORAGE_BUFFER, local_offsets_buf);
    glBufferStorage(GL_SHADER_STORAGE_BUFFER, GLsizeiptr(round_to_power_of_2(calc_thread_blocks_num(arr_len)) * BITSET_SIZE * sizeof(GLuint)), nullptr, GL_DYNAMIC_STORAGE_BIT);
    glGenBuffers(1, &glob_counts_buf); glBindBuffer(GL_SHADER_STORAGE_BUFFER, glob_counts_buf);
    glBufferStorage(GL_SHADER_STORAGE_BUFFER, GLsizeiptr(BITSET_SIZE * sizeof(GLuint)), nullptr, 0);
    glGenBuffers(1, &keys_scratch_buf); glBindBuffer(GL_SHADER_STORAGE_BUFFER, keys_scratch_buf);
    glBufferStorage(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)(arr_len * sizeof(GLuint)), nullptr, 0);
    glGenBuffers(1, &values_scratch_buf); glBindBuffer(GL_SHADER_STORAGE_BUFFER, values_scratch_buf);
    glBufferStorage(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)(arr_len * sizeof(GLuint)), nullptr, 0);
    GLCHK("sort buffers");

    if (arr_len > 1) {                                                                                               // :260-262
        GLuint thread_blocks_num = calc_thread_blocks_num(arr_len);
        GLuint power_of_two_thread_blocks_num = round_to_power_of_2(thread_blocks_num);
        GLuint keys_buffers[2] = { key_buf, keys_scratch_buf };
        GLuint values_buffers[2] = { val_buf, values_scratch_buf };
        const GLuint scan_groups = GLuint(ceil(float(power_of_two_thread_blocks_num) / float(THREADS_PER_BLOCK * ITEMS_PER_THREAD)));
        for (GLuint pass = 0; pass < BITSET_COUNT; pass++) {
            glBindBuffer(GL_SHADER_STORAGE_BUFFER, glob_counts_buf);                                                 // :285-289
            glClearBufferData(GL_SHADER_STORAGE_BUFFER, GL_R32UI, GL_RED_INTEGER, GL_UNSIGNED_INT, &k_zero);
            glBindBuffer(GL_SHADER_STORAGE_BUFFER, local_offsets_buf);
            glClearBufferData(GL_SHADER_STORAGE_BUFFER, GL_R32UI, GL_RED_INTEGER, GL_UNSIGNED_INT, &k_zero);

            glUseProgram(count_p);                                                                                   // :297-311
            glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 0, keys_buffers[pass % 2]);
            glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 1, local_offsets_buf);
            glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 2, glob_counts_buf);
            glUniform1ui(uni(count_p, "u_arr_len"), (GLuint)arr_len);
            glUniform1ui(uni(count_p, "u_bitset_idx"), pass);
            glDispatchCompute(thread_blocks_num, 1, 1);
            glMemoryBarrier(GL_SHADER_STORAGE_BARRIER_BIT);

            glUseProgram(offs_p);                                                                                    // :317-361
            glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 0, local_offsets_buf);
            for (GLuint d = 0; d < GLuint(log2(power_of_two_thread_blocks_num)); d++) {
                glUniform1ui(uni(offs_p, "u_arr_len"), power_of_two_thread_blocks_num);
                glUniform1ui(uni(offs_p, "u_op"), 0);
                glUniform1ui(uni(offs_p, "u_depth"), d);
                glDispatchCompute(scan_groups, 1, 1);
                glMemoryBarrier(GL_SHADER_STORAGE_BARRIER_BIT);
            }
            glUniform1ui(uni(offs_p, "u_arr_len"), power_of_two_thread_blocks_num);
            glUniform1ui(uni(offs_p, "u_op"), 1);
            glDispatchCompute(scan_groups, 1, 1);
            glMemoryBarrier(GL_SHADER_STORAGE_BARRIER_BIT);
            for (GLint d = GLint(log2(power_of_two_thread_blocks_num)) - 1; d >= 0; d--) {
                glUniform1ui(uni(offs_p, "u_arr_len"), power_of_two_thread_blocks_num);
                glUniform1ui(uni(offs_p, "u_op"), 2);
                glUniform1ui(uni(offs_p, "u_depth"), (GLuint)d);
                glDispatchCompute(scan_groups, 1, 1);
                glMemoryBarrier(GL_SHADER_STORAGE_BARRIER_BIT);
            }

            glUseProgram(reorder_p);                                                                                 // :369-388
            glBindBufferRange(GL_SHADER_STORAGE_BUFFER, 0, keys_buffers[pass % 2], 0, (GLsizeiptr)(arr_len * sizeof(GLuint)));
            glBindBufferRange(GL_SHADER_STORAGE_BUFFER, 1, keys_buffers[(pass + 1) % 2], 0, (GLsizeiptr)(arr_len * sizeof(GLuint)));
            glBindBufferRange(GL_SHADER_STORAGE_BUFFER, 2, values_buffers[pass % 2], 0, (GLsizeiptr)(arr_len * sizeof(GLuint)));
            glBindBufferRange(GL_SHADER_STORAGE_BUFFER, 3, values_buffers[(pass + 1) % 2], 0, (GLsizeiptr)(arr_len * sizeof(GLuint)));
            glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 4, local_offsets_buf);
            glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 5, glob_counts_buf);
            glUniform1ui(uni(reorder_p, "u_write_values"), 1);
            glUniform1ui(uni(reorder_p, "u_arr_len"), (GLuint)arr_len);
            glUniform1ui(uni(reorder_p, "u_bitset_idx"), pass);
            glDispatchCompute(thread_blocks_num, 1, 1);
            glMemoryBarrier(GL_SHADER_STORAGE_BARRIER_BIT);
            GLCHK("sort pass");
        }
        glUseProgram(0);
    }
    glFinish();
    std::vector<uint32_t> ko(arr_len), vo(arr_len);
    glBindBuffer(GL_SHADER_STORAGE_BUFFER, key_buf); glGetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, (GLsizeiptr)(4 * arr_len), ko.data());
    glBindBuffer(GL_SHADER_STORAGE_BUFFER, val_buf); glGetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, (GLsizeiptr)(4 * arr_len), vo.data());
    GLCHK("sort read");
    spill(out + ".keys.u32", ko.data(), 4 * arr_len);
    spill(out + ".vals.u32", vo.data(), 4 * arr_len);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) die("usage: refgl info | draw ... | lines ... | sort ...");
    gl_up();
    std::string cmd = argv[1];
    if (cmd == "info") {
        GLint ssbo = 0, wg = 0, inv = 0; glGetIntegerv(GL_MAX_SHADER_STORAGE_BLOCK_SIZE, &ssbo); glGetIntegeri_v(GL_MAX_COMPUTE_WORK_GROUP_COUNT, 0, &wg);
        glGetIntegerv(GL_MAX_COMPUTE_WORK_GROUP_INVOCATIONS, &inv);
        GLint sub = 0; glGetIntegerv(GL_SUBPIXEL_BITS, &sub);
        printf("{\"version\": \"%s\", \"renderer\": \"%s\", \"glsl\": \"%s\", \"max_ssbo_block\": %d, \"max_wg_count\": %d, \"max_wg_invocations\": %d, \"subpixel_bits\": %d}\n",
               glGetString(GL_VERSION), glGetString(GL_RENDERER), glGetString(GL_SHADING_LANGUAGE_VERSION), ssbo, wg, inv, sub);
        return 0;
    }
    if (cmd == "draw") return cmd_draw(argc, argv);
    if (cmd == "lines") return cmd_lines(argc, argv);
    if (cmd == "sort") return cmd_sort(argc, argv);
    die("unknown command %s", cmd.c_str());
}
