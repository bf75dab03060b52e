/* gs4d.h — C ABI of libgs4d.so, the MI355X (gfx950) forward Gaussian-splat rasteriser.
 *
 * This is the drop-in boundary for ONE path of EndMy5uffering/4DGaussianSplatRendering: everything
 * from the upload of the splat SSBO to pixels.  The reference has no FFI of its own:
 * the surface it exposes to its scenes is a set of C++ classes over OpenGL (Renderer, ShareStorageBuffer,
 * Shader, radix_sort::sorter) plus six raw GL calls.  Every entry point below names the reference call
 * it stands in for (file:line relative to the reference tree); the C++ mirror of those classes that a
 * maintainer would compile Scenes.h against lives in 4dgaussiansplatrendering_amd/host/ and is a thin
 * wrapper over these functions.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function returns 0 on success or a
 * negative GS4D_E_* code (no exception crosses the ABI); gs4d_last_error() returns a description.
 * Matrices are 16 floats, column-major (m[4*c + r]), exactly what glUniformMatrix4fv(…, GL_FALSE, …)
 * receives from glm::mat4.  Buffers are named by small integers like GL buffer names; 0 is "none";
 * deleting 0 or an already-deleted name is tolerated (the reference double-deletes, Scenes.h:220-224, 291-299).
 * Calls return as soon as their work is queued; only the read-back / finish calls block.  Results are as if the calls had run one
 * after another.  Internally a context spreads consecutive FRAMES over a few HIP streams ("frame lanes", 4 by default) so that they
 * overlap on the device: a frame is everything from one gs4d_clear / gs4d_keygen / gs4d_sort_pairs that follows a draw up to and
 * including the next draw(s); buffers and the framebuffer carry their own cross-lane ordering.  A gs4d_clear starts rendering into the
 * next image of a small swap chain (one RGBA32F image per lane); gs4d_read_pixels* read the image the last clear/draw used.
 * One context = one GPU; contexts are independent (one process per GPU for multi-GPU runs).
 * The framebuffer is RGBA float32, row 0 = bottom row (OpenGL window origin).
 */
#ifndef GS4D_H
#define GS4D_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define GS4D_API __attribute__((visibility("default")))
#else
#define GS4D_API
#endif

typedef struct gs4d_ctx gs4d_ctx;
typedef uint32_t gs4d_buf;

enum {
    GS4D_OK = 0,
    GS4D_E_INVALID = -1,      /* bad argument / bad buffer name / size mismatch          */
    GS4D_E_DEVICE = -2,       /* HIP runtime error (message in gs4d_last_error)          */
    GS4D_E_UNSUPPORTED = -3,  /* state the path does not implement (e.g. other blend funcs) */
    GS4D_E_NOMEM = -4
};

/* Pipeline selection = which shader pair the reference scene would have bound (Shader::AddShaderSource paths). */
enum {
    GS4D_MODE_4D_SORTED = 0, /* Shader/Splats4D/Splat4DVertexShaderInstanced.GLSL + Splat4DFragShader.GLSL: slot 1 = sortidx[], slot 2 = SplatData[] (96 B) */
    GS4D_MODE_4D_DIRECT = 1, /* Shader/Splats4D/Splat4DVertexShaderMod.GLSL: slot 1 = SplatData[], instance id indexes it directly                         */
    GS4D_MODE_3D_FULL   = 2, /* Shader/Splats3D/Splat3DVertexShaderFull.GLSL + Splat3DFragShaderFull.GLSL: 4 x 72-B vertices per splat (gs4d_draw_quads) */
    GS4D_MODE_2D        = 3  /* Shader/Splats2D/Splat2DVSI.GLSL + Splat2DFragShader.GLSL: slot 1 = 48-B records                                         */
};

enum { GS4D_U_TIME = 0, GS4D_U_MIN_OPACITY = 1 };  /* uTime, uMinOpacity (Scenes.h:331-332) */
enum { GS4D_U_VIEW = 0, GS4D_U_PROJ = 1 };         /* uView, uProj       (Scenes.h:333-334) */

enum { GS4D_KEY_REF_INV_EUCLID = 0,  /* 1/|mean'(t) - cam|, the reference's key (Scenes.h:28-36, 314-319) */
       GS4D_KEY_VIEW_Z = 1 };        /* extra: key = 1/(-z_view) of the time-conditioned mean              */

/* glBlendFunc factors: GL enum values (Application.cpp:137-138, 150); the set is the one the reference's blend menu offers (DebugMenus.h:41-59) */
enum { GS4D_ZERO = 0, GS4D_ONE = 1, GS4D_SRC_COLOR = 0x0300, GS4D_ONE_MINUS_SRC_COLOR = 0x0301, GS4D_SRC_ALPHA = 0x0302, GS4D_ONE_MINUS_SRC_ALPHA = 0x0303,
       GS4D_DST_ALPHA = 0x0304, GS4D_ONE_MINUS_DST_ALPHA = 0x0305, GS4D_DST_COLOR = 0x0306, GS4D_ONE_MINUS_DST_COLOR = 0x0307,
       GS4D_CONSTANT_COLOR = 0x8001, GS4D_ONE_MINUS_CONSTANT_COLOR = 0x8002, GS4D_CONSTANT_ALPHA = 0x8003, GS4D_ONE_MINUS_CONSTANT_ALPHA = 0x8004 };

/* Per-stage device timings of the most recent calls, measured with HIP events on the context's stream. */
enum { GS4D_T_KEYGEN = 0, GS4D_T_SORT = 1, GS4D_T_PREPROCESS = 2, GS4D_T_BINNING = 3, GS4D_T_PAIRSORT = 4, GS4D_T_COMPOSITE = 5, GS4D_T_COUNT = 6 };

/* ---- context (stands in for the GL context + default framebuffer; Application.cpp:89-97, glViewport :69) ---- */
GS4D_API int  gs4d_create(int device, int width, int height, gs4d_ctx** out);
GS4D_API void gs4d_destroy(gs4d_ctx* ctx);
GS4D_API int  gs4d_resize(gs4d_ctx* ctx, int width, int height);                 /* glViewport / Camera::Resize            */
GS4D_API const char* gs4d_last_error(gs4d_ctx* ctx);                             /* ctx may be NULL: error of a failed create */

/* ---- buffers: glGenBuffers+glBufferData / glBufferStorage (ShareStorageBuffer.cpp:3-8, Scenes.h:241-247),
 *      glBufferSubData (ShareStorageBuffer.cpp:30-40, Scenes.h:321-325), glDeleteBuffers (ShareStorageBuffer.cpp:10-13) ---- */
GS4D_API int gs4d_buffer_create(gs4d_ctx* ctx, const void* data /* may be NULL */, size_t bytes, gs4d_buf* out);
GS4D_API int gs4d_buffer_subdata(gs4d_ctx* ctx, gs4d_buf buf, size_t offset, const void* data, size_t bytes);
GS4D_API int gs4d_buffer_read(gs4d_ctx* ctx, gs4d_buf buf, size_t offset, void* out, size_t bytes);  /* blocking; no reference counterpart (tests/tools) */
GS4D_API int gs4d_buffer_destroy(gs4d_ctx* ctx, gs4d_buf buf);
GS4D_API int gs4d_buffer_device_ptr(gs4d_ctx* ctx, gs4d_buf buf, void** dptr, size_t* bytes);        /* zero-copy interop with a caller that owns HIP memory */
/* The caller is about to overwrite `buf` through that pointer with work on its own stream (gs4d_set_stream).  Call this BEFORE queueing
 * the write, every time: (a) the caller's stream is made to wait for every kernel the library has queued that still uses the buffer
 * (without a caller stream the call blocks until they have finished), (b) the contents count as changed from here on — the library keeps
 * derived data per buffer version (a repacked copy of the records, bounds of the sort keys, what a sorted index was sorted by) and
 * otherwise goes on using it.  The library's next call that uses the buffer is ordered after the caller's stream as gs4d_set_stream says. */
GS4D_API int gs4d_buffer_invalidate(gs4d_ctx* ctx, gs4d_buf buf);
/* glBindBufferBase(GL_SHADER_STORAGE_BUFFER, slot, buf) (Scenes.h:336, ShareStorageBuffer.cpp:20-23); slots 0..7 */
GS4D_API int gs4d_bind_storage(gs4d_ctx* ctx, int slot, gs4d_buf buf);

/* ---- pipeline state: Shader::Bind / SetUniform1f / SetUniformMat4f (Shader.cpp:171-174, 206-209), glClearColor
 *      (Application.cpp:125), glBlendFunc (Application.cpp:150), Renderer::Clear (Renderer.cpp:20-23) ---- */
GS4D_API int gs4d_set_mode(gs4d_ctx* ctx, int mode);
GS4D_API int gs4d_set_uniform_1f(gs4d_ctx* ctx, int id, float v);
GS4D_API int gs4d_set_uniform_mat4(gs4d_ctx* ctx, int id, const float m[16]);
GS4D_API int gs4d_set_clear_color(gs4d_ctx* ctx, const float rgba[4]);
/* glBlendFunc(sfactor, dfactor) (Application.cpp:150): equation FUNC_ADD on all four channels, result clamped to [0, 1].  Default
 * (SRC_ALPHA, ONE_MINUS_SRC_ALPHA).  The blend colour is (0, 0, 0, 0) as in the reference (no glBlendColor): CONSTANT_* act as ZERO,
 * ONE_MINUS_CONSTANT_* as ONE.  Any other value: GS4D_E_INVALID (GL_INVALID_ENUM).  Draws with a function other than the default take
 * the instance-ordered tile lists and blend them in draw order. */
GS4D_API int gs4d_set_blend(gs4d_ctx* ctx, int src_factor, int dst_factor);
GS4D_API int gs4d_clear(gs4d_ctx* ctx);

/* ---- ordering ---- */
/* radix_sort::sorter::sort(key_buf, val_buf, n) (radix_sort.hpp:258-392): stable ascending sort of n (uint32 key, uint32 value)
 * pairs, in place.  Output == std::stable_sort by key; n <= 1 is a no-op (radix_sort.hpp:260). */
GS4D_API int gs4d_sort_pairs(gs4d_ctx* ctx, gs4d_buf keys, gs4d_buf vals, size_t n);
/* GPU replacement for the CPU key loop + two glBufferSubData uploads (Scenes.h:314-325): writes keys_f32[i] and idx_u32[i] = i
 * for the n 96-byte SplatData records in `data`. */
GS4D_API int gs4d_keygen(gs4d_ctx* ctx, gs4d_buf data, float t, const float cam_pos[3], gs4d_buf keys_f32, gs4d_buf idx_u32, size_t n, int key_mode);

/* ---- draw: Renderer::Draw(va, ib, instances) -> glDrawElementsInstanced(GL_TRIANGLES, 6, …, instances) (Renderer.cpp:33-39)
 *      with the state set above; blends `instances` quads into the framebuffer in instance order. ---- */
GS4D_API int gs4d_draw_instanced(gs4d_ctx* ctx, size_t instances);
/* Renderer::Draw(va, ib) -> glDrawElements on 4 vertices x 72 B per splat (Scenes.h:1690-1692): GS4D_MODE_3D_FULL */
GS4D_API int gs4d_draw_quads(gs4d_ctx* ctx, gs4d_buf vertices, size_t nquads);

/* Overlay lines: Renderer::DrawLine / DrawGrid / DrawAxis (Renderer.cpp:41-215) with the flat-colour programs under Shader/Lines.
 * nverts positions of `dims` floats each: dims == 3: gl_Position = viewproj * vec4(p, 1) (LineVert.GLSL:11); dims == 2: gl_Position =
 * vec4(p, 0, 1), i.e. NDC (Line2DVert.GLSL:11; viewproj may be NULL).  strip == 0: GL_LINES (vertices 2k, 2k+1), else GL_LINE_STRIP.
 * Segments are clipped to the view volume and rasterised as non-antialiased lines of `width` pixels (rounded, at least 1) — GL 4.4
 * section 14.5.2 — and every fragment is blended into the current image with the context's blend function, after what was drawn
 * before and before what is drawn next.  All fragments of one call carry the same colour. */
GS4D_API int gs4d_draw_lines(gs4d_ctx* ctx, const float* verts, size_t nverts, int dims, int strip, const float viewproj[16], const float rgba[4], float width);

/* ---- read-back (no reference counterpart: the reference never reads its framebuffer) ---- */
GS4D_API int gs4d_read_pixels(gs4d_ctx* ctx, float* rgba, size_t bytes);          /* blocking; bytes == width*height*16     */
GS4D_API int gs4d_read_pixels_device(gs4d_ctx* ctx, void* dptr, size_t bytes);    /* device-to-device, asynchronous: ordered into the caller's stream if one was given (gs4d_set_stream), else call gs4d_finish before using dptr */
/* Presentation format of the reference's window framebuffer (RGBA8 unorm, Application.cpp:89): clamp to [0,1], round to nearest.
 * Device-to-device, asynchronous like gs4d_read_pixels_device, width*height*4 bytes. */
GS4D_API int gs4d_read_pixels_rgba8_device(gs4d_ctx* ctx, void* dptr, size_t bytes);
/* The same for an image of the swap chain: frames_back 0 = the current image, 1 = the image the last gs4d_clear moved away from (the
 * previous frame), which stays intact until its lane comes round again.  An application that reads frame f-1 after queueing frame f
 * (clear, keygen, sort, draw) never waits for frame f: its host thread stays ahead of the device.  GS4D_E_INVALID when there is no such
 * image (frames_back > 1, one frame lane, or no gs4d_clear yet). */
GS4D_API int gs4d_read_frame_rgba8_device(gs4d_ctx* ctx, int frames_back, void* dptr, size_t bytes);
/* The same, for callers that cycle through several destination buffers (a double-buffered gather batch): the pack does NOT wait for
 * everything queued on the caller's stream — which would include the transfer still reading the OTHER buffer — but only for `hip_event`
 * (a hipEvent_t passed as void*, recorded by the caller behind the last work that used `dptr`; NULL: `dptr` is free, wait for nothing).
 * As with the call above, work the caller queues on its stream afterwards sees the pixels. */
GS4D_API int gs4d_read_frame_rgba8_device_after(gs4d_ctx* ctx, int frames_back, void* dptr, size_t bytes, void* hip_event);
/* Name the caller's HIP stream (hipStream_t passed as void*; NULL: none — note that the legacy default stream's handle IS NULL: give a
 * stream of your own).  The library keeps running on its own streams, but from now on
 *  (a) a buffer the caller rewrites on that stream (gs4d_buffer_device_ptr + gs4d_buffer_invalidate, in that call order, THEN the
 *      caller's writes) is not used by the library before those writes are done: the first call that uses the buffer after
 *      gs4d_buffer_invalidate records an event on the caller's stream and every frame lane waits for it before touching the buffer;
 *  (b) a device read-back waits for what the caller queued before it (the destination may still be read by, e.g., the RCCL send of the
 *      previous batch), and whatever the caller queues after it sees the pixels — e.g. an RCCL gather of the frames.
 * No host synchronisation.  Calls that hand nothing over (clear, uniforms, keygen / sort / draw on buffers the caller did not announce)
 * do not look at the caller's stream: frames keep overlapping across the lanes. */
GS4D_API int gs4d_set_stream(gs4d_ctx* ctx, void* hip_stream);
/* Single-frame sharding over several GPUs (secondary mode; config 5): rows of 8x8-pixel tiles are dealt round-robin,
 * tile row ty belongs to rank ty % world.  After gs4d_set_tile_shard(rank, world) a draw bins and composites only the context's own
 * tile rows (every rank still generates keys and sorts all splats: the blend order is global); the other rows of its image keep the
 * clear colour.  gs4d_read_band_rgba8_device packs the context's rows — band_rows pixel rows, its first tile row on top (bottom-up like
 * the framebuffer) — for a gather; bytes == band_rows * width * 4.  rank 0 / world 1 restores the default. */
GS4D_API int gs4d_set_tile_shard(gs4d_ctx* ctx, int rank, int world);
GS4D_API int gs4d_band_rows(gs4d_ctx* ctx, int* rows);
GS4D_API int gs4d_read_band_rgba8_device(gs4d_ctx* ctx, void* dptr, size_t bytes);
GS4D_API int gs4d_finish(gs4d_ctx* ctx);                                          /* blocks until every lane is idle; reports device-side check failures */

/* ---- measurement / test hooks ---- */
GS4D_API int gs4d_set_profiling(gs4d_ctx* ctx, int stage_mask);                   /* bit (1 << GS4D_T_x) times stage x; 0 = off, 0x3F = every stage; bits 8..15 = k: time only every k-th frame (0 = every frame).
                                                                                      Each timed stage costs two event records in a timed frame (they break back-to-back kernel dispatch: ~2 us each on the device) */
GS4D_API int gs4d_get_timings(gs4d_ctx* ctx, float ms[GS4D_T_COUNT]);             /* blocking; -1.0f for stages that did not run */
/* Start and end of every timed stage of the frames recorded so far (at most 128), in ms since the first timed stage of frame 0:
 * ms[frame][stage][2].  Shows how consecutive frames overlap.  Blocking; does not restart the ring (gs4d_get_timings does). */
GS4D_API int gs4d_get_timeline(gs4d_ctx* ctx, float* ms, int max_frames, int* frames);
GS4D_API int gs4d_get_stats(gs4d_ctx* ctx, uint64_t stats[8]);                    /* [0] low 32 bits: tile-list entries of the last draw, high 32 bits: draws so far whose projection kernel wrote the list entries itself (staged lists: DESIGN.md 3c), [1] low 40 bits: capacity, high 24 bits: staged draws whose guess did not fit and that were re-run exactly, [2] low 32 bits: re-runs after overflow, high 32 bits: draws that aborted on the device and were cleared away unobserved (never re-run; a frame loop without read-backs checks this stays 0), [3] low 32 bits: tiles, bits 32-39: bytes per record the last 4D draw's projection read (64: static 3D splats, 72: symmetric sig, 96: anything), bits 40-63: tiles the compositing kernel of the last unordered draw was launched for (a staged draw: the box of tiles that held entries in the frames before, a few tiles wider — an entry outside it is found on the device and the draw re-run exactly, counted with the staged misses),
                                                                                      [4] low 32 bits: radix passes launched by the last gs4d_sort_pairs, high 32 bits: candidate streams gs4d_create discarded because they shared a hardware queue with a frame lane chosen before them (0 in a process without other streams), [5] low 32 bits: by the last draw's tile sort (0: the draw built unordered tile lists), high 32 bits: gs4d_keygen calls that gave their output buffers fresh storage instead of waiting for another frame lane (one key / index pair shared by all frames),
                                                                                      [6] bits 0..15: frame lanes, bits 16..31: lanes whose stream shares a hardware queue with another lane's (0 unless the process has fewer free queues than lanes: such a context runs ~10 % slower), high 32 bits: draws that generated the depth keys of the preceding gs4d_keygen themselves (see gs4d_keygen), [7] low 32 bits: draws so far on the unordered tile-list path, high 32 bits: longest tile list of the last such draw */
/* Projected records of the last draw, 16 floats per record in record order:
 * cx, cy, a0x, a0y, a1x, a1y, alpha, r, g, b, tile-rect (2 words, bit patterns), hx, hy, valid(1/0), 0 */
GS4D_API int gs4d_debug_read_projected(gs4d_ctx* ctx, float* out16, size_t nrecords);

/* ---- host-side parameterisation (CPU code inside libgs4d.so; mirrors the reference's host math so that a caller
 *      without GLM can build SSBO contents).  Quaternions are w,x,y,z (GLM 0.9.9.9 order). ---- */
GS4D_API void gs4d_host_look_at(const float eye[3], const float orientation[3], const float up[3], float view[16]);            /* Camera.cpp:50-53 */
GS4D_API void gs4d_host_perspective(float fov_deg, int width, int height, float znear, float zfar, float proj[16]);             /* Camera.cpp:55-58 */
GS4D_API void gs4d_host_quat_look_at(const float dir[3], const float up[3], float q_wxyz[4]);                                   /* Scenes.h:268     */
GS4D_API void gs4d_host_splat3d_cov(const float q_wxyz[4], const float scale[3], float cov9[9]);                                /* Splat.h:334-344  */
GS4D_API void gs4d_host_splat3d_mesh(const float pos3[3], const float q_wxyz[4], const float scale[3], const float color4[4], float verts72[72]); /* Splat.h:433-473, Geometry.h:37-50: the 4 x 72-byte vertices gs4d_draw_quads takes */
GS4D_API void gs4d_host_splat2d_sigma_inv(const float v0[2], float l0, float l1, float sigma_inv4[4]);                             /* Splat.h:551-582  */
GS4D_API void gs4d_host_gaussians2d_record(float angle, float s0, float s1, float px, float py, const float rgb[3], float rec12[12]); /* Scenes.h:1490-1496: one 48-byte GS4D_MODE_2D record */
GS4D_API void gs4d_host_splat4d_cov(const float q_wxyz[4], const float scale[3], float lifetime, float fade, const float dir[3], float cov16[16]); /* Splat.h:132-159 */
GS4D_API void gs4d_host_splat4d_cov2q(const float q0_wxyz[4], const float q1_wxyz[4], const float scale4[4], float cov16[16]);   /* Splat.h:91-130   */
/* Batch builders: n splats -> n 96-byte SplatData records (Scenes.h:22-37 layout).
 * static 3D embedding (Scenes.h:2487 ObjectDisplay): Sigma3 in the upper 3x3, Sigma[i][3]=Sigma[3][i]=0, Sigma44=1, mu_t=0. */
GS4D_API void gs4d_host_build_records_3d(size_t n, const float* pos3, const float* q_wxyz, const float* scale3, const float* rgba, float* records24);
GS4D_API void gs4d_host_build_records_4d(size_t n, const float* pos4, const float* q_wxyz, const float* scale3, const float* lifetime, const float* fade,
                                const float* dir3, const float* rgba, float* records24);

/* Scene generators and the .vdata loader (f2): the CPU loops that fill the SSBO before the path starts. */
GS4D_API void gs4d_host_scene_linear(size_t nverts, const float* verts6, int steps, float time_multiplier, float object_scale, const float splat_scale[3],
                                     float lifetime, float fade, float speed, float* records24);                      /* Scenes.h:258-279, defaults :186-201 */
GS4D_API void gs4d_host_scene_nonlinear(size_t nverts, const float* verts6, int steps, float angle_multiplier, float radius, float object_scale,
                                        const float splat_scale[3], float lifetime, float fade, float speed, size_t max_records, float* records24); /* Scenes.h:517-545, defaults :451-467 */
GS4D_API void gs4d_host_scene_rotation(size_t nverts, const float* verts6, int steps, float angle_multiplier, float object_scale, const float splat_scale[3],
                                       float lifetime, float fade, float speed, size_t max_records, float* records24);  /* Scenes.h:775-803, defaults :711-727 */
GS4D_API void gs4d_host_scene_combined(size_t nverts, const float* verts6, int steps, float angle_multiplier, float lin_multiplier, float amplitude, float frequency,
                                       float object_scale, const float splat_scale[3], float lifetime, float fade, float speed, size_t max_records,
                                       float* records24);                                                               /* Scenes.h:1035-1068, defaults :959-976 */
GS4D_API void gs4d_host_scene_broken(size_t nverts, const float* verts6, int steps, float object_scale, const float splat_scale[3],
                                     float lifetime, float fade, float speed, size_t max_records, float* records24);    /* Scenes.h:1965-1989, defaults :1899-1912 */
GS4D_API void gs4d_host_scene_square(size_t nverts, const float* verts6, int steps, float square_size, float object_scale, const float splat_scale[3],
                                     float lifetime, float fade, float speed, size_t max_records, float* records24);    /* Scenes.h:2216-2259, defaults :2151-2165 */
GS4D_API long gs4d_host_parse_vdata(const char* path, float* verts6, size_t cap_vertices);                            /* VDataParser.h:25-58 */
/* .sd splat files (23 numbers per splat) -> 96-byte records as ObjectDisplay::init builds them; returns the splat count or -1 */
GS4D_API long gs4d_host_parse_sd(const char* path, float object_scale, float* records24, size_t cap_records);           /* VDataParser.h:60-123, Scenes.h:2483-2491 */

/* Camera input model: Camera::HandleInput / HandleCamRotation / SetIsViewFixedOnPoint / GetViewport / GetFocal
 * (Camera.cpp:90-99, 116-220) as a pure state machine — no window: the caller says which keys are down and where the cursor is. */
enum { GS4D_CAMKEY_W = 1, GS4D_CAMKEY_S = 2, GS4D_CAMKEY_A = 4, GS4D_CAMKEY_D = 8, GS4D_CAMKEY_E = 16, GS4D_CAMKEY_Q = 32, GS4D_CAMKEY_SPACE = 64,
       GS4D_CAMKEY_LCTRL = 128, GS4D_CAMKEY_LSHIFT = 256, GS4D_CAMKEY_C = 512, GS4D_CAMKEY_ESC = 1024 };
typedef struct gs4d_camera_state {
    float position[3], orientation[3], up[3];
    int width, height;
    float sensitivity, speed, fast_speed;                 /* Camera.h:78-80: 100, 0.5, 2 */
    int capture_mouse, first_capture, fix_view, fix_position, lock_x, lock_y;
} gs4d_camera_state;
typedef struct gs4d_camera_input { unsigned keys; double mouse_x, mouse_y; int imgui_active; } gs4d_camera_input;
/* One HandleInput call.  *recenter_cursor != 0: the reference moved the cursor to the window centre (glfwSetCursorPos) during the call;
 * *hide_cursor != 0: it hid the cursor (glfwSetInputMode).  When C captures the mouse in this very call the rotation that follows reads
 * the re-centred cursor, as the reference does. */
GS4D_API void gs4d_host_camera_input(gs4d_camera_state* st, const gs4d_camera_input* in, int* recenter_cursor, int* hide_cursor);
GS4D_API void gs4d_host_camera_rotate(gs4d_camera_state* st, double mouse_x, double mouse_y);            /* Camera.cpp:191-207 */
GS4D_API void gs4d_host_camera_look_at_point(gs4d_camera_state* st, const float point[3]);              /* Camera.cpp:209-220 */
GS4D_API void gs4d_host_camera_viewport(int width, int height, float out2[2]);                           /* Camera.cpp:90-93  */
GS4D_API void gs4d_host_camera_focal(float fov, int width, int height, float out2[2]);                   /* Camera.cpp:95-99  */

/* Presentation: an RGBA8 frame as produced by gs4d_read_pixels_rgba8_device (bottom row first) -> PNG file */
GS4D_API int gs4d_host_write_png(const char* path, const uint8_t* rgba8, int width, int height);

GS4D_API const char* gs4d_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GS4D_H */
