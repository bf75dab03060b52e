// gs4d_api.hip — the C ABI (include/gs4d.h): context, buffer objects, pipeline state and the draw sequence.
//
// Stands in for the OpenGL objects and calls the reference's scenes use on this path:
//   ShareStorageBuffer ctor/SubData/Bind                    4DSplatRendering/ShareStorageBuffer.cpp:3-40
//   glGenBuffers/glBufferStorage/glBufferSubData/glBindBufferBase/glDeleteBuffers in the scenes   Scenes.h:241-247, 282-283, 321-325, 336, 220-224
//   Shader::Bind/SetUniform1f/SetUniformMat4f               4DSplatRendering/Shader.cpp:171-174, 206-209
//   radix_sort::sorter::sort                                Dependencies/GPU_RADIX_SORT/radix_sort.hpp:258-392
//   Renderer::Clear / Renderer::Draw                        4DSplatRendering/Renderer.cpp:20-39
// One HIP stream per context; API calls enqueue work in order and return; only read-back/finish calls block, plus the
// validation of a draw's tile-list capacity, which is deferred to the next call that could observe it (see resolve_pending).
#include "gs4d_internal.h"
#include <cstdio>
#include <chrono>
#include <cstring>
#include <new>
#include <algorithm>
#include <cmath>

using namespace gs4d;

namespace {

struct Buffer {
    void* d = nullptr;
    size_t bytes = 0;
    uint64_t version = 0;          // bumped by every write; SoA shadow and pending-draw tracking compare against it
    float4* soa = nullptr;         // lazily built SoA shadow of 96-B SplatData records
    size_t soa_n = 0;
    uint64_t soa_version = ~0ull;
    uint32_t* bbox_dev = nullptr;  // 16 words: bounding box of pos / mu_t / velocity, reduced by the repack kernel
    double bb_lo[7] = { 0 }, bb_hi[7] = { 0 }; bool bb_ok = false;
    uint64_t order_seq = 0;        // last draw that reads its sort index from this buffer (its binning kernel does, on the raster stream)
    uint64_t data_seq = 0;         // last draw that reads its records from this buffer (the SoA repack / projection kernels do)
    bool alive = false;
};

struct DrawArgs {
    int mode = 0;
    Uniforms u;
    gs4d_buf data = 0, order = 0;
    uint64_t data_version = 0, order_version = 0;
    size_t instances = 0;
    bool quads = false;
    bool fb_was_clear = false;     // framebuffer state the composite of this draw must start from (kept for a re-run)
    int pre_idx = 0;               // which projected-record buffer the draw uses
    uint64_t seq = 0;              // draw number (1, 2, ...): indexes the ring of binning-done events
};

thread_local std::string g_create_error;

} // namespace

struct gs4d_ctx {
    int device = 0;
    int W = 0, H = 0, tiles_x = 0, tiles_y = 0;
    hipStream_t st = nullptr;          // stream in use (own_st unless the caller supplied one)
    hipStream_t own_st = nullptr;
    bool single_stream = false;
    // Two streams.  `st` (the caller-visible one) carries the ORDER stage of a frame: SoA refresh, key generation, depth sort.
    // `rs` carries the RASTER stage: preprocess, tile binning, tile sort, composite, read-back.  The raster stage of frame f and the
    // order stage of frame f+1 touch disjoint data (the draw keeps a private copy of the sort index it was given), so consecutive
    // frames overlap on the GPU; the events below carry the few true dependencies.
    hipStream_t rs = nullptr;
    hipStream_t ps = nullptr;             // preprocess stream: the projection of frame f+1 runs beside both stages (double-buffered outputs)
    hipEvent_t ev_pre_done = nullptr;     // ps: projected records written                  -> rs waits before binning
    hipEvent_t ev_raster_done[2] = { nullptr, nullptr };   // rs: the composite that read proj[i] has finished -> ps waits before overwriting proj[i]
    hipEvent_t ev_soa = nullptr;          // st: SoA shadow rebuilt                          -> rs waits before preprocess
    hipEvent_t ev_order_ready = nullptr;  // st: everything queued before the draw call      -> rs waits before binning reads the sort index
    static constexpr int EMIT_RING = 8;
    hipEvent_t ev_emit[EMIT_RING] = { nullptr };   // rs: draw `seq`'s binning has read (and copied) the sort index -> st waits on slot seq % 8 before overwriting that buffer
    uint64_t draw_seq = 0;                // draws enqueued so far
    uint64_t done_seq = 0;                // draws known to have finished entirely (set by sync_all)
    uint64_t emit_known = 0;              // draws whose binning kernel is known to have finished (set by resolve_pending)
    hipEvent_t ev_readback = nullptr;     // rs: device-side read-back enqueued               -> st waits so the caller's stream sees it
    uint32_t* order_copy = nullptr; size_t order_cap = 0;   // private copy of the last draw's sort index (for a re-run after overflow)
    std::string err;
    std::vector<Buffer> bufs;          // index = name; bufs[0] unused
    gs4d_buf slots[8] = { 0 };
    int mode = GS4D_MODE_4D_SORTED;
    Uniforms u;
    float clear[4] = { 0.0f, 0.0f, 0.0f, 0.0f };     // GL's initial clear colour; the app sets its own (Application.cpp:125)
    float4* fb = nullptr;
    bool fb_is_clear = true;           // framebuffer content == clear colour, not yet materialised
    // per-draw scratch
    float4* proj2[2] = { nullptr, nullptr }; size_t proj_cap = 0; size_t proj_n = 0;   // projected records, double-buffered across draws
    uint2* rects2[2] = { nullptr, nullptr };   // compact pixel rectangles, one per projected record
    int pre_idx = 0;                   // buffer the latest draw projected into
    uint32_t* pair_keys = nullptr; uint32_t* pair_vals = nullptr; size_t pair_cap = 0;
    SortScratch depth_sort, pair_sort;
    BinScratch bin;
    uint32_t* host_total = nullptr;    // pinned + mapped: [0..3] the binning total of the last draw, [4] the error word kernels raise when a bounded spin times out
    uint32_t* host_total_dev = nullptr; // the same memory as the device sees it
    uint32_t* dev_err = nullptr;       // = host_total_dev + 4
    gs4d_buf kg_buf = 0; uint64_t kg_ver = 0; size_t kg_n = 0;   // key buffer whose digit histograms k_keygen left for the next sort
    bool pending = false;
    DrawArgs pending_args;
    uint64_t stat_entries = 0, stat_reruns = 0;
    // profiling
    // profiling: a ring of per-frame event pairs; a frame ends with its draw
    static constexpr int PROF_FRAMES = 128;
    unsigned profiling = 0;                    // bit s set: stage s is timed
    int prof_frame = 0;
    std::vector<hipEvent_t> ev0, ev1;          // [PROF_FRAMES][GS4D_T_COUNT], created on first use
    std::vector<uint8_t> ran;
};

namespace {

int fail(gs4d_ctx* c, int code, const char* msg) { if (c) c->err = msg; else g_create_error = msg; return code; }
int hipfail(gs4d_ctx* c, hipError_t e, const char* where) {
    char b[256]; snprintf(b, sizeof b, "%s: %s", where, hipGetErrorString(e));
    if (c) c->err = b; else g_create_error = b;
    return e == hipErrorOutOfMemory ? GS4D_E_NOMEM : GS4D_E_DEVICE;
}
#define HIPCHK(c, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return hipfail((c), e__, #call); } while (0)

Buffer* getbuf(gs4d_ctx* c, gs4d_buf b) { return (b != 0 && b < c->bufs.size() && c->bufs[b].alive) ? &c->bufs[b] : nullptr; }

struct StageTimer {
    gs4d_ctx* c; int slot; hipStream_t s;
    StageTimer(gs4d_ctx* c_, int id, hipStream_t stream = nullptr) : c(c_), slot(-1), s(stream ? stream : c_->st) {
        if (((c->profiling >> id) & 1u) && c->prof_frame < gs4d_ctx::PROF_FRAMES) { slot = c->prof_frame * GS4D_T_COUNT + id; (void)hipEventRecord(c->ev0[slot], s); }
    }
    ~StageTimer() { if (slot >= 0) { (void)hipEventRecord(c->ev1[slot], s); c->ran[slot] = 1; } }
};

int sync_all(gs4d_ctx* c) {
    HIPCHK(c, hipStreamSynchronize(c->st));
    HIPCHK(c, hipStreamSynchronize(c->ps));
    HIPCHK(c, hipStreamSynchronize(c->rs));
    c->done_seq = c->draw_seq;
    return GS4D_OK;
}

int ensure_soa(gs4d_ctx* c, Buffer& b) {
    const size_t n = b.bytes / 96;
    if (b.soa && b.soa_n == n && b.soa_version == b.version) return GS4D_OK;
    if (b.data_seq > c->done_seq) { int rc = sync_all(c); if (rc) return rc; }      // a running draw still projects from the old shadow
    if (!b.soa || b.soa_n != n) {
        if (b.soa) { int rc = sync_all(c); if (rc) return rc; (void)hipFree(b.soa); b.soa = nullptr; }
        if (n) HIPCHK(c, hipMalloc(&b.soa, n * 96));
        b.soa_n = n;
    }
    // bounding box of everything the sort key depends on (upload-time work: one small read-back per refresh)
    uint32_t init[16]; for (int i = 0; i < 16; ++i) init[i] = i < 7 ? 0xFFFFFFFFu : 0u;
    if (!b.bbox_dev) HIPCHK(c, hipMalloc(&b.bbox_dev, 64));
    HIPCHK(c, hipMemcpyAsync(b.bbox_dev, init, 64, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, launch_soa_repack(c->st, (const float*)b.d, n, b.soa, b.bbox_dev));
    HIPCHK(c, hipEventRecord(c->ev_soa, c->st));
    uint32_t got[16];
    HIPCHK(c, hipMemcpyAsync(got, b.bbox_dev, 64, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    auto ord2f = [](uint32_t u) { u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; float f; memcpy(&f, &u, 4); return (double)f; };
    b.bb_ok = n > 0 && got[14] == 0;
    for (int k = 0; k < 7; ++k) { b.bb_lo[k] = ord2f(got[k]); b.bb_hi[k] = ord2f(got[7 + k]); if (!(b.bb_lo[k] <= b.bb_hi[k])) b.bb_ok = false; }
    b.soa_version = b.version;
    return GS4D_OK;
}

int ensure_pairs(gs4d_ctx* c, size_t cap) {
    if (c->pair_cap >= cap) return GS4D_OK;
    { int rc = sync_all(c); if (rc) return rc; }
    if (c->pair_keys) (void)hipFree(c->pair_keys);
    if (c->pair_vals) (void)hipFree(c->pair_vals);
    c->pair_keys = c->pair_vals = nullptr; c->pair_cap = 0;
    HIPCHK(c, hipMalloc(&c->pair_keys, cap * 4));
    HIPCHK(c, hipMalloc(&c->pair_vals, cap * 4));
    c->pair_cap = cap;
    return GS4D_OK;
}

// Enqueue binning -> tile sort -> ranges -> composite on the raster stream for projected records already in c->proj.
int enqueue_raster(gs4d_ctx* c, const uint32_t* order, uint32_t* order_copy, size_t ninst, size_t nrecords, int premult_c, bool fb_was_clear, int pre_idx, uint64_t seq) {
    const size_t ntiles = (size_t)c->tiles_x * c->tiles_y;
    int tile_bits = 1; while (((size_t)1 << tile_bits) < ntiles) ++tile_bits;
    const int tile_passes = (tile_bits + 7) / 8 < 2 ? 2 : (tile_bits + 7) / 8;
    {
        StageTimer t(c, GS4D_T_BINNING, c->rs);
        hipError_t he = hipSuccess;
        uint32_t* ph = sort_hist_slot(c->rs, c->pair_sort, c->pair_cap, &he);      // the emit kernel also counts the tile-id digits
        if (!ph) return hipfail(c, he, "sort_hist_slot");
        HIPCHK(c, launch_binning(c->rs, c->bin, c->rects2[pre_idx], order, order_copy, ninst, nrecords, c->tiles_x, c->tiles_y, c->pair_keys, c->pair_vals, c->pair_cap, c->dev_err,
                                 ph, tile_passes, c->host_total_dev));
    }
    HIPCHK(c, hipEventRecord(c->ev_emit[seq % gs4d_ctx::EMIT_RING], c->rs));     // the last binning workgroup wrote the total straight into pinned host memory
    {
        StageTimer t(c, GS4D_T_PAIRSORT, c->rs);
        HIPCHK(c, radix_sort_pairs(c->rs, c->pair_sort, c->pair_keys, c->pair_vals, c->pair_cap, c->bin.total, tile_bits, true));
        HIPCHK(c, launch_tile_ranges(c->rs, c->bin, c->pair_keys, c->pair_cap, ntiles));
    }
    {
        StageTimer t(c, GS4D_T_COMPOSITE, c->rs);
        HIPCHK(c, launch_composite(c->rs, c->proj2[pre_idx], c->pair_vals, c->bin.ranges, c->bin.total, c->tiles_x, c->tiles_y, c->W, c->H, premult_c, fb_was_clear ? 1 : 0, c->clear, c->fb));
    }
    HIPCHK(c, hipEventRecord(c->ev_raster_done[pre_idx], c->rs));
    return GS4D_OK;
}

int run_draw(gs4d_ctx* c, const DrawArgs& a, bool preprocess) {
    Buffer* data = getbuf(c, a.data);
    if (!data) return fail(c, GS4D_E_INVALID, "draw: no splat data buffer bound");
    const uint32_t* order = nullptr;
    size_t nrec = 0, npre = 0;
    int premult = 0;
    if (a.quads) { nrec = data->bytes / 288; npre = nrec < a.instances ? nrec : a.instances; premult = 1; }
    else if (a.mode == GS4D_MODE_4D_SORTED) {
        Buffer* ob = getbuf(c, a.order);
        if (!ob) return fail(c, GS4D_E_INVALID, "draw: GS4D_MODE_4D_SORTED needs the sort-index buffer at slot 1");
        if (ob->bytes < a.instances * 4) return fail(c, GS4D_E_INVALID, "draw: sort-index buffer smaller than the instance count");
        order = (const uint32_t*)ob->d;
        nrec = data->bytes / 96; npre = nrec;
    } else if (a.mode == GS4D_MODE_4D_DIRECT) { nrec = data->bytes / 96; npre = nrec < a.instances ? nrec : a.instances; }
    else if (a.mode == GS4D_MODE_2D) { nrec = data->bytes / 48; npre = nrec < a.instances ? nrec : a.instances; }
    else return fail(c, GS4D_E_INVALID, "draw: mode does not match the draw call");
    if (a.instances == 0 || nrec == 0) return GS4D_OK;
    if (a.instances >= 0xFFFFFFFFull || nrec >= 0xFFFFFFFFull) return fail(c, GS4D_E_UNSUPPORTED, "draw: more than 2^32-1 instances");

    HIPCHK(c, bin_scratch_reserve(c->rs, c->bin, a.instances, (size_t)c->tiles_x * c->tiles_y));
    uint32_t* order_copy = nullptr;
    if (order) {
        if (c->order_cap < a.instances) {
            int rc = sync_all(c); if (rc) return rc;
            if (c->order_copy) (void)hipFree(c->order_copy);
            c->order_copy = nullptr; c->order_cap = 0;
            HIPCHK(c, hipMalloc(&c->order_copy, a.instances * 4));
            c->order_cap = a.instances;
        }
        if (preprocess) order_copy = c->order_copy;       // first run: the emit kernel reads the caller's buffer and keeps a copy
        else order = c->order_copy;                        // re-run: the caller's buffer may have been overwritten since
    }
    if (preprocess) {
        if (c->proj_cap < npre) {
            int rc = sync_all(c); if (rc) return rc;
            for (int i = 0; i < 2; ++i) { if (c->proj2[i]) (void)hipFree(c->proj2[i]); if (c->rects2[i]) (void)hipFree(c->rects2[i]); c->proj2[i] = nullptr; c->rects2[i] = nullptr; }
            c->proj_cap = 0;
            for (int i = 0; i < 2; ++i) { HIPCHK(c, hipMalloc(&c->proj2[i], npre * 64)); HIPCHK(c, hipMalloc(&c->rects2[i], npre * 8)); }
            c->proj_cap = npre;
        }
        if (!a.quads && (a.mode == GS4D_MODE_4D_SORTED || a.mode == GS4D_MODE_4D_DIRECT)) { int rc = ensure_soa(c, *data); if (rc) return rc; }
        // The projection depends on neither the sort nor the previous frame's raster stage: it has a stream of its own and two
        // output buffers, so it runs beside the key generation / depth sort of this frame and the compositing of the last one.
        HIPCHK(c, hipStreamWaitEvent(c->ps, c->ev_soa, 0));
        HIPCHK(c, hipStreamWaitEvent(c->ps, c->ev_raster_done[a.pre_idx], 0));      // the draw two back, which read this buffer
        {
            StageTimer t(c, GS4D_T_PREPROCESS, c->ps);
            const PreOut po = { c->proj2[a.pre_idx], c->rects2[a.pre_idx] };
            if (a.quads) HIPCHK(c, launch_preprocess_3d(c->ps, (const float*)data->d, npre, a.u, c->W, c->H, po));
            else if (a.mode == GS4D_MODE_2D) HIPCHK(c, launch_preprocess_2d(c->ps, (const float*)data->d, npre, a.u, c->W, c->H, po));
            else HIPCHK(c, launch_preprocess_4d(c->ps, data->soa, npre, a.u, c->W, c->H, po));
        }
        HIPCHK(c, hipEventRecord(c->ev_pre_done, c->ps));
        HIPCHK(c, hipStreamWaitEvent(c->rs, c->ev_pre_done, 0));
        c->proj_n = npre;
        // binning reads the sort index (and any buffer the caller filled on `st` before this call)
        HIPCHK(c, hipEventRecord(c->ev_order_ready, c->st));
        HIPCHK(c, hipStreamWaitEvent(c->rs, c->ev_order_ready, 0));
    }
    size_t want = a.instances * 2 + 65536;
    if (want < c->stat_entries + c->stat_entries / 2) want = c->stat_entries + c->stat_entries / 2;
    if (c->pair_cap < want) { int rc = ensure_pairs(c, want); if (rc) return rc; }
    return enqueue_raster(c, order, order_copy, a.instances, npre, premult, a.fb_was_clear, a.pre_idx, a.seq);
}

// A draw's tile-list capacity is validated after the fact: the entry count comes back through pinned memory behind an event.
// Called by every entry point that could observe the draw's result or overwrite its inputs.  On overflow the raster stages are
// re-run with exact capacity (the projected records are still valid; inputs are unchanged by construction).
int resolve_pending(gs4d_ctx* c) {
    while (c->pending) {
        HIPCHK(c, hipEventSynchronize(c->ev_emit[c->pending_args.seq % gs4d_ctx::EMIT_RING]));     // the entry count is final once the binning kernel has run
        c->pending = false;
        if (c->emit_known < c->pending_args.seq) c->emit_known = c->pending_args.seq;
        if (c->host_total[4]) return fail(c, GS4D_E_DEVICE, "radix sort: a look-back spin timed out on the device (results of this frame are invalid)");
        const uint64_t total = (uint64_t)c->host_total[2] | ((uint64_t)c->host_total[3] << 32);
        if (!c->host_total[1]) { c->stat_entries = total; break; }
        if (total >= 0xFFFFFFF0ull) return fail(c, GS4D_E_UNSUPPORTED, "draw: more than 2^32 tile-list entries (splats cover too many tiles)");
        c->stat_reruns++;
        int rc = ensure_pairs(c, (size_t)(total + total / 8 + 1024));
        if (rc) return rc;
        c->stat_entries = total;
        rc = run_draw(c, c->pending_args, false);
        if (rc) return rc;
        c->pending = true;
    }
    return GS4D_OK;
}

// Is a draw that reads buffer `B` possibly still running?  (The host may be many frames ahead of the device.)
bool in_flight(gs4d_ctx* c, const Buffer& B) { return B.order_seq > c->done_seq || B.data_seq > c->done_seq; }

// The host is about to write (or free) `B`: wait for every draw that reads it.
int host_write_hazard(gs4d_ctx* c, const Buffer& B) {
    if (!in_flight(c, B)) return GS4D_OK;
    int rc = resolve_pending(c); if (rc) return rc;
    return sync_all(c);
}

// The order stage is about to overwrite `b`: if the in-flight draw reads its sort index from it, wait (on the device) until the
// binning kernel has consumed and copied it.  No host synchronisation.
int order_write_hazard(gs4d_ctx* c, gs4d_buf b) {
    Buffer* B = getbuf(c, b);
    if (!B) return GS4D_OK;
    if (B->data_seq > c->done_seq) return host_write_hazard(c, *B);          // overwriting the records of a running draw: rare, settle on the host
    const uint64_t known = c->done_seq > c->emit_known ? c->done_seq : c->emit_known;
    // ring slot seq % 8 holds the record of draw `seq` or of a later one: waiting for it is sufficient either way
    if (B->order_seq > known) HIPCHK(c, hipStreamWaitEvent(c->st, c->ev_emit[B->order_seq % gs4d_ctx::EMIT_RING], 0));
    return GS4D_OK;
}

int materialise_fb(gs4d_ctx* c) {
    if (c->fb_is_clear) { HIPCHK(c, launch_fill(c->rs, c->fb, (size_t)c->W * c->H, c->clear)); c->fb_is_clear = false; }
    return GS4D_OK;
}

int alloc_fb(gs4d_ctx* c, int w, int h) {
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535) return fail(c, GS4D_E_INVALID, "framebuffer size must be 1..65535");
    if (c->fb) { int rc = sync_all(c); if (rc) return rc; (void)hipFree(c->fb); c->fb = nullptr; }
    HIPCHK(c, hipMalloc(&c->fb, (size_t)w * h * 16));
    c->W = w; c->H = h; c->tiles_x = (w + TILE - 1) / TILE; c->tiles_y = (h + TILE - 1) / TILE;
    c->fb_is_clear = true;
    return GS4D_OK;
}

} // namespace

extern "C" {

const char* gs4d_version(void) { return "gs4d 0.1 (gfx950)"; }

const char* gs4d_last_error(gs4d_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int gs4d_create(int device, int width, int height, gs4d_ctx** out) {
    if (!out) return fail(nullptr, GS4D_E_INVALID, "gs4d_create: out == NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(nullptr, GS4D_E_DEVICE, "gs4d_create: no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, GS4D_E_INVALID, "gs4d_create: bad device index");
    if ((e = hipSetDevice(device)) != hipSuccess) return hipfail(nullptr, e, "hipSetDevice");
    gs4d_ctx* c = new (std::nothrow) gs4d_ctx();
    if (!c) return fail(nullptr, GS4D_E_NOMEM, "gs4d_create: out of host memory");
    c->device = device;
    c->bufs.resize(1);
    memset(&c->u, 0, sizeof c->u);
    for (int i = 0; i < 4; ++i) c->u.view[5 * i] = c->u.proj[5 * i] = 1.0f;
    auto bail = [&](int rc) { g_create_error = c->err; gs4d_destroy(c); return rc; };
    if ((e = hipStreamCreateWithFlags(&c->own_st, hipStreamNonBlocking)) != hipSuccess) return bail(hipfail(c, e, "hipStreamCreate"));
    c->st = c->own_st;
    c->single_stream = getenv("GS4D_STREAMS") && atoi(getenv("GS4D_STREAMS")) == 1;     // tuning knob (experiments only): every stage on one stream
    if (c->single_stream) { c->rs = c->ps = c->own_st; }
    else {
        if ((e = hipStreamCreateWithFlags(&c->rs, hipStreamNonBlocking)) != hipSuccess) return bail(hipfail(c, e, "hipStreamCreate"));
        if ((e = hipStreamCreateWithFlags(&c->ps, hipStreamNonBlocking)) != hipSuccess) return bail(hipfail(c, e, "hipStreamCreate"));
    }
    for (hipEvent_t* ev : { &c->ev_soa, &c->ev_order_ready, &c->ev_emit[0], &c->ev_emit[1], &c->ev_emit[2], &c->ev_emit[3], &c->ev_emit[4], &c->ev_emit[5], &c->ev_emit[6], &c->ev_emit[7], &c->ev_readback, &c->ev_pre_done, &c->ev_raster_done[0], &c->ev_raster_done[1] }) {
        if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return bail(hipfail(c, e, "hipEventCreate"));
        if ((e = hipEventRecord(*ev, c->st)) != hipSuccess) return bail(hipfail(c, e, "hipEventRecord"));      // "already happened"
    }
    if ((e = hipHostMalloc((void**)&c->host_total, 64, hipHostMallocMapped)) != hipSuccess) return bail(hipfail(c, e, "hipHostMalloc"));
    memset(c->host_total, 0, 64);
    if ((e = hipHostGetDevicePointer((void**)&c->host_total_dev, c->host_total, 0)) != hipSuccess) return bail(hipfail(c, e, "hipHostGetDevicePointer"));
    c->dev_err = c->host_total_dev + 4;
    c->depth_sort.err = c->dev_err; c->pair_sort.err = c->dev_err;
    { bool ordered = false; if ((e = lds_atomic_order_selftest(c->st, &ordered)) != hipSuccess) return bail(hipfail(c, e, "lds_atomic_order_selftest")); c->depth_sort.atomic_rank = c->pair_sort.atomic_rank = ordered; }
    int rc = alloc_fb(c, width, height);
    if (rc) return bail(rc);
    *out = c;
    return GS4D_OK;
}

void gs4d_destroy(gs4d_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->st) (void)hipStreamSynchronize(c->st);
    if (c->rs) (void)hipStreamSynchronize(c->rs);
    if (c->ps) (void)hipStreamSynchronize(c->ps);
    if (c->order_copy) (void)hipFree(c->order_copy);
    for (auto& b : c->bufs) { if (b.d) (void)hipFree(b.d); if (b.soa) (void)hipFree(b.soa); if (b.bbox_dev) (void)hipFree(b.bbox_dev); }
    if (c->fb) (void)hipFree(c->fb);
    for (int i = 0; i < 2; ++i) { if (c->proj2[i]) (void)hipFree(c->proj2[i]); if (c->rects2[i]) (void)hipFree(c->rects2[i]); }
    if (c->pair_keys) (void)hipFree(c->pair_keys);
    if (c->pair_vals) (void)hipFree(c->pair_vals);
    sort_scratch_free(c->depth_sort); sort_scratch_free(c->pair_sort); bin_scratch_free(c->bin);
    if (c->host_total) (void)hipHostFree(c->host_total);
    for (auto e : c->ev0) if (e) (void)hipEventDestroy(e);
    for (auto e : c->ev1) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t ev : { c->ev_soa, c->ev_order_ready, c->ev_emit[0], c->ev_emit[1], c->ev_emit[2], c->ev_emit[3], c->ev_emit[4], c->ev_emit[5], c->ev_emit[6], c->ev_emit[7], c->ev_readback, c->ev_pre_done, c->ev_raster_done[0], c->ev_raster_done[1] }) if (ev) (void)hipEventDestroy(ev);
    if (c->rs && !c->single_stream) (void)hipStreamDestroy(c->rs);
    if (c->ps && !c->single_stream) (void)hipStreamDestroy(c->ps);
    if (c->own_st) (void)hipStreamDestroy(c->own_st);
    delete c;
}

int gs4d_resize(gs4d_ctx* c, int width, int height) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = resolve_pending(c); if (rc) return rc;
    if (width == c->W && height == c->H) return GS4D_OK;
    return alloc_fb(c, width, height);
}

// ---- buffers ----
int gs4d_buffer_create(gs4d_ctx* c, const void* data, size_t bytes, gs4d_buf* out) {
    if (!c || !out) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    gs4d_buf name = 0;
    for (size_t i = 1; i < c->bufs.size(); ++i) if (!c->bufs[i].alive && !c->bufs[i].d) { name = (gs4d_buf)i; break; }
    if (!name) { c->bufs.emplace_back(); name = (gs4d_buf)(c->bufs.size() - 1); }
    Buffer nb;
    if (bytes) {
        HIPCHK(c, hipMalloc(&nb.d, bytes));
        if (data) { hipError_t e = hipMemcpyAsync(nb.d, data, bytes, hipMemcpyHostToDevice, c->st); if (e == hipSuccess) e = hipStreamSynchronize(c->st); if (e != hipSuccess) { (void)hipFree(nb.d); return hipfail(c, e, "buffer upload"); } }
    }
    nb.bytes = bytes; nb.alive = true; nb.version = 1;
    c->bufs[name] = nb;
    *out = name;
    return GS4D_OK;
}

int gs4d_buffer_subdata(gs4d_ctx* c, gs4d_buf b, size_t offset, const void* data, size_t bytes) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    Buffer* B = getbuf(c, b);
    if (!B) return fail(c, GS4D_E_INVALID, "buffer_subdata: bad buffer name");
    if (offset > B->bytes || bytes > B->bytes - offset) return fail(c, GS4D_E_INVALID, "buffer_subdata: range outside the buffer");   // GL_INVALID_VALUE
    if (!bytes) return GS4D_OK;
    if (!data) return fail(c, GS4D_E_INVALID, "buffer_subdata: data == NULL");
    { int rc = host_write_hazard(c, *B); if (rc) return rc; }
    // the caller keeps ownership of `data` and may reuse it on return (glBufferSubData semantics): copy synchronously
    HIPCHK(c, hipMemcpyAsync((char*)B->d + offset, data, bytes, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    B->version++;
    return GS4D_OK;
}

int gs4d_buffer_read(gs4d_ctx* c, gs4d_buf b, size_t offset, void* out, size_t bytes) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    Buffer* B = getbuf(c, b);
    if (!B) return fail(c, GS4D_E_INVALID, "buffer_read: bad buffer name");
    if (offset > B->bytes || bytes > B->bytes - offset || (!out && bytes)) return fail(c, GS4D_E_INVALID, "buffer_read: range outside the buffer");
    if (!bytes) return GS4D_OK;
    HIPCHK(c, hipMemcpyAsync(out, (const char*)B->d + offset, bytes, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    return GS4D_OK;
}

int gs4d_buffer_destroy(gs4d_ctx* c, gs4d_buf b) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    Buffer* B = getbuf(c, b);
    if (!B) return GS4D_OK;                         // 0, unknown or already deleted: silently ignored, like glDeleteBuffers
    { int rc = resolve_pending(c); if (rc) return rc; rc = sync_all(c); if (rc) return rc; }
    if (B->d) (void)hipFree(B->d);
    if (B->soa) (void)hipFree(B->soa);
    if (B->bbox_dev) (void)hipFree(B->bbox_dev);
    *B = Buffer();
    for (auto& s : c->slots) if (s == b) s = 0;     // a deleted buffer is unbound
    if (c->kg_buf == b) c->kg_buf = 0;
    return GS4D_OK;
}

int gs4d_buffer_device_ptr(gs4d_ctx* c, gs4d_buf b, void** dptr, size_t* bytes) {
    if (!c) return GS4D_E_INVALID;
    Buffer* B = getbuf(c, b);
    if (!B) return fail(c, GS4D_E_INVALID, "buffer_device_ptr: bad buffer name");
    if (dptr) *dptr = B->d;
    if (bytes) *bytes = B->bytes;
    B->version++;                                   // the caller may write through the pointer
    return GS4D_OK;
}

int gs4d_bind_storage(gs4d_ctx* c, int slot, gs4d_buf b) {
    if (!c) return GS4D_E_INVALID;
    if (slot < 0 || slot >= 8) return fail(c, GS4D_E_INVALID, "bind_storage: slot out of range");
    if (b != 0 && !getbuf(c, b)) return fail(c, GS4D_E_INVALID, "bind_storage: bad buffer name");
    c->slots[slot] = b;
    return GS4D_OK;
}

// ---- state ----
int gs4d_set_mode(gs4d_ctx* c, int mode) {
    if (!c) return GS4D_E_INVALID;
    if (mode < GS4D_MODE_4D_SORTED || mode > GS4D_MODE_2D) return fail(c, GS4D_E_INVALID, "set_mode: unknown mode");
    c->mode = mode; return GS4D_OK;
}
int gs4d_set_uniform_1f(gs4d_ctx* c, int id, float v) {
    if (!c) return GS4D_E_INVALID;
    if (id == GS4D_U_TIME) c->u.time = v; else if (id == GS4D_U_MIN_OPACITY) c->u.min_opacity = v; else return fail(c, GS4D_E_INVALID, "set_uniform_1f: unknown uniform");
    return GS4D_OK;
}
int gs4d_set_uniform_mat4(gs4d_ctx* c, int id, const float m[16]) {
    if (!c || !m) return GS4D_E_INVALID;
    if (id == GS4D_U_VIEW) memcpy(c->u.view, m, 64); else if (id == GS4D_U_PROJ) memcpy(c->u.proj, m, 64); else return fail(c, GS4D_E_INVALID, "set_uniform_mat4: unknown uniform");
    return GS4D_OK;
}
int gs4d_set_clear_color(gs4d_ctx* c, const float rgba[4]) {
    if (!c || !rgba) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    if (c->fb_is_clear && memcmp(c->clear, rgba, 16) != 0) { int rc = materialise_fb(c); if (rc) return rc; }  // glClearColor does not touch pixels
    memcpy(c->clear, rgba, 16); return GS4D_OK;
}
int gs4d_set_blend(gs4d_ctx* c, int src, int dst) {
    if (!c) return GS4D_E_INVALID;
    if (src != GS4D_SRC_ALPHA || dst != GS4D_ONE_MINUS_SRC_ALPHA) return fail(c, GS4D_E_UNSUPPORTED, "set_blend: only (SRC_ALPHA, ONE_MINUS_SRC_ALPHA) is implemented");
    return GS4D_OK;
}
int gs4d_clear(gs4d_ctx* c) {
    if (!c) return GS4D_E_INVALID;
    // Whatever a still-unvalidated draw left in the framebuffer is discarded by the clear; its inputs are no longer needed.
    c->pending = false;
    c->fb_is_clear = true;
    return GS4D_OK;
}

// ---- ordering ----
int gs4d_sort_pairs(gs4d_ctx* c, gs4d_buf keys, gs4d_buf vals, size_t n) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    if (n <= 1) return GS4D_OK;
    Buffer* K = getbuf(c, keys); Buffer* V = getbuf(c, vals);
    if (!K || !V) return fail(c, GS4D_E_INVALID, "sort_pairs: bad buffer name");
    if (K == V) return fail(c, GS4D_E_INVALID, "sort_pairs: keys and values must be different buffers");
    if (n >= 0xFFFFFFFFull || K->bytes < n * 4 || V->bytes < n * 4) return fail(c, GS4D_E_INVALID, "sort_pairs: buffers smaller than n elements");
    { int rc = order_write_hazard(c, keys); if (rc) return rc; rc = order_write_hazard(c, vals); if (rc) return rc; }
    // k_keygen leaves the digit histograms of the keys it wrote: no histogram launch when this sort is of exactly those keys
    const bool have_hist = c->depth_sort.hist_pending && keys == c->kg_buf && K->version == c->kg_ver && n == c->kg_n;
    StageTimer t(c, GS4D_T_SORT);
    HIPCHK(c, radix_sort_pairs(c->st, c->depth_sort, (uint32_t*)K->d, (uint32_t*)V->d, n, nullptr, have_hist ? c->depth_sort.hist_bits : 32, have_hist));
    K->version++; V->version++;
    return GS4D_OK;
}

int gs4d_keygen(gs4d_ctx* c, gs4d_buf data, float t, const float cam[3], gs4d_buf keys, gs4d_buf idx, size_t n, int key_mode) {
    if (!c || !cam) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    Buffer* D = getbuf(c, data); Buffer* K = getbuf(c, keys); Buffer* I = getbuf(c, idx);
    if (!D || !K || !I) return fail(c, GS4D_E_INVALID, "keygen: bad buffer name");
    if (key_mode != GS4D_KEY_REF_INV_EUCLID && key_mode != GS4D_KEY_VIEW_Z) return fail(c, GS4D_E_INVALID, "keygen: unknown key mode");
    if (n >= 0xFFFFFFFFull || D->bytes < n * 96 || K->bytes < n * 4 || I->bytes < n * 4) return fail(c, GS4D_E_INVALID, "keygen: buffers smaller than n elements");
    if (n == 0) return GS4D_OK;
    { int rc = order_write_hazard(c, keys); if (rc) return rc; rc = order_write_hazard(c, idx); if (rc) return rc; }
    int rc = ensure_soa(c, *D); if (rc) return rc;
    hipError_t he = hipSuccess;
    uint32_t* kh = sort_hist_slot(c->st, c->depth_sort, n, &he);
    if (!kh) return hipfail(c, he, "sort_hist_slot");
    // A proven lower bound of every key (1 / farthest possible distance, from the bounding box of the records) is subtracted inside
    // the sort's digit extraction: when the keys span less than 2^24 bit patterns above it (camera outside the cloud, far/near < 4)
    // the top digit becomes constant and its pass is skipped on the device.  k_keygen re-checks the bound for every key.
    // When the camera is provably outside the box the same reasoning gives an upper bound, hence the number of key bits above the
    // bias: with <= 24 the sort is launched with three passes instead of four (no launch for the constant digit at all).
    uint32_t bias = 0, span = 0xFFFFFFFFu;
    if (key_mode == GS4D_KEY_REF_INV_EUCLID && D->bb_ok) {          // the box covers all records of the buffer, a superset of the n keyed
        const double c_lo = (double)t - D->bb_hi[3], c_hi = (double)t - D->bb_lo[3];
        double d2 = 0.0, n2 = 0.0;
        for (int ax = 0; ax < 3; ++ax) {
            const double v_lo = D->bb_lo[4 + ax], v_hi = D->bb_hi[4 + ax];
            const double p1 = v_lo * c_lo, p2 = v_lo * c_hi, p3 = v_hi * c_lo, p4 = v_hi * c_hi;
            const double m_lo = D->bb_lo[ax] + std::min(std::min(p1, p2), std::min(p3, p4));
            const double m_hi = D->bb_hi[ax] + std::max(std::max(p1, p2), std::max(p3, p4));
            const double far = std::max(std::fabs(m_lo - (double)cam[ax]), std::fabs(m_hi - (double)cam[ax]));
            d2 += far * far;
            const double near = std::max(0.0, std::max(m_lo - (double)cam[ax], (double)cam[ax] - m_hi));
            n2 += near * near;
        }
        const double dmax = std::sqrt(d2) * (1.0 + 1e-4) + 1e-3;       // generous against float rounding in the kernel's own arithmetic
        const float lb = (float)((1.0 / dmax) * (1.0 - 1e-5));
        if (std::isfinite(dmax) && lb > 0.0f && std::isfinite(lb)) {
            memcpy(&bias, &lb, 4);
            const double dmin = std::sqrt(n2) * (1.0 - 1e-4) - 1e-3;
            if (dmin > 0.0) {
                const float ub = (float)((1.0 / dmin) * (1.0 + 1e-5));
                uint32_t ubits; memcpy(&ubits, &ub, 4);
                if (std::isfinite(ub) && ubits >= bias) span = ubits - bias;
            }
        }
    }
    StageTimer tm(c, GS4D_T_KEYGEN);
    HIPCHK(c, launch_keygen(c->st, D->soa, D->soa + 5 * D->soa_n, n, t, cam, c->u.view, key_mode, (float*)K->d, (uint32_t*)I->d, kh, bias, span, c->dev_err));
    c->depth_sort.hist_bias = bias;
    c->depth_sort.hist_bits = span < (1u << 8) ? 8 : span < (1u << 16) ? 16 : span < (1u << 24) ? 24 : 32;
    K->version++; I->version++;
    c->kg_buf = keys; c->kg_ver = K->version; c->kg_n = n;
    return GS4D_OK;
}

// ---- draw ----
static int draw_common(gs4d_ctx* c, DrawArgs& a) {
    (void)hipSetDevice(c->device);
    int rc = resolve_pending(c); if (rc) return rc;
    Buffer* d = getbuf(c, a.data); Buffer* o = getbuf(c, a.order);
    a.data_version = d ? d->version : 0; a.order_version = o ? o->version : 0;
    a.fb_was_clear = c->fb_is_clear;
    a.pre_idx = c->pre_idx ^ 1;
    a.seq = c->draw_seq + 1;
    const size_t before = c->proj_n;
    c->proj_n = 0;
    rc = run_draw(c, a, true);
    if (rc) { c->proj_n = before; return rc; }
    if (c->proj_n) { c->pending = true; c->pending_args = a; c->fb_is_clear = false; c->pre_idx = a.pre_idx; c->draw_seq = a.seq; if (d) d->data_seq = a.seq; if (o && a.mode == GS4D_MODE_4D_SORTED && !a.quads) o->order_seq = a.seq; }   // proj_n != 0 <=> raster work was enqueued
    if (c->profiling && c->prof_frame < gs4d_ctx::PROF_FRAMES) c->prof_frame++;
    return GS4D_OK;
}

int gs4d_draw_instanced(gs4d_ctx* c, size_t instances) {
    if (!c) return GS4D_E_INVALID;
    DrawArgs a; a.mode = c->mode; a.u = c->u; a.instances = instances; a.quads = false;
    if (c->mode == GS4D_MODE_4D_SORTED) { a.data = c->slots[2]; a.order = c->slots[1]; }
    else if (c->mode == GS4D_MODE_4D_DIRECT || c->mode == GS4D_MODE_2D) { a.data = c->slots[1]; a.order = 0; }
    else return fail(c, GS4D_E_INVALID, "draw_instanced: GS4D_MODE_3D_FULL draws with gs4d_draw_quads");
    return draw_common(c, a);
}

int gs4d_draw_quads(gs4d_ctx* c, gs4d_buf vertices, size_t nquads) {
    if (!c) return GS4D_E_INVALID;
    if (c->mode != GS4D_MODE_3D_FULL) return fail(c, GS4D_E_INVALID, "draw_quads: mode must be GS4D_MODE_3D_FULL");
    DrawArgs a; a.mode = c->mode; a.u = c->u; a.instances = nquads; a.quads = true; a.data = vertices; a.order = 0;
    return draw_common(c, a);
}

// ---- read-back ----
int gs4d_finish(gs4d_ctx* c) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = resolve_pending(c); if (rc) return rc;
    rc = sync_all(c); if (rc) return rc;
    if (c->host_total[4]) return fail(c, GS4D_E_DEVICE, "device-side check failed (a bounded look-back wait timed out, or a sort key fell below its proven bound): results are invalid");
    return GS4D_OK;
}

int gs4d_read_pixels(gs4d_ctx* c, float* rgba, size_t bytes) {
    if (!c || !rgba) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    if (bytes != (size_t)c->W * c->H * 16) return fail(c, GS4D_E_INVALID, "read_pixels: bytes != width*height*16");
    int rc = resolve_pending(c); if (rc) return rc;
    rc = materialise_fb(c); if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(rgba, c->fb, bytes, hipMemcpyDeviceToHost, c->rs));
    HIPCHK(c, hipStreamSynchronize(c->rs));
    if (c->host_total[4]) return fail(c, GS4D_E_DEVICE, "device-side check failed (a bounded look-back wait timed out, or a sort key fell below its proven bound): results are invalid");
    return GS4D_OK;
}

int gs4d_read_pixels_device(gs4d_ctx* c, void* dptr, size_t bytes) {
    if (!c || !dptr) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    if (bytes != (size_t)c->W * c->H * 16) return fail(c, GS4D_E_INVALID, "read_pixels_device: bytes != width*height*16");
    int rc = resolve_pending(c); if (rc) return rc;
    rc = materialise_fb(c); if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(dptr, c->fb, bytes, hipMemcpyDeviceToDevice, c->rs));
    HIPCHK(c, hipEventRecord(c->ev_readback, c->rs));
    HIPCHK(c, hipStreamWaitEvent(c->st, c->ev_readback, 0));      // work the caller queues on `st` after this call sees the pixels
    return GS4D_OK;
}

int gs4d_read_pixels_rgba8_device(gs4d_ctx* c, void* dptr, size_t bytes) {
    if (!c || !dptr) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    if (bytes != (size_t)c->W * c->H * 4) return fail(c, GS4D_E_INVALID, "read_pixels_rgba8_device: bytes != width*height*4");
    int rc = resolve_pending(c); if (rc) return rc;
    rc = materialise_fb(c); if (rc) return rc;
    HIPCHK(c, launch_pack_rgba8(c->rs, c->fb, (size_t)c->W * c->H, (uint32_t*)dptr));
    HIPCHK(c, hipEventRecord(c->ev_readback, c->rs));
    HIPCHK(c, hipStreamWaitEvent(c->st, c->ev_readback, 0));      // work the caller queues on `st` after this call sees the pixels
    return GS4D_OK;
}

int gs4d_set_stream(gs4d_ctx* c, void* hip_stream) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = resolve_pending(c); if (rc) return rc;
    rc = sync_all(c); if (rc) return rc;             // everything queued so far completes before work moves to the other stream
    c->st = hip_stream ? (hipStream_t)hip_stream : c->own_st;
    if (c->single_stream) c->rs = c->ps = c->st;
    return GS4D_OK;
}

// ---- measurement / test hooks ----
int gs4d_set_profiling(gs4d_ctx* c, int stage_mask) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    if (stage_mask && c->ev0.empty()) {
        const size_t n = (size_t)gs4d_ctx::PROF_FRAMES * GS4D_T_COUNT;
        c->ev0.assign(n, nullptr); c->ev1.assign(n, nullptr); c->ran.assign(n, 0);
        for (size_t i = 0; i < n; ++i) { HIPCHK(c, hipEventCreate(&c->ev0[i])); HIPCHK(c, hipEventCreate(&c->ev1[i])); }
    }
    c->profiling = (unsigned)stage_mask & 0x3Fu;
    c->prof_frame = 0;
    std::fill(c->ran.begin(), c->ran.end(), 0);
    return GS4D_OK;
}

int gs4d_get_timings(gs4d_ctx* c, float ms[GS4D_T_COUNT]) {
    if (!c || !ms) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = resolve_pending(c); if (rc) return rc;
    rc = sync_all(c); if (rc) return rc;
    // average per stage over the frames recorded since profiling was switched on (or since the last call); then restart
    for (int i = 0; i < GS4D_T_COUNT; ++i) {
        double sum = 0; int cnt = 0;
        for (int f = 0; f < gs4d_ctx::PROF_FRAMES && !c->ran.empty(); ++f) {
            const int slot = f * GS4D_T_COUNT + i;
            if (!c->ran[slot]) continue;
            float t = 0;
            if (hipEventElapsedTime(&t, c->ev0[slot], c->ev1[slot]) == hipSuccess) { sum += t; ++cnt; }
        }
        ms[i] = cnt ? (float)(sum / cnt) : -1.0f;
    }
    c->prof_frame = 0;
    std::fill(c->ran.begin(), c->ran.end(), 0);
    return GS4D_OK;
}

int gs4d_get_timeline(gs4d_ctx* c, float* ms, int max_frames, int* frames_out) {
    if (!c || !ms || !frames_out || max_frames < 0) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = resolve_pending(c); if (rc) return rc;
    rc = sync_all(c); if (rc) return rc;
    const int frames = std::min(std::min(max_frames, c->prof_frame), (int)gs4d_ctx::PROF_FRAMES);
    *frames_out = frames;
    if (c->ran.empty() || frames == 0) { *frames_out = 0; return GS4D_OK; }
    int base = -1;                                   // first stage of frame 0 that ran: time zero
    for (int i = 0; i < GS4D_T_COUNT && base < 0; ++i) if (c->ran[i]) base = i;
    if (base < 0) { *frames_out = 0; return GS4D_OK; }
    for (int f = 0; f < frames; ++f)
        for (int i = 0; i < GS4D_T_COUNT; ++i) {
            const int slot = f * GS4D_T_COUNT + i;
            float t0 = -1.0f, t1 = -1.0f;
            if (c->ran[slot]) {
                if (hipEventElapsedTime(&t0, c->ev0[base], c->ev0[slot]) != hipSuccess) t0 = -1.0f;
                if (hipEventElapsedTime(&t1, c->ev0[base], c->ev1[slot]) != hipSuccess) t1 = -1.0f;
            }
            ms[(size_t)slot * 2] = t0; ms[(size_t)slot * 2 + 1] = t1;
        }
    return GS4D_OK;
}

int gs4d_get_stats(gs4d_ctx* c, uint64_t stats[4]) {
    if (!c || !stats) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = resolve_pending(c); if (rc) return rc;
    stats[0] = c->stat_entries; stats[1] = c->pair_cap; stats[2] = c->stat_reruns; stats[3] = (uint64_t)c->tiles_x * c->tiles_y;
    return GS4D_OK;
}

int gs4d_debug_read_projected(gs4d_ctx* c, float* out16, size_t nrecords) {
    if (!c || !out16) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    if (nrecords > c->proj_n) return fail(c, GS4D_E_INVALID, "debug_read_projected: more records than the last draw projected");
    int rc = resolve_pending(c); if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(out16, c->proj2[c->pre_idx], nrecords * 64, hipMemcpyDeviceToHost, c->rs));
    HIPCHK(c, hipStreamSynchronize(c->rs));
    // expose the layout documented in gs4d.h: cx,cy,a0x,a0y,a1x,a1y,alpha,r,g,b,rect0,rect1,hx,hy,valid,0  (already the storage order)
    return GS4D_OK;
}

} // extern "C"
