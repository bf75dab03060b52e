// gs4d_api.hip — the C ABI (include/gs4d.h): context, buffer objects, pipeline state and the draw sequence.
//
// Stands in for the OpenGL objects and calls the reference's scenes use on this path:
//   ShareStorageBuffer ctor/SubData/Bind                    4DSplatRendering/ShareStorageBuffer.cpp:3-40
//   glGenBuffers/glBufferStorage/glBufferSubData/glBindBufferBase/glDeleteBuffers in the scenes   Scenes.h:241-247, 282-283, 321-325, 336, 220-224
//   Shader::Bind/SetUniform1f/SetUniformMat4f               4DSplatRendering/Shader.cpp:171-174, 206-209
//   radix_sort::sorter::sort                                Dependencies/GPU_RADIX_SORT/radix_sort.hpp:258-392
//   Renderer::Clear / Renderer::Draw                        4DSplatRendering/Renderer.cpp:20-39
//
// Execution model: FRAME LANES.  A context owns a few lanes (4 by default, GS4D_LANES=1..8); a lane is one HIP stream with its own framebuffer,
// projected records, tile lists and sort scratch.  Every call of one frame (key generation, depth sort, draw) is queued on the
// current lane, in order — no events inside a frame.  The first frame-starting call (clear, keygen, sort) after a draw moves to the
// next lane, so whole frames overlap on the device: the latency-bound kernels of one frame (the radix sort's chained scans) fill the
// gaps of the bandwidth-bound kernels of its neighbours.  Lanes only meet through buffer objects and framebuffers; each of those
// remembers which lane wrote it last and which lanes read it since, and a lane about to touch it waits on the other lane's event —
// on the device, never on the host.  Only read-back/finish calls block, plus the validation of a draw's tile-list capacity, which
// is deferred to the next call that could observe the draw (see resolve_pending).
#include "gs4d_internal.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>
#include <algorithm>
#include <vector>
#include <cmath>

using namespace gs4d;

namespace {

constexpr int MAX_LANES = 8;

struct Buffer {
    void* d = nullptr;
    size_t bytes = 0;
    uint64_t version = 0;          // bumped by every write; the SoA shadow and the keygen-histogram hand-off compare against it
    float4* soa = nullptr;         // lazily built SoA shadow of 96-B SplatData records
    SoaInfo soa_info;              // ... its layout (preprocess.hip): 64 B/record for static 3D splats, 72 for a symmetric sig, 96 otherwise
    size_t soa_n = 0;
    uint64_t soa_version = ~0ull;
    uint32_t* bbox_dev = nullptr;  // 16 words: bounding box of pos / mu_t / velocity, reduced by the repack kernel
    double bb_lo[7] = { 0 }, bb_hi[7] = { 0 }; bool bb_ok = false;
    // cross-lane hazards
    int wr_lane = -1;              // lane whose kernels wrote the buffer last (-1: the host did, synchronously)
    unsigned ordered_mask = 0;     // lanes that have already ordered themselves after that write
    unsigned rd_mask = 0;          // lanes whose DRAWS have read it since that write (their binning-done event covers the reads)
    unsigned tail_mask = 0;        // lanes whose other kernels (key generation) have read it since that write (their tail event does)
    uint64_t touch = 0;            // context op counter at the last device-side use (host writes compare it with the last full sync)
    bool alive = false;
    bool ptr_exposed = false;      // gs4d_buffer_device_ptr has handed the storage's address out: the storage may never be exchanged (see Lane::spare)
    // The caller announced (gs4d_buffer_invalidate) that work on ITS stream rewrites the buffer: every lane that uses it afterwards first
    // waits for an event recorded on that stream at the first such use (the caller has queued the writes by then: that is the contract).
    hipEvent_t ev_fill = nullptr; unsigned fill_mask = 0; bool fill_recorded = false;
    // Provenance of a sort index: set when gs4d_sort_pairs has sorted exactly the keys and the identity index gs4d_keygen wrote for
    // `prov_data` — the contents are then "records of prov_data in ascending (depth key, record index)" for as long as `version`
    // still equals prov_ver, and a draw that binds it can take its blend order from the keys instead of reading it (tilelist.hip).
    bool prov_valid = false; gs4d_buf prov_data = 0; uint64_t prov_data_ver = 0, prov_ver = 0; size_t prov_n = 0; int prov_bits = 32; uint32_t prov_span = 0xFFFFFFFFu;
    KeySrc prov_ks;
};

struct DrawArgs {
    int mode = 0;
    Uniforms u;
    gs4d_buf data = 0, order = 0;
    size_t instances = 0;
    bool quads = false;
    int lane = 0, fb = 0;          // where the draw ran
    bool v2 = false;               // unordered tile lists (tilelist.hip); false: instance-ordered lists (binning.hip)
    KeySrc ks; int keybits = 32;   // v2: where the blend order comes from
    uint32_t key_span = 0xFFFFFFFFu; // ... and the host-proven largest blend key (the depth slabs divide [0, key_span])
    // the draw also executes the gs4d_keygen + gs4d_sort_pairs that were queued for it (first run only: a re-run finds the buffers sorted)
    bool fuse = false; gs4d_buf fuse_keys = 0, fuse_idx = 0; uint32_t fuse_span = 0xFFFFFFFFu;
    int blend_src = GS4D_SRC_ALPHA, blend_dst = GS4D_ONE_MINUS_SRC_ALPHA;       // glBlendFunc state at the draw
    float clear[4] = { 0, 0, 0, 0 };   // what "clear" meant for the image when the draw was issued (a re-run must not pick up a later glClearColor)
    int shard_rank = 0, shard_world = 1;       // ... and the tile-row shard
    // The re-run of an unordered draw on the ordered path: the draw never read the caller's sort index (it took its order from the keys), and by
    // now the application may have overwritten it for a later frame.  The re-run regenerates "records in ascending (key, index)" from `ks`
    // into the lane's private index instead of reading the caller's buffer.
    bool regen_order = false;
    bool exact_lists = false;         // the re-run of a staged draw whose blocks, runs or buckets did not fit: builds its lists exactly (scan + scatter)
    uint64_t stage_geom = 0;        // set by run_draw: the list geometry the draw's bucket statistics belong to (resolve_lane files them under it)
};

struct Framebuffer {
    float4* mem = nullptr;
    uint32_t* linecnt = nullptr;   // per-pixel fragment counters of the overlay-line kernels (lines.hip), allocated on first use, all-zero between calls
    bool is_clear = true;          // no draw has touched the image since its gs4d_clear
    // Tile state ("fast clear", composite.hip): tstate[tile] == epoch <=> the tile's pixels are in memory; any other tile is still the clear
    // colour.  gs4d_clear takes a new epoch (no memset, no fill); the compositing kernels write the tiles that have list entries; readers
    // substitute the clear colour (RGBA8 packs) or have the rest written first (materialise_fb: host read-backs, float copies, overlay lines).
    uint32_t* tstate = nullptr; uint32_t epoch = 1; bool all_in_memory = false;
    float clear[4] = { 0, 0, 0, 0 };   // the clear colour: the context's at the gs4d_clear that cleared this image
    int last_lane = -1;            // lane that touched it last
};

struct Lane {
    hipStream_t s = nullptr;
    hipEvent_t ev_emit = nullptr;      // the lane's latest binning kernel has finished (it read and copied the sort index; its entry count is final)
    hipEvent_t ev_tail = nullptr;      // recorded when the lane is left: everything queued on it so far
    bool drawn = false;                // the lane's current frame has a draw in it: the next frame-starting call moves on
    float4* proj = nullptr; uint32_t* trects = nullptr; size_t proj_cap = 0, proj_n = 0;   // projected records (64 B) and their packed tile rectangles (4 B)
    bool trects_in_order = false;      // the last draw's tile rectangles are in INSTANCE order (they went through its depth sort), not in record order
    uint32_t* pair_keys = nullptr; uint32_t* pair_vals = nullptr; size_t pair_cap = 0;   // tile-list entries: one allocation of 8 * pair_cap bytes — (tile ids | records) on the ordered path, (key, record) pairs on the unordered one
    TileLists tl;                      // unordered path: per-tile counts / starts / cursors, per-record blend keys
    uint32_t* order_copy = nullptr; size_t order_cap = 0;   // private copy of the last draw's sort index (for a re-run after overflow)
    uint32_t* regen_keys = nullptr; size_t regen_cap = 0;   // keys of a regenerated sort index (DrawArgs::regen_order), allocated on first use
    SortScratch depth_sort, pair_sort;
    BinScratch bin;
    float* line_verts = nullptr; size_t line_cap = 0;   // device copy of the vertices of the latest gs4d_draw_lines (the lane's stream orders its reuse)
    uint32_t* host_total = nullptr;     // pinned + mapped: [0..3] the binning total of the last draw, [4] the error word kernels raise
    uint32_t* host_total_dev = nullptr; // the same memory as the device sees it
    gs4d_buf kg_buf = 0; uint64_t kg_ver = 0; size_t kg_n = 0;   // key buffer whose digit histograms k_keygen left for the next sort
    gs4d_buf kg_idx = 0, kg_data = 0; uint64_t kg_idx_ver = 0, kg_data_ver = 0; KeySrc kg_ks; int kg_bits = 32; uint32_t kg_span = 0xFFFFFFFFu;   // ... the identity index it wrote beside them, and what the keys were computed from
    // Storage renaming for per-frame key / index buffers.  A buffer object is a NAME; its device storage is the library's.  gs4d_keygen
    // overwrites its two output buffers entirely, so when their storage is still being written or read by ANOTHER lane's frame (an
    // application with one key / index pair for all frames: the reference's layout, Scenes.h m_key_buf / m_values_buf) the new frame
    // does not wait for it: the buffers exchange their storage with this lane's spare pair and the earlier frame finishes on what is now
    // the spare.  When storage leaves a buffer everything that uses it has already been queued (API calls are sequential): an event is
    // recorded right then on every lane that wrote or still reads it, and the lane that takes the storage back later waits for exactly
    // those events — not for the lanes' tail events, which by then cover later frames as well and would chain the lanes to each other.
    struct Spare { void* d = nullptr; size_t bytes = 0; uint64_t touch = 0; unsigned wait_mask = 0; hipEvent_t ev[MAX_LANES] = { nullptr }; } spare[2];
    bool pending = false;              // the lane's last draw has not had its tile-list capacity validated yet
    bool discarded = false;            // ... and its image has been cleared since: validated (counted, learned from) but never re-run
    DrawArgs pending_args;
};

thread_local std::string g_create_error;

} // namespace

struct gs4d_ctx {
    int device = 0;
    int W = 0, H = 0, tiles_x = 0, tiles_y = 0;
    int nlanes = 4, cur = 0;
    Lane lanes[MAX_LANES];
    Framebuffer fbs[MAX_LANES];        // fbs[i] belongs to lane i; a clear makes the current lane's own framebuffer the current one
    int cur_fb = 0;
    hipStream_t user = nullptr;        // the caller's stream (gs4d_set_stream), or null
    hipEvent_t ev_user = nullptr;      // user stream -> lane: what the caller queued before an API call
    hipEvent_t ev_readback = nullptr;  // lane -> user stream: a device-side read-back
    std::string err;
    std::vector<Buffer> bufs;          // index = name; bufs[0] unused
    gs4d_buf slots[8] = { 0 };
    int mode = GS4D_MODE_4D_SORTED;
    Uniforms u;
    float clear[4] = { 0.0f, 0.0f, 0.0f, 0.0f };     // GL's initial clear colour; the app sets its own (Application.cpp:125)
    bool atomic_rank = false;          // result of the LDS-atomic ordering self-test
    uint64_t ops = 0, synced = 0;      // device-side uses so far / at the last sync of every lane
    int shard_rank = 0, shard_world = 1;   // single-frame sharding: this context bins and composites the tile rows ty % world == rank
    int prev_fb = -1;                  // the image the last gs4d_clear moved away from (still intact until its lane comes round again)
    uint64_t stat_entries = 0, stat_reruns = 0, stat_depth_passes = 0, stat_tile_passes = 0;
    uint64_t stat_aborted_discarded = 0;   // draws that aborted on the device (capacity, list length) and were cleared away before anybody observed them
    // draw path selection: the unordered path needs lists short enough to be sorted in LDS (<= V2_MAX_LIST entries per tile)
    int path_pref = 0;                 // GS4D_DRAW_PATH: 0 auto, 1 ordered path only, 2 = auto (kept for symmetry)
    bool long_lists = false;           // the last unordered draw met a list longer than V2_MAX_LIST: draws use the ordered path ...
    uint64_t ordered_draws = 0;        // ... and probe the unordered one again every so often when the lists look short on average
    // A gs4d_keygen (and the gs4d_sort_pairs of its output) is not launched at once: if the draw that follows takes its blend order from
    // exactly that sort, the projection kernel generates the keys as a by-product (it recomputes them anyway) and the sort is queued
    // behind the draw.  Any other call that could observe the buffers launches the stand-alone kernels first (flush_order).
    struct { bool keygen = false, sorted = false; int lane = 0; gs4d_buf data = 0, keys = 0, idx = 0; size_t n = 0; float t = 0; float cam[3] = { 0, 0, 0 };
             int key_mode = 0; uint32_t bias = 0, span = 0xFFFFFFFFu; float view[16] = { 0 }; } po;
    int blend_src = GS4D_SRC_ALPHA, blend_dst = GS4D_ONE_MINUS_SRC_ALPHA;     // glBlendFunc state (Application.cpp:137-138, 150)
    bool defer_order = true;           // GS4D_FUSE_KEYGEN=0 switches the deferral off (test hook)
    uint64_t stat_composited_tiles = 0;      // tiles the compositing kernel of the last unordered draw was launched for (staged draws: the launch box)
    uint64_t stat_fused = 0, stat_renamed = 0, stat_shadow_bytes = 0, stat_streams_rejected = 0, stat_lanes_sharing = 0;      // lanes_sharing: lanes that had to take a stream which shares a hardware queue with another lane
    bool rename_storage = true;        // GS4D_RENAME=0 switches the storage exchange off (test hook)
    int shrink_votes = 0;
    // Two ways to get a tile's list into blend order.  Lists of up to V2_MAX_LIST entries: built unordered, ordered by the wave that
    // composites the tile (k_composite_v2).  Longer lists, or a blend order that is not a key the library knows: the instance-ordered path
    // (binning.hip) — at 10^7 splats (1500 entries on the average non-empty tile) it is the faster one by 7-17 % against every way of
    // keeping such lists on the unordered path that round 3 built and measured: depth slabs of the lists, a bucket-wide LDS sort.
    uint32_t slabs = 1;                // depth slabs per tile list (tilelist.hip): only GS4D_SLABS sets it — an experiment knob whose mechanism stays tested
    uint32_t list_hint = 256;          // LDS list capacity the compositor is launched with (64 << k); grows on demand, validated per draw on the device
    // Staged lists (tilelist.hip): once a draw of a scene has reported its fullest segment, its longest (bucket, segment) run and its fullest bucket,
    // the draws that follow let the projection kernel write the list entries itself — one dense block per segment, sized by those statistics plus a
    // margin — no scan and no scatter kernel.  The device checks the guess; a draw that does not fit is re-run exactly.  GS4D_STAGED=0 switches it
    // off (test hook).
    bool stage_enable = true, stage_known = false, stage_box_enable = true;
    uint32_t stage_max_run = 0, stage_max_bucket = 0, stage_max_seg = 0; uint64_t stage_geom = 0;
    uint32_t stage_box_margin = 1;  // blocks added on every side of it; doubled (up to 16) whenever a draw had entries outside its box
    uint32_t stage_box = BOX_NONE;  // blocks of tiles that held entries in the last staged draw of this geometry (TileLists::box); BOX_NONE: not known
    uint64_t stat_staged = 0, stat_staged_misses = 0;
    uint64_t stat_v2_draws = 0, stat_longest = 0;
    // profiling: a ring of per-frame event pairs; a frame ends with its draw
    static constexpr int PROF_FRAMES = 128;
    unsigned profiling = 0;                    // bit s set: stage s is timed
    int prof_frame = 0;
    int prof_every = 1;                        // only every prof_every-th frame is timed ...
    uint64_t prof_tick = 0;                    // ... counted here (a frame ends with its draw)
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

const char* const DEVICE_CHECK_MSG = "device-side check failed (a bounded look-back wait timed out, a sort key fell outside its proven bounds, or a key buffer changed under a queued sort without gs4d_buffer_invalidate): results are invalid";

Buffer* getbuf(gs4d_ctx* c, gs4d_buf b) { return (b != 0 && b < c->bufs.size() && c->bufs[b].alive) ? &c->bufs[b] : nullptr; }
Lane& lane(gs4d_ctx* c) { return c->lanes[c->cur]; }

struct StageTimer {
    gs4d_ctx* c; int slot; hipStream_t s;
    StageTimer(gs4d_ctx* c_, int id) : c(c_), slot(-1), s(c_->lanes[c_->cur].s) {      // run_draw re-runs make their lane current first (resolve_lane)
        if (((c->profiling >> id) & 1u) && c->prof_frame < gs4d_ctx::PROF_FRAMES && c->prof_tick % (uint64_t)c->prof_every == 0) { slot = c->prof_frame * GS4D_T_COUNT + id; (void)hipEventRecord(c->ev0[slot], s); }
    }
    ~StageTimer() { if (slot >= 0) { (void)hipEventRecord(c->ev1[slot], s); c->ran[slot] = 1; } }
};

int sync_all(gs4d_ctx* c) {
    for (int i = 0; i < c->nlanes; ++i) HIPCHK(c, hipStreamSynchronize(c->lanes[i].s));
    c->synced = c->ops;
    return GS4D_OK;
}

bool device_error(gs4d_ctx* c) { for (int i = 0; i < c->nlanes; ++i) if (c->lanes[i].host_total[4]) return true; return false; }

// The current frame is complete (it has a draw in it) and a new one starts: move to the next lane.
int next_frame_if_drawn(gs4d_ctx* c) {
    Lane& L = lane(c);
    if (!L.drawn) return GS4D_OK;
    HIPCHK(c, hipEventRecord(L.ev_tail, L.s));
    L.drawn = false;
    c->cur = (c->cur + 1) % c->nlanes;
    lane(c).drawn = false;
    return GS4D_OK;
}

// What the caller queued on its own stream before this API call happens before what the call queues.
int after_user_stream(gs4d_ctx* c) {
    if (!c->user) return GS4D_OK;
    HIPCHK(c, hipEventRecord(c->ev_user, c->user));
    HIPCHK(c, hipStreamWaitEvent(lane(c).s, c->ev_user, 0));
    return GS4D_OK;
}

// The current lane is about to read (or overwrite) buffer B with a kernel: order it after the other lanes' kernels that wrote B
// (or, for a write, still read it).  Device-side waits only.  A lane other than the current one has been left since it last touched
// B, so its tail event (recorded on leaving) covers that use; a draw's reads are already covered by its binning-done event.
int resolve_lane(gs4d_ctx* c, int li);
int after_user_fill(gs4d_ctx* c, Buffer& B) {
    const unsigned me = 1u << c->cur;
    if (!(B.fill_mask & me)) return GS4D_OK;
    B.fill_mask &= ~me;
    if (!c->user) { B.fill_mask = 0; return GS4D_OK; }
    if (!B.ev_fill) HIPCHK(c, hipEventCreateWithFlags(&B.ev_fill, hipEventDisableTiming));
    if (!B.fill_recorded) { HIPCHK(c, hipEventRecord(B.ev_fill, c->user)); B.fill_recorded = true; }
    HIPCHK(c, hipStreamWaitEvent(lane(c).s, B.ev_fill, 0));
    return GS4D_OK;
}
int lane_access(gs4d_ctx* c, Buffer& B, bool write) {
    { int rc = after_user_fill(c, B); if (rc) return rc; }
    // (An unordered draw never read its sort index, and its re-run does not either — DrawArgs::regen_order — so overwriting the index a
    // still-unvalidated draw was given needs no validation first: an application with ONE key / index buffer pair, the reference's layout
    // (Scenes.h m_key_buf / m_values_buf), is ordered lane after lane on the device by the events below, never on the host.)
    Lane& L = lane(c);
    const unsigned me = 1u << c->cur;
    if (B.wr_lane >= 0 && B.wr_lane != c->cur && !(B.ordered_mask & me)) {
        HIPCHK(c, hipStreamWaitEvent(L.s, c->lanes[B.wr_lane].ev_tail, 0));
        B.ordered_mask |= me;
    }
    if (write) {
        for (int r = 0; r < c->nlanes; ++r) {
            if (r == c->cur) continue;
            if ((B.tail_mask >> r) & 1u) HIPCHK(c, hipStreamWaitEvent(L.s, c->lanes[r].ev_tail, 0));
            else if ((B.rd_mask >> r) & 1u) HIPCHK(c, hipStreamWaitEvent(L.s, c->lanes[r].ev_emit, 0));
        }
        B.rd_mask = 0; B.tail_mask = 0; B.wr_lane = c->cur; B.ordered_mask = me;
    }
    B.touch = ++c->ops;
    return GS4D_OK;
}

// The host is about to write, read or free B: wait for every kernel that may still use it.
int host_access(gs4d_ctx* c, Buffer& B);

int ensure_pairs(gs4d_ctx* c, Lane& L, size_t cap) {
    if (L.pair_cap >= cap) return GS4D_OK;
    HIPCHK(c, hipStreamSynchronize(L.s));
    if (L.pair_keys) (void)hipFree(L.pair_keys);
    L.pair_keys = L.pair_vals = nullptr; L.pair_cap = 0;
    HIPCHK(c, hipMalloc(&L.pair_keys, cap * 16));      // ordered path: tile ids | records (4 + 4 bytes per slot); unordered path: two arrays of (key, record)
    L.pair_vals = L.pair_keys + cap;
    L.pair_cap = cap;
    return GS4D_OK;
}

int ensure_soa(gs4d_ctx* c, Buffer& b) {
    const size_t n = b.bytes / 96;
    if (b.soa && b.soa_n == n && b.soa_version == b.version) return GS4D_OK;
    if (b.touch > c->synced) { int rc = sync_all(c); if (rc) return rc; }      // a running draw may still project from the old shadow
    { int rc = after_user_fill(c, b); if (rc) return rc; }                     // the repack reads what the caller's stream is writing
    Lane& L = lane(c);
    if (!b.soa || b.soa_n != n) {
        if (b.soa) { (void)hipFree(b.soa); b.soa = nullptr; }
        if (n) HIPCHK(c, hipMalloc(&b.soa, n * 96));
        b.soa_n = n;
    }
    // bounding box of everything the sort key depends on (upload-time work: one small read-back per refresh)
    uint32_t init[16]; for (int i = 0; i < 16; ++i) init[i] = i < 7 ? 0xFFFFFFFFu : 0u;
    if (!b.bbox_dev) HIPCHK(c, hipMalloc(&b.bbox_dev, 64));
    uint32_t got[16];
    // the most compact layout the records allow: static 3D splats (the time row / column of sig and mu_t the same in every record: those of
    // record 0), else a symmetric sig, else everything.  The kernel verifies the assumption for every record; a violation sends the round again.
    const bool allow_compact = !(getenv("GS4D_SOA_FULL") && atoi(getenv("GS4D_SOA_FULL")));             // test hook: always the 96-byte layout (upload-time code: read per repack)
    float rec0[24] = { 0 };
    if (n) { HIPCHK(c, hipMemcpyAsync(rec0, b.d, 96, hipMemcpyDeviceToHost, L.s)); HIPCHK(c, hipStreamSynchronize(L.s)); }
    static const int order[3] = { SOA_STATIC3D, SOA_SYM, SOA_FULL };
    for (int attempt = allow_compact ? 0 : 2; attempt < 3; ++attempt) {
        b.soa_info = SoaInfo();
        b.soa_info.layout = order[attempt];
        if (order[attempt] == SOA_STATIC3D) { const float cs[8] = { rec0[3], rec0[11], rec0[15], rec0[19], rec0[20], rec0[21], rec0[22], rec0[23] }; memcpy(b.soa_info.consts, cs, sizeof cs); }
        HIPCHK(c, hipMemcpyAsync(b.bbox_dev, init, 64, hipMemcpyHostToDevice, L.s));
        HIPCHK(c, launch_soa_repack(L.s, (const float*)b.d, n, b.soa, b.bbox_dev, b.soa_info));
        HIPCHK(c, hipMemcpyAsync(got, b.bbox_dev, 64, hipMemcpyDeviceToHost, L.s));
        HIPCHK(c, hipStreamSynchronize(L.s));      // the shadow is complete before any lane can be asked to read it
        if (order[attempt] == SOA_FULL || got[15] == 0) break;
    }
    auto ord2f = [](uint32_t u) { u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; float f; memcpy(&f, &u, 4); return (double)f; };
    b.bb_ok = n > 0 && got[14] == 0;
    for (int k = 0; k < 7; ++k) { b.bb_lo[k] = ord2f(got[k]); b.bb_hi[k] = ord2f(got[7 + k]); if (!(b.bb_lo[k] <= b.bb_hi[k])) b.bb_ok = false; }
    b.soa_version = b.version;
    return GS4D_OK;
}

// The current lane is about to use framebuffer F: order it after the lane that touched F last.
int fb_access(gs4d_ctx* c, Framebuffer& F) {
    if (F.last_lane >= 0 && F.last_lane != c->cur) HIPCHK(c, hipStreamWaitEvent(lane(c).s, c->lanes[F.last_lane].ev_tail, 0));
    F.last_lane = c->cur;
    return GS4D_OK;
}

// every tile of the current image into memory (the lazily clear ones get their clear colour): for whoever reads or writes single pixels
int materialise_fb(gs4d_ctx* c) {
    Framebuffer& F = c->fbs[c->cur_fb];
    if (!F.all_in_memory) {
        int rc = fb_access(c, F); if (rc) return rc;
        HIPCHK(c, launch_fill_unwritten(lane(c).s, F.mem, F.tstate, F.epoch, c->tiles_x, c->tiles_y, c->W, c->H, F.clear));
        F.all_in_memory = true;
        F.is_clear = false;
    }
    return GS4D_OK;
}

// Enqueue binning -> tile sort -> ranges -> composite for the projected records in L.proj.
int enqueue_raster(gs4d_ctx* c, Lane& L, Framebuffer& F, const DrawArgs& a, const uint32_t* order, uint32_t* order_copy, size_t ninst, size_t nrecords, int premult_c) {
    const int blend_src = a.blend_src, blend_dst = a.blend_dst;
    const size_t ntiles = (size_t)c->tiles_x * c->tiles_y;
    int tile_bits = 1; while (((size_t)1 << tile_bits) < ntiles) ++tile_bits;
    // (4K: 129 600 tiles = 17 bits — three passes of 8-bit digits, two of 9-bit ones)
    const int tile_rb = L.pair_sort.hist_rb = sort_plan_rb(L.pair_sort, L.pair_cap, tile_bits), tile_passes = sort_plan_passes(tile_bits, tile_rb);
    {
        StageTimer t(c, GS4D_T_BINNING);
        hipError_t he = hipSuccess;
        uint32_t* ph = sort_hist_slot(L.s, L.pair_sort, L.pair_cap, &he);      // the emit kernel also counts the tile-id digits
        if (!ph) return hipfail(c, he, "sort_hist_slot");
        HIPCHK(c, launch_binning(L.s, L.bin, L.trects, L.trects_in_order, L.proj, order, order_copy, ninst, nrecords, c->tiles_x, c->tiles_y, L.pair_keys, L.pair_vals, L.pair_cap, L.host_total_dev + 4,
                                 ph, tile_passes | (tile_rb << 8), L.host_total_dev, a.shard_rank, a.shard_world));
    }
    HIPCHK(c, hipEventRecord(L.ev_emit, L.s));     // the last binning workgroup wrote the total straight into pinned host memory
    {
        StageTimer t(c, GS4D_T_PAIRSORT);
        HIPCHK(c, radix_sort_pairs(L.s, L.pair_sort, L.pair_keys, L.pair_vals, L.pair_cap, L.bin.total, tile_bits, true));
    }
    c->stat_tile_passes = (uint64_t)tile_passes;
    {
        StageTimer t(c, GS4D_T_COMPOSITE);      // the per-tile ranges and the compositing kernel
        HIPCHK(c, launch_tile_ranges(L.s, L.bin, L.pair_keys, L.pair_cap, ntiles));
        HIPCHK(c, launch_composite(L.s, L.proj, L.pair_vals, L.bin.ranges, L.bin.total, c->tiles_x, c->tiles_y, c->W, c->H, premult_c, F.tstate, F.epoch, a.clear, F.mem, blend_src, blend_dst));
    }
    return GS4D_OK;
}

// Unordered path: the projection kernel has counted the entries per tile; scan, scatter, composite (tilelist.hip, composite2.hip).
int enqueue_raster_v2(gs4d_ctx* c, Lane& L, Framebuffer& F, const DrawArgs& a, size_t nrecords, int premult_c) {
    const size_t ntiles = (size_t)c->tiles_x * c->tiles_y;
    uint2* tmp = (uint2*)L.pair_keys;              // the lane's entry storage holds 16 bytes per slot: [0, cap) bucket order, [cap, 2 cap) tile order
    uint2* entries = tmp + L.pair_cap;
    int recbits = 1; while (recbits < 32 && ((size_t)1 << recbits) < nrecords) ++recbits;
    {
        StageTimer t(c, GS4D_T_BINNING);
        bool skip_lists = false;
#ifdef GS4D_TUNING
        { static const bool sk = getenv("GS4D_ABLATE_SCATTER") != nullptr; static int warm = 0; skip_lists = sk && ++warm > 16; }      // ablation: steady-state frames reuse the bucket array of an earlier frame of their lane (same scene): what scan + scatter cost the frame
#endif
        const uint32_t* fused_keys = nullptr;
        if (a.fuse) { Buffer* K = getbuf(c, a.fuse_keys); if (K) fused_keys = (const uint32_t*)K->d; }      // the projection wrote the keys there and nowhere else
        if (L.tl.staged) {
            // staged: the projection kernel wrote the segment blocks; one kernel turns them into tile lists and checks what the host guessed
            HIPCHK(c, launch_bucket_tiles_staged(L.s, L.tl, ntiles, c->tiles_x, L.bin.total, entries, c->list_hint));
            c->stat_staged++;
        } else {
            if (!skip_lists) {
            HIPCHK(c, launch_bucket_scan(L.s, L.tl, L.bin.total, L.host_total_dev, L.pair_cap));
            HIPCHK(c, launch_bucket_scatter(L.s, L.tl, L.trects, L.proj, fused_keys, a.ks.bias, nrecords, L.bin.total, tmp, c->tiles_x, a.shard_rank, a.shard_world));
            }
            HIPCHK(c, launch_bucket_tiles(L.s, L.tl, ntiles, L.bin.total, tmp, entries, c->list_hint));
        }
    }
    c->stat_tile_passes = 0;
#ifdef GS4D_TUNING
    { static const bool skip = getenv("GS4D_ABLATE_COMPOSITE") != nullptr; if (skip) { HIPCHK(c, hipEventRecord(L.ev_emit, L.s)); return GS4D_OK; } }      // ablation: the frame without its compositing kernel (steady state only: the verdict words keep their last values)
#endif
    {
        StageTimer t(c, GS4D_T_COMPOSITE);
        HIPCHK(c, launch_composite_v2(L.s, L.proj, entries, L.tl.tstart, L.tl.tcnt, L.bin.total, L.host_total_dev, c->tiles_x, c->tiles_y, c->W, c->H, premult_c, F.tstate, F.epoch, a.clear, F.mem,
                                      c->list_hint, a.keybits, recbits, L.tl.slabs, L.tl.bstat, L.tl.nb, L.tl.sstat, L.tl.rows, L.tl.staged ? L.tl.seq : 0u, 0xFFFFFFFFu, L.tl.scap, L.tl.bcap, L.tl.box));
    }
    { const uint32_t b = L.tl.staged ? L.tl.box : BOX_NONE; c->stat_composited_tiles = b == BOX_NONE ? ntiles : (uint64_t)std::min<uint32_t>((((b >> 16) & 255u) - (b & 255u) + 1u) * BOX_BLOCK, (uint32_t)c->tiles_x) * std::min<uint32_t>(((b >> 24) - ((b >> 8) & 255u) + 1u) * BOX_BLOCK, (uint32_t)c->tiles_y); }
    HIPCHK(c, hipEventRecord(L.ev_emit, L.s));         // totals, flags and the longest list are in pinned host memory behind this event (the compositor's first workgroup wrote them)
    return GS4D_OK;
}

// `preprocess` false: the re-run of a draw whose tile lists overflowed (projected records and the sort-index copy are still valid).
// width of a key span in bits (what the depth sort and the compositor's list sort have to look at)
static int span_bits(uint32_t span) { return span == 0xFFFFFFFFu ? 32 : std::max(1, 32 - __builtin_clz(span | 1u)); }

int run_draw(gs4d_ctx* c, const DrawArgs& a, bool preprocess) {
    const int fuse_bits = span_bits(a.fuse_span);          // of the keys a fused draw generates (from the draw's own arguments: the lane's sorter may have been told about a later frame's keys by now)
    Lane& L = c->lanes[a.lane];
    Framebuffer& F = c->fbs[a.fb];
    Buffer* data = getbuf(c, a.data);
    if (!data) return fail(c, GS4D_E_INVALID, "draw: no splat data buffer bound");
    const uint32_t* order = nullptr;
    Buffer* ob = nullptr;
    size_t nrec = 0, npre = 0;
    int premult = 0;
    if (a.quads) { nrec = data->bytes / 288; npre = nrec < a.instances ? nrec : a.instances; premult = 1; }
    else if (a.mode == GS4D_MODE_4D_SORTED) {
        if (!a.regen_order) {
            ob = getbuf(c, a.order);
            if (!ob) return fail(c, GS4D_E_INVALID, "draw: GS4D_MODE_4D_SORTED needs the sort-index buffer at slot 1");
            if (ob->bytes < a.instances * 4) return fail(c, GS4D_E_INVALID, "draw: sort-index buffer smaller than the instance count");
            order = (const uint32_t*)ob->d;
        }
        nrec = data->bytes / 96; npre = nrec;
    } else if (a.mode == GS4D_MODE_4D_DIRECT) { nrec = data->bytes / 96; npre = nrec < a.instances ? nrec : a.instances; }
    else if (a.mode == GS4D_MODE_2D) { nrec = data->bytes / 48; npre = nrec < a.instances ? nrec : a.instances; }
    else return fail(c, GS4D_E_INVALID, "draw: mode does not match the draw call");
    if (a.instances == 0 || nrec == 0) return GS4D_OK;
    if (a.instances >= 0xFFFFFFFFull || nrec >= 0xFFFFFFFFull) return fail(c, GS4D_E_UNSUPPORTED, "draw: more than 2^32-1 instances");

    HIPCHK(c, bin_scratch_reserve(L.s, L.bin, a.instances, (size_t)c->tiles_x * c->tiles_y));
    bool v2 = a.v2 && tile_lists_plan(L.tl, (size_t)c->tiles_x * c->tiles_y, npre, c->slabs, a.keybits, a.key_span);
    if (v2) { HIPCHK(c, tile_lists_reserve(L.s, L.tl, (size_t)c->tiles_x * c->tiles_y, npre)); preprocess = true; order = nullptr; }   // an unordered draw is always re-run from the projection
    L.tl.staged = false; L.tl.scap = L.tl.bcap = 0; L.tl.box = BOX_NONE;
    if (v2) {
        // the statistics of a draw belong to a list geometry (buckets, segments, tiles, records, shard, data buffer): another one starts from scratch
        uint64_t geom = 0xcbf29ce484222325ull;
        for (uint64_t v : { (uint64_t)L.tl.nb, (uint64_t)L.tl.rows, (uint64_t)L.tl.seg, (uint64_t)c->tiles_x, (uint64_t)c->tiles_y, (uint64_t)npre, (uint64_t)a.shard_world, (uint64_t)a.shard_rank, (uint64_t)a.data })
            geom = (geom ^ v) * 0x100000001b3ull;
        const_cast<DrawArgs&>(a).stage_geom = geom;
        if (c->stage_enable && c->stage_known && c->stage_geom == geom && !a.exact_lists && L.tl.seg <= (uint32_t)(STAGE_R * SEG_THREADS)) {
            // margins: an eighth on the fullest segment and on the fullest bucket (the longest run is no capacity of anything any more: statistics only)
            const uint64_t scap = ((uint64_t)c->stage_max_seg + c->stage_max_seg / 8 + 64 + 63) & ~63ull;
            const uint64_t bcap = ((uint64_t)c->stage_max_bucket + c->stage_max_bucket / 8 + 512 + 63) & ~63ull;
            if (scap <= STAGE_MAX_SCAP && bcap <= 32u * 512u && (uint64_t)L.tl.nb * bcap < 0xFFFFFFF0ull) {      // (k_bucket_tiles_staged: a thread holds at most 32 of its bucket's entries)
                HIPCHK(c, tile_lists_reserve_blocks(L.s, L.tl, (size_t)L.tl.rows * scap));
                L.tl.staged = true; L.tl.scap = (uint32_t)scap; L.tl.bcap = (uint32_t)bcap;
                if (++L.tl.seq == 0u) L.tl.seq = 1u;
                // the compositor's launch box: where the last staged draw had entries, stage_box_margin blocks (of 4 x 4 tiles) wider on every side
                const uint32_t nbx = (uint32_t)(c->tiles_x + BOX_BLOCK - 1) / BOX_BLOCK, nby = (uint32_t)(c->tiles_y + BOX_BLOCK - 1) / BOX_BLOCK;
                if (c->stage_box_enable && c->stage_box != BOX_NONE && c->stage_box != BOX_EMPTY && nbx <= 256u && nby <= 256u) {
                    const uint32_t b = c->stage_box, x0 = b & 255u, y0 = (b >> 8) & 255u, x1 = (b >> 16) & 255u, y1 = b >> 24;
                    const uint32_t m = c->stage_box_margin;
                    L.tl.box = box_pack(x0 > m ? x0 - m : 0u, y0 > m ? y0 - m : 0u, std::min(x1 + m, nbx - 1u), std::min(y1 + m, nby - 1u));
                }
            }
        }
    }
    uint32_t* order_copy = nullptr;
    const bool regen = a.regen_order && !v2 && a.mode == GS4D_MODE_4D_SORTED && !a.quads;
    if (order || regen) {
        if (L.order_cap < a.instances) {
            HIPCHK(c, hipStreamSynchronize(L.s));
            if (L.order_copy) (void)hipFree(L.order_copy);
            L.order_copy = nullptr; L.order_cap = 0;
            HIPCHK(c, hipMalloc(&L.order_copy, a.instances * 4));
            L.order_cap = a.instances;
        }
        if (regen) order = L.order_copy;                   // filled below (first re-run) or by an earlier re-run of this draw
        else if (preprocess) order_copy = L.order_copy;   // first run: the emit kernel reads the caller's buffer and keeps a copy
        else order = L.order_copy;                         // re-run: the caller's buffer may have been overwritten since
    }
    if (preprocess) {
        if (L.proj_cap < npre) {
            HIPCHK(c, hipStreamSynchronize(L.s));
            if (L.proj) (void)hipFree(L.proj);
            if (L.trects) (void)hipFree(L.trects);
            L.proj = nullptr; L.trects = nullptr; L.proj_cap = 0;
            HIPCHK(c, hipMalloc(&L.proj, npre * 64)); HIPCHK(c, hipMalloc(&L.trects, npre * 4));
            L.proj_cap = npre;
        }
        if (!a.quads && (a.mode == GS4D_MODE_4D_SORTED || a.mode == GS4D_MODE_4D_DIRECT)) { int rc = ensure_soa(c, *data); if (rc) return rc; }
        { int rc = lane_access(c, *data, false); if (rc) return rc; data->rd_mask |= 1u << a.lane; }
        if (ob && !v2) { int rc = lane_access(c, *ob, false); if (rc) return rc; ob->rd_mask |= 1u << a.lane; }
        {
            StageTimer t(c, GS4D_T_PREPROCESS);
            const PreOut po = { L.proj, (v2 && L.tl.staged) ? nullptr : L.trects };
            L.trects_in_order = false;
            TileCount tc;
            if (v2 && L.tl.staged) { tc.stage_out = L.tl.blocks; tc.scap = L.tl.scap; tc.offs = L.tl.hist + L.tl.hist_cap; tc.abort_word = L.bin.total + TL_ABORT_WORD; tc.seq = L.tl.seq; }
            if (v2) { tc.sstat = L.tl.sstat; tc.hist = L.tl.hist; tc.skey = L.tl.skey; tc.nb = L.tl.nb; tc.seg = L.tl.seg; tc.rows = L.tl.rows; tc.tiles_x = c->tiles_x; tc.shard_rank = a.shard_rank; tc.shard_world = a.shard_world; tc.ks = a.ks; }
            if (a.fuse) {
                tc.ks = a.ks;
                Buffer* K = getbuf(c, a.fuse_keys); Buffer* I = getbuf(c, a.fuse_idx);
                if (!K || !I) return fail(c, GS4D_E_INVALID, "draw: the key buffers of the queued key generation have been deleted");
                hipError_t he = hipSuccess;
                uint32_t* kh = sort_hist_slot(L.s, L.depth_sort, npre, &he);
                if (!kh) return hipfail(c, he, "sort_hist_slot");
                tc.keys_out = (float*)K->d; tc.idx_out = nullptr /* the depth sort below makes the identity index up */; (void)I; tc.ghist = kh; tc.span = a.fuse_span; tc.err = L.host_total_dev + 4;
                L.depth_sort.hist_bias = a.ks.bias;
                tc.hist_rb = L.depth_sort.hist_rb = sort_plan_rb(L.depth_sort, npre, fuse_bits);
            }
            if (a.quads) HIPCHK(c, launch_preprocess_3d(L.s, (const float*)data->d, npre, a.u, c->W, c->H, po, tc));
            else if (a.mode == GS4D_MODE_2D) HIPCHK(c, launch_preprocess_2d(L.s, (const float*)data->d, npre, a.u, c->W, c->H, po, tc));
            else {
                HIPCHK(c, launch_preprocess_4d(L.s, data->soa, data->soa_n, data->soa_info, npre, a.u, c->W, c->H, po, tc));
                c->stat_shadow_bytes = data->soa_info.layout == SOA_STATIC3D ? 64 : data->soa_info.layout == SOA_SYM ? 72 : 96;
            }
        }
        L.proj_n = npre;
        if (regen) {
            // "records in ascending (depth key, record index)" — what the caller's index held when the draw was issued — from the key source
            // the draw carries: k_keygen + the stable sort, into the lane's own buffers
            if (L.regen_cap < npre) {
                HIPCHK(c, hipStreamSynchronize(L.s));
                if (L.regen_keys) (void)hipFree(L.regen_keys);
                L.regen_keys = nullptr; L.regen_cap = 0;
                HIPCHK(c, hipMalloc(&L.regen_keys, npre * 4));
                L.regen_cap = npre;
            }
            hipError_t he = hipSuccess;
            uint32_t* kh = sort_hist_slot(L.s, L.depth_sort, npre, &he);
            if (!kh) return hipfail(c, he, "sort_hist_slot");
            const float cam[3] = { a.ks.camx, a.ks.camy, a.ks.camz };
            float view[16] = { 0 }; view[2] = a.ks.vr0; view[6] = a.ks.vr1; view[10] = a.ks.vr2; view[14] = a.ks.vr3;
            L.depth_sort.hist_rb = sort_plan_rb(L.depth_sort, npre, a.keybits);
            HIPCHK(c, launch_keygen(L.s, data->soa, soa_sig3(data->soa, data->soa_n, data->soa_info), data->soa_info, npre, a.ks.t, cam, view, a.ks.mode == KEYSRC_VIEWZ ? GS4D_KEY_VIEW_Z : GS4D_KEY_REF_INV_EUCLID,
                                    (float*)L.regen_keys, L.order_copy, kh, L.depth_sort.hist_rb, a.ks.bias, 0xFFFFFFFFu, L.host_total_dev + 4));
            L.depth_sort.hist_bias = a.ks.bias;
            HIPCHK(c, radix_sort_pairs(L.s, L.depth_sort, L.regen_keys, L.order_copy, npre, nullptr, a.keybits, true));
        }
        { int rc = fb_access(c, F); if (rc) return rc; }
        if (a.fuse && !v2) {
            // ordered path: the sort the application asked for runs here, between the projection (which wrote its keys and digit
            // histograms) and the binning (which reads the sorted index)
            Buffer* K = getbuf(c, a.fuse_keys); Buffer* I = getbuf(c, a.fuse_idx);
            StageTimer t(c, GS4D_T_SORT);
            HIPCHK(c, radix_sort_pairs(L.s, L.depth_sort, (uint32_t*)K->d, (uint32_t*)I->d, npre, nullptr, fuse_bits, true, true));
        }
    }
    size_t want = a.instances * 2 + 65536;
    if (want < c->stat_entries + c->stat_entries / 2) want = c->stat_entries + c->stat_entries / 2;
    if (v2 && L.tl.staged && want < (size_t)L.tl.nb * L.tl.bcap) want = (size_t)L.tl.nb * L.tl.bcap;      // staged: the tile-ordered array holds a region of bcap entries per bucket
    if (L.pair_cap < want) { int rc = ensure_pairs(c, L, want); if (rc) return rc; }
    if (v2) {
        int rc = enqueue_raster_v2(c, L, F, a, npre, premult);
        if (rc == GS4D_OK && a.fuse) {
            // the depth sort the application asked for: its keys and digit histograms came out of the projection kernel; nothing in this draw
            // waits for it (the draw took its order from the keys), it fills the caller's buffers for whoever reads them next
            Buffer* K = getbuf(c, a.fuse_keys); Buffer* I = getbuf(c, a.fuse_idx);
            StageTimer t(c, GS4D_T_SORT);
#ifdef GS4D_TUNING
            static const bool skip_sort = getenv("GS4D_ABLATE_SORT") != nullptr;      // ablation (make TUNING=1): what the frame costs without its depth sort — an upper bound for any re-scheduling of it
            if (skip_sort) { L.depth_sort.hist_pending = false; L.depth_sort.flip ^= 1; return rc; }
#endif
            HIPCHK(c, radix_sort_pairs(L.s, L.depth_sort, (uint32_t*)K->d, (uint32_t*)I->d, npre, nullptr, fuse_bits, true, true));
        }
        return rc;
    }
    return enqueue_raster(c, L, F, a, order, order_copy, a.instances, npre, premult);
}

// A draw's tile-list capacity is validated after the fact: the entry count comes back through pinned memory behind an event.
// Called by every entry point that could observe the draw's result.  On overflow the raster stages are re-run with exact capacity
// (the projected records and the copy of the sort index are still valid).
int resolve_lane(gs4d_ctx* c, int li) {
    Lane& L = c->lanes[li];
    while (L.pending) {
        HIPCHK(c, hipEventSynchronize(L.ev_emit));     // the entry count is final once the binning kernel (ordered path) / the tile scan (unordered path) has run
        L.pending = false;
        const bool discarded = L.discarded;             // the image was cleared before anybody looked: learn from the draw, do not repeat it
        L.discarded = false;
        if (L.host_total[4]) return fail(c, GS4D_E_DEVICE, DEVICE_CHECK_MSG);
        const uint64_t total = (uint64_t)L.host_total[2] | ((uint64_t)L.host_total[3] << 32);
        const uint32_t flags = L.host_total[1];
        if (L.pending_args.v2 && !(flags & 1u)) {
            // longest (bucket, segment) run and fullest bucket of this draw: what sizes the staged blocks of the draws that follow
            c->stage_max_run = L.host_total[6]; c->stage_max_bucket = L.host_total[7]; c->stage_max_seg = L.host_total[8]; c->stage_geom = L.pending_args.stage_geom; c->stage_known = true;
            c->stage_box = L.host_total[9];
            const uint32_t u = L.host_total[9];
            if ((flags & 4u) && L.tl.box != BOX_NONE && u != BOX_NONE && u != BOX_EMPTY && !(box_holds(L.tl.box, u & 255u, (u >> 8) & 255u) && box_holds(L.tl.box, (u >> 16) & 255u, u >> 24)))
                c->stage_box_margin = std::min(16u, c->stage_box_margin * 2u);      // the picture moves faster than the margin allowed
        }
        if (L.pending_args.v2) {
            c->stat_longest = L.host_total[5];
            // the compositor's occupancy falls with the list capacity it is launched for: give capacity back when the lists stay short
            const uint32_t fit = v2_list_capacity(std::min<uint32_t>(V2_MAX_LIST, L.host_total[5] + L.host_total[5] / 8u));
            if (!flags && fit < c->list_hint) { if (++c->shrink_votes >= 8) { c->list_hint = fit; c->shrink_votes = 0; } } else c->shrink_votes = 0;
        }
        if (!flags) { c->stat_entries = total; break; }
        if (total >= 0xFFFFFFF0ull) { if (discarded) break; return fail(c, GS4D_E_UNSUPPORTED, "draw: more than 2^32 tile-list entries (splats cover too many tiles)"); }
        if (discarded) c->stat_aborted_discarded++; else c->stat_reruns++;
        c->stat_entries = total;
        const bool was_v2 = L.pending_args.v2;     // an unordered draw kept no copy of its sort index: whatever path the re-run takes, it starts from the projection
        if (L.pending_args.v2 && (flags & 4u)) { L.pending_args.exact_lists = true; c->stat_staged_misses++; }      // a run or a bucket did not fit the guess: exact lists this time
        if (L.pending_args.v2 && (flags & 2u)) {
            // A (sub-)list longer than the compositing wave was launched for.  What it can hold is a launch parameter (64 entries per lane
            // register: v2_list_capacity) that costs registers and LDS.  Up to V2_MAX_LIST entries: a capacity that fits.  Longer: this is a
            // scene for the instance-ordered path (the re-run regenerates the order the draw was issued with: DrawArgs::regen_order).
            const uint32_t longest = L.host_total[5];
            if (longest <= V2_MAX_LIST) c->list_hint = std::max(c->list_hint, v2_list_capacity(longest + longest / 8u));
            else { c->long_lists = true; c->ordered_draws = 0; L.pending_args.v2 = false; L.pending_args.regen_order = true; }
        }
        int rc = ensure_pairs(c, L, (size_t)(total + total / 8 + 1024));
        if (rc) return rc;
        if (discarded) break;                           // the lane's next draw starts with what this one found out
        // the re-run goes to the draw's own lane: make it current while its kernels are queued
        const int saved = c->cur;
        c->cur = li;
        rc = run_draw(c, L.pending_args, was_v2);
        // Other lanes order themselves after this lane through its tail event (fb_access, lane_access), which was recorded when the lane was
        // left — before this re-run.  Record it again so that it keeps covering everything queued on the lane.
        if (rc == GS4D_OK && li != saved) { hipError_t he = hipEventRecord(L.ev_tail, L.s); if (he != hipSuccess) { c->cur = saved; return hipfail(c, he, "hipEventRecord"); } }
        c->cur = saved;
        if (rc) return rc;
        L.pending = true;
    }
    return GS4D_OK;
}

int flush_order(gs4d_ctx* c);

// every lane (calls that observe or tear down everything)
int resolve_pending(gs4d_ctx* c) {
    { int rc = flush_order(c); if (rc) return rc; }
    for (int i = 0; i < c->nlanes; ++i) { int rc = resolve_lane(c, i); if (rc) return rc; }
    return GS4D_OK;
}

// the draws that rendered into image `fb` (calls that observe that image)
int resolve_image(gs4d_ctx* c, int fb) {
    for (int i = 0; i < c->nlanes; ++i)
        if (c->lanes[i].pending && c->lanes[i].pending_args.fb == fb) { int rc = resolve_lane(c, i); if (rc) return rc; }
    return GS4D_OK;
}

int host_access(gs4d_ctx* c, Buffer& B) {
    if (B.touch <= c->synced) return GS4D_OK;
    int rc = resolve_pending(c); if (rc) return rc;
    rc = sync_all(c); if (rc) return rc;
    B.wr_lane = -1; B.rd_mask = 0; B.tail_mask = 0; B.ordered_mask = 0;
    return GS4D_OK;
}

// Launch a queued gs4d_keygen (+ gs4d_sort_pairs) as kernels of their own: somebody is about to look at the buffers, or the draw that
// follows cannot use them.  The lane has not changed since they were queued (only frame-starting calls change it, and they flush first).
int flush_order(gs4d_ctx* c) {
    if (!c->po.keygen) return GS4D_OK;
    auto po = c->po;
    c->po.keygen = c->po.sorted = false;
    Lane& L = c->lanes[po.lane];
    Buffer* D = getbuf(c, po.data); Buffer* K = getbuf(c, po.keys); Buffer* I = getbuf(c, po.idx);
    if (!D || !K || !I || !D->soa) return fail(c, GS4D_E_INVALID, "queued keygen: a buffer it names has been deleted");
    {
        hipError_t he = hipSuccess;
        uint32_t* kh = sort_hist_slot(L.s, L.depth_sort, po.n, &he);
        if (!kh) return hipfail(c, he, "sort_hist_slot");
        StageTimer tm(c, GS4D_T_KEYGEN);
        L.depth_sort.hist_rb = sort_plan_rb(L.depth_sort, po.n, span_bits(po.span));
        HIPCHK(c, launch_keygen(L.s, D->soa, soa_sig3(D->soa, D->soa_n, D->soa_info), D->soa_info, po.n, po.t, po.cam, po.view, po.key_mode, (float*)K->d, (uint32_t*)I->d, kh, L.depth_sort.hist_rb, po.bias, po.span, L.host_total_dev + 4));
        L.depth_sort.hist_bias = po.bias;
    }
    if (po.sorted) {
        StageTimer t(c, GS4D_T_SORT);
        HIPCHK(c, radix_sort_pairs(L.s, L.depth_sort, (uint32_t*)K->d, (uint32_t*)I->d, po.n, nullptr, span_bits(po.span), true));
    }
    return GS4D_OK;
}

int alloc_fbs(gs4d_ctx* c, int w, int h) {
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535) return fail(c, GS4D_E_INVALID, "framebuffer size must be 1..65535");
    { int rc = sync_all(c); if (rc) return rc; }
    for (int i = 0; i < c->nlanes; ++i) {
        if (c->fbs[i].mem) { (void)hipFree(c->fbs[i].mem); c->fbs[i].mem = nullptr; }
        if (c->fbs[i].tstate) { (void)hipFree(c->fbs[i].tstate); c->fbs[i].tstate = nullptr; }
        if (c->fbs[i].linecnt) { (void)hipFree(c->fbs[i].linecnt); c->fbs[i].linecnt = nullptr; }
        HIPCHK(c, hipMalloc(&c->fbs[i].mem, (size_t)w * h * 16));
        const size_t nt = (size_t)((w + TILE - 1) / TILE) * ((h + TILE - 1) / TILE);
        HIPCHK(c, hipMalloc(&c->fbs[i].tstate, nt * 4));
        HIPCHK(c, hipMemset(c->fbs[i].tstate, 0, nt * 4));      // no tile is in memory: epochs start at 1
        c->fbs[i].epoch = 1; c->fbs[i].all_in_memory = false;
        c->fbs[i].is_clear = true; c->fbs[i].last_lane = -1; memcpy(c->fbs[i].clear, c->clear, 16);
    }
    c->W = w; c->H = h; c->tiles_x = (w + TILE - 1) / TILE; c->tiles_y = (h + TILE - 1) / TILE;
    c->cur_fb = c->cur; c->prev_fb = -1;
    return GS4D_OK;
}

} // namespace

namespace {
// ---- frame-lane streams on hardware queues of their own ----
// HIP maps streams onto a few hardware queues (four by default) in an order that depends on every stream the process has created so far;
// two streams on one queue run their kernels one after the other.  Two frame lanes that share a queue do not overlap — measured: 0.110 ->
// 0.122 ms/frame at 10^6 splats when ONE foreign stream (a communicator's, a framework's) was alive while the context was created, or a
// context with another number of lanes had existed before (tools/order_effect.py, tools/queue_probe.hip).  So the lanes' streams are chosen
// by experiment: a candidate stream is accepted if a short kernel on it runs WHILE a spinning kernel occupies each stream accepted so far.
__global__ void k_lane_probe_spin(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }                // bounded: the 100 MHz counter always advances
    out[0] = wall_clock64();
}
__global__ void k_lane_probe_stamp(unsigned long long* out) { out[0] = wall_clock64(); }

// false also when anything fails: the caller then simply keeps the stream
static bool streams_run_concurrently(hipStream_t a, hipStream_t b, unsigned long long* scratch /* device, 2 words */) {
    k_lane_probe_spin<<<dim3(1), dim3(1), 0, a>>>(10000ull /* 100 us of the 100 MHz counter: long against a launch, even under a tracing tool */, scratch);
    k_lane_probe_stamp<<<dim3(1), dim3(1), 0, b>>>(scratch + 1);
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
    unsigned long long h[2] = { 0, 0 };
    if (hipMemcpy(h, scratch, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return false;
    return h[1] < h[0];                                     // the stamp was taken before the spin ended
}

#ifdef GS4D_TUNING
static unsigned long long* g_tuning_spin_out() { static unsigned long long* p = nullptr; if (!p && hipMalloc(&p, 16) != hipSuccess) p = nullptr; return p; }
#endif

static hipError_t create_lane_streams(gs4d_ctx* c) {
    hipError_t e;
    const bool probe = c->nlanes > 1 && !(getenv("GS4D_PROBE_QUEUES") && atoi(getenv("GS4D_PROBE_QUEUES")) == 0);      // test hook: 0 = take the streams as they come
    unsigned long long* scratch = nullptr;
    if (probe && hipMalloc(&scratch, 16) != hipSuccess) scratch = nullptr;
    std::vector<hipStream_t> good, rejected;
    const int max_tries = 3 * c->nlanes + 4;
    for (int t = 0; (int)good.size() < c->nlanes && t < max_tries; ++t) {
        hipStream_t s = nullptr;
        if ((e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess) break;
        bool ok = true;
        // (the test is a race against a 100-us spin: a host stall between the two launches looks like a shared queue — a stream is rejected only if it
        // fails against the same lane twice)
        if (scratch) for (hipStream_t g : good) if (!streams_run_concurrently(g, s, scratch) && !streams_run_concurrently(g, s, scratch)) { ok = false; break; }
        (ok ? good : rejected).push_back(s);
    }
    c->stat_lanes_sharing = 0;
    while ((int)good.size() < c->nlanes && !rejected.empty()) { good.push_back(rejected.back()); rejected.pop_back(); c->stat_lanes_sharing++; }      // fewer hardware queues than lanes: lanes will share
    c->stat_streams_rejected = rejected.size();
    for (hipStream_t s : rejected) (void)hipStreamDestroy(s);
    if (scratch) (void)hipFree(scratch);
    if ((int)good.size() < c->nlanes) { for (hipStream_t s : good) (void)hipStreamDestroy(s); return hipErrorOutOfMemory; }
    for (int i = 0; i < c->nlanes; ++i) c->lanes[i].s = good[i];
    return hipSuccess;
}

} // namespace

extern "C" {

const char* gs4d_version(void) { return "gs4d 0.2 (gfx950)"; }

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
    if (const char* ev = getenv("GS4D_LANES")) { const int v = atoi(ev); if (v >= 1 && v <= MAX_LANES) c->nlanes = v; }     // tuning knob
    if (const char* ev = getenv("GS4D_FUSE_KEYGEN")) c->defer_order = atoi(ev) != 0;                                       // test hook: 0 = launch key generation and sort at once
    if (const char* ev = getenv("GS4D_DRAW_PATH")) { if (!strcmp(ev, "ordered")) c->path_pref = 1; }
    if (const char* ev = getenv("GS4D_RENAME")) c->rename_storage = atoi(ev) != 0;
    if (const char* ev = getenv("GS4D_STAGED_BOX")) c->stage_box_enable = atoi(ev) != 0;                                   // test hook: 0 = the compositor of a staged draw is launched for every tile
    if (const char* ev = getenv("GS4D_STAGED")) c->stage_enable = atoi(ev) != 0;                                          // test hook: 0 = every unordered draw builds its lists exactly (scan + scatter)
    if (const char* ev = getenv("GS4D_SLABS")) { const int v = atoi(ev); if (v >= 1 && v <= (int)V2_MAX_SLABS) { c->slabs = 1; while ((int)c->slabs < v) c->slabs *= 2u; } }      // test hook: depth slabs (a power of two)                         // test hook: instance-ordered tile lists for every draw
    auto bail = [&](int rc) { g_create_error = c->err; gs4d_destroy(c); return rc; };
    if ((e = create_lane_streams(c)) != hipSuccess) return bail(hipfail(c, e, "hipStreamCreate"));
    for (int i = 0; i < c->nlanes; ++i) {
        Lane& L = c->lanes[i];
        for (hipEvent_t* ev : { &L.ev_emit, &L.ev_tail }) {
            if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return bail(hipfail(c, e, "hipEventCreate"));
            if ((e = hipEventRecord(*ev, L.s)) != hipSuccess) return bail(hipfail(c, e, "hipEventRecord"));      // "already happened"
        }
        if ((e = hipHostMalloc((void**)&L.host_total, 64, hipHostMallocMapped)) != hipSuccess) return bail(hipfail(c, e, "hipHostMalloc"));
        memset(L.host_total, 0, 64);
        if ((e = hipHostGetDevicePointer((void**)&L.host_total_dev, L.host_total, 0)) != hipSuccess) return bail(hipfail(c, e, "hipHostGetDevicePointer"));
        L.depth_sort.err = L.pair_sort.err = L.host_total_dev + 4;
    }
    for (hipEvent_t* ev : { &c->ev_user, &c->ev_readback }) {
        if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return bail(hipfail(c, e, "hipEventCreate"));
    }
    {
        bool ordered = false;
        hipStream_t ls[MAX_LANES];
        for (int i = 0; i < c->nlanes; ++i) ls[i] = c->lanes[i].s;
        if ((e = lds_atomic_order_selftest(ls, c->nlanes, &ordered)) != hipSuccess) return bail(hipfail(c, e, "lds_atomic_order_selftest"));      // on every lane at once: beside other waves on the CUs
        c->atomic_rank = ordered;
    }
    const int shape_knob = getenv("GS4D_SORT_SHAPE") ? atoi(getenv("GS4D_SORT_SHAPE")) : 0, rank_knob = getenv("GS4D_SORT_RANK") ? atoi(getenv("GS4D_SORT_RANK")) : 0;
    for (int i = 0; i < c->nlanes; ++i) {
        for (SortScratch* ss : { &c->lanes[i].depth_sort, &c->lanes[i].pair_sort }) { ss->atomic_rank = c->atomic_rank; ss->shape_knob = shape_knob; ss->rank_knob = rank_knob; ss->rb_knob = getenv("GS4D_SORT_RB") ? atoi(getenv("GS4D_SORT_RB")) : 0; }
    }
    int rc = alloc_fbs(c, width, height);
    if (rc) return bail(rc);
    *out = c;
    return GS4D_OK;
}

void gs4d_destroy(gs4d_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (int i = 0; i < MAX_LANES; ++i) if (c->lanes[i].s) (void)hipStreamSynchronize(c->lanes[i].s);
    for (auto& b : c->bufs) { if (b.d) (void)hipFree(b.d); if (b.soa) (void)hipFree(b.soa); if (b.bbox_dev) (void)hipFree(b.bbox_dev); if (b.ev_fill) (void)hipEventDestroy(b.ev_fill); }
    for (int i = 0; i < MAX_LANES; ++i) {
        Lane& L = c->lanes[i];
        if (c->fbs[i].mem) (void)hipFree(c->fbs[i].mem);
        if (c->fbs[i].tstate) (void)hipFree(c->fbs[i].tstate);
        if (c->fbs[i].linecnt) (void)hipFree(c->fbs[i].linecnt);
        if (L.line_verts) (void)hipFree(L.line_verts);
        if (L.order_copy) (void)hipFree(L.order_copy);
        if (L.regen_keys) (void)hipFree(L.regen_keys);
        for (auto& sp : L.spare) { if (sp.d) (void)hipFree(sp.d); for (hipEvent_t e : sp.ev) if (e) (void)hipEventDestroy(e); }
        if (L.proj) (void)hipFree(L.proj);
        if (L.trects) (void)hipFree(L.trects);
        if (L.pair_keys) (void)hipFree(L.pair_keys);
        tile_lists_free(L.tl);
        sort_scratch_free(L.depth_sort); sort_scratch_free(L.pair_sort); bin_scratch_free(L.bin);
        if (L.host_total) (void)hipHostFree(L.host_total);
        if (L.ev_emit) (void)hipEventDestroy(L.ev_emit);
        if (L.ev_tail) (void)hipEventDestroy(L.ev_tail);
        if (L.s) (void)hipStreamDestroy(L.s);
    }
    for (auto e : c->ev0) if (e) (void)hipEventDestroy(e);
    for (auto e : c->ev1) if (e) (void)hipEventDestroy(e);
    if (c->ev_user) (void)hipEventDestroy(c->ev_user);
    if (c->ev_readback) (void)hipEventDestroy(c->ev_readback);
    delete c;
}

int gs4d_resize(gs4d_ctx* c, int width, int height) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = resolve_pending(c); if (rc) return rc;
    if (width == c->W && height == c->H) return GS4D_OK;
    return alloc_fbs(c, width, height);
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
        if (data) { hipError_t e = hipMemcpy(nb.d, data, bytes, hipMemcpyHostToDevice); if (e != hipSuccess) { (void)hipFree(nb.d); return hipfail(c, e, "buffer upload"); } }
    }
    nb.bytes = bytes; nb.alive = true; nb.version = 1;
    c->bufs[name] = nb;
    *out = name;
    return GS4D_OK;
}

int gs4d_buffer_subdata(gs4d_ctx* c, gs4d_buf b, size_t offset, const void* data, size_t bytes) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    { int rcq = flush_order(c); if (rcq) return rcq; }
    Buffer* B = getbuf(c, b);
    if (!B) return fail(c, GS4D_E_INVALID, "buffer_subdata: bad buffer name");
    if (offset > B->bytes || bytes > B->bytes - offset) return fail(c, GS4D_E_INVALID, "buffer_subdata: range outside the buffer");   // GL_INVALID_VALUE
    if (!bytes) return GS4D_OK;
    if (!data) return fail(c, GS4D_E_INVALID, "buffer_subdata: data == NULL");
    { int rc = host_access(c, *B); if (rc) return rc; }
    // the caller keeps ownership of `data` and may reuse it on return (glBufferSubData semantics): copy synchronously
    HIPCHK(c, hipMemcpy((char*)B->d + offset, data, bytes, hipMemcpyHostToDevice));
    B->version++;
    return GS4D_OK;
}

int gs4d_buffer_read(gs4d_ctx* c, gs4d_buf b, size_t offset, void* out, size_t bytes) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    { int rcq = flush_order(c); if (rcq) return rcq; }
    Buffer* B = getbuf(c, b);
    if (!B) return fail(c, GS4D_E_INVALID, "buffer_read: bad buffer name");
    if (offset > B->bytes || bytes > B->bytes - offset || (!out && bytes)) return fail(c, GS4D_E_INVALID, "buffer_read: range outside the buffer");
    if (!bytes) return GS4D_OK;
    // after the kernels that wrote it: they sit on the current lane, or on the lane recorded in the buffer
    HIPCHK(c, hipStreamSynchronize(lane(c).s));
    if (B->wr_lane >= 0 && B->wr_lane != c->cur) HIPCHK(c, hipStreamSynchronize(c->lanes[B->wr_lane].s));
    HIPCHK(c, hipMemcpy(out, (const char*)B->d + offset, bytes, hipMemcpyDeviceToHost));
    if (device_error(c)) return fail(c, GS4D_E_DEVICE, DEVICE_CHECK_MSG);
    return GS4D_OK;
}

int gs4d_buffer_destroy(gs4d_ctx* c, gs4d_buf b) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    { int rcq = flush_order(c); if (rcq) return rcq; }
    Buffer* B = getbuf(c, b);
    if (!B) return GS4D_OK;                         // 0, unknown or already deleted: silently ignored, like glDeleteBuffers
    { int rc = resolve_pending(c); if (rc) return rc; rc = sync_all(c); if (rc) return rc; }
    if (B->d) (void)hipFree(B->d);
    if (B->soa) (void)hipFree(B->soa);
    if (B->bbox_dev) (void)hipFree(B->bbox_dev);
    if (B->ev_fill) (void)hipEventDestroy(B->ev_fill);
    *B = Buffer();
    for (auto& s : c->slots) if (s == b) s = 0;     // a deleted buffer is unbound
    for (int i = 0; i < c->nlanes; ++i) if (c->lanes[i].kg_buf == b) c->lanes[i].kg_buf = 0;
    return GS4D_OK;
}

int gs4d_buffer_device_ptr(gs4d_ctx* c, gs4d_buf b, void** dptr, size_t* bytes) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    { int rcq = flush_order(c); if (rcq) return rcq; }
    Buffer* B = getbuf(c, b);
    if (!B) return fail(c, GS4D_E_INVALID, "buffer_device_ptr: bad buffer name");
    if (dptr) *dptr = B->d;
    if (bytes) *bytes = B->bytes;
    B->ptr_exposed = true;                          // the address is the caller's to keep: this storage stays with this name
    B->version++;                                   // the caller may write through the pointer
    return GS4D_OK;
}

int gs4d_buffer_invalidate(gs4d_ctx* c, gs4d_buf b) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    { int rcq = flush_order(c); if (rcq) return rcq; }
    Buffer* B = getbuf(c, b);
    if (!B) return fail(c, GS4D_E_INVALID, "buffer_invalidate: bad buffer name");
    int rc = resolve_pending(c); if (rc) return rc;           // a draw that has to be re-run reads the old contents: settle those first
    if (B->touch > c->synced) {
        if (c->user) {
            // the caller's stream waits for everything queued on the lanes so far (a superset of the kernels that use this buffer)
            for (int i = 0; i < c->nlanes; ++i) {
                HIPCHK(c, hipEventRecord(c->lanes[i].ev_tail, c->lanes[i].s));
                HIPCHK(c, hipStreamWaitEvent(c->user, c->lanes[i].ev_tail, 0));
            }
        } else { rc = sync_all(c); if (rc) return rc; }
    }
    B->version++;                                             // SoA shadow, key bounds, histogram hand-off and sort-index provenance all compare against it
    B->prov_valid = false;
    if (c->user) { B->fill_mask = (1u << c->nlanes) - 1u; B->fill_recorded = false; }      // whoever uses it next waits for the caller's stream
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
    memcpy(c->clear, rgba, 16);          // glClearColor does not touch pixels: an image that is (lazily) clear keeps the colour it was cleared with (Framebuffer::clear)
    return GS4D_OK;
}
int gs4d_set_blend(gs4d_ctx* c, int src, int dst) {
    if (!c) return GS4D_E_INVALID;
    auto known = [](int f) { return f == GS4D_ZERO || f == GS4D_ONE || (f >= GS4D_SRC_COLOR && f <= GS4D_ONE_MINUS_DST_COLOR) || (f >= GS4D_CONSTANT_COLOR && f <= GS4D_ONE_MINUS_CONSTANT_ALPHA); };
    if (!known(src) || !known(dst)) return fail(c, GS4D_E_INVALID, "set_blend: not a glBlendFunc factor of the reference's menu (GL_INVALID_ENUM)");
    c->blend_src = src; c->blend_dst = dst;              // like the GL's: state for the draws that follow
    return GS4D_OK;
}
int gs4d_clear(gs4d_ctx* c) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = next_frame_if_drawn(c); if (rc) return rc;
    // the new frame renders into the current lane's own framebuffer (a swap chain with one image per lane); the clear itself is
    // lazy: the compositing kernel starts from the clear colour instead of reading the pixels.  The image left behind stays intact
    // (and readable: gs4d_read_frame_*) until its lane comes round again.
    if (c->cur != c->cur_fb) c->prev_fb = c->cur_fb;
    else if (c->nlanes == 1) c->prev_fb = -1;
    c->cur_fb = c->cur;
    {
        Framebuffer& F = c->fbs[c->cur_fb];
        F.is_clear = true; F.all_in_memory = false;
        if (++F.epoch == 0u) {                              // the 32-bit epoch wraps: forget every old tile word (the lane that used the image last has long finished)
            HIPCHK(c, fb_access(c, F) == GS4D_OK ? hipMemsetAsync(F.tstate, 0, (size_t)c->tiles_x * c->tiles_y * 4, lane(c).s) : hipErrorUnknown);
            F.epoch = 1;
        }
    }
    memcpy(c->fbs[c->cur_fb].clear, c->clear, 16);
    // Whatever a still-unvalidated draw left in the image that is being cleared is discarded with it — the draw is never re-run — but its
    // verdict is still read (resolve_lane, at the latest when its lane draws again): what it found out about the scene's lists feeds the next
    // draw, and a draw that had aborted on the device is counted (gs4d_get_stats: a frame loop that never reads back can prove its frames complete).
    for (int i = 0; i < c->nlanes; ++i) if (c->lanes[i].pending && c->lanes[i].pending_args.fb == c->cur_fb) c->lanes[i].discarded = true;
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
    if (c->po.keygen && !c->po.sorted && keys == c->po.keys && vals == c->po.idx && n == c->po.n && c->cur == c->po.lane && !lane(c).drawn) {
        // the sort of a queued key generation's own output: queued with it; what the index will have been sorted by is known now
        Lane& Lq = lane(c);
        { int rc = lane_access(c, *K, true); if (rc) return rc; rc = lane_access(c, *V, true); if (rc) return rc; }
        c->po.sorted = true;
        c->stat_depth_passes = (uint64_t)sort_plan_passes(Lq.depth_sort.hist_bits, sort_plan_rb(Lq.depth_sort, n, Lq.depth_sort.hist_bits));
        K->version++; V->version++;
        V->prov_valid = true; V->prov_data = Lq.kg_data; V->prov_data_ver = Lq.kg_data_ver; V->prov_ver = V->version; V->prov_n = n; V->prov_bits = Lq.kg_bits; V->prov_ks = Lq.kg_ks; V->prov_span = Lq.kg_span;
        return GS4D_OK;
    }
    { int rc = flush_order(c); if (rc) return rc; rc = next_frame_if_drawn(c); if (rc) return rc; }
    { int rc = lane_access(c, *K, true); if (rc) return rc; rc = lane_access(c, *V, true); if (rc) return rc; }
    Lane& L = lane(c);
    // k_keygen leaves the digit histograms of the keys it wrote: no histogram launch when this sort is of exactly those keys
    const bool have_hist = L.depth_sort.hist_pending && keys == L.kg_buf && K->version == L.kg_ver && n == L.kg_n;
    const int key_bits = have_hist ? L.depth_sort.hist_bits : 32;
    c->stat_depth_passes = (uint64_t)sort_plan_passes(key_bits, have_hist ? L.depth_sort.hist_rb : sort_plan_rb(L.depth_sort, n, key_bits));
    // ... and when the payload is the identity index the same call wrote, the sorted payload is "the records in ascending (key, index)"
    const bool identity_payload = have_hist && vals == L.kg_idx && V->version == L.kg_idx_ver;
    StageTimer t(c, GS4D_T_SORT);
    HIPCHK(c, radix_sort_pairs(L.s, L.depth_sort, (uint32_t*)K->d, (uint32_t*)V->d, n, nullptr, key_bits, have_hist));
    K->version++; V->version++;
    V->prov_valid = identity_payload;
    if (identity_payload) { V->prov_data = L.kg_data; V->prov_data_ver = L.kg_data_ver; V->prov_ver = V->version; V->prov_n = n; V->prov_bits = L.kg_bits; V->prov_ks = L.kg_ks; V->prov_span = L.kg_span; }
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
    { int rc = flush_order(c); if (rc) return rc; rc = next_frame_if_drawn(c); if (rc) return rc; }
    int rc = ensure_soa(c, *D); if (rc) return rc;
    {
        // both outputs are overwritten entirely: if another lane's frame still uses their storage, take this lane's spare storage instead of waiting (Lane::spare)
        Lane& Lr = lane(c);
        auto busy_elsewhere = [&](const Buffer& B) { return (B.wr_lane >= 0 && B.wr_lane != c->cur) || ((B.rd_mask | B.tail_mask) & ~(1u << c->cur)) != 0u; };
        auto renamable = [&](const Buffer& B) { return !B.ptr_exposed && B.bytes == n * 4 && B.fill_mask == 0u && B.touch > c->synced; };
        if (c->rename_storage && c->nlanes > 1 && K != I && K != D && I != D && renamable(*K) && renamable(*I) && (busy_elsewhere(*K) || busy_elsewhere(*I))) {
            Buffer* out[2] = { K, I };
            bool ok = true;
            for (int k = 0; k < 2 && ok; ++k) {
                Lane::Spare& sp = Lr.spare[k];
                if (sp.d && sp.bytes != n * 4) {            // another size than last time: the old spare is given back once nothing can still use it
                    if (sp.touch > c->synced) { rc = sync_all(c); if (rc) return rc; }
                    (void)hipFree(sp.d);
                    sp.d = nullptr; sp.bytes = 0; sp.touch = 0; sp.wait_mask = 0u;      // the events stay with the spare (gs4d_destroy destroys them): resetting the whole struct leaked them
                }
                if (!sp.d) { if (hipMalloc(&sp.d, n * 4) != hipSuccess) { (void)hipGetLastError(); sp.d = nullptr; ok = false; } else sp.bytes = n * 4; }
            }
            if (ok) {
                for (int k = 0; k < 2; ++k) {
                    Buffer& B = *out[k]; Lane::Spare& sp = Lr.spare[k];
                    // the storage coming in: after what used it when it left its buffer (waits capture the events as recorded now)
                    for (int r = 0; r < c->nlanes; ++r) if (((sp.wait_mask >> r) & 1u) && r != c->cur) HIPCHK(c, hipStreamWaitEvent(Lr.s, sp.ev[r], 0));
                    // the storage going out: mark, on every other lane that uses it, the point up to which it does
                    unsigned users = B.rd_mask | B.tail_mask | (B.wr_lane >= 0 ? 1u << B.wr_lane : 0u);
                    users &= ~(1u << c->cur);                // this lane's own earlier uses are ordered by its stream
                    for (int r = 0; r < c->nlanes; ++r) if ((users >> r) & 1u) {
                        if (!sp.ev[r]) HIPCHK(c, hipEventCreateWithFlags(&sp.ev[r], hipEventDisableTiming));
                        HIPCHK(c, hipEventRecord(sp.ev[r], c->lanes[r].s));
                    }
                    sp.wait_mask = users;
                    std::swap(B.d, sp.d);
                    std::swap(B.touch, sp.touch);
                    B.wr_lane = -1; B.ordered_mask = 0; B.rd_mask = 0; B.tail_mask = 0;      // nothing but the waits above stands between this lane and the storage
                    B.touch = ++c->ops;                      // (kernels may still be running on it: a host access has to synchronise)
                }
                c->stat_renamed++;
            }
        }
    }
    { rc = lane_access(c, *D, false); if (rc) return rc; D->tail_mask |= 1u << c->cur; rc = lane_access(c, *K, true); if (rc) return rc; rc = lane_access(c, *I, true); if (rc) return rc; }
    Lane& L = lane(c);
    // A proven lower bound of every key (1 / farthest possible distance, from the bounding box of the records) is subtracted inside
    // the sort's digit extraction: when the keys span less than 2^24 bit patterns above it (camera outside the cloud, far/near < 4)
    // the top digit becomes constant.  When the camera is provably outside the box the same reasoning gives an upper bound, hence
    // the number of key bits above the bias: with <= 24 the sort is launched with three passes instead of four.  k_keygen re-checks
    // both bounds for every key and raises the error word if one does not hold.
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
    // Not launched yet (see gs4d_ctx::po): everything the launch needs is recorded, everything a later call may ask about the buffers
    // (versions, what the index will have been sorted by) is settled now.
    c->po.keygen = true; c->po.sorted = false; c->po.lane = c->cur; c->po.data = data; c->po.keys = keys; c->po.idx = idx; c->po.n = n; c->po.t = t;
    c->po.cam[0] = cam[0]; c->po.cam[1] = cam[1]; c->po.cam[2] = cam[2]; c->po.key_mode = key_mode; c->po.bias = bias; c->po.span = span;
    memcpy(c->po.view, c->u.view, sizeof c->po.view);
    L.depth_sort.hist_bits = span_bits(span);
    K->version++; I->version++;
    L.kg_buf = keys; L.kg_ver = K->version; L.kg_n = n;
    L.kg_idx = idx; L.kg_idx_ver = I->version; L.kg_data = data; L.kg_data_ver = D->version; L.kg_bits = L.depth_sort.hist_bits; L.kg_span = span;
    L.kg_ks.mode = key_mode == GS4D_KEY_REF_INV_EUCLID ? KEYSRC_REF : KEYSRC_VIEWZ;
    L.kg_ks.t = t; L.kg_ks.camx = cam[0]; L.kg_ks.camy = cam[1]; L.kg_ks.camz = cam[2];
    L.kg_ks.vr0 = c->u.view[2]; L.kg_ks.vr1 = c->u.view[6]; L.kg_ks.vr2 = c->u.view[10]; L.kg_ks.vr3 = c->u.view[14];
    L.kg_ks.bias = bias;
    if (!c->defer_order) return flush_order(c);
    return GS4D_OK;
}

// ---- draw ----
static int draw_common(gs4d_ctx* c, DrawArgs& a) {
    (void)hipSetDevice(c->device);
    if (a.quads || a.mode == GS4D_MODE_4D_SORTED || a.mode == GS4D_MODE_4D_DIRECT) {
        // The quad set-up takes uProj * vec4(offset, 0, 1) + ps as "centre + (P00 * x, P11 * y)" at w = 1, which is what the shader computes
        // for a matrix with glm::perspective's sparsity (Camera.cpp:55-58) — the only kind the reference produces.  Anything else is refused
        // rather than drawn differently.
        const float* P = a.u.proj;
        const bool ok = P[1] == 0.0f && P[2] == 0.0f && P[3] == 0.0f && P[4] == 0.0f && P[6] == 0.0f && P[7] == 0.0f && P[8] == 0.0f && P[9] == 0.0f
                     && P[12] == 0.0f && P[13] == 0.0f && P[15] == 0.0f && P[11] != 0.0f && P[0] != 0.0f && P[5] != 0.0f;
        if (!ok) return fail(c, GS4D_E_UNSUPPORTED, "draw: uProj must have the sparsity of glm::perspective (P00, P11, P22, P23, P32 only)");
    }
    // the lane's scratch still belongs to its previous draw, and the image this draw blends onto must be complete: validate those
    // (not the other lanes' draws: their frames are still in flight and nothing here depends on them)
    int rc = resolve_lane(c, c->cur); if (rc) return rc;
    rc = resolve_image(c, c->cur_fb); if (rc) return rc;
    Lane& L = lane(c);
    a.lane = c->cur; a.fb = c->cur_fb;
    memcpy(a.clear, c->fbs[c->cur_fb].clear, 16);
    a.shard_rank = c->shard_rank; a.shard_world = c->shard_world;
    // Which path: the unordered one whenever the blend order is known without reading a sort index — instance k draws record k, or the
    // bound index is this library's sort of its own depth keys for exactly these records — and the lists are short enough to be
    // ordered in LDS (validated on the device; a draw that turns out otherwise is re-run on the ordered path).
    a.v2 = false;
    a.blend_src = c->blend_src; a.blend_dst = c->blend_dst;
    const bool over = a.blend_src == GS4D_SRC_ALPHA && a.blend_dst == GS4D_ONE_MINUS_SRC_ALPHA;       // any other function is applied in draw order: instance-ordered lists
    if (c->atomic_rank && c->path_pref != 1 && over) {
        Buffer* data = getbuf(c, a.data);
        bool ok = false;
        size_t nkeys = 0;
        if (data && (a.quads || a.mode == GS4D_MODE_4D_DIRECT || a.mode == GS4D_MODE_2D)) {
            nkeys = std::min(a.instances, data->bytes / (a.quads ? 288 : a.mode == GS4D_MODE_2D ? 48 : 96));
            a.ks = KeySrc(); a.keybits = 1; while (a.keybits < 32 && ((size_t)1 << a.keybits) < nkeys) ++a.keybits;
            a.key_span = nkeys ? (uint32_t)(nkeys - 1) : 0u;
            ok = nkeys > 0;
        } else if (data && a.mode == GS4D_MODE_4D_SORTED) {
            const Buffer* ob = getbuf(c, a.order);
            if (ob && ob->prov_valid && ob->version == ob->prov_ver && a.data == ob->prov_data && data->version == ob->prov_data_ver && a.instances == ob->prov_n && data->bytes / 96 == ob->prov_n) {
                a.ks = ob->prov_ks; a.keybits = ob->prov_bits; a.key_span = ob->prov_span; ok = true;
            }
        }
        if (ok && (nkeys > V2_MAX_RECORDS || (a.mode == GS4D_MODE_4D_SORTED && a.instances > V2_MAX_RECORDS) || (size_t)c->tiles_x * c->tiles_y > 256u * 1024u)) ok = false;   // tilelist.hip's entry format
        if (ok && c->long_lists) {
            // the lists were too long last time: stay on the ordered path, but probe again now and then if they look short on average
            const uint64_t tiles = (uint64_t)c->tiles_x * c->tiles_y;
            if (++c->ordered_draws >= 64 && c->stat_entries / (tiles ? tiles : 1) <= V2_MAX_LIST / 8) c->long_lists = false; else ok = false;
        }
        a.v2 = ok;
    }
    a.fuse = false;
    if (c->po.keygen) {
        // a queued key generation + sort: executed by this draw if it is the draw they were made for (it takes its blend order from exactly
        // that sort, on the same lane, and the unordered path can run), else launched on their own first
        const Buffer* pd = getbuf(c, a.data);
        bool mine = a.mode == GS4D_MODE_4D_SORTED && !a.quads && c->po.sorted && c->po.idx == a.order && c->po.data == a.data && c->po.lane == c->cur
                 && pd && a.instances == c->po.n && pd->bytes / 96 == c->po.n;
        if (mine && a.v2 && !tile_lists_plan(lane(c).tl, (size_t)c->tiles_x * c->tiles_y, a.instances, c->slabs, a.keybits, a.key_span)) a.v2 = false;
        if (mine) {
            // on the ordered path too: the projection writes the keys, the sort follows it, the binning reads the sorted index
            if (!a.v2) { a.ks = lane(c).kg_ks; a.keybits = lane(c).kg_bits; a.key_span = lane(c).kg_span; }
            a.fuse = true; a.fuse_keys = c->po.keys; a.fuse_idx = c->po.idx; a.fuse_span = c->po.span; c->po.keygen = c->po.sorted = false; c->stat_fused++;
        }
        else { int rc2 = flush_order(c); if (rc2) return rc2; }
    }
    const size_t before = L.proj_n;
    L.proj_n = 0;
    rc = run_draw(c, a, true);
    a.fuse = false;                    // a re-run of this draw finds the keys written and the sort queued
    if (rc) { L.proj_n = before; return rc; }
    if (L.proj_n) { L.pending = true; L.pending_args = a; c->fbs[c->cur_fb].is_clear = false; L.drawn = true; if (a.v2) c->stat_v2_draws++; }   // proj_n != 0 <=> raster work was enqueued
    else L.proj_n = before;
    if (c->profiling) { if (c->prof_frame < gs4d_ctx::PROF_FRAMES && c->prof_tick % (uint64_t)c->prof_every == 0) c->prof_frame++; c->prof_tick++; }
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

// ---- overlay lines (Renderer.cpp:41-215) ----
int gs4d_draw_lines(gs4d_ctx* c, const float* verts, size_t nverts, int dims, int strip, const float viewproj[16], const float rgba[4], float width) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    { int rcq = flush_order(c); if (rcq) return rcq; }
    if (dims != 2 && dims != 3) return fail(c, GS4D_E_INVALID, "draw_lines: dims must be 2 (NDC positions) or 3 (positions transformed by viewproj)");
    if (!rgba || (dims == 3 && !viewproj) || (nverts && !verts)) return fail(c, GS4D_E_INVALID, "draw_lines: NULL argument");
    if (nverts < 2) return GS4D_OK;
    if (nverts > 0x7FFFFFFFull) return fail(c, GS4D_E_UNSUPPORTED, "draw_lines: too many vertices");
    // lines blend into the image in call order: a splat draw into it that still awaits validation (and may be re-run) goes first
    int rc = resolve_image(c, c->cur_fb); if (rc) return rc;
    rc = materialise_fb(c); if (rc) return rc;
    Framebuffer& F = c->fbs[c->cur_fb];
    rc = fb_access(c, F); if (rc) return rc;
    Lane& L = lane(c);
    if (!F.linecnt) {
        HIPCHK(c, hipMalloc(&F.linecnt, (size_t)c->W * c->H * 4));
        HIPCHK(c, hipMemsetAsync(F.linecnt, 0, (size_t)c->W * c->H * 4, L.s));
    }
    const size_t floats = nverts * (size_t)dims;
    if (L.line_cap < floats) {
        HIPCHK(c, hipStreamSynchronize(L.s));
        if (L.line_verts) (void)hipFree(L.line_verts);
        L.line_verts = nullptr; L.line_cap = 0;
        HIPCHK(c, hipMalloc(&L.line_verts, std::max<size_t>(floats, 4096) * 4));
        L.line_cap = std::max<size_t>(floats, 4096);
    }
    HIPCHK(c, hipMemcpyAsync(L.line_verts, verts, floats * 4, hipMemcpyHostToDevice, L.s));     // the caller's array is reusable on return (pageable source)
    LineParams p;
    for (int i = 0; i < 16; ++i) p.vp[i] = viewproj ? viewproj[i] : (i % 5 == 0 ? 1.0f : 0.0f);
    for (int i = 0; i < 4; ++i) p.rgba[i] = std::min(std::max(rgba[i], 0.0f), 1.0f);      // the GL clamps fragment colours before blending into a fixed-point framebuffer
    p.W = c->W; p.H = c->H; p.blend_src = c->blend_src; p.blend_dst = c->blend_dst;
    HIPCHK(c, launch_lines(L.s, L.line_verts, nverts, dims, strip ? 1 : 0, p, width, F.linecnt, F.mem));
    ++c->ops;
    return GS4D_OK;
}

// ---- read-back ----
int gs4d_finish(gs4d_ctx* c) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = resolve_pending(c); if (rc) return rc;
    rc = sync_all(c); if (rc) return rc;
    if (device_error(c)) return fail(c, GS4D_E_DEVICE, DEVICE_CHECK_MSG);
    return GS4D_OK;
}

int gs4d_read_pixels(gs4d_ctx* c, float* rgba, size_t bytes) {
    if (!c || !rgba) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    { int rcq = flush_order(c); if (rcq) return rcq; }
    if (bytes != (size_t)c->W * c->H * 16) return fail(c, GS4D_E_INVALID, "read_pixels: bytes != width*height*16");
    int rc = resolve_image(c, c->cur_fb); if (rc) return rc;
    rc = materialise_fb(c); if (rc) return rc;
    Framebuffer& F = c->fbs[c->cur_fb];
    rc = fb_access(c, F); if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(rgba, F.mem, bytes, hipMemcpyDeviceToHost, lane(c).s));
    HIPCHK(c, hipStreamSynchronize(lane(c).s));
    if (device_error(c)) return fail(c, GS4D_E_DEVICE, DEVICE_CHECK_MSG);
    return GS4D_OK;
}

// frames_back 0: the image the last clear / draw used.  1: the image the last gs4d_clear moved away from (the previous frame of the
// swap chain) — it is packed on the lane that rendered it, behind its compositing kernel, so an application that reads frame f-1
// after queueing frame f never waits for frame f.
static int read_device_common(gs4d_ctx* c, int frames_back, void* dptr, bool rgba8, bool named_event = false, hipEvent_t after = nullptr) {
    if (frames_back != 0 && frames_back != 1) return fail(c, GS4D_E_INVALID, "read_frame: frames_back must be 0 or 1");
    const int fi = frames_back == 0 ? c->cur_fb : c->prev_fb;
    if (fi < 0) return fail(c, GS4D_E_INVALID, "read_frame: no previous image is retained (one frame lane, or no gs4d_clear yet)");
    int rc = resolve_image(c, fi); if (rc) return rc;
    Framebuffer& F = c->fbs[fi];
    const int li = (fi == c->cur_fb || F.last_lane < 0) ? c->cur : F.last_lane;
    Lane& L = c->lanes[li];
    if (named_event) { if (after) HIPCHK(c, hipStreamWaitEvent(L.s, after, 0)); }      // the caller says exactly what the destination has to wait for
    else if (c->user) {                                     // the destination may still be in use by the caller's earlier work
        HIPCHK(c, hipEventRecord(c->ev_user, c->user));
        HIPCHK(c, hipStreamWaitEvent(L.s, c->ev_user, 0));
    }
    if (li == c->cur) { rc = fb_access(c, F); if (rc) return rc; }
    if (rgba8) HIPCHK(c, launch_pack_rgba8(L.s, F.mem, F.tstate, F.epoch, F.clear, c->W, c->H, c->tiles_x, (uint32_t*)dptr));      // lazily clear tiles are packed as the clear colour
    else {
        if (!F.all_in_memory) { HIPCHK(c, launch_fill_unwritten(L.s, F.mem, F.tstate, F.epoch, c->tiles_x, c->tiles_y, c->W, c->H, F.clear)); F.all_in_memory = true; F.is_clear = false; }
        HIPCHK(c, hipMemcpyAsync(dptr, F.mem, (size_t)c->W * c->H * 16, hipMemcpyDeviceToDevice, L.s));
    }
    if (li != c->cur) HIPCHK(c, hipEventRecord(L.ev_tail, L.s));      // the lane's tail event keeps covering everything queued on it
    if (c->user) {                                          // work the caller queues on its stream after this call sees the pixels
        HIPCHK(c, hipEventRecord(c->ev_readback, L.s));
        HIPCHK(c, hipStreamWaitEvent(c->user, c->ev_readback, 0));
    }
    return GS4D_OK;
}

int gs4d_read_pixels_device(gs4d_ctx* c, void* dptr, size_t bytes) {
    if (!c || !dptr) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    { int rcq = flush_order(c); if (rcq) return rcq; }
    if (bytes != (size_t)c->W * c->H * 16) return fail(c, GS4D_E_INVALID, "read_pixels_device: bytes != width*height*16");
    return read_device_common(c, 0, dptr, false);
}

int gs4d_read_pixels_rgba8_device(gs4d_ctx* c, void* dptr, size_t bytes) {
    if (!c || !dptr) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    { int rcq = flush_order(c); if (rcq) return rcq; }
    if (bytes != (size_t)c->W * c->H * 4) return fail(c, GS4D_E_INVALID, "read_pixels_rgba8_device: bytes != width*height*4");
    return read_device_common(c, 0, dptr, true);
}

int gs4d_read_frame_rgba8_device(gs4d_ctx* c, int frames_back, void* dptr, size_t bytes) {
    if (!c || !dptr) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    { int rcq = flush_order(c); if (rcq) return rcq; }
    if (bytes != (size_t)c->W * c->H * 4) return fail(c, GS4D_E_INVALID, "read_frame_rgba8_device: bytes != width*height*4");
    return read_device_common(c, frames_back, dptr, true);
}

int gs4d_read_frame_rgba8_device_after(gs4d_ctx* c, int frames_back, void* dptr, size_t bytes, void* hip_event) {
    if (!c || !dptr) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    { int rcq = flush_order(c); if (rcq) return rcq; }
    if (bytes != (size_t)c->W * c->H * 4) return fail(c, GS4D_E_INVALID, "read_frame_rgba8_device_after: bytes != width*height*4");
    return read_device_common(c, frames_back, dptr, true, true, (hipEvent_t)hip_event);
}

static int band_pixel_rows(const gs4d_ctx* c) {
    int rows = 0;
    for (int ty = c->shard_rank; ty < c->tiles_y; ty += c->shard_world) rows += std::min(TILE, c->H - ty * TILE);
    return rows;
}

int gs4d_set_tile_shard(gs4d_ctx* c, int rank, int world) {
    if (!c) return GS4D_E_INVALID;
    if (world < 1 || world > 1024 || rank < 0 || rank >= world) return fail(c, GS4D_E_INVALID, "set_tile_shard: need 0 <= rank < world <= 1024");
    c->shard_rank = rank; c->shard_world = world;
    return GS4D_OK;
}

int gs4d_band_rows(gs4d_ctx* c, int* rows) {
    if (!c || !rows) return GS4D_E_INVALID;
    *rows = band_pixel_rows(c);
    return GS4D_OK;
}

int gs4d_read_band_rgba8_device(gs4d_ctx* c, void* dptr, size_t bytes) {
    if (!c || !dptr) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    { int rcq = flush_order(c); if (rcq) return rcq; }
    const int rows = band_pixel_rows(c);
    if (bytes != (size_t)rows * c->W * 4) return fail(c, GS4D_E_INVALID, "read_band_rgba8_device: bytes != band_rows*width*4");
    int rc = resolve_image(c, c->cur_fb); if (rc) return rc;
    rc = after_user_stream(c); if (rc) return rc;
    Framebuffer& F = c->fbs[c->cur_fb];
    rc = fb_access(c, F); if (rc) return rc;
    Lane& L = lane(c);
    HIPCHK(c, launch_pack_rgba8_band(L.s, F.mem, F.tstate, F.epoch, F.clear, c->W, c->H, c->tiles_x, c->shard_rank, c->shard_world, rows, (uint32_t*)dptr));
    if (c->user) {
        HIPCHK(c, hipEventRecord(c->ev_readback, L.s));
        HIPCHK(c, hipStreamWaitEvent(c->user, c->ev_readback, 0));
    }
    return GS4D_OK;
}

int gs4d_set_stream(gs4d_ctx* c, void* hip_stream) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = resolve_pending(c); if (rc) return rc;
    rc = sync_all(c); if (rc) return rc;             // everything queued so far completes before the ordering contract changes
    c->user = (hipStream_t)hip_stream;
    return GS4D_OK;
}

// ---- measurement / test hooks ----
int gs4d_set_profiling(gs4d_ctx* c, int stage_mask) {
    if (!c) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    { int rcq = flush_order(c); if (rcq) return rcq; }
    if (stage_mask && c->ev0.empty()) {
        const size_t n = (size_t)gs4d_ctx::PROF_FRAMES * GS4D_T_COUNT;
        c->ev0.assign(n, nullptr); c->ev1.assign(n, nullptr); c->ran.assign(n, 0);
        for (size_t i = 0; i < n; ++i) { HIPCHK(c, hipEventCreate(&c->ev0[i])); HIPCHK(c, hipEventCreate(&c->ev1[i])); }
    }
    c->profiling = (unsigned)stage_mask & 0x3Fu;
    c->prof_every = ((stage_mask >> 8) & 0xFF) ? ((stage_mask >> 8) & 0xFF) : 1;
    c->prof_tick = 0;
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

int gs4d_get_stats(gs4d_ctx* c, uint64_t stats[8]) {
    if (!c || !stats) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = resolve_pending(c); if (rc) return rc;
    stats[0] = (c->stat_entries & 0xFFFFFFFFull) | (c->stat_staged << 32); stats[1] = ((uint64_t)lane(c).pair_cap & 0xFFFFFFFFFFull) | (c->stat_staged_misses << 40); stats[2] = (c->stat_reruns & 0xFFFFFFFFull) | (c->stat_aborted_discarded << 32); stats[3] = (uint64_t)c->tiles_x * c->tiles_y | ((c->stat_shadow_bytes & 0xFFull) << 32) | ((c->stat_composited_tiles & 0xFFFFFFull) << 40);
    stats[4] = (c->stat_depth_passes & 0xFFFFFFFFull) | (c->stat_streams_rejected << 32); stats[5] = (c->stat_tile_passes & 0xFFFFFFFFull) | (c->stat_renamed << 32); stats[6] = (uint64_t)(c->nlanes & 0xFFFF) | (c->stat_lanes_sharing << 16) | (c->stat_fused << 32); stats[7] = c->stat_v2_draws | (c->stat_longest << 32);
    return GS4D_OK;
}

#ifdef GS4D_TUNING
// tuning builds only (make TUNING=1; looked up by name by host/gs4d_sweep --fake-comm-us): occupy `stream` for `usec` microseconds with one spinning
// thread — a stand-in for a communication kernel that holds the stream's hardware queue while it moves data over a slow link
__attribute__((visibility("default"))) int gs4d_tuning_spin(void* stream, unsigned usec) {
    k_lane_probe_spin<<<dim3(1), dim3(1), 0, (hipStream_t)stream>>>((unsigned long long)usec * 100ull, g_tuning_spin_out());
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
#endif

int gs4d_debug_read_projected(gs4d_ctx* c, float* out16, size_t nrecords) {
    if (!c || !out16) return GS4D_E_INVALID;
    (void)hipSetDevice(c->device);
    { int rcq = flush_order(c); if (rcq) return rcq; }
    Lane& L = lane(c);
    if (nrecords > L.proj_n) return fail(c, GS4D_E_INVALID, "debug_read_projected: more records than the last draw projected");
    int rc = resolve_pending(c); if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(out16, L.proj, nrecords * 64, hipMemcpyDeviceToHost, L.s));
    HIPCHK(c, hipStreamSynchronize(L.s));
    // expose the layout documented in gs4d.h: cx,cy,a0x,a0y,a1x,a1y,alpha,r,g,b,rect0,rect1,hx,hy,valid,0
    // stored (gs4d_internal.h):               cx,cy,a0x,a1x,a0y,a1y,r,g,b,alpha,rect0,rect1,hx,hy,valid,0
    for (size_t i = 0; i < nrecords; ++i) {
        float* r = out16 + 16 * i;
        const float a1x = r[3], a0y = r[4], cr = r[6], cg = r[7], cb = r[8], al = r[9];
        r[3] = a0y; r[4] = a1x; r[6] = al; r[7] = cr; r[8] = cg; r[9] = cb;
    }
    return GS4D_OK;
}

} // extern "C"
