// gs4d_internal.h — shared declarations of libgs4d.so's translation units (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include "../../include/gs4d.h"

namespace gs4d {

constexpr int TILE = 8;             // 8x8-pixel tiles: one wave64 composites one tile
constexpr int PROJ_FLOATS = 16;     // projected record: 64 B, one aligned segment per splat

// Projected record layout (float4 A,B,C,D), written by preprocess, gathered by binning/composite.
//  A = cx, cy, a0x, a1x      B = a0y, a1y, r, g      C = b, alpha, rect0 (x0 | y0<<16), rect1 (x1 | y1<<16)   [pixel rect, inclusive]
//  D = hx, hy, valid(1/0), 0      (gs4d_debug_read_projected hands them out in the order documented in gs4d.h)
// rect0 > rect1 in x (x0 = 1, x1 = 0) marks "no coverage".

struct Uniforms {
    float view[16];
    float proj[16];
    float time;
    float min_opacity;
};

// ---- sort.hip ----
struct SortScratch {
    uint32_t* keys2 = nullptr; uint32_t* vals2 = nullptr; size_t cap = 0;   // scratch B and C (keys2[2*cap], vals2[2*cap]; one allocation) — radix_sort.hpp:192-216 scratch
    uint32_t* hist = nullptr; size_t hist_cap = 0;                          // two [OS_REPL][4][256] digit-histogram slots (alternating), then the look-back words
    int flip = 0;
    bool hist_pending = false;         // the current slot holds a histogram accumulated by a producer kernel, not yet consumed by a sort
    int acc_flip = 0;                  // which accumulator set the next launch uses (the other one it zeroes)
    int hist_bits = 32;                // host-proven width of (key - hist_bias): passes above it are not even launched
    int hist_rb = 8;                   // digit width the producer of the pending histogram counted in (sort_plan_rb, chosen together with hist_bits)
    int rb_knob = 0;                   // test hook GS4D_SORT_RB (8 / 9): the digit width of every sort whose tile shape allows it
    uint32_t hist_bias = 0;            // ... of (key - hist_bias): a lower bound of all keys, which makes the high digits constant (and their passes skipped)
    uint32_t epoch = 0;                // tag of the look-back words of the latest pass launch
    bool atomic_rank = false;          // LDS-atomic ranking verified on this device (lds_atomic_order_selftest)
    int shape_knob = 0, rank_knob = 0; // test / tuning hooks read at context creation: GS4D_SORT_SHAPE (1..6: tile shape of a pass), GS4D_SORT_RANK (1 = ballot ranking, 2 = LDS-atomic ranking)
    uint32_t* totals = nullptr;                                             // [256] spare words (err word when `err` is not set)
    uint32_t* err = nullptr;                                                // not owned: device word raised when a look-back spin times out
    // persistent workgroups a pass may launch (occupancy x compute units of THIS scratch's device), per kernel instance: asked of the runtime once per
    // scratch — a scratch belongs to one context, a context to one device; no process-wide cache (two contexts on two devices, or created on two threads)
    struct Resident { const void* fn = nullptr; uint32_t groups = 0; } resident[16];
};
hipError_t sort_scratch_reserve(hipStream_t st, SortScratch& s, size_t n);
void sort_scratch_free(SortScratch& s);
// Stable LSD radix sort of (key,val) pairs on bits [0, key_bits).  n_dev == nullptr: n is exact.  Otherwise the element
// count is read on the device from *n_dev (<= n, n is the launch capacity); the result always lands back in keys/vals.
// have_hist: the digit histograms of `keys` were already accumulated (by the kernel that wrote the keys) into sort_hist_slot(s).
// identity_vals: the payload is the identity index 0..n-1 and `vals` has NOT been written: the first pass that moves keys makes the indices up
// instead of reading them (4 bytes per key less to write for whoever produced the keys, 4 less to read here).
hipError_t radix_sort_pairs(hipStream_t st, SortScratch& s, uint32_t* keys, uint32_t* vals, size_t n, const uint32_t* n_dev, int key_bits, bool have_hist, bool identity_vals = false);
uint32_t* sort_hist_slot(hipStream_t st, SortScratch& s, size_t n_hint, hipError_t* e_out);
int sort_plan_rb(const SortScratch& s, size_t n, int key_bits);      // digit width (8 or 9 bits) of a sort of key_bits-bit keys
int sort_plan_passes(int key_bits, int rb);                          // ... and the launches it takes
hipError_t lds_atomic_order_selftest(const hipStream_t* streams, int nstreams, bool* ordered);      // on all the streams at once
// Layout of the SoA shadow (preprocess.hip).  The repack kernel verifies what a compact layout assumes, bit for bit, for every record, and
// reports a violation in bbox[15]; the caller then repacks in the next layout down.
//   SOA_STATIC3D  64 B/record: a static 3D splat in the reference's 4D record — mu_t, the time row and the time column of sig are the same
//                 eight values in every record (SoaInfo::consts; Scenes.h ObjectDisplay / a 3D covariance with Sigma44 = 1) — keeps position,
//                 colour and the nine spatial elements of sig: planes (px py pz s00), col, (s01 s02 s10 s11), (s12 s20 s21 s22)
//   SOA_SYM       72 B/record: a symmetric sig without its mirrored half
//   SOA_FULL      96 B/record: pos, col, sig[0..3]
enum { SOA_FULL = 0, SOA_SYM = 1, SOA_STATIC3D = 2 };
struct SoaInfo { int layout = SOA_FULL; float consts[8] = { 0 }; };      // consts: pos.w, sig[0][3], sig[1][3], sig[2][3], sig[3][0..3] of a static set
// sig3 == nullptr: a static set (SOA_STATIC3D) — mu_t and sig[3] are info.consts for every record
hipError_t launch_keygen(hipStream_t st, const float4* pos, const float4* sig3, const SoaInfo& info, size_t n, float t, const float cam[3], const float view[16], int key_mode, float* keys, uint32_t* idx, uint32_t* ghist, int rb,
                         uint32_t bias, uint32_t span, uint32_t* err);

// ---- digit histograms for the radix sort, accumulated by whichever kernel produces the keys ----
constexpr int OS_MAX_PASSES = 4;
constexpr int OS_REPL = 8;                                   // replicas of the global histogram: bounds same-address atomic traffic
constexpr uint32_t OS_MAX_BINS = 512;                        // digits are 8 or 9 bits wide (sort_plan_rb); a histogram row always has room for 512 bins
constexpr size_t OS_SLOT_WORDS = (size_t)OS_REPL * OS_MAX_PASSES * OS_MAX_BINS;
#ifdef __HIPCC__
// (threads 0..255 of the workgroup call clear and flush: each looks after bins tid and tid + 256)
__device__ __forceinline__ void os_hist_clear(uint32_t (*h)[OS_MAX_BINS], uint32_t tid) {
#pragma unroll
    for (int p = 0; p < OS_MAX_PASSES; ++p) { h[p][tid] = 0; h[p][tid + 256u] = 0; }
}
// Wave-cooperative add of one key per active lane (`in` marks the lanes that carry a key; call with the whole wave converged).
// Skewed digits (e.g. the sign/exponent bytes of depth keys take 2-3 values) would serialise 64 LDS atomics on 2-3 addresses:
// the lanes sharing the digit of the first unresolved lane are counted with a ballot and added by one lane, twice; the lanes
// left after that (most lanes of a uniformly distributed digit, almost none of a skewed one) use plain LDS atomics.
__device__ __forceinline__ void os_hist_add(uint32_t (*h)[OS_MAX_BINS], uint32_t key, bool in, int passes, int rb) {
    const uint64_t act = __ballot(in);
    if (act == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int p = 0; p < OS_MAX_PASSES; ++p) {
        if (p >= passes) break;
        const uint32_t d = (key >> (rb * p)) & ((1u << rb) - 1u);
        uint64_t rem = act;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            if (rem == 0ull) break;
            const uint32_t first = (uint32_t)__ffsll((long long)rem) - 1u;
            const uint32_t d0 = __shfl(d, (int)first, 64);
            const uint64_t peers = __ballot(((rem >> lane) & 1ull) && d == d0);
            if (lane == first) atomicAdd(&h[p][d0], (uint32_t)__popcll(peers));
            rem &= ~peers;
        }
        if ((rem >> lane) & 1ull) atomicAdd(&h[p][d], 1u);
    }
}
__device__ __forceinline__ void os_hist_flush(uint32_t (*h)[OS_MAX_BINS], uint32_t* __restrict__ ghist, int passes, uint32_t tid) {
    uint32_t* g = ghist + (size_t)(blockIdx.x % OS_REPL) * OS_MAX_PASSES * OS_MAX_BINS;
    for (int p = 0; p < passes; ++p) {
        const uint32_t v = h[p][tid], v2 = h[p][tid + 256u];
        if (v) atomicAdd(&g[p * OS_MAX_BINS + tid], v);
        if (v2) atomicAdd(&g[p * OS_MAX_BINS + tid + 256u], v2);
    }
}
#endif

// ---- the unordered draw path (tilelist.hip, composite2.hip): per-tile lists built with atomics, ordered inside the compositor ----
// Blend order of the instances = ascending (key, record index).  Where that key comes from:
//   KEYSRC_INDEX   instance k draws record k (4D-direct, 3D-full, 2D): key = k
//   KEYSRC_REF / KEYSRC_VIEWZ   the bound sort index is the library's own stable sort of the keys gs4d_keygen produced for exactly
//                  these records (tracked by buffer versions): "instance order" == ascending (depth key, record index), so the key
//                  is recomputed per record with k_keygen's arithmetic and the sort index is never read.
enum { KEYSRC_INDEX = 0, KEYSRC_REF = 1, KEYSRC_VIEWZ = 2 };
struct KeySrc {
    int mode = KEYSRC_INDEX;
    float t = 0, camx = 0, camy = 0, camz = 0;
    float vr0 = 0, vr1 = 0, vr2 = 0, vr3 = 0;   // view row 2 at keygen time (KEYSRC_VIEWZ)
    uint32_t bias = 0;                           // subtracted from the key's bit pattern (host-proven lower bound, as in the depth sort)
};
struct TileCount {                               // hist == nullptr: the ordered path (no counting in the projection kernel)
    uint32_t* hist = nullptr;                    // [nb][rows]: entries that the records of segment `row` put into bucket b = tile % nb
    uint32_t rows = 0;
    uint32_t* skey = nullptr;                    // [records] blend-order key of every record
    uint32_t nb = 0, seg = 0;                    // buckets (a power of two), records per segment (a multiple of SEG_THREADS; one workgroup walks one segment)
    int tiles_x = 0, shard_rank = 0, shard_world = 1;
    KeySrc ks;
    // Fused key generation (the draw executes a gs4d_keygen + gs4d_sort_pairs that were queued just before it): the projection kernel also
    // writes the caller's key and index buffers and accumulates the digit histograms of the depth sort, exactly as k_keygen would
    float* keys_out = nullptr; uint32_t* idx_out = nullptr; uint32_t* ghist = nullptr; int hist_rb = 8; uint32_t span = 0xFFFFFFFFu; uint32_t* err = nullptr;
    uint32_t* sstat = nullptr;                   // [rows]: entries of every segment (statistics for the host; every counting launch writes them)
    // Staged lists (tilelist.hip): the projection kernel itself WRITES the entries.  A workgroup counts its segment's entries per bucket, scans
    // the counts, places the entries bucket by bucket in LDS and writes them out as ONE dense block: stage_out[segment * scap + offs[b][segment] + k],
    // k < hist[b][segment].  A segment with more than scap entries stores `seq` into *abort_word instead.  null: count only.
    uint2* stage_out = nullptr; uint32_t scap = 0; uint32_t* offs = nullptr; uint32_t* abort_word = nullptr; uint32_t seq = 0;
};
constexpr uint32_t V2_MAX_LIST = 1024;           // longest list the compositor sorts in LDS (beyond ~1000 entries per tile its LDS footprint costs more occupancy than the ordered path's two sort passes cost time).  Longer per-tile lists are cut into depth slabs (below); beyond V2_MAX_SLABS a draw uses the ordered path
constexpr uint32_t V2_MAX_SLABS = 64;           // a tile's list is kept as `slabs` sub-lists by equal ranges of the blend key: far slab first, each ordered by itself in the compositor (one wave lane holds a sub-list's table entry: <= 64)
constexpr int STAGE_R = 4;                       // staged lists: records per thread of a segment (kept in registers between the counting and the placing pass): seg <= STAGE_R * SEG_THREADS
constexpr uint32_t STAGE_MAX_SCAP = 5120;        // ... and entries of a segment block (LDS: 8 bytes each beside the 12 KB the projection kernel has already)
constexpr int SEG_THREADS = 512;                 // workgroup size of the kernels that walk a segment of records (k_preprocess<.., true>, k_bucket_scatter)
// list capacities the compositor is instantiated for (64 entries per lane-register): the smallest one >= n
inline uint32_t v2_list_capacity(uint32_t n) {
    static const uint32_t ladder[] = { 64, 128, 192, 256, 384, 512, 768, 1024 };
    for (uint32_t c : ladder) if (n <= c) return c;
    return V2_MAX_LIST;
}
constexpr uint32_t V2_MAX_RECORDS = 1u << 24;    // an entry carries (tile / nb) in the top byte of its record word
struct TileLists {
    uint32_t* hist = nullptr; size_t hist_cap = 0;            // [nb][rows] counts, turned in place into the slot of every (segment, bucket) run inside its bucket
    uint32_t* bbase = nullptr; uint32_t* btot = nullptr; uint32_t* tstart = nullptr; uint32_t* tcnt = nullptr; size_t tiles_cap = 0, nb_cap = 0, slabs_cap = 0;   // [nb + 1] bucket starts, [nb] bucket totals; per tile: first entry, entries
    uint32_t* skey = nullptr; size_t skey_cap = 0;
    uint32_t counters = 0;                                    // LDS counters of k_bucket_tiles: (tiles per bucket) * slabs
    uint32_t nb = 0, rows = 0, seg = 0, slabs = 1, slab_shift = 0;   // geometry of the current draw (tile_lists_plan): slab of an entry = min(slabs - 1, key >> slab_shift)
    // staged lists: segment blocks [rows][scap] entries; per-bucket statistics {entries, longest run, longest list, 0} written by k_bucket_tiles_staged
    // (staged draws) or k_bucket_scan (exact draws), per-segment entry counts written by the projection kernel; the compositing kernel's first
    // workgroup reduces both for the host
    uint2* blocks = nullptr; size_t blocks_cap = 0;           // in entries
    uint4* bstat = nullptr;                                    // [nb_cap]
    uint32_t* sstat = nullptr;                                 // [1024]
    bool staged = false; uint32_t scap = 0, bcap = 0;         // the current draw is staged; entries a segment block holds, bucket capacity in the tile-ordered entry array
    uint32_t seq = 0;                                          // sequence number of the lane's staged draws: total[TL_ABORT_WORD] == seq <=> this draw was aborted
    // staged draws: the BOX of tiles (in blocks of BOX_BLOCK x BOX_BLOCK tiles, both ends inclusive, box_pack) outside which the host expects no entry — a third guess
    // from the previous frames' statistics (bstat[b].w: the box of bucket b's non-empty tiles), checked by k_bucket_tiles_staged like the two capacities;
    // the compositor is launched for these tiles only.  BOX_NONE: the whole image.
    uint32_t box = 0xFFFFFFFFu;
};
constexpr uint32_t BOX_BLOCK = 4u, BOX_NONE = 0xFFFFFFFFu, BOX_EMPTY = 0x0000FFFFu;      // BOX_EMPTY: min 255, max 0 in both directions — the neutral element of box_join
__host__ __device__ __forceinline__ uint32_t box_pack(uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) { return x0 | (y0 << 8) | (x1 << 16) | (y1 << 24); }
__host__ __device__ __forceinline__ uint32_t box_join(uint32_t a, uint32_t b) {
    const uint32_t x0 = (a & 255u) < (b & 255u) ? (a & 255u) : (b & 255u), y0 = ((a >> 8) & 255u) < ((b >> 8) & 255u) ? ((a >> 8) & 255u) : ((b >> 8) & 255u);
    const uint32_t x1 = ((a >> 16) & 255u) > ((b >> 16) & 255u) ? ((a >> 16) & 255u) : ((b >> 16) & 255u), y1 = (a >> 24) > (b >> 24) ? (a >> 24) : (b >> 24);
    return box_pack(x0, y0, x1, y1);
}
__host__ __device__ __forceinline__ bool box_holds(uint32_t box, uint32_t bx, uint32_t by) {      // block (bx, by) inside the box
    return box == BOX_NONE || (bx >= (box & 255u) && bx <= ((box >> 16) & 255u) && by >= ((box >> 8) & 255u) && by <= (box >> 24));
}
constexpr int TL_ABORT_WORD = 9;                              // index into BinScratch::total
// false: this frame / record count cannot use the unordered path (more than 256 * 1024 tiles, or 2^24 records)
// key_span: host-proven largest blend key (the slabs divide [0, key_span] evenly)
bool tile_lists_plan(TileLists& t, size_t ntiles, size_t nrecords, uint32_t slabs, int keybits, uint32_t key_span = 0xFFFFFFFFu, size_t expect_entries = 0);
hipError_t tile_lists_reserve(hipStream_t st, TileLists& t, size_t ntiles, size_t nrecords);
void tile_lists_free(TileLists& t);
// total[0] entries (saturated), [1] abort flags (1: more entries than `cap`, 2: a list longer than `hint`), [2..3] 64-bit entry count, [4] longest list,
// [6] workgroups of k_bucket_tiles that have finished, [7] of k_bucket_scan (back to 0 when the kernel ends); total_host (pinned, mapped) receives [0..3] and the longest list at [5]
hipError_t launch_bucket_scan(hipStream_t st, TileLists& t, uint32_t* total, uint32_t* total_host, size_t cap);
// skey == nullptr: the blend keys the projection left in t.skey; else an array of key bit patterns from which skey_bias is still to be subtracted (the caller's key buffer of a fused draw)
hipError_t launch_bucket_scatter(hipStream_t st, TileLists& t, const uint32_t* trects, const float4* proj, const uint32_t* skey, uint32_t skey_bias, size_t nrecords, const uint32_t* total, uint2* tmp, int tiles_x, int shard_rank, int shard_world);
hipError_t launch_bucket_tiles(hipStream_t st, TileLists& t, size_t ntiles, uint32_t* total, const uint2* tmp, uint2* entries, uint32_t hint);
// staged draws: the segment blocks the projection kernel wrote (t.blocks, t.scap) -> tile lists at entries[b * t.bcap ...]; per-bucket statistics into t.bstat;
// total[TL_ABORT_WORD] = t.seq when a run, a bucket or a list does not fit
hipError_t launch_bucket_tiles_staged(hipStream_t st, TileLists& t, size_t ntiles, int tiles_x, uint32_t* total, uint2* entries, uint32_t hint);
hipError_t tile_lists_reserve_blocks(hipStream_t st, TileLists& t, size_t entries);
// bstat / nb, sstat / rows: per-bucket and per-segment statistics for the host report; stage_seq != 0: a staged draw (aborted <=> total[TL_ABORT_WORD] == stage_seq,
// the entry total is the sum of the statistics); rcap / scap / bcap: what the host guessed for it (longest run: no limit any more, 0xFFFFFFFF; fullest segment; fullest bucket)
hipError_t launch_composite_v2(hipStream_t st, const float4* proj, const uint2* entries, const uint32_t* tstart, const uint32_t* tcnt, const uint32_t* total, uint32_t* total_host, int tiles_x, int tiles_y, int W, int H,
                               int premult_c, uint32_t* tstate, uint32_t epoch, const float clear[4], float4* fb, uint32_t hint, int keybits, int recbits, uint32_t slabs,
                               const uint4* bstat = nullptr, uint32_t nb = 0, const uint32_t* sstat = nullptr, uint32_t rows = 0, uint32_t stage_seq = 0, uint32_t rcap = 0, uint32_t scap = 0, uint32_t bcap = 0, uint32_t box_blocks = 0xFFFFFFFFu);

#ifdef __HIPCC__
// tiles touched by a pixel rectangle (x0|y0<<16, x1|y1<<16; x0 > x1: none), restricted to the tile rows ty % world == rank
struct TRect { uint32_t tx0, ty0, wx, rows, tstep, count; };
// tiles tx0 .. tx0 + wx - 1, ty0 .. ty1
__device__ __forceinline__ TRect tile_rect_of(uint32_t tx0, uint32_t ty0, uint32_t wx, uint32_t ty1, uint32_t shard_rank, uint32_t shard_world) {
    TRect r{ tx0, ty0, wx, ty1 - ty0 + 1u, 1u, 0u };
    if (shard_world > 1u) {
        const uint32_t first = r.ty0 + (shard_rank + shard_world - r.ty0 % shard_world) % shard_world;
        r.rows = first > ty1 ? 0u : (ty1 - first) / shard_world + 1u;
        r.ty0 = first; r.tstep = shard_world;
    }
    r.count = r.wx * r.rows;
    return r;
}
__device__ __forceinline__ TRect tile_rect(uint32_t rect0, uint32_t rect1, uint32_t shard_rank, uint32_t shard_world) {
    const uint32_t x0 = rect0 & 0xFFFFu, y0 = rect0 >> 16, x1 = rect1 & 0xFFFFu, y1 = rect1 >> 16;
    if (x0 > x1 || y0 > y1) return TRect{ 0u, 0u, 0u, 0u, 1u, 0u };
    return tile_rect_of(x0 / TILE, y0 / TILE, x1 / TILE - x0 / TILE + 1u, y1 / TILE, shard_rank, shard_world);
}
// The per-record input of the list-building kernels: the tile rectangle packed into 4 bytes — tx0:10 | ty0:10 | wx-1:6 | wy-1:6 — read in
// record order by the scatter (streaming) and in depth order by the binning (a gather: 4 bytes instead of the 8-byte pixel rectangle of
// rounds 1-2 halves the table).  TRECT_NONE: the record touches no tile.  TRECT_WIDE: more than 63 tiles across or down, or beyond tile
// 1022 (images wider than 8184 pixels): the consumer takes the pixel rectangle from the projected record (C.z, C.w) instead.
constexpr uint32_t TRECT_NONE = 0xFFFFFFFFu, TRECT_WIDE = 0xFFFFFFFEu;
__device__ __forceinline__ uint32_t pack_trect(uint32_t rect0, uint32_t rect1) {
    const uint32_t x0 = rect0 & 0xFFFFu, y0 = rect0 >> 16, x1 = rect1 & 0xFFFFu, y1 = rect1 >> 16;
    if (x0 > x1 || y0 > y1) return TRECT_NONE;
    const uint32_t tx0 = x0 / TILE, ty0 = y0 / TILE, wx = x1 / TILE - tx0, wy = y1 / TILE - ty0;      // widths minus one
    if (tx0 >= 1023u || ty0 >= 1023u || wx >= 63u || wy >= 63u) return TRECT_WIDE;
    return tx0 | (ty0 << 10) | (wx << 20) | (wy << 26);
}
__device__ __forceinline__ TRect unpack_trect(uint32_t w, const float4* __restrict__ proj, uint32_t rec, uint32_t shard_rank, uint32_t shard_world) {
    if (w == TRECT_NONE) return TRect{ 0u, 0u, 0u, 0u, 1u, 0u };
    if (w == TRECT_WIDE) { const float4 c = proj[(size_t)rec * 4 + 2]; return tile_rect(__float_as_uint(c.z), __float_as_uint(c.w), shard_rank, shard_world); }
    const uint32_t tx0 = w & 1023u, ty0 = (w >> 10) & 1023u;
    return tile_rect_of(tx0, ty0, ((w >> 20) & 63u) + 1u, ty0 + (w >> 26), shard_rank, shard_world);
}
__device__ __forceinline__ uint32_t tile_of(const TRect& r, uint32_t j, uint32_t tiles_x) { return (r.ty0 + (j / r.wx) * r.tstep) * tiles_x + r.tx0 + j % r.wx; }
// every tile of a small footprint, row by row (no division: this runs once per record in two kernels)
template <class F> __device__ __forceinline__ void for_each_tile(const TRect& r, uint32_t tiles_x, F f) {
    uint32_t rowbase = r.ty0 * tiles_x + r.tx0;
    for (uint32_t y = 0; y < r.rows; ++y) { for (uint32_t x = 0; x < r.wx; ++x) f(rowbase + x); rowbase += r.tstep * tiles_x; }
}
#endif

// ---- preprocess.hip ----
hipError_t launch_soa_repack(hipStream_t st, const float* aos96, size_t n, float4* soa /* n * 96 bytes */, uint32_t* bbox /* [16], preset: min = ~0, max = 0 */, const SoaInfo& info);
// plane 0 holds pos.xyz in every layout; the plane of sig[3] (what key generation reads beside it), or null when it is one of the constants
inline const float4* soa_sig3(const float4* soa, size_t n, const SoaInfo& info) { return info.layout == SOA_STATIC3D ? nullptr : soa + (info.layout == SOA_SYM ? 3 : 5) * n; }
// Each preprocess launch also writes the packed tile rectangle of every record (pack_trect).
struct PreOut { float4* proj; uint32_t* trects; };
hipError_t launch_preprocess_4d(hipStream_t st, const float4* soa, size_t soa_n /* records in the buffer: the plane stride */, const SoaInfo& info, size_t n, const Uniforms& u, int W, int H, PreOut out, const TileCount& tc);
hipError_t launch_preprocess_3d(hipStream_t st, const float* verts72, size_t n, const Uniforms& u, int W, int H, PreOut out, const TileCount& tc);
hipError_t launch_preprocess_2d(hipStream_t st, const float* rec48, size_t n, const Uniforms& u, int W, int H, PreOut out, const TileCount& tc);

// ---- binning.hip ----
struct BinScratch {
    uint32_t* total = nullptr;        // [0] = number of tile-list entries (saturated), [1] = overflow flag, [2..3] = 64-bit count
    uint32_t* ranges = nullptr; size_t tiles_cap = 0;           // [2*tiles] start,end — followed in the same allocation by
    unsigned long long* status = nullptr; size_t block_cap = 0; // the chained-scan words of the binning workgroups (epoch-tagged, never zeroed)
    uint32_t epoch = 0;
    uint32_t ticket_base = 0;         // total[8] is the ticket counter of the binning workgroups
    // ranges are all-zero between draws: k_tile_ranges fills the non-empty tiles, the composite kernel clears each range it has read
};
hipError_t bin_scratch_reserve(hipStream_t st, BinScratch& b, size_t ninst, size_t ntiles);
void bin_scratch_free(BinScratch& b);
// order == nullptr: instance k draws record k
// trects_in_order: trects[k] belongs to INSTANCE k (it travelled through the depth sort as a second payload); else to record k (gathered through `order`)
hipError_t launch_binning(hipStream_t st, BinScratch& b, const uint32_t* trects, bool trects_in_order, const float4* proj, const uint32_t* order, uint32_t* order_copy, size_t ninst, size_t nrecords, int tiles_x, int tiles_y,
                          uint32_t* pair_keys, uint32_t* pair_vals, size_t pair_cap, uint32_t* err, uint32_t* ghist, int passes, uint32_t* total_host, int shard_rank, int shard_world);
hipError_t launch_tile_ranges(hipStream_t st, BinScratch& b, const uint32_t* pair_keys, size_t pair_cap, size_t ntiles);

// ---- composite.hip ----
// tstate / epoch: the image's tile state (composite.hip): tstate[tile] == epoch <=> the tile's pixels are in memory, else it is still the clear colour
hipError_t launch_composite(hipStream_t st, const float4* proj, const uint32_t* pair_vals, uint32_t* ranges, const uint32_t* total, int tiles_x, int tiles_y,
                            int W, int H, int premult_c, uint32_t* tstate, uint32_t epoch, const float clear[4], float4* fb, int blend_src, int blend_dst);
hipError_t launch_fill_unwritten(hipStream_t st, float4* fb, uint32_t* tstate, uint32_t epoch, int tiles_x, int tiles_y, int W, int H, const float clear[4]);
hipError_t launch_pack_rgba8(hipStream_t st, const float4* fb, const uint32_t* tstate, uint32_t epoch, const float clear[4], int W, int H, int tiles_x, uint32_t* out);
// the pixel rows of the tile rows ty % world == rank, top of the band = the context's first tile row; band_rows pixel rows in all
hipError_t launch_pack_rgba8_band(hipStream_t st, const float4* fb, const uint32_t* tstate, uint32_t epoch, const float clear[4], int W, int H, int tiles_x, int rank, int world, int band_rows, uint32_t* out);

// ---- lines.hip ----
struct LineParams { float vp[16]; float rgba[4]; int W, H; int blend_src, blend_dst; };
// verts_dev: nverts positions of `dims` floats on the device; cnt: W*H fragment counters, all-zero between calls
hipError_t launch_lines(hipStream_t st, const float* verts_dev, size_t nverts, int dims, int strip, const LineParams& p, float width, uint32_t* cnt, float4* fb);

} // namespace gs4d
