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
//  A = cx, cy, a0x, a0y      B = a1x, a1y, alpha, r      C = g, b, rect0 (x0 | y0<<16), rect1 (x1 | y1<<16)   [pixel rect, inclusive]
//  D = hx, hy, valid(1/0), 0
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
    uint32_t hist_bias = 0;            // ... of (key - hist_bias): a lower bound of all keys, which makes the high digits constant (and their passes skipped)
    uint32_t epoch = 0;                // tag of the look-back words of the latest pass launch
    uint32_t ticket_base = 0;          // value of the ticket counter (totals[64]) at the start of the next pass launch
    bool atomic_rank = false;          // LDS-atomic ranking verified on this device (lds_atomic_order_selftest)
    uint32_t* totals = nullptr;                                             // [256] spare words (err word when `err` is not set)
    uint32_t* err = nullptr;                                                // not owned: device word raised when a look-back spin times out
};
hipError_t sort_scratch_reserve(hipStream_t st, SortScratch& s, size_t n);
void sort_scratch_free(SortScratch& s);
// Stable LSD radix sort of (key,val) pairs on bits [0, key_bits).  n_dev == nullptr: n is exact.  Otherwise the element
// count is read on the device from *n_dev (<= n, n is the launch capacity); the result always lands back in keys/vals.
// have_hist: the digit histograms of `keys` were already accumulated (by the kernel that wrote the keys) into sort_hist_slot(s).
hipError_t radix_sort_pairs(hipStream_t st, SortScratch& s, uint32_t* keys, uint32_t* vals, size_t n, const uint32_t* n_dev, int key_bits, bool have_hist);
uint32_t* sort_hist_slot(hipStream_t st, SortScratch& s, size_t n_hint, hipError_t* e_out);
hipError_t lds_atomic_order_selftest(hipStream_t st, bool* ordered);
hipError_t launch_keygen(hipStream_t st, const float4* pos, const float4* sig3, size_t n, float t, const float cam[3], const float view[16], int key_mode, float* keys, uint32_t* idx, uint32_t* ghist,
                         uint32_t bias, uint32_t span, uint32_t* err);

// ---- digit histograms for the radix sort, accumulated by whichever kernel produces the keys ----
constexpr int OS_MAX_PASSES = 4;
constexpr int OS_REPL = 8;                                   // replicas of the global histogram: bounds same-address atomic traffic
constexpr size_t OS_SLOT_WORDS = (size_t)OS_REPL * OS_MAX_PASSES * 256;
#ifdef __HIPCC__
__device__ __forceinline__ void os_hist_clear(uint32_t (*h)[256], uint32_t tid) {
#pragma unroll
    for (int p = 0; p < OS_MAX_PASSES; ++p) h[p][tid] = 0;
}
// Wave-cooperative add of one key per active lane (`in` marks the lanes that carry a key; call with the whole wave converged).
// Skewed digits (e.g. the sign/exponent bytes of depth keys take 2-3 values) would serialise 64 LDS atomics on 2-3 addresses:
// the lanes sharing the digit of the first unresolved lane are counted with a ballot and added by one lane, twice; the lanes
// left after that (most lanes of a uniformly distributed digit, almost none of a skewed one) use plain LDS atomics.
__device__ __forceinline__ void os_hist_add(uint32_t (*h)[256], uint32_t key, bool in, int passes) {
    const uint64_t act = __ballot(in);
    if (act == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int p = 0; p < OS_MAX_PASSES; ++p) {
        if (p >= passes) break;
        const uint32_t d = (key >> (8 * p)) & 255u;
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
__device__ __forceinline__ void os_hist_flush(uint32_t (*h)[256], uint32_t* __restrict__ ghist, int passes, uint32_t tid) {
    uint32_t* g = ghist + (size_t)(blockIdx.x % OS_REPL) * OS_MAX_PASSES * 256;
    for (int p = 0; p < passes; ++p) { const uint32_t v = h[p][tid]; if (v) atomicAdd(&g[p * 256 + tid], v); }
}
#endif

// ---- preprocess.hip ----
hipError_t launch_soa_repack(hipStream_t st, const float* aos96, size_t n, float4* soa /* 6 planes of n float4 */, uint32_t* bbox /* [16], preset: min = ~0, max = 0 */);
// Each preprocess launch also writes the compact pixel rectangle of every record.
struct PreOut { float4* proj; uint2* rects; };
hipError_t launch_preprocess_4d(hipStream_t st, const float4* soa, size_t n, const Uniforms& u, int W, int H, PreOut out);
hipError_t launch_preprocess_3d(hipStream_t st, const float* verts72, size_t n, const Uniforms& u, int W, int H, PreOut out);
hipError_t launch_preprocess_2d(hipStream_t st, const float* rec48, size_t n, const Uniforms& u, int W, int H, PreOut out);

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
hipError_t launch_binning(hipStream_t st, BinScratch& b, const uint2* rects, const uint32_t* order, uint32_t* order_copy, size_t ninst, size_t nrecords, int tiles_x, int tiles_y,
                          uint32_t* pair_keys, uint32_t* pair_vals, size_t pair_cap, uint32_t* err, uint32_t* ghist, int passes, uint32_t* total_host, int shard_rank, int shard_world);
hipError_t launch_tile_ranges(hipStream_t st, BinScratch& b, const uint32_t* pair_keys, size_t pair_cap, size_t ntiles);

// ---- composite.hip ----
hipError_t launch_composite(hipStream_t st, const float4* proj, const uint32_t* pair_vals, uint32_t* ranges, const uint32_t* total, int tiles_x, int tiles_y,
                            int W, int H, int premult_c, int fb_is_clear, const float clear[4], float4* fb);
hipError_t launch_fill(hipStream_t st, float4* fb, size_t npix, const float clear[4]);
hipError_t launch_pack_rgba8(hipStream_t st, const float4* fb, size_t npix, uint32_t* out);
// the pixel rows of the tile rows ty % world == rank, top of the band = the context's first tile row; band_rows pixel rows in all
hipError_t launch_pack_rgba8_band(hipStream_t st, const float4* fb, int W, int H, int rank, int world, int band_rows, uint32_t* out);

} // namespace gs4d
