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
    uint32_t* hist = nullptr; size_t hist_cap = 0;                          // two [4][256] digit histograms (alternating), then the look-back words
    int flip = 0;
    uint32_t* totals = nullptr;                                             // [256] spare words (err word when `err` is not set)
    uint32_t* err = nullptr;                                                // not owned: device word raised when a look-back spin times out
};
hipError_t sort_scratch_reserve(hipStream_t st, SortScratch& s, size_t n);
void sort_scratch_free(SortScratch& s);
// Stable LSD radix sort of (key,val) pairs on bits [0, key_bits).  n_dev == nullptr: n is exact.  Otherwise the element
// count is read on the device from *n_dev (<= n, n is the launch capacity); the result always lands back in keys/vals.
hipError_t radix_sort_pairs(hipStream_t st, SortScratch& s, uint32_t* keys, uint32_t* vals, size_t n, const uint32_t* n_dev, int key_bits);
hipError_t launch_keygen(hipStream_t st, const float4* pos, const float4* sig3, size_t n, float t, const float cam[3], const float view[16], int key_mode, float* keys, uint32_t* idx);

// ---- preprocess.hip ----
hipError_t launch_soa_repack(hipStream_t st, const float* aos96, size_t n, float4* soa /* 6 planes of n float4 */);
// Each preprocess launch also writes the compact pixel rectangle of every record and zeroes `zero_words` words at `zero` (the
// binning control block of the same draw), so the draw needs no memset launch.
struct PreOut { float4* proj; uint2* rects; uint32_t* zero; uint32_t zero_words; };
hipError_t launch_preprocess_4d(hipStream_t st, const float4* soa, size_t n, const Uniforms& u, int W, int H, PreOut out);
hipError_t launch_preprocess_3d(hipStream_t st, const float* verts72, size_t n, const Uniforms& u, int W, int H, PreOut out);
hipError_t launch_preprocess_2d(hipStream_t st, const float* rec48, size_t n, const Uniforms& u, int W, int H, PreOut out);

// ---- binning.hip ----
struct BinScratch {
    uint32_t* total = nullptr;        // [0] = number of tile-list entries (saturated), [1] = overflow flag, [2..3] = 64-bit count
    uint32_t* ranges = nullptr; size_t tiles_cap = 0;           // [2*tiles] start,end — followed in the same allocation by
    unsigned long long* status = nullptr; size_t block_cap = 0; // the chained-scan words of the binning workgroups
    // ranges+status are zeroed together at the start of every draw (by the preprocess kernel, or a memset when a draw is re-run)
    size_t zero_words() const { return 2 * tiles_cap + 2 * block_cap; }
};
hipError_t bin_scratch_reserve(hipStream_t st, BinScratch& b, size_t ninst, size_t ntiles);
void bin_scratch_free(BinScratch& b);
// order == nullptr: instance k draws record k
hipError_t launch_binning(hipStream_t st, BinScratch& b, const uint2* rects, const uint32_t* order, size_t ninst, size_t nrecords, int tiles_x, int tiles_y,
                          uint32_t* pair_keys, uint32_t* pair_vals, size_t pair_cap, uint32_t* err);
hipError_t launch_tile_ranges(hipStream_t st, BinScratch& b, const uint32_t* pair_keys, size_t pair_cap, size_t ntiles);

// ---- composite.hip ----
hipError_t launch_composite(hipStream_t st, const float4* proj, const uint32_t* pair_vals, const uint32_t* ranges, const uint32_t* total, int tiles_x, int tiles_y,
                            int W, int H, int premult_c, int fb_is_clear, const float clear[4], float4* fb);
hipError_t launch_fill(hipStream_t st, float4* fb, size_t npix, const float clear[4]);
hipError_t launch_pack_rgba8(hipStream_t st, const float4* fb, size_t npix, uint32_t* out);

} // namespace gs4d
