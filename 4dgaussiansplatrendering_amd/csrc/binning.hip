// binning.hip — per-tile splat lists that preserve ONE global blend order.
//
// The reference has no tiles: the hardware rasteriser walks instances in order and the ROP blends in that order
// (Renderer.cpp:33-39 -> glDrawElementsInstanced; Application.cpp:150-154).  Here every instance k (record
// i = sortidx[k], Splat4DVertexShaderInstanced.GLSL:9) emits one (tile, i) entry per 8x8 tile its pixel rectangle
// touches, in instance order; a stable radix sort on the tile id (sort.hip) then yields, per tile, the entries in
// instance order — the order the ROP would have blended them in.
//
// Counts stay on the device: the total number of entries is written to total[0] (and an overflow flag to total[1] when it
// exceeds the preallocated capacity); later kernels read it there, so a frame needs no host synchronisation.
#include "gs4d_internal.h"

namespace gs4d {

constexpr int BIN_THREADS = 256;
constexpr int BIN_ITEMS = 4;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(v, off, 64);
        if ((threadIdx.x & 63) >= (unsigned)off) v += t;
    }
    return v;
}

// exclusive scan over the 256 threads of a block; returns the exclusive prefix, *total = block sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wsum /* __shared__[4] */, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    __syncthreads();                      // protect wsum from the previous round
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) if ((unsigned)k < w) base += wsum[k];
    total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return base + inc - v;
}

struct Rect { uint32_t tx0, ty0, tx1, ty1, count; };

__device__ __forceinline__ Rect load_rect(const float4* __restrict__ proj, const uint32_t* __restrict__ order, uint32_t k, uint32_t ninst, uint32_t nrecords, uint32_t& rec) {
    Rect r = { 0, 0, 0, 0, 0 };
    rec = 0;
    if (k >= ninst) return r;
    rec = order ? order[k] : k;
    if (rec >= nrecords) return r;           // index past the bound SSBO: GL would read undefined data; we draw nothing
    float4 c = proj[(size_t)rec * 4 + 2];
    uint32_t r0 = __float_as_uint(c.z), r1 = __float_as_uint(c.w);
    uint32_t x0 = r0 & 0xFFFFu, y0 = r0 >> 16, x1 = r1 & 0xFFFFu, y1 = r1 >> 16;
    if (x0 > x1 || y0 > y1) return r;
    r.tx0 = x0 / TILE; r.ty0 = y0 / TILE; r.tx1 = x1 / TILE; r.ty1 = y1 / TILE;
    r.count = (r.tx1 - r.tx0 + 1u) * (r.ty1 - r.ty0 + 1u);
    return r;
}

__global__ __launch_bounds__(BIN_THREADS) void k_bin_count(const float4* __restrict__ proj, const uint32_t* __restrict__ order, uint32_t ninst, uint32_t nrecords,
                                                           uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t wsum[4];
    uint32_t s = 0, rec;
    const uint32_t base = blockIdx.x * (BIN_THREADS * BIN_ITEMS);
#pragma unroll
    for (int j = 0; j < BIN_ITEMS; ++j) s += load_rect(proj, order, base + j * BIN_THREADS + threadIdx.x, ninst, nrecords, rec).count;
    uint32_t total;
    (void)block_excl_scan(s, wsum, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single workgroup: exclusive scan of block_sums in place; total[0] = sum (saturated), total[1] = (sum > cap), total[2..3] = 64-bit sum
__global__ __launch_bounds__(BIN_THREADS) void k_bin_scan(uint32_t* __restrict__ block_sums, uint32_t nblocks, uint32_t cap, uint32_t* __restrict__ total) {
    __shared__ uint32_t wsum[4];
    unsigned long long run = 0;             // 64-bit: a close-up scene can exceed 2^32 entries; that must read as overflow, not wrap
    for (uint32_t b0 = 0; b0 < nblocks; b0 += BIN_THREADS) {
        uint32_t b = b0 + threadIdx.x;
        uint32_t v = b < nblocks ? block_sums[b] : 0u, tot;
        uint32_t ex = block_excl_scan(v, wsum, tot);
        if (b < nblocks) block_sums[b] = (uint32_t)run + ex;     // only meaningful while run <= cap < 2^32
        run += tot;
    }
    if (threadIdx.x == 0) {
        total[0] = run > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)run;
        total[1] = run > (unsigned long long)cap ? 1u : 0u;
        total[2] = (uint32_t)run; total[3] = (uint32_t)(run >> 32);
    }
}

__device__ __forceinline__ void emit_tiles(const Rect& r, uint32_t off, uint32_t rec, uint32_t tiles_x, uint32_t first, uint32_t stride,
                                           uint32_t* __restrict__ pk, uint32_t* __restrict__ pv) {
    const uint32_t wx = r.tx1 - r.tx0 + 1u;
    for (uint32_t j = first; j < r.count; j += stride) {
        uint32_t ty = r.ty0 + j / wx, tx = r.tx0 + j % wx;
        pk[off + j] = ty * tiles_x + tx;
        pv[off + j] = rec;
    }
}

__global__ __launch_bounds__(BIN_THREADS) void k_bin_emit(const float4* __restrict__ proj, const uint32_t* __restrict__ order, uint32_t ninst, uint32_t nrecords,
                                                          const uint32_t* __restrict__ block_sums, const uint32_t* __restrict__ total, uint32_t tiles_x,
                                                          uint32_t* __restrict__ pk, uint32_t* __restrict__ pv) {
    __shared__ uint32_t wsum[4];
    if (total[1]) return;                     // capacity overflow: the host re-runs the draw with larger lists
    uint32_t run = block_sums[blockIdx.x];
    const uint32_t base = blockIdx.x * (BIN_THREADS * BIN_ITEMS);
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int j = 0; j < BIN_ITEMS; ++j) {
        uint32_t rec;
        Rect r = load_rect(proj, order, base + j * BIN_THREADS + threadIdx.x, ninst, nrecords, rec);
        uint32_t tot;
        uint32_t off = run + block_excl_scan(r.count, wsum, tot);
        run += tot;
        const bool big = r.count > 32u;
        if (!big) emit_tiles(r, off, rec, tiles_x, 0u, 1u, pk, pv);
        // large footprints: the whole wave writes one splat's entries
        uint64_t m = __ballot(big);
        while (m) {
            int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            Rect rr;
            rr.tx0 = __shfl(r.tx0, src, 64); rr.ty0 = __shfl(r.ty0, src, 64); rr.tx1 = __shfl(r.tx1, src, 64); rr.ty1 = __shfl(r.ty1, src, 64);
            rr.count = __shfl(r.count, src, 64);
            uint32_t o2 = __shfl(off, src, 64), rec2 = __shfl(rec, src, 64);
            emit_tiles(rr, o2, rec2, tiles_x, lane, 64u, pk, pv);
        }
    }
}

__global__ __launch_bounds__(256) void k_tile_ranges(const uint32_t* __restrict__ pk, const uint32_t* __restrict__ total, uint32_t ntiles, uint32_t* __restrict__ ranges) {
    if (total[1]) return;
    const uint32_t m = total[0];
    uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= m) return;
    uint32_t cur = pk[j];
    if (cur >= ntiles) return;
    if (j == 0) ranges[2 * cur] = 0;
    else {
        uint32_t prev = pk[j - 1];
        if (prev != cur) { ranges[2 * cur] = j; if (prev < ntiles) ranges[2 * prev + 1] = j; }
    }
    if (j == m - 1) ranges[2 * cur + 1] = m;
}

hipError_t bin_scratch_reserve(hipStream_t st, BinScratch& b, size_t ninst, size_t ntiles) {
    hipError_t e;
    size_t nb = (ninst + BIN_THREADS * BIN_ITEMS - 1) / (BIN_THREADS * BIN_ITEMS);
    if (nb < 1) nb = 1;
    if (b.block_cap < nb) {
        if (b.block_sums) { (void)hipStreamSynchronize(st); (void)hipFree(b.block_sums); }
        b.block_sums = nullptr; b.block_cap = 0;
        if ((e = hipMalloc(&b.block_sums, nb * 4)) != hipSuccess) return e;
        b.block_cap = nb;
    }
    if (!b.total) {
        if ((e = hipMalloc(&b.total, 16)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(b.total, 0, 16, st)) != hipSuccess) return e;
    }
    if (b.tiles_cap < ntiles) {
        if (b.ranges) { (void)hipStreamSynchronize(st); (void)hipFree(b.ranges); }
        b.ranges = nullptr; b.tiles_cap = 0;
        if ((e = hipMalloc(&b.ranges, ntiles * 8)) != hipSuccess) return e;
        b.tiles_cap = ntiles;
    }
    return hipSuccess;
}

void bin_scratch_free(BinScratch& b) {
    if (b.block_sums) (void)hipFree(b.block_sums);
    if (b.total) (void)hipFree(b.total);
    if (b.ranges) (void)hipFree(b.ranges);
    b = BinScratch();
}

hipError_t launch_binning(hipStream_t st, BinScratch& b, const float4* proj, const uint32_t* order, size_t ninst, size_t nrecords, int tiles_x, int tiles_y,
                          uint32_t* pair_keys, uint32_t* pair_vals, size_t pair_cap) {
    (void)tiles_y;
    const uint32_t nb = (uint32_t)((ninst + BIN_THREADS * BIN_ITEMS - 1) / (BIN_THREADS * BIN_ITEMS));
    k_bin_count<<<dim3(nb), dim3(BIN_THREADS), 0, st>>>(proj, order, (uint32_t)ninst, (uint32_t)nrecords, b.block_sums);
    k_bin_scan<<<dim3(1), dim3(BIN_THREADS), 0, st>>>(b.block_sums, nb, (uint32_t)pair_cap, b.total);
    k_bin_emit<<<dim3(nb), dim3(BIN_THREADS), 0, st>>>(proj, order, (uint32_t)ninst, (uint32_t)nrecords, b.block_sums, b.total, (uint32_t)tiles_x, pair_keys, pair_vals);
    return hipGetLastError();
}

hipError_t launch_tile_ranges(hipStream_t st, BinScratch& b, const uint32_t* pair_keys, size_t pair_cap, size_t ntiles) {
    hipError_t e = hipMemsetAsync(b.ranges, 0, ntiles * 8, st);
    if (e != hipSuccess) return e;
    k_tile_ranges<<<dim3((unsigned)((pair_cap + 255) / 256)), dim3(256), 0, st>>>(pair_keys, b.total, (uint32_t)ntiles, b.ranges);
    return hipGetLastError();
}

} // namespace gs4d
