// tilelist.hip — per-tile splat lists for the unordered draw path: count -> scan -> scatter, no workgroup ever waits for another.
//
// The reference has no tiles: the hardware rasteriser walks the instances in order and the ROP blends in that order
// (Renderer.cpp:33-39 -> glDrawElementsInstanced; Application.cpp:150-154).  The ordered path (binning.hip) reproduces that order by
// emitting (tile, record) entries in instance order — a chained scan — and stable-sorting them by tile: four launches, three of
// them chained scans whose cost at 10^6 splats is hand-off latency, not bandwidth.  Here the order is restored where it is consumed:
//   k_preprocess_* (preprocess.hip) counts the entries of every tile with no-return atomics while it projects the records,
//   k_tilescan     turns the counts into list starts (one workgroup; 32 400 tiles at 1080p) and validates capacity and list length,
//   k_tile_scatter puts (blend-order key, record) on the lists in whatever order its returning atomics resolve,
//   k_composite_v2 (composite2.hip) sorts each tile's list by (key, record) in LDS before it blends.
// Which key gives "instance order" is decided on the host (KeySrc, gs4d_internal.h): the record index when instance k draws record k,
// the depth key when the bound sort index is the library's own sort of gs4d_keygen's keys (the sort index is then never read).
// Lists longer than the compositor can hold, or more entries than the preallocated capacity, raise a flag in k_tilescan: the
// scatter and the compositor then do nothing and the host re-runs the draw (larger capacity, longer lists, or the ordered path).
#include "gs4d_internal.h"
#include <algorithm>

namespace gs4d {

typedef unsigned long long u64;

constexpr int SCAN_THREADS = 1024;

// tstart[t] = number of entries on the lists of tiles < t; cursor[t] = the same (the scatter's running position); tcount cleared.
__global__ __launch_bounds__(SCAN_THREADS) void k_tilescan(uint32_t* __restrict__ tcount, uint32_t ntiles, uint32_t* __restrict__ tstart, uint32_t* __restrict__ cursor,
                                                           uint32_t* __restrict__ total, uint32_t* __restrict__ total_host, uint32_t cap, uint32_t hint) {
    __shared__ u64 wsum[SCAN_THREADS / 64];
    __shared__ uint32_t wmax[SCAN_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint32_t per = (ntiles + SCAN_THREADS - 1u) / SCAN_THREADS;
    const uint32_t t0 = min(tid * per, ntiles), t1 = min(t0 + per, ntiles);
    u64 sum = 0; uint32_t mx = 0;
    for (uint32_t t = t0; t < t1; ++t) { const uint32_t c = tcount[t]; sum += c; mx = max(mx, c); }
    u64 inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const u64 v = __shfl_up(inc, off, 64); if (lane >= (unsigned)off) inc += v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, off, 64));
    if (lane == 63u) wsum[w] = inc;
    if (lane == 0u) wmax[w] = mx;
    __syncthreads();
    u64 base = 0, grand = 0; uint32_t gmax = 0;
#pragma unroll
    for (int k = 0; k < SCAN_THREADS / 64; ++k) { const u64 s = wsum[k]; if ((unsigned)k < w) base += s; grand += s; gmax = max(gmax, wmax[k]); }
    u64 run = base + inc - sum;
    const bool over = grand > (u64)cap;
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t c = tcount[t];
        const uint32_t r32 = run > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)run;     // only meaningful when the draw is not aborted
        tstart[t] = r32; cursor[t] = r32; tcount[t] = 0u;
        run += c;
    }
    if (tid == 0) {
        const uint32_t sat = grand > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)grand;
        const uint32_t flags = (over ? 1u : 0u) | (gmax > hint ? 2u : 0u);
        tstart[ntiles] = sat;
        total[0] = sat; total[1] = flags; total[2] = (uint32_t)grand; total[3] = (uint32_t)(grand >> 32); total[4] = gmax;
        total_host[0] = sat; total_host[1] = flags; total_host[2] = (uint32_t)grand; total_host[3] = (uint32_t)(grand >> 32); total_host[5] = gmax;
    }
}

// One thread per record: (key, record) onto the list of every tile its pixel rectangle touches.  The position inside a list is
// whatever the returning atomic hands out — the compositor orders the list.
__global__ __launch_bounds__(256) void k_tile_scatter(const uint2* __restrict__ rects, const uint32_t* __restrict__ skey, uint32_t n, uint32_t* __restrict__ cursor,
                                                      const uint32_t* __restrict__ total, uint2* __restrict__ entries, uint32_t tiles_x, uint32_t shard_rank, uint32_t shard_world) {
    if (total[1]) return;                                  // aborted draw: positions may lie beyond the capacity
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint2 rc = rects[i];
    const TRect r = tile_rect(rc.x, rc.y, shard_rank, shard_world);
    const uint32_t key = r.count ? skey[i] : 0u;
    const bool big = r.count > 16u;
    if (!big) for (uint32_t j = 0; j < r.count; ++j) { const uint32_t pos = atomicAdd(&cursor[tile_of(r, j, tiles_x)], 1u); entries[pos] = make_uint2(key, i); }
    uint64_t m = __ballot(big);
    const uint32_t lane = threadIdx.x & 63u;
    while (m) {                                            // large footprints: the whole wave writes one record's entries
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        TRect rr;
        rr.tx0 = __shfl(r.tx0, src, 64); rr.ty0 = __shfl(r.ty0, src, 64); rr.wx = __shfl(r.wx, src, 64);
        rr.rows = __shfl(r.rows, src, 64); rr.tstep = __shfl(r.tstep, src, 64); rr.count = __shfl(r.count, src, 64);
        const uint32_t key2 = __shfl(key, src, 64), rec2 = __shfl(i, src, 64);
        for (uint32_t j = lane; j < rr.count; j += 64u) { const uint32_t pos = atomicAdd(&cursor[tile_of(rr, j, tiles_x)], 1u); entries[pos] = make_uint2(key2, rec2); }
    }
}

hipError_t tile_lists_reserve(hipStream_t st, TileLists& t, size_t ntiles, size_t nrecords) {
    hipError_t e;
    if (t.tiles_cap < ntiles) {
        if (t.tcount) { (void)hipStreamSynchronize(st); (void)hipFree(t.tcount); }
        t.tcount = t.tstart = t.cursor = nullptr; t.tiles_cap = 0;
        const size_t words = 3 * ntiles + 4;
        if ((e = hipMalloc(&t.tcount, words * 4)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(t.tcount, 0, words * 4, st)) != hipSuccess) return e;      // the counts stay zero between draws: k_tilescan clears what it reads
        t.tstart = t.tcount + ntiles; t.cursor = t.tstart + ntiles + 1;
        t.tiles_cap = ntiles;
    }
    if (t.skey_cap < nrecords) {
        if (t.skey) { (void)hipStreamSynchronize(st); (void)hipFree(t.skey); }
        t.skey = nullptr; t.skey_cap = 0;
        if ((e = hipMalloc(&t.skey, nrecords * 4)) != hipSuccess) return e;
        t.skey_cap = nrecords;
    }
    return hipSuccess;
}

void tile_lists_free(TileLists& t) {
    if (t.tcount) (void)hipFree(t.tcount);
    if (t.skey) (void)hipFree(t.skey);
    t = TileLists();
}

hipError_t launch_tilescan(hipStream_t st, TileLists& t, size_t ntiles, uint32_t* total, uint32_t* total_host, size_t cap, uint32_t hint) {
    // the list starts are laid out for `tiles_cap` tiles (tstart has tiles_cap + 1 words): a smaller frame uses a prefix
    k_tilescan<<<dim3(1), dim3(SCAN_THREADS), 0, st>>>(t.tcount, (uint32_t)ntiles, t.tstart, t.cursor, total, total_host, (uint32_t)std::min<size_t>(cap, 0xFFFFFFFFull), hint);
    return hipGetLastError();
}

hipError_t launch_tile_scatter(hipStream_t st, TileLists& t, const uint2* rects, size_t nrecords, const uint32_t* total, uint2* entries, int tiles_x, int shard_rank, int shard_world) {
    if (nrecords == 0) return hipSuccess;
    k_tile_scatter<<<dim3((unsigned)((nrecords + 255) / 256)), dim3(256), 0, st>>>(rects, t.skey, (uint32_t)nrecords, t.cursor, total, entries, (uint32_t)tiles_x, (uint32_t)shard_rank, (uint32_t)shard_world);
    return hipGetLastError();
}

} // namespace gs4d
