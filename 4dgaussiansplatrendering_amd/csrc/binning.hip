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
// Counting, scanning and emitting are ONE launch (chained scan over workgroup totals); the per-record tile rectangles are read
// from a compact 4-byte array (pack_trect, gs4d_internal.h) instead of the 64-byte projected records — gathered through the sort index,
// or, when the depth sort carried them along as a second payload (a draw that generated its own keys), streamed in instance order.
#include "gs4d_internal.h"
#include <cstdlib>

namespace gs4d {

constexpr int BIN_THREADS = 512;     // 2048 instances per workgroup: one ticket (a same-address atomic, ~11 ns each chip-wide) per 2048
constexpr int BIN_ITEMS = 4;         // (8 per thread held 154 VGPRs = ONE workgroup per CU: gathers, scan, look-back and emit of a CU ran one after the other.
                                     //  4: 90 VGPRs, two workgroups per CU, 10^7 rectangle gathers in 248 us instead of 322 = 84 % of the device's gather rate)
constexpr int BIN_WAVES = BIN_THREADS / 64;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(v, off, 64);
        if ((threadIdx.x & 63) >= (unsigned)off) v += t;
    }
    return v;
}

// exclusive scan over the threads of a block; returns the exclusive prefix, total = block sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wsum /* __shared__[BIN_WAVES] */, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    __syncthreads();                      // protect wsum from the previous round
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < BIN_WAVES; ++k) { const uint32_t s = wsum[k]; if ((unsigned)k < w) base += s; tot += s; }
    total = tot;
    return base + inc - v;
}

struct Rect { uint32_t tx0, ty0, tx1, ty1, count, tstep; };     // tile rows ty0, ty0 + tstep, ... <= ty1 (tstep > 1: this context owns every tstep-th row)

// writes the entries of one splat and counts their tile-id digits for the radix sort that follows (no separate histogram launch)
__device__ __forceinline__ void emit_tiles(const Rect& r, uint32_t off, uint32_t rec, uint32_t tiles_x, uint32_t first, uint32_t stride,
                                           uint32_t* __restrict__ pk, uint32_t* __restrict__ pv, uint32_t (*h)[OS_MAX_BINS], int passes, int rb) {
    const uint32_t wx = r.tx1 - r.tx0 + 1u;
    for (uint32_t j = first; j < r.count; j += stride) {
        const uint32_t ty = r.ty0 + (j / wx) * r.tstep, tx = r.tx0 + j % wx;
        const uint32_t id = ty * tiles_x + tx;
        pk[off + j] = id;
        pv[off + j] = rec;
        if (h) for (int p = 0; p < passes; ++p) atomicAdd(&h[p][(id >> (rb * p)) & ((1u << rb) - 1u)], 1u);
    }
}

// status[b] is one 64-bit {epoch:22, flag:2, value:40} granule per workgroup, written/polled with agent-scope relaxed atomics (same
// hand-off form as the radix sort's look-back words).  The epoch (one per launch) makes the words of earlier draws read as "not
// published", so they are never zeroed.
constexpr unsigned long long BS_VAL = (1ull << 40) - 1ull;
__device__ __forceinline__ unsigned long long bs_word(uint32_t epoch, unsigned long long flag, unsigned long long v) { return ((unsigned long long)epoch << 42) | (flag << 40) | (v > BS_VAL ? BS_VAL : v); }
__device__ __forceinline__ uint32_t bs_flag(unsigned long long w, uint32_t epoch) { return (uint32_t)(w >> 42) == epoch ? (uint32_t)(w >> 40) & 3u : 0u; }

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(BIN_THREADS) void k_bin_emit(const uint32_t* __restrict__ trects, int trects_in_order, const float4* __restrict__ proj, const uint32_t* __restrict__ order, uint32_t* __restrict__ order_copy, uint32_t ninst, uint32_t nrecords,
                                                          unsigned long long* status, unsigned long long* gstatus, uint32_t* __restrict__ total, uint32_t cap, uint32_t tiles_x,
                                                          uint32_t* __restrict__ pk, uint32_t* __restrict__ pv, uint32_t* err,
                                                          uint32_t* __restrict__ ghist, int passes_rb /* passes | digit bits << 8 of the tile sort that follows */, uint32_t* __restrict__ total_host, uint32_t epoch, uint32_t* ticket, uint32_t ticket_base, int dbg_arg,
                                                          uint32_t shard_rank, uint32_t shard_world /* tile row ty is ours iff ty % shard_world == shard_rank */) {
#ifdef GS4D_TUNING
    const int dbg = dbg_arg;             // ablation bits (GS4D_EMIT_DBG): tuning builds only (make TUNING=1)
#else
    constexpr int dbg = 0; (void)dbg_arg;
#endif
    __shared__ uint32_t wsum[BIN_WAVES];
    __shared__ unsigned long long s_prefix;
    __shared__ uint32_t s_blk;
    __shared__ uint32_t h[OS_MAX_PASSES][OS_MAX_BINS];
    const int passes = passes_rb & 255, rb = passes_rb >> 8;
    if (threadIdx.x < 256u) os_hist_clear(h, threadIdx.x);
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    // the slice of instances is handed out by ticket (start order), so the look-back only ever waits for running workgroups
    if (tid == 0) s_blk = atomicAdd(ticket, 1u) - ticket_base;
    __syncthreads();
    const uint32_t blk = s_blk;
    const uint32_t base = blk * (BIN_THREADS * BIN_ITEMS);
    Rect r[BIN_ITEMS]; uint32_t rec[BIN_ITEMS], off[BIN_ITEMS];
    uint32_t run = 0;
#pragma unroll
    for (int j = 0; j < BIN_ITEMS; ++j) {
        const uint32_t k = base + j * BIN_THREADS + tid;
        r[j] = Rect{ 0, 0, 0, 0, 0, 1 }; rec[j] = 0;
        if (k < ninst) {
            rec[j] = order ? order[k] : k;
            if (order_copy) order_copy[k] = rec[j];      // the draw keeps its own copy: the caller may refill the buffer for the next frame
            if (rec[j] < nrecords) {              // an index past the bound SSBO: GL would read undefined data; we draw nothing
                const uint32_t word = (dbg & 1) ? ((k % 230u) | ((k % 130u) << 10)) : trects[trects_in_order ? k : rec[j]];
                const TRect t = unpack_trect(word, proj, rec[j], shard_rank, shard_world);      // only the tile rows this context owns (single-frame sharding over several GPUs)
                if (t.count) { r[j].tx0 = t.tx0; r[j].ty0 = t.ty0; r[j].tx1 = t.tx0 + t.wx - 1u; r[j].ty1 = t.ty0 + (t.rows - 1u) * t.tstep; r[j].tstep = t.tstep; r[j].count = t.count; }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < BIN_ITEMS; ++j) { uint32_t tot; off[j] = run + block_excl_scan(r[j].count, wsum, tot); run += tot; }
    // Publish this workgroup's total, then look back for the exclusive prefix (wave 0).  Two levels, as in the radix sort: the 64
    // workgroups of a group are read in one wave-wide load; the last workgroup of a group also publishes the group's total (AGG)
    // and, once it knows it, the inclusive prefix at the end of the group (INCL).  When ~1000 workgroups start together a
    // one-level walk needs one memory round trip per 64 predecessors; this needs two or three in all.
    if (tid < 64) {
        unsigned long long prefix = 0;
        const uint32_t grp = blk / 64u, r = blk % 64u;
        uint32_t spins = 0;
        bool failed = false;
        if (lane == 0) __hip_atomic_store(status + blk, bs_word(epoch, 1ull, run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (r > 0) {                                              // workgroups grp*64 .. blk-1
            while (true) {
                const unsigned long long v = lane < r ? __hip_atomic_load(status + (blk - 1u - lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : bs_word(epoch, 1ull, 0ull);
                if (__ballot(bs_flag(v, epoch) == 0u) == 0ull) { prefix = wave_sum_u64(v & BS_VAL); break; }
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 21)) { failed = true; break; }
            }
        }
        if (r == 63u && lane == 0) __hip_atomic_store(gstatus + grp, bs_word(epoch, 1ull, prefix + run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int64_t p = (int64_t)grp - 1;
        while (p >= 0 && !failed) {                              // groups before this one, 64 per step, until an INCL
            const int64_t idx = p - (int64_t)lane;
            const unsigned long long v = idx >= 0 ? __hip_atomic_load(gstatus + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : bs_word(epoch, 2ull, 0ull);   // before group 0: inclusive 0
            const uint32_t f = bs_flag(v, epoch);
            const uint64_t none = __ballot(f == 0u), incl = __ballot(f == 2u);
            const uint64_t low = incl & (0ull - incl);                                     // nearest INCL
            const uint64_t upto = incl ? (low | (low - 1ull)) : ~0ull;                       // lanes up to and including it
            if (none & upto) {                                    // a needed word is not published yet
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 21)) failed = true;
                continue;
            }
            prefix += wave_sum_u64(((upto >> lane) & 1ull) ? (v & BS_VAL) : 0ull);
            if (incl) break;
            p -= 64;
        }
        if (failed && lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (r == 63u && lane == 0) __hip_atomic_store(gstatus + grp, bs_word(epoch, 2ull, prefix + run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) {
            s_prefix = prefix;
            if (blk == gridDim.x - 1) {                     // the last workgroup knows the grand total
                const unsigned long long g = prefix + run;
                total[0] = g > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)g;
                total[1] = g > (unsigned long long)cap ? 1u : 0u;
                total[2] = (uint32_t)g; total[3] = (uint32_t)(g >> 32);
                // copy for the host (pinned, mapped): read after the draw's event, no copy launch
                total_host[0] = total[0]; total_host[1] = total[1]; total_host[2] = total[2]; total_host[3] = total[3];
            }
        }
    }
    __syncthreads();
    const unsigned long long pre = s_prefix;
#pragma unroll
    for (int j = 0; j < BIN_ITEMS; ++j) {
        const unsigned long long o64 = pre + off[j];
        const bool fits = o64 + r[j].count <= (unsigned long long)cap;      // entries beyond the capacity are not written; the draw is re-run
        const uint32_t o = (uint32_t)o64;
        const bool big = fits && r[j].count > 32u;
        if (fits && !big && !(dbg & 4)) emit_tiles(r[j], o, rec[j], tiles_x, 0u, 1u, pk, pv, (dbg & 2) ? nullptr : h, passes, rb);
        // large footprints: the whole wave writes one splat's entries
        uint64_t m = __ballot(big);
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            Rect rr;
            rr.tx0 = __shfl(r[j].tx0, src, 64); rr.ty0 = __shfl(r[j].ty0, src, 64); rr.tx1 = __shfl(r[j].tx1, src, 64); rr.ty1 = __shfl(r[j].ty1, src, 64);
            rr.count = __shfl(r[j].count, src, 64); rr.tstep = __shfl(r[j].tstep, src, 64);
            const uint32_t o2 = __shfl(o, src, 64), rec2 = __shfl(rec[j], src, 64);
            emit_tiles(rr, o2, rec2, tiles_x, lane, 64u, pk, pv, h, passes, rb);
        }
    }
    __syncthreads();
    if (tid < 256u) os_hist_flush(h, ghist, passes, tid);
}

// ranges[2t], ranges[2t+1] = first and one-past-last entry of tile t in the sorted tile lists.  Four entries per thread.
__global__ __launch_bounds__(256) void k_tile_ranges(const uint32_t* __restrict__ pk, const uint32_t* __restrict__ total, uint32_t ntiles, uint32_t* __restrict__ ranges) {
    if (total[1]) return;
    const uint32_t m = total[0];
    const uint32_t j0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (j0 >= m) return;
    uint32_t k[4];
    if (j0 + 4u <= m) { const uint4 v = *reinterpret_cast<const uint4*>(pk + j0); k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w; }
    else { for (int q = 0; q < 4; ++q) k[q] = j0 + q < m ? pk[j0 + q] : 0xFFFFFFFFu; }
    uint32_t prev = j0 ? pk[j0 - 1u] : 0xFFFFFFFFu;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t j = j0 + q, cur = k[q];
        if (j < m && cur < ntiles) {
            if (j == 0) ranges[2 * cur] = 0;
            else if (prev != cur) { ranges[2 * cur] = j; if (prev < ntiles) ranges[2 * prev + 1] = j; }
            if (j == m - 1) ranges[2 * cur + 1] = m;
        }
        prev = cur;
    }
}

hipError_t bin_scratch_reserve(hipStream_t st, BinScratch& b, size_t ninst, size_t ntiles) {
    hipError_t e;
    size_t nb = (ninst + BIN_THREADS * BIN_ITEMS - 1) / (BIN_THREADS * BIN_ITEMS);
    if (nb < 1) nb = 1;
    if (!b.total) {
        if (const char* e0 = getenv("GS4D_TEST_EPOCH0")) b.epoch = ((uint32_t)strtoul(e0, nullptr, 0) << 4) | 0xFu;     // test hook: 22-bit wrap within a short test
        if ((e = hipMalloc(&b.total, 64)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(b.total, 0, 64, st)) != hipSuccess) return e;
        b.ticket_base = 0;
    }
    if (b.block_cap < nb || b.tiles_cap < ntiles) {
        if (b.ranges) { (void)hipStreamSynchronize(st); (void)hipFree(b.ranges); }
        const size_t nb2 = nb > b.block_cap ? nb : b.block_cap, nt2 = ntiles > b.tiles_cap ? ntiles : b.tiles_cap;
        b.ranges = nullptr; b.status = nullptr; b.block_cap = b.tiles_cap = 0;
        // one allocation: [ranges: 2*tiles u32][status: blocks u64][group status: blocks/64 u64]
        const size_t ng = nb2 / 64 + 2;
        if ((e = hipMalloc(&b.ranges, nt2 * 8 + (nb2 + ng) * 8)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(b.ranges, 0, nt2 * 8 + (nb2 + ng) * 8, st)) != hipSuccess) return e;    // ranges stay zero between draws: the composite clears what it reads
        b.status = reinterpret_cast<unsigned long long*>(b.ranges + 2 * nt2);
        b.block_cap = nb2; b.tiles_cap = nt2;
    }
    return hipSuccess;
}

void bin_scratch_free(BinScratch& b) {
    if (b.total) (void)hipFree(b.total);
    if (b.ranges) (void)hipFree(b.ranges);
    b = BinScratch();
}

hipError_t launch_binning(hipStream_t st, BinScratch& b, const uint32_t* trects, bool trects_in_order, const float4* proj, const uint32_t* order, uint32_t* order_copy, size_t ninst, size_t nrecords, int tiles_x, int tiles_y,
                          uint32_t* pair_keys, uint32_t* pair_vals, size_t pair_cap, uint32_t* err, uint32_t* ghist, int passes, uint32_t* total_host, int shard_rank, int shard_world) {
    (void)tiles_y;
#ifdef GS4D_TUNING
    static const int dbg = getenv("GS4D_EMIT_DBG") ? atoi(getenv("GS4D_EMIT_DBG")) : 0;      // tuning aid (ablation): 1 no rect gather, 2 no histogram, 4 no writes
#else
    const int dbg = 0;
#endif
    if (++b.epoch >= (1u << 22)) {            // epoch wrap: forget every old word
        hipError_t e = hipMemsetAsync(b.status, 0, (b.block_cap + b.block_cap / 64 + 2) * 8, st);
        if (e != hipSuccess) return e;
        b.epoch = 1;
    }
    const uint32_t nb = (uint32_t)((ninst + BIN_THREADS * BIN_ITEMS - 1) / (BIN_THREADS * BIN_ITEMS));
    k_bin_emit<<<dim3(nb), dim3(BIN_THREADS), 0, st>>>(trects, trects_in_order ? 1 : 0, proj, order, order_copy, (uint32_t)ninst, (uint32_t)nrecords, b.status, b.status + b.block_cap, b.total, (uint32_t)pair_cap, (uint32_t)tiles_x, pair_keys, pair_vals, err, ghist, passes, total_host, b.epoch, b.total + 8, b.ticket_base, dbg, (uint32_t)shard_rank, (uint32_t)shard_world);
    b.ticket_base += nb;
    return hipGetLastError();
}

hipError_t launch_tile_ranges(hipStream_t st, BinScratch& b, const uint32_t* pair_keys, size_t pair_cap, size_t ntiles) {
    k_tile_ranges<<<dim3((unsigned)((pair_cap + 1023) / 1024)), dim3(256), 0, st>>>(pair_keys, b.total, (uint32_t)ntiles, b.ranges);
    return hipGetLastError();
}

} // namespace gs4d
