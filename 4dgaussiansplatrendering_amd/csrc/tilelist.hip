// tilelist.hip — per-tile splat lists for the unordered draw path.  No workgroup waits for another, no global atomics.
//
// The reference has no tiles: the hardware rasteriser walks the instances in order and the ROP blends in that order
// (Renderer.cpp:33-39 -> glDrawElementsInstanced; Application.cpp:150-154).  The ordered path (binning.hip) reproduces that order by
// emitting (tile, record) entries in instance order — a chained scan — and stable-sorting them by tile: four launches, three of
// them chained scans whose cost at 10^6 splats is hand-off latency, not bandwidth.  Here the order is restored where it is consumed
// (composite2.hip sorts each tile's list by (key, record) in LDS), so the lists may be built in ANY order — a two-level bucket
// sort by tile id in which every write position is computed, never negotiated:
//
//   bucket b = tile % nb   (interleaved: the dense image centre spreads over all buckets),   nb a power of two, tile / nb < 256
//
//   k_preprocess<.., true>  (preprocess.hip) walks segment w of the records (one workgroup, `seg` records) and leaves
//                           hist[b][w] = entries its records put into bucket b            (LDS atomics, plain stores)
//   k_bucket_scan           one wave per bucket: exclusive scan along w -> offs[w][b], the slot of run (w, b) inside bucket b; the workgroup
//                           that finishes last turns the bucket totals into bucket starts and checks the capacity
//   k_bucket_scatter        walks the same segments again: entry -> tmp[bucket start + run slot + LDS counter]   (key, tile/nb << 24 | record)
//   k_bucket_tiles          one workgroup per bucket: counts its entries per tile in LDS, writes the tile table (first entry, count)
//                           and moves the entries to their tile's list; the longest list is checked against the compositor's capacity
//                           (the verdict travels to the host through the first workgroup of the compositing kernel)
//   k_composite_v2          (composite2.hip)
//
// Staged lists (round 4): once a draw of a scene has reported its fullest segment, longest (bucket, segment) run and fullest bucket, the draws
// that follow skip the scan and the scatter: k_project_count<.., 2> counts, scans and PLACES its segment's entries in LDS and writes them out as
// one dense block, k_bucket_tiles_staged reads bucket b's run out of every block.  Capacities are guesses (statistics + margin) that the
// device checks; a draw that does not fit raises total[TL_ABORT_WORD] and is re-run with the exact kernels above.
//
// (A first version counted entries per tile with global atomics in the projection kernel and scattered with returning atomics:
// 1.4e6 device-scope atomics cost 150 us each way on MI355X — they execute at the memory side, ~9e9/s when scattered.  Hence this.)
// Which key gives "instance order" is decided on the host (KeySrc, gs4d_internal.h).  Lists longer than the compositor can hold, or
// more entries than the preallocated capacity, raise a flag: the later kernels then do nothing and the host re-runs the draw
// (larger capacity, longer lists, or the ordered path).
#include "gs4d_internal.h"
#include <algorithm>
#include <cstdlib>

namespace gs4d {

typedef unsigned long long u64;


// exclusive scan of the bucket totals by a 256-thread workgroup (nb <= 1024): bucket starts into s_base[0..nb], returns the grand total
__device__ __forceinline__ unsigned long long bucket_bases(const uint32_t* btot, uint32_t nb, uint32_t* s_base /* [1025] */, unsigned long long* s_ws /* [4] */) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    uint32_t v[4]; unsigned long long sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const uint32_t b = tid * 4u + k; v[k] = b < nb ? __hip_atomic_load(btot + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u; sum += v[k]; }
    unsigned long long inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned long long t = __shfl_up(inc, off, 64); if (lane >= (unsigned)off) inc += t; }
    if (lane == 63u) s_ws[w] = inc;
    __syncthreads();
    unsigned long long base = 0, grand = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const unsigned long long t = s_ws[k]; if ((unsigned)k < w) base += t; grand += t; }
    unsigned long long run = base + inc - sum;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const uint32_t b = tid * 4u + k; if (b < nb) s_base[b] = run > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)run; run += v[k]; }
    if (tid == 0) s_base[nb] = grand > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)grand;
    __syncthreads();
    return grand;
}

// hist[b][w] (counts, one row per bucket) -> offs[w][b] = slot of run (w, b) inside bucket b (exclusive scan along w); btot[b] = entries of
// bucket b.  One wave per bucket; the <= 1024 counts of a bucket are all requested before any is used (a single workgroup walking the
// whole matrix, load after dependent load, cost 100 us).  The workgroup that finishes last turns the bucket totals into bucket starts and
// checks the capacity — for the kernels and the host that come after.
__global__ __launch_bounds__(256) void k_bucket_scan(const uint32_t* __restrict__ hist, uint32_t rows, uint32_t nb, uint32_t* __restrict__ offs, uint32_t* __restrict__ btot,
                                                     uint32_t* __restrict__ bbase, uint32_t* __restrict__ total, uint32_t* __restrict__ total_host, uint32_t cap, uint4* __restrict__ bstat) {
    __shared__ uint32_t s_base[1025];
    __shared__ unsigned long long s_ws[4];
    __shared__ uint32_t s_last;
    const uint32_t lane = threadIdx.x & 63u, b = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (b < nb) {
        const uint32_t* __restrict__ h = hist + (size_t)b * rows;
        uint32_t c[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { const uint32_t w = (uint32_t)j * 64u + lane; c[j] = w < rows ? h[w] : 0u; }      // rows <= 1024 (tile_lists_plan)
        uint32_t carry = 0, mxr = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if ((uint32_t)j * 64u >= rows) break;
            mxr = max(mxr, c[j]);
            uint32_t inc = c[j];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(inc, off, 64); if (lane >= (unsigned)off) inc += v; }
            const uint32_t w = (uint32_t)j * 64u + lane;
            if (w < rows) offs[(size_t)w * nb + b] = carry + inc - c[j];
            carry += __shfl(inc, 63, 64);
        }
        if (lane == 0) __hip_atomic_store(btot + b, carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // what a later draw of this scene needs to size its staged blocks (the compositing kernel's first workgroup reduces these for the host)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mxr = max(mxr, (uint32_t)__shfl_xor(mxr, off, 64));
        if (lane == 0) bstat[b] = make_uint4(carry, mxr, 0u, 0u);
    }
    // the bucket totals were stored write-through (agent-scope atomic stores) and are read back with agent-scope loads: all that is
    // needed before the arrival counter is that this workgroup's stores have left (s_waitcnt) — no cache write-back, which would push
    // out the XCD's whole L2 once per workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool last = atomicAdd(&total[7], 1u) == gridDim.x - 1u;
        if (last) total[7] = 0u;
        s_last = last ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    const unsigned long long grand = bucket_bases(btot, nb, s_base, s_ws);
    for (uint32_t k = threadIdx.x; k <= nb; k += 256u) bbase[k] = s_base[k];
    if (threadIdx.x == 0) {
        const bool over = grand > (unsigned long long)cap;
        const uint32_t sat = grand > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)grand;
        total[0] = sat; total[1] = over ? 1u : 0u; total[2] = (uint32_t)grand; total[3] = (uint32_t)(grand >> 32); total[4] = 0u; total[6] = 0u;
        // an aborted draw stops here: tell the host now (otherwise k_bucket_tiles reports, once the longest list is known)
        if (over) { total_host[0] = sat; total_host[1] = 1u; total_host[2] = (uint32_t)grand; total_host[3] = (uint32_t)(grand >> 32); total_host[5] = 0u; }
    }
}

// Segment w again: every entry goes to tmp[bucket start + slot of run (w, bucket) + a counter in LDS].  SEG_THREADS threads, one record each per round.
__global__ __launch_bounds__(SEG_THREADS) void k_bucket_scatter(const uint32_t* __restrict__ trects, const float4* __restrict__ proj, const uint32_t* __restrict__ skey, uint32_t skey_bias, uint32_t n, uint32_t seg, uint32_t nb,
                                                        const uint32_t* __restrict__ offs, const uint32_t* __restrict__ bbase, const uint32_t* __restrict__ total, uint2* __restrict__ tmp,
                                                        uint32_t tiles_x, uint32_t shard_rank, uint32_t shard_world) {
    __shared__ uint32_t cur[1024];
    if (total[1] & 1u) return;                             // aborted draw: slots would lie beyond the capacity
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    for (uint32_t b = tid; b < nb; b += SEG_THREADS) cur[b] = bbase[b] + offs[(size_t)blockIdx.x * nb + b];
    __syncthreads();
    const uint32_t nbm = nb - 1u, nbs = (uint32_t)__ffs((int)nb) - 1u;
    const uint32_t i0 = blockIdx.x * seg, i1 = min(n, i0 + seg);
    constexpr int SC_ITEMS = 4;                            // records per thread and round: all their loads are in flight before the first LDS atomic
    for (uint32_t ib = i0; ib < i1; ib += SEG_THREADS * SC_ITEMS) {      // uniform trip count: every lane stays to the end
        uint32_t rc[SC_ITEMS], kk[SC_ITEMS];
#pragma unroll
        for (int q = 0; q < SC_ITEMS; ++q) {
            const uint32_t i = ib + (uint32_t)q * SEG_THREADS + tid;
            rc[q] = i < i1 ? trects[i] : TRECT_NONE;
            kk[q] = i < i1 ? skey[i] - skey_bias : 0u;                        // (a draw that generated the depth keys itself reads them where it wrote them: the caller's key buffer, bit patterns above the bias)
        }
#pragma unroll
        for (int q = 0; q < SC_ITEMS; ++q) {
            const uint32_t i = ib + (uint32_t)q * SEG_THREADS + tid;
            const TRect r = unpack_trect(rc[q], proj, i, shard_rank, shard_world);
            const uint32_t key = kk[q];
            const bool big = r.count > 16u;
            if (!big) for_each_tile(r, tiles_x, [&](uint32_t t) {
                const uint32_t pos = atomicAdd(&cur[t & nbm], 1u);
                tmp[pos] = make_uint2(key, ((t >> nbs) << 24) | i);
            });
            uint64_t m = __ballot(big);
            while (m) {                                    // large footprints: the whole wave writes one record's entries
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1ull;
                TRect rr;
                rr.tx0 = __shfl(r.tx0, src, 64); rr.ty0 = __shfl(r.ty0, src, 64); rr.wx = __shfl(r.wx, src, 64);
                rr.rows = __shfl(r.rows, src, 64); rr.tstep = __shfl(r.tstep, src, 64); rr.count = __shfl(r.count, src, 64);
                const uint32_t key2 = __shfl(key, src, 64), rec2 = __shfl(i, src, 64);
                for (uint32_t j = lane; j < rr.count; j += 64u) {
                    const uint32_t t = tile_of(rr, j, tiles_x);
                    const uint32_t pos = atomicAdd(&cur[t & nbm], 1u);
                    tmp[pos] = make_uint2(key2, ((t >> nbs) << 24) | rec2);
                }
            }
        }
    }
}

// Bucket b -> the lists of its tiles (tile = h * nb + b, h < tpb <= 256): count per list, scan, place.  A tile's list is kept as `slabs`
// sub-lists by depth (far slab first; slab of an entry = min(slabs - 1, key >> slab_shift)): counter = h * slabs + slab.  The compositor
// orders every sub-list by itself in LDS, so `slabs` is what bounds the list it has to hold, not the tile's whole list (10^7 splats at
// 1080p put up to 4 400 entries on a tile).  The counters live in dynamic LDS (tpb * slabs of them, at most BT_MAX_COUNTERS).
// A thread keeps up to 8 entries in registers (all loads in flight at once); a bucket of up to 8192 entries is read once, a longer one
// in rounds of 8192, twice.
constexpr int BT_THREADS = 1024, BT_ITEMS = 8;
constexpr uint32_t BT_MAX_COUNTERS = 12288;      // 48 KB of dynamic LDS beside at most 8.2 KB of static (k_bucket_tiles_staged): inside the 64 KB a workgroup may have without an attribute; tile_lists_plan sends larger frames to the ordered path
__global__ __launch_bounds__(BT_THREADS) void k_bucket_tiles(const uint2* __restrict__ tmp, const uint32_t* __restrict__ bbase, uint32_t nb, uint32_t ntiles, uint32_t slabs, uint32_t slab_shift, uint32_t nc,
                                                             uint32_t* __restrict__ tstart, uint32_t* __restrict__ tcnt, uint2* __restrict__ entries,
                                                             uint32_t* __restrict__ total, uint32_t hint) {
    extern __shared__ uint32_t cnt[];                      // [nc]
    __shared__ uint32_t ws[BT_THREADS / 64];
    if (total[1] & 1u) return;                             // capacity overflow (set by k_bucket_scan); bit 1 is raised HERE by other workgroups and must not stop this one
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6, b = blockIdx.x;
    const uint32_t lo = bbase[b], hi = bbase[b + 1];
    const uint32_t sl = (uint32_t)__ffs((int)slabs) - 1u;  // log2(slabs)
    constexpr uint32_t ROUND = BT_THREADS * BT_ITEMS;
    const bool single = hi - lo <= ROUND;
    for (uint32_t k = tid; k < nc; k += BT_THREADS) cnt[k] = 0u;
    __syncthreads();
    auto counter_of = [&](const uint2& e) { return min(nc - 1u, ((e.y >> 24) << sl) | min(slabs - 1u, e.x >> slab_shift)); };      // (the outer min only keeps a corrupt entry inside LDS)
    uint2 e[BT_ITEMS];
    for (uint32_t r0 = lo; r0 < hi; r0 += ROUND) {
#pragma unroll
        for (int j = 0; j < BT_ITEMS; ++j) { const uint32_t i = r0 + (uint32_t)j * BT_THREADS + tid; e[j] = i < hi ? tmp[i] : make_uint2(0u, 0xFFFFFFFFu); }
#pragma unroll
        for (int j = 0; j < BT_ITEMS; ++j) if (e[j].y != 0xFFFFFFFFu) atomicAdd(&cnt[counter_of(e[j])], 1u);
    }
    __syncthreads();
    // exclusive scan of the nc counters (thread t owns counters [t * cpt, (t + 1) * cpt)), the tile table, the longest sub-list
    const uint32_t cpt = (nc + BT_THREADS - 1u) / BT_THREADS, q0 = tid * cpt;
    uint32_t sum = 0, mx = 0;
    for (uint32_t k = 0; k < cpt; ++k) { const uint32_t q = q0 + k; const uint32_t c = q < nc ? cnt[q] : 0u; sum += c; mx = max(mx, c); }
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(inc, off, 64); if (lane >= (unsigned)off) inc += v; }
    if (lane == 63u) ws[w] = inc;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, off, 64));
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int k = 0; k < BT_THREADS / 64; ++k) if ((unsigned)k < w) base += ws[k];
    uint32_t run = lo + base + inc - sum;
    for (uint32_t k = 0; k < cpt; ++k) {
        const uint32_t q = q0 + k;
        if (q < nc) {
            const uint32_t c = cnt[q];
            cnt[q] = run;                                  // from here on: the sub-list's running position (only this thread touches counter q in this phase)
            const uint32_t tile = (q >> sl) * nb + b;
            if (tile < ntiles) { tstart[(size_t)tile * slabs + (q & (slabs - 1u))] = run; tcnt[(size_t)tile * slabs + (q & (slabs - 1u))] = c; }
            run += c;
        }
    }
    if (lane == 0u && mx) { atomicMax(&total[4], mx); if (mx > hint) atomicOr(&total[1], 2u); }
    __syncthreads();
    for (uint32_t r0 = lo; r0 < hi; r0 += ROUND) {
        if (!single) {
#pragma unroll
            for (int j = 0; j < BT_ITEMS; ++j) { const uint32_t i = r0 + (uint32_t)j * BT_THREADS + tid; e[j] = i < hi ? tmp[i] : make_uint2(0u, 0xFFFFFFFFu); }
        }
#pragma unroll
        for (int j = 0; j < BT_ITEMS; ++j) if (e[j].y != 0xFFFFFFFFu) {
            const uint32_t pos = atomicAdd(&cnt[counter_of(e[j])], 1u);
            entries[pos] = make_uint2(e[j].x, e[j].y & 0x00FFFFFFu);
        }
    }
    // (the entry count, the longest list and the flags reach the host through the compositing kernel that follows: no hand-off here)
}

// Staged draws: no scan and no scatter kernel ran.  The projection kernel (preprocess.hip, k_project_count<.., 2>) wrote every segment's entries as
// one dense block, bucket after bucket: bucket b's run of segment w is blocks[w * scap + offs[b][w] + k], k < hist[b][w].  This workgroup reads its
// bucket's counts and offsets, turns the counts into a prefix (entry i of the bucket -> run, slot), and gives every thread the same number of
// CONSECUTIVE entries of the bucket (ceil(T / THREADS) <= KMAX, kept in registers between counting and placing): one binary search for a thread's first
// entry, then it walks on through the runs.  (A first version gave every thread fixed 8-slot chunks of the runs — capacity for the longest run in every
// run: 28 % of the lanes carried an entry in the LDS-atomic loops, 80 registers of entries per thread, and the longest run was one more guess that could
// miss.)  It checks what the host only guessed — bcap entries in the bucket, `hint` in a tile's list — and otherwise does what k_bucket_tiles does.
// The tile lists of bucket b go to entries[b * bcap ...].  Statistics {entries, longest run, longest list} go to bstat[b]; a bucket that does not fit
// stores the draw's sequence number into *abort_word (every writer stores the same value).
template <int KMAX, int THREADS>
__global__ __launch_bounds__(THREADS) void k_bucket_tiles_staged(const uint2* __restrict__ blocks, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offs, uint32_t rows, uint32_t scap,
                                                                    uint32_t bcap, uint32_t nb, uint32_t ntiles, uint32_t slabs, uint32_t slab_shift, uint32_t nc, uint32_t* __restrict__ tstart,
                                                                    uint32_t* __restrict__ tcnt, uint2* __restrict__ entries, uint4* __restrict__ bstat, uint32_t* __restrict__ abort_word, uint32_t seq, uint32_t hint,
                                                                    uint32_t tiles_x, uint32_t box) {
    extern __shared__ uint32_t cnt[];                      // [nc]
    __shared__ uint32_t ws[THREADS / 64], wm[THREADS / 64], wb[THREADS / 64];
    __shared__ uint32_t rp[1025], ro[1024];                // rp[w]: entries of the bucket before run w (rp[rows] = all of them); ro[w]: where run w starts in `blocks` (rows <= 1024: tile_lists_plan)
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6, b = blockIdx.x;
    const uint32_t sl = (uint32_t)__ffs((int)slabs) - 1u;
    for (uint32_t k = tid; k < nc; k += THREADS) cnt[k] = 0u;
    // ---- the bucket's counts -> prefix over the runs; total, longest run.  Thread t looks after runs [t * RPT, (t + 1) * RPT) ----
    constexpr uint32_t RPT = 1024u / THREADS;
    uint32_t c[RPT], tsum = 0, tmax = 0;
#pragma unroll
    for (uint32_t k = 0; k < RPT; ++k) {
        const uint32_t wseg = tid * RPT + k;
        c[k] = wseg < rows ? hist[(size_t)b * rows + wseg] : 0u;
        ro[wseg] = wseg < rows ? wseg * scap + offs[(size_t)b * rows + wseg] : 0u;
        tsum += c[k]; tmax = max(tmax, c[k]);
    }
    uint32_t pinc = tsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(pinc, off, 64); if (lane >= (unsigned)off) pinc += v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tmax = max(tmax, (uint32_t)__shfl_xor(tmax, off, 64));
    if (lane == 63u) ws[w] = pinc;
    if (lane == 0u) wm[w] = tmax;
    __syncthreads();
    uint32_t T = 0, maxrun = 0, pbase = 0;
#pragma unroll
    for (int k = 0; k < THREADS / 64; ++k) { if ((unsigned)k < w) pbase += ws[k]; T += ws[k]; maxrun = max(maxrun, wm[k]); }
    {
        uint32_t run = pbase + pinc - tsum;
#pragma unroll
        for (uint32_t k = 0; k < RPT; ++k) { rp[tid * RPT + k] = run; run += c[k]; }
        if (tid == THREADS - 1u) rp[1024] = run;          // (= T; runs beyond `rows` are empty, so rp[rows..1024] == T)
    }
    // (a segment that overflowed its block wrote no entries and raised the abort word itself; its counts are still true)
    const uint32_t per = (T + THREADS - 1u) / THREADS;     // entries per thread
    const bool fits = T <= bcap && per <= (uint32_t)KMAX;   // uniform
    if (!fits) {
        if (tid == 0u) { bstat[b] = make_uint4(T, maxrun, 0u, BOX_EMPTY); *abort_word = seq; }
        return;
    }
    __syncthreads();                                        // rp / ro complete; ws / wm are reused below
    auto counter_of = [&](const uint2& e) { return min(nc - 1u, ((e.y >> 24) << sl) | min(slabs - 1u, e.x >> slab_shift)); };
    uint2 e[KMAX];
    const uint32_t s0 = min(T, tid * per), s1 = min(T, s0 + per);
    {
        // the run that holds entry s0: the last w with rp[w] <= s0 among 0..1023 (empty runs share their successor's prefix: the walk below skips them)
        uint32_t lo = 0, hi = 1024;
#pragma unroll
        for (int it = 0; it < 10; ++it) { const uint32_t mid = (lo + hi) >> 1; if (rp[mid] <= s0) lo = mid; else hi = mid; }
        uint32_t run = lo, next = rp[min(run + 1u, 1024u)], first = rp[run];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const uint32_t i = s0 + (uint32_t)k;
            if (i < s1) {
                while (i >= next) { ++run; first = next; next = rp[min(run + 1u, 1024u)]; }      // (i < T = rp[1024]: terminates)
                e[k] = blocks[ro[run] + (i - first)];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) if (s0 + (uint32_t)k < s1) atomicAdd(&cnt[counter_of(e[k])], 1u);
    __syncthreads();
    // exclusive scan of the nc counters, the tile table, the longest sub-list (as k_bucket_tiles)
    const uint32_t cpt = (nc + THREADS - 1u) / THREADS, q0c = tid * cpt;
    uint32_t sum = 0, mx = 0;
    for (uint32_t k = 0; k < cpt; ++k) { const uint32_t q = q0c + k; const uint32_t cq = q < nc ? cnt[q] : 0u; sum += cq; mx = max(mx, cq); }
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(inc, off, 64); if (lane >= (unsigned)off) inc += v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, off, 64));
    if (lane == 63u) ws[w] = inc;
    if (lane == 0u) wm[w] = mx;
    __syncthreads();
    uint32_t wbase = 0, longest = 0;
#pragma unroll
    for (int k = 0; k < THREADS / 64; ++k) { if ((unsigned)k < w) wbase += ws[k]; longest = max(longest, wm[k]); }
    const uint32_t lo_e = b * bcap;
    uint32_t runpos = lo_e + wbase + inc - sum;
    uint32_t mybox = BOX_EMPTY;                             // the blocks of this thread's tiles that hold entries
    for (uint32_t k = 0; k < cpt; ++k) {
        const uint32_t q = q0c + k;
        if (q < nc) {
            const uint32_t cq = cnt[q];
            cnt[q] = runpos;
            const uint32_t tile = (q >> sl) * nb + b;
            if (tile < ntiles) { tstart[(size_t)tile * slabs + (q & (slabs - 1u))] = runpos; tcnt[(size_t)tile * slabs + (q & (slabs - 1u))] = cq; }
            if (cq) { const uint32_t bx = min(255u, (tile % tiles_x) / BOX_BLOCK), by = min(255u, (tile / tiles_x) / BOX_BLOCK); mybox = box_join(mybox, box_pack(bx, by, bx, by)); }
            runpos += cq;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mybox = box_join(mybox, (uint32_t)__shfl_xor((int)mybox, off, 64));
    if (lane == 0u) wb[w] = mybox;
    __syncthreads();
    if (tid == 0u) {
        uint32_t bb = BOX_EMPTY;
#pragma unroll
        for (int k = 0; k < THREADS / 64; ++k) bb = box_join(bb, wb[k]);
        bstat[b] = make_uint4(T, maxrun, longest, bb);
        // an entry outside the box the compositor is launched for (a guess, like the capacities): the draw is re-run exactly, over the whole image
        const bool outside = bb != BOX_EMPTY && !(box_holds(box, bb & 255u, (bb >> 8) & 255u) && box_holds(box, (bb >> 16) & 255u, bb >> 24));
        if (longest > hint || outside) *abort_word = seq;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KMAX; ++k) if (s0 + (uint32_t)k < s1) {
        const uint32_t pos = atomicAdd(&cnt[counter_of(e[k])], 1u);
        entries[pos] = make_uint2(e[k].x, e[k].y & 0x00FFFFFFu);
    }
}

bool tile_lists_plan(TileLists& t, size_t ntiles, size_t nrecords, uint32_t slabs, int keybits, uint32_t key_span, size_t expect_entries) {
    if (nrecords == 0 || nrecords > V2_MAX_RECORDS) return false;
    // buckets: enough that tile / nb < 256, and about 8192 entries each (k_bucket_tiles reads a bucket of that size once) at ~1.4 entries per
    // record, or at what an earlier draw of the scene counted
    uint32_t nb = 64;
    const size_t est = expect_entries ? expect_entries + expect_entries / 8 : nrecords + nrecords / 2;
    while (((size_t)nb * 256 < ntiles || (size_t)nb * 8192 < est) && nb < 1024) nb *= 2;
    if (const char* ev = getenv("GS4D_NB")) {            // tuning knob: number of buckets (a power of two, 64..1024), subject to the entry format
        const uint32_t v = (uint32_t)atoi(ev);
        if (v >= 64 && v <= 1024 && (v & (v - 1)) == 0 && (size_t)v * 256 >= ntiles) nb = v;
    }
    if ((size_t)nb * 256 < ntiles) return false;
    // segments of >= 2048 records (at 10^6 records 4096 left the projection with 245 workgroups = 8 waves per CU in flight: 44.6 us; longer
    // segments mean longer runs per bucket and a smaller count matrix), at most 1024 of them (k_bucket_scan keeps a bucket's counts in registers)
    size_t rows = std::min<size_t>((nrecords + 2047) / 2048, 1024);
    size_t seg = ((nrecords + rows - 1) / rows + SEG_THREADS - 1) / SEG_THREADS * SEG_THREADS;
    rows = (nrecords + seg - 1) / seg;
    if (slabs < 1) slabs = 1;
    if (slabs > V2_MAX_SLABS) slabs = V2_MAX_SLABS;
    int sl = 0; while ((1u << sl) < slabs) ++sl;
    if (keybits < sl) return false;
    // the slabs divide [0, key_span] (the host-proven range of the blend keys; 2^keybits - 1 where nothing tighter is known) into equal
    // key ranges: the smallest shift that leaves fewer than `slabs` values — between slabs / 2 and slabs of the sub-lists are then in use
    const uint32_t top = keybits >= 32 ? 0xFFFFFFFFu : ((1u << keybits) - 1u);
    const uint32_t span = std::min(key_span, top);
    uint32_t shift = 0; while (shift < 32u && (span >> shift) >= (1u << sl)) ++shift;
    if (sl == 0) shift = 31u;                            // one slab: whatever the key, slab 0 (min(0, ...))
    const uint32_t tpb = (uint32_t)((ntiles + nb - 1) / nb);
    if (((size_t)tpb << sl) > BT_MAX_COUNTERS) return false;      // k_bucket_tiles keeps one counter per (tile of the bucket, slab) in LDS
    t.nb = nb; t.rows = (uint32_t)rows; t.seg = (uint32_t)seg; t.slabs = 1u << sl; t.slab_shift = std::min(shift, 31u); t.counters = tpb << sl;
    return true;
}

hipError_t tile_lists_reserve(hipStream_t st, TileLists& t, size_t ntiles, size_t nrecords) {
    hipError_t e;
    const size_t hist_words = (size_t)t.rows * t.nb;
    if (t.hist_cap < hist_words) {
        if (t.hist) { (void)hipStreamSynchronize(st); (void)hipFree(t.hist); }
        t.hist = nullptr; t.hist_cap = 0;
        if ((e = hipMalloc(&t.hist, hist_words * 8)) != hipSuccess) return e;      // [nb][rows] counts, then [rows][nb] run slots
        t.hist_cap = hist_words;
    }
    if (t.tiles_cap < ntiles || t.nb_cap < t.nb || t.slabs_cap < t.slabs) {
        if (t.bbase) { (void)hipStreamSynchronize(st); (void)hipFree(t.bbase); }
        t.bbase = t.btot = t.tstart = t.tcnt = nullptr; t.bstat = nullptr; t.sstat = nullptr;
        const size_t nt = std::max(ntiles, t.tiles_cap), nbc = std::max<size_t>(t.nb, t.nb_cap), sc = std::max<size_t>(std::max<size_t>(t.slabs, t.slabs_cap), 4);
        if ((e = hipMalloc(&t.bbase, (2 * nbc + 4 + 2 * nt * sc) * 4 + nbc * 16 + 1024 * 4)) != hipSuccess) return e;     // the tile table holds `slabs` sub-lists per tile
        t.btot = t.bbase + nbc + 1; t.tstart = t.btot + nbc; t.tcnt = t.tstart + nt * sc;
        t.bstat = reinterpret_cast<uint4*>(t.bbase + ((2 * nbc + 1 + 2 * nt * sc + 3) & ~(size_t)3));       // [nbc], 16-byte aligned
        t.sstat = reinterpret_cast<uint32_t*>(t.bstat + nbc);                                                  // [1024]
        if ((e = hipMemsetAsync(t.bstat, 0, nbc * 16 + 1024 * 4, st)) != hipSuccess) return e;
        t.tiles_cap = nt; t.nb_cap = nbc; t.slabs_cap = sc;
    }
    if (t.skey_cap < nrecords) {
        if (t.skey) { (void)hipStreamSynchronize(st); (void)hipFree(t.skey); }
        t.skey = nullptr; t.skey_cap = 0;
        if ((e = hipMalloc(&t.skey, nrecords * 4)) != hipSuccess) return e;
        t.skey_cap = nrecords;
    }
    return hipSuccess;
}

hipError_t tile_lists_reserve_blocks(hipStream_t st, TileLists& t, size_t entries) {
    if (t.blocks_cap >= entries) return hipSuccess;
    if (t.blocks) { (void)hipStreamSynchronize(st); (void)hipFree(t.blocks); }
    t.blocks = nullptr; t.blocks_cap = 0;
    hipError_t e = hipMalloc(&t.blocks, entries * 8);
    if (e != hipSuccess) return e;
    t.blocks_cap = entries;
    return hipSuccess;
}

void tile_lists_free(TileLists& t) {
    if (t.blocks) (void)hipFree(t.blocks);
    if (t.hist) (void)hipFree(t.hist);
    if (t.bbase) (void)hipFree(t.bbase);
    if (t.skey) (void)hipFree(t.skey);
    t = TileLists();
}

hipError_t launch_bucket_scan(hipStream_t st, TileLists& t, uint32_t* total, uint32_t* total_host, size_t cap) {
    k_bucket_scan<<<dim3((t.nb + 3) / 4), dim3(256), 0, st>>>(t.hist, t.rows, t.nb, t.hist + t.hist_cap, t.btot, t.bbase, total, total_host, (uint32_t)std::min<size_t>(cap, 0xFFFFFFFFull), t.bstat);
    return hipGetLastError();
}

hipError_t launch_bucket_scatter(hipStream_t st, TileLists& t, const uint32_t* trects, const float4* proj, const uint32_t* skey, uint32_t skey_bias, size_t nrecords, const uint32_t* total, uint2* tmp, int tiles_x, int shard_rank, int shard_world) {
    k_bucket_scatter<<<dim3(t.rows), dim3(SEG_THREADS), 0, st>>>(trects, proj, skey ? skey : t.skey, skey ? skey_bias : 0u, (uint32_t)nrecords, t.seg, t.nb, t.hist + t.hist_cap, t.bbase, total, tmp, (uint32_t)tiles_x, (uint32_t)shard_rank, (uint32_t)shard_world);
    return hipGetLastError();
}

hipError_t launch_bucket_tiles(hipStream_t st, TileLists& t, size_t ntiles, uint32_t* total, const uint2* tmp, uint2* entries, uint32_t hint) {
    k_bucket_tiles<<<dim3(t.nb), dim3(BT_THREADS), t.counters * 4u, st>>>(tmp, t.bbase, t.nb, (uint32_t)ntiles, t.slabs, t.slab_shift, t.counters, t.tstart, t.tcnt, entries, total, hint);
    return hipGetLastError();
}

hipError_t launch_bucket_tiles_staged(hipStream_t st, TileLists& t, size_t ntiles, int tiles_x, uint32_t* total, uint2* entries, uint32_t hint) {
    // 512 threads: the same time alone as with 1024, 4 % more frames per second with the frame lanes overlapping (0.0955-0.0961 against 0.0992-0.1018 ms per
    // frame at C2, alternating runs) — an 8-wave workgroup finds room on a busy CU sooner than a 16-wave one.  A thread holds up to 16 (buckets of <= 8192
    // entries: what tile_lists_plan aims at) or 32 entries.
#define GS4D_BTS(K) k_bucket_tiles_staged<K, 512><<<dim3(t.nb), dim3(512), t.counters * 4u, st>>>(t.blocks, t.hist, t.hist + t.hist_cap, t.rows, t.scap, t.bcap, t.nb, (uint32_t)ntiles, t.slabs, t.slab_shift, \
                                                                                                 t.counters, t.tstart, t.tcnt, entries, t.bstat, total + TL_ABORT_WORD, t.seq, hint, (uint32_t)tiles_x, t.box)
    if (t.bcap <= 16u * 512u) GS4D_BTS(16);
    else if (t.bcap <= 32u * 512u) GS4D_BTS(32);
    else return hipErrorInvalidValue;                      // run_draw does not stage such a draw
#undef GS4D_BTS
    return hipGetLastError();
}

} // namespace gs4d
