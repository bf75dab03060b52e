// sort.hip — depth-key generation and the stable LSD radix sort (gfx950, wave64).
//
// Replaces, behind gs4d_keygen / gs4d_sort_pairs:
//   * the single-threaded CPU key loop + two uploads            4DSplatRendering/Scenes.h:28-36, 314-325
//   * radix_sort::sorter::sort and its three GLSL kernels       Dependencies/GPU_RADIX_SORT/radix_sort.hpp:258-392,
//                                                               resources/radix_sort_{count,local_offsets,reorder}.comp.glsl
// Contract kept: output == stable ascending sort by the uint32 key, payload follows (bit-exact permutation).
// Design (not a translation of the GLSL): 8-bit digits x 4 passes instead of 4-bit x 8; ONE histogram launch for all digits and
// ONE launch per pass (chained scan with decoupled look-back) instead of 2*log2(P2)+3 dispatches per pass; keys are ranked with
// wave64 ballot match + popcount (no 16 KB Blelloch tables, no per-pass re-sort of the block) and reordered in LDS so that each
// digit's run leaves the workgroup as consecutive addresses.
//
// Compiled with -ffp-contract=off: the key must be bit-identical to the CPU expression
// 1.0f / sqrtf(dx*dx + dy*dy + dz*dz) (IEEE-correct sqrt and divide are hipcc's default).
#include "gs4d_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <vector>

namespace gs4d {

// ------------------------------------------------------------------------------------------------
// keygen: reads 32 B/splat from the SoA planes (pos, sig[3]) instead of the 96-B record
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_keygen(const float4* __restrict__ pos, const float4* __restrict__ sig3, uint32_t n, float t,
                                                float camx, float camy, float camz, float4 vrow2 /* view row 2: V[2],V[6],V[10],V[14] */, int key_mode,
                                                float* __restrict__ keys, uint32_t* __restrict__ idx, uint32_t* __restrict__ ghist /* digit histograms of (key - bias), for the sort */, int rb /* ... in digits of rb bits */,
                                                uint32_t bias /* host-proven lower bound of every key's bit pattern */, uint32_t span /* ... and of (key - bias) from above */, uint32_t* __restrict__ err,
                                                float4 csig3, float cmut /* sig3 == nullptr (a static set: SOA_STATIC3D): sig[3] and mu_t of every record */) {
    __shared__ uint32_t h[OS_MAX_PASSES][OS_MAX_BINS];
    os_hist_clear(h, threadIdx.x);
    __syncthreads();
    for (uint32_t i0 = blockIdx.x * 256u; i0 < n; i0 += gridDim.x * 256u) {      // uniform trip count per workgroup
        const uint32_t i = i0 + threadIdx.x;
        const bool in = i < n;
        float key = 0.0f;
        if (in) {
            float4 p = pos[i];
            float4 s = csig3;
            if (sig3) s = sig3[i]; else p.w = cmut;        // uniform
            if (key_mode == GS4D_KEY_REF_INV_EUCLID) {
                float ct = t - p.w;                        // Scenes.h:30
                float x = p.x + s.x * ct;                  // :31-33  (sig[3].xyz, NOT divided by Sigma44)
                float y = p.y + s.y * ct;
                float z = p.z + s.z * ct;
                float dx = x - camx, dy = y - camy, dz = z - camz;          // :317
                key = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);            // :318
            } else {
                // extra mode: view-space depth of the shader's conditioned mean (Splat4DVertexShaderInstanced.GLSL:86)
                float k = (1.0f / s.w) * (t - p.w);
                float x = p.x + k * s.x, y = p.y + k * s.y, z = p.z + k * s.z;
                float zv = ((vrow2.x * x + vrow2.y * y) + vrow2.z * z) + vrow2.w;
                key = 1.0f / fmaxf(-zv, 1e-20f);
            }
            keys[i] = key;
            idx[i] = i;
        }
        const uint32_t kb = __float_as_uint(key);
        if (in && (kb < bias || kb - bias > span)) __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // the bound did not hold: reported, never silently mis-sorted
        os_hist_add(h, kb - bias, in, OS_MAX_PASSES, rb);
    }
    __syncthreads();
    os_hist_flush(h, ghist, OS_MAX_PASSES, threadIdx.x);
}

hipError_t launch_keygen(hipStream_t st, const float4* pos, const float4* sig3, const SoaInfo& info, size_t n, float t, const float cam[3], const float view[16], int key_mode, float* keys, uint32_t* idx, uint32_t* ghist, int rb,
                         uint32_t bias, uint32_t span, uint32_t* err) {
    if (n == 0) return hipSuccess;
    float4 vr = make_float4(view[2], view[6], view[10], view[14]);
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 1024);     // grid-stride: bounds the histogram flush to 1024 workgroups
    k_keygen<<<dim3(blocks), dim3(256), 0, st>>>(pos, sig3, (uint32_t)n, t, cam[0], cam[1], cam[2], vr, key_mode, keys, idx, ghist, rb, bias, span, err,
                                                     make_float4(info.consts[4], info.consts[5], info.consts[6], info.consts[7]), info.consts[0]);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// radix sort: one launch per 8-bit digit, chained scan with a two-level decoupled look-back
//
//   histogram  : global digit histograms of ALL passes, [4][256].  Produced by k_os_hist (keys read once, LDS-privatised) — or,
//                for free, by the kernel that wrote the keys (k_keygen for depth keys, k_bin_emit for tile ids), in which case no
//                histogram launch happens at all.
//   k_os_pass  : a workgroup ranks the keys of its tile (wave64 ballot match), publishes its per-digit counts, looks back over its
//                predecessors to obtain its exclusive prefix, reorders the tile in LDS and writes each digit's run to its final
//                place with consecutive lanes on consecutive addresses.  Traffic per pass: keys+values read once, written once.
//   A pass whose digit is the same for every key (e.g. the sign+exponent byte of positive depth keys) is a stable identity and is
//   skipped ON THE DEVICE: every workgroup derives from the histograms which passes are live and which of three buffers (caller's,
//   scratch B, scratch C) it reads and writes, so that the last live pass lands in the caller's buffers without a copy.
// Inter-workgroup hand-off: each look-back word is ONE naturally aligned 64-bit granule {epoch:30, flag:2, value:32}, written by one
// agent-scope relaxed atomic store and polled with agent-scope relaxed atomic loads (bypass L1, write-through) — the data-tagged
// granule form of cdna_hip_programming.md Guideline 16 (R2): no separate flag, hence no ordering requirement.  The epoch (one per
// launch) makes words of earlier launches read as "not published", so the words are never zeroed.
// Look-back is two-level (tiles in groups of OS_GROUP; the last tile of a group publishes the group's total): when all tiles of a
// small sort start together, a tile needs ~3 memory round trips instead of one per ~32 predecessors.
// Tile ids are tickets (an atomic counter, one per started workgroup): a tile waits only for tiles that are already running, whatever
// the dispatch order.  Every spin is bounded as well: on time-out the kernel raises `err` and leaves, and the host reports the
// frame as failed instead of hanging the GPU or returning wrong data.
// ------------------------------------------------------------------------------------------------
// Per-tile wall-clock stamps of one pass (tools/stamps.py) exist only in tuning builds (make TUNING=1): release kernels carry no
// ablation branches.
#ifdef GS4D_TUNING
#define OS_STAMP(k) do { if (stamps && tid == 0) stamps[tile * 8 + (k)] = wall_clock64(); } while (0)
#else
#define OS_STAMP(k) do { } while (0)
#endif
constexpr uint32_t OS_GROUP = 16;         // tiles per look-back group (every look-back word in flight at once is a live register: 32 x 32 held the kernel at 177 VGPRs = one workgroup per CU)
constexpr uint32_t OS_SUPER = 16;         // groups per super-group
typedef unsigned long long u64;

struct OsBufs { uint32_t* k[3]; uint32_t* v[3]; };      // [0] caller's buffers, [1],[2] scratch

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(v, off, 64); if (lane >= (unsigned)off) v += t; }
    return v;
}
__global__ __launch_bounds__(256) void k_os_hist(const uint32_t* __restrict__ keys, uint32_t n_cap, const uint32_t* __restrict__ n_dev, int passes, int rb,
                                                 uint32_t* __restrict__ ghist /* [OS_REPL][4][OS_MAX_BINS], zero on entry */) {
    __shared__ uint32_t h[OS_MAX_PASSES][OS_MAX_BINS];
    const uint32_t n = n_dev ? min(*n_dev, n_cap) : n_cap;
    const uint32_t tid = threadIdx.x;
    os_hist_clear(h, tid);
    __syncthreads();
    const uint32_t nvec = n / 4u;
    const uint4* k4 = reinterpret_cast<const uint4*>(keys);
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t i0 = blockIdx.x * 256u + tid; i0 < nvec; i0 += 4u * stride) {
        uint4 kk[4];
        bool in[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const uint32_t i = i0 + u * stride; in[u] = i < nvec; kk[u] = in[u] ? k4[i] : make_uint4(0, 0, 0, 0); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            os_hist_add(h, kk[u].x, in[u], passes, rb); os_hist_add(h, kk[u].y, in[u], passes, rb);
            os_hist_add(h, kk[u].z, in[u], passes, rb); os_hist_add(h, kk[u].w, in[u], passes, rb);
        }
    }
    if (blockIdx.x == 0) { const bool in = tid < (n & 3u); os_hist_add(h, in ? keys[nvec * 4u + tid] : 0u, in, passes, rb); }      // tail keys
    __syncthreads();
    os_hist_flush(h, ghist, passes, tid);
}

// Which passes are live, and which buffers pass `p` reads and writes.  Returns false when pass p is skipped.
// A single live pass would end in a scratch buffer: one identity pass is then run as well (a stable copy), so the number of
// executed passes is 0, 2, 3 or 4 and the result always lands in buffer 0.
// `live` is a register copy of the workgroup's live mask (bit q: pass q moves keys): every thread derives the same schedule from
// the same word — nothing is amended in shared memory, so waves cannot disagree about it.
// `executed` = how many passes move keys before pass p (0: pass p is the first to read the caller's buffers), `total` = how many do at all.
__device__ __forceinline__ bool os_schedule(uint32_t live, int passes, int p, int& src, int& dst, int& executed, int& total) {
    const uint32_t all = (1u << passes) - 1u;
    live &= all;
    int k = __popc(live);
    if (k == 1) { const uint32_t deadm = all & ~live; live |= deadm & (0u - deadm); ++k; }      // revive the first dead pass (passes >= 2)
    total = k;
    const int j = __popc(live & ((1u << p) - 1u));
    executed = j;
    if (!((live >> p) & 1u)) return false;
    // buffer after i executed passes: even k: 0,1,0,1,...   odd k (>= 3): 0,1,2,0,1,0,...
    auto buf_at = [k](int i) { if (k & 1) { if (i <= 2) return i; return (i - 3) & 1; } return i & 1; };
    src = buf_at(j);
    dst = buf_at(j + 1);
    return true;
}

// ---- look-back words ----
// tile level : 32-bit {epoch:18, count:14}: the tile's count of one digit (<= TILE_KEYS < 2^14).  Published once per launch.
// group level: 64-bit {epoch:30, flag:2, value:32}: AGG = total of the group's tiles, INCL = inclusive global prefix at its end.
// Small words matter: with every tile of a 10^6-key sort in flight at once, look-back reads are (tiles^2/group) KB of uncached
// traffic per pass — as many bytes as the keys themselves if the words are wide or the tiles small.
__device__ __forceinline__ uint32_t os_tword(uint32_t epoch, uint32_t count) { return ((epoch & 0x3FFFFu) << 14) | count; }
__device__ __forceinline__ bool os_tpublished(uint32_t w, uint32_t epoch) { return (w >> 14) == (epoch & 0x3FFFFu); }

// ---- look-back reads ----
// A round of agent-scope loads costs 1.5-3 us on a busy device, so the look-back is built to need ONE: the words of the group's earlier
// tiles and the accumulators of the super-group's earlier groups are all requested before any of them is examined.
// Polling discipline when something is still missing: spin on the FIRST missing word alone (one load per thread and round, with a
// sleep), then read the rest again.  Re-reading everything every round is a polling storm — a few hundred tiles asking for tens of
// TB/s — and the stores everybody is waiting for queue up behind it.

// rows hi-k0 .. hi-(rows-1) of the tile words are outstanding and row hi-k0 is known to be missing
template <int LB, int BINS>
__device__ __forceinline__ uint32_t os_tiles_finish(const uint32_t* st, int32_t hi, int rows, int k0, uint32_t sum, uint32_t tid, uint32_t epoch, uint32_t* err) {
    uint32_t spins = 0;
    while (true) {
        while (true) {
            __builtin_amdgcn_s_sleep(8);
            const uint32_t w = __hip_atomic_load(st + (size_t)(hi - k0) * (uint32_t)BINS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (os_tpublished(w, epoch)) { sum += w & 0x3FFFu; break; }
            if (++spins > (1u << 20)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return sum; }
        }
        if (++k0 >= rows) return sum;
        uint32_t sv[LB];
#pragma unroll
        for (int k = 0; k < LB; ++k) sv[k] = (k >= k0 && k < rows) ? __hip_atomic_load(st + (size_t)(hi - k) * (uint32_t)BINS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        bool missing = false;
#pragma unroll
        for (int k = 0; k < LB; ++k) {
            if (missing || k < k0 || k >= rows) continue;
            if (!os_tpublished(sv[k], epoch)) { k0 = k; missing = true; continue; }
            sum += sv[k] & 0x3FFFu;
        }
        if (!missing) return sum;
    }
}

// accumulator rows t+k0 .. t+rows-1 are outstanding and row t+k0 is known to be incomplete (complete = `expect` arrivals)
template <int LB, int BINS>
__device__ __forceinline__ uint32_t os_acc_finish(const uint32_t* acc, uint32_t t, int rows, int k0, uint32_t sum, uint32_t expect, uint32_t tid, uint32_t* err) {
    uint32_t spins = 0;
    while (true) {
        while (true) {
            __builtin_amdgcn_s_sleep(8);
            const uint32_t w = __hip_atomic_load(acc + (size_t)(t + k0) * (uint32_t)BINS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((w >> 24) == expect) { sum += w & 0xFFFFFFu; break; }
            if (++spins > (1u << 20)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return sum; }
        }
        if (++k0 >= rows) return sum;
        uint32_t sv[LB];
#pragma unroll
        for (int k = 0; k < LB; ++k) sv[k] = (k >= k0 && k < rows) ? __hip_atomic_load(acc + (size_t)(t + k) * (uint32_t)BINS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        bool missing = false;
#pragma unroll
        for (int k = 0; k < LB; ++k) {
            if (missing || k < k0 || k >= rows) continue;
            if ((sv[k] >> 24) != expect) { k0 = k; missing = true; continue; }
            sum += sv[k] & 0xFFFFFFu;
        }
        if (!missing) return sum;
    }
}

// Exclusive prefix of tile `tile` for digit `tid`: earlier tiles of its group + earlier groups of its super-group + earlier super-groups.
template <int BINS>
__device__ __forceinline__ uint32_t os_lookback(const uint32_t* st, const uint32_t* acc, uint32_t acc_groups, uint32_t tile, uint32_t tid, uint32_t epoch, uint32_t* err) {
    const uint32_t grp = tile / OS_GROUP, sup = grp / OS_SUPER;
    const int rows_t = (int)(tile - grp * OS_GROUP), rows_g = (int)(grp - sup * OS_SUPER);
    const int32_t hi = (int32_t)tile - 1;
    const uint32_t g0 = sup * OS_SUPER;
    uint32_t tv[OS_GROUP], gv[OS_SUPER];
#pragma unroll
    for (int k = 0; k < (int)OS_GROUP; ++k) tv[k] = k < rows_t ? __hip_atomic_load(st + (size_t)(hi - k) * (uint32_t)BINS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
    for (int k = 0; k < (int)OS_SUPER; ++k) gv[k] = k < rows_g ? __hip_atomic_load(acc + (size_t)(g0 + k) * (uint32_t)BINS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    uint32_t sum_t = 0, sum_g = 0, sum_s = 0;
    int miss_t = -1, miss_g = -1;
#pragma unroll
    for (int k = 0; k < (int)OS_GROUP; ++k) {
        if (miss_t >= 0 || k >= rows_t) continue;
        if (!os_tpublished(tv[k], epoch)) { miss_t = k; continue; }
        sum_t += tv[k] & 0x3FFFu;
    }
#pragma unroll
    for (int k = 0; k < (int)OS_SUPER; ++k) {
        if (miss_g >= 0 || k >= rows_g) continue;
        if ((gv[k] >> 24) != OS_GROUP) { miss_g = k; continue; }
        sum_g += gv[k] & 0xFFFFFFu;
    }
    // earlier super-groups (sorts of more than 1024 tiles only), 16 at a time
    const uint32_t* sacc = acc + (size_t)acc_groups * (uint32_t)BINS;
    for (uint32_t t = 0; t < sup; t += 16u) {
        const int rows = (int)min(16u, sup - t);
        uint32_t sv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) sv[k] = k < rows ? __hip_atomic_load(sacc + (size_t)(t + k) * (uint32_t)BINS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        int miss = -1;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (miss >= 0 || k >= rows) continue;
            if ((sv[k] >> 24) != OS_SUPER) { miss = k; continue; }
            sum_s += sv[k] & 0xFFFFFFu;
        }
        if (miss >= 0) sum_s = os_acc_finish<16, BINS>(sacc, t, rows, miss, sum_s, OS_SUPER, tid, err);
    }
    if (miss_t >= 0) sum_t = os_tiles_finish<(int)OS_GROUP, BINS>(st, hi, rows_t, miss_t, sum_t, tid, epoch, err);
    if (miss_g >= 0) sum_g = os_acc_finish<(int)OS_SUPER, BINS>(acc, g0, rows_g, miss_g, sum_g, OS_GROUP, tid, err);
    return sum_t + sum_g + sum_s;
}

// exclusive scan of one value per digit (threads 0..BINS-1 carry a value, all THREADS threads take part in the barriers)
template <int THREADS, int BINS>
__device__ __forceinline__ uint32_t digit_excl_scan(uint32_t v, uint32_t* tmp /* __shared__[BINS / 64] */, uint32_t tid) {
    const uint32_t lane = tid & 63u, w = tid >> 6;
    const uint32_t inc = wave_incl_scan_u32(tid < (uint32_t)BINS ? v : 0u, lane);
    __syncthreads();
    if (lane == 63u && w < (uint32_t)(BINS / 64)) tmp[w] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int k = 0; k < BINS / 64 - 1; ++k) if ((unsigned)k < w) base += tmp[k];
    return base + inc - v;
}

// Persistent workgroups: the grid is what fits the device at once (or one workgroup per tile if that is fewer); a workgroup draws a
// ticket, sorts that tile, draws the next.  What does not depend on the tile (histograms -> live passes and digit bases, the
// housekeeping for the next launch) happens once per workgroup, and a finished tile's successor starts without a dispatch.
template <int THREADS, int ITEMS, bool ATOMIC_RANK, int RB>
__global__ __launch_bounds__(THREADS) void k_os_pass(OsBufs bufs, uint32_t n_cap, const uint32_t* __restrict__ n_dev, int pass, int passes,
                                                     const uint32_t* __restrict__ ghist /* [OS_REPL][4][OS_MAX_BINS] */, uint32_t* __restrict__ ghist_other /* zeroed by pass 0 */,
                                                     uint32_t* status /* [tiles][BINS] */, uint32_t* acc /* [acc_groups + supers][BINS], zero at launch */, uint32_t* acc_next /* zeroed here */, uint32_t acc_groups, uint32_t acc_words,
                                                     uint32_t epoch, uint32_t* err,
                                                     uint32_t* ticket /* zero at launch */, uint32_t* ticket_next /* zeroed here */, uint32_t bias, int identity_vals /* the payload is the identity index and has NOT been written: see radix_sort_pairs */,
                                                     u64* stamps /* tuning aid, may be null */) {
    constexpr uint32_t TILE_KEYS = THREADS * ITEMS;
    constexpr int WAVES = THREADS / 64;
    constexpr uint32_t BINS = 1u << RB;                      // digits of RB bits: thread d < BINS owns digit d in everything per digit below
    static_assert(BINS <= (uint32_t)THREADS && BINS <= OS_MAX_BINS && BINS >= 64u, "one thread per digit");
    static_assert(TILE_KEYS < (1u << 14), "tile-level look-back words carry 14-bit counts");
    static_assert((uint64_t)TILE_KEYS * OS_GROUP * OS_SUPER < (1u << 24) && OS_GROUP < 256u && OS_SUPER < 256u, "accumulators are {arrivals:8, sum:24}");
    __shared__ uint32_t skeys[TILE_KEYS];
    __shared__ uint32_t svals[TILE_KEYS];
    __shared__ uint32_t wcnt[WAVES][BINS];
    __shared__ uint32_t loff[BINS];     // first local slot of digit d in the reordered tile
    __shared__ uint32_t gpos[BINS];     // global slot of that first element
    __shared__ uint32_t s_tmp[BINS / 64];
    __shared__ uint32_t s_dead;          // bit q: one digit of pass q holds every key (the pass is a stable identity)
    __shared__ uint32_t s_tile;

    const uint32_t n = n_dev ? min(*n_dev, n_cap) : n_cap;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    // Tile ids are handed out by ticket, in the order they are STARTED: a tile only ever waits for tiles that are already running.
    // (With blockIdx as the tile id, two chained-scan kernels running side by side on different streams can dead-lock each other:
    // workgroups are dispatched per XCD, so each kernel's late tiles can fill the XCD the other kernel's early tiles need.)
    // The ticket's round trip overlaps the histogram loads below, which do not depend on the tile.
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    if (tid == 1u) s_dead = 0u;
    if (tid == 2u && blockIdx.x == 0) __hip_atomic_store(ticket_next, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the next launch on this stream counts from zero again
    for (uint32_t q = tid; q < WAVES * BINS; q += THREADS) (&wcnt[0][0])[q] = 0u;
    uint32_t tot = 0, dead = 0;
    if (tid < BINS) {
        for (int q = 0; q < passes; ++q) {
            uint32_t g = 0;
#pragma unroll
            for (int r = 0; r < OS_REPL; ++r) g += ghist[(r * OS_MAX_PASSES + q) * OS_MAX_BINS + tid];
            if (q == pass) tot = g;
            if (g == n) dead |= 1u << q;                          // one digit holds every key
        }
    }
    __syncthreads();
    if (dead) atomicOr(&s_dead, dead);                            // at most one thread per dead pass
    const uint32_t ntiles = (n + TILE_KEYS - 1u) / TILE_KEYS;
    // housekeeping for the NEXT launch of this sorter: its group accumulators (every workgroup a slice), its histogram slot
    for (uint32_t q = blockIdx.x * THREADS + tid; q < acc_words; q += gridDim.x * THREADS) acc_next[q] = 0u;
    if (pass == 0 && blockIdx.x == 0) { for (uint32_t q = tid; q < OS_SLOT_WORDS; q += THREADS) ghist_other[q] = 0u; }
    __syncthreads();
    uint32_t tile = s_tile;
    if (tile >= ntiles) return;                                   // uniform
    const int shift = RB * pass;
    int src, dst, executed, total;
    if (!os_schedule(~s_dead, passes, pass, src, dst, executed, total)) {          // uniform: this pass is an identity
        // an identity payload nobody has written, and no pass at all will move anything (every key is the same): pass 0's launch writes it
        if (identity_vals && pass == 0 && total == 0) { for (uint32_t i = blockIdx.x * THREADS + tid; i < n; i += gridDim.x * THREADS) bufs.v[0][i] = i; }
        return;
    }
    const bool make_identity = identity_vals && executed == 0;   // uniform: the first pass that moves keys makes up the payload instead of reading it
    const uint32_t* __restrict__ keys_in = bufs.k[src]; const uint32_t* __restrict__ vals_in = bufs.v[src];
    uint32_t* __restrict__ keys_out = bufs.k[dst]; uint32_t* __restrict__ vals_out = bufs.v[dst];
    uint32_t digit_base = 0;
    bool first = true;
    const uint32_t ngroups = (ntiles + OS_GROUP - 1u) / OS_GROUP, nsuper = (ngroups + OS_SUPER - 1u) / OS_SUPER;

    while (true) {                                                // uniform: `tile` is the same in every thread
        OS_STAMP(0);
        const uint32_t tbase = tile * TILE_KEYS;
        const uint32_t wbase = tbase + w * (64u * ITEMS);
        uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint32_t i = wbase + j * 64u + lane;
            const bool valid = i < n;
            key[j] = valid ? keys_in[i] : 0xFFFFFFFFu;
            val[j] = !valid ? 0u : make_identity ? i : vals_in[i];
        }
        if (first) { digit_base = digit_excl_scan<THREADS, (int)BINS>(tot, s_tmp, tid); first = false; }      // once per workgroup, under the first tile's loads
        OS_STAMP(1);

        if (ATOMIC_RANK) {
            // rank = value returned by an LDS atomic add on the wave's digit counter.  Stable only because the LDS unit serialises the
            // lanes of one instruction that hit the same counter in ascending lane order — not an architectural promise, so the
            // library verifies it on the device at context creation (k_lds_order_test) and uses the ballot form below otherwise.
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const bool valid = (wbase + j * 64u + lane) < n;
                const uint32_t d = ((key[j] - bias) >> shift) & (BINS - 1u);
                rank[j] = valid ? atomicAdd(&wcnt[w][d], 1u) : 0u;
            }
        } else {
            const uint64_t lt = (1ull << lane) - 1ull;
            volatile uint32_t* wc = wcnt[w];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const bool valid = (wbase + j * 64u + lane) < n;
                const uint32_t d = ((key[j] - bias) >> shift) & (BINS - 1u);
                uint64_t m = __ballot(valid);
#pragma unroll
                for (int b = 0; b < RB; ++b) {
                    const bool bit = (d >> b) & 1u;
                    const uint64_t bal = __ballot(bit);
                    m &= bit ? bal : ~bal;
                }
                rank[j] = 0;
                if (valid) {
                    const uint32_t c = wc[d];
                    rank[j] = c + (uint32_t)__popcll(m & lt);
                    __builtin_amdgcn_wave_barrier();
                    if ((m & lt) == 0) wc[d] = c + (uint32_t)__popcll(m);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        // thread d: counts per wave -> offsets inside the tile's digit-d run; tile count of digit d; publish it at once
        uint32_t cnt = 0;
        uint32_t garr = 0;
        const uint32_t grp = tile / OS_GROUP, sup = grp / OS_SUPER;
        if (tid < BINS) {
#pragma unroll
            for (int k = 0; k < WAVES; ++k) { const uint32_t t = wcnt[k][tid]; wcnt[k][tid] = cnt; cnt += t; }
            __hip_atomic_store(status + (size_t)tile * BINS + tid, os_tword(epoch, cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // {arrivals:8, sum:24} accumulator of the group (only if a later tile will read it); what it held before comes back after the
            // reorder below
            // (asked for only where a super-group total will be needed: waiting for the returned value costs a memory round trip)
            if (sup + 1u < nsuper) garr = __hip_atomic_fetch_add(acc + (size_t)grp * BINS + tid, (1u << 24) | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (grp + 1u < ngroups) (void)__hip_atomic_fetch_add(acc + (size_t)grp * BINS + tid, (1u << 24) | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        OS_STAMP(2);
        // local run starts, then the reorder inside LDS — none of it needs the other tiles, so it overlaps their publishing
        const uint32_t lo_ = digit_excl_scan<THREADS, (int)BINS>(cnt, s_tmp, tid);
        if (tid < BINS) loff[tid] = lo_;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            if ((wbase + j * 64u + lane) < n) {                   // stable: wave-major, item-major, lane order == memory order
                const uint32_t d = ((key[j] - bias) >> shift) & (BINS - 1u);
                const uint32_t l = loff[d] + wcnt[w][d] + rank[j];
                skeys[l] = key[j];
                svals[l] = val[j];
            }
        }
        OS_STAMP(3);
        if (tid < BINS) {
            // the tile that completes its group (per digit: whichever arrived sixteenth) hands the group's total to the super-group: a
            // super-group's accumulator takes OS_SUPER additions per digit, not one from every tile under it
            if (sup + 1u < nsuper && (garr >> 24) == OS_GROUP - 1u)
                (void)__hip_atomic_fetch_add(acc + (size_t)(acc_groups + sup) * BINS + tid, (1u << 24) | ((garr & 0xFFFFFFu) + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // look back, one memory round trip: the tiles of this group before this one (their words), the groups of this super-group
            // before this group and the super-groups before this one (their accumulators, complete when every member has arrived)
            const uint32_t prefix = os_lookback<(int)BINS>(status, acc, acc_groups, tile, tid, epoch, err);
            gpos[tid] = digit_base + prefix;
        }
        __syncthreads();
        OS_STAMP(4);
        const uint32_t tcount = min(TILE_KEYS, n - tbase);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint32_t l = j * THREADS + tid;
            if (l < tcount) {
                const uint32_t k = skeys[l];
                const uint32_t d = ((k - bias) >> shift) & (BINS - 1u);
                const uint32_t o = gpos[d] + (l - loff[d]);
                // Small sorts (one round of tiles): streaming stores, the runs go to memory as they are written instead of sitting dirty in
                // this XCD's L2 until the end-of-kernel write-back.  Large sorts (the 8192-key shape, several rounds of tiles per workgroup):
                // ordinary stores, so that the L2 merges the partial lines at the ends of neighbouring runs before they reach HBM — measured
                // at 10^7 keys: 75.7 -> 57.7 us per pass; at 10^6 keys the streaming form is the faster one (14.1 against 15.8).
                // `o` is built from the global digit histograms, which an EARLIER kernel accumulated (k_keygen, the projection, k_os_hist), and
                // from this launch's own counts: if the keys changed in between — a caller refilling the buffer through a device pointer
                // without gs4d_buffer_invalidate — the two no longer describe the same array and `o` can point anywhere.  One compare keeps
                // the store inside the buffer and turns the contract violation into the error word (the frame is reported as failed).
                if (o >= n) { __hip_atomic_store(err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); continue; }
                if (ITEMS >= 12) { keys_out[o] = k; vals_out[o] = svals[l]; }
                else { __builtin_nontemporal_store(k, keys_out + o); __builtin_nontemporal_store(svals[l], vals_out + o); }
            }
        }
        OS_STAMP(5);
        // The next ticket is drawn only now, behind the write-out: tiles with later tickets wait for the tile it names to publish, so a
        // ticket must not be held while its holder is still busy with something else (drawn before the write-out, 8 us at 10^7 keys, every
        // successor's look-back waited that long).  The other workgroup of the CU covers the round trip.
        if (gridDim.x >= ntiles) return;                          // uniform: every tile has a workgroup of its own, nothing is left to draw
        uint32_t next = 0;
        if (tid == 0) next = atomicAdd(ticket, 1u);
        __syncthreads();                                          // everybody is done with the tile's LDS
        if (tid == 0) s_tile = next;
        for (uint32_t q = tid; q < WAVES * BINS; q += THREADS) (&wcnt[0][0])[q] = 0u;
        __syncthreads();
        tile = s_tile;
        if (tile >= ntiles) return;                               // uniform; every workgroup ends on a ticket past the last tile
    }
}

// Does an LDS atomic add executed by a whole wave return, to the lanes that hit the same address, their rank in ascending lane order?
__global__ __launch_bounds__(256) void k_lds_order_test(uint32_t* __restrict__ bad) {
    __shared__ uint32_t cnt[4][256];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    uint32_t errors = 0;
    for (uint32_t round = 0; round < 96u; ++round) {
#pragma unroll
        for (int k = 0; k < 4; ++k) cnt[k][tid] = 0u;
        __syncthreads();
        const uint32_t bins = round % 6u == 0u ? 1u : round % 6u == 1u ? 2u : round % 6u == 2u ? 3u : round % 6u == 3u ? 7u : round % 6u == 4u ? 64u : 256u;
        uint32_t x = (blockIdx.x * 256u + tid) * 2654435761u + round * 40503u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        const uint32_t d = x % bins;
        const bool active = ((x >> 20) & 7u) != 0u;              // some lanes sit out, as at the end of an array
        uint64_t m = __ballot(active);
#pragma unroll
        for (int b = 0; b < 8; ++b) { const bool bit = (d >> b) & 1u; const uint64_t bal = __ballot(bit); m &= bit ? bal : ~bal; }
        if (active) {
            const uint32_t got = atomicAdd(&cnt[w][d], 1u);
            const uint32_t want = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (got != want) ++errors;
            const uint32_t got2 = atomicAdd(&cnt[w][d], 1u);     // a second instruction continues after all lanes of the first
            if (got2 != (uint32_t)__popcll(m) + want) ++errors;
        }
        __syncthreads();
    }
    if (errors) atomicAdd(bad, errors);
}

// The test runs on every stream given, at the same time (the frame lanes' streams sit on hardware queues of their own): the workgroups of the
// launches share the CUs, so each wave is ranked beside waves of another kernel instance — the situation the product's kernels are in.
hipError_t lds_atomic_order_selftest(const hipStream_t* streams, int nstreams, bool* ordered) {
    *ordered = false;
    if (nstreams < 1) return hipErrorInvalidValue;
    if (nstreams > 8) nstreams = 8;
    uint32_t* d = nullptr; uint32_t h[8] = { 1, 1, 1, 1, 1, 1, 1, 1 };
    hipError_t e = hipMalloc(&d, 4 * (size_t)nstreams);
    if (e != hipSuccess) return e;
    for (int i = 0; i < nstreams && e == hipSuccess; ++i) e = hipMemsetAsync(d + i, 0, 4, streams[i]);
    for (int i = 0; i < nstreams && e == hipSuccess; ++i) { k_lds_order_test<<<dim3(512), dim3(256), 0, streams[i]>>>(d + i); e = hipGetLastError(); }
    for (int i = 0; i < nstreams && e == hipSuccess; ++i) e = hipMemcpyAsync(&h[i], d + i, 4, hipMemcpyDeviceToHost, streams[i]);
    for (int i = 0; i < nstreams; ++i) { const hipError_t e2 = hipStreamSynchronize(streams[i]); if (e == hipSuccess) e = e2; }
    (void)hipFree(d);
    bool ok = e == hipSuccess;
    for (int i = 0; i < nstreams; ++i) ok = ok && h[i] == 0;
    *ordered = ok;
    return e;
}

hipError_t sort_scratch_reserve(hipStream_t st, SortScratch& s, size_t n) {
    hipError_t e;
    if (s.cap < n) {
        if (s.keys2) { (void)hipStreamSynchronize(st); (void)hipFree(s.keys2); }
        s.keys2 = s.vals2 = nullptr; s.cap = 0;
        if ((e = hipMalloc(&s.keys2, n * 16)) != hipSuccess) return e;        // scratch B and C, keys and values
        s.vals2 = s.keys2 + 2 * n;
        s.cap = n;
    }
    // control block: two [OS_REPL][4][OS_MAX_BINS] histogram slots (alternating), then the look-back words [tiles + groups][bins], sized for the widest digit
    const size_t tiles = (s.cap + 1023) / 1024;      // smallest tile = 1024 keys; sized for the largest sort seen (the layout depends on it)
    const size_t groups = tiles / OS_GROUP + 2, supers = groups / OS_SUPER + 2;
    const size_t words = 2 * OS_SLOT_WORDS + (tiles + 2 + 2 * 2 * (groups + supers)) * OS_MAX_BINS;       // tile words, two accumulator sets
    if (s.hist_cap < words) {
        uint32_t* nh = nullptr;
        if ((e = hipMalloc(&nh, words * 4)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(nh, 0, words * 4, st)) != hipSuccess) return e;
        if (s.hist) {
            // keep a histogram that a producer kernel has already accumulated into the current slot
            if ((e = hipMemcpyAsync(nh, s.hist, 2 * OS_SLOT_WORDS * 4, hipMemcpyDeviceToDevice, st)) != hipSuccess) return e;
            (void)hipStreamSynchronize(st); (void)hipFree(s.hist);
        }
        s.hist = nh; s.hist_cap = words;
    }
    if (!s.totals) {
        // test hook: start the epoch counter close to its 18-bit wrap so that a short test crosses it
        if (const char* e0 = getenv("GS4D_TEST_EPOCH0")) s.epoch = (uint32_t)strtoul(e0, nullptr, 0);
    }
    if (!s.totals) { if ((e = hipMalloc(&s.totals, 256 * 4)) != hipSuccess) return e; if ((e = hipMemsetAsync(s.totals, 0, 1024, st)) != hipSuccess) return e; }
    return hipSuccess;
}

void sort_scratch_free(SortScratch& s) {
    if (s.keys2) (void)hipFree(s.keys2);
    if (s.hist) (void)hipFree(s.hist);
    if (s.totals) (void)hipFree(s.totals);
    s = SortScratch();
}

uint32_t* sort_hist_slot(hipStream_t st, SortScratch& s, size_t n_hint, hipError_t* e_out) {
    // The slot a producer kernel (k_keygen, k_bin_emit) accumulates the digit histograms of the NEXT sort into.  It is zero unless an
    // earlier producer's histogram was never consumed by a sort; then it is cleared first.
    hipError_t e = sort_scratch_reserve(st, s, n_hint);
    if (e == hipSuccess && s.hist_pending) e = hipMemsetAsync(s.hist + (s.flip ? OS_SLOT_WORDS : 0), 0, OS_SLOT_WORDS * 4, st);
    if (e_out) *e_out = e;
    if (e != hipSuccess) return nullptr;
    s.hist_pending = true;
    return s.hist + (s.flip ? OS_SLOT_WORDS : 0);
}

template <int THREADS, int ITEMS, bool ATOMIC_RANK, int RB>
static hipError_t onesweep(hipStream_t st, SortScratch& s, uint32_t* keys, uint32_t* vals, size_t n, const uint32_t* n_dev, int passes, bool have_hist, bool identity_vals) {
    const uint32_t tile_keys = THREADS * ITEMS;
    const uint32_t tiles = (uint32_t)((n + tile_keys - 1) / tile_keys);
    // persistent workgroups: as many as the device holds at once (asked of the runtime once per scratch and kernel instance: the scratch's context has
    // made its device current)
    uint32_t resident = 0;
    {
        const void* const fn = reinterpret_cast<const void*>(&k_os_pass<THREADS, ITEMS, ATOMIC_RANK, RB>);
        SortScratch::Resident* slot = nullptr;
        for (auto& r : s.resident) { if (r.fn == fn) { resident = r.groups; break; } if (!r.fn && !slot) slot = &r; }
        if (!resident) {
            int per_cu = 0, dev = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_os_pass<THREADS, ITEMS, ATOMIC_RANK, RB>, THREADS, 0) != hipSuccess || per_cu < 1) per_cu = 1;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
            resident = (uint32_t)per_cu * (uint32_t)cus;
            if (slot) { slot->fn = fn; slot->groups = resident; }
        }
    }
    const uint32_t grid = std::min(tiles, resident);
    uint32_t* ghist = s.hist + (s.flip ? OS_SLOT_WORDS : 0);
    uint32_t* ghist_other = s.hist + (s.flip ? 0 : OS_SLOT_WORDS);
    hipError_t e;
    const uint32_t bias = have_hist ? s.hist_bias : 0u;
    s.hist_bias = 0;
    if (!have_hist) {
        if (s.hist_pending) { if ((e = hipMemsetAsync(ghist, 0, OS_SLOT_WORDS * 4, st)) != hipSuccess) return e; }   // someone else's histogram sits in the slot
        const uint32_t hist_blocks = (uint32_t)std::min<size_t>((n / 16 + 255) / 256 + 1, 256);          // few workgroups: each flushes 256 global atomics per pass
        k_os_hist<<<dim3(hist_blocks), dim3(256), 0, st>>>(keys, (uint32_t)n, n_dev, passes, RB, ghist);
    }
    s.hist_pending = false;
    s.flip ^= 1;
#ifdef GS4D_TUNING
    static const char* stampf = getenv("GS4D_SORT_STAMP_FILE");
    static const int stamp_pass = getenv("GS4D_SORT_STAMP_PASS") ? atoi(getenv("GS4D_SORT_STAMP_PASS")) : 0;
#else
    const char* stampf = nullptr; const int stamp_pass = 0;
#endif
    u64* stamps = nullptr;
    if (stampf && hipMalloc(&stamps, (size_t)tiles * 64) != hipSuccess) stamps = nullptr;
    uint32_t* status = s.hist + 2 * OS_SLOT_WORDS;
    // accumulator sets sit behind the tile words of the LARGEST sort this scratch was sized for (so they never move between launches)
    const size_t cap_tiles = (s.cap + 1023) / 1024, cap_groups = cap_tiles / OS_GROUP + 2, cap_supers = cap_groups / OS_SUPER + 2;
    const size_t acc_words = (cap_groups + cap_supers) * OS_MAX_BINS;
    uint32_t* acc_base = status + (cap_tiles + 2) * OS_MAX_BINS;
    OsBufs b;
    b.k[0] = keys; b.v[0] = vals;
    b.k[1] = s.keys2; b.v[1] = s.vals2;
    b.k[2] = s.keys2 + s.cap; b.v[2] = s.vals2 + s.cap;
    for (int p = 0; p < passes; ++p) {
        ++s.epoch;
        if ((s.epoch & 0x3FFFFu) == 0u) {    // the 18-bit tile-level epoch wraps: forget every old word
            if ((e = hipMemsetAsync(status, 0, (s.hist_cap - 2 * OS_SLOT_WORDS) * 4, st)) != hipSuccess) return e;
            ++s.epoch;
        }
        k_os_pass<THREADS, ITEMS, ATOMIC_RANK, RB><<<dim3(grid), dim3(THREADS), 0, st>>>(b, (uint32_t)n, n_dev, p, passes, ghist, ghist_other, status, acc_base + (s.acc_flip ? acc_words : 0), acc_base + (s.acc_flip ? 0 : acc_words), (uint32_t)cap_groups, (uint32_t)acc_words,
                                                                         s.epoch & 0x3FFFFFFFu,
                                                                         s.err ? s.err : s.totals, s.totals + 64 + (s.acc_flip ? 1 : 0), s.totals + 64 + (s.acc_flip ? 0 : 1), bias, identity_vals ? 1 : 0, (stampf && p == stamp_pass) ? stamps : nullptr);
        s.acc_flip ^= 1;
    }
    if (stampf && stamps) {   // tuning aid: dump per-tile wall-clock stamps (100 MHz) of one pass
        (void)hipStreamSynchronize(st);
        std::vector<u64> hs((size_t)tiles * 8);
        (void)hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
        FILE* f = fopen(stampf, "w");
        if (f) { for (uint32_t t = 0; t < tiles; ++t) { for (int k = 0; k < 6; ++k) fprintf(f, "%llu ", hs[t * 8 + k] - hs[0]); fprintf(f, "\n"); } fclose(f); }
        (void)hipFree(stamps);
    }
    return hipGetLastError();
}

// Digit width of a sort of `key_bits`-bit keys: 9-bit digits (512 bins) where they save a pass over 8-bit ones — 17-18 and 25-27 key bits, e.g. the
// depth keys of slowly moving (4D) splats, whose host-proven span is a little wider than 2^24; the tile shape needs a thread per bin.
int sort_plan_rb(const SortScratch& s, size_t n, int key_bits) {
    (void)n;
    if (s.rb_knob == 8 || s.rb_knob == 9) return (s.rb_knob == 9 && (s.shape_knob == 1 || s.shape_knob == 4)) ? 8 : s.rb_knob;      // test hook (the 256-thread shapes have no 512-bin form)
    if (s.shape_knob == 1 || s.shape_knob == 4) return 8;
    const int p8 = std::max(2, (key_bits + 7) / 8), p9 = std::max(2, (key_bits + 8) / 9);
    // (Beyond 27 bits both take four launches.  Preferring 9-bit digits there — their last digit is bits 27-31 only and is skipped on the device when
    // no key reaches 2^27 above the bias — was measured on the 4D sweep of BASELINE.json configs[3], whose bound is open-ended: 39.36 against 39.26 ms, nothing.)
    return p9 < p8 ? 9 : 8;
}
int sort_plan_passes(int key_bits, int rb) { return std::min(OS_MAX_PASSES, std::max(2, (key_bits + rb - 1) / rb)); }      // an even number of executed passes always exists (see os_schedule)

hipError_t radix_sort_pairs(hipStream_t st, SortScratch& s, uint32_t* keys, uint32_t* vals, size_t n, const uint32_t* n_dev, int key_bits, bool have_hist, bool identity_vals) {
    if (n <= 1) { if (n == 1 && identity_vals) return hipMemsetAsync(vals, 0, 4, st); return hipSuccess; }      // radix_sort.hpp:260
    if (n >= (1ull << 32) - 1) return hipErrorInvalidValue;
    hipError_t e = sort_scratch_reserve(st, s, n);
    if (e != hipSuccess) return e;
    // the producer of the histograms (have_hist) counted digits of s.hist_rb bits: whoever set hist_bits chose it with sort_plan_rb
    const int rb = have_hist ? s.hist_rb : sort_plan_rb(s, n, key_bits);
    if (rb != 8 && rb != 9) return hipErrorInvalidValue;
    const int passes = sort_plan_passes(key_bits, rb);
    // 8192-key tiles of 512 threads x 16 keys for every size.  Rounds 2-3 ran sorts of <= 1.5M keys as 1024 x 8 (the same 12.7 us per pass alone at 10^6 keys;
    // 2-3 % more frames per second with the frame lanes overlapping, as the frame then was).  With this round's frame (staged tile lists: two launches fewer,
    // a longer projection kernel) the 512-thread form wins by 3 % at C2 — 0.0980-0.0984 against 0.1010-0.1017 ms/frame, three alternating runs each; 512 x 12:
    // 0.0983-0.0997; 512 x 8: 0.104; 256 x 16: 0.105 — and 1.4 % on the 4D set (0.1213 against 0.1231): a 16-wave workgroup finds a free CU later than an
    // 8-wave one beside the other lanes' kernels.  2048-key tiles are slower alone (20.3 us) and overlapped.
    // 512 bins: the large sorts use 512 x 12 (48 KB of keys and values + 16 KB of per-wave counters: still two workgroups per CU).
    int shape = s.shape_knob ? s.shape_knob : 5;
    (void)n;
    if (rb == 9 && shape == 5 && !s.shape_knob) shape = 6;
    const bool atomic_rank = s.rank_knob ? s.rank_knob == 2 : s.atomic_rank;
#define GS4D_OS8(T, I) (atomic_rank ? onesweep<T, I, true, 8>(st, s, keys, vals, n, n_dev, passes, have_hist, identity_vals) : onesweep<T, I, false, 8>(st, s, keys, vals, n, n_dev, passes, have_hist, identity_vals))
#define GS4D_OS(T, I) (rb == 9 ? (atomic_rank ? onesweep<T, I, true, 9>(st, s, keys, vals, n, n_dev, passes, have_hist, identity_vals) : onesweep<T, I, false, 9>(st, s, keys, vals, n, n_dev, passes, have_hist, identity_vals)) : GS4D_OS8(T, I))
    switch (shape) {
    case 1: return rb == 8 ? GS4D_OS8(256, 8) : hipErrorInvalidValue;
    case 2: return GS4D_OS(512, 8);
    case 3: return GS4D_OS(1024, 8);
    case 4: return rb == 8 ? GS4D_OS8(256, 16) : hipErrorInvalidValue;
    case 5: return GS4D_OS(512, 16);
    case 6: return GS4D_OS(512, 12);
    default: return GS4D_OS(512, 4);
    }
#undef GS4D_OS
#undef GS4D_OS8
}

} // namespace gs4d
