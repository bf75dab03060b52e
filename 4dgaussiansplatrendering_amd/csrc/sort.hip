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
                                                float* __restrict__ keys, uint32_t* __restrict__ idx) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 p = pos[i];
    float4 s = sig3[i];
    float key;
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

hipError_t launch_keygen(hipStream_t st, const float4* pos, const float4* sig3, size_t n, float t, const float cam[3], const float view[16], int key_mode, float* keys, uint32_t* idx) {
    if (n == 0) return hipSuccess;
    float4 vr = make_float4(view[2], view[6], view[10], view[14]);
    k_keygen<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(pos, sig3, (uint32_t)n, t, cam[0], cam[1], cam[2], vr, key_mode, keys, idx);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// radix sort
// ------------------------------------------------------------------------------------------------
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;

// ------------------------------------------------------------------------------------------------
// Single-pass-per-digit radix sort with a chained scan (decoupled look-back)
//
//   k_os_hist : one launch builds the global digit histograms of ALL passes (keys read once, LDS-privatised) and zeroes the
//               look-back words of this sort and the histogram of the NEXT sort (so no memset launch is needed).
//   k_os_pass : one launch per 8-bit digit.  A workgroup ranks the keys of its tile (wave64 ballot match), publishes its
//               per-digit counts as {flag,value} words, looks back over its predecessors' words to obtain its exclusive prefix,
//               reorders the tile in LDS and writes each digit's run to its final place with consecutive lanes on consecutive
//               addresses.  Traffic per pass: keys+values read once, written once.
//   A pass whose digit is the same for every key (e.g. sign+exponent byte of the positive depth keys) is a stable identity and
//   is skipped ON THE DEVICE: every workgroup derives, from the histograms, which passes are live and which of three buffers
//   (caller's, scratch B, scratch C) it reads and writes, so that the last live pass lands in the caller's buffers.
// Inter-workgroup hand-off: each status word is ONE naturally aligned 32-bit {flag:2,value:30} granule written by one
// agent-scope relaxed atomic store and polled with agent-scope relaxed atomic loads (bypass L1, write-through) — the
// data-tagged granule form of cdna_hip_programming.md Guideline 16 (R2): no separate flag, hence no ordering requirement.
// Tile id = blockIdx.x: a tile waits only for lower-numbered tiles, which the dispatcher has started earlier (observed in-order
// dispatch; not an API guarantee), so every spin is bounded: on time-out the kernel raises `err` and leaves, and the host
// reports the frame as failed instead of hanging the GPU or returning wrong data.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t OS_FLAG_AGG = 1u << 30, OS_FLAG_INCL = 2u << 30, OS_VAL_MASK = (1u << 30) - 1u;
constexpr int OS_MAX_PASSES = 4;
constexpr uint32_t OS_GROUP = 32;         // tiles per look-back group

struct OsBufs { uint32_t* k[3]; uint32_t* v[3]; };      // [0] caller's buffers, [1],[2] scratch

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(v, off, 64); if (lane >= (unsigned)off) v += t; }
    return v;
}
// exclusive scan over 256 threads (4 waves); `tmp` = __shared__ uint32_t[4]; two barriers
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* tmp, uint32_t tid) {
    const uint32_t lane = tid & 63u, w = tid >> 6;
    const uint32_t inc = wave_incl_scan_u32(v, lane);
    __syncthreads();
    if (lane == 63u) tmp[w] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) if ((unsigned)k < w) base += tmp[k];
    return base + inc - v;
}

__global__ __launch_bounds__(256) void k_os_hist(const uint32_t* __restrict__ keys, uint32_t n_cap, const uint32_t* __restrict__ n_dev, int passes,
                                                 uint32_t* __restrict__ ghist /* [4][256], zero on entry */, uint32_t* __restrict__ ghist_next /* zeroed here */,
                                                 uint32_t* __restrict__ status, uint32_t status_words) {
    __shared__ uint32_t h[OS_MAX_PASSES][256];
    const uint32_t n = n_dev ? min(*n_dev, n_cap) : n_cap;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
#pragma unroll
    for (int p = 0; p < OS_MAX_PASSES; ++p) h[p][tid] = 0;
    // housekeeping for the passes of this sort and the histogram of the next one
    for (uint32_t i = blockIdx.x * 256u + tid; i < status_words; i += gridDim.x * 256u) status[i] = 0u;
    if (blockIdx.x == 0) { for (int p = 0; p < OS_MAX_PASSES; ++p) ghist_next[p * 256 + tid] = 0u; }
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
            const uint32_t kv[4] = { kk[u].x, kk[u].y, kk[u].z, kk[u].w };
            const uint64_t act = __ballot(in[u]);
            if (act == 0ull) continue;
            const uint32_t first = (uint32_t)__ffsll((long long)act) - 1u;
#pragma unroll
            for (int p = 0; p < OS_MAX_PASSES; ++p) {
                if (p >= passes) break;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t d = (kv[c] >> (8 * p)) & 255u;
                    // skewed digits (e.g. the exponent byte of depth keys) would serialise LDS atomics: one add per wave when uniform
                    const uint32_t d0 = __shfl(d, (int)first, 64);
                    const bool uni = __ballot(in[u] && d == d0) == act;
                    if (uni) { if (lane == first) atomicAdd(&h[p][d0], (uint32_t)__popcll(act)); }
                    else if (in[u]) atomicAdd(&h[p][d], 1u);
                }
            }
        }
    }
    if (blockIdx.x == 0 && tid < (n & 3u)) {                                                 // tail keys
        const uint32_t k = keys[nvec * 4u + tid];
        for (int p = 0; p < passes; ++p) atomicAdd(&h[p][(k >> (8 * p)) & 255u], 1u);
    }
    __syncthreads();
    for (int p = 0; p < passes; ++p) { const uint32_t v = h[p][tid]; if (v) atomicAdd(&ghist[p * 256 + tid], v); }
}

// Which passes are live, and which buffers pass `p` reads and writes.  Returns false when pass p is skipped.
__device__ __forceinline__ bool os_schedule(const uint32_t* s_live /* [4] 0/1 */, int passes, int p, int& src, int& dst) {
    int k = 0, j = 0;
    for (int q = 0; q < passes; ++q) { if (s_live[q]) { if (q < p) ++j; ++k; } }
    if (!s_live[p]) return false;
    // buffer sequence ending in buffer 0: even k: 0,1,0,1,...  odd k >= 3: 0,1,2,0,1,0,...  k == 1: 0 -> 1 (copied back afterwards)
    auto buf_at = [k](int i) { if (k & 1) { if (k == 1) return i; if (i <= 2) return i; return (i - 3) & 1; } return i & 1; };
    src = buf_at(j);
    dst = buf_at(j + 1);
    return true;
}

// Batched descending look-back over rows hi, hi-1, ..., lo of a [row][256] status array for digit `tid`.  Adds the values of the
// rows visited to `sum`; stops early (returns true) at a row flagged INCL.  Up to LB loads are in flight together, so a batch
// costs one memory round trip; an unpublished row is polled alone (bounded) before the walk resumes.
template <int LB>
__device__ __forceinline__ bool os_lookback(const uint32_t* st, int32_t hi, int32_t lo, uint32_t tid, uint32_t& sum, uint32_t* err) {
    int32_t t = hi;
    uint32_t spins = 0;
    while (t >= lo) {
        uint32_t sv[LB];
#pragma unroll
        for (int k = 0; k < LB; ++k) {
            const int32_t tt = t - k;
            sv[k] = tt >= lo ? __hip_atomic_load(st + (size_t)tt * 256u + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        }
        int used = 0;
        bool incl = false;
#pragma unroll
        for (int k = 0; k < LB; ++k) {
            if (incl || used != k || t - k < lo) continue;
            const uint32_t f = sv[k] >> 30;
            if (f == 0u) continue;                                // not published yet
            sum += sv[k] & OS_VAL_MASK;
            used = k + 1;
            if (f != 1u) incl = true;
        }
        if (incl) return true;
        t -= used;
        if (t >= lo && used < LB) {                               // row t is unpublished: poll that one word
            const uint32_t* p = st + (size_t)t * 256u + tid;
            while ((__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 30) == 0u) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 21)) { atomicExch(err, 1u); return true; }
            }
        }
    }
    return false;
}

template <int ITEMS>
__global__ __launch_bounds__(RS_THREADS) void k_os_pass(OsBufs bufs, uint32_t n_cap, const uint32_t* __restrict__ n_dev, int pass, int passes,
                                                        const uint32_t* __restrict__ ghist /* [4][256] */,
                                                        uint32_t* status /* [tiles][256] of this pass, zeroed */, uint32_t* gstatus /* [groups][256], zeroed */, uint32_t* err, int dbg, unsigned long long* stamps) {
    constexpr uint32_t TILE_KEYS = RS_THREADS * ITEMS;
    __shared__ uint32_t skeys[TILE_KEYS];
    __shared__ uint32_t svals[TILE_KEYS];
    __shared__ uint32_t wcnt[RS_WAVES][256];
    __shared__ uint32_t loff[256];      // first local slot of digit d in the reordered tile
    __shared__ uint32_t gpos[256];      // global slot of that first element
    __shared__ uint32_t s_tmp[4];
    __shared__ uint32_t s_live[OS_MAX_PASSES];

    const uint32_t n = n_dev ? min(*n_dev, n_cap) : n_cap;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint32_t tile = blockIdx.x;
    const uint32_t ntiles = (n + TILE_KEYS - 1u) / TILE_KEYS;
    if (tile >= ntiles) return;                                   // uniform
    if (stamps && tid == 0) stamps[tile * 8 + 0] = wall_clock64();
    const int shift = 8 * pass;
    if (tid < OS_MAX_PASSES) s_live[tid] = 1u;
#pragma unroll
    for (int k = 0; k < RS_WAVES; ++k) wcnt[k][tid] = 0;
    __syncthreads();
    uint32_t tot = 0;
    for (int q = 0; q < passes; ++q) { const uint32_t g = ghist[q * 256 + tid]; if (q == pass) tot = g; if (g == n) s_live[q] = 0u; }   // one digit holds every key
    __syncthreads();
    int src, dst;
    if (!os_schedule(s_live, passes, pass, src, dst)) return;    // uniform: this pass is an identity
    const uint32_t* __restrict__ keys_in = bufs.k[src]; const uint32_t* __restrict__ vals_in = bufs.v[src];
    uint32_t* __restrict__ keys_out = bufs.k[dst]; uint32_t* __restrict__ vals_out = bufs.v[dst];

    const uint32_t tbase = tile * TILE_KEYS;
    const uint32_t wbase = tbase + w * (64u * ITEMS);
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t i = wbase + j * 64u + lane;
        const bool valid = i < n;
        key[j] = valid ? keys_in[i] : 0xFFFFFFFFu;
        val[j] = valid ? vals_in[i] : 0u;
    }
    const uint32_t digit_base = block_excl_scan_256(tot, s_tmp, tid);          // overlaps the loads above
    if (stamps && tid == 0) { stamps[tile * 8 + 1] = wall_clock64(); stamps[tile * 8 + 6] = key[0]; }

    const uint64_t lt = (1ull << lane) - 1ull;
    volatile uint32_t* wc = wcnt[w];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const bool valid = (wbase + j * 64u + lane) < n;
        const uint32_t d = (key[j] >> shift) & 255u;
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
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
    __syncthreads();
    // thread d: counts per wave -> offsets inside the tile's digit-d run; tile count of digit d
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < RS_WAVES; ++k) { const uint32_t t = wcnt[k][tid]; wcnt[k][tid] = cnt; cnt += t; }
    if (stamps && tid == 0) stamps[tile * 8 + 2] = wall_clock64();
    // Publish, then look back — two levels, so that the walk costs ~3 memory round trips however many tiles start together:
    // tiles in groups of OS_GROUP; the last tile of a group also publishes the group's aggregate / inclusive prefix.
    uint32_t* my = status + (size_t)tile * 256u + tid;
    const uint32_t grp = tile / OS_GROUP;
    const bool last_in_group = (tile % OS_GROUP) == OS_GROUP - 1u;
    uint32_t prefix = 0;
    if (tile == 0 || (dbg & 2)) {
        __hip_atomic_store(my, OS_FLAG_INCL | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __hip_atomic_store(my, OS_FLAG_AGG | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool full = os_lookback<16>(status, (int32_t)tile - 1, (int32_t)(grp * OS_GROUP), tid, prefix, err);
        if (!full) {                                              // reached the start of the group: need the groups before it
            if (last_in_group) __hip_atomic_store(gstatus + (size_t)grp * 256u + tid, OS_FLAG_AGG | ((prefix + cnt) & OS_VAL_MASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)os_lookback<16>(gstatus, (int32_t)grp - 1, 0, tid, prefix, err);
        }
        __hip_atomic_store(my, OS_FLAG_INCL | ((prefix + cnt) & OS_VAL_MASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (last_in_group) __hip_atomic_store(gstatus + (size_t)grp * 256u + tid, OS_FLAG_INCL | ((prefix + cnt) & OS_VAL_MASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (stamps && tid == 0) stamps[tile * 8 + 3] = wall_clock64();
    // exclusive scan of the tile's digit counts -> local run starts
    loff[tid] = block_excl_scan_256(cnt, s_tmp, tid);
    gpos[tid] = digit_base + prefix;
    __syncthreads();
    // reorder inside LDS: stable (wave-major, item-major, lane order == memory order)
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if ((wbase + j * 64u + lane) < n) {
            const uint32_t d = (key[j] >> shift) & 255u;
            const uint32_t l = loff[d] + wcnt[w][d] + rank[j];
            skeys[l] = key[j];
            svals[l] = val[j];
        }
    }
    __syncthreads();
    if (stamps && tid == 0) stamps[tile * 8 + 4] = wall_clock64();
    const uint32_t tcount = min(TILE_KEYS, n - tbase);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t l = j * RS_THREADS + tid;
        if (l < tcount) {
            const uint32_t k = skeys[l];
            const uint32_t d = (k >> shift) & 255u;
            const uint32_t o = gpos[d] + (l - loff[d]);
            keys_out[o] = k;
            vals_out[o] = svals[l];
        }
    }
    if (stamps && tid == 0) stamps[tile * 8 + 5] = wall_clock64();
}

// Exactly one live pass leaves the result in scratch buffer 1: copy it back (device-side decision; otherwise a no-op launch).
__global__ __launch_bounds__(256) void k_os_copyback(OsBufs bufs, uint32_t n_cap, const uint32_t* __restrict__ n_dev, int passes, const uint32_t* __restrict__ ghist) {
    __shared__ uint32_t s_live[OS_MAX_PASSES];
    const uint32_t n = n_dev ? min(*n_dev, n_cap) : n_cap;
    if (threadIdx.x < OS_MAX_PASSES) s_live[threadIdx.x] = 1u;
    __syncthreads();
    for (int q = 0; q < passes; ++q) if (ghist[q * 256 + threadIdx.x] == n) s_live[q] = 0u;
    __syncthreads();
    int k = 0;
    for (int q = 0; q < passes; ++q) k += s_live[q] ? 1 : 0;
    if (k != 1) return;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) { bufs.k[0][i] = bufs.k[1][i]; bufs.v[0][i] = bufs.v[1][i]; }
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
    // control block: [ghist A 4x256][ghist B 4x256] then status [4 passes][tiles][256]
    const size_t tiles = (n + 1023) / 1024;          // smallest tile = 1024 keys
    const size_t words = 2 * 4 * 256 + (size_t)OS_MAX_PASSES * (tiles + tiles / OS_GROUP + 1) * 256;
    if (s.hist_cap < words) {
        if (s.hist) { (void)hipStreamSynchronize(st); (void)hipFree(s.hist); }
        s.hist = nullptr; s.hist_cap = 0;
        if ((e = hipMalloc(&s.hist, words * 4)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(s.hist, 0, 2 * 4 * 256 * 4, st)) != hipSuccess) return e;     // both histograms start zero; each sort re-zeroes the other one
        s.hist_cap = words;
        s.flip = 0;
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

template <int ITEMS>
static hipError_t onesweep(hipStream_t st, SortScratch& s, uint32_t* keys, uint32_t* vals, size_t n, const uint32_t* n_dev, int passes) {
    static const int dbgk = getenv("GS4D_SORT_DBG") ? atoi(getenv("GS4D_SORT_DBG")) : 0;
    static const char* stampf = getenv("GS4D_SORT_STAMP_FILE");
    unsigned long long* stamps = nullptr;
    if (stampf) { if (hipMalloc(&stamps, (size_t)((n + RS_THREADS * ITEMS - 1) / (RS_THREADS * ITEMS)) * 64) != hipSuccess) stamps = nullptr; }
    const uint32_t tile_keys = RS_THREADS * ITEMS;
    const uint32_t tiles = (uint32_t)((n + tile_keys - 1) / tile_keys);
    uint32_t* ghist = s.hist + (s.flip ? 1024 : 0);
    uint32_t* ghist_next = s.hist + (s.flip ? 0 : 1024);
    s.flip ^= 1;
    uint32_t* status = s.hist + 2048;
    const uint32_t groups = tiles / OS_GROUP + 1;
    const size_t per_pass = (size_t)(tiles + groups) * 256;
    const uint32_t status_words = (uint32_t)((size_t)passes * per_pass);
    OsBufs b;
    b.k[0] = keys; b.v[0] = vals;
    b.k[1] = s.keys2; b.v[1] = s.vals2;
    b.k[2] = s.keys2 + s.cap; b.v[2] = s.vals2 + s.cap;
    const uint32_t hist_blocks = (uint32_t)std::min<size_t>((n / 16 + 255) / 256 + 1, 256);      // few workgroups: each flushes 256 global atomics per pass
    k_os_hist<<<dim3(hist_blocks), dim3(256), 0, st>>>(keys, (uint32_t)n, n_dev, passes, ghist, ghist_next, status, status_words);
    for (int p = 0; p < passes; ++p)
        k_os_pass<ITEMS><<<dim3(tiles), dim3(RS_THREADS), 0, st>>>(b, (uint32_t)n, n_dev, p, passes, ghist, status + p * per_pass, status + p * per_pass + (size_t)tiles * 256, s.err ? s.err : s.totals, dbgk, (stampf && p == 0) ? stamps : nullptr);
    k_os_copyback<<<dim3(512), dim3(256), 0, st>>>(b, (uint32_t)n, n_dev, passes, ghist);
    if (stampf) {   // debugging aid: dump per-tile wall-clock stamps (100 MHz) of pass 0
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h((size_t)tiles * 8);
        (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        FILE* f = fopen(stampf, "w");
        if (f) { for (uint32_t t = 0; t < tiles; ++t) { for (int k = 0; k < 6; ++k) fprintf(f, "%llu ", h[t * 8 + k] - h[0]); fprintf(f, "\n"); } fclose(f); }
        (void)hipFree(stamps);
    }
    return hipGetLastError();
}

hipError_t radix_sort_pairs(hipStream_t st, SortScratch& s, uint32_t* keys, uint32_t* vals, size_t n, const uint32_t* n_dev, int key_bits) {
    if (n <= 1) return hipSuccess;                    // radix_sort.hpp:260
    if (n >= (1u << 30)) return hipErrorInvalidValue;  // status words carry 30-bit counts
    hipError_t e = sort_scratch_reserve(st, s, n);
    if (e != hipSuccess) return e;
    const int passes = (key_bits + 7) / 8;
    static const int knob = getenv("GS4D_SORT_ITEMS") ? atoi(getenv("GS4D_SORT_ITEMS")) : 0;      // tuning knob (experiments only)
    const int items = knob ? knob : (n <= ((size_t)4 << 20) ? 8 : 16);
    if (items == 4) return onesweep<4>(st, s, keys, vals, n, n_dev, passes);
    if (items == 8) return onesweep<8>(st, s, keys, vals, n, n_dev, passes);
    return onesweep<16>(st, s, keys, vals, n, n_dev, passes);
}

} // namespace gs4d
