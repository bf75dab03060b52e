// sort.hip — depth-key generation and the stable LSD radix sort (gfx950, wave64).
//
// Replaces, behind gs4d_keygen / gs4d_sort_pairs:
//   * the single-threaded CPU key loop + two uploads            4DSplatRendering/Scenes.h:28-36, 314-325
//   * radix_sort::sorter::sort and its three GLSL kernels       Dependencies/GPU_RADIX_SORT/radix_sort.hpp:258-392,
//                                                               resources/radix_sort_{count,local_offsets,reorder}.comp.glsl
// Contract kept: output == stable ascending sort by the uint32 key, payload follows (bit-exact permutation).
// Design (not a translation of the GLSL): 8-bit digits x 4 passes instead of 4-bit x 8; per pass one LDS-privatised
// histogram kernel, one row-scan kernel (one workgroup per digit) and one scatter kernel that ranks keys with wave64
// ballot match + popcount (no 16 KB Blelloch tables, no per-pass re-sort of the block), 3 launches per pass instead of
// 2*log2(P2)+3.
//
// Compiled with -ffp-contract=off: the key must be bit-identical to the CPU expression
// 1.0f / sqrtf(dx*dx + dy*dy + dz*dz) (IEEE-correct sqrt and divide are hipcc's default).
#include "gs4d_internal.h"

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

template <int ITEMS>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint32_t* __restrict__ keys, uint32_t n_cap, const uint32_t* __restrict__ n_dev, int shift,
                                                        uint32_t* __restrict__ hist, uint32_t nblocks) {
    __shared__ uint32_t h[256];
    const uint32_t n = n_dev ? min(*n_dev, n_cap) : n_cap;
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (RS_THREADS * ITEMS);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        uint32_t i = base + j * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// one workgroup per digit: exclusive scan of hist[d][0..nblocks) in place, totals[d] = row sum
__global__ __launch_bounds__(256) void k_rs_scan(uint32_t* __restrict__ hist, uint32_t nblocks, uint32_t* __restrict__ totals) {
    __shared__ uint32_t part[256];
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    const uint32_t chunk = (nblocks + 255u) / 256u;
    const uint32_t b0 = min(threadIdx.x * chunk, nblocks), b1 = min(b0 + chunk, nblocks);
    uint32_t s = 0;
    for (uint32_t b = b0; b < b1; ++b) s += row[b];
    part[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over 256 partials
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t v = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;       // exclusive prefix of this thread's chunk
    for (uint32_t b = b0; b < b1; ++b) { uint32_t v = row[b]; row[b] = run; run += v; }
    if (threadIdx.x == 255) totals[blockIdx.x] = part[255];
}

template <int ITEMS>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                           uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                           uint32_t n_cap, const uint32_t* __restrict__ n_dev, int shift,
                                                           const uint32_t* __restrict__ hist, uint32_t nblocks, const uint32_t* __restrict__ totals) {
    __shared__ uint32_t wcnt[RS_WAVES][256];
    __shared__ uint32_t scan[256];
    const uint32_t n = n_dev ? min(*n_dev, n_cap) : n_cap;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint32_t wbase = blockIdx.x * (RS_THREADS * ITEMS) + w * (64u * ITEMS);
    if (wbase - w * (64u * ITEMS) >= n) return;        // whole block past the end (uniform)

#pragma unroll
    for (int k = 0; k < RS_WAVES; ++k) wcnt[k][tid] = 0;
    // digit bases: exclusive scan of the 256 digit totals
    const uint32_t tot = totals[tid];
    scan[tid] = tot;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t v = tid >= (unsigned)off ? scan[tid - off] : 0u;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    const uint32_t digit_base = scan[tid] - tot;

    uint32_t key[ITEMS], rank[ITEMS];
    const uint64_t lt = (1ull << lane) - 1ull;
    volatile uint32_t* wc = wcnt[w];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t i = wbase + j * 64u + lane;
        const bool valid = i < n;
        key[j] = valid ? keys_in[i] : 0xFFFFFFFFu;
        const uint32_t d = (key[j] >> shift) & 255u;
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        if (valid) {
            const uint32_t c = wc[d];
            rank[j] = c + (uint32_t)__popcll(m & lt);
            __builtin_amdgcn_wave_barrier();
            if ((m & lt) == 0) wc[d] = c + (uint32_t)__popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {   // thread d: turn per-wave counts into per-wave global start positions for digit d
        uint32_t run = digit_base + hist[tid * nblocks + blockIdx.x];
#pragma unroll
        for (int k = 0; k < RS_WAVES; ++k) { uint32_t t = wcnt[k][tid]; wcnt[k][tid] = run; run += t; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t i = wbase + j * 64u + lane;
        if (i < n) {
            const uint32_t d = (key[j] >> shift) & 255u;
            const uint32_t pos = wcnt[w][d] + rank[j];
            keys_out[pos] = key[j];
            vals_out[pos] = vals_in[i];
        }
    }
}

hipError_t sort_scratch_reserve(SortScratch& s, size_t n) {
    hipError_t e;
    if (s.cap < n) {
        if (s.keys2) (void)hipFree(s.keys2);
        if (s.vals2) (void)hipFree(s.vals2);
        s.keys2 = s.vals2 = nullptr; s.cap = 0;
        if ((e = hipMalloc(&s.keys2, n * 4)) != hipSuccess) return e;
        if ((e = hipMalloc(&s.vals2, n * 4)) != hipSuccess) return e;
        s.cap = n;
    }
    size_t nb = (n + 1023) / 1024;              // smallest block = 1024 keys
    if (s.hist_cap < nb * 256) {
        if (s.hist) (void)hipFree(s.hist);
        s.hist = nullptr; s.hist_cap = 0;
        if ((e = hipMalloc(&s.hist, nb * 256 * 4)) != hipSuccess) return e;
        s.hist_cap = nb * 256;
    }
    if (!s.totals) { if ((e = hipMalloc(&s.totals, 256 * 4)) != hipSuccess) return e; }
    return hipSuccess;
}

void sort_scratch_free(SortScratch& s) {
    if (s.keys2) (void)hipFree(s.keys2);
    if (s.vals2) (void)hipFree(s.vals2);
    if (s.hist) (void)hipFree(s.hist);
    if (s.totals) (void)hipFree(s.totals);
    s = SortScratch();
}

template <int ITEMS>
static hipError_t sort_passes(hipStream_t st, SortScratch& s, uint32_t* keys, uint32_t* vals, size_t n, const uint32_t* n_dev, int key_bits) {
    const uint32_t kpb = RS_THREADS * ITEMS;
    const uint32_t nblocks = (uint32_t)((n + kpb - 1) / kpb);
    uint32_t* kin = keys; uint32_t* vin = vals; uint32_t* kout = s.keys2; uint32_t* vout = s.vals2;
    int passes = (key_bits + 7) / 8;
    for (int p = 0; p < passes; ++p) {
        k_rs_hist<ITEMS><<<dim3(nblocks), dim3(RS_THREADS), 0, st>>>(kin, (uint32_t)n, n_dev, 8 * p, s.hist, nblocks);
        k_rs_scan<<<dim3(256), dim3(256), 0, st>>>(s.hist, nblocks, s.totals);
        k_rs_scatter<ITEMS><<<dim3(nblocks), dim3(RS_THREADS), 0, st>>>(kin, vin, kout, vout, (uint32_t)n, n_dev, 8 * p, s.hist, nblocks, s.totals);
        uint32_t* t;
        t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (passes & 1) {   // odd number of passes: result sits in the scratch buffers
        if ((e = hipMemcpyAsync(keys, s.keys2, n * 4, hipMemcpyDeviceToDevice, st)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(vals, s.vals2, n * 4, hipMemcpyDeviceToDevice, st)) != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t radix_sort_pairs(hipStream_t st, SortScratch& s, uint32_t* keys, uint32_t* vals, size_t n, const uint32_t* n_dev, int key_bits) {
    if (n <= 1) return hipSuccess;                    // radix_sort.hpp:260
    hipError_t e = sort_scratch_reserve(s, n);
    if (e != hipSuccess) return e;
    if (n <= (size_t)3 << 20) return sort_passes<4>(st, s, keys, vals, n, n_dev, key_bits);
    return sort_passes<16>(st, s, keys, vals, n, n_dev, key_bits);
}

} // namespace gs4d
