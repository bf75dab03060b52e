// fetch_calib.hip — known-bytes micro-kernels for calibrating rocprofv3's FETCH_SIZE on gfx950 in THIS library's access patterns.
// MI355X_MICROARCH.md says FETCH_SIZE reports half the bytes of a wide coalesced streaming read and leaves other widths uncalibrated;
// the compositor gathers 48 of every 64-byte record and (round 1) the binning kernel gathered 8-byte rectangles.  Each kernel below
// touches a table far larger than the 256 MiB Infinity Cache exactly once per element it names, so the bytes it MUST fetch are known:
//   k_stream16   every lane 16 B, consecutive                       requested = distinct 64-B sectors = distinct 128-B lines = N * 16
//   k_gather8    every lane 8 B at a random 8-B slot                requested 8 B;  sectors 64 B;  lines 128 B  (slots are distinct lines)
//   k_gather48   every lane 48 of a random 64-B record (3 x 16 B)   requested 48 B; sectors 64 B;  lines 128 B when the neighbour is not taken
//   k_gather64   every lane a random 64-B record (4 x 16 B)         requested 64 B; sectors 64 B
// Build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib ; run under  rocprofv3 --pmc FETCH_SIZE  (tools/fetch_calib.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void k_stream16(const float4* __restrict__ t, size_t n, float* __restrict__ sink) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = t[i];
    if (v.x == 12345.678f) sink[0] = v.y + v.z + v.w;
}
// slots = number of 128-B lines in the table; the permutation i -> (i * odd) mod 2^k visits distinct lines
__global__ void k_gather8(const uint2* __restrict__ t, uint32_t nlines_log2, size_t n, float* __restrict__ sink) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t line = (uint32_t)((i * 2654435761ull) & ((1ull << nlines_log2) - 1ull));
    const uint2 v = t[(size_t)line * 16 + (mix((uint32_t)i) & 15u)];           // one 8-B slot of a 128-B line nobody else touches
    if (v.x == 0xDEADBEEFu) sink[0] = (float)v.y;
}
__global__ void k_gather48(const float4* __restrict__ t, uint32_t nrec_log2, size_t n, float* __restrict__ sink) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t rec = (uint32_t)((i * 2654435761ull) & ((1ull << nrec_log2) - 1ull));       // distinct 64-B records (a random half of the 128-B lines' halves)
    const float4* r = t + (size_t)rec * 4;
    const float4 a = r[0], b = r[1], c = r[2];
    if (a.x == 12345.678f) sink[0] = b.x + c.x;
}
__global__ void k_gather64(const float4* __restrict__ t, uint32_t nrec_log2, size_t n, float* __restrict__ sink) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t rec = (uint32_t)((i * 2654435761ull) & ((1ull << nrec_log2) - 1ull));
    const float4* r = t + (size_t)rec * 4;
    const float4 a = r[0], b = r[1], c = r[2], d = r[3];
    if (a.x == 12345.678f) sink[0] = b.x + c.x + d.x;
}

int main() {
    const size_t bytes = (size_t)4 << 30;                   // 4 GiB table: 16x the Infinity Cache
    void* t = nullptr; float* sink = nullptr;
    if (hipMalloc(&t, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { std::fprintf(stderr, "hipMalloc failed\n"); return 1; }
    (void)hipMemset(t, 0, bytes);
    (void)hipDeviceSynchronize();
    const size_t n16 = bytes / 16;                          // k_stream16: the whole table once
    const uint32_t lines_log2 = 25;                         // 2^25 128-B lines = 4 GiB
    const uint32_t rec_log2 = 26;                           // 2^26 64-B records = 4 GiB
    const size_t ng = (size_t)1 << 24;                      // 16.8 M gathers per gather kernel: distinct lines / records (multiplicative permutation)
    for (int rep = 0; rep < 3; ++rep) {
        k_stream16<<<dim3((unsigned)((n16 + 255) / 256)), dim3(256)>>>((const float4*)t, n16, sink);
        k_gather8<<<dim3((unsigned)((ng + 255) / 256)), dim3(256)>>>((const uint2*)t, lines_log2, ng, sink);
        k_gather48<<<dim3((unsigned)((ng + 255) / 256)), dim3(256)>>>((const float4*)t, rec_log2, ng, sink);
        k_gather64<<<dim3((unsigned)((ng + 255) / 256)), dim3(256)>>>((const float4*)t, rec_log2, ng, sink);
    }
    if (hipDeviceSynchronize() != hipSuccess) { std::fprintf(stderr, "kernel failed\n"); return 1; }
    std::printf("known bytes per launch: k_stream16 requested=%zu ; k_gather8 requested=%zu sectors64=%zu lines128=%zu ; k_gather48 requested=%zu sectors64=%zu ; k_gather64 requested=%zu\n",
                bytes, ng * 8, ng * 64, ng * 128, ng * 48, ng * 64, ng * 64);
    (void)hipFree(t); (void)hipFree(sink);
    return 0;
}
