// composite2.hip — compositing kernel of the unordered draw path: one wave64 per 8x8 tile, the tile's list is ORDERED HERE.
//
// Same arithmetic contract as composite.hip (rasteriser Geometry.h:44-50 / Renderer.cpp:33-39, fragment shaders
// Shader/Splats4D/Splat4DFragShader.GLSL:16-31 and the 3D/2D variants, blend Application.cpp:150-154): the chunk walk is the shared
// composite_chunk().  What differs is where the blend order comes from.  tilelist.hip leaves every tile an UNORDERED list of
// (key, record) entries; "instance order" — the order the reference's ROP blends in — is ascending (key, record) (KeySrc,
// gs4d_internal.h).  The wave reads its whole list into registers (PER entries per lane; k_bucket_tiles guarantees it fits, otherwise the
// draw was aborted and re-run on the ordered path), sorts it with a wave-local LSD radix sort over 8-bit digits — 256 counters, four
// per lane — and then walks it from the end (front-most) as composite.hip does.
//
// Radix pass, one wave, no other wave to wait for: count the digit with no-return LDS atomics, exclusive scan of the 256
// counters (inside each lane, then across the wave), then a RETURNING LDS atomic add on the digit's running position gives every element its slot.  Stable because
// (a) the LDS unit serialises the lanes of one instruction that hit the same counter in ascending lane order — verified on the device
// at context creation (lds_atomic_order_selftest; the unordered path is not used if the test fails) — and (b) a wave's LDS
// instructions execute in program order, so element j*64+lane is ranked before element (j+1)*64+lane'.  Digits on which every key of
// the list agrees are skipped.  Keys that tie are rare (24-bit depth keys, tens of entries): the list is sorted on the key alone and
// checked for equal neighbours; only then is it sorted on the record index first and on the key again.
#include "composite_common.h"
#include <cstdlib>
#include <cstdio>
#include <vector>

namespace gs4d {

constexpr int WS_DIGIT_BITS = 8;                                   // digit of the wave-local sort: 256 counters, four per lane (6-bit digits, one counter per lane, take four passes over 24-bit keys instead of three)
constexpr int WS_BINS = 1 << WS_DIGIT_BITS, WS_CPL = WS_BINS / 64; // counters per lane

template <int PER>
struct WaveSort {
    uint32_t k[PER], r[PER];

    __device__ __forceinline__ void pass(bool on_rec, int shift, uint32_t E, uint32_t* ek, uint32_t* er, uint32_t* cnt /* [WS_BINS] */, uint32_t lane) {
#pragma unroll
        for (int q = 0; q < WS_CPL; ++q) cnt[q * 64 + lane] = 0u;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if ((uint32_t)j * 64u >= E) break;                                     // uniform
            if ((uint32_t)j * 64u + lane < E) { const uint32_t d = ((on_rec ? r[j] : k[j]) >> shift) & (uint32_t)(WS_BINS - 1); __hip_atomic_fetch_add(&cnt[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        }
        __syncthreads();
        // lane l owns counters l * WS_CPL .. l * WS_CPL + WS_CPL - 1 (consecutive digits): exclusive scan inside the lane, then across the wave
        static_assert(WS_CPL == 4, "a lane's counters are read and written as one uint4");
        const uint4 cc = reinterpret_cast<const uint4*>(cnt)[lane];
        const uint32_t c[WS_CPL] = { cc.x, cc.y, cc.z, cc.w };
        const uint32_t sum = (cc.x + cc.y) + (cc.z + cc.w);
        const bool all = cc.x == E || cc.y == E || cc.z == E || cc.w == E;
        if (__ballot(all) != 0ull) return;                                         // uniform: one digit holds every key — a stable identity
        uint32_t inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(inc, off, 64); if (lane >= (unsigned)off) inc += t; }
        uint32_t run = inc - sum;
        __syncthreads();                                                            // (everybody has read the counts)
        reinterpret_cast<uint4*>(cnt)[lane] = make_uint4(run, run + c[0], run + c[0] + c[1], run + c[0] + c[1] + c[2]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if ((uint32_t)j * 64u >= E) break;
            if ((uint32_t)j * 64u + lane < E) {
                const uint32_t d = ((on_rec ? r[j] : k[j]) >> shift) & (uint32_t)(WS_BINS - 1);
                const uint32_t dest = atomicAdd(&cnt[d], 1u);                      // lane-ordered within the instruction, program-ordered across j
                ek[dest] = k[j]; er[dest] = r[j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if ((uint32_t)j * 64u >= E) break;
            if ((uint32_t)j * 64u + lane < E) { k[j] = ek[j * 64 + lane]; r[j] = er[j * 64 + lane]; }
        }
        __syncthreads();
    }
};

template <bool PREMULT_C, int PER>
__global__ __launch_bounds__(64) void k_composite_v2(const float4* __restrict__ proj, const uint2* __restrict__ entries, const uint32_t* __restrict__ tstart, const uint32_t* __restrict__ tcnt,
                                                     const uint32_t* __restrict__ total, uint32_t* __restrict__ total_host, int tiles_x, int W, int H, uint32_t* __restrict__ tstate, uint32_t epoch, float4 clear,
                                                     float4* __restrict__ fb, int key_passes, int rec_passes, uint32_t slabs,
                                                     const uint4* __restrict__ bstat, uint32_t nb, const uint32_t* __restrict__ sstat, uint32_t rows, uint32_t stage_seq, uint32_t rcap, uint32_t scap, uint32_t bcap,
                                                     TileBox box, uint32_t box_blocks, unsigned long long* __restrict__ stamps) {
    // the sort's key plane and the blend's record staging never live at the same time: one piece of LDS serves both
    constexpr int SHARED_WORDS = 64 * PER > 64 * 3 * 4 ? 64 * PER : 64 * 3 * 4;
    __shared__ __attribute__((aligned(16))) uint32_t sh_a[SHARED_WORDS];
    __shared__ uint32_t pmask[64 * 2];
    __shared__ __attribute__((aligned(16))) uint32_t cnt[WS_BINS];
    __shared__ uint32_t er[64 * PER];
    float4* stage = reinterpret_cast<float4*>(sh_a);
    uint32_t* ek = sh_a;
    uint32_t tile;
    const bool real = composite_tile(blockIdx.x, tiles_x, box, tile);     // false: padding of the XCD-aware grid
    const uint32_t lane = threadIdx.x;
#ifdef GS4D_TUNING
    const unsigned long long st0 = wall_clock64(); unsigned long long st1 = 0, st2 = 0; uint32_t stE = 0;      // per-tile stamps (100 MHz): start, list ordered, (end), entries
#endif
    // what the list kernels found out — entries, longest list, abort flags — goes to the host from here (pinned, mapped memory behind
    // the lane's event): they are complete now, and none of them has to wait for a hand-off of its own
    // A staged draw (stage_seq != 0, tilelist.hip) has no scan kernel that could have added anything up: its verdict is the abort word and the
    // per-bucket statistics {entries, longest run, longest list} and per-segment entry counts, reduced here.  An exact draw reports its statistics
    // the same way (they size the blocks, runs and buckets of the staged draws that follow) beside the totals its scan kernel left.
    // Everything the tile needs before it can ask for its list is requested HERE, in one go — the verdict word, the tile's table row, its state
    // word: three independent loads, one round trip.  (Issued one behind the other's branch they were three round trips: per-tile stamps of a
    // TUNING build showed 3.9 us between a workgroup's start and its list being in order for lists of <= 64 entries — tools/v2_timeline.py.)
    const uint32_t verdict = stage_seq ? total[TL_ABORT_WORD] : total[1];
    uint32_t my_start = 0u, my_cnt = 0u;                    // lane s: sub-list s of this tile (one load for the whole table row)
    if (real && lane < slabs) { my_start = tstart[(size_t)tile * slabs + lane]; my_cnt = tcnt[(size_t)tile * slabs + lane]; }
    const uint32_t tstate_word = real ? tstate[tile] : 0u;
    const bool aborted = stage_seq ? verdict == stage_seq : verdict != 0u;
    if (blockIdx.x == 0u) {
        unsigned long long sum = 0ull; uint32_t mrun = 0u, mbucket = 0u, mlist = 0u, mseg = 0u, used = BOX_EMPTY;
        if (bstat) for (uint32_t b = lane; b < nb; b += 64u) { const uint4 v = bstat[b]; sum += v.x; mrun = max(mrun, v.y); mbucket = max(mbucket, v.x); mlist = max(mlist, v.z); used = box_join(used, v.w); }
        if (sstat) for (uint32_t w = lane; w < rows; w += 64u) mseg = max(mseg, sstat[w]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sum += __shfl_xor(sum, off, 64); mrun = max(mrun, (uint32_t)__shfl_xor(mrun, off, 64));
            mbucket = max(mbucket, (uint32_t)__shfl_xor(mbucket, off, 64)); mlist = max(mlist, (uint32_t)__shfl_xor(mlist, off, 64)); mseg = max(mseg, (uint32_t)__shfl_xor(mseg, off, 64));
            used = box_join(used, (uint32_t)__shfl_xor((int)used, off, 64));
        }
        if (lane == 0u) {
            total_host[6] = mrun; total_host[7] = mbucket; total_host[8] = mseg;
            total_host[9] = stage_seq ? used : BOX_NONE;         // the blocks of tiles that hold entries (staged draws: k_bucket_tiles_staged knows; BOX_EMPTY: none)
            if (stage_seq) {
                // flags: 4 = a segment, a run or a bucket did not fit what the host guessed, or an entry lies outside the launch box (re-run exactly), 2 = a list longer than the compositor was launched for.
                // (A segment that overflowed wrote no entries: the bucket statistics then count entries that are not there, and the sum is still the true total.)
                const bool in_box = used == BOX_EMPTY || (box_holds(box_blocks, used & 255u, (used >> 8) & 255u) && box_holds(box_blocks, (used >> 16) & 255u, used >> 24));
                const bool guess_ok = mrun <= rcap && mbucket <= bcap && mseg <= scap && in_box;
                const uint32_t fl = (guess_ok ? 0u : 4u) | ((aborted && guess_ok) ? 2u : 0u);
                total_host[0] = sum > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sum; total_host[2] = (uint32_t)sum; total_host[3] = (uint32_t)(sum >> 32); total_host[5] = mlist; total_host[1] = fl;
            } else { total_host[0] = total[0]; total_host[2] = total[2]; total_host[3] = total[3]; total_host[5] = total[4]; total_host[1] = total[1]; }
        }
    }
    if (aborted || !real) return;                           // aborted draw (capacity or list length): the host re-runs it
    const int tx0 = (int)(tile % (uint32_t)tiles_x) * TILE, ty0 = (int)(tile / (uint32_t)tiles_x) * TILE;
    const int px = tx0 + (int)(lane & 7u), py = ty0 + (int)(lane >> 3);
    const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    float T = 1.0f, Cr = 0.0f, Cg = 0.0f, Cb = 0.0f, A = 0.0f;
    // the tile's list is `slabs` sub-lists by the top bits of the key; larger key = nearer: the last slab is blended first
    if (__ballot(my_cnt != 0u) == 0ull) return;             // nothing is drawn on this tile: its pixels, or its being clear (composite.hip: tile state), stay as they are
    const bool fb_is_clear = tstate_word != epoch;          // uniform: the tile's pixels are not in memory yet
    for (int sb = (int)slabs - 1; sb >= 0; --sb) {
        const uint32_t start = __shfl(my_start, sb, 64);
        const uint32_t E = min((uint32_t)__shfl(my_cnt, sb, 64), (uint32_t)(64 * PER));         // k_bucket_tiles guarantees the bound; min() keeps a broken promise inside LDS
        if (E == 0u) continue;
        if (E > 1u) {
            WaveSort<PER> ws;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                ws.k[j] = 0u; ws.r[j] = 0u;
                if ((uint32_t)j * 64u + lane < E) { const uint2 e = entries[start + j * 64 + lane]; ws.k[j] = e.x; ws.r[j] = e.y; }
            }
            for (int p = 0; p < key_passes; ++p) ws.pass(false, WS_DIGIT_BITS * p, E, ek, er, cnt, lane);
            // the sorted keys go to LDS for the neighbour test (the last pass may have been an identity that wrote nothing)
#pragma unroll
            for (int j = 0; j < PER; ++j) { if ((uint32_t)j * 64u + lane < E) { ek[j * 64 + lane] = ws.k[j]; er[j * 64 + lane] = ws.r[j]; } }
            __syncthreads();
            bool tie = false;
#pragma unroll
            for (int j = 0; j < PER; ++j) { const uint32_t i = (uint32_t)j * 64u + lane; if (i + 1u < E) tie |= ek[i] == ek[i + 1u]; }
            if (__ballot(tie) != 0ull) {                    // equal keys: instance order among them is ascending record index
                __syncthreads();
                for (int p = 0; p < rec_passes; ++p) ws.pass(true, WS_DIGIT_BITS * p, E, ek, er, cnt, lane);
                for (int p = 0; p < key_passes; ++p) ws.pass(false, WS_DIGIT_BITS * p, E, ek, er, cnt, lane);
#pragma unroll
                for (int j = 0; j < PER; ++j) { if ((uint32_t)j * 64u + lane < E) er[j * 64 + lane] = ws.r[j]; }
            }
            __syncthreads();
        } else {
            if (lane == 0) er[0] = entries[start].y;
            __syncthreads();
        }
#ifdef GS4D_TUNING
        if (!st1) st1 = wall_clock64();
        stE += E;
#endif
        for (uint32_t hi = E; hi > 0u;) {
            const uint32_t c = min(64u, hi);
            const uint32_t rec = lane < c ? er[hi - 1u - lane] : 0u;       // lane s holds list entry hi-1-s : s = 0 is the front-most of the chunk
            composite_chunk<PREMULT_C>(proj, rec, c, lane, tx0, ty0, fx, fy, stage, pmask, 0, T, Cr, Cg, Cb, A);
            hi -= c;
            if (__ballot(T > 0.0f) == 0ull) break;          // exact: every remaining contribution is multiplied by T == 0
        }
        if (__ballot(T > 0.0f) == 0ull) break;
    }
    if (px < W && py < H) {
        const size_t o = (size_t)py * W + px;
        const float4 d = fb_is_clear ? clear : fb[o];
        fb[o] = make_float4(Cr + T * d.x, Cg + T * d.y, Cb + T * d.z, A + T * d.w);
    }
    if (lane == 0u) tstate[tile] = epoch;
#ifdef GS4D_TUNING
    if (stamps && lane == 0u) { st2 = wall_clock64(); unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(hw)); unsigned long long* o = stamps + (size_t)blockIdx.x * 6; o[0] = st0; o[1] = st1; o[2] = st2; o[3] = stE; o[4] = tile; o[5] = hw; }
#endif
}

template <bool PREMULT_C>
static hipError_t launch_v2(hipStream_t st, int per, dim3 grid, const float4* proj, const uint2* entries, const uint32_t* tstart, const uint32_t* tcnt, const uint32_t* total, uint32_t* total_host, int tiles_x, int W, int H,
                            uint32_t* tstate, uint32_t epoch, float4 c, float4* fb, int kp, int rp, uint32_t slabs, const uint4* bstat, uint32_t nb, const uint32_t* sstat, uint32_t rows, uint32_t stage_seq, uint32_t rcap, uint32_t scap, uint32_t bcap, TileBox box, uint32_t box_blocks, unsigned long long* stamps) {
#define GS4D_V2(P) k_composite_v2<PREMULT_C, P><<<grid, dim3(64), 0, st>>>(proj, entries, tstart, tcnt, total, total_host, tiles_x, W, H, tstate, epoch, c, fb, kp, rp, slabs, bstat, nb, sstat, rows, stage_seq, rcap, scap, bcap, box, box_blocks, stamps)
    switch (per) {
    case 1: GS4D_V2(1); break;
    case 2: GS4D_V2(2); break;
    case 3: GS4D_V2(3); break;
    case 4: GS4D_V2(4); break;
    case 6: GS4D_V2(6); break;
    case 8: GS4D_V2(8); break;
    case 12: GS4D_V2(12); break;
    default: GS4D_V2(16); break;
    }
#undef GS4D_V2
    return hipGetLastError();
}

hipError_t launch_composite_v2(hipStream_t st, const float4* proj, const uint2* entries, const uint32_t* tstart, const uint32_t* tcnt, const uint32_t* total, uint32_t* total_host, int tiles_x, int tiles_y, int W, int H,
                               int premult_c, uint32_t* tstate, uint32_t epoch, const float clear[4], float4* fb, uint32_t hint, int keybits, int recbits, uint32_t slabs,
                               const uint4* bstat, uint32_t nb, const uint32_t* sstat, uint32_t rows, uint32_t stage_seq, uint32_t rcap, uint32_t scap, uint32_t bcap, uint32_t box_blocks) {
    if (hint > V2_MAX_LIST) return hipErrorInvalidValue;
    const int per = (int)(v2_list_capacity(hint) / 64u);
    const float4 c = make_float4(clear[0], clear[1], clear[2], clear[3]);
    // staged draws: only the box of tiles the list kernel has checked every entry to lie in (TileLists::box); the first workgroup reports, so there is always one
    TileBox box = tile_box(stage_seq ? box_blocks : BOX_NONE, tiles_x, tiles_y);
    if (box.w == 0u || box.h == 0u) box = TileBox{ 0u, 0u, 1u, 1u };
    const dim3 grid(composite_grid((int)box.w, (int)box.h));
    int kp = (keybits + WS_DIGIT_BITS - 1) / WS_DIGIT_BITS, rp = (recbits + WS_DIGIT_BITS - 1) / WS_DIGIT_BITS;
#ifdef GS4D_TUNING
    { static const bool nosort = getenv("GS4D_V2_NOSORT") != nullptr; if (nosort) kp = rp = 0; }      // ablation: what the wave-local list sort costs the kernel (the image is then wrong)
#endif
    unsigned long long* stamps = nullptr;
#ifdef GS4D_TUNING
    // tuning aid: per-tile wall-clock stamps of the GS4D_V2_STAMP_CALL-th launch, dumped to GS4D_V2_STAMP_FILE (tools/v2_timeline.py reads it)
    static const char* stampf = getenv("GS4D_V2_STAMP_FILE");
    static const int stamp_call = getenv("GS4D_V2_STAMP_CALL") ? atoi(getenv("GS4D_V2_STAMP_CALL")) : 40;
    static int calls = 0;
    const bool stamp_now = stampf && ++calls == stamp_call;
    if (stamp_now && (hipMalloc(&stamps, (size_t)grid.x * 48) != hipSuccess || hipMemsetAsync(stamps, 0, (size_t)grid.x * 48, st) != hipSuccess)) stamps = nullptr;
#endif
    const hipError_t le = premult_c ? launch_v2<true>(st, per, grid, proj, entries, tstart, tcnt, total, total_host, tiles_x, W, H, tstate, epoch, c, fb, kp, rp, slabs, bstat, nb, sstat, rows, stage_seq, rcap, scap, bcap, box, box_blocks, stamps)
                                    : launch_v2<false>(st, per, grid, proj, entries, tstart, tcnt, total, total_host, tiles_x, W, H, tstate, epoch, c, fb, kp, rp, slabs, bstat, nb, sstat, rows, stage_seq, rcap, scap, bcap, box, box_blocks, stamps);
#ifdef GS4D_TUNING
    if (stamps) {
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> hs((size_t)grid.x * 6);
        (void)hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
        if (FILE* f = fopen(stampf, "w")) { for (uint32_t t = 0; t < grid.x; ++t) { for (int k = 0; k < 6; ++k) fprintf(f, "%llu ", hs[(size_t)t * 6 + k]); fprintf(f, "\n"); } fclose(f); }
        (void)hipFree(stamps);
    }
#endif
    return le;
}

} // namespace gs4d
