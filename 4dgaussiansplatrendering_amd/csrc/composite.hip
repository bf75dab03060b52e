// composite.hip — per-tile ordered alpha compositing into the RGBA32F framebuffer.
//
// Replaces the fixed-function half of the reference's draw: rasterisation of the instanced quads
// (Geometry.h:44-50, Renderer.cpp:33-39), the fragment shaders (Shader/Splats4D/Splat4DFragShader.GLSL:16-31,
// Shader/Splats3D/Splat3DFragShaderFull.GLSL:16-24, Shader/Splats2D/Splat2DFragShader.GLSL:10-25) and the ROP blend
// dst = src*src.a + dst*(1-src.a) on all four channels in instance order (Application.cpp:150-154).
//
// One wave64 owns one 8x8-pixel tile (lane = pixel; a tile row is one 128-B line of the framebuffer).  The tile's list
// is walked from its END (nearest splat when the instances are depth-sorted far->near) to its start, i.e. front to back:
//   C += T*a*rgb,  A += T*a*a,  T *= (1-a);   result = C + T*dst_rgb,  A + T*dst_a
// which is algebraically the reference's back-to-front "over" result.  Chunks of 64 list entries are gathered with one
// coalesced index load + one 48-B record gather per lane, staged in LDS, and broadcast to all lanes with uniform LDS reads.
//
// Coverage rule (identical arithmetic in the CPU checker, oracle/gs4d_oracle.cpp gs4do_covered):
//   dx = (i+0.5) - cx, dy = (j+0.5) - cy, u = fma(a0x,dx, a0y*dy), v = fma(a1x,dx, a1y*dy), covered iff |u|<=0.5 && |v|<=0.5
// Fragment: c = exp(-0.5 * x^T Sigma'^-1 x) with x = 8*R*S*(u,v)  ==  exp(-32*(u*u+v*v))   (R orthonormal, Sigma' = R S S R^T);
// discarded when c < 1e-4 (Splat4DFragShader.GLSL:30).
#include "composite_common.h"
#include <cstdlib>

namespace gs4d {

// ---- framebuffer tile state ("fast clear") -----------------------------------------------------------------------------------------
// tstate[tile] == epoch  <=>  the tile's 64 pixels are in memory.  Any other value: the tile is still the clear colour of the
// gs4d_clear that started the frame (each clear of an image takes a new epoch: no memset).  The compositing kernels write only the
// tiles that have list entries — at 1080p three tiles in four of the cube scenes have none — and everything that reads an image
// substitutes the clear colour for the others (the RGBA8 packs) or writes it first (k_fill_unwritten: host read-backs, float copies,
// overlay lines, which touch single pixels).  It is what a GL driver's fast clear does with glClear.

// one wave per tile: tiles whose pixels are not in memory get the clear colour
__global__ __launch_bounds__(64) void k_fill_unwritten(float4* __restrict__ fb, uint32_t* __restrict__ tstate, uint32_t epoch, float4 c, int tiles_x, int W, int H) {
    const uint32_t tile = blockIdx.x, lane = threadIdx.x;
    if (tstate[tile] == epoch) return;                      // uniform
    const int px = (int)(tile % (uint32_t)tiles_x) * TILE + (int)(lane & 7u), py = (int)(tile / (uint32_t)tiles_x) * TILE + (int)(lane >> 3);
    if (px < W && py < H) fb[(size_t)py * W + px] = c;
    if (lane == 0u) tstate[tile] = epoch;
}

hipError_t launch_fill_unwritten(hipStream_t st, float4* fb, uint32_t* tstate, uint32_t epoch, int tiles_x, int tiles_y, int W, int H, const float clear[4]) {
    k_fill_unwritten<<<dim3((unsigned)(tiles_x * tiles_y)), dim3(64), 0, st>>>(fb, tstate, epoch, make_float4(clear[0], clear[1], clear[2], clear[3]), tiles_x, W, H);
    return hipGetLastError();
}

__device__ __forceinline__ uint32_t pack8(float4 v) {
    auto q = [](float x) { return (uint32_t)__float2int_rn(fminf(fmaxf(x, 0.0f), 1.0f) * 255.0f); };
    return q(v.x) | (q(v.y) << 8) | (q(v.z) << 16) | (q(v.w) << 24);
}

__global__ __launch_bounds__(256) void k_pack_rgba8(const float4* __restrict__ fb, const uint32_t* __restrict__ tstate, uint32_t epoch, float4 c, uint32_t W, uint32_t npix, uint32_t tiles_x, uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= npix) return;
    const uint32_t x = i % W, y = i / W;
    const bool have = tstate[(y / (uint32_t)TILE) * tiles_x + x / (uint32_t)TILE] == epoch;
    out[i] = pack8(have ? fb[i] : c);
}
// W % 4 == 0: four pixels of a row per thread (they share a tile: TILE % 4 == 0), 64 bytes in, 16 out
__global__ __launch_bounds__(256) void k_pack_rgba8_x4(const float4* __restrict__ fb, const uint32_t* __restrict__ tstate, uint32_t epoch, float4 c, uint32_t W, uint32_t nquads, uint32_t tiles_x, uint4* __restrict__ out) {
    static_assert(TILE % 4 == 0, "a quad of pixels lies in one tile");
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= nquads) return;
    const uint32_t i = q * 4u, x = i % W, y = i / W;
    const bool have = tstate[(y / (uint32_t)TILE) * tiles_x + x / (uint32_t)TILE] == epoch;
    float4 v0 = c, v1 = c, v2 = c, v3 = c;
    if (have) { v0 = fb[i]; v1 = fb[i + 1u]; v2 = fb[i + 2u]; v3 = fb[i + 3u]; }
    out[q] = make_uint4(pack8(v0), pack8(v1), pack8(v2), pack8(v3));
}

hipError_t launch_pack_rgba8(hipStream_t st, const float4* fb, const uint32_t* tstate, uint32_t epoch, const float clear[4], int W, int H, int tiles_x, uint32_t* out) {
    const size_t npix = (size_t)W * H;
    const float4 c = make_float4(clear[0], clear[1], clear[2], clear[3]);
    if (W % 4 == 0 && ((uintptr_t)out & 15u) == 0) k_pack_rgba8_x4<<<dim3((unsigned)((npix / 4 + 255) / 256)), dim3(256), 0, st>>>(fb, tstate, epoch, c, (uint32_t)W, (uint32_t)(npix / 4), (uint32_t)tiles_x, (uint4*)out);
    else k_pack_rgba8<<<dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st>>>(fb, tstate, epoch, c, (uint32_t)W, (uint32_t)npix, (uint32_t)tiles_x, out);
    return hipGetLastError();
}

// Band of a tile-row-sharded frame: output row b belongs to the context's (b / TILE)-th tile row, i.e. tile row rank + (b / TILE) * world.
__global__ __launch_bounds__(256) void k_pack_rgba8_band(const float4* __restrict__ fb, const uint32_t* __restrict__ tstate, uint32_t epoch, float4 c, uint32_t W, uint32_t H, uint32_t tiles_x,
                                                         uint32_t rank, uint32_t world, uint32_t band_rows, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= band_rows * W) return;
    const uint32_t b = i / W, x = i % W;
    const uint32_t ty = rank + (b / (uint32_t)TILE) * world;
    const uint32_t y = ty * (uint32_t)TILE + b % (uint32_t)TILE;
    if (y >= H) return;                                    // cannot happen for a band_rows computed by band_pixel_rows(); kept as a guard
    const bool have = tstate[ty * tiles_x + x / (uint32_t)TILE] == epoch;
    out[i] = pack8(have ? fb[(size_t)y * W + x] : c);
}

hipError_t launch_pack_rgba8_band(hipStream_t st, const float4* fb, const uint32_t* tstate, uint32_t epoch, const float clear[4], int W, int H, int tiles_x, int rank, int world, int band_rows, uint32_t* out) {
    if (band_rows <= 0) return hipSuccess;
    const size_t n = (size_t)band_rows * W;
    k_pack_rgba8_band<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(fb, tstate, epoch, make_float4(clear[0], clear[1], clear[2], clear[3]), (uint32_t)W, (uint32_t)H, (uint32_t)tiles_x, (uint32_t)rank, (uint32_t)world, (uint32_t)band_rows, out);
    return hipGetLastError();
}

template <bool PREMULT_C, bool GENERAL>
__global__ __launch_bounds__(64) void k_composite(const float4* __restrict__ proj, const uint32_t* __restrict__ pair_vals, uint32_t* __restrict__ ranges,
                                                  const uint32_t* __restrict__ total, int tiles_x, int W, int H, uint32_t* __restrict__ tstate, uint32_t epoch, float4 clear,
                                                  float4* __restrict__ fb, int dbg_arg, BlendFn bf) {
#ifdef GS4D_TUNING
    const int dbg = dbg_arg;             // GS4D_COMPOSITE_DBG: tuning builds only (make TUNING=1)
#else
    constexpr int dbg = 0; (void)dbg_arg;
#endif
    __shared__ float4 stage[64 * 3];
    __shared__ uint32_t pmask[64 * 2];                      // per pixel: 64-bit mask of the chunk entries that cover it
    if (total[1]) return;                                   // tile lists overflowed: nothing was emitted, the host re-runs
    uint32_t tile;
    if (!composite_tile(blockIdx.x, tiles_x, (H + TILE - 1) / TILE, tile)) return;        // uniform: padding of the XCD-aware grid
    const uint32_t lane = threadIdx.x;
    const int tx0 = (int)(tile % (uint32_t)tiles_x) * TILE, ty0 = (int)(tile / (uint32_t)tiles_x) * TILE;
    const int px = tx0 + (int)(lane & 7u), py = ty0 + (int)(lane >> 3);
    const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    const uint32_t start = ranges[2 * tile], end = ranges[2 * tile + 1];
    if (lane < 2u && end != 0u) ranges[2 * tile + lane] = 0u;      // leave the table all-zero for the next draw (no memset launch)
    if (start >= end) return;                               // uniform: nothing is drawn on this tile — its pixels, or its being clear, stay as they are
    const bool fb_is_clear = tstate[tile] != epoch;         // uniform: the tile's pixels are not in memory yet

    float T = 1.0f, Cr = 0.0f, Cg = 0.0f, Cb = 0.0f, A = 0.0f;
    if (GENERAL) {
        // a blend function other than the default: no transmittance form — the pixel's value is taken through the list in draw order
        const bool in = px < W && py < H;
        const size_t o = in ? (size_t)py * W + px : 0;
        const float4 d = (fb_is_clear || !in) ? clear : fb[o];
        Cr = d.x; Cg = d.y; Cb = d.z; A = d.w;
        for (uint32_t lo = start; lo < end; lo += 64u) {
            const uint32_t cnt = min(64u, end - lo);
            const uint32_t rec = lane < cnt ? pair_vals[lo + lane] : 0u;         // lane s holds list entry lo+s: s = 0 is drawn first
            composite_chunk<PREMULT_C, true>(proj, rec, cnt, lane, tx0, ty0, fx, fy, stage, pmask, dbg, T, Cr, Cg, Cb, A, bf);
        }
        if (in) fb[o] = make_float4(Cr, Cg, Cb, A);
        if (lane == 0u) tstate[tile] = epoch;
        return;
    }
    for (uint32_t hi = end; hi > start;) {
        const uint32_t cnt = min(64u, hi - start);
        // lane s holds list entry hi-1-s : s = 0 is the LAST (front-most) entry of this chunk
        const uint32_t rec = lane < cnt ? pair_vals[hi - 1u - lane] : 0u;
        composite_chunk<PREMULT_C>(proj, rec, cnt, lane, tx0, ty0, fx, fy, stage, pmask, dbg, T, Cr, Cg, Cb, A);
        hi -= cnt;
        if (__ballot(T > 0.0f) == 0ull) break;              // exact: every remaining contribution is multiplied by T == 0
    }
    if (px < W && py < H) {
        const size_t o = (size_t)py * W + px;
        const float4 d = fb_is_clear ? clear : fb[o];
        fb[o] = make_float4(Cr + T * d.x, Cg + T * d.y, Cb + T * d.z, A + T * d.w);
    }
    if (lane == 0u) tstate[tile] = epoch;
}

hipError_t launch_composite(hipStream_t st, const float4* proj, const uint32_t* pair_vals, uint32_t* ranges, const uint32_t* total, int tiles_x, int tiles_y,
                            int W, int H, int premult_c, uint32_t* tstate, uint32_t epoch, const float clear[4], float4* fb, int blend_src, int blend_dst) {
    const float4 c = make_float4(clear[0], clear[1], clear[2], clear[3]);
    const dim3 grid(composite_grid(tiles_x, tiles_y));
#ifdef GS4D_TUNING
    static const int dbg = getenv("GS4D_COMPOSITE_DBG") ? atoi(getenv("GS4D_COMPOSITE_DBG")) : 0;   // tuning knob: 1 = broadcast only, 2 = splat-parallel only
#else
    const int dbg = 0;
#endif
    const BlendFn bf{ blend_src, blend_dst };
    const bool general = !(blend_src == GS4D_SRC_ALPHA && blend_dst == GS4D_ONE_MINUS_SRC_ALPHA);
    if (general) {
        if (premult_c) k_composite<true, true><<<grid, dim3(64), 0, st>>>(proj, pair_vals, ranges, total, tiles_x, W, H, tstate, epoch, c, fb, dbg, bf);
        else           k_composite<false, true><<<grid, dim3(64), 0, st>>>(proj, pair_vals, ranges, total, tiles_x, W, H, tstate, epoch, c, fb, dbg, bf);
    }
    else if (premult_c) k_composite<true, false><<<grid, dim3(64), 0, st>>>(proj, pair_vals, ranges, total, tiles_x, W, H, tstate, epoch, c, fb, dbg, bf);
    else                k_composite<false, false><<<grid, dim3(64), 0, st>>>(proj, pair_vals, ranges, total, tiles_x, W, H, tstate, epoch, c, fb, dbg, bf);
    return hipGetLastError();
}

} // namespace gs4d
