// composite_common.h — what the two compositing kernels (composite.hip: ordered lists; composite2.hip: lists ordered in LDS) share.
#pragma once
#include "gs4d_internal.h"

namespace gs4d {

// One chunk of <= 64 list entries is processed in one of two ways (wave-uniform choice):
//  * splat-parallel (small footprints, the 10^6..10^7-splat cube configs: a quad covers ~3 of the tile's 64 pixels):
//      phase A, lane = entry: evaluates the coverage rule only for the pixels of its own bounding box inside the tile and ORs its
//               bit into the per-pixel hit masks in LDS;
//      phase B, lane = pixel: walks the set bits of its own mask front to back, fetches that entry from LDS and blends.
//      Work is proportional to the number of covered pixels instead of 64 x entries.
//  * pixel-parallel broadcast (large footprints, e.g. the teapot scenes): every lane tests every entry of the chunk.
// Entries whose box inside the tile is wider or taller than SMALL_SIDE pixels are always tested pixel-parallel.
// (Round 2 walked each small box cell by cell up to 16 cells, x fastest, with a row wrap inside the loop: ~25 wave instructions per cell for
// three nested divergent regions.  A fixed SMALL_SIDE x SMALL_SIDE window walked by two UNIFORM loops — as many rows and columns as the
// widest box of the chunk has — needs no wrap and keeps everything of a row in registers: ~12 per cell.)
constexpr int SMALL_SIDE = 4;

// Workgroup -> tile.  Workgroups are dealt to the 8 XCDs round-robin (workgroup i runs on XCD i % 8) and each XCD has an L2 of its own;
// a splat that touches several tiles (1.38 entries per splat in the cube configs) is gathered once per tile.  With tile = workgroup id
// the tiles that share a record sit on different XCDs and every gather goes to memory.  Here the image is cut into blocks of 4 x 4
// tiles, block b belongs to XCD b % 8 (small blocks: the busy middle of the image is spread over all XCDs), and the workgroups of an
// XCD walk its blocks one after another: neighbours in the image are neighbours in time on one L2.
constexpr uint32_t XCDS = 8, TBLOCK = 4;
__host__ __device__ __forceinline__ uint32_t composite_grid(int tiles_x, int tiles_y) {
    const uint32_t nblocks = (uint32_t)((tiles_x + TBLOCK - 1) / TBLOCK) * (uint32_t)((tiles_y + TBLOCK - 1) / TBLOCK);
    return (nblocks + XCDS - 1) / XCDS * XCDS * (TBLOCK * TBLOCK);
}
__device__ __forceinline__ bool composite_tile(uint32_t wg, int tiles_x, int tiles_y, uint32_t& tile) {
    const uint32_t x = wg % XCDS, j = wg / XCDS;
    const uint32_t block = x + XCDS * (j / (TBLOCK * TBLOCK)), t = j % (TBLOCK * TBLOCK);
    const uint32_t bxn = (uint32_t)(tiles_x + TBLOCK - 1) / TBLOCK;
    const uint32_t tx = (block % bxn) * TBLOCK + (t % TBLOCK), ty = (block / bxn) * TBLOCK + (t / TBLOCK);
    tile = ty * (uint32_t)tiles_x + tx;
    return tx < (uint32_t)tiles_x && ty < (uint32_t)tiles_y;
}
// The same walk over a BOX of tiles (x0, y0, w, h — in tiles) of an image that is tiles_x tiles wide: the grid is composite_grid(w, h).
struct TileBox { uint32_t x0, y0, w, h; };
static_assert(TBLOCK == BOX_BLOCK, "a launch box is whole blocks of the compositor's walk");
// box: TileLists::box (blocks, inclusive; BOX_NONE = everything), clipped to the image
__host__ __device__ __forceinline__ TileBox tile_box(uint32_t box, int tiles_x, int tiles_y) {
    if (box == BOX_NONE) return TileBox{ 0u, 0u, (uint32_t)tiles_x, (uint32_t)tiles_y };
    const uint32_t x0 = (box & 255u) * BOX_BLOCK, y0 = ((box >> 8) & 255u) * BOX_BLOCK;
    const uint32_t x1 = (((box >> 16) & 255u) + 1u) * BOX_BLOCK, y1 = ((box >> 24) + 1u) * BOX_BLOCK;
    const uint32_t cx1 = x1 < (uint32_t)tiles_x ? x1 : (uint32_t)tiles_x, cy1 = y1 < (uint32_t)tiles_y ? y1 : (uint32_t)tiles_y;
    return TileBox{ x0, y0, cx1 > x0 ? cx1 - x0 : 0u, cy1 > y0 ? cy1 - y0 : 0u };
}
__device__ __forceinline__ bool composite_tile(uint32_t wg, int tiles_x, TileBox box, uint32_t& tile) {
    uint32_t t;
    const bool real = composite_tile(wg, (int)box.w, (int)box.h, t);
    tile = (box.y0 + t / box.w) * (uint32_t)tiles_x + box.x0 + t % box.w;
    return real;
}

// glBlendFunc factors (GL enum values; the set the reference's menu offers, DebugMenus.h:41-59).  The reference never calls glBlendColor:
// the blend colour stays (0, 0, 0, 0), so CONSTANT_* = 0 and ONE_MINUS_CONSTANT_* = 1.
struct BlendFn { int sf, df; };
__device__ __forceinline__ float blend_factor(int f, float sc, float sa, float dc, float da) {
    switch (f) {
    case GS4D_ONE: case GS4D_ONE_MINUS_CONSTANT_COLOR: case GS4D_ONE_MINUS_CONSTANT_ALPHA: return 1.0f;
    case GS4D_SRC_COLOR: return sc;
    case GS4D_ONE_MINUS_SRC_COLOR: return 1.0f - sc;
    case GS4D_SRC_ALPHA: return sa;
    case GS4D_ONE_MINUS_SRC_ALPHA: return 1.0f - sa;
    case GS4D_DST_ALPHA: return da;
    case GS4D_ONE_MINUS_DST_ALPHA: return 1.0f - da;
    case GS4D_DST_COLOR: return dc;
    case GS4D_ONE_MINUS_DST_COLOR: return 1.0f - dc;
    default: return 0.0f;                                  // GS4D_ZERO, GS4D_CONSTANT_COLOR, GS4D_CONSTANT_ALPHA
    }
}
// dst = clamp(src * S + dst * D) on all four channels (OpenGL 4.4, 17.3.8: FUNC_ADD, fixed-point framebuffer; src is already clamped)
__device__ __forceinline__ void blend_general(BlendFn bf, float sr, float sg, float sb, float sa, float& dr, float& dg, float& db, float& da) {
    const float r = __saturatef(__fadd_rn(__fmul_rn(sr, blend_factor(bf.sf, sr, sa, dr, da)), __fmul_rn(dr, blend_factor(bf.df, sr, sa, dr, da))));
    const float g = __saturatef(__fadd_rn(__fmul_rn(sg, blend_factor(bf.sf, sg, sa, dg, da)), __fmul_rn(dg, blend_factor(bf.df, sg, sa, dg, da))));
    const float b = __saturatef(__fadd_rn(__fmul_rn(sb, blend_factor(bf.sf, sb, sa, db, da)), __fmul_rn(db, blend_factor(bf.df, sb, sa, db, da))));
    const float a = __saturatef(__fadd_rn(__fmul_rn(sa, blend_factor(bf.sf, sa, sa, da, da)), __fmul_rn(da, blend_factor(bf.df, sa, sa, da, da))));
    dr = r; dg = g; db = b; da = a;
}

// GENERAL = false: the default function (SRC_ALPHA, ONE_MINUS_SRC_ALPHA), accumulated front to back: colour C, transmittance T.
// GENERAL = true : any function of the menu, applied fragment by fragment in draw order to the pixel's value (Cr, Cg, Cb, A); T unused.
// exp(-32 q) as ONE multiply and the hardware's 2^x (v_exp_f32); [0, 1] clamps as one v_med3_f32 (finite arguments only reach them).
__device__ __forceinline__ float gauss_weight(float u, float v) { return __builtin_amdgcn_exp2f((u * u + v * v) * -46.16624130844683f); }      // -32 * log2(e)
__device__ __forceinline__ float clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }

template <bool PREMULT_C, bool GENERAL = false>
__device__ __forceinline__ void blend_fragment(float u, float v, float alpha, float r_, float g_, float b_, float& T, float& Cr, float& Cg, float& Cb, float& A, BlendFn bf = BlendFn{ 0, 0 }) {
    const float cg = gauss_weight(u, v);
    if (GENERAL) {
        if (cg >= 0.0001f) {                               // Splat4DFragShader.GLSL:30 discard
            const float al = clamp01(alpha * cg);
            if (PREMULT_C) { r_ = clamp01(r_ * cg); g_ = clamp01(g_ * cg); b_ = clamp01(b_ * cg); }
            blend_general(bf, r_, g_, b_, al, Cr, Cg, Cb, A);
        }
        return;
    }
    // The reference blends into a fixed-point (RGBA8) framebuffer: the GL clamps the fragment's colour and alpha to [0, 1] before the
    // blend (OpenGL 4.4, 17.3.8).  Colours that are not premultiplied were clamped once per record by the projection kernel.
    // No branch: a discarded fragment (Splat4DFragShader.GLSL:30) — and a lane that has no fragment at all, which passes alpha = 0 — blends
    // with al = 0, which leaves C and T as they are, exactly (C + 0 * c, T * 1).  The compositor is bound by instruction issue; a divergent
    // region costs four wave instructions whether or not a lane takes it.
    const float al = cg >= 0.0001f ? clamp01(alpha * cg) : 0.0f;
    const float w = T * al;
    if (PREMULT_C) { r_ = clamp01(r_ * cg); g_ = clamp01(g_ * cg); b_ = clamp01(b_ * cg); }   // Splat3DFragShaderFull.GLSL:22
    Cr += w * r_; Cg += w * g_; Cb += w * b_; A += w * al;
    T *= (1.0f - al);
}

// One chunk of the tile's list, front to back: lane s < cnt carries record `rec` of list entry (end of chunk - 1 - s), so s = 0 is the
// front-most entry.  stage: 64 x 3 float4, pmask: 64 x 2 words (per pixel: 64-bit mask of the chunk entries that cover it).
// GENERAL: the chunk is walked in DRAW order instead — lane s carries list entry (start of chunk + s).
template <bool PREMULT_C, bool GENERAL = false>
__device__ __forceinline__ void composite_chunk(const float4* __restrict__ proj, uint32_t rec, uint32_t cnt, uint32_t lane, int tx0, int ty0, float fx, float fy,
                                                float4* stage, uint32_t* pmask, int dbg, float& T, float& Cr, float& Cg, float& Cb, float& A, BlendFn bf = BlendFn{ 0, 0 }) {
    // lane s holds list entry hi-1-s : s = 0 is the LAST (front-most) entry of this chunk
    int lx0 = 0, ly0 = 0, bw = 0, bh = 0;
    float4 ra = make_float4(0, 0, 0, 0), rb = ra;
    if (lane < cnt) {
        const float4* r = proj + (size_t)rec * 4;
        ra = r[0]; rb = r[1];
        const float4 rc = r[2];
        stage[lane * 3 + 0] = ra;
        stage[lane * 3 + 1] = rb;
        stage[lane * 3 + 2] = rc;
        const uint32_t r0 = __float_as_uint(rc.z), r1 = __float_as_uint(rc.w);
        lx0 = max((int)(r0 & 0xFFFFu) - tx0, 0); ly0 = max((int)(r0 >> 16) - ty0, 0);
        const int lx1 = min((int)(r1 & 0xFFFFu) - tx0, TILE - 1), ly1 = min((int)(r1 >> 16) - ty0, TILE - 1);
        bw = lx1 - lx0 + 1;
        bh = ly1 - ly0 + 1;
        if (bw <= 0 || bh <= 0) { bw = 0; bh = 0; }            // no pixel of this tile (cannot happen for a list entry; kept exact)
    }
    const bool big = bw > SMALL_SIDE || bh > SMALL_SIDE;
    const uint64_t bigmask = __ballot(big);
    pmask[lane * 2] = 0u; pmask[lane * 2 + 1] = 0u;
    __syncthreads();
    if (dbg != 2 && (dbg == 1 || (uint32_t)__popcll(bigmask) * 2u > cnt)) {
        // ---- pixel-parallel broadcast over the whole chunk ----
        for (uint32_t s = 0; s < cnt; ++s) {
            const float4 a = stage[s * 3 + 0];          // cx, cy, a0x, a1x      (uniform address: LDS broadcast)
            const float4 b = stage[s * 3 + 1];          // a0y, a1y, r, g
            const float dx = __fsub_rn(fx, a.x), dy = __fsub_rn(fy, a.y);
            const float u = __fmaf_rn(a.z, dx, __fmul_rn(b.x, dy));
            const float v = __fmaf_rn(a.w, dx, __fmul_rn(b.y, dy));
            const bool cov = fabsf(u) <= 0.5f && fabsf(v) <= 0.5f;
            if (__ballot(cov) == 0ull) continue;
            const float4 c = stage[s * 3 + 2];          // b, alpha, -, -
            if (cov) blend_fragment<PREMULT_C, GENERAL>(u, v, c.y, b.z, b.w, c.x, T, Cr, Cg, Cb, A, bf);
        }
    } else {
        // ---- phase A (lane = entry): mark the covered pixels of small footprints ----
        // The box is at most SMALL_SIDE x SMALL_SIDE here.  Two uniform loops, over as many rows and columns as the tallest / widest small box
        // of the chunk: what does not change along a row stays in registers, and (float)(pixel) + 0.5f is formed afresh for every cell —
        // the same bits the pixel-parallel path and the checker use (pixel centres are integers + 0.5 below 2^16: exact).
        const int sw = big ? 0 : bw, sh = big ? 0 : bh;
        // the widest / tallest small box of the chunk, by ballots (scalar work: the vector pipe is the bottleneck here)
        int mw = 0, mh = 0;
#pragma unroll
        for (int k = 0; k < SMALL_SIDE; ++k) { if (__ballot(sw > k) != 0ull) mw = k + 1; if (__ballot(sh > k) != 0ull) mh = k + 1; }
        const float fx0 = (float)(tx0 + lx0) + 0.5f, fy0 = (float)(ty0 + ly0) + 0.5f;
        uint32_t* const pm0 = pmask + (ly0 * TILE + lx0) * 2 + (int)(lane >> 5);
        const uint32_t bit = 1u << (lane & 31u);
        float dxs[SMALL_SIDE];
#pragma unroll
        for (int xx = 0; xx < SMALL_SIDE; ++xx) dxs[xx] = __fsub_rn(fx0 + (float)xx, ra.x);
#pragma unroll
        for (int yy = 0; yy < SMALL_SIDE; ++yy) {
            if (yy >= mh) break;                            // uniform
            const float dyq = __fsub_rn(fy0 + (float)yy, ra.y);
            const float t0 = __fmul_rn(rb.x, dyq), t1 = __fmul_rn(rb.y, dyq);
            const bool rowin = yy < sh;
#pragma unroll
            for (int xx = 0; xx < SMALL_SIDE; ++xx) {
                if (xx >= mw) break;                        // uniform
                const float u = __fmaf_rn(ra.z, dxs[xx], t0);
                const float v = __fmaf_rn(ra.w, dxs[xx], t1);
                if (rowin && xx < sw && fabsf(u) <= 0.5f && fabsf(v) <= 0.5f) atomicOr(pm0 + (yy * TILE + xx) * 2, bit);
            }
        }
        // ---- large footprints of this chunk: every pixel tests them itself ----
        uint64_t mine = 0ull;
        for (uint64_t bm = bigmask; bm; bm &= bm - 1ull) {
            const int e = __ffsll((long long)bm) - 1;
            const float4 a = stage[e * 3 + 0];
            const float4 b = stage[e * 3 + 1];
            const float dx = __fsub_rn(fx, a.x), dy = __fsub_rn(fy, a.y);
            const float u = __fmaf_rn(a.z, dx, __fmul_rn(b.x, dy));
            const float v = __fmaf_rn(a.w, dx, __fmul_rn(b.y, dy));
            if (fabsf(u) <= 0.5f && fabsf(v) <= 0.5f) mine |= 1ull << e;
        }
        __syncthreads();
        // ---- phase B (lane = pixel): blend the entries that hit this pixel, front to back ----
        // One straight body per round (GENERAL keeps its branch: the blend functions of the menu are not the hot path): a pixel that has run
        // out of hits reads entry 0 and blends it with alpha 0 — nothing changes, and no divergent region is opened.
        uint64_t m = mine | (uint64_t)pmask[lane * 2] | ((uint64_t)pmask[lane * 2 + 1] << 32);
        while (__ballot(m != 0ull) != 0ull) {
            const bool on = m != 0ull;
            const int e = on ? __ffsll((long long)m) - 1 : 0;
            m &= m - 1ull;                                  // 0 stays 0
            const float4 a = stage[e * 3 + 0];
            const float4 b = stage[e * 3 + 1];
            const float4 c = stage[e * 3 + 2];
            const float dx = __fsub_rn(fx, a.x), dy = __fsub_rn(fy, a.y);
            const float u = __fmaf_rn(a.z, dx, __fmul_rn(b.x, dy));
            const float v = __fmaf_rn(a.w, dx, __fmul_rn(b.y, dy));
            if (GENERAL) { if (on) blend_fragment<PREMULT_C, true>(u, v, c.y, b.z, b.w, c.x, T, Cr, Cg, Cb, A, bf); }
            else blend_fragment<PREMULT_C, false>(u, v, on ? c.y : 0.0f, b.z, b.w, c.x, T, Cr, Cg, Cb, A);
        }
    }
    __syncthreads();
}

} // namespace gs4d
