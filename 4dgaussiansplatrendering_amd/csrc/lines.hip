// lines.hip — overlay lines blended into the RGBA32F image: Renderer::DrawLine / DrawGrid / DrawAxis
// (4DSplatRendering/Renderer.cpp:41-215) with the flat-colour programs Shader/Lines/LineVert.GLSL + LineFrag.GLSL
// (gl_Position = uViewProj * vec4(aPos, 1), fragColor = uColor) and Line2DVert.GLSL (gl_Position = vec4(aPos, 0, 1)).
//
// The fixed-function part is specified here (OpenGL 4.4 core, sections 13.5 and 14.5.2) and checked against the reference's line
// programs run by Mesa llvmpipe (tests/golden/gl_lines_*.npz, tests/test_gpu_gl.py): of the default scenes' overlays (grid, axes, unit
// line, a path strip: 97 121 line pixels at 1080p) 52 pixels differ, each a pair of neighbours swapped where a line passes within a
// sub-pixel step (1/256) of a pixel boundary at a pixel-centre crossing:
//   * a segment is clipped to the view volume -w <= x, y, z <= w (Liang-Barsky in clip space), divided by w and mapped to window
//     coordinates  xw = (x_ndc + 1) W/2,  yw = (y_ndc + 1) H/2  (pixel centres at +0.5, row 0 at the bottom);
//   * non-antialiased rasterisation of width 1 (14.5.2.1, diamond exit) in its common form: an x-major segment (|dx| >= |dy|) yields
//     one fragment in every pixel column whose centre lies in [min x, max x), in the row the line passes through at that centre; a
//     y-major one the same with the roles swapped; a line that runs exactly ON a boundary between two rows (the X axis through the
//     screen centre of the reference's default cameras does) belongs to the lower one — 14.5.2.1 perturbs such end points by
//     (-eps, -eps^2) — llvmpipe agrees (before this rule 6 942 of those 97 121 pixels differed);
//   * wide lines (14.5.2.2): w = round(width) >= 1; the segment is shifted by -(w - 1)/2 in the minor direction and every fragment is
//     replaced by w fragments stacked in the minor direction;
//   * every fragment is blended  dst = src * src.a + dst * (1 - src.a)  on all four channels (Application.cpp:150-154), depth test off.
// All fragments of one call have the same colour, so the result at a pixel depends only on HOW MANY fragments hit it: a first kernel
// counts fragments per pixel (atomics on a per-image counter plane that is all-zero between calls), a second one lets the first
// fragment that arrives at a pixel take the count, apply the blend that many times and clear the counter.
#include "gs4d_internal.h"
#include "composite_common.h"
#include <algorithm>

namespace gs4d {

struct LineSeg { float ax, ay, bx, by; int ok; };        // window coordinates of the clipped segment

__device__ __forceinline__ bool clip_t(float num, float den, float& t0, float& t1) {       // keep num + t * den >= 0
    if (den == 0.0f) return num >= 0.0f;
    const float t = -num / den;
    if (den > 0.0f) { if (t > t1) return false; if (t > t0) t0 = t; }
    else { if (t < t0) return false; if (t < t1) t1 = t; }
    return true;
}

__device__ __forceinline__ LineSeg line_setup(const float* __restrict__ verts, uint32_t s, int dims, int strip, LineParams p) {
    LineSeg g; g.ok = 0; g.ax = g.ay = g.bx = g.by = 0.0f;
    const uint32_t i0 = strip ? s : 2u * s, i1 = i0 + 1u;
    float c0[4], c1[4];
    if (dims == 3) {
        const float* a = verts + 3 * (size_t)i0; const float* b = verts + 3 * (size_t)i1;
        const float* M = p.vp;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            c0[r] = ((M[r] * a[0] + M[4 + r] * a[1]) + M[8 + r] * a[2]) + M[12 + r] * 1.0f;
            c1[r] = ((M[r] * b[0] + M[4 + r] * b[1]) + M[8 + r] * b[2]) + M[12 + r] * 1.0f;
        }
    } else {
        const float* a = verts + 2 * (size_t)i0; const float* b = verts + 2 * (size_t)i1;
        c0[0] = a[0]; c0[1] = a[1]; c0[2] = 0.0f; c0[3] = 1.0f;
        c1[0] = b[0]; c1[1] = b[1]; c1[2] = 0.0f; c1[3] = 1.0f;
    }
    float t0 = 0.0f, t1 = 1.0f;
    bool vis = true;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        vis = vis && clip_t(c0[3] + c0[ax], (c1[3] - c0[3]) + (c1[ax] - c0[ax]), t0, t1);      // w + x >= 0
        vis = vis && clip_t(c0[3] - c0[ax], (c1[3] - c0[3]) - (c1[ax] - c0[ax]), t0, t1);      // w - x >= 0
    }
    if (!vis || !(t0 <= t1)) return g;
    float e0[4], e1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float d = c1[r] - c0[r]; e0[r] = c0[r] + t0 * d; e1[r] = c0[r] + t1 * d; }
    if (!(e0[3] > 0.0f) || !(e1[3] > 0.0f)) return g;
    const float hw = 0.5f * (float)p.W, hh = 0.5f * (float)p.H;
    g.ax = (e0[0] / e0[3] + 1.0f) * hw; g.ay = (e0[1] / e0[3] + 1.0f) * hh;
    g.bx = (e1[0] / e1[3] + 1.0f) * hw; g.by = (e1[1] / e1[3] + 1.0f) * hh;
    g.ok = isfinite(g.ax) && isfinite(g.ay) && isfinite(g.bx) && isfinite(g.by);
    return g;
}

// fragment f of segment g: f = step * wpx + k.  Returns false when there is no such fragment or it falls outside the image.
__device__ __forceinline__ bool line_fragment(const LineSeg& g, uint32_t f, int wpx, int W, int H, int& px, int& py) {
    if (!g.ok) return false;
    const float dx = g.bx - g.ax, dy = g.by - g.ay;
    const bool xmajor = fabsf(dx) >= fabsf(dy);
    float ma = xmajor ? g.ax : g.ay, mb = xmajor ? g.bx : g.by, na = xmajor ? g.ay : g.ax, nb = xmajor ? g.by : g.bx;
    if (ma > mb) { float t = ma; ma = mb; mb = t; t = na; na = nb; nb = t; }
    if (!(mb > ma)) return false;                            // degenerate
    const float first = ceilf(ma - 0.5f), last = ceilf(mb - 0.5f);      // pixel centres i + 0.5 in [ma, mb)
    const uint32_t step = f / (uint32_t)wpx, k = f % (uint32_t)wpx;
    const float i = first + (float)step;
    if (!(i < last)) return false;
    const float t = ((i + 0.5f) - ma) / (mb - ma);
    const float minor = (na + t * (nb - na)) - 0.5f * (float)(wpx - 1);
    const float j = (ceilf(minor) - 1.0f) + (float)k;        // ON a pixel boundary: the pixel below / to the left (see the header)
    const float x = xmajor ? i : j, y = xmajor ? j : i;
    if (!(x >= 0.0f && y >= 0.0f && x < (float)W && y < (float)H)) return false;
    px = (int)x; py = (int)y;
    return true;
}

__global__ __launch_bounds__(256) void k_lines_count(const float* __restrict__ verts, int dims, int strip, LineParams p, int wpx, uint32_t* __restrict__ cnt) {
    const LineSeg g = line_setup(verts, blockIdx.y, dims, strip, p);
    int px, py;
    if (line_fragment(g, blockIdx.x * 256u + threadIdx.x, wpx, p.W, p.H, px, py)) atomicAdd(&cnt[(size_t)py * p.W + px], 1u);
}

__global__ __launch_bounds__(256) void k_lines_blend(const float* __restrict__ verts, int dims, int strip, LineParams p, int wpx, uint32_t* __restrict__ cnt, float4* __restrict__ fb) {
    const LineSeg g = line_setup(verts, blockIdx.y, dims, strip, p);
    int px, py;
    if (!line_fragment(g, blockIdx.x * 256u + threadIdx.x, wpx, p.W, p.H, px, py)) return;
    const size_t o = (size_t)py * p.W + px;
    const uint32_t k = atomicExch(&cnt[o], 0u);              // the first fragment to arrive blends for all of them
    if (!k) return;
    float4 d = fb[o];
    const float a = p.rgba[3], om = __fsub_rn(1.0f, a);
    const float sr = __fmul_rn(p.rgba[0], a), sg = __fmul_rn(p.rgba[1], a), sb = __fmul_rn(p.rgba[2], a), sa = __fmul_rn(a, a);
    if (p.blend_src == GS4D_SRC_ALPHA && p.blend_dst == GS4D_ONE_MINUS_SRC_ALPHA) {
        for (uint32_t q = 0; q < k; ++q) {
            d.x = __fadd_rn(sr, __fmul_rn(d.x, om)); d.y = __fadd_rn(sg, __fmul_rn(d.y, om));
            d.z = __fadd_rn(sb, __fmul_rn(d.z, om)); d.w = __fadd_rn(sa, __fmul_rn(d.w, om));
        }
    } else {
        const BlendFn bf{ p.blend_src, p.blend_dst };       // any other glBlendFunc: the same fragment, k times over
        for (uint32_t q = 0; q < k; ++q) blend_general(bf, p.rgba[0], p.rgba[1], p.rgba[2], a, d.x, d.y, d.z, d.w);
    }
    fb[o] = d;
}

hipError_t launch_lines(hipStream_t st, const float* verts_dev, size_t nverts, int dims, int strip, const LineParams& p, float width, uint32_t* cnt, float4* fb) {
    const size_t nseg = strip ? (nverts >= 2 ? nverts - 1 : 0) : nverts / 2;
    if (nseg == 0) return hipSuccess;
    int wpx = (int)floorf(width + 0.5f); if (!(wpx >= 1)) wpx = 1; if (wpx > 64) wpx = 64;
    const unsigned frags = (unsigned)((p.W > p.H ? p.W : p.H) + 1) * (unsigned)wpx;
    for (size_t s0 = 0; s0 < nseg; s0 += 65535) {            // grid.y is 16-bit
        const unsigned ns = (unsigned)std::min<size_t>(65535, nseg - s0);
        const float* v = verts_dev + (strip ? s0 : 2 * s0) * (size_t)dims;
        const dim3 grid((frags + 255) / 256, ns);
        k_lines_count<<<grid, dim3(256), 0, st>>>(v, dims, strip, p, wpx, cnt);
        k_lines_blend<<<grid, dim3(256), 0, st>>>(v, dims, strip, p, wpx, cnt, fb);
    }
    return hipGetLastError();
}

} // namespace gs4d
