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
#include "gs4d_internal.h"

namespace gs4d {

__global__ __launch_bounds__(256) void k_fill(float4* __restrict__ fb, uint32_t npix, float4 c) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < npix) fb[i] = c;
}

hipError_t launch_fill(hipStream_t st, float4* fb, size_t npix, const float clear[4]) {
    k_fill<<<dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st>>>(fb, (uint32_t)npix, make_float4(clear[0], clear[1], clear[2], clear[3]));
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_pack_rgba8(const float4* __restrict__ fb, uint32_t npix, uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= npix) return;
    float4 v = fb[i];
    auto q = [](float x) { return (uint32_t)__float2int_rn(fminf(fmaxf(x, 0.0f), 1.0f) * 255.0f); };
    out[i] = q(v.x) | (q(v.y) << 8) | (q(v.z) << 16) | (q(v.w) << 24);
}

hipError_t launch_pack_rgba8(hipStream_t st, const float4* fb, size_t npix, uint32_t* out) {
    k_pack_rgba8<<<dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st>>>(fb, (uint32_t)npix, out);
    return hipGetLastError();
}

template <bool PREMULT_C>
__global__ __launch_bounds__(64) void k_composite(const float4* __restrict__ proj, const uint32_t* __restrict__ pair_vals, const uint32_t* __restrict__ ranges,
                                                  const uint32_t* __restrict__ total, int tiles_x, int W, int H, int fb_is_clear, float4 clear,
                                                  float4* __restrict__ fb) {
    __shared__ float4 stage[64 * 3];
    if (total[1]) return;                                   // tile lists overflowed: nothing was emitted, the host re-runs
    const uint32_t tile = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const int px = (int)(tile % (uint32_t)tiles_x) * TILE + (int)(lane & 7u);
    const int py = (int)(tile / (uint32_t)tiles_x) * TILE + (int)(lane >> 3);
    const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    const uint32_t start = ranges[2 * tile], end = ranges[2 * tile + 1];

    float T = 1.0f, Cr = 0.0f, Cg = 0.0f, Cb = 0.0f, A = 0.0f;
    for (uint32_t hi = end; hi > start;) {
        const uint32_t cnt = min(64u, hi - start);
        // lane s holds list entry hi-1-s : s = 0 is the LAST (front-most) entry of this chunk
        if (lane < cnt) {
            const uint32_t rec = pair_vals[hi - 1u - lane];
            const float4* r = proj + (size_t)rec * 4;
            stage[lane * 3 + 0] = r[0];
            stage[lane * 3 + 1] = r[1];
            stage[lane * 3 + 2] = r[2];
        }
        __syncthreads();
        for (uint32_t s = 0; s < cnt; ++s) {
            const float4 a = stage[s * 3 + 0];              // cx, cy, a0x, a0y      (uniform address: LDS broadcast)
            const float4 b = stage[s * 3 + 1];              // a1x, a1y, alpha, r
            const float dx = __fsub_rn(fx, a.x), dy = __fsub_rn(fy, a.y);
            const float u = __fmaf_rn(a.z, dx, __fmul_rn(a.w, dy));
            const float v = __fmaf_rn(b.x, dx, __fmul_rn(b.y, dy));
            const bool cov = fabsf(u) <= 0.5f && fabsf(v) <= 0.5f;
            if (__ballot(cov) == 0ull) continue;
            const float4 c = stage[s * 3 + 2];              // g, b, -, -
            const float q = u * u + v * v;
            const float cg = __expf(-32.0f * q);
            if (cov && cg >= 0.0001f) {
                const float al = b.z * cg;
                const float w = T * al;
                float r_ = b.w, g_ = c.x, b_ = c.y;
                if (PREMULT_C) { r_ *= cg; g_ *= cg; b_ *= cg; }
                Cr += w * r_; Cg += w * g_; Cb += w * b_; A += w * al;
                T *= (1.0f - al);
            }
        }
        __syncthreads();
        hi -= cnt;
        if (__ballot(T > 0.0f) == 0ull) break;              // exact: every remaining contribution is multiplied by T == 0
    }
    if (px < W && py < H) {
        const size_t o = (size_t)py * W + px;
        const float4 d = fb_is_clear ? clear : fb[o];
        fb[o] = make_float4(Cr + T * d.x, Cg + T * d.y, Cb + T * d.z, A + T * d.w);
    }
}

hipError_t launch_composite(hipStream_t st, const float4* proj, const uint32_t* pair_vals, const uint32_t* ranges, const uint32_t* total, int tiles_x, int tiles_y,
                            int W, int H, int premult_c, int fb_is_clear, const float clear[4], float4* fb) {
    const float4 c = make_float4(clear[0], clear[1], clear[2], clear[3]);
    const dim3 grid((unsigned)(tiles_x * tiles_y));
    if (premult_c) k_composite<true><<<grid, dim3(64), 0, st>>>(proj, pair_vals, ranges, total, tiles_x, W, H, fb_is_clear, c, fb);
    else           k_composite<false><<<grid, dim3(64), 0, st>>>(proj, pair_vals, ranges, total, tiles_x, W, H, fb_is_clear, c, fb);
    return hipGetLastError();
}

} // namespace gs4d
