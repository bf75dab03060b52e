// preprocess.hip — per-splat vertex stage: 4D->3D time conditioning, EWA projection, 2x2 eigen-decomposition,
// window-space quad set-up.  One thread per splat, coalesced float4 loads from SoA planes, one 64-B projected
// record out.  Replaces the vertex shaders the reference runs 4x per splat (once per quad corner):
//   Shader/Splats4D/Splat4DVertexShaderInstanced.GLSL:81-150 (and ...Mod.GLSL:61-130)
//   Shader/Splats3D/Splat3DVertexShaderFull.GLSL:43-98
//   Shader/Splats2D/Splat2DVSI.GLSL:59-94
//
// Compiled with -ffp-contract=off and hipcc's default IEEE divide/sqrt: everything that decides pixel coverage
// (cx, cy, a0, a1) is bit-identical to the CPU checker, which evaluates the same expressions in the same order.
#include "gs4d_internal.h"
#include <algorithm>
#include <cstdlib>

namespace gs4d {

// AoS 96-B SplatData (Scenes.h:22-37) -> planes.  Three layouts (SOA_*, gs4d_internal.h):
//   full      6 planes of float4: pos, col, sig[0], sig[1], sig[2], sig[3]                                                96 B / record
//   sym       pos, col, U = (s00, s01, s02, s11), V = sig[3] = (s03, s13, s23, s33) as float4 planes, W = (s12, s22) as a float2 plane   72 B / record
//   static3d  (px py pz s00), col, (s01 s02 s10 s11), (s12 s20 s21 s22): the 16 values of a STATIC 3D splat that differ from record to record   64 B / record
// sym holds a SYMMETRIC sig without its mirrored half — every covariance the reference's 4D scenes build is one (Splat.h:127, 141-154).
// static3d is for a set of 3D splats kept in the reference's 4D record: mu_t, sig's time column sig[c][3] and its time row sig[3][*] are
// the same eight values in every record (0, 0, 0, 0, (0, 0, 0, 1) for a 3D covariance: Scenes.h ObjectDisplay, the cube configs) and travel
// as kernel arguments.  Both are exact: the repack kernel compares what the layout assumes as bit patterns for every record and raises
// bbox[15] if any record breaks it; the host then repacks in the next layout down.  A third (a quarter) of the projection's read traffic is gone.
// float <-> unsigned with the same order, for atomicMin/atomicMax on floats
__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

// Also reduces the bounding box of what the sort key depends on — position, mu_t and the velocity column sig[3].xyz
// (Scenes.h:28-36) — into bbox[0..6] = min, bbox[7..13] = max (order-preserving uint form), bbox[14] = non-finite input seen.
struct SoaConsts { float v[8]; };
template <int LAYOUT>
__global__ __launch_bounds__(256) void k_soa_repack(const float4* __restrict__ aos, uint32_t n, float4* __restrict__ soa, uint32_t* __restrict__ bbox, SoaConsts cs) {
    constexpr bool COMPACT = LAYOUT != SOA_FULL;
    // one wave moves 64 records = 384 float4, read fully coalesced, written as 6 x 64 contiguous float4
    __shared__ float4 stage[4][384];
    __shared__ uint32_t red[4][16];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    // bounding box: accumulated in registers over the workgroup's whole share of the records, reduced once per workgroup, and only
    // then merged into the 15 global words (they share a cache line: one atomic per wave and chunk made the kernel 30x slower)
    uint32_t lo[7], hi[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) { lo[k] = 0xFFFFFFFFu; hi[k] = 0u; }
    bool bad = false, asym = false;
    for (uint32_t chunk = blockIdx.x; chunk * 256u < n; chunk += gridDim.x) {           // uniform trip count per workgroup
        const uint32_t rec0 = (chunk * 4u + w) * 64u;
        const uint32_t nrec = rec0 < n ? min(64u, n - rec0) : 0u;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            uint32_t f = j * 64u + lane;
            if (f < nrec * 6u) stage[w][f] = aos[(size_t)rec0 * 6u + f];
        }
        __builtin_amdgcn_wave_barrier();               // stage[w] is private to the wave
        if (lane < nrec) {
            auto same = [](float a, float b) { return __float_as_uint(a) == __float_as_uint(b); };
            if (LAYOUT == SOA_STATIC3D) {
                const float4 ps = stage[w][lane * 6u + 0], c0 = stage[w][lane * 6u + 2], c1 = stage[w][lane * 6u + 3], c2 = stage[w][lane * 6u + 4], c3 = stage[w][lane * 6u + 5];
                soa[rec0 + lane] = make_float4(ps.x, ps.y, ps.z, c0.x);
                soa[(size_t)n + rec0 + lane] = stage[w][lane * 6u + 1];
                soa[(size_t)2 * n + rec0 + lane] = make_float4(c0.y, c0.z, c1.x, c1.y);
                soa[(size_t)3 * n + rec0 + lane] = make_float4(c1.z, c2.x, c2.y, c2.z);
                if (!(same(ps.w, cs.v[0]) && same(c0.w, cs.v[1]) && same(c1.w, cs.v[2]) && same(c2.w, cs.v[3]) && same(c3.x, cs.v[4]) && same(c3.y, cs.v[5]) && same(c3.z, cs.v[6]) && same(c3.w, cs.v[7]))) asym = true;
            } else if (LAYOUT == SOA_SYM) {
                const float4 c0 = stage[w][lane * 6u + 2], c1 = stage[w][lane * 6u + 3], c2 = stage[w][lane * 6u + 4], c3 = stage[w][lane * 6u + 5];
                soa[rec0 + lane] = stage[w][lane * 6u + 0];
                soa[(size_t)n + rec0 + lane] = stage[w][lane * 6u + 1];
                soa[(size_t)2 * n + rec0 + lane] = make_float4(c0.x, c0.y, c0.z, c1.y);
                soa[(size_t)3 * n + rec0 + lane] = c3;
                reinterpret_cast<float2*>(soa + (size_t)4 * n)[rec0 + lane] = make_float2(c1.z, c2.z);
                // sig[c][r] == sig[r][c]: c0.y = sig[0][1] vs c1.x = sig[1][0], ...
                if (!(same(c0.y, c1.x) && same(c0.z, c2.x) && same(c0.w, c3.x) && same(c1.z, c2.y) && same(c1.w, c3.y) && same(c2.w, c3.z))) asym = true;
            } else {
#pragma unroll
                for (int p = 0; p < 6; ++p) soa[(size_t)p * n + rec0 + lane] = stage[w][lane * 6u + p];
            }
            const float4 ps = stage[w][lane * 6u + 0], s3 = stage[w][lane * 6u + 5];
            const float q[7] = { ps.x, ps.y, ps.z, ps.w, s3.x, s3.y, s3.z };
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                if (isfinite(q[k])) { const uint32_t o = f2ord(q[k]); lo[k] = min(lo[k], o); hi[k] = max(hi[k], o); }
                else bad = true;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lo[k] = min(lo[k], (uint32_t)__shfl_xor(lo[k], off, 64)); hi[k] = max(hi[k], (uint32_t)__shfl_xor(hi[k], off, 64)); }
    }
    const bool anybad = __ballot(bad) != 0ull;
    if (COMPACT && __ballot(asym) != 0ull && lane == 0) atomicOr(&bbox[15], 1u);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) { red[w][k] = lo[k]; red[w][7 + k] = hi[k]; }
        red[w][14] = anybad ? 1u : 0u;
    }
    __syncthreads();
    if (threadIdx.x < 15u) {
        const uint32_t k = threadIdx.x;
        uint32_t v = red[0][k];
        for (int ww = 1; ww < 4; ++ww) v = k < 7u ? min(v, red[ww][k]) : k < 14u ? max(v, red[ww][k]) : (v | red[ww][k]);
        if (k < 7u) atomicMin(&bbox[k], v); else if (k < 14u) atomicMax(&bbox[k], v); else if (v) atomicOr(&bbox[14], 1u);
    }
}

hipError_t launch_soa_repack(hipStream_t st, const float* aos96, size_t n, float4* soa, uint32_t* bbox, const SoaInfo& info) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
    SoaConsts cs; for (int k = 0; k < 8; ++k) cs.v[k] = info.consts[k];
    if (info.layout == SOA_STATIC3D) k_soa_repack<SOA_STATIC3D><<<dim3(blocks), dim3(256), 0, st>>>((const float4*)aos96, (uint32_t)n, soa, bbox, cs);
    else if (info.layout == SOA_SYM) k_soa_repack<SOA_SYM><<<dim3(blocks), dim3(256), 0, st>>>((const float4*)aos96, (uint32_t)n, soa, bbox, cs);
    else k_soa_repack<SOA_FULL><<<dim3(blocks), dim3(256), 0, st>>>((const float4*)aos96, (uint32_t)n, soa, bbox, cs);
    return hipGetLastError();
}

// ---- shared device math (same expression order as the checker) ----------------------------------
struct PU { float V[16]; float P[16]; float time, min_opacity; int W, H; };

__device__ __forceinline__ float maxf_glsl(float a, float b) { return a >= b ? a : b; }
__device__ __forceinline__ void normalize2(float& x, float& y) { float tx = x * x, ty = y * y; float s = 1.0f / sqrtf(tx + ty); x = x * s; y = y * s; }

struct Quad { float e0x, e0y, e1x, e1y, s0, s1; };   // R columns and the S diagonal used in R*S

// GetEigenValues2x2 / GetEigenVectors2x2 / normalize (…Instanced.GLSL:59-78, 132-138)
__device__ __forceinline__ void eigen2(float u00, float u01, float u10, float u11, bool guard, float& lx, float& ly, Quad& q) {
    float m = (u00 + u11) * 0.5f;
    float p = (u00 * u11) - (u01 * u10);
    float d = sqrtf((m * m) - p);
    lx = maxf_glsl(m - d, 0.000001f);
    ly = maxf_glsl(m + d, 0.000001f);
    float ax, ay, bx, by;
    if (guard && u01 == 0.0f) { ax = 1.0f; ay = 0.0f; bx = 0.0f; by = 1.0f; }
    else {
        ax = u01; ay = lx - u00;
        normalize2(ax, ay);
        bx = ay; by = -ax;
        normalize2(ax, ay); normalize2(bx, by);
    }
    normalize2(ax, ay); normalize2(bx, by);
    q.e0x = ax; q.e0y = ay; q.e1x = bx; q.e1y = by;
}

__device__ __forceinline__ bool fin(float x) { return isfinite(x); }

// Window-space set-up + record store.  ncx,ncy = NDC centre; kx,ky = NDC scale of the quad offset.
__device__ __forceinline__ uint2 emit(const PreOut& out, uint32_t i, bool valid, const Quad& q, float ncx, float ncy, float kx, float ky,
                                      int W, int H, float r, float g, float b, float alpha, bool clamp_rgb) {
    float cx = 0, cy = 0, a0x = 0, a0y = 0, a1x = 0, a1y = 0, hx = 0, hy = 0;
    uint32_t rect0 = 1u, rect1 = 0u;           // empty
    if (valid) {
        float hw = (float)W * 0.5f, hh = (float)H * 0.5f;
        float sx = kx * hw, sy = ky * hh;
        cx = fmaf(ncx, hw, hw);
        cy = fmaf(ncy, hh, hh);
        float r0 = 1.0f / q.s0, r1 = 1.0f / q.s1;
        a0x = (q.e0x * r0) / sx; a0y = (q.e0y * r0) / sy;
        a1x = (q.e1x * r1) / sx; a1y = (q.e1y * r1) / sy;
        hx = 0.5f * (fabsf(q.e0x) * q.s0 + fabsf(q.e1x) * q.s1) * fabsf(sx);
        hy = 0.5f * (fabsf(q.e0y) * q.s0 + fabsf(q.e1y) * q.s1) * fabsf(sy);
        valid = fin(cx) && fin(cy) && fin(a0x) && fin(a0y) && fin(a1x) && fin(a1y) && fin(hx) && fin(hy) && fin(alpha);
        if (valid) {
            // conservative pixel rectangle: candidates are pixels whose centre lies within the bbox grown by a margin that
            // dominates the rounding of the coverage predicate
            float mx = 0.01f + 1e-5f * hx, my = 0.01f + 1e-5f * hy;
            float fx0 = ceilf(cx - hx - mx - 0.5f), fx1 = floorf(cx + hx + mx - 0.5f);
            float fy0 = ceilf(cy - hy - my - 0.5f), fy1 = floorf(cy + hy + my - 0.5f);
            fx0 = fmaxf(fx0, 0.0f); fy0 = fmaxf(fy0, 0.0f);
            fx1 = fminf(fx1, (float)(W - 1)); fy1 = fminf(fy1, (float)(H - 1));
            if (fx0 <= fx1 && fy0 <= fy1) {
                rect0 = (uint32_t)fx0 | ((uint32_t)fy0 << 16);
                rect1 = (uint32_t)fx1 | ((uint32_t)fy1 << 16);
            }
        }
    }
    if (!valid) { cx = cy = a0x = a0y = a1x = a1y = hx = hy = 0.0f; alpha = 0.0f; rect0 = 1u; rect1 = 0u; }
    // the GL clamps a fragment's colour to [0, 1] before blending into the reference's RGBA8 framebuffer; where the fragment shader passes
    // the colour through unchanged that is a per-record operation (the 3D-Full shader multiplies by c first: clamped per fragment)
    if (clamp_rgb) { r = __saturatef(r); g = __saturatef(g); b = __saturatef(b); }
    if (out.trects) out.trects[i] = pack_trect(rect0, rect1);      // (null: nothing downstream reads it — a staged draw writes its list entries itself)
    float4* o = out.proj + (size_t)i * 4;
    // (x components of the two affine rows side by side, likewise y: the compositor forms u and v with packed two-float instructions)
    o[0] = make_float4(cx, cy, a0x, a1x);
    o[1] = make_float4(a0y, a1y, r, g);
    o[2] = make_float4(b, alpha, __uint_as_float(rect0), __uint_as_float(rect1));
    // Nothing but gs4d_debug_read_projected reads the fourth float4 — and it is written all the same: the record is one 64-byte line, and a
    // line written whole goes to memory as it is, while a line with 16 bytes missing has to be merged with what memory holds.  Measured
    // with this store left out (round 3, 10^7 records): the projection kernel 373 -> 459 us; at 10^6 records no difference (40.1 / 40.4 us).
    o[3] = make_float4(hx, hy, valid ? 1.0f : 0.0f, 0.0f);
    return make_uint2(rect0, rect1);
}

// Unordered draw path (tilelist.hip): the projection kernel also counts, per bucket b = tile % nb, the tile-list entries of the segment
// of records its workgroup walks (LDS atomics; the row of counts is stored once at the end) and stores each record's blend-order key.
// Called by all 64 lanes of the wave (`r.count` == 0 for lanes without a record): footprints of more than 16 tiles are counted by the
// whole wave.
__device__ __forceinline__ void count_buckets(uint32_t* h, uint32_t nbm, uint32_t tiles_x, const TRect& r) {
    const bool big = r.count > 16u;
    if (!big) for_each_tile(r, tiles_x, [&](uint32_t t) { atomicAdd(&h[t & nbm], 1u); });
    uint64_t m = __ballot(big);
    const uint32_t lane = threadIdx.x & 63u;
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        TRect rr;
        rr.tx0 = __shfl(r.tx0, src, 64); rr.ty0 = __shfl(r.ty0, src, 64); rr.wx = __shfl(r.wx, src, 64);
        rr.rows = __shfl(r.rows, src, 64); rr.tstep = __shfl(r.tstep, src, 64); rr.count = __shfl(r.count, src, 64);
        for (uint32_t j = lane; j < rr.count; j += 64u) atomicAdd(&h[tile_of(rr, j, tiles_x) & nbm], 1u);
    }
}

// Staged lists, second pass: the same walk again, every entry placed in the workgroup's LDS block at the position a returning LDS atomic on
// its bucket's cursor hands out (the cursors start at the scanned counts of the first pass).
__device__ __forceinline__ void place_buckets(uint32_t* cur, uint32_t nbm, uint32_t nbs, uint32_t tiles_x, const TRect& r, uint32_t key, uint32_t rec, uint2* stage) {
    const bool big = r.count > 16u;
    if (!big) for_each_tile(r, tiles_x, [&](uint32_t t) { stage[atomicAdd(&cur[t & nbm], 1u)] = make_uint2(key, ((t >> nbs) << 24) | rec); });
    uint64_t m = __ballot(big);
    const uint32_t lane = threadIdx.x & 63u;
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        TRect rr;
        rr.tx0 = __shfl(r.tx0, src, 64); rr.ty0 = __shfl(r.ty0, src, 64); rr.wx = __shfl(r.wx, src, 64);
        rr.rows = __shfl(r.rows, src, 64); rr.tstep = __shfl(r.tstep, src, 64); rr.count = __shfl(r.count, src, 64);
        const uint32_t key2 = __shfl(key, src, 64), rec2 = __shfl(rec, src, 64);
        for (uint32_t j = lane; j < rr.count; j += 64u) { const uint32_t t = tile_of(rr, j, tiles_x); stage[atomicAdd(&cur[t & nbm], 1u)] = make_uint2(key2, ((t >> nbs) << 24) | rec2); }
    }
}

// the depth key k_keygen (sort.hip) writes for this record, as a bit pattern relative to the host-proven lower bound
__device__ __forceinline__ uint32_t blend_key_4d(const KeySrc& ks, uint32_t i, const float4& p, const float4& s) {
    if (ks.mode == KEYSRC_INDEX) return i;
    float key;
    if (ks.mode == KEYSRC_REF) {
        float ct = ks.t - p.w;                             // Scenes.h:30
        float x = p.x + s.x * ct;                          // :31-33
        float y = p.y + s.y * ct;
        float z = p.z + s.z * ct;
        float dx = x - ks.camx, dy = y - ks.camy, dz = z - ks.camz;     // :317
        key = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);                // :318
    } else {
        float k = (1.0f / s.w) * (ks.t - p.w);
        float x = p.x + k * s.x, y = p.y + k * s.y, z = p.z + k * s.z;
        float zv = ((ks.vr0 * x + ks.vr1 * y) + ks.vr2 * z) + ks.vr3;
        key = 1.0f / fmaxf(-zv, 1e-20f);
    }
    return __float_as_uint(key) - ks.bias;
}

// …Instanced.GLSL:97-147 == Splat3DVertexShaderFull.GLSL:45-95.  C[c][r] = 3x3 covariance.
__device__ __forceinline__ bool project3d(const PU& u, float mx, float my, float mz, const float C[3][3], Quad& q, float& ncx, float& ncy) {
    const float* V = u.V; const float* P = u.P;
    float pcx = ((V[0] * mx + V[4] * my) + V[8] * mz) + V[12] * 1.0f;
    float pcy = ((V[1] * mx + V[5] * my) + V[9] * mz) + V[13] * 1.0f;
    float pcz = ((V[2] * mx + V[6] * my) + V[10] * mz) + V[14] * 1.0f;
    float pcw = ((V[3] * mx + V[7] * my) + V[11] * mz) + V[15] * 1.0f;
    float psx = ((P[0] * pcx + P[4] * pcy) + P[8] * pcz) + P[12] * pcw;
    float psy = ((P[1] * pcx + P[5] * pcy) + P[9] * pcz) + P[13] * pcw;
    float psz = ((P[2] * pcx + P[6] * pcy) + P[10] * pcz) + P[14] * pcw;
    float psw = ((P[3] * pcx + P[7] * pcy) + P[11] * pcz) + P[15] * pcw;
    float rw = 1.0f / psw;
    psx = rw * psx; psy = rw * psy; psz = rw * psz; psw = rw * psw;
    float z = psz / psw;
    float bound = 1.2f * psw;
    if (z < 0.0f || z > 1.0f || psx < -bound || psx > bound || psy < -bound || psy > bound) return false;
    if (!(fin(psx) && fin(psy) && fin(z))) return false;
    // gl_Position = uProj * vec4(R*S*v, 0, 1) + ps (:147): all four corners get z = ps.z + P[3][2] at w = 1, and the GL clips -w <= z <= w
    { const float zq = psz + P[14]; if (zq < -1.0f || zq > 1.0f) return false; }
    float z2 = pcz * pcz;
    float J[3][3] = { { 1.0f / pcz, 0.0f, -pcx / z2 }, { 0.0f, 1.0f / pcz, -pcy / z2 }, { 0.0f, 0.0f, 0.0f } };
    float Wt[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) Wt[c][r] = V[4 * r + c];
    float T[3][3], Tt[3][3], A[3][3], cov[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) T[c][r] = Wt[0][r] * J[c][0] + Wt[1][r] * J[c][1] + Wt[2][r] * J[c][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) Tt[c][r] = T[r][c];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) A[c][r] = Tt[0][r] * C[c][0] + Tt[1][r] * C[c][1] + Tt[2][r] * C[c][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 2; ++r) cov[c][r] = A[0][r] * T[c][0] + A[1][r] * T[c][1] + A[2][r] * T[c][2];
    float lx, ly;
    eigen2(cov[0][0], cov[0][1], cov[1][0], cov[1][1], false, lx, ly, q);
    q.s0 = sqrtf(lx); q.s1 = sqrtf(ly);
    ncx = psx; ncy = psy;
    return true;
}

// One record each: returns its pixel rectangle, and through `key` its blend-order key (unordered path).
struct Src4D { const float4* soa; uint32_t stride; };      // six planes of `stride` float4 (the buffer's record count, not the draw's)
struct Src4DSym { const float4* soa; uint32_t stride; };   // the compact layout of a symmetric sig: pos, col, U, V as float4 planes, W as a float2 plane
struct Src4DStatic { const float4* soa; uint32_t stride; SoaConsts cs; };      // static 3D splats: four float4 planes, the time row / column of sig and mu_t as constants
struct Src3D { const float* verts; };
struct Src2D { const float* recs; };

__device__ __forceinline__ uint2 project_4d(const float4& pos, const float4& col, const float4& s0, const float4& s1, const float4& s2, const float4& s3,
                                            uint32_t i, const PU& u, const PreOut& out, const KeySrc& ks, uint32_t& key);
__device__ __forceinline__ uint2 project_record(const Src4D& src, uint32_t n, uint32_t i, const PU& u, const PreOut& out, const KeySrc& ks, uint32_t& key) {
    const float4* __restrict__ soa = src.soa;
    const size_t ps = src.stride;
    const float4 pos = soa[i], col = soa[ps + i];
    const float4 s0 = soa[2 * ps + i], s1 = soa[3 * ps + i], s2 = soa[4 * ps + i], s3 = soa[5 * ps + i];
    return project_4d(pos, col, s0, s1, s2, s3, i, u, out, ks, key);
}
__device__ __forceinline__ uint2 project_record(const Src4DSym& src, uint32_t n, uint32_t i, const PU& u, const PreOut& out, const KeySrc& ks, uint32_t& key) {
    const float4* __restrict__ soa = src.soa;
    const size_t ps = src.stride;
    const float4 pos = soa[i], col = soa[ps + i], U = soa[2 * ps + i], V = soa[3 * ps + i];
    const float2 W = reinterpret_cast<const float2*>(soa + 4 * ps)[i];
    // sig[c] = column c; the mirrored elements are the same bits (verified by the repack kernel)
    return project_4d(pos, col, make_float4(U.x, U.y, U.z, V.x), make_float4(U.y, U.w, W.x, V.y), make_float4(U.z, W.x, W.y, V.z), V, i, u, out, ks, key);
}
__device__ __forceinline__ uint2 project_record(const Src4DStatic& src, uint32_t n, uint32_t i, const PU& u, const PreOut& out, const KeySrc& ks, uint32_t& key) {
    const float4* __restrict__ soa = src.soa;
    const size_t ps = src.stride;
    const float4 A = soa[i], col = soa[ps + i], B = soa[2 * ps + i], C = soa[3 * ps + i];
    const float* c = src.cs.v;
    return project_4d(make_float4(A.x, A.y, A.z, c[0]), col, make_float4(A.w, B.x, B.y, c[1]), make_float4(B.z, B.w, C.x, c[2]), make_float4(C.y, C.z, C.w, c[3]), make_float4(c[4], c[5], c[6], c[7]),
                      i, u, out, ks, key);
}
__device__ __forceinline__ uint2 project_4d(const float4& pos, const float4& col, const float4& s0, const float4& s1, const float4& s2, const float4& s3,
                                            uint32_t i, const PU& u, const PreOut& out, const KeySrc& ks, uint32_t& key) {
    float s44 = s3.w;
    float dt = u.time - pos.w;
    float ot = maxf_glsl(expf(-0.5f * dt * (1.0f / s44) * dt), u.min_opacity);     // :48-51, 83
    float ax = s0.w, ay = s1.w, az = s2.w;                                       // iSig[0][3], [1][3], [2][3]
    float k = (1.0f / s44) * dt;
    float mx = pos.x + k * ax, my = pos.y + k * ay, mz = pos.z + k * az;         // :86
    float inv = 1.0f / s44;
    float tv[3] = { inv * s3.x, inv * s3.y, inv * s3.z };                        // :87
    float a[3] = { ax, ay, az };
    float S[3][3] = { { s0.x, s0.y, s0.z }, { s1.x, s1.y, s1.z }, { s2.x, s2.y, s2.z } };
    float C[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) C[c][r] = S[c][r] - a[r] * tv[c];              // :89-95
    Quad q; float ncx = 0, ncy = 0;
    bool valid = project3d(u, mx, my, mz, C, q, ncx, ncy);
    key = blend_key_4d(ks, i, pos, s3);
    return emit(out, i, valid, q, ncx, ncy, u.P[0], u.P[5], u.W, u.H, col.x, col.y, col.z, ot * col.w, true);
}

__device__ __forceinline__ uint2 project_record(const Src3D& src, uint32_t, uint32_t i, const PU& u, const PreOut& out, const KeySrc&, uint32_t& key) {
    const float* v = src.verts + (size_t)72 * i;      // vertex 0 of the quad: {vpos2, spos3, col4, sig9}
    float C[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) C[c][r] = v[9 + 3 * c + r];
    Quad q; float ncx = 0, ncy = 0;
    bool valid = project3d(u, v[2], v[3], v[4], C, q, ncx, ncy);
    key = i;
    return emit(out, i, valid, q, ncx, ncy, u.P[0], u.P[5], u.W, u.H, v[5], v[6], v[7], v[8], false);
}

__device__ __forceinline__ uint2 project_record(const Src2D& src, uint32_t, uint32_t i, const PU& u, const PreOut& out, const KeySrc&, uint32_t& key) {
    const float* rec = src.recs + (size_t)12 * i;
    const float* P = u.P;
    float x = rec[0], y = rec[1];
    float psx = ((P[0] * x + P[4] * y) + P[8] * -1.0f) + P[12] * 1.0f;             // Splat2DVSI.GLSL:64
    float psy = ((P[1] * x + P[5] * y) + P[9] * -1.0f) + P[13] * 1.0f;
    float psz = ((P[2] * x + P[6] * y) + P[10] * -1.0f) + P[14] * 1.0f;
    float psw = ((P[3] * x + P[7] * y) + P[11] * -1.0f) + P[15] * 1.0f;
    float rw = 1.0f / psw;
    psx = rw * psx; psy = rw * psy; psz = rw * psz; psw = rw * psw;
    Quad q; float lx, ly;
    eigen2(rec[8], rec[9], rec[10], rec[11], true, lx, ly, q);
    float l0 = sqrtf(lx * 2.0f), l1 = sqrtf(ly * 2.0f);                           // :68-69
    q.s0 = l1; q.s1 = l0;                                                         // S = mat2(l1,0,0,l0) :76
    float zc = -5.0f + psz, wc4 = 1.0f + psw;
    float clipw = P[11] * zc + P[15] * wc4;
    float clipz = P[10] * zc + P[14] * wc4;
    bool valid = (clipw > 0.0f) && !(clipz < -clipw || clipz > clipw);
    float kx = P[0] / clipw, ky = P[5] / clipw;
    key = i;
    return emit(out, i, valid, q, kx * psx, ky * psy, kx, ky, u.W, u.H, rec[4], rec[5], rec[6], rec[7], true);
}

// One thread per record (the ordered path).
template <class SRC>
__global__ __launch_bounds__(256) void k_preprocess(SRC src, uint32_t n, PU u, PreOut out, TileCount tc) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t key;
    (void)project_record(src, n, i, u, out, tc.ks, key);
}

// Unordered path (COUNT): workgroup w walks records [w * seg, (w + 1) * seg), SEG_THREADS per round, and leaves the counts of its segment
// per bucket.
// FUSE_KEYS: it is also k_keygen (sort.hip) for these records — the key it derives the blend order from IS the depth key: written to the
// caller's key buffer with the identity index beside it, bounds-checked, and counted into the depth sort's digit histograms.  Without
// COUNT that is all it adds to the projection: the ordered path's form (the sort follows, then binning.hip reads the sorted index).
template <class SRC, bool FUSE_KEYS, int COUNT>      // COUNT: 0 none (ordered path), 1 count per bucket, 2 count + place + write the segment's entries (staged lists)
__global__ __launch_bounds__(SEG_THREADS) void k_project_count(SRC src, uint32_t n, PU u, PreOut out, TileCount tc) {
    __shared__ uint32_t h[COUNT ? 1024 : 1];
    __shared__ uint32_t kh[FUSE_KEYS ? OS_MAX_PASSES : 1][OS_MAX_BINS];
    __shared__ uint32_t ws[SEG_THREADS / 64 + 1];
    extern __shared__ uint2 stage[];                       // COUNT == 2: tc.scap entries
    if (COUNT) for (uint32_t b = threadIdx.x; b < tc.nb; b += SEG_THREADS) h[b] = 0u;
    if (FUSE_KEYS && threadIdx.x < 256u) os_hist_clear(kh, threadIdx.x);
    __syncthreads();
    const uint32_t i0 = blockIdx.x * tc.seg, i1 = min(n, i0 + tc.seg);
    // staged lists: what the placing pass needs of every record this thread projected (seg <= STAGE_R * SEG_THREADS: tile_lists_plan / run_draw)
    uint32_t s_key[COUNT == 2 ? STAGE_R : 1], s_r0[COUNT == 2 ? STAGE_R : 1], s_r1[COUNT == 2 ? STAGE_R : 1];
#pragma unroll
    for (int q = 0; q < (COUNT == 2 ? STAGE_R : 1); ++q) { s_key[q] = 0u; s_r0[q] = 1u; s_r1[q] = 0u; }
    auto round = [&](uint32_t ib, int it) {                 // one record per thread; every lane stays for the wave-wide counting
        const uint32_t i = ib + threadIdx.x;
        TRect r{ 0u, 0u, 0u, 0u, 1u, 0u };
        uint32_t key = 0u;
        if (i < i1) {
            const uint2 rect = project_record(src, n, i, u, out, tc.ks, key);
            if (COUNT) {
                if (!FUSE_KEYS && COUNT == 1) tc.skey[i] = key;               // fused: k_bucket_scatter takes the key from the caller's key buffer, written just below
                r = tile_rect(rect.x, rect.y, (uint32_t)tc.shard_rank, (uint32_t)tc.shard_world);
                if (COUNT == 2) { s_key[it] = key; s_r0[it] = rect.x; s_r1[it] = rect.y; }
            }
            if (FUSE_KEYS) {
                tc.keys_out[i] = __uint_as_float(key + tc.ks.bias);          // the blend key is the depth key's bit pattern relative to the bias
                if (tc.idx_out) tc.idx_out[i] = i;                            // null: the sort that follows makes up the identity payload itself (radix_sort_pairs)
                if (key > tc.span) __hip_atomic_store(tc.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // below the bias wraps to a huge value: caught too
            }
        }
        if (FUSE_KEYS) os_hist_add(kh, key, i < i1, OS_MAX_PASSES, tc.hist_rb);
        if (COUNT) count_buckets(h, tc.nb - 1u, (uint32_t)tc.tiles_x, r);
    };
    if (COUNT == 2) {
#pragma unroll
        for (int it = 0; it < STAGE_R; ++it) { const uint32_t ib = i0 + (uint32_t)it * SEG_THREADS; if (ib >= i1) break; round(ib, it); }      // uniform trip count, registers indexed at compile time
    } else {
        for (uint32_t ib = i0; ib < i1; ib += SEG_THREADS) round(ib, 0);    // uniform trip count
    }
    __syncthreads();
    if (COUNT) {
        // the row of counts (one row per bucket: k_bucket_scan / k_bucket_tiles_staged walk along the segments), the segment's total and — staged —
        // the exclusive scan of the counts: where each bucket's run starts inside the segment's block.  Thread t looks after buckets 2t, 2t + 1.
        const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6, b0 = 2u * tid;
        const uint32_t c0 = b0 < tc.nb ? h[b0] : 0u, c1 = b0 + 1u < tc.nb ? h[b0 + 1u] : 0u, sum = c0 + c1;
        uint32_t inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(inc, off, 64); if (lane >= (unsigned)off) inc += v; }
        if (lane == 63u) ws[w] = inc;
        __syncthreads();
        uint32_t base = 0, total = 0;
#pragma unroll
        for (int k = 0; k < SEG_THREADS / 64; ++k) { const uint32_t v = ws[k]; if ((unsigned)k < w) base += v; total += v; }
        const uint32_t o0 = base + inc - sum, o1 = o0 + c0;
        if (b0 < tc.nb) tc.hist[(size_t)b0 * tc.rows + blockIdx.x] = c0;
        if (b0 + 1u < tc.nb) tc.hist[(size_t)(b0 + 1u) * tc.rows + blockIdx.x] = c1;
        if (tid == 0u) tc.sstat[blockIdx.x] = total;
        if (COUNT == 2) {
            if (b0 < tc.nb) { tc.offs[(size_t)b0 * tc.rows + blockIdx.x] = o0; h[b0] = o0; }
            if (b0 + 1u < tc.nb) { tc.offs[(size_t)(b0 + 1u) * tc.rows + blockIdx.x] = o1; h[b0 + 1u] = o1; }
            const bool fits = total <= tc.scap;            // uniform (every thread added up the same ws[])
            if (!fits && tid == 0u) *tc.abort_word = tc.seq;      // the block cannot hold this segment: the draw is re-run with exact lists (every writer stores the same value)
            __syncthreads();
            if (fits) {
                const uint32_t nbs = (uint32_t)__ffs((int)tc.nb) - 1u;
#pragma unroll
                for (int q = 0; q < STAGE_R; ++q) {
                    const uint32_t i = i0 + (uint32_t)q * SEG_THREADS + tid;
                    if (i0 + (uint32_t)q * SEG_THREADS >= i1) break;          // uniform
                    const TRect r = i < i1 ? tile_rect(s_r0[q], s_r1[q], (uint32_t)tc.shard_rank, (uint32_t)tc.shard_world) : TRect{ 0u, 0u, 0u, 0u, 1u, 0u };
                    place_buckets(h, tc.nb - 1u, nbs, (uint32_t)tc.tiles_x, r, s_key[q], i, stage);
                }
                __syncthreads();
                // the block leaves the workgroup in one piece: consecutive threads, consecutive 16-byte pieces
                uint4* __restrict__ dst = reinterpret_cast<uint4*>(tc.stage_out + (size_t)blockIdx.x * tc.scap);
                const uint4* src4 = reinterpret_cast<const uint4*>(stage);
                for (uint32_t k = tid; k < (total + 1u) / 2u; k += SEG_THREADS) dst[k] = src4[k];
            }
        }
    }
    if (FUSE_KEYS && threadIdx.x < 256u) os_hist_flush(kh, tc.ghist, OS_MAX_PASSES, threadIdx.x);
}

static PU make_pu(const Uniforms& un, int W, int H) {
    PU u;
    for (int i = 0; i < 16; ++i) { u.V[i] = un.view[i]; u.P[i] = un.proj[i]; }
    u.time = un.time; u.min_opacity = un.min_opacity; u.W = W; u.H = H;
    return u;
}

template <class SRC>
static hipError_t launch_pre(hipStream_t st, SRC src, size_t n, const Uniforms& un, int W, int H, PreOut out, const TileCount& tc) {
    if (n == 0) return hipSuccess;
    if (tc.hist && tc.stage_out && tc.keys_out) k_project_count<SRC, true, 2><<<dim3((unsigned)((n + tc.seg - 1) / tc.seg)), dim3(SEG_THREADS), (size_t)tc.scap * 8, st>>>(src, (uint32_t)n, make_pu(un, W, H), out, tc);
    else if (tc.hist && tc.stage_out) k_project_count<SRC, false, 2><<<dim3((unsigned)((n + tc.seg - 1) / tc.seg)), dim3(SEG_THREADS), (size_t)tc.scap * 8, st>>>(src, (uint32_t)n, make_pu(un, W, H), out, tc);
    else if (tc.hist && tc.keys_out) k_project_count<SRC, true, 1><<<dim3((unsigned)((n + tc.seg - 1) / tc.seg)), dim3(SEG_THREADS), 0, st>>>(src, (uint32_t)n, make_pu(un, W, H), out, tc);
    else if (tc.hist) k_project_count<SRC, false, 1><<<dim3((unsigned)((n + tc.seg - 1) / tc.seg)), dim3(SEG_THREADS), 0, st>>>(src, (uint32_t)n, make_pu(un, W, H), out, tc);
    else if (tc.keys_out) {
        // segments sized so that at most 2048 workgroups flush digit histograms (1024 global atomics each)
        TileCount t2 = tc;
        t2.seg = (uint32_t)std::max<size_t>(4096, (((n + 2047) / 2048) + SEG_THREADS - 1) / SEG_THREADS * SEG_THREADS);
        k_project_count<SRC, true, 0><<<dim3((unsigned)((n + t2.seg - 1) / t2.seg)), dim3(SEG_THREADS), 0, st>>>(src, (uint32_t)n, make_pu(un, W, H), out, t2);
    }
    else k_preprocess<SRC><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(src, (uint32_t)n, make_pu(un, W, H), out, tc);
    return hipGetLastError();
}
hipError_t launch_preprocess_4d(hipStream_t st, const float4* soa, size_t soa_n, const SoaInfo& info, size_t n, const Uniforms& un, int W, int H, PreOut out, const TileCount& tc) {
    if (info.layout == SOA_STATIC3D) { Src4DStatic s{ soa, (uint32_t)soa_n, SoaConsts() }; for (int k = 0; k < 8; ++k) s.cs.v[k] = info.consts[k]; return launch_pre(st, s, n, un, W, H, out, tc); }
    return info.layout == SOA_SYM ? launch_pre(st, Src4DSym{ soa, (uint32_t)soa_n }, n, un, W, H, out, tc) : launch_pre(st, Src4D{ soa, (uint32_t)soa_n }, n, un, W, H, out, tc);
}
hipError_t launch_preprocess_3d(hipStream_t st, const float* verts72, size_t n, const Uniforms& un, int W, int H, PreOut out, const TileCount& tc) { return launch_pre(st, Src3D{ verts72 }, n, un, W, H, out, tc); }
hipError_t launch_preprocess_2d(hipStream_t st, const float* rec48, size_t n, const Uniforms& un, int W, int H, PreOut out, const TileCount& tc) { return launch_pre(st, Src2D{ rec48 }, n, un, W, H, out, tc); }

} // namespace gs4d
