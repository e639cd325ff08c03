// gsr_tile_cull.h -- exact culling of tile instances at EMISSION (round 4).
//
// The reference emits one instance for every tile of a gaussian's bounding rect (3DGS rasterizer_impl.cu:70-111, auxiliary.h:46-56): on the
// SURVEY 8d scene 37 % of those instances cannot reach alpha >= 1/255 on any pixel of their tile (the rect is the box of a 3-sigma ellipse), yet they
// are sorted, range-scanned, depth-sorted, cull-tested and staged by every later stage.  Here a tile is counted (preprocess) and emitted
// (k_duplicate) only if the gaussian's cull record -- the exact region where alpha can reach 1/255, the one the blend kernels already trust at
// 8x8 / 4x4 granularity (gsr_blend_common.h cull_hit_rec) -- meets the tile's 16x16 pixel rectangle.  A dropped instance fails the reference's own
// alpha gate on every pixel of the tile, so every OUTPUT (images, radii, out_observe, gradients) is unchanged; only the private per-tile list
// positions (point_list, n_contrib) are those of the filtered list.  GSR_TILE_CULL=0 keeps the reference-shaped list.
//
// The count and the emission must agree instance for instance: both evaluate tile_cull_hit<V> below on the same two float4 words.  The function is
// written with IEEE divisions and `fp contract(off)`, so its result does not depend on the flags of the translation unit that inlines it.
#pragma once
#include "gsr_common.h"

// minimum over the rectangle [X0, X0+ext] x [Y0, Y0+ext] (relative to the conic centre) of q(d) = A dx^2 + 2 B dx dy + C dy^2:
// 0 when the centre is inside, else the best of the four edges, each a clamped 1-D parabola.  mBC = -B / C, mBA = -B / A: the two IEEE divisions are
// taken once per gaussian (tc_slopes) instead of four times per candidate tile.
__device__ __forceinline__ float2 tc_slopes(const float4 a, const float4 b)
{
#pragma clang fp contract(off)
    return make_float2(-a.w / b.x, -a.w / a.z);
}
__device__ __forceinline__ float tc_conic_min(float A, float B, float C, float mBC, float mBA, float X0, float Y0, float ext)
{
#pragma clang fp contract(off)
    const float X1 = X0 + ext, Y1 = Y0 + ext;
    if (X0 <= 0.f && X1 >= 0.f && Y0 <= 0.f && Y1 >= 0.f) return 0.f;
    float dy = fminf(fmaxf(mBC * X0, Y0), Y1);
    float qmin = A * X0 * X0 + 2.f * B * X0 * dy + C * dy * dy;
    dy = fminf(fmaxf(mBC * X1, Y0), Y1);
    qmin = fminf(qmin, A * X1 * X1 + 2.f * B * X1 * dy + C * dy * dy);
    float dx = fminf(fmaxf(mBA * Y0, X0), X1);
    qmin = fminf(qmin, A * dx * dx + 2.f * B * dx * Y0 + C * Y0 * Y0);
    dx = fminf(fmaxf(mBA * Y1, X0), X1);
    qmin = fminf(qmin, A * dx * dx + 2.f * B * dx * Y1 + C * Y1 * Y1);
    return qmin;
}

// can the gaussian with cull record (a, b) -- and slopes sl = tc_slopes(a, b) -- reach alpha >= 1/255 on some pixel of the 16x16 tile whose first pixel
// is (ox, oy)?  Conservative (the continuous rectangle, the inflated records of gsr_preprocess.hip); NaN keeps the instance (a record with
// A = B = C = 0, "culling disabled", has NaN slopes and a zero form: kept).
template <int V>
__device__ __forceinline__ bool tile_cull_hit(const float4 a, const float4 b, const float2 sl, float ox, float oy)
{
#pragma clang fp contract(off)
    const float ext = (float)(GSR_TILE - 1);
    if (V == GSR_SURFEL) {
        if (!(b.y >= 0.f)) return false;                 // opacity below 1/255: no pixel anywhere
        const float ex0 = ox - b.z, ey0 = oy - b.w;      // low-pass disc of radius^2 b.y about (b.z, b.w)
        const float ddx = fmaxf(fmaxf(ex0, -(ex0 + ext)), 0.f), ddy = fmaxf(fmaxf(ey0, -(ey0 + ext)), 0.f);
        if (ddx * ddx + ddy * ddy <= b.y) return true;
        return !(tc_conic_min(a.z, a.w, b.x, sl.x, sl.y, ox - a.x, oy - a.y, ext) > 1.0f);
    } else {
        if (!(b.y > 0.f)) return false;
        return !(tc_conic_min(a.z, a.w, b.x, sl.x, sl.y, ox - a.x, oy - a.y, ext) > b.y);
    }
}

// Wave-cooperative walk over the tile rects of the wave's 64 gaussians, lane = candidate instance (y outer, x inner inside a rect, gaussians in
// lane order: the reference's emission order).  `excl` = exclusive prefix over the lanes of the UNCULLED rect areas, `total` their sum
// (wave-uniform).  s_cull / s_slope / s_rect: the wave's LDS copies of the 64 cull records, their slopes (tc_slopes) and rects.  For candidate k the functor receives
// (in_range, source lane s, tile x, tile y, hit).
struct TcCand { uint32_t s, tx, ty; bool hit; };
template <int V>
__device__ __forceinline__ TcCand tc_candidate(uint32_t k, uint32_t total, uint32_t excl, const float4* __restrict__ s_cull, const float2* __restrict__ s_slope,
                                               const ushort4* __restrict__ s_rect)
{
    TcCand c;
    uint32_t s = 0;
#pragma unroll
    for (uint32_t step = 32; step >= 1; step >>= 1) {
        const uint32_t t = s + step;
        const uint32_t e = (uint32_t)__shfl((int)excl, (int)(t & 63u), 64);
        if (t < 64u && e <= k) s = t;
    }
    // zero-area lanes share their successor's prefix value: the search lands on the LAST lane with excl <= k, which is the one that owns k
    const uint32_t j = k - (uint32_t)__shfl((int)excl, (int)s, 64);
    const ushort4 r = s_rect[s];
    const uint32_t w = (uint32_t)r.z - (uint32_t)r.x;
    const uint32_t q = (uint32_t)(((float)j + 0.5f) / (float)(w ? w : 1u));      // j / w: exact for j < 2^20
    c.s = s; c.tx = (uint32_t)r.x + (j - q * w); c.ty = (uint32_t)r.y + q;
    c.hit = false;
    if (k < total) c.hit = tile_cull_hit<V>(s_cull[2 * s], s_cull[2 * s + 1], s_slope[s], (float)(c.tx * GSR_TILE), (float)(c.ty * GSR_TILE));
    return c;
}
