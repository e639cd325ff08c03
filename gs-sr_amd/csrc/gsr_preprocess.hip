// gsr_preprocess.hip -- per-gaussian stages (one thread per gaussian, HBM-bound):
//   forward : frustum cull, EWA / surfel projection, tile rect, SH->RGB, packed blend record, conservative cull box
//   backward: conic/transMat gradients -> means3D, scales, rotations, SH
//   helpers : visible_filter (scaffold-filter), mark_visible
// Built with -ffp-contract=off (see gsr_math.h).  Reference behaviour cited per kernel.
#include "gsr_common.h"
#include "gsr_tile_sort.h"
#include "gsr_tile_cull.h"
#include <stdlib.h>
#include "gsr_math.h"

using namespace gsr;

struct PreParams {
    int P, D, M, W, H, gx, gy, variant;
    float tanfovx, tanfovy, fx, fy, mod;
    const float *means3D, *shs, *colors, *opac, *scales, *rots, *cov3D_pre, *all_map, *view, *proj, *campos;
    int32_t* radii;
    GeomView g;
    const uint8_t* in_mask;   // FILTER only: optional per-gaussian pre-mask (Octree LOD); masked-out entries get radii 0 without any work
    int scale_stride;         // floats between consecutive scale triples (3, or 6 when fed get_scaling directly)
    int no_cull;           // GSR_NO_CULL=1 (diagnostic): every visible gaussian passes the sub-tile cull -> outputs must not change
    uint32_t* zero_ptr; uint32_t zero_n;      // first group-histogram buffer of the depth sort that follows (gsr_binning.hip): cleared here
    // per-tile depth order (instances emitted in id order): the block-local inclusive prefix of tiles_touched and the block totals are written here, by
    // this kernel, instead of by a k_offsets_local launch of their own (5 us); nullptr: not wanted (global depth order, visible_filter)
    uint32_t* scan_offsets; uint32_t* scan_sums;
    // GaussianRasterizationSettings.prefiltered: the caller asserts that no gaussian fails the frustum test.  The reference prints "Point is filtered
    // although prefiltered is set" and traps the device (auxiliary.h:156-160); here the kernel stores 1 into this mapped word and the forward call
    // that owns it fails with that message.  nullptr: not checked.
    uint32_t* prefiltered_err;
    uint32_t* mode_word; uint32_t mode;      // GeomView::counters[GSR_CNT_MODE] <- GSR_MODE_TILE / GSR_MODE_GLOBAL: the depth order this arena is being laid out for
    int tile_cull;            // 1 (default): tiles_touched counts only the tiles of the rect the cull record can reach (gsr_tile_cull.h); 0 (GSR_TILE_CULL=0): the whole rect
};
// what the tile-instance count needs from the per-gaussian pass: the unculled tile count, the rect and the cull record (also stored in the geom arena)
struct PreTc { uint32_t tiles; ushort4 rect; float4 c0, c1; };

__device__ __forceinline__ void load16(const float* p, float* m)
{
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = p[i];
}

// Conservative half-extent scale for "alpha >= 1/255 is possible":  o * exp(-q/2) >= 1/255  <=>  q <= 2*ln(255*o).
// Returns 2*tau (>0) or a negative number when the gaussian can never reach 1/255.
__device__ __forceinline__ float two_tau(float o)
{
    float v = 255.0f * o;
    if (!(v > 1.0f)) return -1.0f;
    return 2.0f * logf(v) * 1.001f + 1e-3f;
}

// ------------------------------------------------------------------------------------------------ EWA / PLANE
// 3DGS forward.cu:156-256 (PLANE forward.cu:156-268 is identical); FILTER forward.cu:268-340 when FILTER_ONLY.
template <bool FILTER_ONLY>
__device__ __forceinline__ PreTc pre_ewa_one(const PreParams& p, const int idx, const float* sh_staged = nullptr)      // -> rect tiles, rect, cull record
{
    PreTc tc; tc.tiles = 0u; tc.rect = make_ushort4(0, 0, 0, 0); tc.c0 = make_float4(0, 0, 0, 0); tc.c1 = make_float4(0, -1, 0, 0);
    if (FILTER_ONLY && p.in_mask && !p.in_mask[idx]) { p.radii[idx] = 0; return tc; }
    float view[16], proj[16];
    load16(p.view, view); load16(p.proj, proj);

    int radius_out = 0; uint32_t tiles = 0, key = 0xFFFFFFFFu, clamped = 0;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0, cull0 = q0, cull1 = make_float4(0, -1, 0, 0);

    const float3 p_orig = make_float3(p.means3D[3 * idx], p.means3D[3 * idx + 1], p.means3D[3 * idx + 2]);
    const float3 p_view = xform_point4x3(p_orig, view);
    do {
        if (p_view.z <= 0.2f) {                                         // in_frustum, auxiliary.h:139-164
            if (p.prefiltered_err) __hip_atomic_store(p.prefiltered_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        float4 p_hom = xform_point4x4(p_orig, proj);
        float p_w = 1.0f / (p_hom.w + 0.0000001f);
        float projx = p_hom.x * p_w, projy = p_hom.y * p_w;
        float cov3D[6];
        if (p.cov3D_pre) {
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D[i] = p.cov3D_pre[6 * idx + i];
        } else {
            cov3d_from_scale_rot(p.scales + (size_t)p.scale_stride * idx, p.mod, p.rots + 4 * idx, cov3D);
        }
        Cov2D cv = cov2d_project(p_orig, p.fx, p.fy, p.tanfovx, p.tanfovy, cov3D, view);
        float det = (cv.a * cv.c - cv.b * cv.b);
        if (det == 0.0f) break;
        float det_inv = 1.f / det;
        float conA = cv.c * det_inv, conB = -cv.b * det_inv, conC = cv.a * det_inv;
        float mid = 0.5f * (cv.a + cv.c);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        float pix = ndc2pix(projx, p.W), piy = ndc2pix(projy, p.H);
        tile_rect(pix, piy, (int)my_radius, p.gx, p.gy, x0, y0, x1, y1);
        if ((x1 - x0) * (y1 - y0) == 0) break;
        radius_out = (int)my_radius;
        if (FILTER_ONLY) break;

        float3 rgb;
        if (p.colors) rgb = make_float3(p.colors[3 * idx], p.colors[3 * idx + 1], p.colors[3 * idx + 2]);
        else rgb = sh_to_rgb(p.D, sh_staged ? sh_staged : p.shs + (size_t)idx * p.M * 3, p_orig,
                             make_float3(p.campos[0], p.campos[1], p.campos[2]), clamped);
        const float o = p.opac[idx];
        key = __float_as_uint(p_view.z);
        tiles = (uint32_t)((y1 - y0) * (x1 - x0));
        q0 = make_float4(pix, piy, conA, conB);
        q1 = make_float4(conC, o, rgb.x, rgb.y);
        q2.x = rgb.z;
        if (p.variant == GSR_PLANE && p.all_map) {
            const float* am = p.all_map + 5 * idx;
            q2.y = am[0]; q2.z = am[1]; q2.w = am[2]; q3.x = am[3]; q3.y = am[4];
        }
        // cull record for the blend kernels: conic form + 2*tau (the pair contributes iff dT Q d <= 2 tau)
        cull0 = make_float4(pix, piy, conA, conB);
        cull1 = make_float4(conC, two_tau(o), 0.f, 0.f);
        // the block test (gsr_blend.hip conic_min_over_block) assumes a positive-definite form: fp32 cancellation in det = a c - b^2 can
        // leave a huge / near-degenerate splat with an indefinite conic, whose per-pixel gates the reference would still evaluate.
        // Anything doubtful disables culling for this splat (A = B = C = 0 with tt > 0 always hits), as for surfels.
        // the same goes for a conic whose float32 evaluation at the far corner of the rect cancels below the 2 tau margin (huge needles)
        {
            const float D = 1.5f * my_radius + 8.f;
            if (!(conA > 0.f && conC > 0.f && conA * conC - conB * conB > 0.f) || !((conA + 2.f * fabsf(conB) + conC) * D * D < 2.5e4f)) {
                cull0.z = 0.f; cull0.w = 0.f; cull1.x = 0.f; if (!(cull1.y > 0.f)) cull1.y = -1.f;
            }
        }
        if (p.no_cull) { cull0 = make_float4(pix, piy, 0.f, 0.f); cull1 = make_float4(0.f, 1.0f, 0.f, 0.f); }
    } while (0);

    p.radii[idx] = radius_out;
    if (FILTER_ONLY) return tc;
    p.g.depth_key[idx] = key;
    tc.tiles = tiles; tc.rect = make_ushort4((unsigned short)x0, (unsigned short)y0, (unsigned short)x1, (unsigned short)y1); tc.c0 = cull0; tc.c1 = cull1;
    p.g.rect[idx] = tc.rect;
    p.g.cull[2 * (size_t)idx] = cull0;
    p.g.cull[2 * (size_t)idx + 1] = cull1;
    p.g.clamped[idx] = clamped;
    const int st = (p.variant == GSR_PLANE) ? GSR_REC_PLANE : GSR_REC_EWA;
    float4* rec = p.g.rec + (size_t)idx * st;
    rec[0] = q0; rec[1] = q1; rec[2] = q2;
    if (p.variant == GSR_PLANE) rec[3] = q3;
    return tc;
}
// tiles_touched of the wave's 64 gaussians with the tiles their cull record cannot reach left out (gsr_tile_cull.h): the wave walks the rects of its
// gaussians 64 candidate tiles at a time (lane = candidate), one region test each, and every gaussian counts the hits inside its own span of the
// ballot.  k_duplicate (gsr_binning.hip) repeats the same walk with the same function on the same words when it emits the instances.
template <int V>
__device__ __forceinline__ uint32_t pre_tile_count(const PreParams& p, const int idx, const PreTc& tc)
{
    uint32_t tiles = tc.tiles;
    if (p.tile_cull) {                                // kernel-uniform
        __shared__ float4 s_cull[2 * 256];
        __shared__ float2 s_slope[256];
        __shared__ ushort4 s_rect[256];
        const uint32_t lane = tds_lane(), wbase = threadIdx.x & ~63u;
        s_cull[2 * threadIdx.x] = tc.c0; s_cull[2 * threadIdx.x + 1] = tc.c1; s_slope[threadIdx.x] = tc_slopes(tc.c0, tc.c1); s_rect[threadIdx.x] = tc.rect;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t incl = tds_wave_incl_scan(tc.tiles);
        const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64), excl = incl - tc.tiles;
        uint32_t kept = 0u;
        for (uint32_t k0 = 0; k0 < total; k0 += 64u) {
            const TcCand c = tc_candidate<V>(k0 + lane, total, excl, s_cull + 2 * wbase, s_slope + wbase, s_rect + wbase);
            const uint64_t hits = __ballot(c.hit);
            // this gaussian's candidates occupy [excl, incl) of the walk: bits [lo, hi) of this batch
            const uint32_t lo = max(excl, k0) - k0, hi = min(incl, k0 + 64u) > k0 ? min(incl, k0 + 64u) - k0 : 0u;
            if (hi > lo) {
                const uint64_t span = (hi >= 64u ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
                kept += (uint32_t)__popcll(hits & span);
            }
        }
        tiles = kept;
    }
    if (idx < p.P) p.g.tiles_touched[idx] = tiles;
    return tiles;
}
// the prefix of tiles_touched the id-order binning needs, block part: inclusive scan over the block's 256 gaussians + the block total
__device__ __forceinline__ void pre_block_scan(const PreParams& p, const int idx, const uint32_t tiles)
{
    if (!p.scan_offsets) return;                 // kernel-uniform
    __shared__ uint32_t lds[17];
    uint32_t tot;
    const uint32_t incl = tds_block_incl_scan(tiles, lds, &tot);
    if (idx < p.P) p.scan_offsets[idx] = incl;
    if (threadIdx.x == 0) p.scan_sums[blockIdx.x] = tot;
}
template <bool FILTER_ONLY>
__global__ void __launch_bounds__(256) k_preprocess_ewa(PreParams p)
{
    if (p.mode_word && blockIdx.x == 0 && threadIdx.x == 0) { *p.mode_word = p.mode; p.mode_word[GSR_CNT_CULL_MISMATCH - GSR_CNT_MODE] = 0u; }
    for (uint32_t z = (uint32_t)(blockIdx.x * blockDim.x + threadIdx.x); z < p.zero_n; z += gridDim.x * blockDim.x) p.zero_ptr[z] = 0u;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    PreTc tc; tc.tiles = 0u; tc.rect = make_ushort4(0, 0, 0, 0); tc.c0 = make_float4(0, 0, 0, 0); tc.c1 = make_float4(0, -1, 0, 0);
    if (idx < p.P) tc = pre_ewa_one<FILTER_ONLY>(p, idx);
    if (!FILTER_ONLY) pre_block_scan(p, idx, pre_tile_count<GSR_EWA>(p, idx, tc));
}

// The same with the SH coefficients (16 x 3 floats = 192 B per gaussian) staged through LDS.  Read straight from global memory, lane l's 48 floats
// sit 192 B from lane l+1's: every load instruction of the wave touches 64 different cache lines and the texture addresser, which takes a line per
// cycle, becomes the kernel (46 us against 15 us with precomputed colours at P = 300k).  Here the wave copies its 64 gaussians' 12 KB with 12 fully
// coalesced 16-byte-per-lane loads into its private LDS slice, one row of 49 words per gaussian (odd stride: lane l reading word k of row l is
// conflict-free), and sh_to_rgb reads from there.  Same arithmetic on the same values: bit-identical results.  M = 16 only (degree-3 models).
#define GSR_SH_ROW 49
// ---- SH rows through LDS (see k_preprocess_ewa_sh16): the wave's 64 gaussians x 48 floats, 16 bytes per lane and instruction
__device__ __forceinline__ void sh_rows_load(float* my, const float* __restrict__ shs, size_t g0, int lane, size_t total)
{
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const int w = 4 * (i * 64 + lane);                     // word offset inside the wave's 64 x 48 block: 1 KB per instruction
        if (g0 * 48 + w + 3 < total) {
            const float4 v = *reinterpret_cast<const float4*>(shs + g0 * 48 + w);
            const int g = w / 48, j = w - g * 48;              // 48 is a multiple of 4: the four words stay inside one gaussian's row
            float* d = my + g * GSR_SH_ROW + j;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void sh_rows_store(const float* my, float* __restrict__ dst, size_t g0, int lane, size_t total)
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const int w = 4 * (i * 64 + lane);
        if (g0 * 48 + w + 3 < total) {
            const int g = w / 48, j = w - g * 48;
            const float* r = my + g * GSR_SH_ROW + j;
            *reinterpret_cast<float4*>(dst + g0 * 48 + w) = make_float4(r[0], r[1], r[2], r[3]);
        }
    }
}
__global__ void __launch_bounds__(256) k_preprocess_ewa_sh16(PreParams p)
{
    if (p.mode_word && blockIdx.x == 0 && threadIdx.x == 0) { *p.mode_word = p.mode; p.mode_word[GSR_CNT_CULL_MISMATCH - GSR_CNT_MODE] = 0u; }
    __shared__ float s_sh[4 * 64 * GSR_SH_ROW];
    for (uint32_t z = (uint32_t)(blockIdx.x * blockDim.x + threadIdx.x); z < p.zero_n; z += gridDim.x * blockDim.x) p.zero_ptr[z] = 0u;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* my = s_sh + wave * 64 * GSR_SH_ROW;
    sh_rows_load(my, p.shs, (size_t)(idx - lane), lane, (size_t)p.P * 48);
    PreTc tc; tc.tiles = 0u; tc.rect = make_ushort4(0, 0, 0, 0); tc.c0 = make_float4(0, 0, 0, 0); tc.c1 = make_float4(0, -1, 0, 0);
    if (idx < p.P) tc = pre_ewa_one<false>(p, idx, my + lane * GSR_SH_ROW);
    pre_block_scan(p, idx, pre_tile_count<GSR_EWA>(p, idx, tc));
}

// ------------------------------------------------------------------------------------------------ SURFEL
// SURFEL forward.cu:149-251
__device__ __forceinline__ PreTc pre_surfel_one(const PreParams& p, const int idx, const float* sh_staged = nullptr)      // -> rect tiles, rect, cull record
{
    PreTc tc;
    float view[16], proj[16];
    load16(p.view, view); load16(p.proj, proj);

    int radius_out = 0; uint32_t tiles = 0, key = 0xFFFFFFFFu, clamped = 0;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    float T[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    float3 normal = make_float3(0, 0, 0), rgb = make_float3(0, 0, 0);
    float pix = 0, piy = 0, o = 0;
    float4 cull0 = make_float4(0, 0, 0, 0), cull1 = make_float4(0, -1.f, 0, 0);     // culled gaussians never hit (r2 < 0)

    const float3 p_orig = make_float3(p.means3D[3 * idx], p.means3D[3 * idx + 1], p.means3D[3 * idx + 2]);
    const float3 p_view = xform_point4x3(p_orig, view);
    do {
        if (p_view.z <= 0.2f) {
            if (p.prefiltered_err) __hip_atomic_store(p.prefiltered_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        if (!p.cov3D_pre) {
            M3 R;
            surfel_transmat(p_orig, p.scales + 2 * idx, p.mod, p.rots + 4 * idx, proj, view, p.W, p.H, T, normal, R);
        } else {
#pragma unroll
            for (int i = 0; i < 9; i++) T[i] = p.cov3D_pre[9 * idx + i];
            normal = make_float3(0.0f, 0.0f, 1.0f);
        }
        // DUAL_VISIABLE (forward.cu:209-214)
        float cosv = -(p_view.x * normal.x + p_view.y * normal.y + p_view.z * normal.z);
        if (cosv == 0) break;
        float mult = cosv > 0 ? 1.f : -1.f;
        normal = make_float3(mult * normal.x, mult * normal.y, mult * normal.z);
        float ex, ey;
        if (!surfel_aabb(T, 9.0f, pix, piy, ex, ey)) break;
        float radius = ceilf(fmaxf(fmaxf(ex, ey), 3.0f * 0.707106f));
        tile_rect(pix, piy, (int)radius, p.gx, p.gy, x0, y0, x1, y1);
        if ((x1 - x0) * (y1 - y0) == 0) break;
        radius_out = (int)radius;
        if (p.colors) rgb = make_float3(p.colors[3 * idx], p.colors[3 * idx + 1], p.colors[3 * idx + 2]);
        else rgb = sh_to_rgb(p.D, sh_staged ? sh_staged : p.shs + (size_t)idx * p.M * 3, p_orig,
                             make_float3(p.campos[0], p.campos[1], p.campos[2]), clamped);
        o = p.opac[idx];
        key = __float_as_uint(p_view.z);
        tiles = (uint32_t)((y1 - y0) * (x1 - x0));
        // Sub-tile cull record: the region where alpha >= 1/255, i.e. {min(rho3d, rho2d) <= tt}, is the union of
        //   (i) the projected ellipse of the sqrt(tt)-sigma contour, d^T Sigma'^-1 d <= 1 about its centre c, with
        //       Sigma' = c c^T - S/w from the dual conic Q* = sum_i t_i T_i T_i^T, t = (tt, tt, -1)  (the same quantities compute_aabb,
        //       SURFEL forward.cu:119-145, takes its centre and extents from), valid while the contour stays in front of the camera, and
        //   (ii) the screen-space low-pass disc |xy - pix|^2 <= tt / 2.
        // The blend kernels test both exactly against the 8x8 block (gsr_blend.hip cull_hit); a box of the union let 21 % of the
        // (splat, block) pairs through that touch no pixel.  Inflation: +2 % on the form, +0.25 px^2 and the fp32 cancellation bound
        // 1e-6 c^2 on the diagonal.  Anything doubtful disables culling for this splat (A = B = C = 0 always hits).
        float tt = two_tau(o);
        if (tt > 0) {
            cull0 = make_float4(pix, piy, 0.f, 0.f);
            cull1 = make_float4(0.f, 0.5f * tt * 1.001f + 0.01f, pix, piy);
            // The dual-conic sums cancel (tt Tu.x Tw.x + tt Tu.y Tw.y - Tu.z Tw.z, and c c^T - S/w with both terms ~1/w^2 for splats
            // close to edge-on): in float32 the centre of a x20 needle came out wrong by a few pixels and the blend skipped a splat with
            // alpha 0.2-0.3 on five pixels (randomised sweep, surfel seed 28, rounds 1-2 -- the float64 and float32 replays of the
            // reference formulas agree with the oracle there, the culled kernel did not).  Evaluated in float64 the cancellation error is
            // 1e-9 of what it was; the inflation terms stay.  ~60 DP operations per visible surfel, once per forward.
            const double Tud[3] = { T[0], T[1], T[2] }, Tvd[3] = { T[3], T[4], T[5] }, Twd[3] = { T[6], T[7], T[8] };
            const double t3[3] = { (double)tt, (double)tt, -1.0 };
            auto qd = [&](const double* X, const double* Y) { return t3[0] * X[0] * Y[0] + t3[1] * X[1] * Y[1] + t3[2] * X[2] * Y[2]; };
            const double w = qd(Twd, Twd);
            const double tw2 = Twd[2] * Twd[2];
            if (w < -1e-4 * tw2 && T[8] > 0) {
                const double iw = 1.0 / w;
                const double cx = qd(Tud, Twd) * iw, cy = qd(Tvd, Twd) * iw;
                double sxx = cx * cx - qd(Tud, Tud) * iw, syy = cy * cy - qd(Tvd, Tvd) * iw;
                const double sxy = cx * cy - qd(Tud, Tvd) * iw;
                sxx += 0.25 + 1e-6 * cx * cx; syy += 0.25 + 1e-6 * cy * cy;
                const double det = sxx * syy - sxy * sxy;
                if (sxx > 0 && syy > 0 && det > 1e-6 * sxx * syy && isfinite(cx) && isfinite(cy) && isfinite(det)) {
                    const double k = 1.0 / (1.02 * det);
                    const float A = (float)(syy * k), B = (float)(-sxy * k), C = (float)(sxx * k);
                    // the float32 form the blend kernels evaluate must itself be a usable ellipse (huge needles: A C - B^2 can round to <= 0)
                    // ... and its float32 evaluation at distance D from the centre cancels three terms of size ~ (A + 2|B| + C) D^2 down to ~1:
                    // keep the conic only while that rounding error stays inside the 2 % inflation (D: centre to the far corner of the rect)
                    const float D = sqrtf((float)((cx - pix) * (cx - pix) + (cy - piy) * (cy - piy))) + 1.5f * radius + 8.f;
                    if (A > 0.f && C > 0.f && A * C - B * B > 0.f && isfinite(A) && isfinite(C) && (A + 2.f * fabsf(B) + C) * D * D < 2.5e4f) {
                        cull0 = make_float4((float)cx, (float)cy, A, B);
                        cull1.x = C;
                    }
                }
            }
        } else {
            cull0 = make_float4(pix, piy, 0.f, 0.f);
            cull1 = make_float4(0.f, -1.f, pix, piy);      // r2 < 0: never hits
        }
        if (p.no_cull) { cull0 = make_float4(pix, piy, 0.f, 0.f); cull1 = make_float4(0.f, 1e30f, pix, piy); }
    } while (0);

    p.radii[idx] = radius_out;
    p.g.depth_key[idx] = key;
    tc.tiles = tiles; tc.rect = make_ushort4((unsigned short)x0, (unsigned short)y0, (unsigned short)x1, (unsigned short)y1); tc.c0 = cull0; tc.c1 = cull1;
    p.g.rect[idx] = tc.rect;
    p.g.cull[2 * (size_t)idx] = cull0;
    p.g.cull[2 * (size_t)idx + 1] = cull1;
    p.g.clamped[idx] = clamped;
    float4* rec = p.g.rec + (size_t)idx * GSR_REC_SURFEL;
    rec[0] = make_float4(T[0], T[1], T[2], T[3]);
    rec[1] = make_float4(T[4], T[5], T[6], T[7]);
    rec[2] = make_float4(T[8], pix, piy, o);
    rec[3] = make_float4(normal.x, normal.y, normal.z, rgb.x);
    // D = det[Tu Tv Tw] = p . Tw for every pixel's ray-splat vector p = k x l (k, l differ from -Tu, -Tv by multiples of Tw): the blend kernels take the
    // depth s . Tw.xy + Tw.z of SURFEL forward.cu:367 as D / p.z (gsr_blend_sp.hip sp_surf_setup).  Its terms are of size |Tu.z Tv.z Tw| ~ 1e7 against a
    // result of ~1e4, so it is formed in float64 from the float32 T (9 DP multiplies per visible surfel).
    const double detT = (double)T[0] * ((double)T[4] * T[8] - (double)T[5] * T[7]) - (double)T[1] * ((double)T[3] * T[8] - (double)T[5] * T[6])
                        + (double)T[2] * ((double)T[3] * T[7] - (double)T[4] * T[6]);
    rec[4] = make_float4(rgb.y, rgb.z, (float)detT, 0.f);
    return tc;
}
template <bool SH16>
__global__ void __launch_bounds__(256) k_preprocess_surfel(PreParams p)
{
    if (p.mode_word && blockIdx.x == 0 && threadIdx.x == 0) { *p.mode_word = p.mode; p.mode_word[GSR_CNT_CULL_MISMATCH - GSR_CNT_MODE] = 0u; }
    for (uint32_t z = (uint32_t)(blockIdx.x * blockDim.x + threadIdx.x); z < p.zero_n; z += gridDim.x * blockDim.x) p.zero_ptr[z] = 0u;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const float* sh_row = nullptr;
    if constexpr (SH16) {          // SH coefficients through LDS, as in k_preprocess_ewa_sh16
        __shared__ float s_sh[4 * 64 * GSR_SH_ROW];
        const int lane = threadIdx.x & 63;
        float* my = s_sh + (threadIdx.x >> 6) * 64 * GSR_SH_ROW;
        sh_rows_load(my, p.shs, (size_t)(idx - lane), lane, (size_t)p.P * 48);
        sh_row = my + lane * GSR_SH_ROW;
    }
    PreTc tc; tc.tiles = 0u; tc.rect = make_ushort4(0, 0, 0, 0); tc.c0 = make_float4(0, 0, 0, 0); tc.c1 = make_float4(0, -1, 0, 0);
    if (idx < p.P) tc = pre_surfel_one(p, idx, sh_row);
    pre_block_scan(p, idx, pre_tile_count<GSR_SURFEL>(p, idx, tc));
}

// 3DGS rasterizer_impl.cu:54-66
__global__ void k_mark_visible(int P, const float* means3D, const float* viewp, uint8_t* present)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    float view[16]; load16(viewp, view);
    float3 pv = xform_point4x3(make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]), view);
    present[idx] = !(pv.z <= 0.2f);
}

static PreParams make_params(const gsr_cfg* cfg, const gsr_inputs* in, GeomView g, int32_t* radii)
{
    PreParams p;
    p.P = cfg->P; p.D = cfg->D; p.M = cfg->M; p.W = cfg->W; p.H = cfg->H;
    p.gx = (cfg->W + GSR_TILE - 1) / GSR_TILE; p.gy = (cfg->H + GSR_TILE - 1) / GSR_TILE;
    p.variant = cfg->variant;
    p.tanfovx = cfg->tanfovx; p.tanfovy = cfg->tanfovy;
    p.fy = cfg->H / (2.0f * cfg->tanfovy);       // rasterizer_impl.cu:220-221
    p.fx = cfg->W / (2.0f * cfg->tanfovx);
    p.mod = cfg->scale_modifier;
    p.means3D = in->means3D; p.shs = in->shs; p.colors = in->colors_precomp; p.opac = in->opacities;
    p.scales = in->scales; p.rots = in->rotations; p.cov3D_pre = in->cov3D_precomp; p.all_map = in->all_map;
    p.view = cfg->viewmatrix; p.proj = cfg->projmatrix; p.campos = cfg->campos;
    { const char* e = getenv("GSR_NO_CULL"); p.no_cull = (e && atoi(e) != 0) ? 1 : 0; }
    p.in_mask = nullptr; p.scale_stride = 3;
    p.scan_offsets = nullptr; p.scan_sums = nullptr; p.prefiltered_err = nullptr; p.mode_word = nullptr; p.mode = 0u;
    p.tile_cull = gsr_tile_cull_enabled() ? 1 : 0;
    p.radii = radii; p.g = g;
    p.zero_ptr = nullptr; p.zero_n = 0;
    return p;
}

int gsr_launch_preprocess(const gsr_cfg* cfg, const gsr_inputs* in, GeomView g, int32_t* radii, hipStream_t s, bool global_order, uint32_t* prefiltered_err)
{
    PreParams p = make_params(cfg, in, g, radii);
    p.prefiltered_err = cfg->prefiltered ? prefiltered_err : nullptr;
    p.zero_ptr = g.hist; p.zero_n = global_order ? gsr_sort_group_words((uint32_t)cfg->P, false, 256) : 0u;      // first group histogram of the global depth sort
    p.mode_word = g.counters + GSR_CNT_MODE; p.mode = global_order ? GSR_MODE_GLOBAL : GSR_MODE_TILE;
    // per-tile depth order: instances are emitted in id order, the block-local prefix of tiles_touched is written here (256-gaussian blocks)
    if (!global_order) { p.scan_offsets = g.offsets; p.scan_sums = g.scan_tmp; }
    dim3 grid(gsr_div_up(cfg->P, 256)), block(256);
    // SH colours of degree-3 models: the 16 x 3 coefficients go through LDS (coalesced 16-byte loads); other shapes read them directly
    const bool sh16 = p.shs && !p.colors && cfg->M == 16 && (((uintptr_t)p.shs) & 15) == 0;
    if (cfg->variant == GSR_SURFEL) { if (sh16) hipLaunchKernelGGL(k_preprocess_surfel<true>, grid, block, 0, s, p); else hipLaunchKernelGGL(k_preprocess_surfel<false>, grid, block, 0, s, p); }
    else if (sh16) hipLaunchKernelGGL(k_preprocess_ewa_sh16, grid, block, 0, s, p);
    else hipLaunchKernelGGL(k_preprocess_ewa<false>, grid, block, 0, s, p);
    return gsr_check_launch("preprocess", s, cfg->debug);
}

extern "C" int gsr_visible_filter(const gsr_cfg* cfg, const float* means3D, const float* scales, const float* rotations,
                                  const float* cov3D_precomp, int32_t* radii, void* stream)
{
    if (cfg->P == 0) return 0;
    gsr_inputs in = {};
    in.means3D = means3D; in.scales = scales; in.rotations = rotations; in.cov3D_precomp = cov3D_precomp;
    GeomView g = {};
    gsr_cfg c = *cfg; c.variant = GSR_EWA;
    PreParams p = make_params(&c, &in, g, radii);
    hipLaunchKernelGGL(k_preprocess_ewa<true>, dim3(gsr_div_up(cfg->P, 256)), dim3(256), 0, (hipStream_t)stream, p);
    return gsr_check_launch("visible_filter", (hipStream_t)stream, cfg->debug);
}

// ---- Octree-GS level-of-detail mask (OctreeGaussianModel.set_anchor_mask + map_to_int_level, gssr/gaussian/octree_gaussian.py:184-203,
// 255-267) fused with the frustum/extent prefilter (OctreeScene.prefilter_voxel, gssr/scene/octree_scene.py:136-172): ~20 torch ops, three
// boolean-index gathers (each a host sync) and a masked scatter in the reference; two launches and no sync here.
struct LodArgs {
    int Na; const float* anchor; const int32_t* level; const float* extra_level; const float* campos;
    gsr_lod_cfg c;
    uint8_t* anchor_mask; float* prog_ratio; uint8_t* transition_mask;
};
__global__ void __launch_bounds__(256) k_lod_mask(LodArgs p)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.Na) return;
    const int lvl = p.level[i];
    const float half = (p.c.voxel_size / 2) / powf(p.c.fork, (float)lvl);                   // :256
    const float dx = (p.anchor[3 * i] + half) - p.campos[0], dy = (p.anchor[3 * i + 1] + half) - p.campos[1],
                dz = (p.anchor[3 * i + 2] + half) - p.campos[2];
    const float dist = sqrtf(dx * dx + dy * dy + dz * dz) * p.c.resolution_scale;           // :257
    float pred = log2f(p.c.standard_dist / dist) / log2f(p.c.fork) + (p.extra_level ? p.extra_level[i] : 0.0f);      // :258
    const int cur = p.c.coarse_index - 1;
    int il;
    if (p.c.mode == 3) {                                                                      // 'progressive' :194-198
        pred = fminf(fmaxf(pred + 1.0f, 0.9999f), (float)cur + 0.9999f);
        il = (int)floorf(pred);
        if (p.prog_ratio) p.prog_ratio[i] = pred - truncf(pred);
        if (p.transition_mask) p.transition_mask[i] = (lvl == il) ? 1 : 0;
    } else {
        const float r = p.c.mode == 0 ? floorf(pred) : (p.c.mode == 1 ? rintf(pred) : ceilf(pred));     // torch.round = half to even
        il = (int)r;
        il = il < 0 ? 0 : (il > cur ? cur : il);
    }
    p.anchor_mask[i] = (lvl <= il) ? 1 : 0;                                                   // :267
}

extern "C" int gsr_octree_visible(const gsr_cfg* cfg, const gsr_lod_cfg* lod, const float* anchor, const int32_t* level,
                                  const float* extra_level, const float* scales, int32_t scale_stride, const float* rotations,
                                  uint8_t* anchor_mask, int32_t* radii, float* prog_ratio, uint8_t* transition_mask, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (!cfg || !lod) { gsr_set_error("octree_visible: null cfg"); return 1; }
    if (cfg->P == 0) return 0;
    if (!anchor || !level || !scales || !rotations || !anchor_mask || !radii || scale_stride < 3) {
        gsr_set_error("octree_visible: anchor/level/scales/rotations/anchor_mask/radii must be provided (scale_stride >= 3)"); return 1;
    }
    if (lod->mode < 0 || lod->mode > 3) { gsr_set_error("Unknown dist2level: %d", lod->mode); return 1; }
    LodArgs la;
    la.Na = cfg->P; la.anchor = anchor; la.level = level; la.extra_level = extra_level; la.campos = cfg->campos; la.c = *lod;
    la.anchor_mask = anchor_mask; la.prog_ratio = prog_ratio; la.transition_mask = transition_mask;
    hipLaunchKernelGGL(k_lod_mask, dim3(gsr_div_up(cfg->P, 256)), dim3(256), 0, s, la);
    gsr_inputs in = {};
    in.means3D = anchor; in.scales = scales; in.rotations = rotations;
    GeomView g = {};
    gsr_cfg c = *cfg; c.variant = GSR_EWA;
    PreParams p = make_params(&c, &in, g, radii);
    p.in_mask = anchor_mask; p.scale_stride = scale_stride;
    hipLaunchKernelGGL(k_preprocess_ewa<true>, dim3(gsr_div_up(cfg->P, 256)), dim3(256), 0, s, p);
    return gsr_check_launch("octree_visible", s, cfg->debug);
}

extern "C" int gsr_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                                uint8_t* present, void* stream)
{
    (void)projmatrix;
    if (P == 0) return 0;
    hipLaunchKernelGGL(k_mark_visible, dim3(gsr_div_up(P, 256)), dim3(256), 0, (hipStream_t)stream, P, means3D, viewmatrix, present);
    return gsr_check_launch("mark_visible", (hipStream_t)stream, false);
}

// ================================================================================================ backward
struct PreBwdParams {
    int P, D, M, W, H, variant;
    float tanfovx, tanfovy, fx, fy, mod;
    const float *means3D, *shs, *scales, *rots, *cov3D_pre, *view, *proj, *campos;
    const int32_t* radii;
    const uint32_t* clamped;
    const float4* rec;
    const float* acc;
    float* acc_clear;       // == acc when the accumulator rows are to be left zeroed (gsr_backward_ex GSR_BWD_LEAVE_ZERO), else nullptr
    gsr_in_grads ig;
};
// zero one accumulator row (AS floats, 16-byte aligned: AS is a multiple of 4) after the thread has read it
template <int STRIDE, int USED> __device__ __forceinline__ void clear_acc_row(float* acc_clear, int idx)
{
    float4* r = reinterpret_cast<float4*>(acc_clear + (size_t)idx * STRIDE);
#pragma unroll
    for (int k = 0; k < USED / 4; k++) r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// SH backward, 3DGS backward.cu:20-139.  Writes dL_dsh[idx] and returns the view-direction term of dL/dmean.
// Two phases over the same statements: first everything that READS sh (the direction term), then everything that WRITES dL_dsh -- the two are
// independent, and with the reads done first `sh` and `dL_dsh` may be the same LDS row (k_preprocess_bwd_ewa<true> stages both through one buffer).
template <bool WRITE>
__device__ __forceinline__ void sh_backward_phase(int deg, const float* sh, float x, float y, float z, const float* dRGB, float* dL_dsh,
                                                  float* dRGBdx, float* dRGBdy, float* dRGBdz)
{
#define SHSET(k, coef) do { if constexpr (WRITE) { float cf_ = (coef); for (int c = 0; c < 3; c++) dL_dsh[(k) * 3 + c] = cf_ * dRGB[c]; } } while (0)
    SHSET(0, GSR_SH_C0);
    if (deg > 0) {
        SHSET(1, -GSR_SH_C1 * y); SHSET(2, GSR_SH_C1 * z); SHSET(3, -GSR_SH_C1 * x);
        if constexpr (!WRITE) for (int c = 0; c < 3; c++) {
            dRGBdx[c] = -GSR_SH_C1 * sh[3 * 3 + c]; dRGBdy[c] = -GSR_SH_C1 * sh[1 * 3 + c]; dRGBdz[c] = GSR_SH_C1 * sh[2 * 3 + c];
        }
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            SHSET(4, GSR_SH_C2[0] * xy); SHSET(5, GSR_SH_C2[1] * yz); SHSET(6, GSR_SH_C2[2] * (2.f * zz - xx - yy));
            SHSET(7, GSR_SH_C2[3] * xz); SHSET(8, GSR_SH_C2[4] * (xx - yy));
            if constexpr (!WRITE) for (int c = 0; c < 3; c++) {
                dRGBdx[c] += GSR_SH_C2[0] * y * sh[4 * 3 + c] + GSR_SH_C2[2] * 2.f * -x * sh[6 * 3 + c] + GSR_SH_C2[3] * z * sh[7 * 3 + c] + GSR_SH_C2[4] * 2.f * x * sh[8 * 3 + c];
                dRGBdy[c] += GSR_SH_C2[0] * x * sh[4 * 3 + c] + GSR_SH_C2[1] * z * sh[5 * 3 + c] + GSR_SH_C2[2] * 2.f * -y * sh[6 * 3 + c] + GSR_SH_C2[4] * 2.f * -y * sh[8 * 3 + c];
                dRGBdz[c] += GSR_SH_C2[1] * y * sh[5 * 3 + c] + GSR_SH_C2[2] * 2.f * 2.f * z * sh[6 * 3 + c] + GSR_SH_C2[3] * x * sh[7 * 3 + c];
            }
            if (deg > 2) {
                SHSET(9, GSR_SH_C3[0] * y * (3.f * xx - yy));
                SHSET(10, GSR_SH_C3[1] * xy * z);
                SHSET(11, GSR_SH_C3[2] * y * (4.f * zz - xx - yy));
                SHSET(12, GSR_SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                SHSET(13, GSR_SH_C3[4] * x * (4.f * zz - xx - yy));
                SHSET(14, GSR_SH_C3[5] * z * (xx - yy));
                SHSET(15, GSR_SH_C3[6] * x * (xx - 3.f * yy));
                if constexpr (!WRITE) for (int c = 0; c < 3; c++) {
                    dRGBdx[c] += (GSR_SH_C3[0] * sh[9 * 3 + c] * 3.f * 2.f * xy + GSR_SH_C3[1] * sh[10 * 3 + c] * yz +
                                  GSR_SH_C3[2] * sh[11 * 3 + c] * -2.f * xy + GSR_SH_C3[3] * sh[12 * 3 + c] * -3.f * 2.f * xz +
                                  GSR_SH_C3[4] * sh[13 * 3 + c] * (-3.f * xx + 4.f * zz - yy) + GSR_SH_C3[5] * sh[14 * 3 + c] * 2.f * xz +
                                  GSR_SH_C3[6] * sh[15 * 3 + c] * 3.f * (xx - yy));
                    dRGBdy[c] += (GSR_SH_C3[0] * sh[9 * 3 + c] * 3.f * (xx - yy) + GSR_SH_C3[1] * sh[10 * 3 + c] * xz +
                                  GSR_SH_C3[2] * sh[11 * 3 + c] * (-3.f * yy + 4.f * zz - xx) + GSR_SH_C3[3] * sh[12 * 3 + c] * -3.f * 2.f * yz +
                                  GSR_SH_C3[4] * sh[13 * 3 + c] * -2.f * xy + GSR_SH_C3[5] * sh[14 * 3 + c] * -2.f * yz +
                                  GSR_SH_C3[6] * sh[15 * 3 + c] * -3.f * 2.f * xy);
                    dRGBdz[c] += (GSR_SH_C3[1] * sh[10 * 3 + c] * xy + GSR_SH_C3[2] * sh[11 * 3 + c] * 4.f * 2.f * yz +
                                  GSR_SH_C3[3] * sh[12 * 3 + c] * 3.f * (2.f * zz - xx - yy) + GSR_SH_C3[4] * sh[13 * 3 + c] * 4.f * 2.f * xz +
                                  GSR_SH_C3[5] * sh[14 * 3 + c] * (xx - yy));
                }
            }
        }
    }
#undef SHSET
}
__device__ __forceinline__ float3 sh_backward(int deg, int M, const float* sh, float3 mean, float3 campos, uint32_t clamped,
                                              const float* dL_dcolor, float* dL_dsh)
{
    (void)M;
    float3 dir_orig = make_float3(mean.x - campos.x, mean.y - campos.y, mean.z - campos.z);
    float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
    float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
    float dRGB[3];
#pragma unroll
    for (int c = 0; c < 3; c++) dRGB[c] = dL_dcolor[c] * (((clamped >> c) & 1u) ? 0.0f : 1.0f);
    float dRGBdx[3] = { 0, 0, 0 }, dRGBdy[3] = { 0, 0, 0 }, dRGBdz[3] = { 0, 0, 0 };
    sh_backward_phase<false>(deg, sh, x, y, z, dRGB, dL_dsh, dRGBdx, dRGBdy, dRGBdz);
    sh_backward_phase<true>(deg, sh, x, y, z, dRGB, dL_dsh, dRGBdx, dRGBdy, dRGBdz);
    float3 dL_ddir = make_float3(dot3(dRGBdx, dRGB), dot3(dRGBdy, dRGB), dot3(dRGBdz, dRGB));
    return dnormvdv(dir_orig, dL_ddir);
}

// EWA/PLANE: computeCov2DCUDA + preprocessCUDA + computeCov3D of 3DGS backward.cu:144-396 fused in one pass.
// SH16: the 16 x 3 SH coefficients and their gradients go through one LDS row per gaussian (see k_preprocess_ewa_sh16: coalesced 16-byte-per-lane
// copies instead of 48 loads + 48 stores per lane that each touch 64 cache lines) -- sh_backward reads the row, then overwrites it with dL_dsh.
template <bool SH16>
__global__ void __launch_bounds__(256) k_preprocess_bwd_ewa(PreBwdParams p)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    float* sh_row = nullptr;
    const int lane = threadIdx.x & 63;
    const size_t g0 = (size_t)(idx - lane);
    if constexpr (SH16) {
        __shared__ float s_sh[4 * 64 * GSR_SH_ROW];
        float* my = s_sh + (threadIdx.x >> 6) * 64 * GSR_SH_ROW;
        sh_row = my + lane * GSR_SH_ROW;
        sh_rows_load(my, p.shs, g0, lane, (size_t)p.P * 48);
    }
    auto body = [&]() {
    const int AS = (p.variant == GSR_PLANE) ? GSR_ACC_PLANE : GSR_ACC_EWA;
    const float* a = p.acc + (size_t)idx * AS;
    const float dcol[3] = { a[0], a[1], a[2] };
    const float dop = a[3], dm2x = a[4], dm2y = a[5];
    const float dcx = a[6], dcy = a[7], dcz = a[8];
    gsr_in_grads& ig = p.ig;
    ig.dL_dcolors[3 * idx] = dcol[0]; ig.dL_dcolors[3 * idx + 1] = dcol[1]; ig.dL_dcolors[3 * idx + 2] = dcol[2];
    ig.dL_dopacity[idx] = dop;
    ig.dL_dmeans2D[3 * idx] = dm2x; ig.dL_dmeans2D[3 * idx + 1] = dm2y; ig.dL_dmeans2D[3 * idx + 2] = 0.f;
    if (p.variant == GSR_PLANE) {
        if (ig.dL_dmeans2D_abs) { ig.dL_dmeans2D_abs[3 * idx] = a[9]; ig.dL_dmeans2D_abs[3 * idx + 1] = a[10]; ig.dL_dmeans2D_abs[3 * idx + 2] = 0.f; }
        if (ig.dL_dall_map) for (int c = 0; c < 5; c++) ig.dL_dall_map[5 * idx + c] = a[11 + c];
    }
    float dmean[3] = { 0, 0, 0 }, dcov[6] = { 0, 0, 0, 0, 0, 0 }, dsc[3] = { 0, 0, 0 }, drot[4] = { 0, 0, 0, 0 };
    const bool vis = p.radii[idx] > 0;
    if (vis) {
        float view[16], proj[16];
        load16(p.view, view); load16(p.proj, proj);
        const float3 mean = make_float3(p.means3D[3 * idx], p.means3D[3 * idx + 1], p.means3D[3 * idx + 2]);
        float cov3D[6];
        if (p.cov3D_pre) { for (int i = 0; i < 6; i++) cov3D[i] = p.cov3D_pre[6 * idx + i]; }
        else cov3d_from_scale_rot(p.scales + 3 * idx, p.mod, p.rots + 4 * idx, cov3D);
        // ---- conic -> cov2D -> cov3D, mean (backward.cu:144-274)
        Cov2D cv = cov2d_project(mean, p.fx, p.fy, p.tanfovx, p.tanfovy, cov3D, view);
        const M3& T = cv.T; const M3& Vrk = cv.Vrk;
        M3 Wm = m3_cols(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
        const float ca = cv.a, cb = cv.b, cc = cv.c;
        float denom = ca * cc - cb * cb;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        if (denom2inv != 0) {
            dL_da = denom2inv * (-cc * cc * dcx + 2 * cb * cc * dcy + (denom - ca * cc) * dcz);
            dL_dc = denom2inv * (-ca * ca * dcz + 2 * ca * cb * dcy + (denom - ca * cc) * dcx);
            dL_db = denom2inv * 2 * (cb * cc * dcx - (denom + 2 * cb * cb) * dcy + ca * cb * dcz);
            dcov[0] = (T.c[0][0] * T.c[0][0] * dL_da + T.c[0][0] * T.c[1][0] * dL_db + T.c[1][0] * T.c[1][0] * dL_dc);
            dcov[3] = (T.c[0][1] * T.c[0][1] * dL_da + T.c[0][1] * T.c[1][1] * dL_db + T.c[1][1] * T.c[1][1] * dL_dc);
            dcov[5] = (T.c[0][2] * T.c[0][2] * dL_da + T.c[0][2] * T.c[1][2] * dL_db + T.c[1][2] * T.c[1][2] * dL_dc);
            dcov[1] = 2 * T.c[0][0] * T.c[0][1] * dL_da + (T.c[0][0] * T.c[1][1] + T.c[0][1] * T.c[1][0]) * dL_db + 2 * T.c[1][0] * T.c[1][1] * dL_dc;
            dcov[2] = 2 * T.c[0][0] * T.c[0][2] * dL_da + (T.c[0][0] * T.c[1][2] + T.c[0][2] * T.c[1][0]) * dL_db + 2 * T.c[1][0] * T.c[1][2] * dL_dc;
            dcov[4] = 2 * T.c[0][2] * T.c[0][1] * dL_da + (T.c[0][1] * T.c[1][2] + T.c[0][2] * T.c[1][1]) * dL_db + 2 * T.c[1][1] * T.c[1][2] * dL_dc;
        }
        float dL_dT00 = 2 * (T.c[0][0] * Vrk.c[0][0] + T.c[0][1] * Vrk.c[0][1] + T.c[0][2] * Vrk.c[0][2]) * dL_da + (T.c[1][0] * Vrk.c[0][0] + T.c[1][1] * Vrk.c[0][1] + T.c[1][2] * Vrk.c[0][2]) * dL_db;
        float dL_dT01 = 2 * (T.c[0][0] * Vrk.c[1][0] + T.c[0][1] * Vrk.c[1][1] + T.c[0][2] * Vrk.c[1][2]) * dL_da + (T.c[1][0] * Vrk.c[1][0] + T.c[1][1] * Vrk.c[1][1] + T.c[1][2] * Vrk.c[1][2]) * dL_db;
        float dL_dT02 = 2 * (T.c[0][0] * Vrk.c[2][0] + T.c[0][1] * Vrk.c[2][1] + T.c[0][2] * Vrk.c[2][2]) * dL_da + (T.c[1][0] * Vrk.c[2][0] + T.c[1][1] * Vrk.c[2][1] + T.c[1][2] * Vrk.c[2][2]) * dL_db;
        float dL_dT10 = 2 * (T.c[1][0] * Vrk.c[0][0] + T.c[1][1] * Vrk.c[0][1] + T.c[1][2] * Vrk.c[0][2]) * dL_dc + (T.c[0][0] * Vrk.c[0][0] + T.c[0][1] * Vrk.c[0][1] + T.c[0][2] * Vrk.c[0][2]) * dL_db;
        float dL_dT11 = 2 * (T.c[1][0] * Vrk.c[1][0] + T.c[1][1] * Vrk.c[1][1] + T.c[1][2] * Vrk.c[1][2]) * dL_dc + (T.c[0][0] * Vrk.c[1][0] + T.c[0][1] * Vrk.c[1][1] + T.c[0][2] * Vrk.c[1][2]) * dL_db;
        float dL_dT12 = 2 * (T.c[1][0] * Vrk.c[2][0] + T.c[1][1] * Vrk.c[2][1] + T.c[1][2] * Vrk.c[2][2]) * dL_dc + (T.c[0][0] * Vrk.c[2][0] + T.c[0][1] * Vrk.c[2][1] + T.c[0][2] * Vrk.c[2][2]) * dL_db;
        float dL_dJ00 = Wm.c[0][0] * dL_dT00 + Wm.c[0][1] * dL_dT01 + Wm.c[0][2] * dL_dT02;
        float dL_dJ02 = Wm.c[2][0] * dL_dT00 + Wm.c[2][1] * dL_dT01 + Wm.c[2][2] * dL_dT02;
        float dL_dJ11 = Wm.c[1][0] * dL_dT10 + Wm.c[1][1] * dL_dT11 + Wm.c[1][2] * dL_dT12;
        float dL_dJ12 = Wm.c[2][0] * dL_dT10 + Wm.c[2][1] * dL_dT11 + Wm.c[2][2] * dL_dT12;
        float tz = 1.f / cv.t.z, tz2 = tz * tz, tz3 = tz2 * tz;
        float dL_dtx = cv.xgm * -p.fx * tz2 * dL_dJ02;
        float dL_dty = cv.ygm * -p.fy * tz2 * dL_dJ12;
        float dL_dtz = -p.fx * tz2 * dL_dJ00 - p.fy * tz2 * dL_dJ11 + (2 * p.fx * cv.t.x) * tz3 * dL_dJ02 + (2 * p.fy * cv.t.y) * tz3 * dL_dJ12;
        float3 dm = xform_vec4x3_t(make_float3(dL_dtx, dL_dty, dL_dtz), view);
        dmean[0] = dm.x; dmean[1] = dm.y; dmean[2] = dm.z;
        // ---- projection term (backward.cu:369-387)
        float4 mh = xform_point4x4(mean, proj);
        float m_w = 1.0f / (mh.w + 0.0000001f);
        float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
        dmean[0] += (proj[0] * m_w - proj[3] * mul1) * dm2x + (proj[1] * m_w - proj[3] * mul2) * dm2y;
        dmean[1] += (proj[4] * m_w - proj[7] * mul1) * dm2x + (proj[5] * m_w - proj[7] * mul2) * dm2y;
        dmean[2] += (proj[8] * m_w - proj[11] * mul1) * dm2x + (proj[9] * m_w - proj[11] * mul2) * dm2y;
        // ---- SH (backward.cu:389-391)
        if (p.shs) {
            float* dsh = SH16 ? sh_row : ig.dL_dsh + (size_t)idx * p.M * 3;
            float3 ds = sh_backward(p.D, p.M, SH16 ? sh_row : p.shs + (size_t)idx * p.M * 3, mean, make_float3(p.campos[0], p.campos[1], p.campos[2]),
                                    p.clamped[idx], dcol, dsh);
            // coefficients above the active degree are never written by sh_backward: zero them
            const int used = (p.D + 1) * (p.D + 1);
            for (int k = used; k < p.M; k++) for (int c = 0; c < 3; c++) dsh[k * 3 + c] = 0.f;
            dmean[0] += ds.x; dmean[1] += ds.y; dmean[2] += ds.z;
        }
        // ---- cov3D -> scale, rotation (backward.cu:278-341; no quaternion-normalisation Jacobian)
        if (p.scales) {
            const float* sc = p.scales + 3 * idx; const float* q = p.rots + 4 * idx;
            float r = q[0], x = q[1], y = q[2], z = q[3];
            M3 R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                           2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                           2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
            float s[3] = { p.mod * sc[0], p.mod * sc[1], p.mod * sc[2] };
            M3 S = m3_cols(s[0], 0, 0, 0, s[1], 0, 0, 0, s[2]);
            M3 Mm = m3_mul(S, R);
            M3 dSg = m3_cols(dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3], 0.5f * dcov[4], 0.5f * dcov[2], 0.5f * dcov[4], dcov[5]);
            M3 M2 = Mm;
            for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) M2.c[j][i] = 2.0f * Mm.c[j][i];
            M3 dM = m3_mul(M2, dSg);
            M3 Rt = m3_t(R), dMt = m3_t(dM);
            dsc[0] = dot3(Rt.c[0], dMt.c[0]); dsc[1] = dot3(Rt.c[1], dMt.c[1]); dsc[2] = dot3(Rt.c[2], dMt.c[2]);
            for (int i = 0; i < 3; i++) { dMt.c[0][i] *= s[0]; dMt.c[1][i] *= s[1]; dMt.c[2][i] *= s[2]; }
            drot[0] = 2 * z * (dMt.c[0][1] - dMt.c[1][0]) + 2 * y * (dMt.c[2][0] - dMt.c[0][2]) + 2 * x * (dMt.c[1][2] - dMt.c[2][1]);
            drot[1] = 2 * y * (dMt.c[1][0] + dMt.c[0][1]) + 2 * z * (dMt.c[2][0] + dMt.c[0][2]) + 2 * r * (dMt.c[1][2] - dMt.c[2][1]) - 4 * x * (dMt.c[2][2] + dMt.c[1][1]);
            drot[2] = 2 * x * (dMt.c[1][0] + dMt.c[0][1]) + 2 * r * (dMt.c[2][0] - dMt.c[0][2]) + 2 * z * (dMt.c[1][2] + dMt.c[2][1]) - 4 * y * (dMt.c[2][2] + dMt.c[0][0]);
            drot[3] = 2 * r * (dMt.c[0][1] - dMt.c[1][0]) + 2 * x * (dMt.c[2][0] + dMt.c[0][2]) + 2 * y * (dMt.c[1][2] + dMt.c[2][1]) - 4 * z * (dMt.c[1][1] + dMt.c[0][0]);
        }
    } else if (p.shs && ig.dL_dsh) {
        float* dsh = SH16 ? sh_row : ig.dL_dsh + (size_t)idx * p.M * 3;
        for (int k = 0; k < p.M * 3; k++) dsh[k] = 0.f;
    }
    for (int i = 0; i < 3; i++) ig.dL_dmeans3D[3 * idx + i] = dmean[i];
    for (int i = 0; i < 6; i++) ig.dL_dcov3D[6 * idx + i] = dcov[i];
    for (int i = 0; i < 3; i++) ig.dL_dscales[3 * idx + i] = dsc[i];
    for (int i = 0; i < 4; i++) ig.dL_drotations[4 * idx + i] = drot[i];
    if (p.acc_clear) { if (p.variant == GSR_PLANE) clear_acc_row<GSR_ACC_PLANE, GSR_ACC_USED_PLANE>(p.acc_clear, idx); else clear_acc_row<GSR_ACC_EWA, GSR_ACC_USED_EWA>(p.acc_clear, idx); }
    };
    if (idx < p.P) body();
    if constexpr (SH16) sh_rows_store(sh_row - lane * GSR_SH_ROW, p.ig.dL_dsh, g0, lane, (size_t)p.P * 48);      // the wave's 64 rows of dL_dsh
}

// SURFEL backward.cu:450-637
template <bool SH16>
__global__ void __launch_bounds__(256) k_preprocess_bwd_surfel(PreBwdParams p)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    float* sh_row = nullptr;
    const int lane = threadIdx.x & 63;
    const size_t g0 = (size_t)(idx - lane);
    if constexpr (SH16) {          // SH coefficients and their gradients through one LDS row per gaussian, as in k_preprocess_bwd_ewa<true>
        __shared__ float s_sh[4 * 64 * GSR_SH_ROW];
        float* my = s_sh + (threadIdx.x >> 6) * 64 * GSR_SH_ROW;
        sh_row = my + lane * GSR_SH_ROW;
        sh_rows_load(my, p.shs, g0, lane, (size_t)p.P * 48);
    }
    auto body = [&]() {
    const float* a = p.acc + (size_t)idx * GSR_ACC_SURFEL;
    const float dcol[3] = { a[0], a[1], a[2] };
    const float dop = a[3];
    // accumulator layout (gsr_blend.hip): 0-2 colour, 3 opacity, 4-6 normal, 7-15 transMat, 16-17 mean2D
    float gmx = a[16], gmy = a[17];
    const float dnrm[3] = { a[4], a[5], a[6] };
    float dT[3][3];
    for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) dT[j][i] = a[7 + 3 * j + i];
    gsr_in_grads& ig = p.ig;
    ig.dL_dcolors[3 * idx] = dcol[0]; ig.dL_dcolors[3 * idx + 1] = dcol[1]; ig.dL_dcolors[3 * idx + 2] = dcol[2];
    ig.dL_dopacity[idx] = dop;
    float dmean[3] = { 0, 0, 0 }, dsc[2] = { 0, 0 }, drot[4] = { 0, 0, 0, 0 };
    float dTout[9];
    for (int k = 0; k < 9; k++) dTout[k] = a[7 + k];
    float m2x = gmx, m2y = gmy;     // value returned for dL_dmeans2D when the gaussian is not visible (always 0 then)
    const bool vis = p.radii[idx] > 0;
    if (vis) {
        float view[16], proj[16];
        load16(p.view, view); load16(p.proj, proj);
        // fork quirk: W,H re-derived in float32 (backward.cu:614-615)
        const int Wb = (int)(p.fx * p.tanfovx * 2), Hb = (int)(p.fy * p.tanfovy * 2);
        const bool precomp = (p.scales == nullptr);
        float T[9]; float3 normal = make_float3(0, 0, 0); M3 R = {};
        float3 p_orig = make_float3(0, 0, 0);
        if (precomp) { for (int i = 0; i < 9; i++) T[i] = p.cov3D_pre[9 * idx + i]; }
        else {
            p_orig = make_float3(p.means3D[3 * idx], p.means3D[3 * idx + 1], p.means3D[3 * idx + 2]);
            // fork quirk (backward.cu:488): scale_modifier not applied here
            surfel_transmat(p_orig, p.scales + 2 * idx, 1.0f, p.rots + 4 * idx, proj, view, Wb, Hb, T, normal, R);
        }
        bool done = false;
        if (gmx != 0 || gmy != 0) {
            const float* Tu = T; const float* Tv = T + 3; const float* Tw = T + 6;
            float tv[3] = { 9.0f, 9.0f, -1.0f };
            float ww[3] = { Tw[0] * Tw[0], Tw[1] * Tw[1], Tw[2] * Tw[2] };
            float d = dot3(tv, ww);
            float inv = 1.0f / d;
            float f[3] = { tv[0] * inv, tv[1] * inv, tv[2] * inv };
            float dT0[3], dT1[3], dT3[3], dL_df[3];
            for (int i = 0; i < 3; i++) {
                dT0[i] = gmx * f[i] * Tw[i];
                dT1[i] = gmy * f[i] * Tw[i];
                dT3[i] = gmx * f[i] * Tu[i] + gmy * f[i] * Tv[i];
                dL_df[i] = gmx * Tu[i] * Tw[i] + gmy * Tv[i] * Tw[i];
            }
            float dL_dd = (float)(dot3(dL_df, f) * (-1.0 / d));
            for (int i = 0; i < 3; i++) {
                float dd_dT3 = tv[i] * Tw[i] * 2.0f;
                dT3[i] += dL_dd * dd_dT3;
                dT[0][i] += dT0[i]; dT[1][i] += dT1[i]; dT[2][i] += dT3[i];
            }
            if (precomp) {
                for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) dTout[3 * j + i] = dT[j][i];
                done = true;
            }
        }
        if (!precomp && !done) {
            float n00 = (float)((float)Wb / 2.0), n03 = (float)((float)(Wb - 1) / 2.0);
            float n11 = (float)((float)Hb / 2.0), n13 = (float)((float)(Hb - 1) / 2.0);
            float dM[3][4];
            for (int k = 0; k < 4; k++) {
                float a0 = proj[4 * k + 0], a1 = proj[4 * k + 1], a2 = proj[4 * k + 2], a3 = proj[4 * k + 3];
                float P0 = a0 * n00 + a1 * 0.0f + a2 * 0.0f + a3 * n03;
                float P1 = a0 * 0.0f + a1 * n11 + a2 * 0.0f + a3 * n13;
                float P2 = a0 * 0.0f + a1 * 0.0f + a2 * 0.0f + a3 * 1.0f;
                for (int j = 0; j < 3; j++) dM[j][k] = P0 * dT[0][j] + P1 * dT[1][j] + P2 * dT[2][j];
            }
            float3 dtn = xform_vec4x3_t(make_float3(dnrm[0], dnrm[1], dnrm[2]), view);
            float3 pv = xform_point4x3(p_orig, view);
            float cosv = -(pv.x * normal.x + pv.y * normal.y + pv.z * normal.z);
            float mult = cosv > 0 ? 1.f : -1.f;
            dtn = make_float3(mult * dtn.x, mult * dtn.y, mult * dtn.z);
            const float* sc = p.scales + 2 * idx;
            M3 dRS = m3_cols(dM[0][0], dM[0][1], dM[0][2], dM[1][0], dM[1][1], dM[1][2], dtn.x, dtn.y, dtn.z);
            M3 dR = m3_cols(dRS.c[0][0] * sc[0], dRS.c[0][1] * sc[0], dRS.c[0][2] * sc[0],
                            dRS.c[1][0] * sc[1], dRS.c[1][1] * sc[1], dRS.c[1][2] * sc[1],
                            dRS.c[2][0], dRS.c[2][1], dRS.c[2][2]);
            // quat_to_rotmat_vjp (auxiliary.h:241-284): w.r.t. the normalised quaternion, no normalisation Jacobian
            const float* q = p.rots + 4 * idx;
            float s = 1.0f / sqrtf(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
            float w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
            drot[0] = 2.f * (x * (dR.c[1][2] - dR.c[2][1]) + y * (dR.c[2][0] - dR.c[0][2]) + z * (dR.c[0][1] - dR.c[1][0]));
            drot[1] = 2.f * (-2.f * x * (dR.c[1][1] + dR.c[2][2]) + y * (dR.c[0][1] + dR.c[1][0]) + z * (dR.c[0][2] + dR.c[2][0]) + w * (dR.c[1][2] - dR.c[2][1]));
            drot[2] = 2.f * (x * (dR.c[0][1] + dR.c[1][0]) - 2.f * y * (dR.c[0][0] + dR.c[2][2]) + z * (dR.c[1][2] + dR.c[2][1]) + w * (dR.c[2][0] - dR.c[0][2]));
            drot[3] = 2.f * (x * (dR.c[0][2] + dR.c[2][0]) + y * (dR.c[1][2] + dR.c[2][1]) - 2.f * z * (dR.c[0][0] + dR.c[1][1]) + w * (dR.c[0][1] - dR.c[1][0]));
            dsc[0] = dot3(dRS.c[0], R.c[0]);
            dsc[1] = dot3(dRS.c[1], R.c[1]);
            dmean[0] = dM[2][0]; dmean[1] = dM[2][1]; dmean[2] = dM[2][2];
        }
        if (p.shs) {
            float3 mean = make_float3(p.means3D[3 * idx], p.means3D[3 * idx + 1], p.means3D[3 * idx + 2]);
            float* dsh = SH16 ? sh_row : ig.dL_dsh + (size_t)idx * p.M * 3;
            float3 ds = sh_backward(p.D, p.M, SH16 ? sh_row : p.shs + (size_t)idx * p.M * 3, mean, make_float3(p.campos[0], p.campos[1], p.campos[2]),
                                    p.clamped[idx], dcol, dsh);
            const int used = (p.D + 1) * (p.D + 1);
            for (int k = used; k < p.M; k++) for (int c = 0; c < 3; c++) dsh[k * 3 + c] = 0.f;
            dmean[0] += ds.x; dmean[1] += ds.y; dmean[2] += ds.z;
        }
        // densification proxy (backward.cu:633-636): uses the transMat the blend used and the (possibly updated) dL_dtransMat
        const float depth = p.rec[(size_t)idx * GSR_REC_SURFEL + 2].x;
        m2x = (float)(dTout[2] * depth * 0.5 * (float)Wb);
        m2y = (float)(dTout[5] * depth * 0.5 * (float)Hb);
    } else if (p.shs && ig.dL_dsh) {
        float* dsh = SH16 ? sh_row : ig.dL_dsh + (size_t)idx * p.M * 3;
        for (int k = 0; k < p.M * 3; k++) dsh[k] = 0.f;
    }
    ig.dL_dmeans2D[3 * idx] = m2x; ig.dL_dmeans2D[3 * idx + 1] = m2y; ig.dL_dmeans2D[3 * idx + 2] = 0.f;
    for (int i = 0; i < 3; i++) ig.dL_dmeans3D[3 * idx + i] = dmean[i];
    for (int i = 0; i < 9; i++) ig.dL_dcov3D[9 * idx + i] = dTout[i];
    ig.dL_dscales[2 * idx] = dsc[0]; ig.dL_dscales[2 * idx + 1] = dsc[1];
    for (int i = 0; i < 4; i++) ig.dL_drotations[4 * idx + i] = drot[i];
    if (p.acc_clear) clear_acc_row<GSR_ACC_SURFEL, GSR_ACC_USED_SURFEL>(p.acc_clear, idx);
    };
    if (idx < p.P) body();
    if constexpr (SH16) sh_rows_store(sh_row - lane * GSR_SH_ROW, p.ig.dL_dsh, g0, lane, (size_t)p.P * 48);
}

int gsr_launch_preprocess_bwd(const gsr_cfg* cfg, const gsr_inputs* in, const int32_t* radii, GeomView g,
                              float* acc, const gsr_in_grads* ig, bool leave_zero, hipStream_t s)
{
    PreBwdParams p;
    p.P = cfg->P; p.D = cfg->D; p.M = cfg->M; p.W = cfg->W; p.H = cfg->H; p.variant = cfg->variant;
    p.tanfovx = cfg->tanfovx; p.tanfovy = cfg->tanfovy;
    p.fy = cfg->H / (2.0f * cfg->tanfovy);
    p.fx = cfg->W / (2.0f * cfg->tanfovx);
    p.mod = cfg->scale_modifier;
    p.means3D = in->means3D; p.shs = in->shs; p.scales = in->scales; p.rots = in->rotations; p.cov3D_pre = in->cov3D_precomp;
    p.view = cfg->viewmatrix; p.proj = cfg->projmatrix; p.campos = cfg->campos;
    p.radii = radii; p.clamped = g.clamped; p.rec = g.rec; p.acc = acc; p.acc_clear = leave_zero ? acc : nullptr; p.ig = *ig;
    dim3 grid(gsr_div_up(cfg->P, 256)), block(256);
    const bool sh16 = p.shs && p.ig.dL_dsh && cfg->M == 16 && ((((uintptr_t)p.shs) | ((uintptr_t)p.ig.dL_dsh)) & 15) == 0;
    if (cfg->variant == GSR_SURFEL) { if (sh16) hipLaunchKernelGGL(k_preprocess_bwd_surfel<true>, grid, block, 0, s, p); else hipLaunchKernelGGL(k_preprocess_bwd_surfel<false>, grid, block, 0, s, p); }
    else if (sh16) hipLaunchKernelGGL(k_preprocess_bwd_ewa<true>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(k_preprocess_bwd_ewa<false>, grid, block, 0, s, p);
    return gsr_check_launch("preprocess_bwd", s, cfg->debug);
}
