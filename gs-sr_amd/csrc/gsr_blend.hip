// gsr_blend.hip -- per-tile alpha blend, forward and backward, for the EWA / PLANE / SURFEL variants.
//
// CDNA4 design (not the reference's 16x16-thread / shared-memory-batch / __syncthreads scheme):
//   * one 64-lane wavefront owns an 8x8 pixel sub-tile (4 waves = one 16x16 tile, same tile list, NO barriers,
//     each wave terminates on its own as soon as its 64 pixels are saturated);
//   * the tile's depth-sorted splat list is consumed 64 at a time: lane l fetches instance l's conservative
//     screen box, tests it against the wave's sub-tile, and a 64-bit ballot gives the queue of splats that can
//     touch this sub-tile at all -- non-contributing (pixel, splat) pairs are skipped a whole wave at a time;
//   * forward: the surviving candidates' packed records are staged in wave-private LDS by the lanes that tested them and read back
//     through a wave-uniform address (broadcast ds_read_b128), so the blend maths has VGPR operands only; the pixel-parallel backward
//     (GSR_BWD=px) fetches them with scalar loads (3-5 s_load_dwordx4 per pair);
//   * backward: per-pixel partial gradients are summed across the wave with DPP row reductions and ONE lane issues
//     the atomics into a packed per-gaussian accumulator (<= 20 floats, one or two cache lines) -- 64x fewer atomics
//     than one-atomic-per-pixel.
// Behaviour follows 3DGS forward.cu:261-374 / backward.cu:399-557, PLANE forward.cu:273-407 / backward.cu:399-614,
// SURFEL forward.cu:256-448 / backward.cu:143-447 (thresholds, ordering, recurrences); see DESIGN.md.
#include "gsr_blend_common.h"
#include <algorithm>
#include "gsr_tile_sort.h"

// =================================================================================================== forward
template <int V>
__global__ void __launch_bounds__(256) k_blend_fwd(BlendParams p)
{
    constexpr int ST = (V == GSR_EWA) ? GSR_REC_EWA : (V == GSR_PLANE ? GSR_REC_PLANE : GSR_REC_SURFEL);
    // [wave][record quarter][candidate slot]: private to the wave, no barrier.  20 KB for every variant (SURFEL needs them; eight workgroups per CU --
    // the wave limit -- fit either way): the sort prologue borrows the buffer and takes lists of up to 2048 entries in LDS instead of sending EWA's
    // beyond 1024 to the global-memory radix path (EWA at P = 1.5 M: 624 -> 663 it/s; neutral at 300k: blend forward 0.2207 vs 0.2204 ms)
    __shared__ float4 s_rec[(4 * ST * 64 > 1280) ? 4 * ST * 64 : 1280];
    const int tile = tile_of_block(blockIdx.x, p.gx * p.gy, p.xcd_remap, p.tile_order, p.static_map);
    const int tx = tile % p.gx, ty = tile / p.gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ox = tx * GSR_TILE + (wave & 1) * GSR_SUB, oy = ty * GSR_TILE + (wave >> 1) * GSR_SUB;
    const uint2 range = p.ranges[tile];
    if (p.depth_key) {
        // per-tile depth order, fused: the four waves put the tile's list in (depth, id) order before any of them blends (the only barriers of
        // the kernel; s_rec is free until the first batch is staged).  The backward reads the same list afterwards.
        TdsScratch sc; sc.tile_keys = p.tile_keys; sc.keys = p.scratch_keys; sc.ids = p.scratch_ids;
        tds_sort_tile_wg<(int)sizeof(s_rec)>(s_rec, p.list_rw + range.x, range.y > range.x ? range.y - range.x : 0u, range.x, (uint32_t)tile, p.depth_key, sc, p.sort_buckets != 0, p.list_any_order != 0);
    }
    if (p.status && blockIdx.x == 0 && threadIdx.x == 0) { const uint32_t R = *p.status_total; p.status[0] = R; if (R > p.status_cap) p.status[1] = 1u; }
    if (p.long_word && threadIdx.x == 0 && range.y > range.x && range.y - range.x > p.long_len) *p.long_word = range.y - range.x;      // feedback for the launch order
    if (ox >= p.W || oy >= p.H) return;                 // wave-uniform: this sub-tile is outside the image
    const int px = ox + (lane & 7), py = oy + (lane >> 3);
    const bool inside = px < p.W && py < p.H;
    const float pxf = (float)px, pyf = (float)py;
    const float fdx = (float)(lane & 7), fdy = (float)(lane >> 3);      // SURFEL: pixel offset inside the sub-tile
    const size_t HW = (size_t)p.W * p.H;
    const uint32_t pix_id = (uint32_t)p.W * py + px;

    float T = 1.0f;
    uint32_t last_contributor = 0;
    float C0 = 0, C1 = 0, C2 = 0;
    // Which pixels still blend, and every gate of the pair loop, are kept as 64-bit wave masks in SCALAR registers: the compares write them directly
    // (v_cmp -> SGPR pair, __builtin_amdgcn_fcmpf), they are combined on the scalar unit, and the lanes that apply the splat are selected through
    // inverse_ballot.  With `done` as a loop-carried bool the early-exit test __ballot(!done) cost a v_cndmask + v_cmp per pair on top of the
    // scalar bookkeeping of the bool's phi.  Predicates (LLVM fcmp numbering): 2 ogt, 4 olt, 5 ole, 11 uge, 13 ule, 14 une -- the unordered
    // forms reproduce the reference's negated tests (`!(a < b)` is true for NaN).
    uint64_t live = __ballot(inside);
    // SURFEL
    float N0 = 0, N1 = 0, N2 = 0, Dd = 0, M1 = 0, M2 = 0, distortion = 0, median_depth = 0;
    uint32_t median_contributor = 0; int surf_idx = -1;
    float mn0 = 0, mn1 = 0, mn2 = 0;
    // PLANE
    float A0 = 0, A1 = 0, A2 = 0, A3 = 0, A4 = 0;

    for (uint32_t base = range.x; base < range.y; base += GSR_WAVE) {
        if (live == 0) break;
        const uint32_t i = base + lane;
        const bool v = i < range.y;
        const uint32_t id = v ? p.point_list[i] : 0u;
        const bool hit = v && cull_hit<V>(p.cull, id, (float)ox, (float)oy);
        uint64_t m = __ballot(hit);
        // the ballot is what the splat-parallel backward needs to know about this (batch, quadrant): keep it (8 bytes per 64 entries)
        if (p.qmask && lane == 0) p.qmask[(((size_t)(range.x >> 6) + (size_t)tile + ((base - range.x) >> 6)) << 2) + wave] = m;
        // Every lane whose candidate survives the cull stages that candidate's packed record in the wave's LDS slots: one burst of vector
        // loads per 64 candidates instead of a scalar-load round trip per pair.  The pair loop reads the record back through a wave-uniform
        // LDS address (broadcast ds_read_b128), so the blend maths runs on VGPR operands only -- on gfx950 a VALU instruction with an SGPR
        // operand issues in ~4.2 cycles against ~2.7 for VGPR-only fma / mul / add (tools/microbench/valu_rate.hip), and the 6 v_mov the
        // two-SGPR fmas of the surfel intersection needed are gone.  Measured: surfel 0.234 -> 0.201 ms, EWA 0.233 -> 0.189, PLANE 0.199 -> 0.176.
#ifndef FWD_DIAG_NO_STAGE
        if (hit) {
#else
        if (hit && p.W < 0) {
#endif
            const float4* __restrict__ rr = p.rec + (size_t)id * ST;
            if constexpr (V == GSR_SURFEL) {
                // SURFEL: the staging lane also does the per-(splat, sub-tile) part of the ray-splat intersection.  p = k x l (k = px Tw - Tu, l = py Tw - Tv,
                // SURFEL forward.cu:351-357) is affine in the pixel: with (ox, oy) the sub-tile's first pixel, k0 = ox Tw - Tu, l0 = oy Tw - Tv,
                //     p(ox + dx, oy + dy) = k0 x l0 + dx (Tw x l0) + dy (k0 x Tw),
                // and the depth s . Tw.xy + Tw.z equals det[Tu Tv Tw] / p.z (record word D, gsr_preprocess.hip).  Staged: {P0, Px, Py, D, Tw.z, opacity,
                // centre - origin, normal, rgb} = 20 floats, so that a pixel pays 6 FMAs + 1 multiply for p and the depth instead of 15 instructions.
                const float4 r0 = rr[0], r1 = rr[1], r2 = rr[2], r3 = rr[3], r4 = rr[4];
                const float fox = (float)ox, foy = (float)oy;
                const float Tw0 = r1.z, Tw1 = r1.w, Tw2 = r2.x;
                const float kx = fox * Tw0 - r0.x, ky = fox * Tw1 - r0.y, kz = fox * Tw2 - r0.z;
                const float lx = foy * Tw0 - r0.w, ly = foy * Tw1 - r1.x, lz = foy * Tw2 - r1.y;
                float4* dst = s_rec + wave * ST * 64 + lane;
                dst[0 * 64] = make_float4(ky * lz - kz * ly, kz * lx - kx * lz, kx * ly - ky * lx, Tw1 * lz - Tw2 * ly);
                dst[1 * 64] = make_float4(Tw2 * lx - Tw0 * lz, Tw0 * ly - Tw1 * lx, ky * Tw2 - kz * Tw1, kz * Tw0 - kx * Tw2);
                dst[2 * 64] = make_float4(kx * Tw1 - ky * Tw0, r4.z, Tw2, r2.w);
                dst[3 * 64] = make_float4(r2.y - fox, r2.z - foy, r3.x, r3.y);
                dst[4 * 64] = make_float4(r3.z, r3.w, r4.x, r4.y);
            } else {
                // (staging the conic times log2(e), so that a pair pays exp2 instead of a multiply and a v_exp_f32: PLANE 0.1705 -> 0.168 ms, EWA 0.1695 -> 0.171, and
                // the three separately rounded products break the cancellation inside `power` of x20 needle splats -- test_needle_splats_are_not_culled_away;
                // removed, EXPERIMENTS.md (74))
#pragma unroll
                for (int k = 0; k < ST; k++) s_rec[(wave * ST + k) * 64 + lane] = rr[k];      // [wave][k][slot]: lane-contiguous 16-byte stores
            }
        }
#ifdef FWD_DIAG_NO_PAIRS      // diagnostic build only: the kernel without its pair loop (results are then wrong)
        m = 0;
#endif
#ifdef FWD_DIAG_NO_STAGE
        m = 0; if (lane == 99) s_rec[0] = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
        while (m) {
            const int j = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            const uint32_t gid = (uint32_t)__builtin_amdgcn_readlane((int)id, j);
            const uint32_t contributor = base - range.x + (uint32_t)j + 1u;
            const float4* lr = s_rec + wave * ST * 64 + j;
#define FWD_LD(k) lr[(k) * 64]
            if (V != GSR_SURFEL) {
                const float4 q0 = FWD_LD(0), q1 = FWD_LD(1), q2 = FWD_LD(2);
                const float dx = q0.x - pxf, dy = q0.y - pyf;
                const float power = -0.5f * (q0.z * dx * dx + q1.x * dy * dy) - q0.w * dx * dy;
                const float alpha = fminf(0.99f, q1.y * __expf(power));
                uint64_t okm = live & __builtin_amdgcn_fcmpf(power, 0.0f, 13) & __builtin_amdgcn_fcmpf(alpha, 1.0f / 255.0f, 11);
                const float test_T = T * (1 - alpha);
                const uint64_t stopm = okm & __builtin_amdgcn_fcmpf(test_T, 0.0001f, 4);
                live &= ~stopm; okm &= ~stopm;
                if (V == GSR_PLANE) {
                    const uint64_t ob = okm & __builtin_amdgcn_fcmpf(T, 0.5f, 2);
                    if (ob != 0 && lane == 0) atomicAdd(&p.out_observe[gid], (int)__popcll(ob));
                }
                const bool ok = __builtin_amdgcn_inverse_ballot_w64(okm);
                if (ok) {
                    const float w = alpha * T;
                    C0 += q1.z * w; C1 += q1.w * w; C2 += q2.x * w;
                    if (V == GSR_PLANE && p.render_geo) {
                        const float4 q3 = FWD_LD(3);
                        A0 += q2.y * w; A1 += q2.z * w; A2 += q2.w * w; A3 += q3.x * w; A4 += q3.y * w;
                    }
                    T = test_T;
                    last_contributor = contributor;
                }
            } else {
                const float4 q0 = FWD_LD(0), q1 = FWD_LD(1), q2 = FWD_LD(2), q3 = FWD_LD(3), q4 = FWD_LD(4);
                // p = P0 + dx Px + dy Py, (dx, dy) = this lane's pixel inside the sub-tile (staged layout: see the staging block above)
                const float ppx = fmaf(fdy, q1.z, fmaf(fdx, q0.w, q0.x)), ppy = fmaf(fdy, q1.w, fmaf(fdx, q1.x, q0.y)), ppz = fmaf(fdy, q2.x, fmaf(fdx, q1.y, q0.z));
                const float rpz = rcp_nr(ppz);
                const float sx = ppx * rpz, sy = ppy * rpz;
                const float rho3d = sx * sx + sy * sy;
                const float dx = q3.x - fdx, dy = q3.y - fdy;
                const float rho2d = FILTER_INV_SQ * (dx * dx + dy * dy);
                const float rho = fminf(rho3d, rho2d);
                const float depth = (rho3d <= rho2d) ? q2.y * rpz : q2.z;
                const float alpha = fminf(0.99f, q2.w * __builtin_amdgcn_exp2f(rho * (-0.5f * 1.4426950408889634f)));      // exp(-rho / 2)
                // (the reference's `power > 0` gate, forward.cu:389, cannot fire: rho is a minimum of two sums of squares)
                uint64_t okm = live & __builtin_amdgcn_fcmpf(ppz, 0.0f, 14) & __builtin_amdgcn_fcmpf(depth, NEAR_N, 11) & __builtin_amdgcn_fcmpf(alpha, 1.0f / 255.0f, 11);
                const float test_T = T * (1 - alpha);
                const uint64_t stopm = okm & __builtin_amdgcn_fcmpf(test_T, 0.0001f, 4);
                live &= ~stopm; okm &= ~stopm;
                const bool ok = __builtin_amdgcn_inverse_ballot_w64(okm);
                if (ok) {
                    const float w = alpha * T;
                    const float A = 1 - T;
                    const float mm = fmaf(-(FAR_N * NEAR_N) / (FAR_N - NEAR_N), rcp_(depth), FAR_N / (FAR_N - NEAR_N));      // far / (far - near) (1 - near / depth)
                    distortion += (mm * mm * A + M2 - 2 * mm * M1) * w;
                    Dd += depth * w; M1 += mm * w; M2 += mm * mm * w;
                    if (T > 0.5f) {
                        median_depth = depth; surf_idx = (int)gid;
                        mn0 = q3.z; mn1 = q3.w; mn2 = q4.x;
                        median_contributor = contributor;
                    }
                    N0 += q3.z * w; N1 += q3.w * w; N2 += q4.x * w;
                    C0 += q4.y * w; C1 += q4.z * w; C2 += q4.w * w;
                    T = test_T;
                    last_contributor = contributor;
                }
            }
            if (live == 0) break;
        }
    }

    if (inside) {
        p.final_T[pix_id] = T;
        p.n_contrib[pix_id] = last_contributor;
        p.out_color[0 * HW + pix_id] = C0 + T * p.bg[0];
        p.out_color[1 * HW + pix_id] = C1 + T * p.bg[1];
        p.out_color[2 * HW + pix_id] = C2 + T * p.bg[2];
        if (V == GSR_SURFEL) {
            p.n_contrib[pix_id + HW] = median_contributor;
            p.final_T[pix_id + HW] = M1;
            p.final_T[pix_id + 2 * HW] = M2;
            float* o = p.out_others;
            o[pix_id + 0 * HW] = Dd;
            o[pix_id + 1 * HW] = 1 - T;
            o[pix_id + 2 * HW] = N0; o[pix_id + 3 * HW] = N1; o[pix_id + 4 * HW] = N2;
            o[pix_id + 5 * HW] = median_depth;
            o[pix_id + 6 * HW] = distortion;
            o[pix_id + 7 * HW] = (float)surf_idx;
            o[pix_id + 8 * HW] = mn0; o[pix_id + 9 * HW] = mn1; o[pix_id + 10 * HW] = mn2;
        }
        if (V == GSR_PLANE && p.render_geo) {
            p.out_all_map[0 * HW + pix_id] = A0; p.out_all_map[1 * HW + pix_id] = A1; p.out_all_map[2 * HW + pix_id] = A2;
            p.out_all_map[3 * HW + pix_id] = A3; p.out_all_map[4 * HW + pix_id] = A4;
            const float rayx = (pxf - (float)(p.W * 0.5f)) / p.fx, rayy = (pyf - (float)(p.H * 0.5f)) / p.fy;
            p.out_plane_depth[pix_id] = (float)(A4 / -(double)((A0 * rayx + A1 * rayy + A2) + 1.0e-8));
        }
    }
}

// =================================================================================================== backward
// one component: wave-reduce and let lane 63 add it to acc[slot]
#define GSR_REDUCE_ADD(slot, val)                                         \
    do {                                                                  \
        float s_ = wave_sum_to_lane63(val);                               \
        if (lane == 63) atomic_addf(accg + (slot), s_);                   \
    } while (0)

template <int V>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) k_blend_bwd(BlendParams p)
{
    constexpr int ST = (V == GSR_EWA) ? GSR_REC_EWA : (V == GSR_PLANE ? GSR_REC_PLANE : GSR_REC_SURFEL);
    constexpr int AS = (V == GSR_EWA) ? GSR_ACC_EWA : (V == GSR_PLANE ? GSR_ACC_PLANE : GSR_ACC_SURFEL);
    const int tile = tile_of_block(blockIdx.x, p.gx * p.gy, p.xcd_remap, p.tile_order, p.static_map);
    const int tx = tile % p.gx, ty = tile / p.gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ox = tx * GSR_TILE + (wave & 1) * GSR_SUB, oy = ty * GSR_TILE + (wave >> 1) * GSR_SUB;
    if (ox >= p.W || oy >= p.H) return;
    const int px = ox + (lane & 7), py = oy + (lane >> 3);
    const bool inside = px < p.W && py < p.H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = p.ranges[tile];
    const size_t HW = (size_t)p.W * p.H;
    const uint32_t pix_id = inside ? (uint32_t)p.W * py + px : 0u;

    const float T_final = inside ? p.final_T[pix_id] : 0.f;
    float T = T_final;
    const uint32_t last_contributor = inside ? p.n_contrib[pix_id] : 0u;
    const uint32_t max_last = wave_max_u32(last_contributor);
    if (max_last == 0) return;

    float dLp0 = 0, dLp1 = 0, dLp2 = 0;
    if (inside && p.dL_dcolor) { dLp0 = p.dL_dcolor[pix_id]; dLp1 = p.dL_dcolor[HW + pix_id]; dLp2 = p.dL_dcolor[2 * HW + pix_id]; }
    const float bg_dot_dpixel = p.bg[0] * dLp0 + p.bg[1] * dLp1 + p.bg[2] * dLp2;
    float arA = 0;      // collapsed accum recurrence (all channels)
    const float ddelx_dx = 0.5f * p.W, ddely_dy = 0.5f * p.H;

    // PLANE (backward.cu:433,460-490)
    const bool geo = (V == GSR_PLANE) && p.render_geo;
    float dA[5] = { 0, 0, 0, 0, 0 };
    if (geo && inside) {
        const float rayx = (float)((pxf - p.W * 0.5) / p.fx), rayy = (float)((pyf - p.H * 0.5) / p.fy);
        if (p.dL_dout_all_map)
            for (int c = 0; c < 5; c++) dA[c] = p.dL_dout_all_map[c * HW + pix_id];
        const float nx = p.all_map_pixels[pix_id], ny = p.all_map_pixels[HW + pix_id], nz = p.all_map_pixels[2 * HW + pix_id];
        const float distance = p.all_map_pixels[4 * HW + pix_id];
        const float tmp = (float)(nx * rayx + ny * rayy + nz + 1.0e-8);
        const float dpd = p.dL_dplane_depth ? p.dL_dplane_depth[pix_id] : 0.f;
        dA[4] += (-dpd / tmp);
        dA[0] += dpd * (distance / (tmp * tmp) * rayx);
        dA[1] += dpd * (distance / (tmp * tmp) * rayy);
        dA[2] += dpd * (distance / (tmp * tmp));
    }
    // SURFEL (backward.cu:205-243)
    float dL_dreg = 0, dL_ddepth = 0, dL_daccum = 0, dN0 = 0, dN1 = 0, dN2 = 0, dL_dmedian_depth = 0;
    float dMN0 = 0, dMN1 = 0, dMN2 = 0;
    uint32_t median_contributor = 0;
    float final_D = 0, final_D2 = 0, final_A = 0;
    if (V == GSR_SURFEL && inside) {
        median_contributor = p.n_contrib[pix_id + HW];
        if (p.dL_dothers) {
            const float* g = p.dL_dothers;
            dL_ddepth = g[0 * HW + pix_id]; dL_daccum = g[1 * HW + pix_id]; dL_dreg = g[6 * HW + pix_id];
            dN0 = g[2 * HW + pix_id]; dN1 = g[3 * HW + pix_id]; dN2 = g[4 * HW + pix_id];
            dL_dmedian_depth = g[5 * HW + pix_id];
            dMN0 = g[8 * HW + pix_id]; dMN1 = g[9 * HW + pix_id]; dMN2 = g[10 * HW + pix_id];
        }
        final_D = p.final_T[pix_id + HW]; final_D2 = p.final_T[pix_id + 2 * HW]; final_A = 1 - T_final;
    }

    // walk the list back to front, starting at the deepest splat any pixel of this wave used
    const uint32_t end = range.x + max_last;
    for (uint32_t top = end; top > range.x; top = (top - range.x > GSR_WAVE) ? top - GSR_WAVE : range.x) {
        const bool v = (top - range.x) > (uint32_t)lane;
        const uint32_t i = top - 1u - (uint32_t)lane;               // lane 0 = deepest
        const uint32_t id = v ? p.point_list[i] : 0u;
        const bool hit = v && cull_hit<V>(p.cull, id, (float)ox, (float)oy);
        uint64_t m = __ballot(hit);
        while (m) {
            const int j = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            const uint32_t gid = (uint32_t)__builtin_amdgcn_readlane((int)id, j);
            const float4* __restrict__ r = p.rec + (size_t)gid * ST;
            float* accg = p.acc + (size_t)gid * AS;
            const uint32_t idx0 = (top - 1u - (uint32_t)j) - range.x;       // 0-based position == reference's `contributor`
            const bool active = inside && (idx0 < last_contributor);

            if (V != GSR_SURFEL) {
                const float4 q0 = ldc(r, 0), q1 = ldc(r, 1), q2 = ldc(r, 2);
                // pins every loop-carried recurrence to one register across both back edges (the early-out below and the loop latch):
                // without it the register allocator reconciles them with a block of v_mov per pair (ISA: -10 VALU in the surfel loop)
                asm volatile("" : "+v"(T), "+v"(arA));
                const float dx = q0.x - pxf, dy = q0.y - pyf;
                const float power = -0.5f * (q0.z * dx * dx + q1.x * dy * dy) - q0.w * dx * dy;
                const float G = __expf(power);
                const float alpha = fminf(0.99f, q1.y * G);
                const bool ok = active && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
                if (__ballot(ok) == 0) continue;
                // Branch-free: a lane that does not contribute runs the same instructions with alpha = 0 (its recurrence becomes the
                // identity) and dL_dalpha = 0 (all its gradient terms vanish), so no per-value zero initialisation and no divergent
                // region is needed; G is sanitised because exp(power > 0) may overflow and inf * 0 would poison the sums.
                const float al = ok ? alpha : 0.0f, Gm = ok ? G : 0.0f;
                const float r1a = rcp_(1.f - al);
                T = T * r1a;
                const float dchannel_dcolor = al * T;
                // Every "(c - accum_c) * dL_dc" term of the reference (backward.cu:503-520; PLANE :563-579) is linear in the channel value, so
                // the per-channel accum recurrences collapse into ONE on u = sum_ch c_ch dL_ch:  dL_dalpha = (u - A) T,  A <- al u + (1 - al) A.
                float u = q1.z * dLp0 + q1.w * dLp1 + q2.x * dLp2;
                const float g_c0 = dchannel_dcolor * dLp0, g_c1 = dchannel_dcolor * dLp1, g_c2 = dchannel_dcolor * dLp2;
                float g_am[5] = { 0, 0, 0, 0, 0 };
                if (geo) {
                    const float4 q3 = ldc(r, 3);
                    const float am[5] = { q2.y, q2.z, q2.w, q3.x, q3.y };
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        u += am[c] * dA[c];
                        g_am[c] = dchannel_dcolor * dA[c];
                    }
                }
                float dL_dalpha = u - arA;
                arA = al * u + (1.f - al) * arA;
                dL_dalpha *= T;
                dL_dalpha += (-T_final * r1a) * bg_dot_dpixel;
                dL_dalpha = ok ? dL_dalpha : 0.0f;
                const float dL_dG = q1.y * dL_dalpha;
                const float gdx = Gm * dx, gdy = Gm * dy;
                const float dG_ddelx = -gdx * q0.z - gdy * q0.w;
                const float dG_ddely = -gdy * q1.x - gdx * q0.w;
                const float g_mx = dL_dG * dG_ddelx * ddelx_dx;
                const float g_my = dL_dG * dG_ddely * ddely_dy;
                const float g_ax = fabsf(g_mx), g_ay = fabsf(g_my);
                const float g_ca = -0.5f * gdx * dx * dL_dG;
                const float g_cb = -0.5f * gdx * dy * dL_dG;
                const float g_cc = -0.5f * gdy * dy * dL_dG;
                const float g_op = Gm * dL_dalpha;
                if (V == GSR_EWA) {
                    const float v8[8] = { g_c0, g_c1, g_c2, g_op, g_mx, g_my, g_ca, g_cb };
                    const float w8 = reduce8(v8, lane);
                    const float w1 = wave_sum_to_lane63(g_cc);
                    if (lane >= 56 && w8 != 0.f) atomic_addf(accg + (lane - 56), w8);
                    if (lane == 63 && w1 != 0.f) atomic_addf(accg + 8, w1);
                } else {
                    const float v16[16] = { g_c0, g_c1, g_c2, g_op, g_mx, g_my, g_ca, g_cb, g_cc, g_ax, g_ay,
                                            g_am[0], g_am[1], g_am[2], g_am[3], g_am[4] };
                    const float w16 = reduce16(v16, lane);
                    if (lane >= 48 && w16 != 0.f) atomic_addf(accg + (lane - 48), w16);
                }
            } else {
                const float4 q0 = ldc(r, 0), q1 = ldc(r, 1), q2 = ldc(r, 2), q3 = ldc(r, 3), q4 = ldc(r, 4);
                asm volatile("" : "+v"(T), "+v"(arA));   // see the EWA path
                const float Tu0 = q0.x, Tu1 = q0.y, Tu2 = q0.z, Tv0 = q0.w, Tv1 = q1.x, Tv2 = q1.y;
                const float Tw0 = q1.z, Tw1 = q1.w, Tw2 = q2.x;
                const float kx = pxf * Tw0 - Tu0, ky = pxf * Tw1 - Tu1, kz = pxf * Tw2 - Tu2;
                const float lx = pyf * Tw0 - Tv0, ly = pyf * Tw1 - Tv1, lz = pyf * Tw2 - Tv2;
                const float ppx = ky * lz - kz * ly, ppy = kz * lx - kx * lz, ppz = kx * ly - ky * lx;
                const float rpz = (ppz == 0.0f) ? 0.0f : rcp_nr(ppz);      // keeps s (hence rho, G <= 1) finite on lanes that will not contribute
                const float sx = ppx * rpz, sy = ppy * rpz;
                const float rho3d = sx * sx + sy * sy;
                const float dx = q2.y - pxf, dy = q2.z - pyf;
                const float rho2d = FILTER_INV_SQ * (dx * dx + dy * dy);
                const float rho = fminf(rho3d, rho2d);
                const float c_d = (rho3d <= rho2d) ? (sx * Tw0 + sy * Tw1) + Tw2 : Tw2;
                const float power = -0.5f * rho;
                const float G = __expf(power);
                const float opa = q2.w;
                const float alpha = fminf(0.99f, opa * G);
                const bool ok = active && !(ppz == 0.0f) && !(c_d < NEAR_N) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
                if (__ballot(ok) == 0) continue;
                // Branch-free (see the EWA path): non-contributing lanes run with alpha = 0, dL_dalpha = 0 and sanitised G, s, depth
                // (depth ~ 0 would give inf/NaN that 0 cannot cancel; p.z == 0 is handled where s is formed).  The rho3d <= rho2d fork is a pair of selects on
                // dL_dG / dL_dz instead of a divergent branch: with dL_dG3 = dL_dz3 = 0 the ray-splat terms vanish identically and
                // g_T[8] reduces to dL_dz, exactly the screen-space-filter branch of backward.cu:434-441.
                const bool b3 = rho3d <= rho2d;
                const float al = ok ? alpha : 0.0f, cd = ok ? c_d : 1.0f, okf = ok ? 1.0f : 0.0f;
                const float Gm = G, sxm = sx, sym = sy;      // finite on every lane (rpz sanitised above, rho >= 0)
                const float r1a = rcp_(1.f - al);
                T = T * r1a;
                const float w = al * T;
                const float g_c0 = w * dLp0, g_c1 = w * dLp1, g_c2 = w * dLp2;
                const float rcd = rcp_(cd);
                const float m_d = (FAR_N / (FAR_N - NEAR_N)) * (1 - NEAR_N * rcd);
                const float dmd_dd = ((FAR_N * NEAR_N) / (FAR_N - NEAR_N)) * rcd * rcd;
                float dL_dz = (ok && idx0 + 1u == median_contributor) ? dL_dmedian_depth : 0.0f;      // contributor == median_contributor-1
                const float dL_dweight = (final_D2 + m_d * m_d * final_A - 2 * m_d * final_D) * dL_dreg;
                const float dL_dmd = 2.0f * w * (m_d * final_A - final_D) * dL_dreg;
                dL_dz += dL_dmd * dmd_dd;
                // colour, distortion weight (last_dL_dT), depth, alpha and normal are all "channels" of the same recurrence (backward.cu:
                // 300-375): collapsed into one on u (see the EWA path) -- nine per-lane recurrences fewer than the literal form
                const float u = (q3.w * dLp0 + q4.x * dLp1 + q4.y * dLp2) + dL_dweight + cd * dL_ddepth + dL_daccum + (q3.x * dN0 + q3.y * dN1 + q3.z * dN2);
                float dL_dalpha = u - arA;
                arA = al * u + (1.f - al) * arA;
                // fork quirk (backward.cu:381): median-normal gradient is added for every contributing splat
                const float g_n0 = w * dN0 + okf * dMN0, g_n1 = w * dN1 + okf * dMN1, g_n2 = w * dN2 + okf * dMN2;
                dL_dalpha *= T;
                dL_dalpha += (-T_final * r1a) * bg_dot_dpixel;
                dL_dalpha = ok ? dL_dalpha : 0.0f;
                const float dL_dG = opa * dL_dalpha;
                dL_dz += w * dL_ddepth;
                const float dL_dG3 = b3 ? dL_dG : 0.0f, dL_dG2 = b3 ? 0.0f : dL_dG, dL_dz3 = b3 ? dL_dz : 0.0f;
                const float dL_dsx = dL_dG3 * -Gm * sxm + dL_dz3 * Tw0;
                const float dL_dsy = dL_dG3 * -Gm * sym + dL_dz3 * Tw1;
                const float rpzm = rpz;
                const float dpx = dL_dsx * rpzm, dpy = dL_dsy * rpzm, dpz = -(dpx * sxm + dpy * sym);
                // dL_dTu = -cross(l, dL_dp) = cross(dL_dp, l); dL_dTv = -cross(dL_dp, k) = cross(k, dL_dp)  (no sign flips)
                const float tux = dpy * lz - dpz * ly, tuy = dpz * lx - dpx * lz, tuz = dpx * ly - dpy * lx;
                const float tvx = ky * dpz - kz * dpy, tvy = kz * dpx - kx * dpz, tvz = kx * dpy - ky * dpx;
                const float g_T[9] = { tux, tuy, tuz, tvx, tvy, tvz, dL_dz3 * sxm - (pxf * tux + pyf * tvx), dL_dz3 * sym - (pxf * tuy + pyf * tvy),
                                       dL_dz - (pxf * tuz + pyf * tvz) };
                const float g_mx = dL_dG2 * (-Gm * FILTER_INV_SQ * dx);
                const float g_my = dL_dG2 * (-Gm * FILTER_INV_SQ * dy);
                const float g_op = Gm * dL_dalpha;
                // accumulator layout (SURFEL): 0-2 colour, 3 opacity, 4-6 normal, 7-15 transMat, 16-17 mean2D
                const float v16[16] = { g_c0, g_c1, g_c2, g_op, g_n0, g_n1, g_n2, g_T[0], g_T[1], g_T[2], g_T[3], g_T[4],
                                        g_T[5], g_T[6], g_T[7], g_T[8] };
                const float w16 = reduce16(v16, lane);
                if (lane >= 48 && w16 != 0.f) atomic_addf(accg + (lane - 48), w16);
                if (__ballot(ok && !b3) != 0) {      // wave-uniform: any pair on the screen-space filter branch
                    const float w2 = reduce2(g_mx, g_my, lane);
                    if (lane >= 62 && w2 != 0.f) atomic_addf(accg + 16 + (lane - 62), w2);
                }
            }
        }
    }
}

// =================================================================================================== launchers
static BlendParams make_bp(const gsr_cfg* cfg, GeomView g, BinView b, ImgView im, hipStream_t s)
{
    BlendParams p = {};
    p.W = cfg->W; p.H = cfg->H;
    p.gx = (cfg->W + GSR_TILE - 1) / GSR_TILE; p.gy = (cfg->H + GSR_TILE - 1) / GSR_TILE;
    p.variant = cfg->variant; p.render_geo = cfg->render_geo;
    {
        // Which tile workgroup b works on (it runs on XCD b % 8, every XCD has its own L2):
        //   GSR_XCD_REMAP=0  raster: b = tile.  Neighbouring tiles -- which share most of their splats -- sit on eight different XCDs.
        //   GSR_XCD_REMAP=1  one contiguous band of tiles per XCD (rounds 1-3): best L2 reuse, but the busy band of a scene whose density is
        //                    not uniform is one XCD's alone: 701 vs 1070 it/s with half of the gaussians in the image centre (tools/ab_tile_order.sh).
        //   GSR_XCD_REMAP=2  (default) 4x4-tile blocks dealt out to the XCDs cyclically (gsr_static_tile_map): a block's tiles share an L2,
        //                    every image region is spread over all eight XCDs.
        static int remap = -1;
        if (remap < 0) { const char* e = getenv("GSR_XCD_REMAP"); remap = e ? atoi(e) : 2; }
        p.xcd_remap = remap == 1 ? 1 : 0;
        p.static_map = remap == 2 ? gsr_static_tile_map(p.gx, p.gy, s) : nullptr;
    }
    p.fy = cfg->H / (2.0f * cfg->tanfovy);
    p.fx = cfg->W / (2.0f * cfg->tanfovx);
    p.tile_order = im.tile_order;         // used when its word T is set: decided per forward (gsr_tile_order_wanted, gsr_api.hip)
    p.long_word = nullptr; p.long_len = 0xFFFFFFFFu; p.status = nullptr; p.status_total = nullptr; p.status_cap = 0u;
    p.qmask = b.qmask;                    // the forward's per-(batch, quadrant) cull ballots, read by the splat-parallel backward
    p.ranges = im.ranges; p.point_list = b.point_list; p.cull = g.cull; p.rec = g.rec; p.bg = cfg->bg;
    p.depth_key = nullptr; p.list_rw = nullptr; p.tile_keys = nullptr; p.scratch_keys = nullptr; p.scratch_ids = nullptr;
    p.sort_buckets = 1; p.list_any_order = 0;
    p.final_T = im.final_T; p.n_contrib = im.n_contrib;
    return p;
}

int gsr_launch_blend_fwd(const gsr_cfg* cfg, const gsr_inputs* in, GeomView g, BinView b, ImgView im,
                         const gsr_outputs* out, hipStream_t s, bool global_order, uint32_t* status_dev, uint32_t status_cap)
{
    (void)in;
    BlendParams p = make_bp(cfg, g, b, im, s);
    if (status_dev) { p.status = status_dev; p.status_total = g.counters; p.status_cap = status_cap; }
    if (!global_order && gsr_tile_sort_is_fused()) {
        p.depth_key = g.depth_key; p.list_rw = b.point_list; p.tile_keys = b.tile_keys; p.scratch_keys = b.keys_b; p.scratch_ids = b.vals_b;
        p.list_any_order = gsr_tile_bucket_chunk(global_order, p.gx * p.gy, b.cap) ? 1 : 0;
    }
    {   // long-list feedback for the launch order of the forwards that follow (gsr_tile_order_wanted): "long" = beyond max(1024, ~4 x the mean list,
        // the mean taken as 5 instances per gaussian over T tiles)
        const long long T = (long long)p.gx * p.gy;
        p.long_word = gsr_long_list_word();
        p.long_len = (uint32_t)std::max(1024ll, 20ll * (long long)cfg->P / std::max(T, 1ll));
    }
    p.out_color = out->out_color; p.out_others = out->out_others; p.out_observe = out->out_observe;
    p.out_all_map = out->out_all_map; p.out_plane_depth = out->out_plane_depth;
    dim3 grid(p.gx * p.gy), block(256);
    switch (cfg->variant) {
    case GSR_EWA: hipLaunchKernelGGL(k_blend_fwd<GSR_EWA>, grid, block, 0, s, p); break;
    case GSR_PLANE: hipLaunchKernelGGL(k_blend_fwd<GSR_PLANE>, grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL(k_blend_fwd<GSR_SURFEL>, grid, block, 0, s, p); break;
    }
    return gsr_check_launch("blend_fwd", s, cfg->debug);
}

bool gsr_blend_bwd_is_sp()
{
    static int use_sp = -1;
    if (use_sp < 0) { const char* e = getenv("GSR_BWD"); use_sp = (e && e[0] == 'p') ? 0 : 1; }
    return use_sp != 0;
}
void gsr_blend_bwd_attach_events(hipEvent_t start, hipEvent_t stop) { gsr_blend_bwd_sp_attach_events(start, stop); }

int gsr_launch_blend_bwd(const gsr_cfg* cfg, const gsr_inputs* in, GeomView g, BinView b, ImgView im,
                         const gsr_out_grads* og, float* acc, hipStream_t s)
{
    (void)in;
    BlendParams p = make_bp(cfg, g, b, im, s);
    p.dL_dcolor = og->dL_dcolor; p.dL_dothers = og->dL_dothers; p.dL_dout_all_map = og->dL_dout_all_map;
    p.dL_dplane_depth = og->dL_dplane_depth; p.all_map_pixels = og->all_map_pixels;
    p.acc = acc;
    // GSR_BWD=sp (default): the splat-parallel backward of gsr_blend_sp.hip; GSR_BWD=px: the pixel-parallel kernel below (round 1's
    // formulation, kept switchable for A/B).  Both pass the full parity suite incl. the 300k / 1080p oracle cases.  Measured on MI355X,
    // 300k splats, 1080p (round 2): surfel sp 0.486 / px 0.524 ms, EWA 0.390 / 0.483, PLANE 0.331 / 0.372 -- DESIGN.md section 4.
    const int use_sp = gsr_blend_bwd_is_sp() ? 1 : 0;
    if (use_sp) {
        if (gsr_launch_blend_bwd_sp(p, cfg->variant, s)) return 1;
        return gsr_check_launch("blend_bwd_sp", s, cfg->debug);
    }
    dim3 grid(p.gx * p.gy), block(256);
    // (the 16-component reduction on the matrix pipe, v_mfma_f32_16x16x4_f32 with a one-hot selector, was measured in round 1 and removed in round 4:
    // 1.65 ms vs 1.04 ms for the DPP tree -- DESIGN Appendix A)
    switch (cfg->variant) {
    case GSR_EWA: hipLaunchKernelGGL(k_blend_bwd<GSR_EWA>, grid, block, 0, s, p); break;
    case GSR_PLANE: hipLaunchKernelGGL(k_blend_bwd<GSR_PLANE>, grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL(k_blend_bwd<GSR_SURFEL>, grid, block, 0, s, p); break;
    }
    return gsr_check_launch("blend_bwd", s, cfg->debug);
}
