// gsr_blend_common.h -- device helpers shared by the blend translation units (gsr_blend.hip: forward + pixel-parallel backward,
// gsr_blend_sp.hip: splat-parallel backward).  gfx950 only.
#pragma once
#include "gsr_common.h"
#include <cstdlib>

struct BlendParams {
    int W, H, gx, gy, variant, render_geo, xcd_remap;
    float fx, fy;
    const uint2* ranges;
    const uint32_t* tile_order;      // ImgView::tile_order [T + 1]: used when word T is set (k_tile_order ran for this forward)
    const uint32_t* static_map;      // [T] blockIdx -> tile, block-cyclic over the XCDs (gsr_static_tile_map, GSR_XCD_REMAP=2), or nullptr
    uint32_t* status; const uint32_t* status_total; uint32_t status_cap;      // gsr_forward_async: workgroup 0 writes status[0] = *status_total and sets status[1] when it exceeds the capacity (sticky); else nullptr
    uint32_t* long_word; uint32_t long_len;     // forward: a tile whose list is longer than long_len stores its length into *long_word (feedback, or nullptr)
    const uint32_t* point_list;
    unsigned long long* qmask;       // BinView::qmask, or nullptr (GSR_CULL_REUSE=0)
    // forward only, per-tile depth order with GSR_TILE_SORT=fused: the list arrives grouped by tile but in no particular order inside the tile and
    // the kernel's prologue sorts it by (depth_key, id) in place (gsr_tile_sort.h); nullptr: the list is already in its final order
    const uint32_t* depth_key;
    uint32_t* list_rw;               // == point_list
    int sort_buckets;                // 1: rank inside depth buckets (gsr_tile_sort.h tds_bucket_rank_wg); 0: all-pairs count / bitonic
    int list_any_order;              // 1: a tile's list arrives in no particular order (one-pass bucket sort, gsr_binning.hip) -- only the long-list fallback cares
    uint32_t* tile_keys; uint32_t* scratch_keys; uint32_t* scratch_ids;      // the long-list fallback's scratch (free ping-pong half of the binning arena)
    const float4* cull;
    const float4* rec;
    const float* bg;
    float* final_T;
    uint32_t* n_contrib;
    // forward outputs
    float* out_color; float* out_others; int32_t* out_observe; float* out_all_map; float* out_plane_depth;
    // backward inputs
    const float* dL_dcolor; const float* dL_dothers; const float* dL_dout_all_map; const float* dL_dplane_depth;
    const float* all_map_pixels;
    float* acc;
};

// Workgroup b is observed to run on XCD b % 8 (MI355X_MICROARCH.md); give each XCD a contiguous band of tiles so that
// neighbouring tiles -- which share most of their splats -- hit the same 4 MiB L2.  Bijective for any T; speed only.
__device__ __forceinline__ int tile_of_block(int b, int T, int remap, const uint32_t* __restrict__ order, const uint32_t* __restrict__ smap = nullptr)
{
    if (order[T]) return (int)order[b];   // longest tile lists first: the launch does not end on a few long tiles (k_tile_order ran for this forward)
    if (smap) return (int)smap[b];        // 4x4-tile blocks dealt out to the XCDs cyclically
    if (!remap) return b;
    const int q = T >> 3, r = T & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// ---- DPP wave reduction: total of v over the 64 lanes, returned wave-uniform ------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v += dpp_f<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v += dpp_f<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v += dpp_f<0x141, 0xF>(v);   // row_half_mirror
    v += dpp_f<0x140, 0xF>(v);   // row_mirror        -> every lane holds its row's sum
    v += dpp_f<0x142, 0xA>(v);   // row_bcast15 into rows 1,3
    v += dpp_f<0x143, 0xC>(v);   // row_bcast31 into rows 2,3 -> lanes 48..63 hold the wave total
    return v;
}
// ---- transpose-reduce: K per-lane components -> ONE register, lane (48 + c) holds the wave total of component c.
// Each level halves the number of live registers instead of running K separate 6-step butterflies:
//   xor 1, xor 2 : select + quad_perm DPP add          (3 VALU per merge)
//   xor 4, xor 8 : bank-masked row_shl/row_shr DPP adds (the DPP bank mask does the select)
//   rows         : lane-wise xor 16 / xor 32 (ds_bpermute)
// 16 components cost ~45 VALU instead of 16 x 7, and the result feeds a single 16-lane atomic instruction.
template <int QP>
__device__ __forceinline__ float merge_quad(float a, float b, bool sel)
{
    const float keep = sel ? b : a, give = sel ? a : b;
    return keep + dpp_f<QP, 0xF>(give);
}
// xor 4 / xor 8 level: lanes with the level bit clear keep a and add a[l + S], the others keep b and add b[l - S].  The DPP bank
// mask does the lane selection, so each half is ONE v_add_f32_dpp accumulating into w.  The compiler cannot form this itself (it
// materialises update_dpp(0, x) with a zero-fill and a separate add: 7 VALU instead of 3 per merge), hence the inline assembly; the
// leading s_nop covers the "VALU write -> DPP read" hazard the assembler does not see inside an asm block.
template <int S>
__device__ __forceinline__ float merge_row(float a, float b, bool sel)
{
    float w = sel ? b : a;
    if (S == 4)
        asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\tv_add_f32_dpp %0, %2, %0 row_shr:4 row_mask:0xf bank_mask:0xa"
            : "+v"(w) : "v"(a), "v"(b));
    else
        asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %0 row_shl:8 row_mask:0xf bank_mask:0x3\n\tv_add_f32_dpp %0, %2, %0 row_shr:8 row_mask:0xf bank_mask:0xc"
            : "+v"(w) : "v"(a), "v"(b));
    return w;
}
// lane-wise sum across the four 16-lane rows (every row ends up with the totals).  row_bcast cannot be used here:
// the lanes of a row hold DIFFERENT components, so the exchange must be lane l <-> l^16, l^32.  gfx950 has VALU-only
// row/half swaps (v_permlane16_swap / v_permlane32_swap), so no trip through the LDS crossbar (ds_bpermute):
//   permlane16_swap(w,w) -> {[r0,r0,r2,r2], [r1,r1,r3,r3]}, permlane32_swap(w,w) -> {[lo,lo], [hi,hi]}.
__device__ __forceinline__ float rows_to_row3(float w)
{
    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
    u2_t a = __builtin_amdgcn_permlane16_swap(__float_as_uint(w), __float_as_uint(w), false, false);
    w = __uint_as_float(a.x) + __uint_as_float(a.y);
    u2_t b = __builtin_amdgcn_permlane32_swap(__float_as_uint(w), __float_as_uint(w), false, false);
    return __uint_as_float(b.x) + __uint_as_float(b.y);
}
// lane 48+c <- total of v[c], c in [0,16)
__device__ __forceinline__ float reduce16(const float* v, int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    float r[8], q[4], p[2];
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = merge_quad<0xB1>(v[2 * i], v[2 * i + 1], b0);
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = merge_quad<0x4E>(r[2 * i], r[2 * i + 1], b1);
#pragma unroll
    for (int i = 0; i < 2; i++) p[i] = merge_row<4>(q[2 * i], q[2 * i + 1], b2);   // row_shl:4 / row_shr:4
    const float w = merge_row<8>(p[0], p[1], b3);                                   // row_shl:8 / row_shr:8
    return rows_to_row3(w);
}
// lane 56+c (and 48+c) <- total of v[c], c in [0,8)
__device__ __forceinline__ float reduce8(const float* v, int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    float r[4], q[2];
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = merge_quad<0xB1>(v[2 * i], v[2 * i + 1], b0);
#pragma unroll
    for (int i = 0; i < 2; i++) q[i] = merge_quad<0x4E>(r[2 * i], r[2 * i + 1], b1);
    float w = merge_row<4>(q[0], q[1], b2);
    w += dpp_f<0x128, 0xF>(w);   // row_ror:8 -> sum over the row, component = lane & 7
    return rows_to_row3(w);
}
// lane 62 <- total of a, lane 63 <- total of b
__device__ __forceinline__ float reduce2(float a, float b, int lane)
{
    float w = merge_quad<0xB1>(a, b, lane & 1);
    w += dpp_f<0x4E, 0xF>(w);    // xor 2
    w += dpp_f<0x124, 0xF>(w);   // row_ror:4
    w += dpp_f<0x128, 0xF>(w);   // row_ror:8
    return rows_to_row3(w);
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return v;
}

__device__ __forceinline__ void atomic_addf(float* p, float v) { unsafeAtomicAdd(p, v); }
// v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division sequence; blend outputs are tolerance-checked.
__device__ __forceinline__ float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
// v_rcp_f32 + one Newton step (2 FMA): <= 0.5 ulp-ish.  Used where the reciprocal is AMPLIFIED by the algorithm -- the ray-splat
// intersection s = p.xy / p.z of an edge-on surfel (p.z ~ 0), where the bare 1-ulp v_rcp_f32 moved five pixels beyond the oracle's own
// FMA noise floor in the round-1 sweep (surfel seed 28).
__device__ __forceinline__ float rcp_nr(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}

// (Round 2 tried the ray-splat intersection p = k x l of the surfel variant with every product and difference rounded on its own -- bit-identical
// to the oracle's p -- while chasing the one parity-sweep case beyond the noise floor; the cause was the cull conic (gsr_preprocess.hip), the
// unfused form cost +9 VALU per pair and changed nothing, so the contracted expressions stay.)
// Packed records are written by the preprocess kernel and are read-only in both blend kernels.  Loading them through
// the constant address space lets the backend use scalar (SMEM) loads for the wave-uniform address even in the
// backward kernel, where the atomics into `acc` would otherwise defeat the no-clobber analysis.
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef const f4_t __attribute__((address_space(4))) * const_rec_ptr;
__device__ __forceinline__ float4 ldc(const float4* p, int k)
{
    const_rec_ptr c = (const_rec_ptr)(p + k);
    const f4_t v = *c;
    return make_float4(v.x, v.y, v.z, v.w);
}

// ---- sub-tile cull: can splat `id` reach alpha >= 1/255 anywhere in the 8x8 block whose first pixel is (ox,oy)?
// Skipping is result-neutral (a skipped splat fails the reference's alpha gate for every pixel of the block), so the
// test only has to be conservative.  SURFEL: bounding box of the contribution region.  EWA/PLANE: exact minimum of
// the conic form q(d) = A dx^2 + 2B dx dy + C dy^2 over the block rectangle (centre inside -> 0, else the best of the
// four edges, each a clamped 1-D parabola) against 2*ln(255*opacity) (with the safety margin added in preprocess).
// minimum of q(d) = A dx^2 + 2B dx dy + C dy^2 over the rectangle [X0, X0+7] x [Y0, Y0+7] (d relative to the conic centre):
// 0 when the centre is inside, else the best of the four edges, each a clamped 1-D parabola.
__device__ __forceinline__ float conic_min_over_block(float A, float B, float C, float X0, float Y0, float ext = 7.f)
{
    const float X1 = X0 + ext, Y1 = Y0 + ext;
    if (X0 <= 0.f && X1 >= 0.f && Y0 <= 0.f && Y1 >= 0.f) return 0.f;
    const float rC = __builtin_amdgcn_rcpf(C), rA = __builtin_amdgcn_rcpf(A);
    float dy = fminf(fmaxf(-B * X0 * rC, Y0), Y1);
    float qmin = A * X0 * X0 + 2.f * B * X0 * dy + C * dy * dy;
    dy = fminf(fmaxf(-B * X1 * rC, Y0), Y1);
    qmin = fminf(qmin, A * X1 * X1 + 2.f * B * X1 * dy + C * dy * dy);
    float dx = fminf(fmaxf(-B * Y0 * rA, X0), X1);
    qmin = fminf(qmin, A * dx * dx + 2.f * B * dx * Y0 + C * Y0 * Y0);
    dx = fminf(fmaxf(-B * Y1 * rA, X0), X1);
    qmin = fminf(qmin, A * dx * dx + 2.f * B * dx * Y1 + C * Y1 * Y1);
    return qmin;
}
// one splat's cull record against the (ext+1) x (ext+1) pixel block whose first pixel is (ox, oy)
template <int V>
__device__ __forceinline__ bool cull_hit_rec(const float4 a, const float4 b, float ox, float oy, float ext = 7.f)
{
    if (V == GSR_SURFEL) {
        // union of the projected ellipse {A dx^2 + 2B dx dy + C dy^2 <= 1} about (a.x, a.y) and the low-pass disc of radius^2 b.y about
        // (b.z, b.w); b.y < 0: no pixel can reach alpha >= 1/255; A = B = C = 0: culling disabled for this splat
        if (!(b.y >= 0.f)) return false;
        const float ex0 = ox - b.z, ey0 = oy - b.w;
        const float ddx = fmaxf(fmaxf(ex0, -(ex0 + ext)), 0.f), ddy = fmaxf(fmaxf(ey0, -(ey0 + ext)), 0.f);
        if (ddx * ddx + ddy * ddy <= b.y) return true;
        return !(conic_min_over_block(a.z, a.w, b.x, ox - a.x, oy - a.y, ext) > 1.0f);      // NaN -> keep
    } else {
        const float A = a.z, B = a.w, C = b.x, tt = b.y;
        if (!(tt > 0.f)) return false;
        const float qmin = conic_min_over_block(A, B, C, ox - a.x, oy - a.y, ext);
        return !(qmin > tt);      // NaN -> keep
    }
}
// ---- the four 4x4 blocks of an 8x8 quadrant at once (splat-parallel backward, stage phase).  Block k has its first pixel at
// (X0 + 4 (k & 1), Y0 + 4 (k >> 1)); qm[k] = the minimum of q over the block's rectangle, exactly what conic_min_over_block(.., ext = 3) returns for
// that block.  The 2 x 2 arrangement shares its lines: a block's rectangle is bounded by two of the four vertical lines x in {X0, X0+3, X0+4, X0+7} and
// two of the four horizontal ones, the unconstrained minimiser along a line (-B x / C) and the line's constant terms are the same for both blocks the
// line bounds, and only the clamp to the block's span and two FMAs are per (line, block): ~120 VALU for the four blocks instead of 4 x 59.
__device__ __forceinline__ void conic_min_2x2(float A, float B, float C, float X0, float Y0, float* qm)
{
    const float rC = __builtin_amdgcn_rcpf(C), rA = __builtin_amdgcn_rcpf(A);
    const float nBrC = -B * rC, nBrA = -B * rA, B2 = B + B;
    const float xs[4] = { X0, X0 + 3.f, X0 + 4.f, X0 + 7.f }, ys[4] = { Y0, Y0 + 3.f, Y0 + 4.f, Y0 + 7.f };
    float v[4][2], h[4][2];      // v[i][r]: minimum along the vertical line xs[i] over the y span of block row r; h[j][r]: horizontal line ys[j], block column r
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float x = xs[i], ystar = nBrC * x, ax2 = A * x * x, bx = B2 * x;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const float dy = fminf(fmaxf(ystar, ys[2 * r]), ys[2 * r + 1]);
            v[i][r] = fmaf(fmaf(C, dy, bx), dy, ax2);
        }
        const float y = ys[i], xstar = nBrA * y, cy2 = C * y * y, by = B2 * y;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const float dx = fminf(fmaxf(xstar, xs[2 * r]), xs[2 * r + 1]);
            h[i][r] = fmaf(fmaf(A, dx, by), dx, cy2);
        }
    }
    const bool inx[2] = { xs[0] <= 0.f && xs[1] >= 0.f, xs[2] <= 0.f && xs[3] >= 0.f }, iny[2] = { ys[0] <= 0.f && ys[1] >= 0.f, ys[2] <= 0.f && ys[3] >= 0.f };
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int cx = k & 1, cy = k >> 1;
        const float q = fminf(fminf(v[2 * cx][cy], v[2 * cx + 1][cy]), fminf(h[2 * cy][cx], h[2 * cy + 1][cx]));
        qm[k] = (inx[cx] && iny[cy]) ? 0.f : q;
    }
}
// bit k: the splat can reach alpha >= 1/255 somewhere in 4x4 block k of the quadrant whose first pixel is (ox, oy) -- the same decisions as four
// cull_hit_rec<V>(a, b, ox + 4 (k & 1), oy + 4 (k >> 1), 3.f)
template <int V>
__device__ __forceinline__ uint32_t cull_hit_quad4(const float4 a, const float4 b, float ox, float oy)
{
    uint32_t m = 0;
    float lim = 1.0f;
    if (V == GSR_SURFEL) {
        if (!(b.y >= 0.f)) return 0u;
        // low-pass disc of radius^2 b.y about (b.z, b.w): squared distance of the disc's centre to each block's rectangle
        const float ex0 = ox - b.z, ey0 = oy - b.w;
        float ddx2[2], ddy2[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const float ex = ex0 + 4.f * c, ey = ey0 + 4.f * c;
            const float ddx = fmaxf(fmaxf(ex, -(ex + 3.f)), 0.f), ddy = fmaxf(fmaxf(ey, -(ey + 3.f)), 0.f);
            ddx2[c] = ddx * ddx; ddy2[c] = ddy * ddy;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) if (ddx2[k & 1] + ddy2[k >> 1] <= b.y) m |= 1u << k;
    } else {
        lim = b.y;
        if (!(lim > 0.f)) return 0u;
    }
    float qm[4];
    conic_min_2x2(a.z, a.w, b.x, ox - a.x, oy - a.y, qm);
#pragma unroll
    for (int k = 0; k < 4; k++) if (!(qm[k] > lim)) m |= 1u << k;      // NaN -> keep
    return m;
}
template <int V>
__device__ __forceinline__ bool cull_hit(const float4* __restrict__ cull, uint32_t id, float ox, float oy)
{
    return cull_hit_rec<V>(cull[2 * (size_t)id], cull[2 * (size_t)id + 1], ox, oy, 7.f);
}

int gsr_launch_blend_bwd_sp(const BlendParams& p, int variant, hipStream_t s);      // gsr_blend_sp.hip
void gsr_blend_bwd_sp_attach_events(hipEvent_t start, hipEvent_t stop);            // the next gsr_launch_blend_bwd_sp of this thread carries them

static constexpr float NEAR_N = 0.2f, FAR_N = 100.0f, FILTER_INV_SQ = 2.0f;

