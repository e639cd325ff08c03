// gsr_blend_sp.hip -- SPLAT-PARALLEL blend backward (round 2).  Same results as k_blend_bwd (gsr_blend.hip) within float rounding.
//
// k_blend_bwd puts one PIXEL in each lane and walks the splats: every (block, splat) step ends with a 16-component cross-lane
// transpose-reduce (~51 VALU) and a 16-lane float atomic, and only 46 % of the lanes of an 8x8 block carry a contributing pair.
// Here the roles are swapped:
//   * lane = SPLAT.  A wave is four independent 16-lane rows; row b owns one 4x4 pixel block of the wave's 8x8 quadrant and
//     holds, one per lane, 16 splats of that block's culled queue (their packed records live in VGPRs for 16 pixel steps).
//     4x4 blocks are culled with the same exact region test as the 8x8 blocks of the forward: 69 % of the lane slots carry a
//     contributing pair instead of 46 % (measured on the BASELINE scene).
//   * the row walks its block's 16 PIXELS; the pixels' constants (upstream gradients, final T, M1, M2, contributor counts) stay in
//     the registers of the lane that loaded them (lane j <-> pixel j of the block) and reach the 16 splat lanes through DPP
//     row_newbcast:i, fused into the consuming VOP2 instruction wherever a constant is used as a plain multiplicand / addend
//     (first version: broadcast reads from an LDS copy -- 8 ds_read_b128 per step kept the LDS pipe 71 % busy, 0.87 ms).  The reference's ten per-pixel recurrences (T, accum colour / depth / alpha / normal, last_dL_dT;
//     SURFEL backward.cu:279-447) collapse algebraically into two: every "(c - accum_c) * dL_dc" term is linear in the channel
//     value, so with u_j = sum_ch c_{j,ch} dL_ch
//           dL_dalpha_j = u_j T_j - (sum_{k deeper than j} w_k u_k + T_final <bg, dL_dC>) / (1 - alpha_j),   w_k = alpha_k T_k,
//           T_j = T_behind / prod_{k deeper or equal j} (1 - alpha_k).
//     Both are SCANS over the row (4 in-place DPP steps each, lane 0 = deepest), chained from one 16-splat load of the row to the
//     next through a per-pixel carry (T, S) held, again, by lane j of the row.  No forward-side checkpoints are needed: the chain starts at the stored
//     final T / last contributor exactly like the reference's replay.
//   * each lane accumulates its splat's 9 / 16 / 18 gradient components in REGISTERS over the 16 pixels -- no cross-lane
//     reduction at all -- and merges them once per load into its WAVE's private LDS table (one row per tile-list entry that
//     reaches the wave's quadrant): the four rows of a load go one after the other as plain read-add-write (a splat can sit in
//     several rows of a load, never twice in one row; DS operations of a wave execute in order), no LDS float atomics.  After
//     a barrier the four waves' tables are combined and flushed with one 16-lane global atomic per entry that received anything
//     (37 % of the tile instances reach no pixel and cost nothing): 1.64 M accumulator atomics per launch (profiles/r02_pmc_summary.json)
//     against 2.02 M for the pixel-parallel kernel.  (The first table version used ds_add_f32: ~2 cycles per LANE, a third of the
//     kernel in the LDS pipe.)
// Reference semantics: 3DGS backward.cu:399-557, SURFEL backward.cu:143-447, PLANE backward.cu:399-614 (gates, thresholds, the
// median-normal quirk, no gradient through the 0.99 clamp test); see DESIGN.md "splat-parallel backward".
#include "gsr_blend_common.h"
#include <hip/hip_ext.h>

// Waves per SIMD the register budget of each variant is sized for, and the rows of a wave's accumulation table (SP_CAP_*): the measurements behind
// these values are in DESIGN.md Appendix A (21), (53).
#ifndef SP_WPE_EWA
#define SP_WPE_EWA 6
#endif
#ifndef SP_WPE_PLANE
#define SP_WPE_PLANE 5
#endif
#ifndef SP_WPE_SURFEL
#define SP_WPE_SURFEL 4
#endif
template <int V> struct SpOcc { static constexpr int WPE = (V == GSR_EWA) ? SP_WPE_EWA : (V == GSR_PLANE ? SP_WPE_PLANE : SP_WPE_SURFEL); };
#define SP_OCC __attribute__((amdgpu_waves_per_eu(SpOcc<V>::WPE, SpOcc<V>::WPE)))
#define SP_CH 256                 // tile-list entries per chunk (queues, masks); longer lists take several chunks
// rows of a wave's private accumulation table (entries of the chunk that reach the wave's quadrant), sized so that the variant keeps its workgroups per CU: EWA 6
// (40-byte rows), PLANE 5 (64-byte rows), SURFEL 4 (72-byte rows); 4 KB of per-pixel constants beside the table in all three.  The block queues hold one byte per row since round 6
// (they were sized for a whole chunk), which is what pays for the surfel's constants.  More rows for EWA / PLANE (136 / 108: 27 072 / 32 400 B of LDS) measured SLOWER,
// 0.272 vs 0.266 and 0.299 vs 0.272 ms: the allocation is rounded up and a workgroup per CU is lost (EXPERIMENTS.md (78)).
#ifndef SP_CAP_EWA
#define SP_CAP_EWA 112
#endif
#ifndef SP_CAP_SURFEL
#define SP_CAP_SURFEL 111
#endif
#ifndef SP_CAP_PLANE
#define SP_CAP_PLANE 91
#endif

template <int V> struct SpTraits;
// NACC: gradient components per splat, TS: floats per table row, NREG: per-lane register accumulators of a load (SURFEL: the nine transMat components
// are accumulated as three moments of dL/dp over the load's 16 pixels plus three depth terms and turned into dL/dTu, dL/dTv, dL/dTw once per load)
// PLANE's table row is exactly its 16 components = 64 bytes, so the 16 lanes of a DPP row (16 different table rows, same column) meet in two 8-byte bank
// pairs: 36 % of the kernel's LDS cycles are bank conflicts (profiles/r04_pmc_summary.json).  Both fixes were built and measured in round 5 and BOTH ARE
// SLOWER (EXPERIMENTS.md (71)): 72-byte rows (-DSP_TS_PLANE=18, with the 84 rows that still fit five workgroups per CU) 0.291 ms against 0.278, an XOR
// swizzle of the float2 slots (16 address instructions per load) 0.2845 against 0.278 -- the kernel waits for its VALU, not for the LDS.
#ifndef SP_TS_PLANE
#define SP_TS_PLANE 16
#endif
// NC4: float4 slots per pixel of per-pixel constants kept in LDS instead of being broadcast from the owning lane's register (round 6, VERDICT r5 #3 i).  The pure
// multiplicands a step uses twice -- once in u, once in an accumulator -- are written to LDS once per kernel and fetched per step with one broadcast ds_read_b128 (the LDS
// pipe idles next to a saturated VALU), so that their consumers are VGPR-only fmas (2.6-3.0 issue cycles) instead of DPP forms (4.2-4.4).  ONE slot per variant, paid for by
// the block queues (one byte per table row since round 6): SURFEL dL/dC + dL/ddepth (8 consumers), EWA dL/dC (6), PLANE dL/dC + the first all_map gradient (8, and 91 table
// rows instead of 96).  Backward kernel in ms, DPP form -> LDS form, same box (EXPERIMENTS.md (78)): SURFEL 300k 0.3812 -> 0.3759, 1 M 1.0615 -> 1.0522, 1600x900
// 0.3615 -> 0.3544; EWA 0.2659 -> 0.2608, 0.4328 -> 0.4239, 0.2711 -> 0.2650; PLANE 0.2718 -> 0.2672, 0.7743 -> 0.7590, 0.2839 -> 0.2839.  MORE does not pay: two slots for
// the surfel (the normal's gradients too: 8 KB, 97 rows) win 3 % at 300k and lose 3-6 % at 1 M / 1600x900 / on concentrated scenes; two slots for PLANE or one slot at the
// price of a workgroup per CU lose 10-20 %.
#ifndef SP_NC4_EWA
#define SP_NC4_EWA 1
#endif
#ifndef SP_NC4_PLANE
#define SP_NC4_PLANE 1
#endif
template <> struct SpTraits<GSR_EWA> { static constexpr int NACC = 9, TS = 10, NREG = 10, NPIN = 9, NC4 = SP_NC4_EWA; };
template <> struct SpTraits<GSR_PLANE> { static constexpr int NACC = 16, TS = SP_TS_PLANE, NREG = 16, NPIN = 16, NC4 = SP_NC4_PLANE; };
template <> struct SpTraits<GSR_SURFEL> { static constexpr int NACC = 18, TS = 18, NREG = 21, NPIN = 21, NC4 = 1; };

// inclusive scans along the 16 lanes of each DPP row, lane 0 first, in place.  A lane whose source would lie outside the row is
// disabled by the hardware (bound_ctrl off) and keeps its value -- exactly the Hillis-Steele step.  s_nop 1 = the two wait states
// between a VALU write and a DPP read of the same VGPR, which the assembler cannot see inside an asm block.
__device__ __forceinline__ float row_scan_mul(float x)
{
    asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(x));
    return x;
}
__device__ __forceinline__ float row_scan_add(float x)
{
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(x));
    return x;
}
__device__ __forceinline__ uint32_t row_max_u32(uint32_t v)      // every lane of a 16-lane row <- the row's maximum
{
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 16));
    return v;
}
__device__ __forceinline__ void lds_addf(float* p, float v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// ---- row broadcast of lane I: DPP row_newbcast (gfx90a+), either materialised (mov) or fused into the consuming VOP2 operation.
// The fused forms are inline assembly (hipcc folds a DPP mov into its consumer only sporadically); their DPP operand is always a
// LONG-LIVED per-pixel constant register, never a value produced by the preceding instructions, so the "VALU write -> DPP read"
// wait states are satisfied by construction -- tools/audit_dpp.py checks that on the generated ISA after every build.
template <int I> __device__ __forceinline__ float bc_mov(float c)
{
    float r;
    asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(c), "n"(I));
    return r;
}
template <int I> __device__ __forceinline__ uint32_t bc_movu(uint32_t c)
{
    uint32_t r;
    asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(c), "n"(I));
    return r;
}
template <int I> __device__ __forceinline__ float bc_mul(float c, float x)               // bcast(c) * x
{
    float r;
    asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(c), "v"(x), "n"(I));
    return r;
}
template <int I> __device__ __forceinline__ float bc_add(float c, float x)               // bcast(c) + x
{
    float r;
    asm volatile("v_add_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(c), "v"(x), "n"(I));
    return r;
}
template <int I> __device__ __forceinline__ float bc_fmac(float acc, float c, float x)   // acc + bcast(c) * x
{
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(x), "n"(I));
    return acc;
}
// lane I of each row -> every lane of the row, for a value that is (re)written inside the loop: builtin form, so that the compiler
// sees the DPP read and places the wait states itself
template <int I> __device__ __forceinline__ float bc_fresh(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + I, 0xf, 0xf, true));
}

// acc + c * x with c either broadcast from lane I's register (fused DPP form) or already fetched from LDS by this lane
template <int I, bool L> __device__ __forceinline__ float pc_fmac(float acc, float kc, float lc, float x) { if constexpr (L) return fmaf(lc, x, acc); else return bc_fmac<I>(acc, kc, x); }

// New carry of pixel I: lane I of every row <- lane 15's value, every other lane keeps its own.  One v_cndmask_b32_dpp per carry (D = vcc ? src1 :
// dpp(src0), vcc = the constant mask "lane != I of its row", written by two scalar moves) instead of a DPP move and a v_cndmask each -- two VALU
// instructions less per pixel step in every variant.  The scalar moves and the s_nop are the wait states between the VALU write of a / b and their DPP read.
template <int I> __device__ __forceinline__ void carry_put(float& Tc, float& Sc, float a, float b)
{
    constexpr int M = (int)~(0x00010001u << I);
    asm volatile("s_mov_b32 vcc_lo, %4\n\ts_mov_b32 vcc_hi, %4\n\ts_nop 1\n\t"
                 "v_cndmask_b32_dpp %0, %2, %0, vcc row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_dpp %1, %3, %1, vcc row_newbcast:15 row_mask:0xf bank_mask:0xf"
                 : "+v"(Tc), "+v"(Sc) : "v"(a), "v"(b), "n"(M) : "vcc");
}

// Opaque in-place use of the accumulators at the end of every pixel step.  The accumulations are pure arithmetic whose only reader is
// the table flush after step 15, so LLVM sinks each step's gradient maths below all later steps and keeps ~8 values per step alive
// for it (216 live VGPRs at step 13); pinned, every step finishes its own work.
template <int N> __device__ __forceinline__ void sp_pin(float* a)
{
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]));
    if constexpr (N > 9) asm volatile("" : "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
    if constexpr (N > 16) asm volatile("" : "+v"(a[16]), "+v"(a[17]));
    if constexpr (N > 18) asm volatile("" : "+v"(a[18]), "+v"(a[19]), "+v"(a[20]));
}

// (Round 5 tried non-temporal loads for the per-pixel inputs -- read once, 41-158 MB per launch -- so that they would not evict the splat records from the 4 MB
// L2: FETCH_SIZE rose instead, EWA 77.7 -> 79.8 MiB, surfel 129.2 -> 142.3, and the surfel kernel slowed from 0.385 to 0.389 ms; EXPERIMENTS.md (73).)
template <typename T> __device__ __forceinline__ T ld_once(const T* p) { return *p; }

// per-pixel constants, lane j of a row holds those of pixel j of the row's 4x4 block
template <int V> struct SpPix;
template <> struct SpPix<GSR_EWA> { float dLp0, dLp1, dLp2, Tc, Sc, rowx, rowy; uint32_t last; };
template <> struct SpPix<GSR_PLANE> { float dLp0, dLp1, dLp2, dA0, dA1, dA2, dA3, dA4, Tc, Sc, rowx, rowy; uint32_t last; };
template <> struct SpPix<GSR_SURFEL> {
    // c0, c1, c2: the distortion terms of a pixel as a quadratic in the splat's mapped depth m (SURFEL backward.cu:347-364, with the final A, M1, M2
    // of the pixel): dL_daccum + dL_dweight = c0 + c1 m + c2 m^2 with c0 = dL_daccum + dL_dreg M2, c1 = -2 dL_dreg M1, c2 = dL_dreg A, and dL_dmd = w (2 c2 m + c1)
    float dLp0, dLp1, dLp2, dLd, c0, c1, c2, dN0, dN1, dN2, dLmd, dMN0, dMN1, dMN2, Tc, Sc, rowx, rowy;
    uint32_t last, med;
};

// SURFEL, per (splat, 4x4 block) load.  The ray-splat vector p = k x l, k = px Tw - Tu, l = py Tw - Tv (SURFEL forward.cu:351-357), is AFFINE in the
// pixel: with (x0, y0) the block's first pixel, k0 = x0 Tw - Tu, l0 = y0 Tw - Tv,
//     p(x0 + dx, y0 + dy) = k0 x l0 + dx (Tw x l0) + dy (k0 x Tw)          (the dx dy term is Tw x Tw = 0).
// The cancellation of px Tw against Tu happens once, in k0 / l0, as it does per pixel in the reference's form; the three cross products are formed
// once per load and a pixel step evaluates p with <= 6 FMAs instead of 12 instructions.  The depth s . Tw.xy + Tw.z equals (p . Tw) / p.z and
// p . Tw = det[Tu Tv Tw] =: D for every pixel (k, l differ from -Tu, -Tv by multiples of Tw), so depth = D / p.z with D from the record.
struct SpSurf { float P0x, P0y, P0z, Pxx, Pxy, Pxz, Pyx, Pyy, Pyz, Tw0, Tw1, Tw2, D, cdx, cdy, oh; };      // EWA / PLANE use cdx, cdy (centre - block origin) and oh = -opacity / 2
__device__ __forceinline__ SpSurf sp_surf_setup(const float4& q0, const float4& q1, const float4& q2, const float4& q4, float x0, float y0)
{
    SpSurf S;
    const float Tu0 = q0.x, Tu1 = q0.y, Tu2 = q0.z, Tv0 = q0.w, Tv1 = q1.x, Tv2 = q1.y, Tw0 = q1.z, Tw1 = q1.w, Tw2 = q2.x;
    const float kx = x0 * Tw0 - Tu0, ky = x0 * Tw1 - Tu1, kz = x0 * Tw2 - Tu2;
    const float lx = y0 * Tw0 - Tv0, ly = y0 * Tw1 - Tv1, lz = y0 * Tw2 - Tv2;
    S.P0x = ky * lz - kz * ly; S.P0y = kz * lx - kx * lz; S.P0z = kx * ly - ky * lx;              // k0 x l0
    S.Pxx = Tw1 * lz - Tw2 * ly; S.Pxy = Tw2 * lx - Tw0 * lz; S.Pxz = Tw0 * ly - Tw1 * lx;        // Tw x l0
    S.Pyx = ky * Tw2 - kz * Tw1; S.Pyy = kz * Tw0 - kx * Tw2; S.Pyz = kx * Tw1 - ky * Tw0;        // k0 x Tw
    S.Tw0 = Tw0; S.Tw1 = Tw1; S.Tw2 = Tw2; S.D = q4.z;
    S.cdx = q2.y - x0; S.cdy = q2.z - y0;
    return S;
}
// end of a load: the moments M0 = sum dp, Mx = sum dx dp, My = sum dy dp (dp = dL/dp of a pixel step, dx, dy its offset inside the block) and
// Z = sum dL_dz (sx, sy, 1) -> the nine transMat gradient components of SURFEL backward.cu:403-433 summed over the load's pixels:
//     sum dL/dTu = sum dp x l           = M0 x l0 + My x Tw
//     sum dL/dTv = sum k x dp           = k0 x M0 + Tw x Mx
//     sum dL/dTw = Z - sum (px dL/dTu + py dL/dTv) = Z - (x0 SdTu + y0 SdTv + Mx x l0 + k0 x My)      (the dx dy moments cancel: Mxy x Tw + Tw x Mxy = 0)
__device__ __forceinline__ void sp_surf_finish(const float4& t0, const float4& t1, float Tw2, float x0, float y0, const float* a /* NREG */, float* g /* 9 */)
{
    const float Tu0 = t0.x, Tu1 = t0.y, Tu2 = t0.z, Tv0 = t0.w, Tv1 = t1.x, Tv2 = t1.y, Tw0 = t1.z, Tw1 = t1.w;
    const float kx = x0 * Tw0 - Tu0, ky = x0 * Tw1 - Tu1, kz = x0 * Tw2 - Tu2;
    const float lx = y0 * Tw0 - Tv0, ly = y0 * Tw1 - Tv1, lz = y0 * Tw2 - Tv2;
    const float m0x = a[7], m0y = a[8], m0z = a[9], mxx = a[10], mxy = a[11], mxz = a[12], myx = a[13], myy = a[14], myz = a[15];
    const float ux = (m0y * lz - m0z * ly) + (myy * Tw2 - myz * Tw1);
    const float uy = (m0z * lx - m0x * lz) + (myz * Tw0 - myx * Tw2);
    const float uz = (m0x * ly - m0y * lx) + (myx * Tw1 - myy * Tw0);
    const float vx = (ky * m0z - kz * m0y) + (Tw1 * mxz - Tw2 * mxy);
    const float vy = (kz * m0x - kx * m0z) + (Tw2 * mxx - Tw0 * mxz);
    const float vz = (kx * m0y - ky * m0x) + (Tw0 * mxy - Tw1 * mxx);
    const float cx = (mxy * lz - mxz * ly) + (ky * myz - kz * myy);
    const float cy = (mxz * lx - mxx * lz) + (kz * myx - kx * myz);
    const float cz = (mxx * ly - mxy * lx) + (kx * myy - ky * myx);
    g[0] = ux; g[1] = uy; g[2] = uz; g[3] = vx; g[4] = vy; g[5] = vz;
    g[6] = a[18] - ((x0 * ux + y0 * vx) + cx);
    g[7] = a[19] - ((x0 * uy + y0 * vy) + cy);
    g[8] = a[20] - ((x0 * uz + y0 * vz) + cz);
}

// one pixel step of a row: pixel I of the block against the 16 splats held by the row's lanes
template <int V, int I, int NACC>
__device__ __forceinline__ void sp_step(SpPix<V>& K, const float4* __restrict__ pcrow, const float4& q0, const float4& q1, const float4& q2, const float4& q3, const float4& q4, const SpSurf& S,
                                        bool valid, uint32_t idx0, int j, bool geo, float ddelx_dx, float ddely_dy, float* acc, uint32_t& okbits)
{
    constexpr bool L = SpTraits<V>::NC4 > 0;
    float4 pc0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (L) pc0 = pcrow[I];      // (dL/dC, dL/ddepth | dA0 | last contributor) of pixel I of this row's block: one address for the row's 16 lanes
    // EWA's slot has a word to spare: the pixel's last-contributor index rides in it (one DPP move less per step)
    // (the surfel's (last, median) indices from a second, 8-byte slot -- 2 KB, 104 table rows -- : 300k 0.3760 -> 0.3721 ms, but 1600x900 +0.9 %, 1 M +1.6 %, concentrated
    // scene +3.1 %: the table rows are worth more.  EXPERIMENTS.md (78).)
    const uint32_t last = (V == GSR_EWA && L) ? __float_as_uint(pc0.w) : bc_movu<I>(K.last);
    if constexpr (V != GSR_SURFEL) {
        constexpr int DX = I & 3, DY = I >> 2;
        // centre - pixel.  EWA: from the block-relative centre of the load, squares shared by the exponent and the conic gradients.  PLANE keeps the
        // round-3 expressions: the same rewrite measured SLOWER there (0.294 vs 0.285 ms at 5 waves / 96 VGPRs: the per-load values the compiler
        // hoists out of the 16 steps push two more registers into scratch).
        float dx, dy, dxx = 0.f, dxy = 0.f, dyy = 0.f, power;
        if constexpr (V == GSR_EWA) {
            dx = S.cdx - (float)DX; dy = S.cdy - (float)DY;
            dxx = dx * dx; dxy = dx * dy; dyy = dy * dy;
            power = fmaf(-0.5f, fmaf(q1.x, dyy, q0.z * dxx), -(q0.w * dxy));
        } else {
            dx = q0.x - (K.rowx + (float)DX); dy = q0.y - (K.rowy + (float)DY);
            power = -0.5f * (q0.z * dx * dx + q1.x * dy * dy) - q0.w * dx * dy;
        }
        const float G = __expf(power);      // (folding log2(e) into the conic once per load: 0.2687 vs 0.2690 ms and two more spills, EXPERIMENTS.md (72))
        const float alpha = fminf(0.99f, q1.y * G);
        const bool ok = valid & (idx0 < last) & !(power > 0.0f) & !(alpha < 1.0f / 255.0f);      // '&': no short-circuit control flow, EXEC stays full for the DPP reads
        const float al = ok ? alpha : 0.0f, Gm = ok ? G : 0.0f;
        const float om = 1.f - al;
        // T_j = T_carry / prod_{k <= j} (1 - alpha_k): the row scan runs on the reciprocals 1 / (1 - alpha_k), which the dL/dalpha term needs anyway --
        // one v_rcp_f32 (8 issue cycles) per step instead of two
        const float r1a = rcp_(om);
        const float Tj = bc_fresh<I>(K.Tc) * row_scan_mul(r1a);
        const float w = al * Tj;
        float u = L ? pc0.x * q1.z : bc_mul<I>(K.dLp0, q1.z);
        u = pc_fmac<I, L>(u, K.dLp1, pc0.y, q1.w);
        u = pc_fmac<I, L>(u, K.dLp2, pc0.z, q2.x);
        if constexpr (V == GSR_PLANE) {
            if (geo) {
                u = pc_fmac<I, L>(u, K.dA0, pc0.w, q2.y); u = bc_fmac<I>(u, K.dA1, q2.z); u = bc_fmac<I>(u, K.dA2, q2.w);
                u = bc_fmac<I>(u, K.dA3, q3.x); u = bc_fmac<I>(u, K.dA4, q3.y);
            }
        }
        const float wu = w * u;
        const float si = row_scan_add(wu);
        const float Scb = bc_fresh<I>(K.Sc);
        const float Sfx = Scb + (si - wu);
        const float dL_dalpha = ok ? (u * Tj - Sfx * r1a) : 0.0f;
        carry_put<I>(K.Tc, K.Sc, Tj, Scb + si);      // new carry of pixel I = the state in front of lane 15's splat
        // Gradients of this pair (3DGS backward.cu:520-545) with gG = G dL_dalpha and h = -opacity gG / 2 = -G dL_dG / 2:
        //   opacity += gG;  conic (xx, xy, yy) += h (dx^2, dx dy, dy^2);  mean2D.x += 2 h (A dx + B dy) W/2,  mean2D.y += 2 h (C dy + B dx) H/2.
        // EWA accumulates the two moments sum h dx, sum h dy and applies A, B, C and the NDC factors once per load (end of the load loop); PLANE needs
        // |mean2D term| per pixel (PLANE backward.cu:552-553) and keeps the literal per-pixel terms.
        acc[0] = pc_fmac<I, L>(acc[0], K.dLp0, pc0.x, w); acc[1] = pc_fmac<I, L>(acc[1], K.dLp1, pc0.y, w); acc[2] = pc_fmac<I, L>(acc[2], K.dLp2, pc0.z, w);
        if constexpr (V == GSR_EWA) {
            const float gG = Gm * dL_dalpha;
            const float h = gG * S.oh;
            acc[3] += gG;
            acc[6] = fmaf(h, dxx, acc[6]); acc[7] = fmaf(h, dxy, acc[7]); acc[8] = fmaf(h, dyy, acc[8]);
            acc[4] = fmaf(h, dx, acc[4]); acc[5] = fmaf(h, dy, acc[5]);
        } else {
            const float dL_dG = q1.y * dL_dalpha;
            const float gdx = Gm * dx, gdy = Gm * dy;
            const float dG_ddelx = -gdx * q0.z - gdy * q0.w;
            const float dG_ddely = -gdy * q1.x - gdx * q0.w;
            const float g_mx = dL_dG * dG_ddelx * ddelx_dx;
            const float g_my = dL_dG * dG_ddely * ddely_dy;
            acc[3] += Gm * dL_dalpha;
            acc[4] += g_mx; acc[5] += g_my;
            acc[6] += -0.5f * gdx * dx * dL_dG;
            acc[7] += -0.5f * gdx * dy * dL_dG;
            acc[8] += -0.5f * gdy * dy * dL_dG;
            acc[9] += fabsf(g_mx); acc[10] += fabsf(g_my);
            if (geo) {
                acc[11] = pc_fmac<I, L>(acc[11], K.dA0, pc0.w, w); acc[12] = bc_fmac<I>(acc[12], K.dA1, w); acc[13] = bc_fmac<I>(acc[13], K.dA2, w);
                acc[14] = bc_fmac<I>(acc[14], K.dA3, w); acc[15] = bc_fmac<I>(acc[15], K.dA4, w);
            }
        }
    } else {
        const float opa = q2.w;
        constexpr int DX = I & 3, DY = I >> 2;
        float ppx = S.P0x, ppy = S.P0y, ppz = S.P0z;
        if constexpr (DX != 0) { ppx = fmaf((float)DX, S.Pxx, ppx); ppy = fmaf((float)DX, S.Pxy, ppy); ppz = fmaf((float)DX, S.Pxz, ppz); }
        if constexpr (DY != 0) { ppx = fmaf((float)DY, S.Pyx, ppx); ppy = fmaf((float)DY, S.Pyy, ppy); ppz = fmaf((float)DY, S.Pyz, ppz); }
        const float rpz = (ppz == 0.0f) ? 0.0f : rcp_nr(ppz);
        const float sx = ppx * rpz, sy = ppy * rpz;
        const float rho3d = sx * sx + sy * sy;
        const float dx = S.cdx - (float)DX, dy = S.cdy - (float)DY;
        const float rho2d = FILTER_INV_SQ * (dx * dx + dy * dy);
        const float rho = fminf(rho3d, rho2d);
        const bool b3 = rho3d <= rho2d;
        const float c_d = b3 ? S.D * rpz : S.Tw2;
        const float G = __builtin_amdgcn_exp2f(rho * (-0.5f * 1.4426950408889634f));      // exp(-rho / 2): one multiply in front of v_exp_f32 instead of two
        const float alpha = fminf(0.99f, opa * G);
        // the reference's `power > 0` gate (forward.cu:389) cannot fire here: rho = min of two sums of squares, power = -rho / 2 <= 0 (NaN passes it as well)
        const bool ok = valid & (idx0 < last) & !(ppz == 0.0f) & !(c_d < NEAR_N) & !(alpha < 1.0f / 255.0f);      // '&': no short-circuit control flow, EXEC stays full for the DPP reads
        const float al = ok ? alpha : 0.0f, cd = ok ? c_d : 1.0f;
        okbits |= ok ? (1u << I) : 0u;              // which of the load's 16 pixels this splat contributed to (the median-normal pass below)
        const float om = 1.f - al;
        // T_j = T_carry / prod_{k <= j} (1 - alpha_k): the row scan runs on the reciprocals 1 / (1 - alpha_k), which the dL/dalpha term needs anyway --
        // one v_rcp_f32 (8 issue cycles) per step instead of two
        const float r1a = rcp_(om);
        const float Tj = bc_fresh<I>(K.Tc) * row_scan_mul(r1a);
        const float w = al * Tj;
        // (1 / c_d without a transcendental -- `ok ? (b3 ? ppz / D : 1 / Tw.z) : 1` with both reciprocals formed once per load -- was built and measured in round 6:
        // 0.3843 ms against 0.3851 over three alternating runs, 128 VGPRs and 0 spills either way: noise.  EXPERIMENTS.md (76).)
        const float rcd = rcp_(cd);
        const float m_d = fmaf(-(FAR_N * NEAR_N) / (FAR_N - NEAR_N), rcd, FAR_N / (FAR_N - NEAR_N));      // far / (far - near) (1 - near / depth)
        const float dmd_dd = ((FAR_N * NEAR_N) / (FAR_N - NEAR_N)) * rcd * rcd;
        float u = bc_mul<I>(K.c2, m_d * m_d);
        u = bc_fmac<I>(u, K.c1, m_d);
        u = bc_add<I>(K.c0, u);
        u = pc_fmac<I, L>(u, K.dLd, pc0.w, cd);
        u = pc_fmac<I, L>(u, K.dLp0, pc0.x, q3.w); u = pc_fmac<I, L>(u, K.dLp1, pc0.y, q4.x); u = pc_fmac<I, L>(u, K.dLp2, pc0.z, q4.y);
        u = bc_fmac<I>(u, K.dN0, q3.x); u = bc_fmac<I>(u, K.dN1, q3.y); u = bc_fmac<I>(u, K.dN2, q3.z);
        const float wu = w * u;
        const float si = row_scan_add(wu);
        const float Scb = bc_fresh<I>(K.Sc);
        const float Sfx = Scb + (si - wu);
        const float dL_dalpha = ok ? (u * Tj - Sfx * r1a) : 0.0f;
        carry_put<I>(K.Tc, K.Sc, Tj, Scb + si);      // new carry of pixel I = the state in front of lane 15's splat
        // (skipping the three instructions below when no pixel of the wave has a median-depth gradient was tried: a wave-uniform branch here splits
        // the step's basic block and costs the taken side 13 us, 0.485 vs 0.472 ms; two copies of the 16 steps spill 29 VGPRs, 0.540 ms)
        const uint32_t med = bc_movu<I>(K.med);
        const float dLmd = bc_mov<I>(K.dLmd);
        float dL_dz = (ok & (idx0 + 1u == med)) ? dLmd : 0.0f;      // contributor == median_contributor-1
        // dL_dmd = 2 w (m A - M1) dL_dreg
        const float dL_dmd = w * bc_add<I>(K.c1, bc_mul<I>(K.c2, m_d + m_d));      // 2 w (m A - M1) dL_dreg
        dL_dz += dL_dmd * dmd_dd;
        dL_dz = pc_fmac<I, L>(dL_dz, K.dLd, pc0.w, w);
        const float dL_dG = opa * dL_dalpha;
        const float dL_dG3 = b3 ? dL_dG : 0.0f, dL_dG2 = b3 ? 0.0f : dL_dG, dL_dz3 = b3 ? dL_dz : 0.0f;
        const float dL_dsx = dL_dG3 * -G * sx + dL_dz3 * S.Tw0;
        const float dL_dsy = dL_dG3 * -G * sy + dL_dz3 * S.Tw1;
        const float dpx = dL_dsx * rpz, dpy = dL_dsy * rpz, dpz = -(dpx * sx + dpy * sy);
        // register accumulators (SURFEL): 0-2 colour, 3 opacity, 4-6 normal, 7-9 M0, 10-12 Mx, 13-15 My, 16-17 mean2D, 18-20 Z (sp_surf_finish)
        acc[0] = pc_fmac<I, L>(acc[0], K.dLp0, pc0.x, w); acc[1] = pc_fmac<I, L>(acc[1], K.dLp1, pc0.y, w); acc[2] = pc_fmac<I, L>(acc[2], K.dLp2, pc0.z, w);
        acc[3] += G * dL_dalpha;
        acc[4] = bc_fmac<I>(acc[4], K.dN0, w); acc[5] = bc_fmac<I>(acc[5], K.dN1, w); acc[6] = bc_fmac<I>(acc[6], K.dN2, w);
        acc[7] += dpx; acc[8] += dpy; acc[9] += dpz;
        if constexpr (DX != 0) { acc[10] = fmaf((float)DX, dpx, acc[10]); acc[11] = fmaf((float)DX, dpy, acc[11]); acc[12] = fmaf((float)DX, dpz, acc[12]); }
        if constexpr (DY != 0) { acc[13] = fmaf((float)DY, dpx, acc[13]); acc[14] = fmaf((float)DY, dpy, acc[14]); acc[15] = fmaf((float)DY, dpz, acc[15]); }
        acc[18] = fmaf(dL_dz3, sx, acc[18]); acc[19] = fmaf(dL_dz3, sy, acc[19]); acc[20] += dL_dz;
        acc[16] += dL_dG2 * (-G * FILTER_INV_SQ * dx);
        acc[17] += dL_dG2 * (-G * FILTER_INV_SQ * dy);
    }
}

// median-normal quirk (SURFEL backward.cu:381): EVERY contributor of a pixel receives dL/d(median normal).  The 2DGS scenes send no gradient to those
// channels (twodgs_scene.py:88-105), so this runs -- once per load, wave-uniformly -- only when some pixel of the wave has one: the pixel steps just
// record which pixels each splat contributed to (one bit per step), instead of three DPP fmacs behind a branch in every step (round 3: the branch's
// two register assignments for the accumulators were reconciled with ~10 v_mov per step).
template <int I> __device__ __forceinline__ void sp_mn_pixel(const SpPix<GSR_SURFEL>& K, uint32_t okbits, float* acc)
{
    const float okf = (float)((okbits >> I) & 1u);
    acc[4] = bc_fmac<I>(acc[4], K.dMN0, okf); acc[5] = bc_fmac<I>(acc[5], K.dMN1, okf); acc[6] = bc_fmac<I>(acc[6], K.dMN2, okf);
}

// which 8x8 quadrants of the tile an entry reaches, when the forward left no ballots (GSR_CULL_REUSE-less callers, debug): out of line, so that the four
// tile-origin constants of this rarely taken path are not hoisted into -- and spilled from -- the kernel's prologue (EWA / PLANE: 4 / 6 spilled registers in round 4)
template <int V> __device__ __attribute__((noinline)) uint32_t sp_quadrant_mask_slow(const float4* __restrict__ cull, uint32_t id, int tx, int ty)
{
    const float4 ca = cull[2 * (size_t)id], cb = cull[2 * (size_t)id + 1];
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (cull_hit_rec<V>(ca, cb, (float)(tx * GSR_TILE + (q & 1) * GSR_SUB), (float)(ty * GSR_TILE + (q >> 1) * GSR_SUB), 7.f)) m |= 1u << q;
    return m;
}

template <int V>
__global__ void __launch_bounds__(256) SP_OCC k_blend_bwd_sp(BlendParams p)
{
    using TR = SpTraits<V>;
    constexpr int ST = (V == GSR_EWA) ? GSR_REC_EWA : (V == GSR_PLANE ? GSR_REC_PLANE : GSR_REC_SURFEL);
    constexpr int AS = (V == GSR_EWA) ? GSR_ACC_EWA : (V == GSR_PLANE ? GSR_ACC_PLANE : GSR_ACC_SURFEL);
    constexpr int NACC = TR::NACC, TS = TR::TS, NREG = TR::NREG;

    constexpr int CAPV = (V == GSR_EWA) ? SP_CAP_EWA : (V == GSR_PLANE ? SP_CAP_PLANE : SP_CAP_SURFEL);      // table rows per wave
    __shared__ float2 s_wtab[4 * CAPV * (TS / 2)];        // [wave][compact entry][component pair]: PRIVATE to the wave, plain read-add-write
    __shared__ uint8_t s_cidx[4 * SP_CH];                   // [wave][entry] -> row of the wave's table, 0xFF: the entry does not reach the quadrant
    __shared__ uint32_t s_ids[SP_CH];
    __shared__ uint32_t s_lo[4];                            // per wave: the first entry of the chunk whose rows still fit its table
    __shared__ uint16_t s_mask[SP_CH];                      // bit q: entry reaches 8x8 quadrant q of the tile
    __shared__ uint8_t s_cent[4 * CAPV];                  // [wave][table row] -> chunk-local entry
    __shared__ uint8_t s_queue[4 * 4 * CAPV];               // [wave][block][position] -> chunk-local entry, list order (a block's queue is a subset of its wave's <= CAPV rows)
    __shared__ uint32_t s_wmax[4];
    constexpr int NC4 = TR::NC4;
    // (the four rows' broadcast reads of a step fall on the same banks: rows 17 slots apart instead of 16 remove the conflicts -- 22 % of the kernel's LDS cycles -- and
    // change nothing, 0.3741 vs 0.3743 ms: the kernel does not wait for the LDS.  EXPERIMENTS.md (78).)
    constexpr int PCW = 64, PCR = 16;
    __shared__ float4 s_pc[NC4 > 0 ? 4 * PCW * NC4 : 1];    // [wave][row][pixel]: the per-pixel multiplicands the steps fetch from LDS (SpTraits::NC4)

    const int tile = tile_of_block(blockIdx.x, p.gx * p.gy, p.xcd_remap, p.tile_order, p.static_map);
    const int tx = tile % p.gx, ty = tile / p.gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = lane >> 4, j = lane & 15;                 // row (= 4x4 block of the quadrant), slot / pixel inside it
    const int qx = tx * GSR_TILE + (wave & 1) * GSR_SUB, qy = ty * GSR_TILE + (wave >> 1) * GSR_SUB;
    const int px = qx + (b & 1) * 4 + (j & 3), py = qy + (b >> 1) * 4 + (j >> 2);
    const bool inside = px < p.W && py < p.H;
    const uint2 range = p.ranges[tile];
    const size_t HW = (size_t)p.W * p.H;
    const uint32_t pix_id = inside ? (uint32_t)p.W * py + px : 0u;

    // ---------------------------------------------------------------- per-pixel constants (lane = pixel), kept in registers
    SpPix<V> K;
    const float T_final = inside ? ld_once(p.final_T + pix_id) : 0.f;
    K.last = inside ? ld_once(p.n_contrib + pix_id) : 0u;
    K.dLp0 = K.dLp1 = K.dLp2 = 0.f;
    if (inside && p.dL_dcolor) { K.dLp0 = ld_once(p.dL_dcolor + pix_id); K.dLp1 = ld_once(p.dL_dcolor + HW + pix_id); K.dLp2 = ld_once(p.dL_dcolor + 2 * HW + pix_id); }
    K.rowx = (float)(qx + (b & 1) * 4); K.rowy = (float)(qy + (b >> 1) * 4);
    K.Tc = T_final;                                          // the chain starts behind the last contributor ...
    K.Sc = T_final * (p.bg[0] * K.dLp0 + p.bg[1] * K.dLp1 + p.bg[2] * K.dLp2);      // ... with the background term of dL_dalpha folded in
    const bool geo = (V == GSR_PLANE) && p.render_geo;
    if constexpr (V == GSR_PLANE) {
        float dA[5] = { 0, 0, 0, 0, 0 };
        if (geo && inside) {                                 // PLANE backward.cu:433,460-490: plane-depth chain folded into the all_map gradients
            const float rayx = (float)(((float)px - p.W * 0.5) / p.fx), rayy = (float)(((float)py - p.H * 0.5) / p.fy);
            if (p.dL_dout_all_map)
                for (int c = 0; c < 5; c++) dA[c] = ld_once(p.dL_dout_all_map + c * HW + pix_id);
            const float nx = ld_once(p.all_map_pixels + pix_id), ny = ld_once(p.all_map_pixels + HW + pix_id), nz = ld_once(p.all_map_pixels + 2 * HW + pix_id);
            const float distance = ld_once(p.all_map_pixels + 4 * HW + pix_id);
            const float tmp = (float)(nx * rayx + ny * rayy + nz + 1.0e-8);
            const float dpd = p.dL_dplane_depth ? ld_once(p.dL_dplane_depth + pix_id) : 0.f;
            dA[4] += (-dpd / tmp);
            dA[0] += dpd * (distance / (tmp * tmp) * rayx);
            dA[1] += dpd * (distance / (tmp * tmp) * rayy);
            dA[2] += dpd * (distance / (tmp * tmp));
        }
        K.dA0 = dA[0]; K.dA1 = dA[1]; K.dA2 = dA[2]; K.dA3 = dA[3]; K.dA4 = dA[4];
    }
    if constexpr (V == GSR_SURFEL) {
        K.dLd = K.c0 = K.c1 = K.c2 = K.dN0 = K.dN1 = K.dN2 = K.dLmd = K.dMN0 = K.dMN1 = K.dMN2 = 0.f;
        K.med = 0;
        if (inside) {                                        // SURFEL backward.cu:205-243
            float dLa = 0.f, dLr = 0.f;
            K.med = ld_once(p.n_contrib + pix_id + HW);
            if (p.dL_dothers) {
                const float* g = p.dL_dothers;
                K.dLd = ld_once(g + 0 * HW + pix_id); dLa = ld_once(g + 1 * HW + pix_id); K.dN0 = ld_once(g + 2 * HW + pix_id); K.dN1 = ld_once(g + 3 * HW + pix_id); K.dN2 = ld_once(g + 4 * HW + pix_id);
                K.dLmd = ld_once(g + 5 * HW + pix_id); dLr = ld_once(g + 6 * HW + pix_id); K.dMN0 = ld_once(g + 8 * HW + pix_id); K.dMN1 = ld_once(g + 9 * HW + pix_id); K.dMN2 = ld_once(g + 10 * HW + pix_id);
            }
            const float fD = ld_once(p.final_T + pix_id + HW), fD2 = ld_once(p.final_T + pix_id + 2 * HW);
            K.c0 = fmaf(dLr, fD2, dLa); K.c1 = -2.0f * dLr * fD; K.c2 = dLr * (1.f - T_final);
        }
    }
    if constexpr (V == GSR_SURFEL) s_pc[wave * PCW + b * PCR + j] = make_float4(K.dLp0, K.dLp1, K.dLp2, K.dLd);
    if constexpr (V == GSR_EWA && NC4 > 0) s_pc[wave * PCW + b * PCR + j] = make_float4(K.dLp0, K.dLp1, K.dLp2, __uint_as_float(K.last));
    if constexpr (V == GSR_PLANE && NC4 > 0) s_pc[wave * PCW + b * PCR + j] = make_float4(K.dLp0, K.dLp1, K.dLp2, K.dA0);
    const float4* pcrow = s_pc + (NC4 > 0 ? wave * PCW + b * PCR : 0);      // the row's 16 pixels (written by the row's own lanes; the __syncthreads below orders the stores)
    bool mn_live = false;
    if constexpr (V == GSR_SURFEL) mn_live = __ballot((K.dMN0 != 0.f) | (K.dMN1 != 0.f) | (K.dMN2 != 0.f)) != 0ull;
    const uint32_t mlast_row = row_max_u32(K.last);                                 // deepest contributor of this 4x4 block
    uint32_t mlast_b[4];
#pragma unroll
    for (int k = 0; k < 4; k++) mlast_b[k] = (uint32_t)__builtin_amdgcn_readlane((int)mlast_row, 16 * k);
    const uint32_t wmax = max(max(mlast_b[0], mlast_b[1]), max(mlast_b[2], mlast_b[3]));
    if (lane == 0) s_wmax[wave] = wmax;
    __syncthreads();
    const uint32_t tile_max = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
    if (tile_max == 0) return;                              // block-uniform

    const uint8_t* myqueue = s_queue + (wave * 4 + b) * CAPV;
    const float ddelx_dx = 0.5f * p.W, ddely_dy = 0.5f * p.H;

    // The tile list is consumed from its deep end in chunks of <= SP_CH entries [cbase, hi).  A chunk whose entries would overflow a wave's SP_CAP-row table
    // (dense scenes) is TRIMMED at its shallow end to the longest suffix that fits all four tables (the entries in front of it are staged again with the
    // next chunk); rounds 2-5 re-staged the whole chunk at half its length (1600x900, lists of 157: 0.425 ms against 0.361 once the table had 88 rows).
    float2* mytab = s_wtab + wave * CAPV * (TS / 2);
    uint8_t* mycidx = s_cidx + wave * SP_CH;
    uint8_t* mycent = s_cent + wave * CAPV;
    uint32_t hi = tile_max;
    while (hi > 0) {
        const uint32_t n = min((uint32_t)SP_CH, hi);
        const uint32_t cbase = hi - n;
        // ------------------------------------------------------------ stage: ids + which 8x8 QUADRANTS of the tile each entry reaches
        {
            const uint32_t t = threadIdx.x;
            if (t < n) {
                const uint32_t id = p.point_list[range.x + cbase + t];
                s_ids[t] = id;
                uint32_t m = 0;
                if (p.qmask) {
                    // the forward kept its per-(batch, quadrant) cull ballots: same test, same inputs -- no second evaluation, no cull-record gather
                    // (valid for every entry a quadrant's wave can use: the forward wave tested all batches up to its last contributor)
                    const uint32_t ei = cbase + t;
                    const unsigned long long* q = p.qmask + (((size_t)(range.x >> 6) + (size_t)tile + (ei >> 6)) << 2);
#pragma unroll
                    for (int w4 = 0; w4 < 4; w4++) m |= (uint32_t)((q[w4] >> (ei & 63u)) & 1ull) << w4;
                } else m = sp_quadrant_mask_slow<V>(p.cull, id, tx, ty);
                s_mask[t] = (uint16_t)m;
            }
        }
        __syncthreads();
        // ------------------------------------------------------------ per wave: how much of the chunk fits its table, counted from the deep end (scalar work)
        {
            uint32_t run = 0, first = 0;
            for (int e0 = (int)((n - 1u) & ~63u); e0 >= 0; e0 -= 64) {
                const uint32_t e = (uint32_t)e0 + lane;
                uint64_t am = __ballot((e < n) && (((uint32_t)s_mask[e] >> wave) & 1u) && (cbase + e < wmax));
                const uint32_t c = (uint32_t)__popcll(am);
                if (run + c > (uint32_t)CAPV) {                  // keep the CAPV - run highest entries of this group
                    while ((uint32_t)__popcll(am) > (uint32_t)CAPV - run) am &= am - 1ull;
                    first = (uint32_t)e0 + (am ? (uint32_t)__ffsll((unsigned long long)am) - 1u : 64u);
                    break;
                }
                run += c;
            }
            if (lane == 0) s_lo[wave] = first;
        }
        __syncthreads();
        const uint32_t lo_e = max(max(s_lo[0], s_lo[1]), max(s_lo[2], s_lo[3]));      // entries [lo_e, n) are this chunk; block-uniform
        // ------------------------------------------------------------ per wave: entries that reach ITS quadrant -> table rows (list order) ...
        uint32_t cany = 0;
        for (uint32_t e0 = 0; e0 < n; e0 += 64) {
            const uint32_t e = e0 + lane;
            const bool any = (e >= lo_e) && (e < n) && (((uint32_t)s_mask[e] >> wave) & 1u) && (cbase + e < wmax);
            const uint64_t am = __ballot(any);
            const uint32_t arank = __builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
            if (e < n) mycidx[e] = any ? (uint8_t)min(cany + arank, 255u) : (uint8_t)0xFF;
            if (any && cany + arank < CAPV) mycent[cany + arank] = (uint8_t)e;
            cany += (uint32_t)__popcll(am);
        }
        // ... and, for those only, the exact test against the quadrant's four 4x4 blocks -> one queue per block (list order).  Two levels
        // (4 quadrant tests per entry, then 4 block tests per (entry, quadrant) hit: 1.35 quadrants per entry) cost ~7.6 region tests
        // per entry instead of 16.
        uint32_t cnt[4] = { 0, 0, 0, 0 };
        for (uint32_t i0 = 0; i0 < min(cany, (uint32_t)CAPV); i0 += 64) {
            const uint32_t i = i0 + lane;
            const bool v = i < min(cany, (uint32_t)CAPV);
            const uint32_t e = v ? (uint32_t)mycent[i] : 0u;
            const uint32_t id = s_ids[e];
            float4 ca = make_float4(0.f, 0.f, 0.f, 0.f), cb = make_float4(0.f, -1.f, 0.f, 0.f);
            if (v) { ca = p.cull[2 * (size_t)id]; cb = p.cull[2 * (size_t)id + 1]; }
            const uint32_t hit4 = cull_hit_quad4<V>(ca, cb, (float)qx, (float)qy);      // the four blocks share their bounding lines (gsr_blend_common.h)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool hit = v && (cbase + e < mlast_b[k]) && ((hit4 >> k) & 1u);
                const uint64_t bm = __ballot(hit);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                if (hit) s_queue[(wave * 4 + k) * CAPV + cnt[k] + rank] = (uint8_t)e;
                cnt[k] += (uint32_t)__popcll(bm);
            }
        }
        for (uint32_t q = lane; q < cany * (TS / 2); q += 64) mytab[q] = make_float2(0.f, 0.f);
        const int Qmine = (int)(b == 0 ? cnt[0] : (b == 1 ? cnt[1] : (b == 2 ? cnt[2] : cnt[3])));
        const int loads = (int)((max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3])) + 15u) >> 4);

        // ------------------------------------------------------------ 16 splats per row x 16 pixel steps
#ifdef SP_DIAG_NO_LOADS
        for (int ld = 0; ld < 0; ld++) {
#else
        for (int ld = 0; ld < loads; ld++) {
#endif
            const int pos = Qmine - 1 - (16 * ld + j);      // lane 0 = deepest entry of this load
            const bool valid = pos >= 0;
            const uint32_t e = valid ? (uint32_t)myqueue[pos] : 0u;
            const uint32_t gid = s_ids[e];
            const uint32_t idx0 = cbase + e;                // 0-based position in the tile list == the reference's `contributor`
            const float4* __restrict__ r = p.rec + (size_t)gid * ST;
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 q0 = valid ? r[0] : z4, q1 = valid ? r[1] : z4, q2 = valid ? r[2] : z4;
            float4 q3 = z4, q4 = z4;
            if (V != GSR_EWA) q3 = valid ? r[3] : z4;
            if (V == GSR_SURFEL) q4 = valid ? r[4] : z4;
            float acc[NREG];
#pragma unroll
            for (int c = 0; c < NREG; c++) acc[c] = 0.f;
            SpSurf S = {};
            uint32_t okbits = 0u;
            if constexpr (V == GSR_SURFEL) S = sp_surf_setup(q0, q1, q2, q4, K.rowx, K.rowy);
            else if constexpr (V == GSR_EWA) { S.cdx = q0.x - K.rowx; S.cdy = q0.y - K.rowy; S.oh = -0.5f * q1.y; }
            // `never` is a wave-uniform, never-true condition the compiler cannot fold: the (untaken) scalar branch after every step ends
            // the basic block, so the 16 unrolled steps are scheduled one at a time -- as ONE block the scheduler overlaps them and the
            // kernel needs 280+ VGPRs (one wave per SIMD); split, every step's temporaries die inside the step.
            const bool never = p.gx == 0x7fffffff;
#define SP_STEP(I) sp_step<V, I, NACC>(K, pcrow, q0, q1, q2, q3, q4, S, valid, idx0, j, geo, ddelx_dx, ddely_dy, acc, okbits); sp_pin<TR::NPIN>(acc); if (never) asm volatile("s_nop 0");
#ifndef SP_DIAG_NO_STEPS      // diagnostic build only (make BLEND_EXTRA=-DSP_DIAG_NO_STEPS): what the kernel costs without its pixel steps (results are then wrong)
            SP_STEP(0) SP_STEP(1) SP_STEP(2) SP_STEP(3) SP_STEP(4) SP_STEP(5) SP_STEP(6) SP_STEP(7)
            SP_STEP(8) SP_STEP(9) SP_STEP(10) SP_STEP(11)
#endif
            // SURFEL: Tu, Tv (dead since sp_surf_setup) come back for sp_surf_finish; requested here, four steps ahead of their use
            float4 t0 = z4, t1 = z4;
            if constexpr (V == GSR_SURFEL) { t0 = r[0]; t1 = r[1]; }
#ifndef SP_DIAG_NO_STEPS
            SP_STEP(12) SP_STEP(13) SP_STEP(14) SP_STEP(15)
#endif
#undef SP_STEP
            if constexpr (V == GSR_EWA) {              // moments -> mean2D terms: 2 (A Sx + B Sy) W/2, 2 (C Sy + B Sx) H/2
                const float sx = acc[4], sy = acc[5];
                acc[4] = (2.0f * ddelx_dx) * fmaf(q0.w, sy, q0.z * sx);
                acc[5] = (2.0f * ddely_dy) * fmaf(q0.w, sx, q1.x * sy);
            }
            if constexpr (V == GSR_SURFEL) {
                if (mn_live) {
#define SP_MN(I) sp_mn_pixel<I>(K, okbits, acc);
                    SP_MN(0) SP_MN(1) SP_MN(2) SP_MN(3) SP_MN(4) SP_MN(5) SP_MN(6) SP_MN(7) SP_MN(8) SP_MN(9) SP_MN(10) SP_MN(11) SP_MN(12) SP_MN(13) SP_MN(14) SP_MN(15)
#undef SP_MN
                }
                float g9[9];
                sp_surf_finish(t0, t1, S.Tw2, K.rowx, K.rowy, acc, g9);
#pragma unroll
                for (int c = 0; c < 9; c++) acc[7 + c] = g9[c];
            }
            // Add the 16 x 4 (block, splat) partials of this load into the wave's table.  The same splat can sit in several ROWS of one
            // load (it reaches several blocks), never twice in one row: the four rows go one after the other, each a plain
            // read-add-write (DS operations of a wave execute in order) -- no LDS float atomics, which cost ~2 cycles per LANE
            // (18 x 64-lane ds_add_f32 per load kept the LDS pipe busy for a third of the first version's run time).
            const uint32_t cid = valid ? (uint32_t)mycidx[e] : 0u;
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                if (b == rr && valid) {
                    float2* t = mytab + cid * (TS / 2);
#pragma unroll
                    for (int c = 0; c < (NACC + 1) / 2; c++) { float2 v = t[c]; v.x += acc[2 * c]; v.y += acc[2 * c + 1]; t[c] = v; }      // (a row may be padded: TS >= NACC)
                }
            }
        }
        __syncthreads();
        // ------------------------------------------------------------ combine the four waves' tables; one 16-lane atomic per entry that received anything
        // (measured alternative: every wave flushing its own rows without this barrier -- 1.35 x the accumulator line operations -- 0.499 vs 0.506 ms)
        for (uint32_t e = lo_e + (threadIdx.x >> 4); e < n; e += 16) {
            const uint32_t c = threadIdx.x & 15u;
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const uint32_t ci = s_cidx[w * SP_CH + e];
                if (ci != 0xFFu) {
                    const float* t = reinterpret_cast<const float*>(s_wtab + (w * CAPV + ci) * (TS / 2));
                    if ((int)c < NACC) v0 += t[c];
                    if (NACC > 16 && (int)c < NACC - 16) v1 += t[16 + c];
                }
            }
            float* dst = p.acc + (size_t)s_ids[e] * AS;
#ifndef SP_DIAG_NO_FLUSH
            if (v0 != 0.f) atomic_addf(dst + c, v0);
            if (NACC > 16 && v1 != 0.f) atomic_addf(dst + 16 + c, v1);
#else
            if (v0 == 123.456f && v1 == 1.f) dst[c] = v0;
#endif
        }
        hi = cbase + lo_e;
        if (hi > 0) __syncthreads();
    }
}

// Stage profiler, single-kernel stage: start / stop events ATTACHED to the dispatch (hipExtLaunchKernel) instead of two hipEventRecord around it.
// An event record is a barrier packet of its own: ~6 us of stream idle time each, on both sides of the kernel it measures
// (profiles/r03_timeline_surfel.json) -- 1.3 % of the iteration bench.py times with this stage's profiling on.  Thread-local, consumed by the next launch.
static thread_local hipEvent_t t_ev_start = nullptr, t_ev_stop = nullptr;
void gsr_blend_bwd_sp_attach_events(hipEvent_t start, hipEvent_t stop) { t_ev_start = start; t_ev_stop = stop; }

int gsr_launch_blend_bwd_sp(const BlendParams& p, int variant, hipStream_t s)
{
    dim3 grid(p.gx * p.gy), block(256);
    if (t_ev_start && t_ev_stop) {
        hipEvent_t a = t_ev_start, b = t_ev_stop;
        t_ev_start = t_ev_stop = nullptr;
        switch (variant) {
        case GSR_EWA: hipExtLaunchKernelGGL(k_blend_bwd_sp<GSR_EWA>, grid, block, 0, s, a, b, 0, p); break;
        case GSR_PLANE: hipExtLaunchKernelGGL(k_blend_bwd_sp<GSR_PLANE>, grid, block, 0, s, a, b, 0, p); break;
        default: hipExtLaunchKernelGGL(k_blend_bwd_sp<GSR_SURFEL>, grid, block, 0, s, a, b, 0, p); break;
        }
        return 0;
    }
    switch (variant) {
    case GSR_EWA: hipLaunchKernelGGL(k_blend_bwd_sp<GSR_EWA>, grid, block, 0, s, p); break;
    case GSR_PLANE: hipLaunchKernelGGL(k_blend_bwd_sp<GSR_PLANE>, grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL(k_blend_bwd_sp<GSR_SURFEL>, grid, block, 0, s, p); break;
    }
    return 0;
}
