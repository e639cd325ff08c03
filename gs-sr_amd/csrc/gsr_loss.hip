// gsr_loss.hip -- fused photometric loss of the training loop, gfx950 only.
//
//   loss = (1 - lambda) * mean|img - gt|  +  lambda * (1 - SSIM(img, gt))            gssr/scene/vanilla_scene.py:29-69
//   SSIM: 11x11 Gaussian window (sigma 1.5), zero padding 5, per channel, C1 = 0.01^2, C2 = 0.03^2, mean over C*H*W.
//
// The reference builds it from five grouped conv2d calls and ~20 elementwise ops (and autograd replays all of them).  Here: two
// HBM-bound streaming kernels over 32x32 pixel tiles with a 5-pixel halo staged in LDS and a separable 11+11 tap filter
// (4 outputs per thread in each pass: 27 LDS reads per pixel instead of 91):
//   k_ssim_fwd : mu1, mu2, E[x^2], E[y^2], E[xy] -> SSIM map value (block-reduced into the loss) and the three partial-derivative
//                maps dS/dmu1, dS/dE[x^2], dS/dE[xy]
//   k_ssim_bwd : dL/dimg = w * dS/dmu1 + 2 img (w * dS/dE[x^2]) + gt (w * dS/dE[xy])  (the window is symmetric, so the adjoint of the
//                zero-padded correlation is the same correlation) + the L1 sign term.
// Algorithmic bytes per pixel-channel: fwd 8 read + 12 written, bwd 12 + 8 read + 4 written = 44 B.
#include "gsr_common.h"

#define SS_T 32                         // output tile width
#ifndef SS_TY
#define SS_TY 30                        // output tile height: 40 patch rows fit four workgroups' LDS on a CU (0.1328 ms against 0.1354 with 32)
#endif
#define SS_R 5
#define SS_P (SS_T + 2 * SS_R)        // 42: staged patch width
#define SS_PY (SS_TY + 2 * SS_R)      // staged patch height
#define SS_LD (SS_P + 1)
#define SS_NLD ((SS_PY * SS_P + 255) / 256)   // patch elements per thread
#define SS_B 4                          // outputs per thread in the horizontal pass (sliding window: 14 LDS reads feed 4 outputs, not 44)
// rows past the patch are only ever read for outputs that are masked off (SS_TY not a multiple of 8)
#define SS_VROW(r) ((SS_BV * 8 == SS_TY) ? (r) : min((r), SS_PY - 1))
#define SS_BV ((SS_TY + 7) / 8)         // output rows per thread in the vertical pass (8 row groups x 32 columns = 256 threads)

__constant__ float c_win[11] = {1.028380124e-03f, 7.598758209e-03f, 3.600077331e-02f, 1.093606874e-01f, 2.130055279e-01f, 2.660117149e-01f,
                                2.130055279e-01f, 1.093606874e-01f, 3.600077331e-02f, 7.598758209e-03f, 1.028380124e-03f};

// The 11 window taps in VGPRs: as wave-uniform constants the compiler keeps them in SGPRs, and on gfx950 a VALU instruction with an SGPR
// operand issues in ~4.2 cycles against ~2.7 for VGPR-only fma (tools/microbench/valu_rate.hip); every one of the ~440 filter fmas per
// thread has a tap as an operand.
#define SS_LOAD_TAPS(w)                                                          \
    float w[11];                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 11; i_++) { w[i_] = c_win[i_]; asm volatile("" : "+v"(w[i_])); }

__device__ __forceinline__ float block_sum256(float v, float* red)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// 32x32 outputs per 256-thread block.  Horizontal pass: a work item = one patch row x 4 adjacent output columns (42 rows x 8 groups);
// vertical pass: one output column x 4 adjacent output rows (32 x 8 = 256 items, one per thread).
__global__ void __launch_bounds__(256) k_ssim_fwd(int H, int W, const float* __restrict__ img, const float* __restrict__ gt,
                                                  float* __restrict__ maps /*[3][C][H][W]*/, size_t plane_all, float2* __restrict__ partial)
{
    __shared__ float sx[SS_PY][SS_LD], sy[SS_PY][SS_LD];
    __shared__ float h[5][SS_PY][SS_T + 1];
    __shared__ float red[4];
    SS_LOAD_TAPS(wt)
    const int c = blockIdx.z, x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_TY;
    const float* ip = img + (size_t)c * H * W;
    const float* gp = gt + (size_t)c * H * W;
    {   // All of a thread's patch loads are issued before the first LDS write: as a rolled loop with the bounds test as a branch this was seven
        // dependent HBM round trips per workgroup (ISA: load, s_waitcnt vmcnt(0), ds_write, branch), with three workgroups per CU to hide them.
        float vx[SS_NLD], vy[SS_NLD];
#pragma unroll
        for (int i = 0; i < SS_NLD; i++) {
            const int e = threadIdx.x + i * 256, ly = e / SS_P, lx = e % SS_P, gy = y0 + ly - SS_R, gx = x0 + lx - SS_R;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W && e < SS_PY * SS_P;
            const size_t o = in ? (size_t)gy * W + gx : 0;          // element 0 always exists: an unconditional load, then a select
            vx[i] = ip[o]; vy[i] = gp[o];
            if (!in) { vx[i] = 0.0f; vy[i] = 0.0f; }
        }
#pragma unroll
        for (int i = 0; i < SS_NLD; i++) {
            const int e = threadIdx.x + i * 256, ly = e / SS_P, lx = e % SS_P;
            if (e < SS_PY * SS_P) { sx[ly][lx] = vx[i]; sy[ly][lx] = vy[i]; }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < SS_PY * (SS_T / SS_B); e += 256) {
        const int ly = e / (SS_T / SS_B), lx = (e % (SS_T / SS_B)) * SS_B;
        float a[SS_B], b[SS_B], aa[SS_B], bb[SS_B], ab[SS_B];
#pragma unroll
        for (int o = 0; o < SS_B; o++) a[o] = b[o] = aa[o] = bb[o] = ab[o] = 0.f;
#pragma unroll
        for (int t = 0; t < 11 + SS_B - 1; t++) {
            const float u = sx[ly][lx + t], v = sy[ly][lx + t];
            const float uu = u * u, vv = v * v, uv = u * v;
#pragma unroll
            for (int o = 0; o < SS_B; o++) {
                if (t - o >= 0 && t - o < 11) {
                    const float w = wt[t - o];
                    a[o] = fmaf(w, u, a[o]); b[o] = fmaf(w, v, b[o]); aa[o] = fmaf(w, uu, aa[o]); bb[o] = fmaf(w, vv, bb[o]); ab[o] = fmaf(w, uv, ab[o]);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < SS_B; o++) { h[0][ly][lx + o] = a[o]; h[1][ly][lx + o] = b[o]; h[2][ly][lx + o] = aa[o]; h[3][ly][lx + o] = bb[o]; h[4][ly][lx + o] = ab[o]; }
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly0 = (threadIdx.x >> 5) * SS_BV, gx = x0 + lx;
    float q[5][SS_BV];
#pragma unroll
    for (int k = 0; k < 5; k++)
#pragma unroll
        for (int o = 0; o < SS_BV; o++) q[k][o] = 0.f;
#pragma unroll
    for (int t = 0; t < 11 + SS_BV - 1; t++) {
        float hv[5];
#pragma unroll
        for (int k = 0; k < 5; k++) hv[k] = h[k][SS_VROW(ly0 + t)][lx];
#pragma unroll
        for (int o = 0; o < SS_BV; o++) {
            if (t - o >= 0 && t - o < 11) {
                const float w = wt[t - o];
#pragma unroll
                for (int k = 0; k < 5; k++) q[k][o] = fmaf(w, hv[k], q[k][o]);
            }
        }
    }
    float ssim = 0.f, l1 = 0.f;
#pragma unroll
    for (int o = 0; o < SS_BV; o++) {
        const int gy = y0 + ly0 + o;
        if (gx < W && gy < H && ly0 + o < SS_TY) {
            const float mu1 = q[0][o], mu2 = q[1][o], e11 = q[2][o], e22 = q[3][o], e12 = q[4][o];
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
            const float s1 = e11 - mu1s, s2 = e22 - mu2s, s12 = e12 - mu12;
            const float A1 = 2.f * mu12 + C1, A2 = 2.f * s12 + C2, B1 = mu1s + mu2s + C1, B2 = s1 + s2 + C2;
            const float inv = 1.0f / (B1 * B2);
            const float sv = A1 * A2 * inv;
            ssim += sv;
            // S as a function of (mu1, E[x^2], E[xy]) with mu2, E[y^2] fixed
            const float dmu = 2.f * mu2 * (A2 - A1) * inv - sv * 2.f * mu1 * (1.0f / B1 - 1.0f / B2);
            const float d11 = -sv / B2;
            const float d12 = 2.f * A1 * inv;
            const size_t oo = ((size_t)c * H + gy) * W + gx;
            maps[oo] = dmu; maps[plane_all + oo] = d11; maps[2 * plane_all + oo] = d12;
            l1 += fabsf(sx[ly0 + o + SS_R][lx + SS_R] - sy[ly0 + o + SS_R][lx + SS_R]);
        }
    }
    const float ts = block_sum256(ssim, red);
    __syncthreads();
    const float tl = block_sum256(l1, red);
    // one partial per block, summed by k_ssim_finish: same-address atomics from thousands of blocks serialise (measured 0.5 ms)
    if (threadIdx.x == 0) partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = make_float2(tl, ts);
}

__global__ void __launch_bounds__(256) k_ssim_bwd(int H, int W, const float* __restrict__ img, const float* __restrict__ gt,
                                                  const float* __restrict__ maps, size_t plane_all, float w_l1, float w_ssim,
                                                  float* __restrict__ dimg)
{
    __shared__ float sm[3][SS_PY][SS_LD];
    __shared__ float h[3][SS_PY][SS_T + 1];
    SS_LOAD_TAPS(wt)
    const int c = blockIdx.z, x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_TY;
    {   // loads first, LDS writes after (see k_ssim_fwd)
        float vm[3][SS_NLD];
#pragma unroll
        for (int i = 0; i < SS_NLD; i++) {
            const int e = threadIdx.x + i * 256, ly = e / SS_P, lx = e % SS_P, gy = y0 + ly - SS_R, gx = x0 + lx - SS_R;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W && e < SS_PY * SS_P;
            const size_t o = ((size_t)c * H + (in ? gy : 0)) * W + (in ? gx : 0);
#pragma unroll
            for (int m = 0; m < 3; m++) { vm[m][i] = maps[m * plane_all + o]; if (!in) vm[m][i] = 0.0f; }
        }
#pragma unroll
        for (int i = 0; i < SS_NLD; i++) {
            const int e = threadIdx.x + i * 256, ly = e / SS_P, lx = e % SS_P;
            if (e < SS_PY * SS_P) {
#pragma unroll
                for (int m = 0; m < 3; m++) sm[m][ly][lx] = vm[m][i];
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < SS_PY * (SS_T / SS_B); e += 256) {
        const int ly = e / (SS_T / SS_B), lx = (e % (SS_T / SS_B)) * SS_B;
        float a[3][SS_B];
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int o = 0; o < SS_B; o++) a[m][o] = 0.f;
#pragma unroll
        for (int t = 0; t < 11 + SS_B - 1; t++) {
            const float v0 = sm[0][ly][lx + t], v1 = sm[1][ly][lx + t], v2 = sm[2][ly][lx + t];
#pragma unroll
            for (int o = 0; o < SS_B; o++) {
                if (t - o >= 0 && t - o < 11) {
                    const float w = wt[t - o];
                    a[0][o] = fmaf(w, v0, a[0][o]); a[1][o] = fmaf(w, v1, a[1][o]); a[2][o] = fmaf(w, v2, a[2][o]);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int o = 0; o < SS_B; o++) h[m][ly][lx + o] = a[m][o];
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly0 = (threadIdx.x >> 5) * SS_BV, gx = x0 + lx;
    float q[3][SS_BV];
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int o = 0; o < SS_BV; o++) q[m][o] = 0.f;
#pragma unroll
    for (int t = 0; t < 11 + SS_BV - 1; t++) {
        const int r = SS_VROW(ly0 + t);
        const float v0 = h[0][r][lx], v1 = h[1][r][lx], v2 = h[2][r][lx];
#pragma unroll
        for (int o = 0; o < SS_BV; o++) {
            if (t - o >= 0 && t - o < 11) {
                const float w = wt[t - o];
                q[0][o] = fmaf(w, v0, q[0][o]); q[1][o] = fmaf(w, v1, q[1][o]); q[2][o] = fmaf(w, v2, q[2][o]);
            }
        }
    }
    if (gx >= W) return;
#pragma unroll
    for (int o = 0; o < SS_BV; o++) {
        const int gy = y0 + ly0 + o;
        if (gy < H && ly0 + o < SS_TY) {
            const size_t oo = ((size_t)c * H + gy) * W + gx;
            const float x = img[oo], y = gt[oo], df = x - y;
            const float sgn = df > 0.f ? 1.0f : (df < 0.f ? -1.0f : 0.0f);
            dimg[oo] = w_l1 * sgn - w_ssim * (q[0][o] + 2.f * x * q[1][o] + y * q[2][o]);
        }
    }
}

__global__ void __launch_bounds__(1024) k_ssim_finish(const float2* __restrict__ partial, int n, float* loss, float inv_n, float lambda)
{
    __shared__ float r1[16], r2[16];
    float a = 0.f, b = 0.f;
    for (int i0 = 0; i0 < n; i0 += 8 * 1024) {           // eight loads in flight per thread, summed in index order
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = i0 + u * 1024 + threadIdx.x; v[u] = partial[i < n ? i : 0]; if (i >= n) v[u] = make_float2(0.f, 0.f); }
#pragma unroll
        for (int u = 0; u < 8; u++) { a += v[u].x; b += v[u].y; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = a; r2[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float sa = 0.f, sb = 0.f;
        for (int w = 0; w < 16; w++) { sa += r1[w]; sb += r2[w]; }
        const float l1 = sa * inv_n, ss = sb * inv_n;
        loss[0] = l1; loss[1] = ss; loss[2] = (1.0f - lambda) * l1 + lambda * (1.0f - ss);
    }
}

static size_t ssim_blocks(int32_t C, int32_t H, int32_t W) { return (size_t)gsr_div_up(W, SS_T) * gsr_div_up(H, SS_TY) * C; }
extern "C" size_t gsr_loss_l1_ssim_scratch_bytes(int32_t C, int32_t H, int32_t W)
{
    return (C > 0 && H > 0 && W > 0) ? gsr_align((size_t)3 * C * H * W * sizeof(float)) + ssim_blocks(C, H, W) * sizeof(float2) : 0;
}

extern "C" int gsr_loss_l1_ssim(int32_t C, int32_t H, int32_t W, const float* img, const float* gt, float lambda_dssim, float* loss_out,
                                float* dL_dimg, void* scratch, size_t scratch_bytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (C <= 0 || H <= 0 || W <= 0) { gsr_set_error("loss_l1_ssim: bad sizes C=%d H=%d W=%d", C, H, W); return 1; }
    if (!img || !gt || !loss_out || !dL_dimg || !scratch || scratch_bytes < gsr_loss_l1_ssim_scratch_bytes(C, H, W)) {
        gsr_set_error("loss_l1_ssim: null pointer or scratch too small"); return 1;
    }
    const size_t plane_all = (size_t)C * H * W;
    const float inv_n = 1.0f / (float)plane_all;
    float2* partial = (float2*)((char*)scratch + gsr_align(3 * plane_all * sizeof(float)));
    const dim3 grid(gsr_div_up(W, SS_T), gsr_div_up(H, SS_TY), C);
    hipLaunchKernelGGL(k_ssim_fwd, grid, dim3(256), 0, s, H, W, img, gt, (float*)scratch, plane_all, partial);
    hipLaunchKernelGGL(k_ssim_bwd, grid, dim3(256), 0, s, H, W, img, gt, (const float*)scratch, plane_all, (1.0f - lambda_dssim) * inv_n,
                       lambda_dssim * inv_n, dL_dimg);
    hipLaunchKernelGGL(k_ssim_finish, dim3(1), dim3(1024), 0, s, (const float2*)partial, (int)ssim_blocks(C, H, W), loss_out, inv_n, lambda_dssim);
    return gsr_check_launch("loss_l1_ssim", s, false);
}

// ------------------------------------------------------------------------------------------------------------------------------
// 2DGS geometric regularisers in ONE kernel (forward values, loss, and dL/dallmap):
//   TwoDGSScene.render post-processing (gssr/scene/twodgs_scene.py:88-115), depth_to_normal (gssr/utils/point_utils.py:9-37),
//   normal + distortion losses (twodgs_scene.py:25-35).  ~35 N-sized torch ops (+ autograd) in the reference.
// GEO_TX x GEO_TY pixel tile per block; the 3x3 normal stencil and its adjoint (a 13-point diamond in depth) are staged through LDS:
//   P   (tile + 4) = depth * ray                   halo 2
//   g   (tile + 2) = d loss / d (dx, dy) per pixel halo 1     (zero for non-interior pixels)
// then every pixel gathers  dP = g_dx(y-1,x) - g_dx(y+1,x) + g_dy(y,x-1) - g_dy(y,x+1)  and chains it into allmap channels 0, 1, 5.
// Algorithmic bytes: 7 channels read + 11 written (+ optional outputs) = 72 B per pixel.
#ifndef GEO_TX
#define GEO_TX 64                      // 64 x 4 pixels: rows of whole 256-byte segments (16 x 16: 0.0567 ms, 32 x 8: 0.0526, 64 x 4: 0.0508; tools/ab_loss_tiles.sh)
#define GEO_TY 4
#endif
#define GEO_PX (GEO_TX + 4)    // P tile, halo 2
#define GEO_PY (GEO_TY + 4)
#define GEO_GX (GEO_TX + 2)    // g tile, halo 1
#define GEO_GY (GEO_TY + 2)
#define GEO_NLD ((GEO_PX * GEO_PY + 255) / 256)
static_assert(GEO_TX * GEO_TY == 256 && (GEO_TX & (GEO_TX - 1)) == 0, "one thread per pixel of the tile");

struct GeoArgs {
    int H, W;
    const float* allmap; const float* ray_mat; const float* normal_rot;
    float depth_ratio, wn, wd;
    float2* partial; float* dL; float* o_depth; float* o_nw; float* o_sn;
};

__device__ __forceinline__ float nan0(float v) { return (isnan(v) || isinf(v)) ? 0.0f : v; }

__global__ void __launch_bounds__(256) k_surfel_geo(GeoArgs p)
{
    __shared__ float sP[3][GEO_PY][GEO_PX + 1];
    __shared__ float sG[6][GEO_GY][GEO_GX + 1];
    __shared__ float sDot[GEO_GY][GEO_GX + 1];
    __shared__ float sN[3][GEO_GY][GEO_GX + 1];
    __shared__ float red[4];
    typedef const float __attribute__((address_space(4))) * cfp;
    float rm[9], nr[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { rm[i] = ((cfp)p.ray_mat)[i]; nr[i] = ((cfp)p.normal_rot)[i]; }
    const int H = p.H, W = p.W, x0 = blockIdx.x * GEO_TX, y0 = blockIdx.y * GEO_TY;
    const size_t N = (size_t)H * W;
    {   // loads of every round first (clamped address + select instead of a branch), LDS writes after: see k_ssim_fwd
        float va[GEO_NLD], v0[GEO_NLD], v5[GEO_NLD];
#pragma unroll
        for (int i = 0; i < GEO_NLD; i++) {
            const int e = threadIdx.x + i * 256, ly = e / GEO_PX, lx = e % GEO_PX, gy = y0 + ly - 2, gx = x0 + lx - 2;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W && e < GEO_PX * GEO_PY;
            const size_t o = in ? (size_t)gy * W + gx : 0;
            va[i] = p.allmap[N + o]; v0[i] = p.allmap[o]; v5[i] = p.allmap[5 * N + o];
        }
#pragma unroll
        for (int i = 0; i < GEO_NLD; i++) {
            const int e = threadIdx.x + i * 256, ly = e / GEO_PX, lx = e % GEO_PX, gy = y0 + ly - 2, gx = x0 + lx - 2;
            if (e < GEO_PX * GEO_PY) {
                float P0 = 0.f, P1 = 0.f, P2 = 0.f;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                    const float d = nan0(v0[i] / va[i]) * (1.0f - p.depth_ratio) + p.depth_ratio * nan0(v5[i]);
                    const float fx = (float)gx, fy = (float)gy;
                    P0 = d * (fx * rm[0] + fy * rm[3] + rm[6]); P1 = d * (fx * rm[1] + fy * rm[4] + rm[7]); P2 = d * (fx * rm[2] + fy * rm[5] + rm[8]);
                }
                sP[0][ly][lx] = P0; sP[1][ly][lx] = P1; sP[2][ly][lx] = P2;
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < GEO_GX * GEO_GY; e += 256) {
        const int ly = e / GEO_GX, lx = e % GEO_GX, gy = y0 + ly - 1, gx = x0 + lx - 1;
        float g[6] = {0, 0, 0, 0, 0, 0}, n[3] = {0, 0, 0}, dot = 0.f;
        if (gy >= 1 && gy <= H - 2 && gx >= 1 && gx <= W - 2) {
            const size_t o = (size_t)gy * W + gx;
            const float a = p.allmap[N + o];
            const float nv0 = p.allmap[2 * N + o], nv1 = p.allmap[3 * N + o], nv2 = p.allmap[4 * N + o];
            float nw[3];
#pragma unroll
            for (int c = 0; c < 3; c++) nw[c] = nv0 * nr[c] + nv1 * nr[3 + c] + nv2 * nr[6 + c];
            const int py = ly + 1, px = lx + 1;                 // position in the P tile
            float dx[3], dy[3];
#pragma unroll
            for (int c = 0; c < 3; c++) { dx[c] = sP[c][py + 1][px] - sP[c][py - 1][px]; dy[c] = sP[c][py][px + 1] - sP[c][py][px - 1]; }
            const float c0 = dx[1] * dy[2] - dx[2] * dy[1], c1 = dx[2] * dy[0] - dx[0] * dy[2], c2 = dx[0] * dy[1] - dx[1] * dy[0];
            const float len = sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
            const float inv = 1.0f / fmaxf(len, 1e-12f);
            n[0] = c0 * inv; n[1] = c1 * inv; n[2] = c2 * inv;
            dot = a * (nw[0] * n[0] + nw[1] * n[1] + nw[2] * n[2]);
            float dn[3], dc[3];
#pragma unroll
            for (int c = 0; c < 3; c++) dn[c] = -p.wn * a * nw[c];
            const float nd = len > 1e-12f ? n[0] * dn[0] + n[1] * dn[1] + n[2] * dn[2] : 0.0f;
#pragma unroll
            for (int c = 0; c < 3; c++) dc[c] = (dn[c] - n[c] * nd) * inv;
            g[0] = dy[1] * dc[2] - dy[2] * dc[1]; g[1] = dy[2] * dc[0] - dy[0] * dc[2]; g[2] = dy[0] * dc[1] - dy[1] * dc[0];
            g[3] = dc[1] * dx[2] - dc[2] * dx[1]; g[4] = dc[2] * dx[0] - dc[0] * dx[2]; g[5] = dc[0] * dx[1] - dc[1] * dx[0];
            n[0] *= a; n[1] *= a; n[2] *= a;                   // surf_normal = n * alpha
        }
#pragma unroll
        for (int c = 0; c < 6; c++) sG[c][ly][lx] = g[c];
#pragma unroll
        for (int c = 0; c < 3; c++) sN[c][ly][lx] = n[c];
        sDot[ly][lx] = dot;
    }
    __syncthreads();
    const int lx = threadIdx.x % GEO_TX, ly = threadIdx.x / GEO_TX, gx = x0 + lx, gy = y0 + ly;
    float err = 0.f, dist = 0.f;
    if (gx < W && gy < H) {
        const size_t o = (size_t)gy * W + gx;
        const int qy = ly + 1, qx = lx + 1;                    // position in the g tile
        float dP[3];
#pragma unroll
        for (int c = 0; c < 3; c++) dP[c] = sG[c][qy - 1][qx] - sG[c][qy + 1][qx] + sG[3 + c][qy][qx - 1] - sG[3 + c][qy][qx + 1];
        const float fx = (float)gx, fy = (float)gy;
        const float dd = dP[0] * (fx * rm[0] + fy * rm[3] + rm[6]) + dP[1] * (fx * rm[1] + fy * rm[4] + rm[7]) + dP[2] * (fx * rm[2] + fy * rm[5] + rm[8]);
        const float a0 = p.allmap[o], a = p.allmap[N + o], m = p.allmap[5 * N + o], q = a0 / a;
        const bool okq = !(isnan(q) || isinf(q)), okm = !(isnan(m) || isinf(m));
        p.dL[o] = okq ? dd * (1.0f - p.depth_ratio) / a : 0.0f;
        p.dL[N + o] = okq ? -dd * (1.0f - p.depth_ratio) * a0 / (a * a) : 0.0f;
        p.dL[5 * N + o] = okm ? dd * p.depth_ratio : 0.0f;
        const float s0 = sN[0][qy][qx], s1 = sN[1][qy][qx], s2 = sN[2][qy][qx];
#pragma unroll
        for (int i = 0; i < 3; i++) p.dL[(2 + i) * N + o] = -p.wn * (nr[3 * i] * s0 + nr[3 * i + 1] * s1 + nr[3 * i + 2] * s2);
        p.dL[6 * N + o] = p.wd;
#pragma unroll
        for (int c = 7; c < 11; c++) p.dL[c * N + o] = 0.0f;
        err = 1.0f - sDot[qy][qx];
        dist = p.allmap[6 * N + o];
        if (p.o_depth) p.o_depth[o] = (okq ? q : 0.0f) * (1.0f - p.depth_ratio) + p.depth_ratio * (okm ? m : 0.0f);
        if (p.o_sn) { p.o_sn[o] = s0; p.o_sn[N + o] = s1; p.o_sn[2 * N + o] = s2; }
        if (p.o_nw) {
            const float nv0 = p.allmap[2 * N + o], nv1 = p.allmap[3 * N + o], nv2 = p.allmap[4 * N + o];
#pragma unroll
            for (int c = 0; c < 3; c++) p.o_nw[c * N + o] = nv0 * nr[c] + nv1 * nr[3 + c] + nv2 * nr[6 + c];
        }
    }
    const float te = block_sum256(err, red);
    __syncthreads();
    const float td = block_sum256(dist, red);
    if (threadIdx.x == 0) p.partial[blockIdx.y * gridDim.x + blockIdx.x] = make_float2(te, td);
}

__global__ void __launch_bounds__(1024) k_geo_finish(const float2* __restrict__ partial, int n, float* loss, float inv_n, float ln, float ld)
{
    __shared__ float r1[16], r2[16];
    float a = 0.f, b = 0.f;
    for (int i0 = 0; i0 < n; i0 += 8 * 1024) {           // eight loads in flight per thread, summed in index order
        float2 q[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = i0 + u * 1024 + threadIdx.x; q[u] = partial[i < n ? i : 0]; if (i >= n) q[u] = make_float2(0.f, 0.f); }
#pragma unroll
        for (int u = 0; u < 8; u++) { a += q[u].x; b += q[u].y; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = a; r2[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float sa = 0.f, sb = 0.f;
        for (int w = 0; w < 16; w++) { sa += r1[w]; sb += r2[w]; }
        loss[0] = sa * inv_n; loss[1] = sb * inv_n; loss[2] = ln * loss[0] + ld * loss[1];
    }
}

extern "C" size_t gsr_loss_surfel_geo_scratch_bytes(int32_t H, int32_t W)
{
    return (H > 0 && W > 0) ? (size_t)gsr_div_up(W, GEO_TX) * gsr_div_up(H, GEO_TY) * sizeof(float2) : 0;
}

extern "C" int gsr_loss_surfel_geo(int32_t H, int32_t W, const float* allmap, const float* ray_mat, const float* normal_rot, float depth_ratio,
                                   float lambda_normal, float lambda_dist, float* loss_out, float* dL_dallmap, float* out_surf_depth,
                                   float* out_normal_world, float* out_surf_normal, void* scratch, size_t scratch_bytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (H <= 0 || W <= 0) { gsr_set_error("loss_surfel_geo: bad sizes H=%d W=%d", H, W); return 1; }
    if (!allmap || !ray_mat || !normal_rot || !loss_out || !dL_dallmap || !scratch || scratch_bytes < gsr_loss_surfel_geo_scratch_bytes(H, W)) {
        gsr_set_error("loss_surfel_geo: null pointer or scratch too small"); return 1;
    }
    GeoArgs a;
    const float inv_n = 1.0f / ((float)H * (float)W);
    a.H = H; a.W = W; a.allmap = allmap; a.ray_mat = ray_mat; a.normal_rot = normal_rot; a.depth_ratio = depth_ratio;
    a.wn = lambda_normal * inv_n; a.wd = lambda_dist * inv_n; a.partial = (float2*)scratch; a.dL = dL_dallmap;
    a.o_depth = out_surf_depth; a.o_nw = out_normal_world; a.o_sn = out_surf_normal;
    const dim3 grid(gsr_div_up(W, GEO_TX), gsr_div_up(H, GEO_TY));
    hipLaunchKernelGGL(k_surfel_geo, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_geo_finish, dim3(1), dim3(1024), 0, s, (const float2*)scratch, (int)(grid.x * grid.y), loss_out, inv_n, lambda_normal,
                       lambda_dist);
    return gsr_check_launch("loss_surfel_geo", s, false);
}

// ------------------------------------------------------------------------------------------------------------------------------
// PGSR single-view normal regulariser (gssr/scene/pgsr_scene.py:105-112,227-238,320; gssr/utils/graphics_utils.py:80-146):
//   P = plane_depth * ([x y 1] * ray_mat)  (camera space, ray_mat = inverse(K^T));  depth_normal = normalize(cross(right - left, top -
//   bottom)) * alpha (alpha detached, zero on the border) -- the same vector as the 2DGS stencil above;
//   loss = lambda * mean(weight * sum_c |depth_normal_c - normal_c|).
// Same tiling as k_surfel_geo: value, dL/dplane_depth and dL/dnormal in one kernel.
struct PlaneGeoArgs {
    int H, W;
    const float* depth; const float* alpha; const float* normal; const float* weight; const float* ray_mat;
    float wl;
    float2* partial; float* dDepth; float* dNormal; float* o_dn;
};
__device__ __forceinline__ float sgn_(float v) { return v > 0.f ? 1.0f : (v < 0.f ? -1.0f : 0.0f); }

__global__ void __launch_bounds__(256) k_plane_geo(PlaneGeoArgs p)
{
    __shared__ float sP[3][GEO_PY][GEO_PX + 1];
    __shared__ float sG[6][GEO_GY][GEO_GX + 1];
    __shared__ float sN[3][GEO_GY][GEO_GX + 1];
    __shared__ float red[4];
    typedef const float __attribute__((address_space(4))) * cfp;
    float rm[9];
#pragma unroll
    for (int i = 0; i < 9; i++) rm[i] = ((cfp)p.ray_mat)[i];
    const int H = p.H, W = p.W, x0 = blockIdx.x * GEO_TX, y0 = blockIdx.y * GEO_TY;
    const size_t N = (size_t)H * W;
    {
        float vd[GEO_NLD];
#pragma unroll
        for (int i = 0; i < GEO_NLD; i++) {
            const int e = threadIdx.x + i * 256, ly = e / GEO_PX, lx = e % GEO_PX, gy = y0 + ly - 2, gx = x0 + lx - 2;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W && e < GEO_PX * GEO_PY;
            vd[i] = p.depth[in ? (size_t)gy * W + gx : 0];
            if (!in) vd[i] = 0.0f;
        }
#pragma unroll
        for (int i = 0; i < GEO_NLD; i++) {
            const int e = threadIdx.x + i * 256, ly = e / GEO_PX, lx = e % GEO_PX, gy = y0 + ly - 2, gx = x0 + lx - 2;
            if (e < GEO_PX * GEO_PY) {
                const float d = vd[i], fx = (float)gx, fy = (float)gy;
                sP[0][ly][lx] = d * (fx * rm[0] + fy * rm[3] + rm[6]); sP[1][ly][lx] = d * (fx * rm[1] + fy * rm[4] + rm[7]);
                sP[2][ly][lx] = d * (fx * rm[2] + fy * rm[5] + rm[8]);
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < GEO_GX * GEO_GY; e += 256) {
        const int ly = e / GEO_GX, lx = e % GEO_GX, gy = y0 + ly - 1, gx = x0 + lx - 1;
        float g[6] = {0, 0, 0, 0, 0, 0}, n[3] = {0, 0, 0};
        if (gy >= 1 && gy <= H - 2 && gx >= 1 && gx <= W - 2) {
            const size_t o = (size_t)gy * W + gx;
            const float a = p.alpha[o], wl = p.wl * (p.weight ? p.weight[o] : 1.0f);
            const int py = ly + 1, px = lx + 1;
            float dx[3], dy[3];
#pragma unroll
            for (int c = 0; c < 3; c++) { dx[c] = sP[c][py + 1][px] - sP[c][py - 1][px]; dy[c] = sP[c][py][px + 1] - sP[c][py][px - 1]; }
            const float c0 = dx[1] * dy[2] - dx[2] * dy[1], c1 = dx[2] * dy[0] - dx[0] * dy[2], c2 = dx[0] * dy[1] - dx[1] * dy[0];
            const float len = sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
            const float inv = 1.0f / fmaxf(len, 1e-12f);
            n[0] = c0 * inv; n[1] = c1 * inv; n[2] = c2 * inv;
            float dn[3], dc[3];
#pragma unroll
            for (int c = 0; c < 3; c++) dn[c] = wl * a * sgn_(a * n[c] - p.normal[c * N + o]);
            const float nd = len > 1e-12f ? n[0] * dn[0] + n[1] * dn[1] + n[2] * dn[2] : 0.0f;
#pragma unroll
            for (int c = 0; c < 3; c++) dc[c] = (dn[c] - n[c] * nd) * inv;
            g[0] = dy[1] * dc[2] - dy[2] * dc[1]; g[1] = dy[2] * dc[0] - dy[0] * dc[2]; g[2] = dy[0] * dc[1] - dy[1] * dc[0];
            g[3] = dc[1] * dx[2] - dc[2] * dx[1]; g[4] = dc[2] * dx[0] - dc[0] * dx[2]; g[5] = dc[0] * dx[1] - dc[1] * dx[0];
            n[0] *= a; n[1] *= a; n[2] *= a;
        }
#pragma unroll
        for (int c = 0; c < 6; c++) sG[c][ly][lx] = g[c];
#pragma unroll
        for (int c = 0; c < 3; c++) sN[c][ly][lx] = n[c];
    }
    __syncthreads();
    const int lx = threadIdx.x % GEO_TX, ly = threadIdx.x / GEO_TX, gx = x0 + lx, gy = y0 + ly;
    float err = 0.f;
    if (gx < W && gy < H) {
        const size_t o = (size_t)gy * W + gx;
        const int qy = ly + 1, qx = lx + 1;
        float dP[3];
#pragma unroll
        for (int c = 0; c < 3; c++) dP[c] = sG[c][qy - 1][qx] - sG[c][qy + 1][qx] + sG[3 + c][qy][qx - 1] - sG[3 + c][qy][qx + 1];
        const float fx = (float)gx, fy = (float)gy;
        p.dDepth[o] = dP[0] * (fx * rm[0] + fy * rm[3] + rm[6]) + dP[1] * (fx * rm[1] + fy * rm[4] + rm[7]) + dP[2] * (fx * rm[2] + fy * rm[5] + rm[8]);
        const float w = p.weight ? p.weight[o] : 1.0f, wl = p.wl * w;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float dn = sN[c][qy][qx] - p.normal[c * N + o];
            err += fabsf(dn);
            p.dNormal[c * N + o] = -wl * sgn_(dn);
            if (p.o_dn) p.o_dn[c * N + o] = sN[c][qy][qx];
        }
        err *= w;
    }
    const float te = block_sum256(err, red);
    if (threadIdx.x == 0) p.partial[blockIdx.y * gridDim.x + blockIdx.x] = make_float2(te, 0.f);
}

extern "C" int gsr_loss_plane_geo(int32_t H, int32_t W, const float* plane_depth, const float* alpha, const float* normal, const float* weight,
                                  const float* ray_mat, float lambda_normal, float* loss_out, float* dL_ddepth, float* dL_dnormal,
                                  float* out_depth_normal, void* scratch, size_t scratch_bytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (H <= 0 || W <= 0) { gsr_set_error("loss_plane_geo: bad sizes H=%d W=%d", H, W); return 1; }
    if (!plane_depth || !alpha || !normal || !ray_mat || !loss_out || !dL_ddepth || !dL_dnormal || !scratch ||
        scratch_bytes < gsr_loss_surfel_geo_scratch_bytes(H, W)) {
        gsr_set_error("loss_plane_geo: null pointer or scratch too small"); return 1;
    }
    PlaneGeoArgs a;
    const float inv_n = 1.0f / ((float)H * (float)W);
    a.H = H; a.W = W; a.depth = plane_depth; a.alpha = alpha; a.normal = normal; a.weight = weight; a.ray_mat = ray_mat;
    a.wl = lambda_normal * inv_n; a.partial = (float2*)scratch; a.dDepth = dL_ddepth; a.dNormal = dL_dnormal; a.o_dn = out_depth_normal;
    const dim3 grid(gsr_div_up(W, GEO_TX), gsr_div_up(H, GEO_TY));
    hipLaunchKernelGGL(k_plane_geo, grid, dim3(256), 0, s, a);
    // loss_out = {mean weighted L1, 0, lambda * mean}
    hipLaunchKernelGGL(k_geo_finish, dim3(1), dim3(1024), 0, s, (const float2*)scratch, (int)(grid.x * grid.y), loss_out, inv_n, lambda_normal, 0.0f);
    return gsr_check_launch("loss_plane_geo", s, false);
}

// ---- scaling regulariser: lambda * mean_i prod_{c < cols} scaling[i, c]  (scaffold_scene.py:184, scaffold_2dgs_scene.py:25, scaffold_pgsr_scene.py:20,
// octree_2dgs_scene.py:25, octree_pgsr_scene.py:23: `lambda_scaling * outputs["scaling"].prod(dim=1).mean()`), value and gradient in one pass.
// d/d scaling[i, c] = lambda / n * product of the row's OTHER columns (what prod's backward gives, without its division -- and without the
// `nonzero` host synchronisation torch's prod backward runs when an entry is 0).  `stride` >= cols floats per row: the first `cols` of a wider
// tensor (2DGS uses two of the decode's three) without a copy; the gradient of the remaining columns is written as 0.  With count_dev (static-shape
// iterations) only the first *count_dev rows are live: the mean divides by that, the rows behind get no value and zero gradient.  loss_out must be zero on entry (one atomic per block).
__global__ void __launch_bounds__(256) k_scaling_prod(int64_t P, int cols, int stride, const float* __restrict__ sc, const int32_t* __restrict__ count_dev,
                                                      float lambda, float* __restrict__ loss_out, float* __restrict__ grad)
{
    __shared__ float red[4];
    const int64_t live = count_dev ? (int64_t)max(*count_dev, 0) : P;          // rows [live, P) are parked: no value, zero gradient
    const float wgt = lambda / (float)max(live, (int64_t)1);
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const float* r = sc + i * stride;
        float* g = grad + i * stride;
        const float x = r[0], y = cols > 1 ? r[1] : 1.f, z = cols > 2 ? r[2] : 1.f;
        const float wi = i < live ? wgt : 0.f;
        if (i < live) acc += (x * y) * z;
        g[0] = wi * (y * z);
        if (cols > 1) g[1] = wi * (x * z);
        if (cols > 2) g[2] = wi * (x * y);
        for (int c = cols; c < stride; c++) g[c] = 0.f;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(loss_out, wgt * ((red[0] + red[1]) + (red[2] + red[3])));
}

extern "C" int gsr_loss_scaling_prod(int64_t P, int32_t cols, int32_t stride, const float* scaling, const int32_t* count_dev, float lambda_scaling,
                                     float* loss_out, float* dL_dscaling, void* stream)
{
    if (P < 0 || cols < 1 || cols > 3 || stride < cols) { gsr_set_error("loss_scaling_prod: P=%lld cols=%d stride=%d", (long long)P, cols, stride); return 1; }
    if (P == 0) return 0;
    if (!scaling || !loss_out || !dL_dscaling) { gsr_set_error("loss_scaling_prod: null pointer"); return 1; }
    const uint32_t blocks = (uint32_t)min((int64_t)1024, (P + 255) / 256);
    hipLaunchKernelGGL(k_scaling_prod, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, (int)cols, (int)stride, scaling, count_dev, lambda_scaling,
                       loss_out, dL_dscaling);
    return gsr_check_launch("loss_scaling_prod", (hipStream_t)stream, false);
}
