// gsr_extra.hip -- adjacent kernels of the hot path:
//   gsr_tsdf_integrate : per-frame TSDF voxel update, the in-repo definition gssr/utils/mesh_utils.py:195-246
//   gsr_dist2          : simple_knn.distCUDA2 (mean squared distance to the 3 nearest neighbours)
#include <algorithm>
#include "gsr_common.h"

// torch.nn.functional.grid_sample(mode='bilinear', padding_mode='border', align_corners=True), one sample.
// Coordinates are clamped to [0, size-1]; corner weights as ATen: (x1 - x), (x - x0); out-of-range corners add 0.
__device__ __forceinline__ float bilinear_border(const float* __restrict__ img, int W, int H, float u, float v)
{
    float x = ((u + 1.f) / 2.f) * (float)(W - 1);
    float y = ((v + 1.f) / 2.f) * (float)(H - 1);
    x = fminf(fmaxf(x, 0.f), (float)(W - 1));
    y = fminf(fmaxf(y, 0.f), (float)(H - 1));
    const int x0 = (int)floorf(x), y0 = (int)floorf(y);
    const int x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = x - (float)x0, wy1 = y - (float)y0, wx0 = (float)x1 - x, wy0 = (float)y1 - y;
    float acc = 0.f;
    if (x0 < W && y0 < H) acc += img[(size_t)y0 * W + x0] * (wx0 * wy0);
    if (x1 < W && y0 < H) acc += img[(size_t)y0 * W + x1] * (wx1 * wy0);
    if (x0 < W && y1 < H) acc += img[(size_t)y1 * W + x0] * (wx0 * wy1);
    if (x1 < W && y1 < H) acc += img[(size_t)y1 * W + x1] * (wx1 * wy1);
    return acc;
}

// Interleaved (r,g,b,d) texels: one 16-byte load per bilinear corner instead of four dword gathers.
__global__ void __launch_bounds__(256) k_pack_rgbd(int64_t n, const float* __restrict__ depth, const float* __restrict__ rgb, float4* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_float4(rgb[i], rgb[n + i], rgb[2 * n + i], depth[i]);
}
// same clamping, weights and corner order as bilinear_border, on all four channels at once (bit-identical per channel)
__device__ __forceinline__ float4 bilinear_border4(const float4* __restrict__ img, int W, int H, float u, float v)
{
    float x = ((u + 1.f) / 2.f) * (float)(W - 1);
    float y = ((v + 1.f) / 2.f) * (float)(H - 1);
    x = fminf(fmaxf(x, 0.f), (float)(W - 1));
    y = fminf(fmaxf(y, 0.f), (float)(H - 1));
    const int x0 = (int)floorf(x), y0 = (int)floorf(y);
    const int x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = x - (float)x0, wy1 = y - (float)y0, wx0 = (float)x1 - x, wy0 = (float)y1 - y;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add = [&](int xx, int yy, float w) {
        if (xx < W && yy < H) {
            const float4 t = img[(size_t)yy * W + xx];
            acc.x += t.x * w; acc.y += t.y * w; acc.z += t.z * w; acc.w += t.w * w;
        }
    };
    add(x0, y0, wx0 * wy0); add(x1, y0, wx1 * wy0); add(x0, y1, wx0 * wy1); add(x1, y1, wx1 * wy1);
    return acc;
}

// One sample point: the update of compute_unbounded_tsdf (mesh_utils.py:208-246).  Returns true when the voxel changed.
__device__ __forceinline__ bool tsdf_point(const float* F, float x, float y, float z3, int W, int H, const float* __restrict__ depth,
                                           const float* __restrict__ rgb, const float4* __restrict__ rgbd, float tr, float& tsdf,
                                           float& weight, float* col /*[3]*/)
{
    const float qx = x * F[0] + y * F[4] + z3 * F[8] + F[12];
    const float qy = x * F[1] + y * F[5] + z3 * F[9] + F[13];
    const float qw = x * F[3] + y * F[7] + z3 * F[11] + F[15];
    const float u = qx / qw, v = qy / qw;
    bool mask = (u > -1.f) && (u < 1.f) && (v > -1.f) && (v < 1.f) && (qw > 0);
    if (!mask) return false;                      // the reference samples depth for every point; the result is masked anyway
    float4 tex;
    if (rgbd) tex = bilinear_border4(rgbd, W, H, u, v);
    else tex.w = bilinear_border(depth, W, H, u, v);
    const float sdf = tex.w - qw;
    if (!(sdf > -tr)) return false;
    if (!rgbd) {
        const size_t HW = (size_t)W * H;
        tex.x = bilinear_border(rgb, W, H, u, v); tex.y = bilinear_border(rgb + HW, W, H, u, v); tex.z = bilinear_border(rgb + 2 * HW, W, H, u, v);
    }
    float s = sdf / tr;
    s = fminf(fmaxf(s, -1.0f), 1.0f);
    const float w = weight, wp = w + 1;
    tsdf = (tsdf * w + s) / wp;
    col[0] = (col[0] * w + tex.x) / wp; col[1] = (col[1] * w + tex.y) / wp; col[2] = (col[2] * w + tex.z) / wp;
    weight = wp;
    return true;
}

// Streaming RMW of (tsdf, weight, rgb): 12 B read per point + 40 B per touched voxel.  Each thread owns FOUR consecutive
// points so every stream is 16-byte vector traffic (3 x dwordx4 for the points, 1 each for tsdf/weight, 3 for rgb)
// instead of stride-3 dword accesses.
__global__ void __launch_bounds__(256) k_tsdf_integrate4(int64_t V4, const float4* __restrict__ points, const float* __restrict__ Fp, int W, int H,
                                                         const float* __restrict__ depth, const float* __restrict__ rgb, const float4* __restrict__ rgbd,
                                                         float sdf_trunc, const float4* __restrict__ trunc_pp, float4* __restrict__ tsdf,
                                                         float4* __restrict__ weight, float4* __restrict__ rgb_acc)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V4) return;
    float F[16];
#pragma unroll
    for (int k = 0; k < 16; k++) F[k] = Fp[k];
    const float4 p0 = points[3 * i], p1 = points[3 * i + 1], p2 = points[3 * i + 2];
    const float px[4] = { p0.x, p0.w, p1.z, p2.y }, py[4] = { p0.y, p1.x, p1.w, p2.z }, pz[4] = { p0.z, p1.y, p2.x, p2.w };
    float4 t4 = tsdf[i], w4 = weight[i];
    float t[4] = { t4.x, t4.y, t4.z, t4.w }, w[4] = { w4.x, w4.y, w4.z, w4.w };
    float tr[4] = { sdf_trunc, sdf_trunc, sdf_trunc, sdf_trunc };
    if (trunc_pp) { const float4 q = trunc_pp[i]; tr[0] = q.x; tr[1] = q.y; tr[2] = q.z; tr[3] = q.w; }
    float4 c0 = rgb_acc[3 * i], c1 = rgb_acc[3 * i + 1], c2 = rgb_acc[3 * i + 2];
    float col[4][3] = { { c0.x, c0.y, c0.z }, { c0.w, c1.x, c1.y }, { c1.z, c1.w, c2.x }, { c2.y, c2.z, c2.w } };
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; k++) any |= tsdf_point(F, px[k], py[k], pz[k], W, H, depth, rgb, rgbd, tr[k], t[k], w[k], col[k]);
    if (any) {
        tsdf[i] = make_float4(t[0], t[1], t[2], t[3]);
        weight[i] = make_float4(w[0], w[1], w[2], w[3]);
        rgb_acc[3 * i] = make_float4(col[0][0], col[0][1], col[0][2], col[1][0]);
        rgb_acc[3 * i + 1] = make_float4(col[1][1], col[1][2], col[2][0], col[2][1]);
        rgb_acc[3 * i + 2] = make_float4(col[2][2], col[3][0], col[3][1], col[3][2]);
    }
}

// scalar version for the tail / unaligned buffers
__global__ void __launch_bounds__(256) k_tsdf_integrate(int64_t V0, int64_t V, const float* __restrict__ points, const float* __restrict__ Fp, int W, int H,
                                                        const float* __restrict__ depth, const float* __restrict__ rgb, const float4* __restrict__ rgbd,
                                                        float sdf_trunc, const float* __restrict__ trunc_pp, float* __restrict__ tsdf,
                                                        float* __restrict__ weight, float* __restrict__ rgb_acc)
{
    const int64_t i = V0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    float F[16];
#pragma unroll
    for (int k = 0; k < 16; k++) F[k] = Fp[k];
    float t = tsdf[i], w = weight[i], col[3] = { rgb_acc[3 * i], rgb_acc[3 * i + 1], rgb_acc[3 * i + 2] };
    if (tsdf_point(F, points[3 * i], points[3 * i + 1], points[3 * i + 2], W, H, depth, rgb, rgbd, trunc_pp ? trunc_pp[i] : sdf_trunc, t, w, col)) {
        tsdf[i] = t; weight[i] = w;
        rgb_acc[3 * i] = col[0]; rgb_acc[3 * i + 1] = col[1]; rgb_acc[3 * i + 2] = col[2];
    }
}

extern "C" int gsr_tsdf_integrate(int64_t V, const float* points, const float* full_proj, int32_t W, int32_t H, const float* depth,
                                  const float* rgb, float sdf_trunc, const float* sdf_trunc_per_point, float* tsdf, float* weight,
                                  float* rgb_acc, void* rgbd_scratch, void* stream)
{
    if (V <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    // optional [H*W] float4 scratch: interleave (r,g,b,d) once per frame so every bilinear corner is one 16-byte load
    const float4* rgbd = nullptr;
    if (rgbd_scratch && (((uintptr_t)rgbd_scratch) & 15) == 0) {
        const int64_t n = (int64_t)W * H;
        hipLaunchKernelGGL(k_pack_rgbd, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, n, depth, rgb, (float4*)rgbd_scratch);
        rgbd = (const float4*)rgbd_scratch;
    }
    const uintptr_t al = (uintptr_t)points | (uintptr_t)tsdf | (uintptr_t)weight | (uintptr_t)rgb_acc | (uintptr_t)sdf_trunc_per_point;
    const int64_t V4 = (al & 15) == 0 ? V / 4 : 0;      // vector path needs 16-byte aligned buffers (torch allocations are)
    if ((V4 + 255) / 256 > 0x7FFFFFFF || (V + 255) / 256 > 0x7FFFFFFF) { gsr_set_error("tsdf: too many points"); return 1; }
    if (V4 > 0)
        hipLaunchKernelGGL(k_tsdf_integrate4, dim3((uint32_t)((V4 + 255) / 256)), dim3(256), 0, s, V4, (const float4*)points, full_proj, W, H,
                           depth, rgb, rgbd, sdf_trunc, (const float4*)sdf_trunc_per_point, (float4*)tsdf, (float4*)weight, (float4*)rgb_acc);
    const int64_t rest = V - 4 * V4;
    if (rest > 0)
        hipLaunchKernelGGL(k_tsdf_integrate, dim3((uint32_t)((rest + 255) / 256)), dim3(256), 0, s, 4 * V4, V, points, full_proj, W, H, depth,
                           rgb, rgbd, sdf_trunc, sdf_trunc_per_point, tsdf, weight, rgb_acc);
    return gsr_check_launch("tsdf_integrate", s, false);
}

// ---- dense-grid Open3D-style integration (see include/gsrast.h; parity unpinned).  One thread per voxel, z fastest:
// consecutive lanes walk a voxel column, so tsdf/weight/color accesses are coalesced 4/4/12-B streams.
struct DenseTsdfParams {
    int nx, ny, nz, W, H;
    float ox, oy, oz, vl, trunc, dtrunc, fx, fy, cx, cy, rfx, rfy, rtrunc;
    float E[12];
};
__global__ void __launch_bounds__(256) k_tsdf_dense(DenseTsdfParams p, const float* __restrict__ depth, const float* __restrict__ rgb,
                                                    float* __restrict__ tsdf, float* __restrict__ weight, float* __restrict__ color)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t V = (int64_t)p.nx * p.ny * p.nz;
    if (i >= V) return;
    const int iz = (int)(i % p.nz), iy = (int)((i / p.nz) % p.ny), ix = (int)(i / ((int64_t)p.nz * p.ny));
    const float x = p.ox + p.vl * ((float)ix + 0.5f), y = p.oy + p.vl * ((float)iy + 0.5f), z = p.oz + p.vl * ((float)iz + 0.5f);
    const float xc = p.E[0] * x + p.E[1] * y + p.E[2] * z + p.E[3];
    const float yc = p.E[4] * x + p.E[5] * y + p.E[6] * z + p.E[7];
    const float zc = p.E[8] * x + p.E[9] * y + p.E[10] * z + p.E[11];
    if (!(zc > 0.f)) return;
    // v_rcp/v_sqrt instead of the IEEE division/sqrt sequences: this kernel was VALU-bound on them (222 VALU per wave),
    // and its definition is parity-unpinned anyway (checked against the CPU restatement to 1e-4)
    const float rz = __builtin_amdgcn_rcpf(zc);
    const float uf = xc * p.fx * rz + p.cx + 0.5f, vf = yc * p.fy * rz + p.cy + 0.5f;
    if (!(uf >= 0.f && uf < (float)p.W && vf >= 0.f && vf < (float)p.H)) return;
    const int u = (int)uf, v = (int)vf;
    const float d = depth[(size_t)v * p.W + u];
    if (!(d > 0.f) || d > p.dtrunc) return;
    const float rx = ((float)u - p.cx) * p.rfx, ry = ((float)v - p.cy) * p.rfy;
    const float sdf = (d - zc) * __builtin_amdgcn_sqrtf(rx * rx + ry * ry + 1.0f);
    if (!(sdf > -p.trunc)) return;
    const float t = fminf(1.0f, sdf * p.rtrunc);
    const float w = weight[i], wp = w + 1.0f, rwp = __builtin_amdgcn_rcpf(wp);
    tsdf[i] = (tsdf[i] * w + t) * rwp;
    const size_t HW = (size_t)p.W * p.H;
#pragma unroll
    for (int c = 0; c < 3; c++) color[3 * i + c] = (color[3 * i + c] * w + rgb[c * HW + (size_t)v * p.W + u]) * rwp;
    weight[i] = wp;
}

extern "C" int gsr_tsdf_integrate_dense(int32_t nx, int32_t ny, int32_t nz, const float* origin, float voxel_length, float sdf_trunc,
                                        float depth_trunc, int32_t W, int32_t H, const float* depth, const float* rgb, float fx, float fy,
                                        float cx, float cy, const float* extrinsic, float* tsdf, float* weight, float* color, void* stream)
{
    const int64_t V = (int64_t)nx * ny * nz;
    if (V <= 0) return 0;
    if ((V + 255) / 256 > 0x7FFFFFFF) { gsr_set_error("tsdf: grid too large"); return 1; }
    DenseTsdfParams p;
    p.nx = nx; p.ny = ny; p.nz = nz; p.W = W; p.H = H;
    p.ox = origin[0]; p.oy = origin[1]; p.oz = origin[2]; p.vl = voxel_length; p.trunc = sdf_trunc; p.dtrunc = depth_trunc;
    p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy; p.rfx = 1.0f / fx; p.rfy = 1.0f / fy; p.rtrunc = 1.0f / sdf_trunc;
    for (int k = 0; k < 12; k++) p.E[k] = extrinsic[k];
    hipLaunchKernelGGL(k_tsdf_dense, dim3((uint32_t)((V + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, depth, rgb, tsdf, weight, color);
    return gsr_check_launch("tsdf_integrate_dense", (hipStream_t)stream, false);
}

// ---- fused L1 + linear-aux loss (forward value + dL/dcolor in one streaming pass; float4 loads, grid-stride)
__global__ void __launch_bounds__(256) k_loss_l1_linear(int64_t n4c, int64_t nc, const float* __restrict__ color, const float* __restrict__ gt,
                                                        float* __restrict__ dcol, int64_t n4a, int64_t na, const float* __restrict__ aux,
                                                        const float* __restrict__ waux, float inv_n, float* __restrict__ loss_out)
{
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll 2
    for (int64_t i = t0; i < n4c; i += stride) {
        const float4 c = reinterpret_cast<const float4*>(color)[i], g = reinterpret_cast<const float4*>(gt)[i];
        const float d0 = c.x - g.x, d1 = c.y - g.y, d2 = c.z - g.z, d3 = c.w - g.w;
        acc += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
        auto sg = [inv_n](float d) { return d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f); };
        reinterpret_cast<float4*>(dcol)[i] = make_float4(sg(d0), sg(d1), sg(d2), sg(d3));
    }
    for (int64_t i = 4 * n4c + t0; i < nc; i += stride) {          // tail
        const float d = color[i] - gt[i];
        acc += fabsf(d);
        dcol[i] = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
    }
    acc *= inv_n;
#pragma unroll 4
    for (int64_t i = t0; i < n4a; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(aux)[i], w = reinterpret_cast<const float4*>(waux)[i];
        acc += (a.x * w.x + a.y * w.y) + (a.z * w.z + a.w * w.w);
    }
    for (int64_t i = 4 * n4a + t0; i < na; i += stride) acc += aux[i] * waux[i];
    // block reduction, one atomic per block
    __shared__ float red[4];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(loss_out, (red[0] + red[1]) + (red[2] + red[3]));
}

extern "C" int gsr_loss_l1_linear(int64_t n_color, const float* color, const float* gt, float* dL_dcolor, int64_t n_aux, const float* aux,
                                  const float* waux, float* loss_out, void* stream)
{
    if (n_color <= 0) return 0;
    const bool al = (((uintptr_t)color | (uintptr_t)gt | (uintptr_t)dL_dcolor | (uintptr_t)aux | (uintptr_t)waux) & 15) == 0;
    const int64_t n4c = al ? n_color / 4 : 0, n4a = (al && n_aux > 0) ? n_aux / 4 : 0;
    hipLaunchKernelGGL(k_loss_l1_linear, dim3(2048), dim3(256), 0, (hipStream_t)stream, n4c, n_color, color, gt, dL_dcolor, n4a,
                       n_aux > 0 ? n_aux : 0, aux, waux, 1.0f / (float)n_color, loss_out);
    return gsr_check_launch("loss_l1_linear", (hipStream_t)stream, false);
}

// ---- fused Adam step (the optimizer of every Gaussian model: `torch.optim.Adam(l, lr=0.0, eps=1e-15)`, gssr/gaussian/vanilla_gaussian.py:120-139,
// stepped once per iteration by gssr/engine/trainer.py:127).  One streaming pass over (param, grad, exp_avg, exp_avg_sq) with the arithmetic of
// torch's single-tensor implementation:  m += (g - m)(1 - b1);  v = v b2 + g g (1 - b2);  p -= step_size * m / (sqrt(v) / sqrt(1 - b2^t) + eps),
// step_size = lr / (1 - b1^t) (both bias corrections formed on the host in double, as torch does).  28 bytes per parameter: HBM-bound.
__global__ void __launch_bounds__(256) k_adam(int64_t n4, int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, float step_size, float w1, float b2, float w2, float inv_bc2_sqrt, float eps,
                                              const float* __restrict__ lr_scale)
{
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    auto upd = [&](float& pp, float gg, float& mm, float& vv, float sc) {
        mm = mm + (gg - mm) * w1;
        vv = vv * b2 + gg * gg * w2;
        const float denom = sqrtf(vv) * inv_bc2_sqrt + eps;
        pp = pp - (step_size * sc) * (mm / denom);
    };
    for (int64_t i = t0; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
        if (lr_scale) sc = reinterpret_cast<const float4*>(lr_scale)[i];
        upd(pp.x, gg.x, mm.x, vv.x, sc.x); upd(pp.y, gg.y, mm.y, vv.y, sc.y); upd(pp.z, gg.z, mm.z, vv.z, sc.z); upd(pp.w, gg.w, mm.w, vv.w, sc.w);
        reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (int64_t i = 4 * n4 + t0; i < n; i += stride) {
        float pp = p[i], mm = m[i], vv = v[i];
        upd(pp, g[i], mm, vv, lr_scale ? lr_scale[i] : 1.0f);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

extern "C" int gsr_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float step_size, double beta1,
                             double beta2, float bias_correction2_sqrt, float eps, const float* lr_scale, void* stream)
{
    if (n <= 0) return 0;
    if (!param || !grad || !exp_avg || !exp_avg_sq) { gsr_set_error("adam_step: null pointer"); return 1; }
    if (!(bias_correction2_sqrt > 0.0f)) { gsr_set_error("adam_step: bias_correction2_sqrt must be > 0 (step >= 1)"); return 1; }
    const bool al = (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)lr_scale) & 15) == 0;
    const int64_t n4 = al ? n / 4 : 0;
    const int64_t work = n4 > 0 ? n4 : n;
    const int blocks = (int)std::min<int64_t>((work + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n4, n, param, grad, exp_avg, exp_avg_sq, step_size, (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), 1.0f / bias_correction2_sqrt, eps, lr_scale);      // 1 - beta in double first, as torch forms its lerp weights
    return gsr_check_launch("adam_step", (hipStream_t)stream, false);
}

// Several parameter tensors in ONE launch (a model has 6-20 of them, most of them tiny: a launch per tensor would cost more than the update).
// The table travels as a kernel argument; block b works on 4096 consecutive elements of the tensor whose [first_block, next first_block) holds b.
#define GSR_ADAM_MAX 24
#define GSR_ADAM_CHUNK 4096
struct AdamEntry { float* p; const float* g; const float* g2; float* m; float* v; const float* sc; int64_t n; float step_size, w1, b2, w2, inv_bc2_sqrt, eps; uint32_t first_block, vec; };
struct AdamTable { int32_t count; int32_t first; const float* hyper; AdamEntry e[GSR_ADAM_MAX]; };      // hyper: device [total][2] or null; first: caller index of e[0]
struct AdamIdx { int32_t idx[GSR_ADAM_MAX]; };

__global__ void __launch_bounds__(256) k_adam_multi(AdamTable T, AdamIdx X)
{
    int k = 0;
#pragma unroll 1
    for (int i = 1; i < T.count; i++) k = (blockIdx.x >= T.e[i].first_block) ? i : k;      // wave-uniform: scalar loop over <= 24 entries
    const AdamEntry& E = T.e[k];
    const int64_t base = (int64_t)(blockIdx.x - E.first_block) * GSR_ADAM_CHUNK;
    const int64_t end = min(E.n, base + GSR_ADAM_CHUNK);
    float step_size = E.step_size, ibc = E.inv_bc2_sqrt;
    const float w1 = E.w1, b2 = E.b2, w2 = E.w2, eps = E.eps;
    if (T.hyper) { step_size = T.hyper[2 * X.idx[k]]; ibc = 1.0f / T.hyper[2 * X.idx[k] + 1]; }      // per-step scalars from device memory (graph replay)
    auto upd = [&](float& pp, float gg, float& mm, float& vv, float sc) {
        mm = mm + (gg - mm) * w1;
        vv = vv * b2 + gg * gg * w2;
        const float denom = sqrtf(vv) * ibc + eps;
        pp = pp - (step_size * sc) * (mm / denom);
    };
    if (E.vec) {                                   // all five pointers 16-byte aligned: base is a multiple of 4096 elements
        for (int64_t i = base + 4 * (int64_t)threadIdx.x; i + 3 < end; i += 1024) {
            float4 pp = *reinterpret_cast<float4*>(E.p + i), mm = *reinterpret_cast<float4*>(E.m + i), vv = *reinterpret_cast<float4*>(E.v + i);
            float4 gg = *reinterpret_cast<const float4*>(E.g + i);
            if (E.g2) { const float4 g2 = *reinterpret_cast<const float4*>(E.g2 + i); gg.x += g2.x; gg.y += g2.y; gg.z += g2.z; gg.w += g2.w; }
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
            if (E.sc) sc = *reinterpret_cast<const float4*>(E.sc + i);
            upd(pp.x, gg.x, mm.x, vv.x, sc.x); upd(pp.y, gg.y, mm.y, vv.y, sc.y); upd(pp.z, gg.z, mm.z, vv.z, sc.z); upd(pp.w, gg.w, mm.w, vv.w, sc.w);
            *reinterpret_cast<float4*>(E.p + i) = pp; *reinterpret_cast<float4*>(E.m + i) = mm; *reinterpret_cast<float4*>(E.v + i) = vv;
        }
        const int64_t tail = base + ((end - base) & ~(int64_t)3);
        for (int64_t i = tail + threadIdx.x; i < end; i += 256) {
            float pp = E.p[i], mm = E.m[i], vv = E.v[i];
            upd(pp, E.g2 ? E.g[i] + E.g2[i] : E.g[i], mm, vv, E.sc ? E.sc[i] : 1.0f);
            E.p[i] = pp; E.m[i] = mm; E.v[i] = vv;
        }
    } else {
        for (int64_t i = base + threadIdx.x; i < end; i += 256) {
            float pp = E.p[i], mm = E.m[i], vv = E.v[i];
            upd(pp, E.g2 ? E.g[i] + E.g2[i] : E.g[i], mm, vv, E.sc ? E.sc[i] : 1.0f);
            E.p[i] = pp; E.m[i] = mm; E.v[i] = vv;
        }
    }
}

static int adam_multi(int32_t count, const gsr_adam_tensor* t, const float* hyper_dev, void* stream);
extern "C" int gsr_adam_step_multi(int32_t count, const gsr_adam_tensor* t, void* stream) { return adam_multi(count, t, nullptr, stream); }
extern "C" int gsr_adam_step_multi_dev(int32_t count, const gsr_adam_tensor* t, const float* hyper_dev, void* stream)
{
    if (!hyper_dev) { gsr_set_error("adam_step_multi_dev: hyper_dev is NULL"); return 1; }
    return adam_multi(count, t, hyper_dev, stream);
}
static int adam_multi(int32_t count, const gsr_adam_tensor* t, const float* hyper_dev, void* stream)
{
    if (count < 0 || (count > 0 && !t)) { gsr_set_error("adam_step_multi: bad table"); return 1; }
    int32_t i = 0;                              // consumed index, carried across launches: empty tensors are skipped without using a table slot,
    while (i < count) {                         // so a batch may consume more than GSR_ADAM_MAX indices and the next one must start behind them
        AdamTable T; T.count = 0; T.first = i; T.hyper = hyper_dev;
        AdamIdx X;
        uint32_t blocks = 0;
        for (; i < count && T.count < GSR_ADAM_MAX; i++) {
            const gsr_adam_tensor& a = t[i];
            if (a.n <= 0) continue;
            if (!a.param || !a.grad || !a.exp_avg || !a.exp_avg_sq || (!hyper_dev && !(a.bias_correction2_sqrt > 0.0f))) {
                gsr_set_error("adam_step_multi: tensor %d: null pointer or bias_correction2_sqrt <= 0", i); return 1;
            }
            X.idx[T.count] = i;
            AdamEntry& E = T.e[T.count++];
            E.p = a.param; E.g = a.grad; E.g2 = a.grad2; E.m = a.exp_avg; E.v = a.exp_avg_sq; E.sc = a.lr_scale; E.n = a.n;
            E.step_size = a.step_size; E.w1 = (float)(1.0 - a.beta1); E.b2 = (float)a.beta2; E.w2 = (float)(1.0 - a.beta2);
            E.inv_bc2_sqrt = 1.0f / a.bias_correction2_sqrt; E.eps = a.eps;
            E.first_block = blocks;
            E.vec = ((((uintptr_t)a.param | (uintptr_t)a.grad | (uintptr_t)a.grad2 | (uintptr_t)a.exp_avg | (uintptr_t)a.exp_avg_sq | (uintptr_t)a.lr_scale) & 15) == 0) ? 1u : 0u;
            blocks += (uint32_t)((a.n + GSR_ADAM_CHUNK - 1) / GSR_ADAM_CHUNK);
        }
        if (blocks) hipLaunchKernelGGL(k_adam_multi, dim3(blocks), dim3(256), 0, (hipStream_t)stream, T, X);
    }
    return gsr_check_launch("adam_step_multi", (hipStream_t)stream, false);
}

// ---- distCUDA2 (simple-knn/simple_knn.cu:186-222): mean squared distance to the 3 nearest neighbours.
// Same strategy as the reference (Morton order -> boxes of consecutive points -> box-pruned exact search), built from
// this library's own primitives: float min/max by order-preserving integer atomics, the stable LSD radix sort of
// gsr_binning.hip on the 30-bit Morton code, 256-point boxes.  The pruning is conservative, so the result is the exact
// 3-NN (bit-identical to an all-pairs search).
#define GSR_KNN_BOX 256
__device__ __forceinline__ uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

__global__ void k_knn_init(uint32_t* mm) { if (threadIdx.x < 3) mm[threadIdx.x] = 0xFFFFFFFFu; else if (threadIdx.x < 6) mm[threadIdx.x] = 0u; }

__global__ void __launch_bounds__(256) k_knn_minmax(int P, const float* __restrict__ pts, uint32_t* __restrict__ mm)
{
    __shared__ uint32_t s[6];
    if (threadIdx.x < 3) s[threadIdx.x] = 0xFFFFFFFFu; else if (threadIdx.x < 6) s[threadIdx.x] = 0u;
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x)
        for (int c = 0; c < 3; c++) { uint32_t o = f2ord(pts[3 * i + c]); atomicMin(&s[c], o); atomicMax(&s[3 + c], o); }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&mm[threadIdx.x], s[threadIdx.x]); else if (threadIdx.x < 6) atomicMax(&mm[threadIdx.x], s[threadIdx.x]);
}

__device__ __forceinline__ uint32_t prep_morton(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FF; x = (x | (x << 8)) & 0x0300F00F; x = (x | (x << 4)) & 0x030C30C3; x = (x | (x << 2)) & 0x09249249;
    return x;
}
// simple_knn.cu:55-61
__global__ void __launch_bounds__(256) k_knn_morton(int P, const float* __restrict__ pts, const uint32_t* __restrict__ mm, uint32_t* __restrict__ codes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    uint32_t m[3];
    for (int c = 0; c < 3; c++) {
        const float lo = ord2f(mm[c]), hi = ord2f(mm[3 + c]);
        const float t = (hi > lo) ? (pts[3 * i + c] - lo) / (hi - lo) : 0.f;
        m[c] = prep_morton((uint32_t)(t * ((1 << 10) - 1)));
    }
    codes[i] = m[0] | (m[1] << 1) | (m[2] << 2);
}

// per-box bounds of GSR_KNN_BOX consecutive points in Morton order (simple_knn.cu:79-118)
__global__ void __launch_bounds__(GSR_KNN_BOX) k_knn_boxes(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order, float* __restrict__ boxes)
{
    __shared__ uint32_t s[6];
    if (threadIdx.x < 3) s[threadIdx.x] = 0xFFFFFFFFu; else if (threadIdx.x < 6) s[threadIdx.x] = 0u;
    __syncthreads();
    const int i = blockIdx.x * GSR_KNN_BOX + threadIdx.x;
    if (i < P) {
        const uint32_t id = order[i];
        for (int c = 0; c < 3; c++) { uint32_t o = f2ord(pts[3 * id + c]); atomicMin(&s[c], o); atomicMax(&s[3 + c], o); }
    }
    __syncthreads();
    if (threadIdx.x < 6) boxes[6 * blockIdx.x + threadIdx.x] = ord2f(s[threadIdx.x]);
}

__device__ __forceinline__ void knn_update(float px, float py, float pz, float qx, float qy, float qz, float& b0, float& b1, float& b2)
{
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    float d = dx * dx + dy * dy + dz * dz;
    if (b0 > d) { const float t = b0; b0 = d; d = t; }
    if (b1 > d) { const float t = b1; b1 = d; d = t; }
    if (b2 > d) { b2 = d; }
}

// simple_knn.cu:148-184
__global__ void __launch_bounds__(256) k_knn_dist(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
                                                  const float* __restrict__ boxes, int nbox, float* __restrict__ out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const uint32_t me = order[idx];
    const float px = pts[3 * me], py = pts[3 * me + 1], pz = pts[3 * me + 2];
    const float FMAX = 3.402823466e+38f;
    float b0 = FMAX, b1 = FMAX, b2 = FMAX;
    for (int i = max(0, idx - 3); i <= min(P - 1, idx + 3); i++) {
        if (i == idx) continue;
        const uint32_t q = order[i];
        knn_update(px, py, pz, pts[3 * q], pts[3 * q + 1], pts[3 * q + 2], b0, b1, b2);
    }
    const float reject = b2;
    b0 = FMAX; b1 = FMAX; b2 = FMAX;
    for (int b = 0; b < nbox; b++) {
        const float* bx = boxes + 6 * b;
        float ddx = 0.f, ddy = 0.f, ddz = 0.f;
        if (px < bx[0] || px > bx[3]) ddx = fminf(fabsf(px - bx[0]), fabsf(px - bx[3]));
        if (py < bx[1] || py > bx[4]) ddy = fminf(fabsf(py - bx[1]), fabsf(py - bx[4]));
        if (pz < bx[2] || pz > bx[5]) ddz = fminf(fabsf(pz - bx[2]), fabsf(pz - bx[5]));
        const float dist = ddx * ddx + ddy * ddy + ddz * ddz;
        if (dist > reject || dist > b2) continue;
        const int e = min(P, (b + 1) * GSR_KNN_BOX);
        for (int i = b * GSR_KNN_BOX; i < e; i++) {
            if (i == idx) continue;
            const uint32_t q = order[i];
            knn_update(px, py, pz, pts[3 * q], pts[3 * q + 1], pts[3 * q + 2], b0, b1, b2);
        }
    }
    out[me] = (b0 + b1 + b2) / 3.0f;
}

struct KnnScratch { uint32_t *mm, *keys_a, *keys_b, *vals_a, *vals_b, *hist; float* boxes; size_t bytes; };
static KnnScratch knn_carve(int P, void* base)
{
    KnnScratch k; char* p = (char*)base;
    const size_t n = (size_t)(P > 0 ? P : 1);
    const uint32_t nblk = gsr_div_up((uint32_t)n, GSR_SORT_BLOCK), nbox = gsr_div_up((uint32_t)n, GSR_KNN_BOX);
    auto take = [&](size_t bytes) { char* r = p; p += gsr_align(bytes); return r; };
    k.mm = (uint32_t*)take(64); k.keys_a = (uint32_t*)take(n * 4); k.keys_b = (uint32_t*)take(n * 4);
    k.vals_a = (uint32_t*)take(n * 4); k.vals_b = (uint32_t*)take(n * 4);
    k.hist = (uint32_t*)take(gsr_sort_hist_words(nblk, 256) * 4); k.boxes = (float*)take((size_t)nbox * 6 * 4);
    k.bytes = (size_t)(p - (char*)base);
    return k;
}
extern "C" size_t gsr_dist2_scratch_bytes(int32_t P) { return knn_carve(P, nullptr).bytes; }
extern "C" int gsr_dist2(int32_t P, const float* points, float* out, void* scratch, size_t scratch_bytes, void* stream)
{
    if (P <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    KnnScratch k = knn_carve(P, scratch);
    if (k.bytes > scratch_bytes) { gsr_set_error("dist2 scratch too small: %zu < %zu", scratch_bytes, k.bytes); return 1; }
    const int nbox = (int)gsr_div_up((uint32_t)P, GSR_KNN_BOX);
    hipLaunchKernelGGL(k_knn_init, dim3(1), dim3(64), 0, s, k.mm);
    hipLaunchKernelGGL(k_knn_minmax, dim3(min(1024u, gsr_div_up((uint32_t)P, 256))), dim3(256), 0, s, P, points, k.mm);
    hipLaunchKernelGGL(k_knn_morton, dim3(gsr_div_up((uint32_t)P, 256)), dim3(256), 0, s, P, points, k.mm, k.keys_a);
    bool in_b = false;
    if (gsr_radix_sort_pairs(k.keys_a, k.vals_a, k.keys_b, k.vals_b, (uint32_t)P, nullptr, 0, 30, 8, true, k.hist, &in_b, s)) return 1;
    const uint32_t* order = in_b ? k.vals_b : k.vals_a;
    hipLaunchKernelGGL(k_knn_boxes, dim3(nbox), dim3(GSR_KNN_BOX), 0, s, P, points, order, k.boxes);
    hipLaunchKernelGGL(k_knn_dist, dim3(gsr_div_up((uint32_t)P, 256)), dim3(256), 0, s, P, points, order, k.boxes, nbox, out);
    return gsr_check_launch("dist2", s, false);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Per-Gaussian `all_map` input of the plane rasterizer (gssr/scene/pgsr_scene.py:241-257 get_smallest_axis / get_normal, :297-304):
//   R = quaternion_to_matrix(q) (pytorch3d: real part first, two_s = 2/(q.q));  n = R[:, argmin(scale)], flipped towards the camera;
//   local_normal = n Wv[:3,:3];  local_distance = |local_normal . (xyz Wv[:3,:3] + Wv[3,:3])|;  all_map = {local_normal, 1, local_distance}.
// The reference spends ~25 torch ops forward and ~40 backward on it every iteration; here one streaming kernel each way (HBM-bound:
// 40 B read + 20 B written per Gaussian forward, 60 B + 28 B backward).
struct PlaneAxis { float n[3], ln[3], pc[3], flip; int k; };

__device__ __forceinline__ void plane_q2m(const float4 q, float* R)
{
    const float r = q.x, i = q.y, j = q.z, k = q.w;
    const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
    R[0] = 1 - two_s * (j * j + k * k); R[1] = two_s * (i * j - k * r); R[2] = two_s * (i * k + j * r);
    R[3] = two_s * (i * j + k * r); R[4] = 1 - two_s * (i * i + k * k); R[5] = two_s * (j * k - i * r);
    R[6] = two_s * (i * k - j * r); R[7] = two_s * (j * k + i * r); R[8] = 1 - two_s * (i * i + j * j);
}

__device__ __forceinline__ PlaneAxis plane_axis(int p, const float* __restrict__ xyz, const float4 q, const float* __restrict__ scale, int ss,
                                                const float* __restrict__ V, const float* __restrict__ cp)
{
    PlaneAxis a;
    float R[9]; plane_q2m(q, R);
    const float s0 = scale[(size_t)p * ss], s1 = scale[(size_t)p * ss + 1], s2 = scale[(size_t)p * ss + 2];
    int k = 0; float sm = s0;
    if (s1 < sm) { k = 1; sm = s1; }
    if (s2 < sm) k = 2;                                               // first minimum wins, like torch.min
    a.k = k;
    a.n[0] = k == 0 ? R[0] : (k == 1 ? R[1] : R[2]); a.n[1] = k == 0 ? R[3] : (k == 1 ? R[4] : R[5]); a.n[2] = k == 0 ? R[6] : (k == 1 ? R[7] : R[8]);
    const float x0 = xyz[3 * (size_t)p], x1 = xyz[3 * (size_t)p + 1], x2 = xyz[3 * (size_t)p + 2];
    const float dot = a.n[0] * (cp[0] - x0) + a.n[1] * (cp[1] - x1) + a.n[2] * (cp[2] - x2);
    a.flip = dot < 0.f ? -1.f : 1.f;
    if (dot < 0.f) { a.n[0] = -a.n[0]; a.n[1] = -a.n[1]; a.n[2] = -a.n[2]; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a.ln[c] = a.n[0] * V[c] + a.n[1] * V[4 + c] + a.n[2] * V[8 + c];
        a.pc[c] = x0 * V[c] + x1 * V[4 + c] + x2 * V[8 + c] + V[12 + c];
    }
    return a;
}

__global__ void __launch_bounds__(256) k_plane_allmap(int P, const float* __restrict__ xyz, const float4* __restrict__ rot, const float* __restrict__ scale,
                                                      int ss, const float* __restrict__ V, const float* __restrict__ cp, float* __restrict__ out)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const PlaneAxis a = plane_axis(p, xyz, rot[p], scale, ss, V, cp);
    float* o = out + 5 * (size_t)p;
    o[0] = a.ln[0]; o[1] = a.ln[1]; o[2] = a.ln[2]; o[3] = 1.0f;
    o[4] = fabsf(a.ln[0] * a.pc[0] + a.ln[1] * a.pc[1] + a.ln[2] * a.pc[2]);
}

__global__ void __launch_bounds__(256) k_plane_allmap_bwd(int P, const float* __restrict__ xyz, const float4* __restrict__ rot,
                                                          const float* __restrict__ scale, int ss, const float* __restrict__ V,
                                                          const float* __restrict__ cp, const float* __restrict__ g_all, float* __restrict__ d_xyz,
                                                          float4* __restrict__ d_rot)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float4 q = rot[p];
    const PlaneAxis a = plane_axis(p, xyz, q, scale, ss, V, cp);
    const float* g = g_all + 5 * (size_t)p;
    const float g4 = g[4];
    const float sd = a.ln[0] * a.pc[0] + a.ln[1] * a.pc[1] + a.ln[2] * a.pc[2];
    const float sg = sd > 0.f ? 1.f : (sd < 0.f ? -1.f : 0.f);
    float dln[3], dpc[3], dn[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { dln[c] = g[c] + sg * g4 * a.pc[c]; dpc[c] = sg * g4 * a.ln[c]; }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        d_xyz[3 * (size_t)p + r] = dpc[0] * V[r * 4] + dpc[1] * V[r * 4 + 1] + dpc[2] * V[r * 4 + 2];
        dn[r] = a.flip * (dln[0] * V[r * 4] + dln[1] * V[r * 4 + 1] + dln[2] * V[r * 4 + 2]);
    }
    const float r = q.x, i = q.y, j = q.z, kk = q.w;
    const float s2 = r * r + i * i + j * j + kk * kk, two_s = 2.0f / s2;
    float dR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) { dR[c] = a.k == c ? dn[0] : 0.f; dR[3 + c] = a.k == c ? dn[1] : 0.f; dR[6 + c] = a.k == c ? dn[2] : 0.f; }
    const float M[9] = {-(j * j + kk * kk), i * j - kk * r, i * kk + j * r, i * j + kk * r, -(i * i + kk * kk), j * kk - i * r,
                        i * kk - j * r, j * kk + i * r, -(i * i + j * j)};
    float dts = 0.f, dM[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) { dts += dR[e] * M[e]; dM[e] = dR[e] * two_s; }
    float4 dq;
    dq.x = -kk * dM[1] + j * dM[2] + kk * dM[3] - i * dM[5] - j * dM[6] + i * dM[7];
    dq.y = j * dM[1] + kk * dM[2] + j * dM[3] - 2 * i * dM[4] - r * dM[5] + kk * dM[6] + r * dM[7] - 2 * i * dM[8];
    dq.z = -2 * j * dM[0] + i * dM[1] + r * dM[2] + i * dM[3] + kk * dM[5] - r * dM[6] + kk * dM[7] - 2 * j * dM[8];
    dq.w = -2 * kk * dM[0] - r * dM[1] + i * dM[2] + r * dM[3] - 2 * kk * dM[4] + j * dM[5] + i * dM[6] + j * dM[7];
    const float sc = dts * (-2.0f / (s2 * s2)) * 2.0f;
    d_rot[p] = make_float4(dq.x + sc * r, dq.y + sc * i, dq.z + sc * j, dq.w + sc * kk);
}

extern "C" int gsr_plane_allmap(int32_t P, const float* means3D, const float* rotations, const float* scales, int32_t scale_stride,
                                const float* viewmatrix, const float* campos, float* all_map, void* stream)
{
    if (P <= 0) return 0;
    if (!means3D || !rotations || !scales || !viewmatrix || !campos || !all_map || scale_stride < 3) {
        gsr_set_error("plane_allmap: null pointer or scale_stride < 3"); return 1;
    }
    if (((uintptr_t)rotations & 15) != 0) { gsr_set_error("plane_allmap: rotations must be 16-byte aligned"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_plane_allmap, dim3(gsr_div_up((uint32_t)P, 256)), dim3(256), 0, s, P, means3D, (const float4*)rotations, scales, scale_stride,
                       viewmatrix, campos, all_map);
    return gsr_check_launch("plane_allmap", s, false);
}

extern "C" int gsr_plane_allmap_backward(int32_t P, const float* means3D, const float* rotations, const float* scales, int32_t scale_stride,
                                         const float* viewmatrix, const float* campos, const float* dL_dall_map, float* dL_dmeans3D,
                                         float* dL_drotations, void* stream)
{
    if (P <= 0) return 0;
    if (!means3D || !rotations || !scales || !viewmatrix || !campos || !dL_dall_map || !dL_dmeans3D || !dL_drotations || scale_stride < 3) {
        gsr_set_error("plane_allmap_backward: null pointer or scale_stride < 3"); return 1;
    }
    if ((((uintptr_t)rotations | (uintptr_t)dL_drotations) & 15) != 0) { gsr_set_error("plane_allmap_backward: rotations must be 16-byte aligned"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_plane_allmap_bwd, dim3(gsr_div_up((uint32_t)P, 256)), dim3(256), 0, s, P, means3D, (const float4*)rotations, scales, scale_stride,
                       viewmatrix, campos, dL_dall_map, dL_dmeans3D, (float4*)dL_drotations);
    return gsr_check_launch("plane_allmap_backward", s, false);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Activations of the explicit-Gaussian models, the step right in front of the rasterizer call of vanilla-3dgs / 2dgs / pgsr
// (gssr/gaussian/vanilla_gaussian.py:86-90,250-269: get_scaling = exp(_scaling), get_rotation = F.normalize(_rotation), get_opacity =
// sigmoid(_opacity)): three torch forward ops and ~8 autograd kernels per iteration there, one streaming kernel each way here.
// normalize: x / max(|x|_2, 1e-12) (torch.nn.functional.normalize, dim 1); its backward (g - r <r, g>) / max(|x|, eps), zero coupling when clamped.
__global__ void __launch_bounds__(256) k_gauss_act_fwd(int P, int S, const float* __restrict__ scl_log, const float4* __restrict__ rot_raw,
                                                       const float* __restrict__ op_raw, float* __restrict__ scl, float4* __restrict__ rot, float* __restrict__ op)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    for (int k = 0; k < S; k++) scl[(size_t)p * S + k] = expf(scl_log[(size_t)p * S + k]);
    const float4 q = rot_raw[p];
    const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    rot[p] = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
    op[p] = 1.0f / (1.0f + expf(-op_raw[p]));
}
__global__ void __launch_bounds__(256) k_gauss_act_bwd(int P, int S, const float* __restrict__ scl, const float4* __restrict__ rot_raw, const float4* __restrict__ rot,
                                                       const float* __restrict__ op, const float* __restrict__ d_scl, const float4* __restrict__ d_rot,
                                                       const float* __restrict__ d_op, float* __restrict__ d_scl_log, float4* __restrict__ d_rot_raw,
                                                       float* __restrict__ d_op_raw)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    for (int k = 0; k < S; k++) d_scl_log[(size_t)p * S + k] = d_scl ? d_scl[(size_t)p * S + k] * scl[(size_t)p * S + k] : 0.0f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d_rot) {
        const float4 q = rot_raw[p], r = rot[p], g = d_rot[p];
        const float nn = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
        if (nn > 1e-12f) {
            const float dot = r.x * g.x + r.y * g.y + r.z * g.z + r.w * g.w;
            o = make_float4((g.x - r.x * dot) / nn, (g.y - r.y * dot) / nn, (g.z - r.z * dot) / nn, (g.w - r.w * dot) / nn);
        } else o = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);      // clamped denominator: y = x / eps
    }
    d_rot_raw[p] = o;
    const float a = op[p];
    d_op_raw[p] = d_op ? d_op[p] * a * (1.0f - a) : 0.0f;
}
extern "C" int gsr_gauss_activations(int32_t P, int32_t scale_dim, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                                     float* scaling, float* rotation, float* opacity, void* stream)
{
    if (P <= 0) return 0;
    if (!scaling_raw || !rotation_raw || !opacity_raw || !scaling || !rotation || !opacity || scale_dim < 1 || scale_dim > 3) {
        gsr_set_error("gauss_activations: null pointer or scale_dim outside 1..3"); return 1;
    }
    if ((((uintptr_t)rotation_raw | (uintptr_t)rotation) & 15) != 0) { gsr_set_error("gauss_activations: rotations must be 16-byte aligned"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_gauss_act_fwd, dim3(gsr_div_up((uint32_t)P, 256)), dim3(256), 0, s, P, scale_dim, scaling_raw, (const float4*)rotation_raw, opacity_raw,
                       scaling, (float4*)rotation, opacity);
    return gsr_check_launch("gauss_activations", s, false);
}
extern "C" int gsr_gauss_activations_backward(int32_t P, int32_t scale_dim, const float* scaling, const float* rotation_raw, const float* rotation,
                                              const float* opacity, const float* dL_dscaling, const float* dL_drotation, const float* dL_dopacity,
                                              float* dL_dscaling_raw, float* dL_drotation_raw, float* dL_dopacity_raw, void* stream)
{
    if (P <= 0) return 0;
    if (!scaling || !rotation_raw || !rotation || !opacity || !dL_dscaling_raw || !dL_drotation_raw || !dL_dopacity_raw || scale_dim < 1 || scale_dim > 3) {
        gsr_set_error("gauss_activations_backward: null pointer or scale_dim outside 1..3"); return 1;
    }
    if ((((uintptr_t)rotation_raw | (uintptr_t)rotation | (uintptr_t)dL_drotation | (uintptr_t)dL_drotation_raw) & 15) != 0) {
        gsr_set_error("gauss_activations_backward: rotations must be 16-byte aligned"); return 1;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_gauss_act_bwd, dim3(gsr_div_up((uint32_t)P, 256)), dim3(256), 0, s, P, scale_dim, scaling, (const float4*)rotation_raw, (const float4*)rotation,
                       opacity, dL_dscaling, (const float4*)dL_drotation, dL_dopacity, dL_dscaling_raw, (float4*)dL_drotation_raw, dL_dopacity_raw);
    return gsr_check_launch("gauss_activations_backward", s, false);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Per-iteration densification statistics of the explicit-Gaussian methods (vanilla_gaussian.py:467-472,428-430; pgsr_gaussian.py:164-172,
// 157-161).  The reference writes them as boolean-mask index assignments (a nonzero() host synchronisation each: 3 for 3DGS/2DGS, 5 for
// PGSR per iteration); one elementwise kernel here.
__global__ void __launch_bounds__(256) k_densify_stats(int P, const uint8_t* __restrict__ filter, const int32_t* __restrict__ radii,
                                                       const int32_t* __restrict__ observe, const float* __restrict__ grad, int gs,
                                                       const float* __restrict__ grad_abs, float* __restrict__ max_radii, float* __restrict__ accum,
                                                       float* __restrict__ denom, float* __restrict__ accum_abs, float* __restrict__ denom_abs)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P || !filter[p]) return;
    if (!observe || observe[p] > 0) { const float r = (float)radii[p]; if (r > max_radii[p]) max_radii[p] = r; }
    const float gx = grad[(size_t)p * gs], gy = grad[(size_t)p * gs + 1];
    accum[p] += sqrtf(gx * gx + gy * gy); denom[p] += 1.f;
    if (grad_abs) {
        const float ax = grad_abs[(size_t)p * gs], ay = grad_abs[(size_t)p * gs + 1];
        accum_abs[p] += sqrtf(ax * ax + ay * ay); denom_abs[p] += 1.f;
    }
}

extern "C" int gsr_densify_stats(int32_t P, const uint8_t* visibility_filter, const int32_t* radii, const int32_t* out_observe,
                                 const float* viewspace_grad, int32_t grad_stride, const float* viewspace_grad_abs, float* max_radii2D,
                                 float* xyz_gradient_accum, float* denom, float* xyz_gradient_accum_abs, float* denom_abs, void* stream)
{
    if (P <= 0) return 0;
    if (!visibility_filter || !radii || !viewspace_grad || !max_radii2D || !xyz_gradient_accum || !denom || grad_stride < 2 ||
        (viewspace_grad_abs && (!xyz_gradient_accum_abs || !denom_abs))) {
        gsr_set_error("densify_stats: null pointer or grad_stride < 2"); return 1;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_densify_stats, dim3(gsr_div_up((uint32_t)P, 256)), dim3(256), 0, s, P, visibility_filter, radii, out_observe, viewspace_grad,
                       grad_stride, viewspace_grad_abs, max_radii2D, xyz_gradient_accum, denom, xyz_gradient_accum_abs, denom_abs);
    return gsr_check_launch("densify_stats", s, false);
}
