// gsr_mvloss.hip -- PGSR multi-view regularisers (geometric consistency + patch NCC), gfx950 only.
//
//   gssr/scene/pgsr_scene.py:113-204 (multi-view branch of get_loss_dict), :60-95 (lncc)
//   gssr/utils/point_utils.py:38-75, gssr/utils/graphics_utils.py:185-198, gssr/cameras/__init__.py:96-121
//
// The reference evaluates this as ~120 torch ops over (H*W,3) and (N,49,2) temporaries with two grid_samples, two host
// synchronisations (d_mask.sum() > 0, np.random.choice on the CPU) and autograd replay.  Here:
//   k_mv_geo : one thread per pixel.  Reprojects the pixel into the neighbour view through its plane depth, samples the neighbour's
//              plane depth (bilinear, border clamp), reprojects back, writes noise / d_mask / weight and the UNSCALED analytic
//              gradients of sum(weight * noise) to both depth maps (4 float atomics per valid pixel into the neighbour map), and
//              block-reduces {sum, count}.
//   k_mv_ncc : 16 lanes per sampled pixel.  Builds the plane-induced homography from the rendered normal / distance, walks the
//              (2h+1)^2 patch twice (sums, then the chain rule back to the homography; 16-lane butterflies), writes ncc / mask and the
//              UNSCALED gradients of sum(ncc * weight) to normal and distance at that pixel (sampled at most once: plain stores).
//   k_mv_finish : {sum, count, sum/count (0 if count == 0)}.
// The means' 1/count and the lambdas are applied by the caller (gsrast.losses) as one device-scalar multiply in backward, so no
// host synchronisation is needed anywhere.  Both kernels are gather-latency bound (random bilinear taps), not HBM-bound.
#include "gsr_common.h"

__device__ __forceinline__ float3 mv_xform(const float* M, float3 p)
{
    return make_float3(p.x * M[0] + p.y * M[3] + p.z * M[6] + M[9], p.x * M[1] + p.y * M[4] + p.z * M[7] + M[10],
                       p.x * M[2] + p.y * M[5] + p.z * M[8] + M[11]);
}
__device__ __forceinline__ float3 mv_xform_t(const float* M, float3 d)      // d @ A^T
{
    return make_float3(d.x * M[0] + d.y * M[1] + d.z * M[2], d.x * M[3] + d.y * M[4] + d.z * M[5], d.x * M[6] + d.y * M[7] + d.z * M[8]);
}

__device__ __forceinline__ float2 mv_block_sum2(float a, float b, float* red)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[4 + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    return make_float2((red[0] + red[1]) + (red[2] + red[3]), (red[4] + red[5]) + (red[6] + red[7]));
}

__global__ void __launch_bounds__(256) k_mv_geo(gsr_mv_cfg c, const float* __restrict__ depth, const float* __restrict__ near_depth,
                                                float* __restrict__ noise_out, uint8_t* __restrict__ dmask_out, float* __restrict__ weight_out,
                                                float* __restrict__ g_depth, float* __restrict__ g_near, float2* __restrict__ partial)
{
    __shared__ float red[8];
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    float s_w = 0.f, s_c = 0.f;
    if (x < c.W && y < c.H) {
        const int p = y * c.W + x;
        const float d = depth[p];
        const float rx = ((float)x - c.cx) / c.fx, ry = ((float)y - c.cy) / c.fy;
        const float3 q = mv_xform(c.v2n, make_float3(rx * d, ry * d, d));
        const float u = q.x * c.nfx / q.z + c.ncx, v = q.y * c.nfy / q.z + c.ncy;
        const bool mask = u > 0.f && u < (float)c.Wn && v > 0.f && v < (float)c.Hn && q.z > 0.1f;
        // grid_sample(align_corners=True, padding_mode='border'): clamp the coordinate (zero coordinate-gradient where clamped)
        const float uc = fminf(fmaxf(u, 0.f), (float)(c.Wn - 1)), vc = fminf(fmaxf(v, 0.f), (float)(c.Hn - 1));
        const float mu = (u < 0.f || u > (float)(c.Wn - 1)) ? 0.f : 1.f, mv = (v < 0.f || v > (float)(c.Hn - 1)) ? 0.f : 1.f;
        const int x0 = (int)floorf(uc), y0 = (int)floorf(vc);
        const float ax = uc - (float)x0, ay = vc - (float)y0;
        const bool xr = x0 + 1 < c.Wn, yb = y0 + 1 < c.Hn;                       // ax (ay) is 0 whenever the right (bottom) tap is out of range
        const float wx0 = 1.f - ax, wy0 = 1.f - ay;
        float t00 = 0.f, t01 = 0.f, t10 = 0.f, t11 = 0.f;
        if (isfinite(u) && isfinite(v)) {
            t00 = near_depth[y0 * c.Wn + x0];
            if (xr) t01 = near_depth[y0 * c.Wn + x0 + 1];
            if (yb) t10 = near_depth[(y0 + 1) * c.Wn + x0];
            if (xr && yb) t11 = near_depth[(y0 + 1) * c.Wn + x0 + 1];
        }
        const float mz = t00 * wx0 * wy0 + t01 * ax * wy0 + t10 * wx0 * ay + t11 * ax * ay;
        const float dmz_du = (-t00 * wy0 + t01 * wy0 - t10 * ay + t11 * ay) * mu, dmz_dv = (-t00 * wx0 - t01 * ax + t10 * wx0 + t11 * ax) * mv;
        const float3 qp = make_float3(q.x / q.z * mz, q.y / q.z * mz, mz);
        const float3 r = mv_xform(c.n2v, qp);
        const float ex = r.x * c.fx / r.z + c.cx - (float)x, ey = r.y * c.fy / r.z + c.cy - (float)y;
        const float noise = sqrtf(ex * ex + ey * ey);
        const bool dm = mask && noise < c.noise_th;
        const float w = dm ? 1.0f / expf(noise) : 0.f;
        noise_out[p] = noise; dmask_out[p] = dm ? 1 : 0; weight_out[p] = w;
        float gd = 0.f;
        if (dm) {
            s_w = w * noise; s_c = 1.f;
            if (noise > 0.f) {
                const float dex = w * ex / noise, dey = w * ey / noise;
                const float3 dr = make_float3(dex * c.fx / r.z, dey * c.fy / r.z, -(dex * r.x * c.fx + dey * r.y * c.fy) / (r.z * r.z));
                const float3 dqp = mv_xform_t(c.n2v, dr);
                const float dmz = dqp.x * q.x / q.z + dqp.y * q.y / q.z + dqp.z;
                float3 dq = make_float3(dqp.x * mz / q.z, dqp.y * mz / q.z, -(dqp.x * q.x + dqp.y * q.y) * mz / (q.z * q.z));
                atomicAdd(&g_near[y0 * c.Wn + x0], dmz * (wx0 * wy0));
                if (xr) atomicAdd(&g_near[y0 * c.Wn + x0 + 1], dmz * (ax * wy0));
                if (yb) atomicAdd(&g_near[(y0 + 1) * c.Wn + x0], dmz * (wx0 * ay));
                if (xr && yb) atomicAdd(&g_near[(y0 + 1) * c.Wn + x0 + 1], dmz * (ax * ay));
                const float gu = dmz * dmz_du, gv = dmz * dmz_dv;
                dq.x += gu * c.nfx / q.z; dq.y += gv * c.nfy / q.z; dq.z += -(gu * q.x * c.nfx + gv * q.y * c.nfy) / (q.z * q.z);
                const float3 dpc = mv_xform_t(c.v2n, dq);
                gd = dpc.x * rx + dpc.y * ry + dpc.z;
            }
        }
        g_depth[p] = gd;
    }
    const float2 t = mv_block_sum2(s_w, s_c, red);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = t;
}

// grid_sample(align_corners=True, padding_mode='zeros') at pixel coordinates; optionally the coordinate gradient
template <bool GRAD>
__device__ __forceinline__ float mv_bilerp0(const float* __restrict__ img, int W, int H, float u, float v, float& du, float& dv)
{
    const float fu = floorf(u), fv = floorf(v);
    if (!(fu >= -1.f && fu < (float)W && fv >= -1.f && fv < (float)H)) { if (GRAD) { du = 0.f; dv = 0.f; } return 0.f; }   // also NaN/inf
    const int x0 = (int)fu, y0 = (int)fv;
    const float ax = u - fu, ay = v - fv;
    const bool xl = x0 >= 0, xr = x0 + 1 < W, yt = y0 >= 0, yb = y0 + 1 < H;
    const float t00 = (xl && yt) ? img[y0 * W + x0] : 0.f, t01 = (xr && yt) ? img[y0 * W + x0 + 1] : 0.f;
    const float t10 = (xl && yb) ? img[(y0 + 1) * W + x0] : 0.f, t11 = (xr && yb) ? img[(y0 + 1) * W + x0 + 1] : 0.f;
    const float wx0 = 1.f - ax, wy0 = 1.f - ay;
    if (GRAD) { du = -t00 * wy0 + t01 * wy0 - t10 * ay + t11 * ay; dv = -t00 * wx0 - t01 * ax + t10 * wx0 + t11 * ax; }
    return t00 * wx0 * wy0 + t01 * ax * wy0 + t10 * wx0 * ay + t11 * ax * ay;
}

// sum over the 16 lanes of a sample group (xor butterflies stay inside a 16-lane row)
__device__ __forceinline__ float mv_sum16(float v)
{
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}

// 16 lanes per sampled pixel (4 samples per wave, 16 per block): a thread-per-sample walk of 2 x 49 dependent gathers leaves only
// ~1.5 waves per SIMD at 102400 samples and runs at gather latency (measured 0.44 ms); spreading the taps over 16 lanes gives 16x the
// waves and 4 short iterations per pass.  With at most 4 taps per lane (patch <= 3) pass 1 keeps value and coordinate-gradient of its taps in
// registers; larger patches re-gather in pass 2.
template <bool CACHE>      // CACHE: (2h+1)^2 <= 64, i.e. at most 4 taps per lane -- pass 1 keeps its taps in registers and pass 2 does not gather again
__global__ void __launch_bounds__(256) k_mv_ncc(gsr_mv_cfg c, int N, const int32_t* __restrict__ idx, const float* __restrict__ weight,
                                                const float* __restrict__ normal, const float* __restrict__ dist, const float* __restrict__ gray,
                                                const float* __restrict__ near_gray, float* __restrict__ ncc_out, uint8_t* __restrict__ mask_out,
                                                float* __restrict__ g_normal, float* __restrict__ g_dist, float2* __restrict__ partial)
{
    __shared__ float red[8];
    const int sub = threadIdx.x & 15;
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
    float s_v = 0.f, s_c = 0.f;
    const int p = i < N ? idx[i] : -1;
    if (p >= 0) {
        const int HW = c.W * c.H, h = c.patch, side = 2 * h + 1, ntap = side * side;
        const int x = p % c.W, y = p / c.W;
        const float s = c.ncc_scale, tps = (float)ntap;
        const float n0 = normal[p], n1 = normal[HW + p], n2 = normal[2 * HW + p], dd = dist[p];
        const float* A = c.v2n; const float* b = c.v2n + 9;
        // Hm = A^T - b n^T / dist;  Hk = K_near(s) Hm Kinv_view(s)
        float Hm[9], T1[9], Hk[9];
        const float nn[3] = {n0, n1, n2};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int e = 0; e < 3; ++e) Hm[a * 3 + e] = A[e * 3 + a] - b[a] * nn[e] / dd;
        const float kfx = c.nfx / s, kfy = c.nfy / s, kcx = c.ncx / s, kcy = c.ncy / s;       // K_near(s)
        const float ifx = s / c.fx, ify = s / c.fy, icx = -c.cx / c.fx, icy = -c.cy / c.fy;    // Kinv_view(s)
#pragma unroll
        for (int e = 0; e < 3; ++e) { T1[e] = kfx * Hm[e] + kcx * Hm[6 + e]; T1[3 + e] = kfy * Hm[3 + e] + kcy * Hm[6 + e]; T1[6 + e] = Hm[6 + e]; }
#pragma unroll
        for (int a = 0; a < 3; ++a) { Hk[a * 3] = T1[a * 3] * ifx; Hk[a * 3 + 1] = T1[a * 3 + 1] * ify; Hk[a * 3 + 2] = T1[a * 3] * icx + T1[a * 3 + 1] * icy + T1[a * 3 + 2]; }
        const float px = (float)x / s, py = (float)y / s;
        float Sr = 0.f, Sn = 0.f, Srr = 0.f, Snn = 0.f, Srn = 0.f, du, dv;
        float c_r[4] = {0.f, 0.f, 0.f, 0.f}, c_n[4] = {0.f, 0.f, 0.f, 0.f}, c_du[4] = {0.f, 0.f, 0.f, 0.f}, c_dv[4] = {0.f, 0.f, 0.f, 0.f};
        if (CACHE) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = sub + 16 * k;
                if (j < ntap) {
                    const float uu = px + (float)(j % side - h), vv = py + (float)(j / side - h);
                    const float rj = mv_bilerp0<false>(gray, c.Wg, c.Hg, uu, vv, du, dv);
                    const float g0 = Hk[0] * uu + Hk[1] * vv + Hk[2], g1 = Hk[3] * uu + Hk[4] * vv + Hk[5], g2 = Hk[6] * uu + Hk[7] * vv + Hk[8] + 1e-10f;
                    const float nj = mv_bilerp0<true>(near_gray, c.Wg, c.Hg, g0 / g2, g1 / g2, du, dv);
                    c_r[k] = rj; c_n[k] = nj; c_du[k] = du; c_dv[k] = dv;
                    Sr += rj; Sn += nj; Srr += rj * rj; Snn += nj * nj; Srn += rj * nj;
                }
            }
        } else {
            for (int j = sub; j < ntap; j += 16) {
                const float uu = px + (float)(j % side - h), vv = py + (float)(j / side - h);
                const float rj = mv_bilerp0<false>(gray, c.Wg, c.Hg, uu, vv, du, dv);
                const float g0 = Hk[0] * uu + Hk[1] * vv + Hk[2], g1 = Hk[3] * uu + Hk[4] * vv + Hk[5], g2 = Hk[6] * uu + Hk[7] * vv + Hk[8] + 1e-10f;
                const float nj = mv_bilerp0<false>(near_gray, c.Wg, c.Hg, g0 / g2, g1 / g2, du, dv);
                Sr += rj; Sn += nj; Srr += rj * rj; Snn += nj * nj; Srn += rj * nj;
            }
        }
        Sr = mv_sum16(Sr); Sn = mv_sum16(Sn); Srr = mv_sum16(Srr); Snn = mv_sum16(Snn); Srn = mv_sum16(Srn);
        const float ravg = Sr / tps, navg = Sn / tps;
        const float cross = Srn - navg * Sr, rvar = Srr - ravg * Sr, nvar = Snn - navg * Sn;
        const float den = rvar * nvar + 1e-8f;
        const float cc = cross * cross / den;
        float ncc = 1.f - cc;
        const bool clamped = !(ncc >= 0.f && ncc <= 2.f);
        ncc = fminf(fmaxf(ncc, 0.f), 2.f);
        const bool m = ncc < 0.9f;
        const float w = weight[p];
        if (sub == 0) {
            if (ncc_out) ncc_out[i] = ncc;
            if (mask_out) mask_out[i] = m ? 1 : 0;
            if (m) { s_v = ncc * w; s_c = 1.f; }
        }
        if (m && !clamped && w != 0.f) {                                    // uniform across the 16 lanes of the sample
            const float dcc = -w;
            const float dcross = dcc * 2.f * cross / den, dnvar = -dcc * cross * cross * rvar / (den * den);
            float dH[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            auto tap_grad = [&](float uu, float vv, float rj, float nj, float tdu, float tdv) {
                const float g0 = Hk[0] * uu + Hk[1] * vv + Hk[2], g1 = Hk[3] * uu + Hk[4] * vv + Hk[5], g2 = Hk[6] * uu + Hk[7] * vv + Hk[8] + 1e-10f;
                const float dn = dcross * (rj - ravg) + dnvar * (2.f * nj - 2.f * navg);
                const float dgx = dn * tdu, dgy = dn * tdv;
                const float d0 = dgx / g2, d1 = dgy / g2, d2 = -(dgx * g0 + dgy * g1) / (g2 * g2);
                dH[0] += d0 * uu; dH[1] += d0 * vv; dH[2] += d0;
                dH[3] += d1 * uu; dH[4] += d1 * vv; dH[5] += d1;
                dH[6] += d2 * uu; dH[7] += d2 * vv; dH[8] += d2;
            };
            if (CACHE) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = sub + 16 * k;
                    if (j < ntap) tap_grad(px + (float)(j % side - h), py + (float)(j / side - h), c_r[k], c_n[k], c_du[k], c_dv[k]);
                }
            } else {
                for (int j = sub; j < ntap; j += 16) {
                    const float uu = px + (float)(j % side - h), vv = py + (float)(j / side - h);
                    const float rj = mv_bilerp0<false>(gray, c.Wg, c.Hg, uu, vv, du, dv);
                    const float g0 = Hk[0] * uu + Hk[1] * vv + Hk[2], g1 = Hk[3] * uu + Hk[4] * vv + Hk[5], g2 = Hk[6] * uu + Hk[7] * vv + Hk[8] + 1e-10f;
                    const float nj = mv_bilerp0<true>(near_gray, c.Wg, c.Hg, g0 / g2, g1 / g2, du, dv);
                    tap_grad(uu, vv, rj, nj, du, dv);
                }
            }
#pragma unroll
            for (int e = 0; e < 9; ++e) dH[e] = mv_sum16(dH[e]);
            // dHm = K_near^T dH Kinv_view^T
            float T2[9], dHm[9];
#pragma unroll
            for (int e = 0; e < 3; ++e) { T2[e] = kfx * dH[e]; T2[3 + e] = kfy * dH[3 + e]; T2[6 + e] = kcx * dH[e] + kcy * dH[3 + e] + dH[6 + e]; }
#pragma unroll
            for (int a = 0; a < 3; ++a) { dHm[a * 3] = T2[a * 3] * ifx + T2[a * 3 + 2] * icx; dHm[a * 3 + 1] = T2[a * 3 + 1] * ify + T2[a * 3 + 2] * icy; dHm[a * 3 + 2] = T2[a * 3 + 2]; }
            float gd = 0.f, gn[3];
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const float tb = dHm[e] * b[0] + dHm[3 + e] * b[1] + dHm[6 + e] * b[2];
                gn[e] = -tb / dd; gd += tb * nn[e] / (dd * dd);
            }
            if (sub < 3) g_normal[sub * HW + p] = sub == 0 ? gn[0] : (sub == 1 ? gn[1] : gn[2]);
            if (sub == 3) g_dist[p] = gd;
        }
    } else if (i < N && sub == 0) {
        if (ncc_out) ncc_out[i] = 0.f;
        if (mask_out) mask_out[i] = 0;
    }
    const float2 t = mv_block_sum2(s_v, s_c, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ void __launch_bounds__(1024) k_mv_finish(const float2* __restrict__ partial, int n, float* stats)
{
    __shared__ float r1[16], r2[16];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) { const float2 q = partial[i]; a += q.x; b += q.y; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = a; r2[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float sa = 0.f, sb = 0.f;
        for (int w = 0; w < 16; w++) { sa += r1[w]; sb += r2[w]; }
        stats[0] = sa; stats[1] = sb; stats[2] = sb > 0.f ? sa / sb : 0.f;
    }
}

// The loss values and the gradient scaling of the two multi-view losses without a chain of scalar framework kernels (round 4: the wrapper spent
// ~14 tiny launches per iteration on lambda * mean, clamp(count), lambda / count and three image-sized multiplies):
//   k_mv_values: out[0] = lambda_geo * stats[2], out[1] = lambda_ncc * stats[5]
//   k_mv_scale : o_depth = g_depth * sg, o_near = g_near * sg, o_am = g_am * sn  with  sg = up_geo * lambda_geo / max(stats[1], 1),
//                sn = up_ncc * lambda_ncc / max(stats[4], 1) (d mean / d x = (d sum / d x) / count; an empty mask leaves zero maps); up_* are the
//                upstream gradients of the two loss values (device scalars; NULL = 1).  One launch, out of place (the saved maps survive a
//                second backward).  Optional addends (the single-view normal loss's maps of the same pixels, times their own upstream scalar)
//                let one autograd node hand back the SUM of all three PGSR geometry losses' gradients: no image-sized framework adds.
__global__ void k_mv_values(const float* __restrict__ stats, float lg, float ln, float* __restrict__ out)
{
    out[0] = lg * stats[2]; out[1] = ln * stats[5];
}
__global__ void __launch_bounds__(256) k_mv_scale(size_t n_d, size_t n_n, size_t n_a, const float* __restrict__ g_d, const float* __restrict__ g_n,
                                                  const float* __restrict__ g_a, const float* __restrict__ stats, float lg, float ln,
                                                  const float* __restrict__ up_geo, const float* __restrict__ up_ncc,
                                                  const float* __restrict__ add_d, const float* __restrict__ add_a, const float* __restrict__ up_add, int have_add,
                                                  float* __restrict__ o_d, float* __restrict__ o_n, float* __restrict__ o_a)
{
    // a NULL up_* with its maps present means "upstream gradient 1"; the caller passes n = 0 for maps whose loss sent no gradient
    const float sg = (up_geo ? up_geo[0] : 1.0f) * lg / fmaxf(stats[1], 1.0f);
    const float sn = (up_ncc ? up_ncc[0] : 1.0f) * ln / fmaxf(stats[4], 1.0f);
    const float sa = have_add ? (up_add ? up_add[0] : 1.0f) : 0.0f;
    const size_t total = n_d + n_n + n_a, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        if (i < n_d) o_d[i] = (g_d ? g_d[i] * sg : 0.0f) + (add_d ? add_d[i] * sa : 0.0f);
        else if (i < n_d + n_n) o_n[i - n_d] = g_n[i - n_d] * sg;
        else { const size_t k = i - n_d - n_n; o_a[k] = (g_a ? g_a[k] * sn : 0.0f) + (add_a ? add_a[k] * sa : 0.0f); }
    }
}
extern "C" int gsr_loss_plane_mv_values(const float* stats, float lambda_geo, float lambda_ncc, float* out2, void* stream)
{
    if (!stats || !out2) { gsr_set_error("loss_plane_mv_values: null pointer"); return 1; }
    hipLaunchKernelGGL(k_mv_values, dim3(1), dim3(1), 0, (hipStream_t)stream, stats, lambda_geo, lambda_ncc, out2);
    return gsr_check_launch("loss_plane_mv_values", (hipStream_t)stream, false);
}
extern "C" int gsr_loss_plane_mv_scale(size_t n_depth, size_t n_near, size_t n_am, const float* g_depth, const float* g_near, const float* g_am,
                                       const float* stats, float lambda_geo, float lambda_ncc, const float* up_geo, const float* up_ncc,
                                       const float* add_depth, const float* add_am, const float* up_add, int32_t have_add,
                                       float* o_depth, float* o_near, float* o_am, void* stream)
{
    if (!stats || (n_depth && ((!g_depth && !add_depth) || !o_depth)) || (n_near && (!g_near || !o_near)) || (n_am && ((!g_am && !add_am) || !o_am))) {
        gsr_set_error("loss_plane_mv_scale: null pointer"); return 1;
    }
    const size_t total = n_depth + n_near + n_am;
    if (total == 0) return 0;
    const unsigned blocks = (unsigned)((total + 1023) / 1024 < 8192 ? (total + 1023) / 1024 : 8192);
    hipLaunchKernelGGL(k_mv_scale, dim3(blocks ? blocks : 1), dim3(256), 0, (hipStream_t)stream, n_depth, n_near, n_am, g_depth, g_near, g_am, stats, lambda_geo,
                       lambda_ncc, up_geo, up_ncc, add_depth, add_am, up_add, have_add ? 1 : 0, o_depth, o_near, o_am);
    return gsr_check_launch("loss_plane_mv_scale", (hipStream_t)stream, false);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Uniform sample WITHOUT replacement of at most `num` set entries of a byte mask, on the device and without a sort
// (pgsr_scene.py:147-151 draws it with np.random.choice on the host).  Every entry gets a 24-bit hashed key from (seed, index); the `num`
// smallest keys are taken: two 4096-bin histogram levels locate the exact 24-bit threshold, then a count / scan / write pass emits the
// selected indices in ascending order (gather locality for k_mv_ncc) followed by the threshold ties needed to fill up, -1 in unused slots.
#define SM_ITEMS 16
#define SM_BLOCK (256 * SM_ITEMS)
struct SmSel { uint32_t bin0, below0, bin1, below1, T24, c_lt, all; };

__device__ __forceinline__ uint32_t sm_key24(uint32_t p, uint32_t s0, uint32_t s1)
{
    uint32_t x = p * 0x9E3779B1u + s0;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x += s1; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x >> 8;
}

template <int LEVEL>
__global__ void __launch_bounds__(256) k_sm_hist(int64_t n, const uint8_t* __restrict__ mask, uint32_t s0, uint32_t s1, const SmSel* __restrict__ sel,
                                                 uint32_t* __restrict__ hist)
{
    __shared__ uint32_t h[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) h[i] = 0;
    __syncthreads();
    const uint32_t b0 = LEVEL ? sel->bin0 : 0u;
    if (!(LEVEL && sel->all)) {
        const int64_t base = (int64_t)blockIdx.x * SM_BLOCK;
#pragma unroll 4
        for (int it = 0; it < SM_ITEMS; ++it) {
            const int64_t p = base + it * 256 + threadIdx.x;
            if (p < n && mask[p]) {
                const uint32_t k = sm_key24((uint32_t)p, s0, s1);
                if (!LEVEL) atomicAdd(&h[k >> 12], 1u);
                else if ((k >> 12) == b0) atomicAdd(&h[k & 4095u], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 256)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// one block: boundary bin = first bin whose inclusive prefix exceeds `need`
template <int LEVEL>
__global__ void __launch_bounds__(1024) k_sm_pick(const uint32_t* __restrict__ hist, uint32_t num, SmSel* sel)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t found_bin, found_below;
    if (threadIdx.x == 0) { found_bin = 4096u; found_below = 0u; }
    const uint32_t need = LEVEL ? num - sel->below0 : num;
    uint32_t v[4], t = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = hist[threadIdx.x * 4 + e]; t += v[e]; }
    uint32_t inc = t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((threadIdx.x & 63) >= d) inc += o; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
    uint32_t excl = woff + inc - t;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (excl <= need && excl + v[e] > need) { found_bin = threadIdx.x * 4 + e; found_below = excl; }   // unique: prefixes are monotone
        excl += v[e];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!LEVEL) { sel->bin0 = found_bin; sel->below0 = found_below; sel->all = found_bin == 4096u; }
        else {
            sel->bin1 = found_bin; sel->below1 = found_below;
            if (sel->all) { sel->T24 = 0x1000000u; sel->c_lt = 0xFFFFFFFFu; }
            else { sel->T24 = (sel->bin0 << 12) | (found_bin & 4095u); sel->c_lt = sel->below0 + found_below; }
        }
    }
}

__global__ void __launch_bounds__(256) k_sm_count(int64_t n, const uint8_t* __restrict__ mask, uint32_t s0, uint32_t s1, const SmSel* __restrict__ sel,
                                                  uint2* __restrict__ blockcnt)
{
    __shared__ float red[8];
    const uint32_t T = sel->T24;
    const int64_t base = (int64_t)blockIdx.x * SM_BLOCK;
    float lt = 0.f, eq = 0.f;                                        // <= 4096 per block: exact in float
    for (int it = 0; it < SM_ITEMS; ++it) {
        const int64_t p = base + it * 256 + threadIdx.x;
        if (p < n && mask[p]) { const uint32_t k = sm_key24((uint32_t)p, s0, s1); lt += k < T ? 1.f : 0.f; eq += k == T ? 1.f : 0.f; }
    }
    const float2 t = mv_block_sum2(lt, eq, red);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = make_uint2((uint32_t)t.x, (uint32_t)t.y);
}

__global__ void __launch_bounds__(1024) k_sm_scan(uint2* __restrict__ blockcnt, uint32_t nblk)
{
    __shared__ uint32_t wa[16], wb[16];
    __shared__ uint32_t carry_a, carry_b;
    if (threadIdx.x == 0) { carry_a = 0; carry_b = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < nblk; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint2 v = i < nblk ? blockcnt[i] : make_uint2(0u, 0u);
        uint32_t a = v.x, b = v.y;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t oa = __shfl_up(a, d, 64), ob = __shfl_up(b, d, 64);
            if ((threadIdx.x & 63) >= d) { a += oa; b += ob; }
        }
        if ((threadIdx.x & 63) == 63) { wa[threadIdx.x >> 6] = a; wb[threadIdx.x >> 6] = b; }
        __syncthreads();
        uint32_t oa = carry_a, ob = carry_b;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) { oa += wa[w]; ob += wb[w]; }
        if (i < nblk) blockcnt[i] = make_uint2(oa + a - v.x, ob + b - v.y);
        __syncthreads();
        if (threadIdx.x == 1023) { carry_a = oa + a; carry_b = ob + b; }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_sm_write(int64_t n, const uint8_t* __restrict__ mask, uint32_t s0, uint32_t s1, const SmSel* __restrict__ sel,
                                                  const uint2* __restrict__ blockoff, uint32_t num, int32_t* __restrict__ idx)
{
    __shared__ uint32_t wl[4], we[4];
    const uint32_t T = sel->T24;
    const uint32_t c_lt = sel->all ? 0xFFFFFFFFu : sel->c_lt;
    const uint2 off = blockoff[blockIdx.x];
    uint32_t run_l = off.x, run_e = off.y;
    const int64_t base = (int64_t)blockIdx.x * SM_BLOCK;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int it = 0; it < SM_ITEMS; ++it) {
        const int64_t p = base + it * 256 + threadIdx.x;
        bool l = false, e = false;
        if (p < n && mask[p]) { const uint32_t k = sm_key24((uint32_t)p, s0, s1); l = k < T; e = k == T; }
        const uint64_t bl = __ballot(l), be = __ballot(e);
        const uint64_t lo = lane ? (~0ull >> (64 - lane)) : 0ull;
        if (lane == 0) { wl[wv] = (uint32_t)__popcll(bl); we[wv] = (uint32_t)__popcll(be); }
        __syncthreads();
        uint32_t pl = 0, pe = 0, tl = 0, te = 0;
        for (int w = 0; w < 4; ++w) { if (w < wv) { pl += wl[w]; pe += we[w]; } tl += wl[w]; te += we[w]; }
        if (l) { const uint32_t slot = run_l + pl + (uint32_t)__popcll(bl & lo); if (slot < num) idx[slot] = (int32_t)p; }
        if (e && c_lt != 0xFFFFFFFFu) { const uint32_t slot = c_lt + run_e + pe + (uint32_t)__popcll(be & lo); if (slot < num) idx[slot] = (int32_t)p; }
        run_l += tl; run_e += te;
        __syncthreads();
    }
}

extern "C" size_t gsr_sample_mask_scratch_bytes(int64_t n)
{
    const size_t nblk = (size_t)((n > 0 ? n : 1) + SM_BLOCK - 1) / SM_BLOCK;
    return gsr_align(2 * 4096 * sizeof(uint32_t)) + gsr_align(sizeof(SmSel)) + gsr_align(nblk * sizeof(uint2));
}

extern "C" int gsr_sample_mask(int64_t n, const uint8_t* mask, int32_t num, uint64_t seed, int32_t* idx_out, void* scratch, size_t scratch_bytes,
                               void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n <= 0 || num <= 0) return 0;
    if (n >= ((int64_t)1 << 31)) { gsr_set_error("sample_mask: n must be < 2^31"); return 1; }
    if (!mask || !idx_out || !scratch || scratch_bytes < gsr_sample_mask_scratch_bytes(n)) { gsr_set_error("sample_mask: null pointer or scratch too small"); return 1; }
    const uint32_t nblk = (uint32_t)((n + SM_BLOCK - 1) / SM_BLOCK);
    char* q = (char*)scratch;
    uint32_t* hist = (uint32_t*)q; q += gsr_align(2 * 4096 * sizeof(uint32_t));
    SmSel* sel = (SmSel*)q; q += gsr_align(sizeof(SmSel));
    uint2* blockcnt = (uint2*)q;
    const uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
    const int64_t slots = n < (int64_t)num ? n : (int64_t)num;
    (void)gsr_memset_async(hist, 0, 2 * 4096 * sizeof(uint32_t), s);
    (void)gsr_memset_async(idx_out, 0xFF, sizeof(int32_t) * (size_t)slots, s);
    hipLaunchKernelGGL(k_sm_hist<0>, dim3(nblk), dim3(256), 0, s, n, mask, s0, s1, (const SmSel*)sel, hist);
    hipLaunchKernelGGL(k_sm_pick<0>, dim3(1), dim3(1024), 0, s, (const uint32_t*)hist, (uint32_t)num, sel);
    hipLaunchKernelGGL(k_sm_hist<1>, dim3(nblk), dim3(256), 0, s, n, mask, s0, s1, (const SmSel*)sel, hist + 4096);
    hipLaunchKernelGGL(k_sm_pick<1>, dim3(1), dim3(1024), 0, s, (const uint32_t*)(hist + 4096), (uint32_t)num, sel);
    hipLaunchKernelGGL(k_sm_count, dim3(nblk), dim3(256), 0, s, n, mask, s0, s1, (const SmSel*)sel, blockcnt);
    hipLaunchKernelGGL(k_sm_scan, dim3(1), dim3(1024), 0, s, blockcnt, nblk);
    hipLaunchKernelGGL(k_sm_write, dim3(nblk), dim3(256), 0, s, n, mask, s0, s1, (const SmSel*)sel, (const uint2*)blockcnt, (uint32_t)slots, idx_out);
    return gsr_check_launch("sample_mask", s, false);
}

static bool mv_cfg_ok(const gsr_mv_cfg* c)
{
    return c && c->W > 0 && c->H > 0 && c->Wn > 0 && c->Hn > 0 && c->Wg > 0 && c->Hg > 0 && c->patch >= 0 && c->patch <= 8 && c->ncc_scale > 0.f &&
           (int64_t)c->W * c->H < (int64_t)1 << 30;
}

extern "C" size_t gsr_loss_plane_mv_scratch_bytes(int32_t W, int32_t H, int32_t n_samples)
{
    const size_t geo = (size_t)gsr_div_up(W > 0 ? W : 1, 32) * gsr_div_up(H > 0 ? H : 1, 8);
    const size_t ncc = (size_t)gsr_div_up(n_samples > 0 ? n_samples : 1, 16);
    return (geo > ncc ? geo : ncc) * sizeof(float2);
}

extern "C" int gsr_loss_plane_mv_geo(const gsr_mv_cfg* cfg, const float* plane_depth, const float* near_plane_depth, float* noise, uint8_t* d_mask,
                                     float* weight, float* stats, float* g_depth, float* g_near, void* scratch, size_t scratch_bytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (!mv_cfg_ok(cfg)) { gsr_set_error("loss_plane_mv_geo: bad configuration"); return 1; }
    if (!plane_depth || !near_plane_depth || !noise || !d_mask || !weight || !stats || !g_depth || !g_near || !scratch ||
        scratch_bytes < gsr_loss_plane_mv_scratch_bytes(cfg->W, cfg->H, 0)) {
        gsr_set_error("loss_plane_mv_geo: null pointer or scratch too small"); return 1;
    }
    const dim3 grid(gsr_div_up(cfg->W, 32), gsr_div_up(cfg->H, 8));
    (void)gsr_memset_async(g_near, 0, sizeof(float) * (size_t)cfg->Wn * cfg->Hn, s);
    hipLaunchKernelGGL(k_mv_geo, grid, dim3(256), 0, s, *cfg, plane_depth, near_plane_depth, noise, d_mask, weight, g_depth, g_near, (float2*)scratch);
    hipLaunchKernelGGL(k_mv_finish, dim3(1), dim3(1024), 0, s, (const float2*)scratch, (int)(grid.x * grid.y), stats);
    return gsr_check_launch("loss_plane_mv_geo", s, false);
}

extern "C" int gsr_loss_plane_mv_ncc(const gsr_mv_cfg* cfg, int32_t n_samples, const int32_t* idx, const float* weight, const float* normal,
                                     const float* distance, const float* gray, const float* near_gray, float* ncc, uint8_t* mask, float* stats,
                                     float* g_normal, float* g_distance, void* scratch, size_t scratch_bytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (!mv_cfg_ok(cfg) || n_samples < 0) { gsr_set_error("loss_plane_mv_ncc: bad configuration"); return 1; }
    if (!weight || !normal || !distance || !gray || !near_gray || !stats || !g_normal || !g_distance || !scratch || (n_samples > 0 && !idx) ||
        scratch_bytes < gsr_loss_plane_mv_scratch_bytes(0, 0, n_samples)) {
        gsr_set_error("loss_plane_mv_ncc: null pointer or scratch too small"); return 1;
    }
    const size_t HW = (size_t)cfg->W * cfg->H;
    (void)gsr_memset_async(g_normal, 0, sizeof(float) * 3 * HW, s);
    (void)gsr_memset_async(g_distance, 0, sizeof(float) * HW, s);
    const int blocks = gsr_div_up(n_samples > 0 ? n_samples : 1, 16);
    const int ntap = (2 * cfg->patch + 1) * (2 * cfg->patch + 1);
    if (ntap <= 64)
        hipLaunchKernelGGL(k_mv_ncc<true>, dim3(blocks), dim3(256), 0, s, *cfg, (int)n_samples, idx, weight, normal, distance, gray, near_gray, ncc, mask,
                           g_normal, g_distance, (float2*)scratch);
    else
        hipLaunchKernelGGL(k_mv_ncc<false>, dim3(blocks), dim3(256), 0, s, *cfg, (int)n_samples, idx, weight, normal, distance, gray, near_gray, ncc, mask,
                           g_normal, g_distance, (float2*)scratch);
    hipLaunchKernelGGL(k_mv_finish, dim3(1), dim3(1024), 0, s, (const float2*)scratch, blocks, stats);
    return gsr_check_launch("loss_plane_mv_ncc", s, false);
}
