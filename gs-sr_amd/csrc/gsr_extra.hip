// gsr_extra.hip -- adjacent kernels of the hot path:
//   gsr_tsdf_integrate : per-frame TSDF voxel update, the in-repo definition gssr/utils/mesh_utils.py:195-246
//   gsr_dist2          : simple_knn.distCUDA2 (mean squared distance to the 3 nearest neighbours)
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

// One thread per voxel/sample point; pure streaming RMW of (tsdf, weight, rgb): 20 B read + 20 B write per touched voxel.
__global__ void __launch_bounds__(256) k_tsdf_integrate(int64_t V, const float* __restrict__ points, const float* __restrict__ Fp, int W, int H,
                                                        const float* __restrict__ depth, const float* __restrict__ rgb, float sdf_trunc,
                                                        const float* __restrict__ trunc_pp, float* __restrict__ tsdf, float* __restrict__ weight,
                                                        float* __restrict__ rgb_acc)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    float F[16];
#pragma unroll
    for (int k = 0; k < 16; k++) F[k] = Fp[k];
    const float x = points[3 * i], y = points[3 * i + 1], z3 = points[3 * i + 2];
    const float qx = x * F[0] + y * F[4] + z3 * F[8] + F[12];
    const float qy = x * F[1] + y * F[5] + z3 * F[9] + F[13];
    const float qw = x * F[3] + y * F[7] + z3 * F[11] + F[15];
    const float u = qx / qw, v = qy / qw;
    bool mask = (u > -1.f) && (u < 1.f) && (v > -1.f) && (v < 1.f) && (qw > 0);
    const float d = bilinear_border(depth, W, H, u, v);
    const float sdf = d - qw;
    const float tr = trunc_pp ? trunc_pp[i] : sdf_trunc;
    mask = mask && (sdf > -tr);
    if (!mask) return;
    float s = sdf / tr;
    s = fminf(fmaxf(s, -1.0f), 1.0f);
    const float w = weight[i], wp = w + 1;
    tsdf[i] = (tsdf[i] * w + s) / wp;
    const size_t HW = (size_t)W * H;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float col = bilinear_border(rgb + c * HW, W, H, u, v);
        rgb_acc[3 * i + c] = (rgb_acc[3 * i + c] * w + col) / wp;
    }
    weight[i] = wp;
}

extern "C" int gsr_tsdf_integrate(int64_t V, const float* points, const float* full_proj, int32_t W, int32_t H, const float* depth,
                                  const float* rgb, float sdf_trunc, const float* sdf_trunc_per_point, float* tsdf, float* weight,
                                  float* rgb_acc, void* stream)
{
    if (V <= 0) return 0;
    const int64_t nb = (V + 255) / 256;
    if (nb > 0x7FFFFFFF) { gsr_set_error("tsdf: too many points"); return 1; }
    hipLaunchKernelGGL(k_tsdf_integrate, dim3((uint32_t)nb), dim3(256), 0, (hipStream_t)stream, V, points, full_proj, W, H, depth, rgb,
                       sdf_trunc, sdf_trunc_per_point, tsdf, weight, rgb_acc);
    return gsr_check_launch("tsdf_integrate", (hipStream_t)stream, false);
}

// ---- distCUDA2 (simple-knn/simple_knn.cu:148-184).  Exact 3-NN by LDS-tiled all-pairs: every block owns 256 query
// points and streams all points through LDS in tiles of 256.  O(P^2) but init-time only; the reference's Morton/box
// pruning yields the same exact answer.
__global__ void __launch_bounds__(256) k_dist2(int P, const float* __restrict__ pts, float* __restrict__ out)
{
    __shared__ float sx[256], sy[256], sz[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float px = 0, py = 0, pz = 0;
    if (i < P) { px = pts[3 * i]; py = pts[3 * i + 1]; pz = pts[3 * i + 2]; }
    float b0 = 3.402823466e+38f, b1 = b0, b2 = b0;
    for (int t = 0; t < P; t += 256) {
        const int j = t + threadIdx.x;
        __syncthreads();
        if (j < P) { sx[threadIdx.x] = pts[3 * j]; sy[threadIdx.x] = pts[3 * j + 1]; sz[threadIdx.x] = pts[3 * j + 2]; }
        __syncthreads();
        const int n = min(256, P - t);
        for (int k = 0; k < n; k++) {
            if (t + k == i) continue;
            const float dx = sx[k] - px, dy = sy[k] - py, dz = sz[k] - pz;
            float d = dx * dx + dy * dy + dz * dz;
            if (b0 > d) { float tmp = b0; b0 = d; d = tmp; }
            if (b1 > d) { float tmp = b1; b1 = d; d = tmp; }
            if (b2 > d) { b2 = d; }
        }
    }
    if (i < P) out[i] = (b0 + b1 + b2) / 3.0f;
}

extern "C" size_t gsr_dist2_scratch_bytes(int32_t P) { (void)P; return 256; }
extern "C" int gsr_dist2(int32_t P, const float* points, float* out, void* scratch, size_t scratch_bytes, void* stream)
{
    (void)scratch; (void)scratch_bytes;
    if (P <= 0) return 0;
    hipLaunchKernelGGL(k_dist2, dim3(gsr_div_up(P, 256)), dim3(256), 0, (hipStream_t)stream, P, points, out);
    return gsr_check_launch("dist2", (hipStream_t)stream, false);
}
