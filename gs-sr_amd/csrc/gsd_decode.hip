// gsd_decode.hip -- fused neural-Gaussian decode (include/gsdecode.h), gfx950 only.
//
// Reference behaviour: ScaffoldScene/OctreeScene.generate_neural_gaussians (gssr/scene/scaffold_scene.py:27-120,
// gssr/scene/octree_scene.py:26-133) -- ~60 torch ops + autograd; here 2 forward kernels and 3 backward kernels.
//
// Mapping to the hardware
//   * forward / per-anchor backward: one lane = one visible anchor, 64 anchors per wave.  The three heads are 32-wide MLPs whose weights
//     are identical for every lane, so they are fetched with SCALAR loads (constant address space -> s_load_dwordx8/16 into SGPRs) and
//     used directly as the SGPR operand of v_fmac: no LDS, no VGPRs for weights, the per-anchor state (x[37], h[32], dh[32], dx[37])
//     stays in registers.  fp32 MFMA has the same peak as packed fp32 FMA on CDNA4, so MFMA buys nothing for these lane-local products.
//   * weight gradients: dW = sum over anchors of (dpre x^T) IS a contraction across lanes, i.e. exactly what MFMA does.  The per-anchor
//     kernel stores x, h and the pre-activation gradients feature-major ([column][anchor], coalesced); k_wgrad then runs
//     v_mfma_f32_16x16x4_f32 with the anchor index as the K dimension (operands are plain float4 loads, 16 consecutive anchors of one
//     column per lane), split-K over persistent waves, followed by a small deterministic reduction of the partial tiles.
//   * layer-1 weights are first repacked (k_pack) to a fixed [head][32][48] layout: columns feat 0..31, view 32..34, dist 35, level 36,
//     effective bias 37 (b1 + W1[:, appearance cols] . appearance) -- absent inputs get zero weights, so the hot kernels carry no flags.
#include "gsr_common.h"
#include "../../include/gsdecode.h"

#define GSD_XC 37            // x = [feat 32, view 3, dist, level]
#define GSD_W1LD 48          // packed layer-1 row stride (floats); column 37 = effective bias
#define GSD_W1P_FLOATS (3 * 32 * GSD_W1LD)
#define GSD_BLOCK 256

typedef const float __attribute__((address_space(4))) * cfp;     // constant address space: wave-uniform index -> scalar load
#define CW(p) ((cfp)(p))

// column blocks of the feature-major backward scratch (units: columns of ld floats each)
#define SC_X 0               // 48 (37 used)
#define SC_H 48              // 3 x 32
#define SC_P1 144            // dpre1: 3 x 32
#define SC_P2O 240           // dpre2 opacity: 16
#define SC_P2K 256           // dpre2 colour: 48
#define SC_P2C 304           // dpre2 cov: 112
#define SC_COLS 416

#define WG_WAVES 128         // persistent waves per weight-gradient problem
#define WG_MAXPROB 24

struct DecArgs {
    gsd_cfg cfg;
    gsd_inputs in;
    const float* W1p;
    const float *W2o, *b2o, *W2c, *b2c, *W2k, *b2k;
};

// ------------------------------------------------------------------------------------------------ small device helpers
__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ void load_x(const DecArgs& p, int a, float (&x)[GSD_XC], float (&vw)[3], float& dist)
{
    const float4* f4 = reinterpret_cast<const float4*>(p.in.feat + (size_t)a * GSD_FEAT);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const float4 t = f4[q];
        x[4 * q] = t.x; x[4 * q + 1] = t.y; x[4 * q + 2] = t.z; x[4 * q + 3] = t.w;
    }
    const float r0 = p.in.anchor[3 * a] - p.in.campos[0], r1 = p.in.anchor[3 * a + 1] - p.in.campos[1],
                r2 = p.in.anchor[3 * a + 2] - p.in.campos[2];
    dist = sqrtf(r0 * r0 + r1 * r1 + r2 * r2);
    vw[0] = r0 / dist; vw[1] = r1 / dist; vw[2] = r2 / dist;
    x[32] = vw[0]; x[33] = vw[1]; x[34] = vw[2];
    x[35] = dist;
    x[36] = p.in.level ? p.in.level[a] : 0.0f;
}

// h = relu(W1p[head] [x;1])
template <int HEAD>
__device__ __forceinline__ void layer1(cfp W1p, const float (&x)[GSD_XC], float (&h)[32])
{
    cfp w = W1p + HEAD * 32 * GSD_W1LD;
#pragma unroll
    for (int j = 0; j < 32; j++) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < GSD_XC; i++) s = fmaf(w[j * GSD_W1LD + i], x[i], s);
        s += w[j * GSD_W1LD + GSD_XC];
        h[j] = fmaxf(s, 0.0f);
    }
}

// dx += W1p[head]^T dpre1
template <int HEAD>
__device__ __forceinline__ void layer1_bwd(cfp W1p, const float (&g)[32], float (&dx)[GSD_XC])
{
    cfp w = W1p + HEAD * 32 * GSD_W1LD;
#pragma unroll
    for (int j = 0; j < 32; j++) {
#pragma unroll
        for (int i = 0; i < GSD_XC; i++) dx[i] = fmaf(w[j * GSD_W1LD + i], g[j], dx[i]);
    }
}

__device__ __forceinline__ float dot32(cfp w, const float (&h)[32])
{
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; i++) s = fmaf(w[i], h[i], s);
    return s;
}

__device__ __forceinline__ void axpy32(cfp w, float g, float (&dh)[32])
{
#pragma unroll
    for (int i = 0; i < 32; i++) dh[i] = fmaf(w[i], g, dh[i]);
}

// ------------------------------------------------------------------------------------------------ scan (exclusive, in place)
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

__global__ void __launch_bounds__(1024) k_scan_block(uint32_t* __restrict__ data, uint32_t n, uint32_t* __restrict__ sums)
{
    __shared__ uint32_t wsum[16];
    const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
    const uint32_t v = i < n ? data[i] : 0u;
    const uint32_t incl = wave_scan_incl(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; w++) base += wsum[w];
    if (i < n) data[i] = base + incl - v;
    if (threadIdx.x == 1023) sums[blockIdx.x] = base + incl;
}

__global__ void __launch_bounds__(1024) k_scan_sums(uint32_t* __restrict__ sums, uint32_t nblk, uint32_t* __restrict__ total)
{
    __shared__ uint32_t wsum[16];
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < nblk; c0 += 1024u) {
        const uint32_t i = c0 + threadIdx.x;
        const uint32_t v = i < nblk ? sums[i] : 0u;
        const uint32_t incl = wave_scan_incl(v);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t base = 0, all = 0;
        for (int w = 0; w < 16; w++) { if (w < wave) base += wsum[w]; all += wsum[w]; }
        if (i < nblk) sums[i] = carry + base + incl - v;
        carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ void __launch_bounds__(1024) k_scan_add(uint32_t* __restrict__ data, uint32_t n, const uint32_t* __restrict__ sums)
{
    const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
    if (i < n) data[i] += sums[blockIdx.x];
}

// data[0..n) -> exclusive prefix in place, *total_dev = sum.  sums: >= div_up(n,1024) words.
static void launch_scan(uint32_t* data, uint32_t n, uint32_t* sums, uint32_t* total_dev, hipStream_t s)
{
    const uint32_t nblk = gsr_div_up(n, 1024u);
    hipLaunchKernelGGL(k_scan_block, dim3(nblk), dim3(1024), 0, s, data, n, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, sums, nblk, total_dev);
    hipLaunchKernelGGL(k_scan_add, dim3(nblk), dim3(1024), 0, s, data, n, sums);
}

// ------------------------------------------------------------------------------------------------ visible-anchor compaction
__global__ void __launch_bounds__(256) k_flags(const uint8_t* __restrict__ mask, uint32_t n, uint32_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) flags[i] = mask[i] ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_scatter_idx(const uint8_t* __restrict__ mask, uint32_t n, const uint32_t* __restrict__ pos,
                                                     int32_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n && mask[i]) out[pos[i]] = (int32_t)i;
}

// ------------------------------------------------------------------------------------------------ weight repack
struct PackArgs {
    gsd_cfg cfg;
    const float *W1o, *b1o, *W1c, *b1c, *W1k, *b1k, *app;
    float* W1p;
};
__global__ void __launch_bounds__(256) k_pack(PackArgs p)
{
    const int lv = p.cfg.level ? 1 : 0;
    for (int e = threadIdx.x; e < 3 * 32 * GSD_W1LD; e += 256) {
        const int head = e / (32 * GSD_W1LD), j = (e / GSD_W1LD) % 32, c = e % GSD_W1LD;
        const float* W = head == 0 ? p.W1o : (head == 1 ? p.W1c : p.W1k);
        const float* b = head == 0 ? p.b1o : (head == 1 ? p.b1c : p.b1k);
        const int dist = head == 0 ? p.cfg.dist_o : (head == 1 ? p.cfg.dist_c : p.cfg.dist_k);
        const int A = head == 2 ? p.cfg.A : 0;
        const int in = 35 + (dist ? 1 : 0) + lv + A;
        float v = 0.0f;
        if (c < 35) v = W[j * in + c];
        else if (c == 35) v = dist ? W[j * in + 35] : 0.0f;
        else if (c == 36) v = lv ? W[j * in + 35 + (dist ? 1 : 0)] : 0.0f;
        else if (c == GSD_XC) {
            v = 0.0f;
            const int base = 35 + (dist ? 1 : 0) + lv;
            for (int i = 0; i < A; i++) v = fmaf(W[j * in + base + i], p.app[i], v);     // appearance is the same for every anchor
            v += b[j];
        }
        p.W1p[e] = v;
    }
}

// ------------------------------------------------------------------------------------------------ forward stage 1: opacity head
__global__ void __launch_bounds__(GSD_BLOCK) k_decode_opacity(DecArgs p, float* __restrict__ neural_opacity, uint8_t* __restrict__ mask,
                                                              uint32_t* __restrict__ counts)
{
    const int v = blockIdx.x * GSD_BLOCK + threadIdx.x;
    const bool active = v < p.cfg.Nv;
    const int a = p.in.vis_idx[active ? v : 0];
    float x[GSD_XC], vw[3], dist, h[32];
    load_x(p, a, x, vw, dist);
    layer1<0>(CW(p.W1p), x, h);
    const float sc = p.in.opacity_scale ? p.in.opacity_scale[a] : 1.0f;
    const int k = p.cfg.k;
    uint32_t cnt = 0;
    for (int j = 0; j < k; j++) {
        float o = tanhf(dot32(CW(p.W2o) + j * 32, h) + CW(p.b2o)[j]);
        if (p.in.opacity_scale) o *= sc;
        if (active) {
            neural_opacity[(size_t)v * k + j] = o;
            mask[(size_t)v * k + j] = o > 0.0f ? 1 : 0;
        }
        cnt += o > 0.0f ? 1u : 0u;
    }
    if (active) counts[v] = cnt;
}

// ------------------------------------------------------------------------------------------------ forward stage 2: cov + colour heads
__global__ void __launch_bounds__(GSD_BLOCK) k_decode_emit(DecArgs p, const float* __restrict__ neural_opacity,
                                                           const uint32_t* __restrict__ row_offset, gsd_outputs out)
{
    const int v = blockIdx.x * GSD_BLOCK + threadIdx.x;
    const bool active = v < p.cfg.Nv;
    const int a = p.in.vis_idx[active ? v : 0];
    const int k = p.cfg.k;
    float x[GSD_XC], vw[3], dist, h[32];
    load_x(p, a, x, vw, dist);
    const uint32_t row0 = active ? row_offset[v] : 0u;
    float S[6], A3[3];
#pragma unroll
    for (int i = 0; i < 6; i++) S[i] = p.in.scaling[(size_t)a * 6 + i];
#pragma unroll
    for (int i = 0; i < 3; i++) A3[i] = p.in.anchor[3 * a + i];
    uint32_t mbits = 0;
    for (int j = 0; j < k; j++) mbits |= (active && neural_opacity[(size_t)v * k + j] > 0.0f) ? (1u << j) : 0u;

    layer1<1>(CW(p.W1p), x, h);
    for (int j = 0; j < k; j++) {
        float sr[7];
#pragma unroll
        for (int r = 0; r < 7; r++) sr[r] = dot32(CW(p.W2c) + (7 * j + r) * 32, h) + CW(p.b2c)[7 * j + r];
        if (mbits & (1u << j)) {
            const uint32_t row = row0 + __popc(mbits & ((1u << j) - 1u));
            const float* off = p.in.offset + ((size_t)a * k + j) * 3;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                out.xyz[(size_t)row * 3 + i] = A3[i] + off[i] * S[i];
                out.scaling[(size_t)row * 3 + i] = S[3 + i] * sigmoid_(sr[i]);
            }
            float n = sqrtf(sr[3] * sr[3] + sr[4] * sr[4] + sr[5] * sr[5] + sr[6] * sr[6]);
            n = fmaxf(n, 1e-12f);
            reinterpret_cast<float4*>(out.rot)[row] = make_float4(sr[3] / n, sr[4] / n, sr[5] / n, sr[6] / n);
            out.opacity[row] = neural_opacity[(size_t)v * k + j];
        }
    }
    layer1<2>(CW(p.W1p), x, h);
    for (int j = 0; j < k; j++) {
        float c[3];
#pragma unroll
        for (int r = 0; r < 3; r++) c[r] = sigmoid_(dot32(CW(p.W2k) + (3 * j + r) * 32, h) + CW(p.b2k)[3 * j + r]);
        if (mbits & (1u << j)) {
            const uint32_t row = row0 + __popc(mbits & ((1u << j) - 1u));
#pragma unroll
            for (int r = 0; r < 3; r++) out.color[(size_t)row * 3 + r] = c[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward, per anchor
__global__ void __launch_bounds__(GSD_BLOCK) k_decode_bwd(DecArgs p, const float* __restrict__ neural_opacity,
                                                          const uint32_t* __restrict__ row_offset, gsd_out_grads og, float* __restrict__ d_anchor,
                                                          float* __restrict__ d_feat, float* __restrict__ d_offset, float* __restrict__ d_scaling,
                                                          float* __restrict__ sc, size_t ld)
{
    const int v = blockIdx.x * GSD_BLOCK + threadIdx.x;        // grid covers the padded anchor count: inactive lanes store zeros
    const bool active = v < p.cfg.Nv;
    const float am = active ? 1.0f : 0.0f;
    const int a = p.in.vis_idx[active ? v : 0];
    const int k = p.cfg.k;
    float x[GSD_XC], vw[3], dist, h[32], dh[32], dx[GSD_XC];
    load_x(p, a, x, vw, dist);
    const uint32_t row0 = active ? row_offset[v] : 0u;
    float S[6], dS[6] = {0, 0, 0, 0, 0, 0}, dA[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < 6; i++) S[i] = p.in.scaling[(size_t)a * 6 + i];
    uint32_t mbits = 0;
    for (int j = 0; j < k; j++) mbits |= (active && neural_opacity[(size_t)v * k + j] > 0.0f) ? (1u << j) : 0u;
    const float osc = p.in.opacity_scale ? p.in.opacity_scale[a] : 1.0f;
#pragma unroll
    for (int i = 0; i < GSD_XC; i++) { dx[i] = 0.0f; sc[(size_t)(SC_X + i) * ld + v] = x[i] * am; }

    // ---- head o: opacity = tanh(.) * scale
    layer1<0>(CW(p.W1p), x, h);
#pragma unroll
    for (int i = 0; i < 32; i++) dh[i] = 0.0f;
    for (int j = 0; j < k; j++) {
        float g = 0.0f;
        if (mbits & (1u << j)) {
            const uint32_t row = row0 + __popc(mbits & ((1u << j) - 1u));
            const float t = tanhf(dot32(CW(p.W2o) + j * 32, h) + CW(p.b2o)[j]);
            g = og.opacity[row] * osc * (1.0f - t * t);
        }
        sc[(size_t)(SC_P2O + j) * ld + v] = g;
        axpy32(CW(p.W2o) + j * 32, g, dh);
    }
#pragma unroll
    for (int i = 0; i < 32; i++) {
        dh[i] = h[i] > 0.0f ? dh[i] : 0.0f;
        sc[(size_t)(SC_H + i) * ld + v] = h[i] * am;
        sc[(size_t)(SC_P1 + i) * ld + v] = dh[i];
    }
    layer1_bwd<0>(CW(p.W1p), dh, dx);

    // ---- head c: scaling_out = S[3:6]*sigmoid(sr[0:3]), rot = normalize(sr[3:7]); also the geometry xyz = anchor + offset*S[0:3]
    layer1<1>(CW(p.W1p), x, h);
#pragma unroll
    for (int i = 0; i < 32; i++) dh[i] = 0.0f;
    for (int j = 0; j < k; j++) {
        float g[7] = {0, 0, 0, 0, 0, 0, 0};
        if (mbits & (1u << j)) {
            const uint32_t row = row0 + __popc(mbits & ((1u << j) - 1u));
            float sr[7];
#pragma unroll
            for (int r = 0; r < 7; r++) sr[r] = dot32(CW(p.W2c) + (7 * j + r) * 32, h) + CW(p.b2c)[7 * j + r];
            const float* off = p.in.offset + ((size_t)a * k + j) * 3;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const float gx = og.xyz[(size_t)row * 3 + i];
                dA[i] += gx;
                d_offset[((size_t)a * k + j) * 3 + i] = gx * S[i];
                dS[i] = fmaf(gx, off[i], dS[i]);
                const float sg = sigmoid_(sr[i]);
                const float gs = og.scaling[(size_t)row * 3 + i];
                g[i] = gs * S[3 + i] * sg * (1.0f - sg);
                dS[3 + i] = fmaf(gs, sg, dS[3 + i]);
            }
            const float4 gr = reinterpret_cast<const float4*>(og.rot)[row];
            const float n = sqrtf(sr[3] * sr[3] + sr[4] * sr[4] + sr[5] * sr[5] + sr[6] * sr[6]);
            if (n > 1e-12f) {
                const float inv = 1.0f / n;
                const float r0 = sr[3] * inv, r1 = sr[4] * inv, r2 = sr[5] * inv, r3 = sr[6] * inv;
                const float dot = r0 * gr.x + r1 * gr.y + r2 * gr.z + r3 * gr.w;
                g[3] = (gr.x - r0 * dot) * inv; g[4] = (gr.y - r1 * dot) * inv; g[5] = (gr.z - r2 * dot) * inv; g[6] = (gr.w - r3 * dot) * inv;
            } else {
                g[3] = gr.x / 1e-12f; g[4] = gr.y / 1e-12f; g[5] = gr.z / 1e-12f; g[6] = gr.w / 1e-12f;
            }
        } else if (active) {
#pragma unroll
            for (int i = 0; i < 3; i++) d_offset[((size_t)a * k + j) * 3 + i] = 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 7; r++) {
            sc[(size_t)(SC_P2C + 7 * j + r) * ld + v] = g[r];
            axpy32(CW(p.W2c) + (7 * j + r) * 32, g[r], dh);
        }
    }
#pragma unroll
    for (int i = 0; i < 32; i++) {
        dh[i] = h[i] > 0.0f ? dh[i] : 0.0f;
        sc[(size_t)(SC_H + 32 + i) * ld + v] = h[i] * am;
        sc[(size_t)(SC_P1 + 32 + i) * ld + v] = dh[i];
    }
    layer1_bwd<1>(CW(p.W1p), dh, dx);

    // ---- head k: colour = sigmoid(.)
    layer1<2>(CW(p.W1p), x, h);
#pragma unroll
    for (int i = 0; i < 32; i++) dh[i] = 0.0f;
    for (int j = 0; j < k; j++) {
        float g[3] = {0, 0, 0};
        if (mbits & (1u << j)) {
            const uint32_t row = row0 + __popc(mbits & ((1u << j) - 1u));
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const float c = sigmoid_(dot32(CW(p.W2k) + (3 * j + r) * 32, h) + CW(p.b2k)[3 * j + r]);
                g[r] = og.color[(size_t)row * 3 + r] * c * (1.0f - c);
            }
        }
#pragma unroll
        for (int r = 0; r < 3; r++) {
            sc[(size_t)(SC_P2K + 3 * j + r) * ld + v] = g[r];
            axpy32(CW(p.W2k) + (3 * j + r) * 32, g[r], dh);
        }
    }
#pragma unroll
    for (int i = 0; i < 32; i++) {
        dh[i] = h[i] > 0.0f ? dh[i] : 0.0f;
        sc[(size_t)(SC_H + 64 + i) * ld + v] = h[i] * am;
        sc[(size_t)(SC_P1 + 64 + i) * ld + v] = dh[i];
    }
    layer1_bwd<2>(CW(p.W1p), dh, dx);

    if (active) {
        float4* f4 = reinterpret_cast<float4*>(d_feat + (size_t)a * GSD_FEAT);
#pragma unroll
        for (int q = 0; q < 8; q++) f4[q] = make_float4(dx[4 * q], dx[4 * q + 1], dx[4 * q + 2], dx[4 * q + 3]);
        const float vd = vw[0] * dx[32] + vw[1] * dx[33] + vw[2] * dx[34];
#pragma unroll
        for (int i = 0; i < 3; i++) d_anchor[3 * a + i] = dA[i] + (dx[32 + i] - vw[i] * vd) / dist + dx[35] * vw[i];
#pragma unroll
        for (int i = 0; i < 6; i++) d_scaling[(size_t)a * 6 + i] = dS[i];
    }
}

// ------------------------------------------------------------------------------------------------ backward, weight gradients (MFMA)
// One problem = one 16-column group of gradient columns (A, rows of dW) against NB 16-column groups of activation columns (B).
struct WgProblem { int a_col; int b_col; int nb; };
struct WgArgs {
    const float* sc; size_t ld; int n_chunks; int n_prob;
    float* part;            // [n_prob][WG_WAVES][3][256] tile partials
    float* bias_part;       // [n_prob][WG_WAVES][16]
    WgProblem prob[WG_MAXPROB];
};
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(64) k_wgrad(WgArgs p)
{
    const WgProblem pr = p.prob[blockIdx.y];
    const int lane = threadIdx.x, i = lane & 15, kk = lane >> 4;
    f32x4 acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; nb++) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.0f;
    const float* colA = p.sc + (size_t)(pr.a_col + i) * p.ld + 16 * kk;
    const float* colB = p.sc + (size_t)(pr.b_col + i) * p.ld + 16 * kk;
    for (int c = blockIdx.x; c < p.n_chunks; c += gridDim.x) {
        const size_t o = (size_t)c * 64;
        float4 a4[4];
#pragma unroll
        for (int q = 0; q < 4; q++) a4[q] = reinterpret_cast<const float4*>(colA + o)[q];
#pragma unroll
        for (int q = 0; q < 4; q++) bsum += (a4[q].x + a4[q].y) + (a4[q].z + a4[q].w);
#pragma unroll
        for (int nb = 0; nb < 3; nb++) {
            if (nb < pr.nb) {
                float4 b4[4];
#pragma unroll
                for (int q = 0; q < 4; q++) b4[q] = reinterpret_cast<const float4*>(colB + (size_t)(16 * nb) * p.ld + o)[q];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q].x, b4[q].x, acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q].y, b4[q].y, acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q].z, b4[q].z, acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q].w, b4[q].w, acc[nb], 0, 0, 0);
                }
            }
        }
    }
    // D layout: lane holds D[m = 4*(lane/16) + r][n = lane%16]
    float* out = p.part + ((size_t)blockIdx.y * WG_WAVES + blockIdx.x) * 3 * 256;
#pragma unroll
    for (int nb = 0; nb < 3; nb++)
#pragma unroll
        for (int r = 0; r < 4; r++) out[nb * 256 + (4 * kk + r) * 16 + i] = acc[nb][r];
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (kk == 0) p.bias_part[((size_t)blockIdx.y * WG_WAVES + blockIdx.x) * 16 + i] = bsum;
}

// sum the partial tiles over the persistent waves and scatter into the nn.Linear-shaped gradients
struct WgOut {
    gsd_cfg cfg;
    gsd_params g;           // destinations
    const float* part; const float* bias_part; int n_waves;
    // per problem: head (0 o, 1 c, 2 k), layer (1 or 2), first output row of the 16-row group
    int head[WG_MAXPROB], layer[WG_MAXPROB], row0[WG_MAXPROB], nb[WG_MAXPROB];
};
__global__ void __launch_bounds__(256) k_wgrad_reduce(WgOut p)
{
    const int pi = blockIdx.x, nbsel = blockIdx.y;
    if (nbsel > p.nb[pi]) return;                       // nbsel == nb  -> the bias vector
    const int head = p.head[pi], layer = p.layer[pi];
    const int k = p.cfg.k, lv = p.cfg.level ? 1 : 0;
    const int dist = head == 0 ? p.cfg.dist_o : (head == 1 ? p.cfg.dist_c : p.cfg.dist_k);
    const int out_dim = layer == 1 ? 32 : (head == 0 ? k : (head == 1 ? 7 * k : 3 * k));
    const int in1 = 35 + (dist ? 1 : 0) + lv + (head == 2 ? p.cfg.A : 0);
    float* W = layer == 1 ? (head == 0 ? p.g.W1o : head == 1 ? p.g.W1c : p.g.W1k) : (head == 0 ? p.g.W2o : head == 1 ? p.g.W2c : p.g.W2k);
    float* B = layer == 1 ? (head == 0 ? p.g.b1o : head == 1 ? p.g.b1c : p.g.b1k) : (head == 0 ? p.g.b2o : head == 1 ? p.g.b2c : p.g.b2k);
    if (nbsel == p.nb[pi]) {
        if (threadIdx.x < 16) {
            float s = 0.0f;
            for (int w = 0; w < p.n_waves; w++) s += p.bias_part[((size_t)pi * WG_WAVES + w) * 16 + threadIdx.x];
            const int row = p.row0[pi] + threadIdx.x;
            if (row < out_dim) B[row] = s;
        }
        return;
    }
    const int m = threadIdx.x >> 4, n = threadIdx.x & 15;
    float s = 0.0f;
    for (int w = 0; w < p.n_waves; w++) s += p.part[(((size_t)pi * WG_WAVES + w) * 3 + nbsel) * 256 + threadIdx.x];
    const int row = p.row0[pi] + m, col = 16 * nbsel + n;
    if (row >= out_dim) return;
    if (layer == 2) { W[row * 32 + col] = s; return; }
    // layer 1: scratch column -> nn.Linear input column
    int dst = -1;
    if (col < 35) dst = col;
    else if (col == 35) dst = dist ? 35 : -1;
    else if (col == 36) dst = lv ? 35 + (dist ? 1 : 0) : -1;
    if (dst >= 0) W[row * in1 + dst] = s;
}

// appearance: dW1k[:, app cols] = db1k (x) app;  dapp = W1k[:, app cols]^T db1k
struct AppArgs { gsd_cfg cfg; const float* W1k; const float* app; float* gW1k; const float* gb1k; float* gapp; };
__global__ void __launch_bounds__(64) k_app_grads(AppArgs p)
{
    const int A = p.cfg.A, lv = p.cfg.level ? 1 : 0, base = 35 + (p.cfg.dist_k ? 1 : 0) + lv, in = base + A;
    const int i = threadIdx.x;
    if (i >= A) return;
    float s = 0.0f;
    for (int j = 0; j < 32; j++) {
        const float g = p.gb1k[j];
        p.gW1k[j * in + base + i] = g * p.app[i];
        s = fmaf(p.W1k[j * in + base + i], g, s);
    }
    p.gapp[i] = s;
}

// ------------------------------------------------------------------------------------------------ host side
static int check_cfg(const gsd_cfg* c, const gsd_inputs* in, const gsd_params* p)
{
    if (!c || !in || !p) { gsr_set_error("gsd: null cfg/inputs/params"); return 1; }
    if (c->k < 1 || c->k > GSD_MAX_K) { gsr_set_error("gsd: n_offsets k=%d out of range [1,%d]", c->k, GSD_MAX_K); return 1; }
    if (c->A < 0 || c->A > GSD_MAX_APP) { gsr_set_error("gsd: appearance_dim A=%d out of range [0,%d]", c->A, GSD_MAX_APP); return 1; }
    if (c->Na < 0 || c->Nv < 0 || c->Nv > c->Na) { gsr_set_error("gsd: bad sizes Na=%d Nv=%d", c->Na, c->Nv); return 1; }
    if (c->Nv && (!in->anchor || !in->feat || !in->offset || !in->scaling || !in->vis_idx || !in->campos)) {
        gsr_set_error("gsd: anchor/feat/offset/scaling/vis_idx/campos must be provided"); return 1;
    }
    if (c->level && !in->level) { gsr_set_error("gsd: add_level set but level is NULL"); return 1; }
    if (!p->W1o || !p->b1o || !p->W2o || !p->b2o || !p->W1c || !p->b1c || !p->W2c || !p->b2c || !p->W1k || !p->b1k || !p->W2k || !p->b2k ||
        (c->A && !p->app)) { gsr_set_error("gsd: missing MLP parameter pointer"); return 1; }
    if (((uintptr_t)in->feat | (uintptr_t)p->W2o | (uintptr_t)p->W2c | (uintptr_t)p->W2k) & 15) {
        gsr_set_error("gsd: feat and the layer-2 weights must be 16-byte aligned"); return 1;
    }
    return 0;
}

static DecArgs make_args(const gsd_cfg* c, const gsd_inputs* in, const gsd_params* p, const float* W1p)
{
    DecArgs a;
    a.cfg = *c; a.in = *in; a.W1p = W1p;
    a.W2o = p->W2o; a.b2o = p->b2o; a.W2c = p->W2c; a.b2c = p->b2c; a.W2k = p->W2k; a.b2k = p->b2k;
    return a;
}

static void launch_pack(const gsd_cfg* c, const gsd_params* p, float* W1p, hipStream_t s)
{
    PackArgs pa;
    pa.cfg = *c; pa.W1o = p->W1o; pa.b1o = p->b1o; pa.W1c = p->W1c; pa.b1c = p->b1c; pa.W1k = p->W1k; pa.b1k = p->b1k; pa.app = p->app;
    pa.W1p = W1p;
    hipLaunchKernelGGL(k_pack, dim3(1), dim3(256), 0, s, pa);
}

// forward scratch: [W1p][total word (256 B)][scan block sums]
static size_t fwd_sums_off() { return gsr_align(GSD_W1P_FLOATS * sizeof(float)) + 256; }
extern "C" size_t gsd_forward_scratch_bytes(int32_t Nv)
{
    return fwd_sums_off() + gsr_align(((size_t)gsr_div_up((uint32_t)(Nv > 0 ? Nv : 1), 1024u) + 1) * sizeof(uint32_t));
}
extern "C" size_t gsd_compact_scratch_bytes(int32_t Na)
{
    const size_t n = (size_t)(Na > 0 ? Na : 1);
    return 256 + gsr_align(n * sizeof(uint32_t)) + gsr_align(((n + 1023) / 1024 + 1) * sizeof(uint32_t));
}

extern "C" int gsd_compact_visible(const uint8_t* mask, int32_t Na, int32_t* vis_idx, uint32_t* count_host, void* scratch,
                                   size_t scratch_bytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (!count_host) { gsr_set_error("gsd_compact_visible: count_host is NULL"); return 1; }
    *count_host = 0;
    if (Na <= 0) return 0;
    if (!mask || !vis_idx || !scratch || scratch_bytes < gsd_compact_scratch_bytes(Na)) {
        gsr_set_error("gsd_compact_visible: null pointer or scratch too small"); return 1;
    }
    uint32_t* total = (uint32_t*)scratch;
    uint32_t* pos = (uint32_t*)((char*)scratch + 256);
    uint32_t* sums = (uint32_t*)((char*)pos + gsr_align((size_t)Na * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_flags, dim3(gsr_div_up(Na, 256)), dim3(256), 0, s, mask, (uint32_t)Na, pos);
    launch_scan(pos, (uint32_t)Na, sums, total, s);
    hipLaunchKernelGGL(k_scatter_idx, dim3(gsr_div_up(Na, 256)), dim3(256), 0, s, mask, (uint32_t)Na, pos, vis_idx);
    GSR_CHECK(hipMemcpyAsync(count_host, total, sizeof(uint32_t), hipMemcpyDeviceToHost, s), "gsd_compact_visible: copy");
    GSR_CHECK(hipStreamSynchronize(s), "gsd_compact_visible: sync");
    return gsr_check_launch("gsd_compact_visible", s, false);
}

extern "C" int gsd_forward_stage1(const gsd_cfg* cfg, const gsd_inputs* in, const gsd_params* p, float* neural_opacity, uint8_t* mask,
                                  uint32_t* row_offset, uint32_t* P_host, void* scratch, size_t scratch_bytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (check_cfg(cfg, in, p)) return 1;
    if (!P_host) { gsr_set_error("gsd_forward_stage1: P_host is NULL"); return 1; }
    *P_host = 0;
    if (!scratch || scratch_bytes < gsd_forward_scratch_bytes(cfg->Nv)) { gsr_set_error("gsd_forward_stage1: scratch too small"); return 1; }
    float* W1p = (float*)scratch;
    launch_pack(cfg, p, W1p, s);
    if (cfg->Nv == 0) return gsr_check_launch("gsd_forward_stage1", s, false);
    if (!neural_opacity || !mask || !row_offset) { gsr_set_error("gsd_forward_stage1: null output"); return 1; }
    uint32_t* total = (uint32_t*)((char*)scratch + gsr_align(GSD_W1P_FLOATS * sizeof(float)));
    uint32_t* sums = (uint32_t*)((char*)scratch + fwd_sums_off());
    DecArgs a = make_args(cfg, in, p, W1p);
    hipLaunchKernelGGL(k_decode_opacity, dim3(gsr_div_up(cfg->Nv, GSD_BLOCK)), dim3(GSD_BLOCK), 0, s, a, neural_opacity, mask, row_offset);
    launch_scan(row_offset, (uint32_t)cfg->Nv, sums, total, s);
    GSR_CHECK(hipMemcpyAsync(P_host, total, sizeof(uint32_t), hipMemcpyDeviceToHost, s), "gsd_forward_stage1: copy");
    GSR_CHECK(hipStreamSynchronize(s), "gsd_forward_stage1: sync");
    return gsr_check_launch("gsd_forward_stage1", s, false);
}

extern "C" int gsd_forward_stage2(const gsd_cfg* cfg, const gsd_inputs* in, const gsd_params* p, const float* neural_opacity,
                                  const uint32_t* row_offset, uint32_t P, const gsd_outputs* out, void* scratch, size_t scratch_bytes,
                                  void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (check_cfg(cfg, in, p)) return 1;
    if (cfg->Nv == 0 || P == 0) return 0;
    if (!scratch || scratch_bytes < gsd_forward_scratch_bytes(cfg->Nv)) { gsr_set_error("gsd_forward_stage2: scratch too small"); return 1; }
    if (!out || !out->xyz || !out->color || !out->opacity || !out->scaling || !out->rot || !neural_opacity || !row_offset) {
        gsr_set_error("gsd_forward_stage2: null output/input"); return 1;
    }
    if ((uintptr_t)out->rot & 15) { gsr_set_error("gsd_forward_stage2: rot must be 16-byte aligned"); return 1; }
    DecArgs a = make_args(cfg, in, p, (const float*)scratch);       // W1p packed by stage 1 into the same scratch
    hipLaunchKernelGGL(k_decode_emit, dim3(gsr_div_up(cfg->Nv, GSD_BLOCK)), dim3(GSD_BLOCK), 0, s, a, neural_opacity, row_offset, *out);
    return gsr_check_launch("gsd_forward_stage2", s, false);
}

// backward scratch: [W1p][feature-major columns SC_COLS x ld][tile partials][bias partials]
static size_t bwd_ld(const gsd_cfg* c) { return (size_t)gsr_div_up((uint32_t)(c->Nv > 0 ? c->Nv : 1), GSD_BLOCK) * GSD_BLOCK; }
extern "C" size_t gsd_backward_scratch_bytes(const gsd_cfg* cfg)
{
    if (!cfg) return 0;
    return gsr_align(GSD_W1P_FLOATS * sizeof(float)) + gsr_align((size_t)SC_COLS * bwd_ld(cfg) * sizeof(float)) +
           gsr_align((size_t)WG_MAXPROB * WG_WAVES * 3 * 256 * sizeof(float)) + gsr_align((size_t)WG_MAXPROB * WG_WAVES * 16 * sizeof(float));
}

extern "C" int gsd_backward(const gsd_cfg* cfg, const gsd_inputs* in, const gsd_params* p, const float* neural_opacity,
                            const uint32_t* row_offset, uint32_t P, const gsd_out_grads* og, const gsd_in_grads* ig, void* scratch,
                            size_t scratch_bytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (check_cfg(cfg, in, p)) return 1;
    if (!ig) { gsr_set_error("gsd_backward: null in_grads"); return 1; }
    const gsd_params& g = ig->params;
    if (!g.W1o || !g.b1o || !g.W2o || !g.b2o || !g.W1c || !g.b1c || !g.W2c || !g.b2c || !g.W1k || !g.b1k || !g.W2k || !g.b2k || (cfg->A && !g.app)) {
        gsr_set_error("gsd_backward: missing parameter-gradient pointer"); return 1;
    }
    if (!scratch || scratch_bytes < gsd_backward_scratch_bytes(cfg)) { gsr_set_error("gsd_backward: scratch too small"); return 1; }
    const int k = cfg->k;
    if (cfg->Nv == 0) {       // no anchors: all parameter gradients are zero
        const int lv = cfg->level ? 1 : 0;
        (void)hipMemsetAsync(g.W1o, 0, sizeof(float) * 32 * (35 + (cfg->dist_o ? 1 : 0) + lv), s); (void)hipMemsetAsync(g.b1o, 0, sizeof(float) * 32, s);
        (void)hipMemsetAsync(g.W1c, 0, sizeof(float) * 32 * (35 + (cfg->dist_c ? 1 : 0) + lv), s); (void)hipMemsetAsync(g.b1c, 0, sizeof(float) * 32, s);
        (void)hipMemsetAsync(g.W1k, 0, sizeof(float) * 32 * (35 + (cfg->dist_k ? 1 : 0) + lv + cfg->A), s); (void)hipMemsetAsync(g.b1k, 0, sizeof(float) * 32, s);
        (void)hipMemsetAsync(g.W2o, 0, sizeof(float) * 32 * k, s); (void)hipMemsetAsync(g.b2o, 0, sizeof(float) * k, s);
        (void)hipMemsetAsync(g.W2c, 0, sizeof(float) * 32 * 7 * k, s); (void)hipMemsetAsync(g.b2c, 0, sizeof(float) * 7 * k, s);
        (void)hipMemsetAsync(g.W2k, 0, sizeof(float) * 32 * 3 * k, s); (void)hipMemsetAsync(g.b2k, 0, sizeof(float) * 3 * k, s);
        if (cfg->A) (void)hipMemsetAsync(g.app, 0, sizeof(float) * cfg->A, s);
        return gsr_check_launch("gsd_backward", s, false);
    }
    if (!og || !og->xyz || !og->color || !og->opacity || !og->scaling || !og->rot || !ig->anchor || !ig->feat || !ig->offset || !ig->scaling ||
        !neural_opacity || !row_offset) { gsr_set_error("gsd_backward: null gradient pointer"); return 1; }
    if (((uintptr_t)og->rot | (uintptr_t)ig->feat) & 15) { gsr_set_error("gsd_backward: dL_drot / d_feat must be 16-byte aligned"); return 1; }
    (void)P;
    const size_t ld = bwd_ld(cfg);
    float* W1p = (float*)scratch;
    float* sc = (float*)((char*)scratch + gsr_align(GSD_W1P_FLOATS * sizeof(float)));
    float* part = (float*)((char*)sc + gsr_align((size_t)SC_COLS * ld * sizeof(float)));
    float* bias_part = (float*)((char*)part + gsr_align((size_t)WG_MAXPROB * WG_WAVES * 3 * 256 * sizeof(float)));
    launch_pack(cfg, p, W1p, s);
    DecArgs a = make_args(cfg, in, p, W1p);
    hipLaunchKernelGGL(k_decode_bwd, dim3(ld / GSD_BLOCK), dim3(GSD_BLOCK), 0, s, a, neural_opacity, row_offset, *og, ig->anchor, ig->feat,
                       ig->offset, ig->scaling, sc, ld);

    // weight-gradient problems
    WgArgs wa; WgOut wo;
    wa.sc = sc; wa.ld = ld; wa.n_chunks = (int)(ld / 64); wa.part = part; wa.bias_part = bias_part;
    wo.cfg = *cfg; wo.g = g; wo.part = part; wo.bias_part = bias_part;
    int np = 0;
    auto add = [&](int a_col, int b_col, int nb, int head, int layer, int row0) {
        wa.prob[np] = {a_col, b_col, nb}; wo.head[np] = head; wo.layer[np] = layer; wo.row0[np] = row0; wo.nb[np] = nb; np++;
    };
    const int p2col[3] = {SC_P2O, SC_P2C, SC_P2K};
    const int outd[3] = {k, 7 * k, 3 * k};
    for (int head = 0; head < 3; head++) {
        for (int r0 = 0; r0 < outd[head]; r0 += 16) add(p2col[head] + r0, SC_H + 32 * head, 2, head, 2, r0);
        for (int r0 = 0; r0 < 32; r0 += 16) add(SC_P1 + 32 * head + r0, SC_X, 3, head, 1, r0);
    }
    wa.n_prob = np;
    const int waves = wa.n_chunks < WG_WAVES ? wa.n_chunks : WG_WAVES;
    wo.n_waves = waves;
    hipLaunchKernelGGL(k_wgrad, dim3(waves, np), dim3(64), 0, s, wa);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(np, 4), dim3(256), 0, s, wo);
    if (cfg->A) {
        AppArgs aa; aa.cfg = *cfg; aa.W1k = p->W1k; aa.app = p->app; aa.gW1k = g.W1k; aa.gb1k = g.b1k; aa.gapp = g.app;
        hipLaunchKernelGGL(k_app_grads, dim3(1), dim3(64), 0, s, aa);
    }
    return gsr_check_launch("gsd_backward", s, false);
}
