// scatter_acc.hip -- what does it cost to move per-(block, splat) partial gradients (16 floats) to a per-gaussian accumulator?
// Standalone (no torch): hipcc --offload-arch=gfx950 -O3 scatter_acc.hip -o scatter_acc && ./scatter_acc
//
// Emulates the write side of k_blend_bwd: NOPS "pair flushes", each adds 16 floats into acc[gid * 20 + c] with gid drawn from
// [0, P).  Each wave issues its ops back to back with a short dependent VALU chain in between (FILL fmas), so the figures are the
// memory-side cost, not issue cost.  Methods:
//   0  16-lane no-return float atomics (lanes 48..63), agent scope         -- what round 1 ships
//   1  the same with the sc1 bit (system scope encoding)
//   2  plain 16-lane dword stores to a private slot per op (scattered 64 B)
//   3  plain stores, slot index == op index in issue order (per wave contiguous: 64 B then the next 64 B ...)
//   4  lane = gaussian: 16 component-major atomic instructions, 64 different lines each (no LDS transpose)
//   5  16-lane atomics, gid restricted to the issuing block's XCD slice (block b -> slice b % 8)
//   6  LDS table per block (ds_add_f32), flushed once per 256 ops with 16-lane atomics of the NON-duplicate entries only
//      (dup = fraction of ops that hit an entry already present: models the 1.39 sub-blocks per tile instance)
//   7  workgroup-scope atomics (__hip_atomic_fetch_add, WORKGROUP) -- encoding check
//   8  no memory operation (fill chains only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int METHOD, int FILL>
__global__ void __launch_bounds__(256) k_scatter(float* acc, float* slots, uint32_t P, uint32_t ops_per_wave, uint32_t total_waves)
{
    const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= total_waves) return;
    float v = 1.0f + lane * 1e-3f;
    __shared__ float table[256 * 17];
    if (METHOD == 6) {
        for (int i = threadIdx.x; i < 256 * 17; i += 256) table[i] = 0.f;
        __syncthreads();
    }
    for (uint32_t k = 0; k < ops_per_wave; k++) {
#pragma unroll
        for (int f = 0; f < FILL; f++) v = fmaf(v, 1.0000001f, 1e-9f);
        const uint32_t op = wave * ops_per_wave + k;
        uint32_t gid = hash32(op * 2654435761u + 12345u) % P;
        if (METHOD == 5) { const uint32_t sl = P / 8; gid = (blockIdx.x & 7) * sl + gid % sl; }
        if (METHOD == 0) {
            if (lane >= 48) unsafeAtomicAdd(acc + (size_t)gid * 20 + (lane - 48), v);
        } else if (METHOD == 1) {
            if (lane >= 48) {
                float* p = acc + (size_t)gid * 20 + (lane - 48);
                asm volatile("global_atomic_add_f32 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
            }
        } else if (METHOD == 2) {
            const uint32_t slot = hash32(op ^ 0x9e3779b9u) % (total_waves * ops_per_wave);
            if (lane >= 48) slots[(size_t)slot * 16 + (lane - 48)] = v;
        } else if (METHOD == 3) {
            if (lane >= 48) slots[(size_t)op * 16 + (lane - 48)] = v;
        } else if (METHOD == 4) {
            if ((k & 63) == 63) {      // one flush per 64 ops: lane = gaussian
                const uint32_t g2 = hash32((op + lane) * 2654435761u + 777u) % P;
#pragma unroll
                for (int c = 0; c < 16; c++) unsafeAtomicAdd(acc + (size_t)g2 * 20 + c, v);
            }
        } else if (METHOD == 5) {
            if (lane >= 48) unsafeAtomicAdd(acc + (size_t)gid * 20 + (lane - 48), v);
        } else if (METHOD == 6) {
            // 28 % of the ops re-hit an existing entry of the block's table (entry index from the op number)
            const uint32_t e = (hash32(op) % 100u < 28u) ? (hash32(op + 1) & 255u) : ((k * 4 + (threadIdx.x >> 6)) & 255u);
            if (lane >= 48) __hip_atomic_fetch_add(&table[e * 17 + (lane - 48)], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((k & 63) == 63) {      // 4 waves x 64 ops = 256 table entries worth of ops: flush 72 % of 256 entries
                __syncthreads();
                for (int e2 = threadIdx.x >> 4; e2 < 184; e2 += 16) {
                    const uint32_t g2 = hash32((op + e2) * 2654435761u + 99u) % P;
                    const float t = table[e2 * 17 + (threadIdx.x & 15)];
                    unsafeAtomicAdd(acc + (size_t)g2 * 20 + (threadIdx.x & 15), t);
                    table[e2 * 17 + (threadIdx.x & 15)] = 0.f;
                }
                __syncthreads();
            }
        } else if (METHOD == 7) {
            if (lane >= 48) __hip_atomic_fetch_add(acc + (size_t)gid * 20 + (lane - 48), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (v == 123.456f) acc[0] = v;
}

template <int METHOD, int FILL>
static float run(float* acc, float* slots, uint32_t P, uint32_t ops_per_wave, uint32_t waves, size_t acc_bytes)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipMemsetAsync(acc, 0, acc_bytes, 0));
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k_scatter<METHOD, FILL>), dim3((waves + 3) / 4), dim3(256), 0, 0, acc, slots, P, ops_per_wave, waves);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

int main()
{
    const uint32_t P = 300000, waves = 8160 * 4, opw = 64;      // 2.09 M ops (round 1 measured 1.92 M wave-splat pairs)
    const size_t acc_bytes = (size_t)P * 20 * 4, slot_bytes = (size_t)waves * opw * 16 * 4;
    float *acc, *slots;
    CK(hipMalloc(&acc, acc_bytes)); CK(hipMalloc(&slots, slot_bytes));
    CK(hipMemset(slots, 0, slot_bytes));
    printf("ops %u (x16 floats), P %u\n", waves * opw, P);
#define RUN(M, F) printf("method %d fill %3d : %.4f ms\n", M, F, run<M, F>(acc, slots, P, opw, waves, acc_bytes));
    RUN(0, 0) RUN(0, 64) RUN(0, 200)
    RUN(1, 0) RUN(1, 200)
    RUN(2, 0) RUN(2, 200)
    RUN(3, 0) RUN(3, 200)
    RUN(4, 0) RUN(4, 200)
    RUN(5, 0) RUN(5, 200)
    RUN(6, 0) RUN(6, 200)
    RUN(7, 0) RUN(7, 200)
    RUN(8, 0) RUN(8, 64) RUN(8, 200)      // no memory operation at all: the cost of the fill chains alone
    return 0;
}
