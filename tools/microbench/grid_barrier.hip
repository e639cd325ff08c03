// grid_barrier.hip -- what does a grid-wide barrier inside ONE persistent kernel cost on MI355X (8 XCDs, one L2 each), against the
// ~4.5 us floor of a dependent kernel launch?  Standalone: hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier && ./grid_barrier
//
// A launch of G workgroups x 256 threads runs NB rounds; in every round each workgroup writes `words` dwords of a buffer shared by the grid,
// passes a barrier, and reads the words its neighbour wrote (checksummed, so the traffic and the cross-workgroup visibility are real).
// Barrier flavours:
//   0  agent-scope fences: __threadfence(); atomicAdd; spin on an atomic load; __threadfence()      -- the portable form
//      (on gfx950 the release is buffer_wbl2 sc1 = write back this XCD's dirty L2 lines, the acquire buffer_inv sc1)
//   1  no fences, data moved with system-scope (sc0 sc1) loads / stores that bypass the non-coherent caches; ordering by s_waitcnt only
//   2  relaxed counter only, data cached normally -- NOT correct across XCDs, the floor of the counter round trip itself
//   3  one kernel launch per round instead of a barrier (the status quo)
// Prints us per round.  (Result on the round-3 box: see DESIGN.md Appendix A-32.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__device__ __forceinline__ void grid_barrier(uint32_t* ctr, uint32_t target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 0) __threadfence();
        else if (MODE == 1) __builtin_amdgcn_s_waitcnt(0);           // all of this wave's stores have left (vmcnt / lgkmcnt = 0) -- the other waves passed __syncthreads
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        if (MODE == 0) __threadfence();
    }
    __syncthreads();
}

template <int MODE>
__device__ __forceinline__ void st(uint32_t* p, uint32_t v)
{
    if (MODE == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); else *p = v;
}
template <int MODE>
__device__ __forceinline__ uint32_t ld(const uint32_t* p)
{
    if (MODE == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return *(const volatile uint32_t*)p;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_rounds(uint32_t* buf, uint32_t words, uint32_t rounds, uint32_t* ctr, uint32_t* out, uint32_t round0)
{
    const uint32_t G = gridDim.x, b = blockIdx.x;
    uint32_t sum = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        uint32_t* mine = buf + ((size_t)((r + round0) & 1) * G + b) * words;
        for (uint32_t i = threadIdx.x; i < words; i += 256) st<MODE>(mine + i, (r + round0) * 0x9e3779b9u + b * 131u + i);
        if (MODE != 3) grid_barrier<MODE>(ctr, G * (r + 1));
        if (MODE != 3) {
            const uint32_t nb = (b + G / 2 + 1) % G;                  // a workgroup that (round robin) sits on another XCD
            const uint32_t* theirs = buf + ((size_t)((r + round0) & 1) * G + nb) * words;
            for (uint32_t i = threadIdx.x; i < words; i += 256) sum += ld<MODE>(theirs + i) ^ ((r + round0) * 0x9e3779b9u + nb * 131u + i);
        }
    }
    if (MODE == 3 && round0 > 0) {                                   // the launch-per-round variant reads what the PREVIOUS launch wrote
        const uint32_t nb = (b + G / 2 + 1) % G;
        const uint32_t* theirs = buf + ((size_t)((round0 - 1) & 1) * G + nb) * words;
        for (uint32_t i = threadIdx.x; i < words; i += 256) sum += theirs[i] ^ ((round0 - 1) * 0x9e3779b9u + nb * 131u + i);
    }
    if (sum) atomicAdd(out, 1u);                                     // every xor is 0 when the data was visible: out counts stale reads
}

template <int MODE>
static float run(int G, uint32_t words, uint32_t rounds, uint32_t* buf, uint32_t* ctr, uint32_t* out, bool coop)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipMemset(ctr, 0, 4)); CK(hipMemset(out, 0, 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (MODE == 3) {
            for (uint32_t r = 0; r < rounds + 1; r++) hipLaunchKernelGGL(k_rounds<3>, dim3(G), dim3(256), 0, 0, buf, words, 1u, ctr, out, r);
        } else if (coop) {
            uint32_t r0 = 0; void* args[] = { &buf, &words, &rounds, &ctr, &out, &r0 };
            CK(hipLaunchCooperativeKernel((const void*)k_rounds<MODE>, dim3(G), dim3(256), args, 0, 0));
        } else {
            hipLaunchKernelGGL(k_rounds<MODE>, dim3(G), dim3(256), 0, 0, buf, words, rounds, ctr, out, 0u);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        uint32_t stale; CK(hipMemcpy(&stale, out, 4, hipMemcpyDeviceToHost));
        if (stale && MODE != 2) printf("    MODE %d: %u workgroups saw stale data!\n", MODE, stale);
        if (ms < best) best = ms;
    }
    return best * 1e3f / rounds;
}

int main()
{
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_rounds<0>, 256, 0));
    printf("%s: %d CUs, cooperativeLaunch=%d, resident 256-thread workgroups per CU (this kernel) = %d\n", pr.name, pr.multiProcessorCount, pr.cooperativeLaunch, occ);
    const uint32_t rounds = 64;
    uint32_t *buf, *ctr, *out;
    CK(hipMalloc(&buf, 2u * 1024 * 8192 * 4)); CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&out, 4));
    for (int G : { 128, 256, 512 }) {
        for (uint32_t words : { 256u, 2048u, 8192u }) {          // 1 KB / 8 KB / 32 KB written per workgroup per round (G = 256: 0.25 / 2 / 8 MB)
            printf("G=%3d  %5.2f MB/round: fences %6.2f us  fences(coop launch) %6.2f us  sc0sc1 data, no fences %6.2f us  counter only %6.2f us  launch per round %6.2f us\n",
                   G, G * words * 4 / 1048576.0, run<0>(G, words, rounds, buf, ctr, out, false), run<0>(G, words, rounds, buf, ctr, out, true),
                   run<1>(G, words, rounds, buf, ctr, out, false), run<2>(G, words, rounds, buf, ctr, out, false), run<3>(G, words, rounds, buf, ctr, out, false));
        }
    }
    return 0;
}
