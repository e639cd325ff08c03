// hbm_granule -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for SPARSE 16-byte accesses (measurement helper, not product code).
// MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced streams (reports half the bytes) and says other widths and WRITE_SIZE are uncalibrated;
// the sparse-TSDF voxel pass reads and writes 16-byte groups of which only a fraction per 64-byte sector / 128-byte line is touched.  Every kernel here makes
// N accesses of 16 bytes, one per STRIDE bytes, over a buffer far larger than the 256 MiB Infinity Cache, so counter / N = bytes moved per access.
//   rd<S>: load   wr<S>: store   rmw<S>: load + store of the same 16 bytes     S in {16, 32, 64, 128, 256}
//   wrw<B,S> / rmww<B,S>: the same with B = 32 / 64 / 128 consecutive bytes per access
// usage: hbm_granule            (prints N and the useful bytes; run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, tools/hbm_granule_report.py)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int S> __global__ void __launch_bounds__(256) rd(const char* __restrict__ p, float4* __restrict__ sink, size_t n)
{
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = *reinterpret_cast<const float4*>(p + i * S);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x == 12345.678f) sink[0] = acc;      // never true: keeps the loads
}
template <int S> __global__ void __launch_bounds__(256) wr(char* __restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        *reinterpret_cast<float4*>(p + i * S) = make_float4(1.f, 2.f, 3.f, (float)i);
}
template <int S> __global__ void __launch_bounds__(256) rmw(char* __restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = *reinterpret_cast<float4*>(p + i * S);
        v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
        *reinterpret_cast<float4*>(p + i * S) = v;
    }
}
// B consecutive bytes per access (32 or 64), one access per S bytes: which unit does the memory side write without a read-modify-write?
template <int B, int S> __global__ void __launch_bounds__(256) wrw(char* __restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
#pragma unroll
        for (int k = 0; k < B / 16; k++) *reinterpret_cast<float4*>(p + i * S + 16 * k) = make_float4(1.f, 2.f, 3.f, (float)i);
}
template <int B, int S> __global__ void __launch_bounds__(256) rmww(char* __restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v[B / 16];
#pragma unroll
        for (int k = 0; k < B / 16; k++) v[k] = *reinterpret_cast<float4*>(p + i * S + 16 * k);
#pragma unroll
        for (int k = 0; k < B / 16; k++) { v[k].x += 1.f; v[k].w += 1.f; *reinterpret_cast<float4*>(p + i * S + 16 * k) = v[k]; }
    }
}
template <int B, int S> static int runw(char* buf, size_t n)
{
    for (int k = 0; k < 3; k++) {
        hipLaunchKernelGGL((wrw<B, S>), dim3(4096), dim3(256), 0, 0, buf, n);
        hipLaunchKernelGGL((rmww<B, S>), dim3(4096), dim3(256), 0, 0, buf, n);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
template <int S> static int run(char* buf, float4* sink, size_t n)
{
    for (int k = 0; k < 3; k++) {
        hipLaunchKernelGGL(rd<S>, dim3(4096), dim3(256), 0, 0, (const char*)buf, sink, n);
        hipLaunchKernelGGL(wr<S>, dim3(4096), dim3(256), 0, 0, buf, n);
        hipLaunchKernelGGL(rmw<S>, dim3(4096), dim3(256), 0, 0, buf, n);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
int main()
{
    const size_t n = (size_t)16 << 20;            // 16 Mi accesses of 16 B = 256 MiB useful per kernel
    const size_t bytes = n * 256;                  // 4 GiB: every stride walks past the Infinity Cache
    char* buf; float4* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, bytes));
    if (run<16>(buf, sink, n) || run<32>(buf, sink, n) || run<64>(buf, sink, n) || run<128>(buf, sink, n) || run<256>(buf, sink, n)) return 1;
    if (runw<32, 64>(buf, n) || runw<32, 128>(buf, n) || runw<64, 128>(buf, n) || runw<64, 256>(buf, n) || runw<128, 256>(buf, n)) return 1;
    printf("{\"accesses\": %zu, \"useful_bytes_per_kernel\": %zu}\n", n, n * 16);
    return 0;
}
