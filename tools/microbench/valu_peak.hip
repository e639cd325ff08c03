// Calibrates the non-packed fp32 VALU issue ceiling of the device: every wave runs `iters` x 16 independent v_fma_f32.
// Used by bench.py's informational `roofline.valu_issue` figure (the microarchitecture guide gives flop peaks, not an issue rate).
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int MODE>
__global__ void __launch_bounds__(256) k_valu(float* out, int iters, float a, float b)
{
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (float)(threadIdx.x + i);
    if (MODE == 3 && (threadIdx.x & 63) >= 16) return;              // only the first quarter-wave stays active: does the SIMD skip empty passes?
    if (MODE == 4 && (threadIdx.x & 63) >= 32) return;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0 || MODE >= 3) x[i] = __builtin_fmaf(x[i], a, b);                          // v_fma_f32 / v_fmac_f32
            else if (MODE == 1) x[i] = x[i] * a;                                        // v_mul_f32
            else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a));    // v_cndmask_b32
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

extern "C" int valu_peak_run(int mode, int blocks, int iters, void* out, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(256), 0, s, (float*)out, iters, 1.0000001f, 1e-9f);
    else if (mode == 1) hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(256), 0, s, (float*)out, iters, 1.0000001f, 0.f);
    else if (mode == 3) hipLaunchKernelGGL(k_valu<3>, dim3(blocks), dim3(256), 0, s, (float*)out, iters, 1.0000001f, 1e-9f);
    else if (mode == 4) hipLaunchKernelGGL(k_valu<4>, dim3(blocks), dim3(256), 0, s, (float*)out, iters, 1.0000001f, 1e-9f);
    else hipLaunchKernelGGL(k_valu<2>, dim3(blocks), dim3(256), 0, s, (float*)out, iters, 1.0000001f, 0.f);
    return (int)hipGetLastError();
}
