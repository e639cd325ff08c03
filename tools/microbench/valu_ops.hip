// Per-opcode wave64 issue rates on gfx950 (16 independent registers per lane, asm so that the compiler cannot fuse or pack).
#include <hip/hip_runtime.h>
#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

template <int MODE>
__global__ void __launch_bounds__(256) k_ops(float* out, int iters, float a, float b)
{
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f + 1.0f;
    for (int it = 0; it < iters; ++it) {
#define OP_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
#define OP_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
#define OP_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(a));
#define OP_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
#define OP_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
#define OP_DPP(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
#define OP_CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc");
#define OP_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : "vcc");
#define OP_MIX_FE(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_exp_f32 %1, %1" : "+v"(x[i]), "+v"(x[(i + 8) & 15]) : "v"(a), "v"(b));
        if (MODE == 0) { REP16(OP_FMA) }
        else if (MODE == 1) { REP16(OP_ADD) }
        else if (MODE == 2) { REP16(OP_MOV) }
        else if (MODE == 3) { REP16(OP_EXP) }
        else if (MODE == 4) { REP16(OP_RCP) }
        else if (MODE == 5) { REP16(OP_DPP) }
        else if (MODE == 6) { REP16(OP_CMP) }
        else if (MODE == 7) { REP16(OP_CND) }
        else { OP_MIX_FE(0) OP_MIX_FE(1) OP_MIX_FE(2) OP_MIX_FE(3) OP_MIX_FE(4) OP_MIX_FE(5) OP_MIX_FE(6) OP_MIX_FE(7) }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

extern "C" int valu_ops_run(int mode, int blocks, int iters, void* out, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    float* o = (float*)out;
    switch (mode) {
    case 0: hipLaunchKernelGGL(k_ops<0>, dim3(blocks), dim3(256), 0, s, o, iters, 1.0000001f, 1e-9f); break;
    case 1: hipLaunchKernelGGL(k_ops<1>, dim3(blocks), dim3(256), 0, s, o, iters, 1.0000001f, 1e-9f); break;
    case 2: hipLaunchKernelGGL(k_ops<2>, dim3(blocks), dim3(256), 0, s, o, iters, 1.0000001f, 1e-9f); break;
    case 3: hipLaunchKernelGGL(k_ops<3>, dim3(blocks), dim3(256), 0, s, o, iters, 1.0000001f, 1e-9f); break;
    case 4: hipLaunchKernelGGL(k_ops<4>, dim3(blocks), dim3(256), 0, s, o, iters, 1.0000001f, 1e-9f); break;
    case 5: hipLaunchKernelGGL(k_ops<5>, dim3(blocks), dim3(256), 0, s, o, iters, 1.0000001f, 1e-9f); break;
    case 6: hipLaunchKernelGGL(k_ops<6>, dim3(blocks), dim3(256), 0, s, o, iters, 1.0000001f, 1e-9f); break;
    case 7: hipLaunchKernelGGL(k_ops<7>, dim3(blocks), dim3(256), 0, s, o, iters, 1.0000001f, 1e-9f); break;
    default: hipLaunchKernelGGL(k_ops<8>, dim3(blocks), dim3(256), 0, s, o, iters, 1.0000001f, 1e-9f); break;
    }
    return (int)hipGetLastError();
}
