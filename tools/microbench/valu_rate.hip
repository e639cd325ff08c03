// Issue rate of plain and packed f32 VALU instructions on gfx950: 16 independent accumulators per lane, 8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 2048
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, float s) {
	float a[16];
	f2 p[8];
	for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 1e-3f + i;
	for (int i = 0; i < 8; i++) p[i] = f2{a[2 * i], a[2 * i + 1]};
	f2 s2 = {s, s * 0.5f};
	for (int r = 0; r < REP; r++) {
		if (MODE == 0) {
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
		} else if (MODE == 1) {
#pragma unroll
			for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(s2));
		} else if (MODE == 2) {
#pragma unroll
			for (int i = 0; i < 8; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
		} else if (MODE == 3) {
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
		} else if (MODE == 4) {     // packed with a broadcast scalar operand (op_sel_hi 0 on src1): acc pair += x pair * w
#pragma unroll
			for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(s2));
		} else if (MODE == 5) {     // fmac DPP broadcast form used by the splat-parallel kernel
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_fmac_f32_dpp %0, %1, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(s));
		} else if (MODE == 6) {     // three distinct VGPR sources
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(s), "v"(s2.y));
		} else if (MODE == 7) {     // VOP2 fmac, two distinct sources
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(s2.y));
		} else if (MODE == 8) {     // row_shr scan step
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
		} else if (MODE == 9) {     // SGPR operand
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(s), "v"(s2.y));
		} else if (MODE == 10) {    // quad_perm dpp
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_fmac_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(s));
		} else if (MODE == 11) {    // mov dpp only
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(s));
		} else if (MODE == 12) {    // v_readlane (to SGPR) + plain fmac with it: the alternative to a fused DPP broadcast
#pragma unroll
			for (int i = 0; i < 16; i++) { float u; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(u) : "v"(a[(i + 1) & 15])); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(u), "v"(s2.y)); }
		} else if (MODE == 13) {    // ds_swizzle-free: v_exp (transcendental)
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
		} else if (MODE == 14) {    // v_rcp
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
		} else if (MODE == 15) {    // v_cndmask
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s));
		} else if (MODE == 16) {    // v_cmp + cndmask pair
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s) : "vcc");
		} else if (MODE == 17) {    // packed with a 64-bit SGPR pair source
#pragma unroll
			for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "s"(s2));
		} else if (MODE == 18) {    // packed, SGPR pair, low half broadcast to both lanes of the pair
#pragma unroll
			for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "s"(s2));
		} else if (MODE == 19) {    // packed add with SGPR pair
#pragma unroll
			for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "s"(s2));
		} else if (MODE == 20) {    // v_sub with sgpr (VOP2, SGPR in src0)
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
		} else if (MODE == 21) {    // inline constant operand
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_fmac_f32 %0, 0.5, %1" : "+v"(a[i]) : "v"(s));
		} else if (MODE == 22) {    // 4 waves/SIMD-independent: v_max (no fp contraction path)
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
		}
	}
	float t = 0;
	for (int i = 0; i < 16; i++) t += a[i];
	for (int i = 0; i < 8; i++) t += p[i].x + p[i].y;
	out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int MODE> void run(const char* name, int per_iter) {
	float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<MODE><<<256 * 8, 256>>>(d, 1.0001f);
	hipEventRecord(e0);
	for (int i = 0; i < 5; i++) k<MODE><<<256 * 8, 256>>>(d, 1.0001f);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
	double winstr = 256.0 * 8 * 4 * REP * per_iter;                     // wave-instructions
	double simd_cycles = ms * 1e-3 * 2.4e9 * 1024;
	printf("%-34s %.3f ms  %.1f G wave-instr/s  %.2f cycles/instr/SIMD @2.4GHz\n", name, ms, winstr / ms * 1e-6, simd_cycles / winstr);
	hipFree(d);
}
int main() {
	run<0>("v_fma_f32", 16); run<3>("v_mul_f32", 16); run<1>("v_pk_fma_f32", 8); run<2>("v_pk_mul_f32", 8); run<4>("v_pk_fma_f32 op_sel_hi bcast", 8);
	run<5>("v_fmac_f32_dpp row_newbcast", 16);
	run<6>("v_fma_f32 3 distinct srcs", 16); run<7>("v_fmac_f32 (VOP2)", 16); run<8>("v_mul_f32_dpp row_shr:1", 16); run<9>("v_fmac_f32 sgpr src", 16);
	run<10>("v_fmac_f32_dpp quad_perm", 16); run<11>("v_mov_b32_dpp row_newbcast", 16); run<12>("v_readlane + v_fmac sgpr (2 instr)", 32);
	run<17>("v_pk_fma_f32 sgpr pair", 8); run<18>("v_pk_fma_f32 sgpr pair op_sel_hi", 8); run<19>("v_pk_add_f32 sgpr pair", 8);
	run<20>("v_sub_f32 sgpr src0", 16); run<21>("v_fmac_f32 inline const", 16); run<22>("v_max_f32", 16);
	run<13>("v_exp_f32", 16); run<14>("v_rcp_f32", 16); run<15>("v_cndmask_b32", 16); run<16>("v_cmp + v_cndmask (2 instr)", 32);
	return 0;
}
