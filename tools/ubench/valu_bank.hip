// Micro-benchmark: does v_bitop3_b32 (3 VGPR sources) pay for register-bank placement or for dependency depth on
// gfx950? Explicit registers in inline asm; 64 instructions per loop iteration.
//   same   : all three sources in one bank (v(4k)), 8 independent chains
//   spread : sources in three different banks, 8 independent chains
//   dep1/2/4: 1, 2, 4 independent dependency chains (sources in different banks)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return; } } while (0)
#define R8(S) S S S S S S S S
template <int OP> __global__ void k(u32 *out, int iters)
{
	u32 seed = threadIdx.x * 2654435761u + blockIdx.x;
	asm volatile("v_mov_b32 v0, %0\nv_mov_b32 v1, %0\nv_mov_b32 v2, %0\nv_mov_b32 v3, %0\nv_mov_b32 v4, %0\nv_mov_b32 v5, %0\nv_mov_b32 v6, %0\nv_mov_b32 v7, %0\n"
	             "v_mov_b32 v8, %0\nv_mov_b32 v9, %0\nv_mov_b32 v10, %0\nv_mov_b32 v11, %0\nv_mov_b32 v12, %0\nv_mov_b32 v13, %0\nv_mov_b32 v14, %0\nv_mov_b32 v15, %0\n"
	             "v_mov_b32 v16, %0\nv_mov_b32 v17, %0\nv_mov_b32 v18, %0\nv_mov_b32 v19, %0\nv_mov_b32 v20, %0\nv_mov_b32 v21, %0\nv_mov_b32 v22, %0\nv_mov_b32 v23, %0\n"
	             "v_mov_b32 v24, %0\nv_mov_b32 v25, %0\nv_mov_b32 v26, %0\nv_mov_b32 v27, %0\nv_mov_b32 v28, %0\nv_mov_b32 v29, %0\nv_mov_b32 v30, %0\nv_mov_b32 v31, %0\n"
	             :: "v"(seed) : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31");
	for (int it = 0; it < iters; it++)
	{
		if (OP == 0) // same bank: dst v(4k), sources v(4k), v(4j), v(4m)
			asm volatile(R8("v_bitop3_b32 v0, v0, v4, v8 bitop3:0x96\nv_bitop3_b32 v4, v4, v8, v12 bitop3:0x96\nv_bitop3_b32 v8, v8, v12, v16 bitop3:0x96\nv_bitop3_b32 v12, v12, v16, v20 bitop3:0x96\n"
			                "v_bitop3_b32 v16, v16, v20, v24 bitop3:0x96\nv_bitop3_b32 v20, v20, v24, v28 bitop3:0x96\nv_bitop3_b32 v24, v24, v28, v0 bitop3:0x96\nv_bitop3_b32 v28, v28, v0, v4 bitop3:0x96\n")
			             ::: "v0","v4","v8","v12","v16","v20","v24","v28");
		if (OP == 1) // spread: sources in banks (k, k+1, k+2)
			asm volatile(R8("v_bitop3_b32 v0, v0, v5, v10 bitop3:0x96\nv_bitop3_b32 v4, v4, v9, v14 bitop3:0x96\nv_bitop3_b32 v8, v8, v13, v18 bitop3:0x96\nv_bitop3_b32 v12, v12, v17, v22 bitop3:0x96\n"
			                "v_bitop3_b32 v16, v16, v21, v26 bitop3:0x96\nv_bitop3_b32 v20, v20, v25, v30 bitop3:0x96\nv_bitop3_b32 v24, v24, v29, v2 bitop3:0x96\nv_bitop3_b32 v28, v28, v1, v6 bitop3:0x96\n")
			             ::: "v0","v4","v8","v12","v16","v20","v24","v28");
		if (OP == 2) // two sources share a bank
			asm volatile(R8("v_bitop3_b32 v0, v0, v4, v9 bitop3:0x96\nv_bitop3_b32 v4, v4, v8, v13 bitop3:0x96\nv_bitop3_b32 v8, v8, v12, v17 bitop3:0x96\nv_bitop3_b32 v12, v12, v16, v21 bitop3:0x96\n"
			                "v_bitop3_b32 v16, v16, v20, v25 bitop3:0x96\nv_bitop3_b32 v20, v20, v24, v29 bitop3:0x96\nv_bitop3_b32 v24, v24, v28, v1 bitop3:0x96\nv_bitop3_b32 v28, v28, v0, v5 bitop3:0x96\n")
			             ::: "v0","v4","v8","v12","v16","v20","v24","v28");
		if (OP == 3) // one dependency chain
			asm volatile(R8(R8("v_bitop3_b32 v0, v0, v5, v10 bitop3:0x96\n")) ::: "v0");
		if (OP == 4) // two chains
			asm volatile(R8("v_bitop3_b32 v0, v0, v5, v10 bitop3:0x96\nv_bitop3_b32 v4, v4, v9, v14 bitop3:0x96\nv_bitop3_b32 v0, v0, v5, v10 bitop3:0x96\nv_bitop3_b32 v4, v4, v9, v14 bitop3:0x96\n"
			                "v_bitop3_b32 v0, v0, v5, v10 bitop3:0x96\nv_bitop3_b32 v4, v4, v9, v14 bitop3:0x96\nv_bitop3_b32 v0, v0, v5, v10 bitop3:0x96\nv_bitop3_b32 v4, v4, v9, v14 bitop3:0x96\n") ::: "v0","v4");
		if (OP == 5) // four chains
			asm volatile(R8("v_bitop3_b32 v0, v0, v5, v10 bitop3:0x96\nv_bitop3_b32 v4, v4, v9, v14 bitop3:0x96\nv_bitop3_b32 v8, v8, v13, v18 bitop3:0x96\nv_bitop3_b32 v12, v12, v17, v22 bitop3:0x96\n"
			                "v_bitop3_b32 v0, v0, v5, v10 bitop3:0x96\nv_bitop3_b32 v4, v4, v9, v14 bitop3:0x96\nv_bitop3_b32 v8, v8, v13, v18 bitop3:0x96\nv_bitop3_b32 v12, v12, v17, v22 bitop3:0x96\n") ::: "v0","v4","v8","v12");
		if (OP == 6) // v_xor (2 sources), 8 chains, same bank
			asm volatile(R8("v_xor_b32 v0, v0, v4\nv_xor_b32 v4, v4, v8\nv_xor_b32 v8, v8, v12\nv_xor_b32 v12, v12, v16\nv_xor_b32 v16, v16, v20\nv_xor_b32 v20, v20, v24\nv_xor_b32 v24, v24, v28\nv_xor_b32 v28, v28, v0\n")
			             ::: "v0","v4","v8","v12","v16","v20","v24","v28");
		if (OP == 7) // dpp mov, 8 chains
			asm volatile(R8("v_mov_b32_dpp v0, v1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp v4, v5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp v8, v9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp v12, v13 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
			                "v_mov_b32_dpp v16, v17 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp v20, v21 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp v24, v25 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp v28, v29 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
			             ::: "v0","v4","v8","v12","v16","v20","v24","v28");
		if (OP == 8) // or3, 8 chains spread
			asm volatile(R8("v_or3_b32 v0, v0, v5, v10\nv_or3_b32 v4, v4, v9, v14\nv_or3_b32 v8, v8, v13, v18\nv_or3_b32 v12, v12, v17, v22\nv_or3_b32 v16, v16, v21, v26\nv_or3_b32 v20, v20, v25, v30\nv_or3_b32 v24, v24, v29, v2\nv_or3_b32 v28, v28, v1, v6\n")
			             ::: "v0","v4","v8","v12","v16","v20","v24","v28");
	}
	u32 r;
	asm volatile("v_xor_b32 %0, v0, v4\nv_xor_b32 %0, %0, v8\nv_xor_b32 %0, %0, v12\nv_xor_b32 %0, %0, v16\nv_xor_b32 %0, %0, v20\nv_xor_b32 %0, %0, v24\nv_xor_b32 %0, %0, v28\n" : "=v"(r));
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> void run(const char *name, int waves_per_simd)
{
	const int threads = 256, blocks = 256 * waves_per_simd;
	u32 *d;
	CK(hipMalloc(&d, (size_t)threads * blocks * 4));
	const int iters = 2000;
	hipEvent_t a, b;
	CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	k<OP><<<blocks, threads>>>(d, 10);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a));
	k<OP><<<blocks, threads>>>(d, iters);
	CK(hipEventRecord(b));
	CK(hipEventSynchronize(b));
	float ms;
	CK(hipEventElapsedTime(&ms, a, b));
	const double insts = (double)iters * 64.0 * waves_per_simd;
	printf("%-26s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instr per SIMD\n", name, waves_per_simd, ms, ms * 1e6 / insts);
	CK(hipFree(d));
}
int main()
{
	for (int w : {1, 2, 3, 4, 8})
	{
		run<0>("bitop3 same bank", w); run<1>("bitop3 three banks", w); run<2>("bitop3 two in one bank", w);
		run<3>("bitop3 1 chain", w); run<4>("bitop3 2 chains", w); run<5>("bitop3 4 chains", w);
		run<6>("xor same bank", w); run<7>("mov dpp wave_shr", w); run<8>("or3", w);
	}
	return 0;
}
