// Micro-benchmark: issue rate of the bit-manipulation VALU ops the CA kernels are built from (gfx950).
// Inline asm keeps the instruction stream exactly as written: 8 independent dependency chains, 64 ops per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32;
#define CHAIN8(INSN)                                                    \
	asm volatile(INSN(0) INSN(1) INSN(2) INSN(3) INSN(4) INSN(5) INSN(6) INSN(7) \
	             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(x), "v"(y));
#define BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n"
#define BITOPM(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0xe8\n"
#define ALIGNB(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
#define XOR2(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define BFI(i) "v_bfi_b32 %" #i ", %" #i ", %8, %9\n"
#define XOR3(i) "v_xor3_b32 %" #i ", %" #i ", %8, %9\n"
#define ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define CNDM(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define CNDE(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"
#define ANDB(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define MOVS(i) "v_mov_b32 %" #i ", s10\n"
template <int OP> __global__ void k(u32 *out, int iters)
{
	u32 r0 = threadIdx.x, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17, r7 = r0 * 19;
	u32 x = blockIdx.x * 2654435761u + threadIdx.x, y = threadIdx.x * 7u + 1;
	asm volatile("s_mov_b64 s[10:11], 0x5555\ns_mov_b64 vcc, 0x3333" ::: "s10", "s11", "vcc");
	for (int it = 0; it < iters; it++)
	{
#pragma unroll
		for (int u = 0; u < 8; u++)
		{
			if (OP == 0) { CHAIN8(BITOP3) }
			if (OP == 1) { CHAIN8(BITOPM) }
			if (OP == 2) { CHAIN8(ALIGNB) }
			if (OP == 3) { CHAIN8(XOR2) }
			if (OP == 4) { CHAIN8(ANDOR) }
			if (OP == 5) { CHAIN8(BFI) }
			if (OP == 7) { CHAIN8(ADD) }
			if (OP == 8) { CHAIN8(CNDM) }
			if (OP == 9) { CHAIN8(CNDE) }
			if (OP == 10) { CHAIN8(ANDB) }
			if (OP == 11) { CHAIN8(LSHL) }
			if (OP == 12) { CHAIN8(MOVS) }
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}
template <int OP> void run(const char *name, int waves_per_simd)
{
	const int threads = 256, blocks = 256 * waves_per_simd;
	u32 *d;
	hipMalloc(&d, (size_t)threads * blocks * 4);
	const int iters = 2000;
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	k<OP><<<blocks, threads>>>(d, 10);
	hipDeviceSynchronize();
	hipEventRecord(a);
	k<OP><<<blocks, threads>>>(d, iters);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	const double insts = (double)iters * 64.0 * waves_per_simd;
	printf("%-12s waves/SIMD %d: %.3f ms -> %.2f clk per wave-instr per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / insts);
	hipFree(d);
}
int main()
{
	for (int w : {1, 2, 4})
	{
		run<0>("bitop3 xor3", w); run<1>("bitop3 maj", w); run<2>("alignbit", w); run<3>("xor", w); run<4>("and_or", w);
		run<5>("bfi", w); run<7>("add_u32", w); run<8>("cndmask vcc", w); run<9>("cndmask e64", w); run<10>("and", w); run<11>("lshlrev", w); run<12>("mov sgpr", w);
	}
	return 0;
}
