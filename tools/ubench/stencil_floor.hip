// How much of the 512^3 step is the LOAD PATTERN? Same geometry as ca_packed_class (one uint4 column per thread,
// ZR planes per thread, 64-launch ping-pong graph), compute replaced by an XOR of everything loaded.
//   YM 0: rows y-1, y+1 of every output plane are loaded (the von Neumann kernel's pattern)
//   YM 1: y neighbours by lane exchange (ds_bpermute), wave-edge rows ignored (upper bound on the gain)
//   YM 2: y neighbours by DPP row shifts (16-lane rows only: timing probe, not a correct stencil)
//   YM 3: no y neighbours at all
//   ZH 1: centre rows of planes z-1 and z+ZR are loaded too; ZH 0: not (with YM 3 this is the plain copy)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 X(uint4 a, uint4 b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }
__device__ __forceinline__ uint4 SH(uint4 a, int src) { return make_uint4(__shfl((int)a.x, src), __shfl((int)a.y, src), __shfl((int)a.z, src), __shfl((int)a.w, src)); }
template <int CTRL> __device__ __forceinline__ u32 dpp(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ uint4 DP(uint4 a) { return make_uint4(dpp<CTRL>(a.x), dpp<CTRL>(a.y), dpp<CTRL>(a.z), dpp<CTRL>(a.w)); }
__device__ __forceinline__ void ST(uint4 *p, uint4 v, int nt)
{
	if (nt) { u32x4 r = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(r, reinterpret_cast<u32x4 *>(p)); }
	else *p = v;
}
template <int ZR, int YM, int ZH, int NTH, int NV = 0>
__global__ __launch_bounds__(NTH) void k(const uint4 *__restrict__ in, uint4 *__restrict__ out, const uint4 *__restrict__ extra, u32 G, int nt, int nostore)
{
	const u32 CV = G / 128, rows_per_block = NTH / CV;
	const u32 b = blockIdx.x, nb = gridDim.x;
	const u32 v = (b & 7u) * (nb >> 3) + (b >> 3);
	const size_t plane = (size_t)G * CV;
	const u32 tiles = G / rows_per_block;
	const u32 zr = v / tiles, tile = v % tiles;
	const u32 t = tile * NTH + threadIdx.x, y = t / CV, cx = t % CV, z = zr * ZR;
	const u32 ym = y == 0 ? 0 : y - 1, yp = y + 1 == G ? 0 : y + 1, zm = z == 0 ? 0 : z - 1, zp = z + ZR == G ? 0 : z + ZR;
	uint4 c[ZR], m[ZR], p[ZR], lo, hi;
#pragma unroll
	for (int q = 0; q < ZR; q++)
	{
		c[q] = in[(z + q) * plane + y * CV + cx];
		if (YM == 0) { m[q] = in[(z + q) * plane + ym * CV + cx]; p[q] = in[(z + q) * plane + yp * CV + cx]; }
	}
	if (ZH == 1) { lo = in[zm * plane + y * CV + cx]; hi = in[zp * plane + y * CV + cx]; }
	if (ZH == 2) { lo = extra[(size_t)(2 * zr) * plane + y * CV + cx]; hi = extra[(size_t)(2 * zr + 1) * plane + y * CV + cx]; } // same volume, lines nobody else reads
	const int lane = threadIdx.x & 63;
	uint4 r[ZR];
#pragma unroll
	for (int q = 0; q < ZR; q++)
	{
		r[q] = c[q];
		if (YM == 0) r[q] = X(r[q], X(m[q], p[q]));
		if (YM == 1) r[q] = X(r[q], X(SH(c[q], lane - (int)CV), SH(c[q], lane + (int)CV)));
		if (YM == 2) r[q] = X(r[q], X(DP<0x114>(c[q]), DP<0x104>(c[q]))); // row_shr:4, row_shl:4
		if (ZH || ZR > 1) r[q] = X(r[q], X(q ? c[q > 0 ? q - 1 : 0] : (ZH ? lo : c[0]), q + 1 < ZR ? c[q + 1 < ZR ? q + 1 : 0] : (ZH ? hi : c[0])));
	}
	if (NV > 0)
	{
		// NV dependent-in-pairs VALU ops per output word, like the adder trees of the real kernel
#pragma unroll
		for (int q = 0; q < ZR; q++)
		{
			u32 w[4] = {r[q].x, r[q].y, r[q].z, r[q].w};
#pragma unroll
			for (int n = 0; n < NV; n++)
#pragma unroll
				for (int i = 0; i < 4; i++) w[i] = __builtin_amdgcn_bitop3_b32(w[i], w[(i + 1) & 3], c[q].x, 0x96);
			r[q] = make_uint4(w[0], w[1], w[2], w[3]);
		}
	}
	if (nostore)
	{
		u32 acc = 0;
#pragma unroll
		for (int q = 0; q < ZR; q++) acc |= r[q].x ^ r[q].y ^ r[q].z ^ r[q].w;
		if (acc != 0x1234567u) return;
	}
#pragma unroll
	for (int q = 0; q < ZR; q++) ST(out + (z + q) * plane + y * CV + cx, r[q], nt);
}
template <int ZR, int YM, int ZH, int NTH, int NV = 0> void run(u32 G, int nt, int nostore)
{
	const size_t bytes = (size_t)G * G * G / 8;
	uint4 *a, *b, *c;
	hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes * 2); hipMemset(c, 3, bytes * 2);
	hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
	hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	const size_t threads = (size_t)G * G * (G / 128) / ZR;
	const int blocks = (int)(threads / NTH);
	hipGraph_t g; hipGraphExec_t ge;
	hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
	for (int i = 0; i < 64; i++) hipLaunchKernelGGL((k<ZR, YM, ZH, NTH, NV>), dim3(blocks), dim3(NTH), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, c, G, nt, nostore);
	hipStreamEndCapture(s, &g);
	hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
	for (int w = 0; w < 4; w++) hipGraphLaunch(ge, s);
	hipStreamSynchronize(s);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0, s);
	for (int w = 0; w < 16; w++) hipGraphLaunch(ge, s);
	hipEventRecord(e1, s);
	hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	static const char *ymn[] = {"y rows loaded", "y by bpermute", "y by DPP", "no y"};
	const int loads = ZR * (YM == 0 ? 3 : 1) + (ZH ? 2 : 0);
	printf("G %4u ZR %d %-14s zhalo %d block %4d (%2d loads / %d outputs) valu/word %3d nostore %d: %6.2f us\n", G, ZR, ymn[YM], ZH, NTH, loads, ZR, NV, nostore, ms * 1e3 / (16 * 64));
	hipFree(a); hipFree(b); hipFree(c); hipStreamDestroy(s);
}
template <int ZR, int NTH> void family(u32 G, int nt, int nostore)
{
	run<ZR, 0, 1, NTH>(G, nt, nostore);
	run<ZR, 1, 1, NTH>(G, nt, nostore);
	run<ZR, 2, 1, NTH>(G, nt, nostore);
	run<ZR, 3, 1, NTH>(G, nt, nostore);
	run<ZR, 3, 2, NTH>(G, nt, nostore);
	run<ZR, 1, 0, NTH>(G, nt, nostore);
	run<ZR, 3, 0, NTH>(G, nt, nostore);
}
template <int NTH> void valu_family(u32 G, int nt)
{
	run<2, 0, 1, NTH, 0>(G, nt, 0);
	run<2, 0, 1, NTH, 4>(G, nt, 0);
	run<2, 0, 1, NTH, 8>(G, nt, 0);
	run<2, 0, 1, NTH, 12>(G, nt, 0);
	run<2, 0, 1, NTH, 16>(G, nt, 0);
	run<2, 0, 1, NTH, 24>(G, nt, 0);
}
int main(int argc, char **argv)
{
	const u32 G = argc > 1 ? (u32)atoi(argv[1]) : 512u;
	const int nt = G <= 512;
	if (argc > 2)
	{
		valu_family<64>(G, nt);
		valu_family<128>(G, nt);
		valu_family<256>(G, nt);
		valu_family<512>(G, nt);
		run<1, 0, 1, 256, 16>(G, nt, 0);
		run<4, 0, 1, 256, 16>(G, nt, 0);
		return 0;
	}

	for (int nostore = 0; nostore < 2; nostore++)
	{
		for (int rep = 0; rep < 2; rep++)
		{
			family<2, 256>(G, nt, nostore);
			family<8, 256>(G, nt, nostore);
			family<2, 64>(G, nt, nostore);
		}
	}
	return 0;
}
