// Can kernel boundaries be hidden by splitting each step into S z-slab kernels with neighbour-only dependencies
// (slab j of step k+1 waits for slabs j-1, j, j+1 of step k), so that the tail of one step overlaps the head of the
// next? Same 512^3 von Neumann load pattern as stencil_floor.hip (8 loads / 2 outputs per thread, XOR "compute").
// Builds the dependency graph explicitly (hipGraphAddKernelNode) and compares with the linear one-kernel-per-step graph.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 X(uint4 a, uint4 b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }
__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ in, uint4 *__restrict__ out, u32 G, u32 z0, u32 nzr)
{
	const u32 CV = G / 128, rows_per_block = 256 / CV, tiles = G / rows_per_block;
	const u32 b = blockIdx.x, nb = gridDim.x;
	const u32 v = (nb & 7u) == 0 ? (b & 7u) * (nb >> 3) + (b >> 3) : b;
	const size_t plane = (size_t)G * CV;
	const u32 zr = v / tiles, tile = v % tiles;
	const u32 t = tile * 256 + threadIdx.x, y = t / CV, cx = t % CV, z = z0 + zr * 2;
	const u32 ym = y == 0 ? 0 : y - 1, yp = y + 1 == G ? 0 : y + 1, zm = z == 0 ? 0 : z - 1, zp = z + 2 == G ? 0 : z + 2;
	const uint4 *p0 = in + z * plane, *p1 = p0 + plane;
	uint4 a[8] = {p0[ym * CV + cx], p0[y * CV + cx], p0[yp * CV + cx], p1[ym * CV + cx], p1[y * CV + cx], p1[yp * CV + cx],
	              in[zm * plane + y * CV + cx], in[zp * plane + y * CV + cx]};
	uint4 r0 = X(X(a[0], a[1]), X(a[2], a[6])), r1 = X(X(a[3], a[4]), X(a[5], a[7]));
	u32x4 v0 = {r0.x, r0.y, r0.z, r0.w}, v1 = {r1.x, r1.y, r1.z, r1.w};
	__builtin_nontemporal_store(v0, reinterpret_cast<u32x4 *>(out + z * plane + y * CV + cx));
	__builtin_nontemporal_store(v1, reinterpret_cast<u32x4 *>(out + (z + 1) * plane + y * CV + cx));
}
static double run_graph(hipGraph_t g, hipStream_t s, int steps, int reps)
{
	hipGraphExec_t ge;
	hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
	for (int w = 0; w < 3; w++) hipGraphLaunch(ge, s);
	hipStreamSynchronize(s);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0, s);
	for (int w = 0; w < reps; w++) hipGraphLaunch(ge, s);
	hipEventRecord(e1, s);
	hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	hipGraphExecDestroy(ge);
	return ms * 1e3 / (reps * steps);
}
int main(int argc, char **argv)
{
	const u32 G = argc > 1 ? (u32)atoi(argv[1]) : 512u;
	const int steps = 64;
	const size_t bytes = (size_t)G * G * G / 8;
	uint4 *buf[2];
	hipMalloc(&buf[0], bytes); hipMalloc(&buf[1], bytes);
	hipMemset(buf[0], 1, bytes); hipMemset(buf[1], 0, bytes);
	hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	const u32 CV = G / 128, tiles = G / (256 / CV);
	for (int S : {1, 2, 4, 8, 16})
	{
		hipGraph_t g; hipGraphCreate(&g, 0);
		std::vector<hipGraphNode_t> prev(S), cur(S);
		const u32 planes = G / S, nzr = planes / 2;
		for (int st = 0; st < steps; st++)
		{
			for (int j = 0; j < S; j++)
			{
				const uint4 *in = buf[st & 1]; uint4 *out = buf[(st + 1) & 1];
				u32 Gv = G, z0 = j * planes, nz = nzr;
				void *args[] = {&in, &out, &Gv, &z0, &nz};
				hipKernelNodeParams p = {};
				p.func = (void *)k; p.gridDim = dim3(tiles * nzr); p.blockDim = dim3(256); p.kernelParams = args;
				std::vector<hipGraphNode_t> deps;
				if (st > 0)
				{
					deps.push_back(prev[j]);
					if (S > 1) { deps.push_back(prev[(j + S - 1) % S]); if (S > 2) deps.push_back(prev[(j + 1) % S]); }
				}
				hipGraphAddKernelNode(&cur[j], g, deps.data(), deps.size(), &p);
			}
			prev = cur;
		}
		const double us = run_graph(g, s, steps, 16);
		printf("G %u: %2d slab kernels per step with neighbour dependencies: %.2f us per step\n", G, S, us);
		hipGraphDestroy(g);
	}
	return 0;
}
