// What does it cost just to stream the 512^3 packed state once (16 MiB in, 16 MiB out) per kernel, back to back?
// Ping-pong copy kernels in a hipGraph: the floor any one-step-per-launch CA kernel sits on.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32;
template <int NT, int PER>
__global__ void copy_k(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n, int mode)
{
	size_t i = (size_t)blockIdx.x * blockDim.x * PER + threadIdx.x;
	uint4 v[PER];
#pragma unroll
	for (int k = 0; k < PER; k++) v[k] = i + (size_t)k * blockDim.x < n ? in[i + (size_t)k * blockDim.x] : make_uint4(0, 0, 0, 0);
#pragma unroll
	for (int k = 0; k < PER; k++)
	{
		if (i + (size_t)k * blockDim.x >= n) continue;
		if (mode == 2 && v[k].x != 0x1234567u) continue; // read only
		if (NT)
		{
			typedef u32 u32x4 __attribute__((ext_vector_type(4)));
			u32x4 r = {v[k].x ^ 1u, v[k].y, v[k].z, v[k].w};
			__builtin_nontemporal_store(r, reinterpret_cast<u32x4 *>(out + i + (size_t)k * blockDim.x));
		}
		else { uint4 r = v[k]; r.x ^= 1u; out[i + (size_t)k * blockDim.x] = r; }
	}
}
template <int NT, int PER> void run(size_t bytes, int mode, const char *name)
{
	const size_t n = bytes / 16;
	uint4 *a, *b;
	hipMalloc(&a, bytes); hipMalloc(&b, bytes);
	hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
	hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	const int threads = 256, blocks = (int)((n + (size_t)threads * PER - 1) / ((size_t)threads * PER));
	hipGraph_t g; hipGraphExec_t ge;
	hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
	for (int k = 0; k < 64; k++) hipLaunchKernelGGL((copy_k<NT, PER>), dim3(blocks), dim3(threads), 0, s, (k & 1) ? b : a, (k & 1) ? a : b, n, mode);
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
	printf("%-28s %6.1f MiB  PER %d: %.2f us per kernel (%.2f TB/s of in+out)\n", name, bytes / 1048576.0, PER, ms * 1e3 / (16 * 64), (mode == 2 ? 1.0 : 2.0) * bytes / (ms * 1e-3 / (16 * 64)) / 1e12);
	hipFree(a); hipFree(b);
}
int main()
{
	for (size_t mb : {16, 128})
	{
		run<0, 1>(mb << 20, 0, "copy plain"); run<1, 1>(mb << 20, 0, "copy nt-store");
		run<0, 2>(mb << 20, 0, "copy plain"); run<1, 2>(mb << 20, 0, "copy nt-store");
		run<0, 4>(mb << 20, 0, "copy plain"); run<1, 4>(mb << 20, 0, "copy nt-store");
		run<0, 1>(mb << 20, 2, "read only");
	}
	return 0;
}
