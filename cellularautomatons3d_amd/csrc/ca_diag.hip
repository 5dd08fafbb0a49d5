// Measurement helper behind ca3d_measure_copy (include/ca3d.h): a float4-per-lane device-to-device copy, the practical
// ceiling of any kernel that reads and writes HBM once (MI355X_MICROARCH.md: 6.29 TB/s of the 8 TB/s spec). Not on any
// product path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ca3d_internal.h"

namespace ca3d
{
namespace
{
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// PER 16-byte elements per thread, all loads issued before the first store; grid-contiguous per load (a wave reads 1 KiB
// per instruction). Non-temporal stores: the destination is not read again.
template <int PER>
__global__ __launch_bounds__(256) void copy_f4(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n)
{
	const size_t i = (size_t)blockIdx.x * (256u * PER) + threadIdx.x;
	u32x4 v[PER];
#pragma unroll
	for (int k = 0; k < PER; k++)
		if (i + (size_t)k * 256u < n) v[k] = __builtin_nontemporal_load(in + i + (size_t)k * 256u);
#pragma unroll
	for (int k = 0; k < PER; k++)
		if (i + (size_t)k * 256u < n) __builtin_nontemporal_store(v[k], out + i + (size_t)k * 256u);
}
// one wave that stays on the chip for `ticks` of the 100 MHz s_memrealtime counter (streams_concurrent below); it always ends
__global__ void spin_ticks(unsigned long long ticks)
{
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
__global__ void noop_kernel() {}
} // namespace

// Are two streams served by DIFFERENT hardware queues — does work on `b` run while a kernel on `a` is still running? HIP maps its
// streams onto a few hardware queues (four by default) by a policy of its own, and two streams on one queue execute in order: the
// engine's two frame lanes (ca3d_api.cpp) overlap nothing then. Probed once per pair: a 2 ms one-wave spin on `a`, an empty kernel
// on `b`; `b` finishing first means separate queues. Both streams must be idle; they are when the call returns.
hipError_t streams_concurrent(hipStream_t a, hipStream_t b, bool *out)
{
	*out = false;
	hipError_t e = hipStreamSynchronize(a);
	if (e == hipSuccess) e = hipStreamSynchronize(b);
	hipEvent_t ea = nullptr, eb = nullptr;
	if (e == hipSuccess) e = hipEventCreateWithFlags(&ea, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&eb, hipEventDisableTiming);
	if (e == hipSuccess)
	{
		hipLaunchKernelGGL(spin_ticks, dim3(1), dim3(64), 0, a, 200000ull);
		e = hipEventRecord(ea, a);
		hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, b);
		if (e == hipSuccess) e = hipEventRecord(eb, b);
		if (e == hipSuccess) e = hipEventSynchronize(eb);
		if (e == hipSuccess)
		{
			*out = hipEventQuery(ea) == hipErrorNotReady;
			(void)hipGetLastError(); // (hipErrorNotReady is the answer, not an error)
			e = hipEventSynchronize(ea);
		}
	}
	if (ea) hipEventDestroy(ea);
	if (eb) hipEventDestroy(eb);
	return e;
}

hipError_t launch_copy_f4(const void *in, void *out, size_t bytes, hipStream_t stream)
{
	constexpr int PER = 4;
	const size_t n = bytes / 16u;
	const unsigned blocks = (unsigned)((n + 256u * PER - 1u) / (256u * PER));
	hipLaunchKernelGGL((copy_f4<PER>), dim3(blocks), dim3(256), 0, stream, (const u32x4 *)in, (u32x4 *)out, n);
	return hipGetLastError();
}

} // namespace ca3d
