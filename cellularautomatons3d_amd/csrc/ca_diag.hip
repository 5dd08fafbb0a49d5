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
} // namespace

hipError_t launch_copy_f4(const void *in, void *out, size_t bytes, hipStream_t stream)
{
	constexpr int PER = 4;
	const size_t n = bytes / 16u;
	const unsigned blocks = (unsigned)((n + 256u * PER - 1u) / (256u * PER));
	hipLaunchKernelGGL((copy_f4<PER>), dim3(blocks), dim3(256), 0, stream, (const u32x4 *)in, (u32x4 *)out, n);
	return hipGetLastError();
}

} // namespace ca3d
