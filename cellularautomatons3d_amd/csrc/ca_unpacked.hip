// Legacy CA step: one u32 (0/1) per cell, one rule-set, toroidal through u32 wrap-around
// (shaders/compute.wgsl:17-47, 49-53, 101, 160-174). The reference does not load this kernel any more
// (main_pathtraced.js:710) but BASELINE names it; it is the layout where the step is genuinely HBM-bound
// (8 B per cell-step).
//
// ca_unpacked_literal: one thread per cell, exact for ANY u32 cell values (the count is the u32 sum of raw
// neighbour values, `state == 1` / `== 0` tests, LUT entries tested with `> 0`), any G.
#include "ca3d_internal.h"

namespace ca3d
{
namespace
{

using u32 = uint32_t;

struct UnpackedArgs
{
	u32 n_offs;
	int8_t off[kMaxOffsets][3];
	u32 survive_gt0[3], born_gt0[3]; // bit k of word k/32 = LUT[k] > 0, k in 0..80
};

__device__ __forceinline__ u32 lut_bit(const u32 (&m)[3], u32 count)
{
	const u32 k = count < (u32)CA3D_LUT_LEN ? count : (u32)CA3D_LUT_LEN - 1u; // robust-access clamp
	return (m[k >> 5] >> (k & 31u)) & 1u;
}

__global__ __launch_bounds__(256) void ca_unpacked_literal(const u32 *__restrict__ in, u32 *__restrict__ out,
                                                           PlaneRange pr, UnpackedArgs a)
{
	const u32 G = pr.G;
	const size_t plane_cells = (size_t)G * G;
	const size_t gid = (size_t)blockIdx.x * 256u + threadIdx.x;
	if (gid >= (size_t)(pr.hi - pr.lo) * plane_cells) return;
	const u32 pj = (u32)(gid / plane_cells);
	const u32 rem = (u32)(gid - (size_t)pj * plane_cells);
	const u32 y = rem / G, x = rem - y * G;
	const u32 j = pr.lo + pj;
	u32 count = 0;
	for (u32 i = 0; i < a.n_offs; i++)
	{
		// vec3u(vec3i(cell) + offset) % G  (compute.wgsl:27, 42): -1 arrives as 0xFFFFFFFF
		const u32 nx = (u32)((int)x + a.off[i][0]) % G;
		const u32 ny = (u32)((int)y + a.off[i][1]) % G;
		u32 nj;
		if (pr.wrap_full) nj = (u32)((int)j + a.off[i][2]) % G;
		else nj = (u32)((int)j + a.off[i][2]); // slab with ghosts: the neighbour plane is physically adjacent
		count += in[(size_t)nj * plane_cells + (size_t)ny * G + nx];
	}
	const size_t idx = (size_t)j * plane_cells + (size_t)y * G + x;
	const u32 st = in[idx];
	u32 o = 0;
	if (st == 1u && lut_bit(a.survive_gt0, count)) o = 1u;
	else if (st == 0u && lut_bit(a.born_gt0, count)) o = 1u;
	out[idx] = o;
}

} // namespace

hipError_t launch_unpacked_step(const UnpackedLaunch &l, hipStream_t stream, const char **kernel_name)
{
	if (kernel_name) *kernel_name = "ca_unpacked_literal";
	if (l.pr.hi <= l.pr.lo) return hipSuccess;
	const CanonRules &r = *l.rules;
	UnpackedArgs a{};
	a.n_offs = r.lists.n[0];
	for (u32 i = 0; i < a.n_offs; i++)
	{
		const u32 code = r.lists.code[0][i];
		a.off[i][0] = (int8_t)((int)(code & 3u) - 1);
		a.off[i][1] = (int8_t)((int)((code >> 2) & 3u) - 1);
		a.off[i][2] = (int8_t)((int)((code >> 4) & 3u) - 1);
	}
	for (int k = 0; k < CA3D_LUT_LEN; k++)
	{
		if (r.survive_raw[k] > 0u) a.survive_gt0[k >> 5] |= 1u << (k & 31);
		if (r.born_raw[k] > 0u) a.born_gt0[k >> 5] |= 1u << (k & 31);
	}
	const size_t total = (size_t)(l.pr.hi - l.pr.lo) * l.pr.G * l.pr.G;
	const size_t blocks = (total + 255u) / 256u;
	if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
	hipLaunchKernelGGL(ca_unpacked_literal, dim3((u32)blocks), dim3(256), 0, stream, l.in, l.out, l.pr, a);
	return hipGetLastError();
}

} // namespace ca3d
