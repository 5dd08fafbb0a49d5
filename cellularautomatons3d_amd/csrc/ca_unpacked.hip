// Legacy CA step: one u32 (0/1) per cell, one rule-set, toroidal through u32 wrap-around
// (shaders/compute.wgsl:17-47, 49-53, 101, 160-174). The reference does not load this kernel any more
// (main_pathtraced.js:710) but BASELINE names it; it is the layout where the step is genuinely HBM-bound
// (8 B per cell-step).
//
// ca_unpacked_literal: one thread per cell, exact for ANY u32 cell values (the count is the u32 sum of raw
// neighbour values, `state == 1` / `== 0` tests, LUT entries tested with `> 0`), any G.
//
// ca_unpacked_ballot: the HBM-bound path (8 B per cell-step) for 0/1 states on power-of-two grids, where the
// legacy kernel is a true torus. A workgroup packs a tile (+1 halo in y and z, full x) into bit-planes in LDS
// with one __ballot per 64 cells — the loads are plain coalesced dwords —, updates it with the same bit-sliced
// adders and rule programs as the packed kernel, and expands the result back to one u32 per cell on the way
// out. HBM sees each cell ~1.2 times in and once out instead of 7-27 cached re-reads per cell.
#include <mutex>

#include "ca3d_internal.h"

namespace ca3d
{
namespace
{

#include "ca_bitslice.inc"

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

constexpr int kBTY = 32; // tile rows; planes per tile (kBTZ) is a template parameter: 16, or 8 on grids with few tiles // tile rows x planes (interior)
constexpr int kBThreads = 1024;     // 16 waves per workgroup: the pack / unpack phases are load-latency-bound

template <int MAIN, bool FAST, int kBTZ>
__global__ __launch_bounds__(kBThreads) void ca_unpacked_ballot(const u32 *__restrict__ in, u32 *__restrict__ out, PlaneRange pr,
                                                          u32 ny, u32 cv_shift, PackedRuleArgs rules_in, u32 pack_loads)
{
	extern __shared__ __attribute__((aligned(16))) u32 lds[]; // in bits [(TZ+2)][(TY+2)][C], then out bits [TZ][TY][C]
	const u32 G = pr.G, C = G / 32u, X64 = G / 64u;
	const size_t plane_cells = (size_t)G * G;
	const u32 tid = threadIdx.x, lane = tid & 63u;
	const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6)); // wave-uniform: keeps the row maths scalar
	const u32 tz = blockIdx.x / ny, ty = blockIdx.x - tz * ny;
	const int y0 = (int)(ty * kBTY), z0 = (int)pr.lo + (int)(tz * kBTZ);
	constexpr u32 RY = kBTY + 2, RZ = kBTZ + 2;
	u32 *in_bits = lds;
	u32 *out_bits = lds + RZ * RY * C;

	// ---- phase 1: pack (tile + halo) into bits. Each lane loads 4 cells (one coalesced dwordx4: 1 KiB per wave),
	// folds them into a nibble, and three lane exchanges gather 8 nibbles into the word the first lane of every
	// octet writes to LDS. (One __ballot per 64 dword-loaded cells does the same with 4x the load instructions;
	// the phase is load-latency-bound, so bytes in flight per wave are what counts.)
	constexpr u32 NW = kBThreads / 64u;
	const u32 X256 = G / 256u; // dwordx4 wave-loads per row (G >= 256), else the ballot path below
	if (X256 > 0 && pack_loads == 0u)
	{
		// (round-1 form, kept for A/B timing: CA3D_UNPACKED_LOADS=0) one row at a time, X256 loads in flight per wave
		for (u32 row = wave; row < RZ * RY; row += NW)
		{
			const u32 zz = row / RY, yy = row - zz * RY;
			int gy = y0 + (int)yy - 1, gz = z0 + (int)zz - 1;
			gy = gy < 0 ? gy + (int)G : (gy >= (int)G ? gy - (int)G : gy);
			if (pr.wrap_full) gz = gz < 0 ? gz + (int)G : (gz >= (int)G ? gz - (int)G : gz);
			else gz = gz < 0 ? 0 : (gz >= (int)pr.nplanes ? (int)pr.nplanes - 1 : gz);
			const uint4 *src = reinterpret_cast<const uint4 *>(in + (size_t)gz * plane_cells + (size_t)gy * G) + lane;
			u32 *dst = in_bits + row * C + (lane >> 3);
			for (u32 x0 = 0; x0 < X256; x0 += 4u)
			{
				uint4 vals[4];
#pragma unroll
				for (int k = 0; k < 4; k++) vals[k] = x0 + (u32)k < X256 ? src[(x0 + (u32)k) * 64u] : make_uint4(0, 0, 0, 0);
#pragma unroll
				for (int k = 0; k < 4; k++)
				{
					u32 n = (vals[k].x != 0u ? 1u : 0u) | (vals[k].y != 0u ? 2u : 0u) | (vals[k].z != 0u ? 4u : 0u) | (vals[k].w != 0u ? 8u : 0u);
					n |= dpp_mov<0x101>(n) << 4;  // row_shl:1 = lane + 1 (the eight lanes of a word sit in one DPP row)
					n |= dpp_mov<0x102>(n) << 8;
					n |= dpp_mov<0x104>(n) << 16;
					if ((lane & 7u) == 0 && x0 + (u32)k < X256) dst[(x0 + (u32)k) * 8u] = n;
				}
			}
		}
	}
	else if (X256 > 0)
	{
		// Eight dwordx4 loads (8 KiB per wave) are issued before the first one is used — the rows of a wave are taken 8 / X256 at a
		// time: with one row (2 loads at 512^3) in flight per wave the phase waited a memory round trip per row
		// (r2: 238 us per 512^3 step; the loads of a workgroup's 16 waves did not cover the latency).
		const u32 xshift = X256 >= 4u ? 2u : (X256 == 2u ? 1u : 0u); // 1, 2 or 4 loads per row (G = 256, 512, 1024: what fits the LDS)
		const u32 xper = 1u << xshift, rif = max(1u, pack_loads >> xshift), nrows = RZ * RY;
		auto row_src = [&](u32 row) -> const uint4 * {
			const u32 zz = row / RY, yy = row - zz * RY;
			int gy = y0 + (int)yy - 1, gz = z0 + (int)zz - 1;
			gy = gy < 0 ? gy + (int)G : (gy >= (int)G ? gy - (int)G : gy); // toroidal (power-of-two G)
			if (pr.wrap_full) gz = gz < 0 ? gz + (int)G : (gz >= (int)G ? gz - (int)G : gz);
			else gz = gz < 0 ? 0 : (gz >= (int)pr.nplanes ? (int)pr.nplanes - 1 : gz); // slab: ghosts are adjacent planes
			return reinterpret_cast<const uint4 *>(in + (size_t)gz * plane_cells + (size_t)gy * G) + lane;
		};
		for (u32 base = wave; base < nrows; base += NW * rif)
		{
			uint4 vals[8];
#pragma unroll
			for (int q = 0; q < 8; q++)
			{
				const u32 row = base + NW * ((u32)q >> xshift), xc = (u32)q & (xper - 1u);
				vals[q] = (row < nrows && (u32)q < rif * xper) ? row_src(row)[xc * 64u] : make_uint4(0, 0, 0, 0);
			}
#pragma unroll
			for (int q = 0; q < 8; q++)
			{
				const u32 row = base + NW * ((u32)q >> xshift), xc = (u32)q & (xper - 1u);
				u32 n = (vals[q].x != 0u ? 1u : 0u) | (vals[q].y != 0u ? 2u : 0u) | (vals[q].z != 0u ? 4u : 0u) | (vals[q].w != 0u ? 8u : 0u);
				n |= dpp_mov<0x101>(n) << 4;  // row_shl:1 = lane + 1 (the eight lanes of a word sit in one DPP row)
				n |= dpp_mov<0x102>(n) << 8;
				n |= dpp_mov<0x104>(n) << 16;
				if ((lane & 7u) == 0 && row < nrows && (u32)q < rif * xper) in_bits[row * C + (lane >> 3) + xc * 8u] = n;
			}
		}
	}
	else
	{
		for (u32 row = wave; row < RZ * RY; row += NW)
		{
			const u32 zz = row / RY, yy = row - zz * RY;
			int gy = y0 + (int)yy - 1, gz = z0 + (int)zz - 1;
			gy = gy < 0 ? gy + (int)G : (gy >= (int)G ? gy - (int)G : gy);
			if (pr.wrap_full) gz = gz < 0 ? gz + (int)G : (gz >= (int)G ? gz - (int)G : gz);
			else gz = gz < 0 ? 0 : (gz >= (int)pr.nplanes ? (int)pr.nplanes - 1 : gz);
			const u32 *src = in + (size_t)gz * plane_cells + (size_t)gy * G + lane;
			u32 *dst = in_bits + row * C;
			for (u32 xc = 0; xc < X64; xc++)
			{
				const unsigned long long m = __ballot(src[xc * 64u] != 0u);
				if (lane == 0) { dst[xc * 2u] = (u32)m; dst[xc * 2u + 1u] = (u32)(m >> 32); }
			}
		}
	}
	__syncthreads();

	// ---- phase 2: the packed update on the tile (4 words per item), torus in x through the row ends
	FastRules<MAIN, false, false> frules;
	if (FAST) frules = expand_rules<MAIN, false, false>(rules_in);
	const u32 CV = C / 4u;
	const u32 nitems = (kBTZ * kBTY) << cv_shift;
	for (u32 it = tid; it < nitems; it += (u32)kBThreads)
	{
		const u32 cxv = it & (CV - 1u), rrow = it >> cv_shift;
		const u32 ry = rrow % kBTY, rz = rrow / kBTY;
		const u32 cx0 = cxv * 4u;
		auto seg = [&](u32 zz, u32 yy) {
			SegT<4> sg;
			const u32 *row = in_bits + (zz * RY + yy) * C;
			const uint4 w = *reinterpret_cast<const uint4 *>(row + cx0);
			sg.w[0] = w.x; sg.w[1] = w.y; sg.w[2] = w.z; sg.w[3] = w.w;
			sg.lo = row[cx0 == 0 ? C - 1u : cx0 - 1u];
			sg.hi = row[cx0 + 4u == C ? 0u : cx0 + 4u];
			return sg;
		};
		PlaneRowsT<4> P[3];
#pragma unroll
		for (int dz = 0; dz < 3; dz++)
		{
			P[dz].ym = seg(rz + (u32)dz, ry);
			P[dz].c = seg(rz + (u32)dz, ry + 1u);
			P[dz].yp = seg(rz + (u32)dz, ry + 2u);
		}
		u32 o[4];
		if (FAST) evolve<4, MAIN, false, false>(P[0], P[1], P[2], 0xFFFFFFFFu, frules, o);
		else evolve<4, MAIN, false, false>(P[0], P[1], P[2], 0xFFFFFFFFu, rules_in, o);
		uint4 ov;
		ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
		*reinterpret_cast<uint4 *>(out_bits + rrow * C + cx0) = ov;
	}
	__syncthreads();

	// ---- phase 3: expand bits to one u32 per cell: 4 cells per lane, coalesced dwordx4 stores, whole rows per wave
	for (u32 row = wave; row < kBTZ * kBTY; row += NW)
	{
		const u32 rz = row / kBTY, ry = row - rz * kBTY;
		const int gy = y0 + (int)ry, gz = z0 + (int)rz;
		if (gy >= (int)G || gz >= (int)pr.hi) continue;
		u32 *drow = out + (size_t)gz * plane_cells + (size_t)gy * G;
		if (X256 > 0)
		{
			const u32 *srcb = out_bits + row * C + (lane >> 3);
			const u32 sh = (lane & 7u) * 4u;
			for (u32 xc = 0; xc < X256; xc++)
			{
				const u32 w = srcb[xc * 8u] >> sh;
				uint4 v;
				v.x = w & 1u; v.y = (w >> 1) & 1u; v.z = (w >> 2) & 1u; v.w = (w >> 3) & 1u;
				reinterpret_cast<uint4 *>(drow)[xc * 64u + lane] = v;
			}
		}
		else
		{
			const u32 *srcb = out_bits + row * C + (lane >> 5);
			for (u32 xc = 0; xc < X64; xc++) drow[xc * 64u + lane] = (srcb[xc * 2u] >> (lane & 31u)) & 1u;
		}
	}
}

// ---------------------------------------------------------------------------------------------- pipelined form
// ca_unpacked_ballot runs pack -> update -> unpack once per tile, and every workgroup of a 512^3 launch is resident at
// once (512 tiles, two per CU): the whole chip reads, then computes, then writes — HBM is never read and written at the
// same time (262 us per step = 4.1 TB/s of the 6.5 TB/s a copy reaches). Here a workgroup walks its tile in groups of SUB
// planes: while group k is expanded and stored, the rows group k + 1 needs are loaded and packed — every wave alternates
// a row of loads with a row of stores, so both directions are in flight for most of the launch. Same tile, same LDS image
// of input bits, same update code; the output bits are double-buffered per group.
template <int MAIN, bool FAST, int kBTZ, int kSub>
__global__ __launch_bounds__(kBThreads) void ca_unpacked_pipe(const u32 *__restrict__ in, u32 *__restrict__ out, PlaneRange pr,
                                                              u32 ny, u32 cv_shift, PackedRuleArgs rules_in)
{
	extern __shared__ __attribute__((aligned(16))) u32 lds[]; // in bits [(TZ+2)][(TY+2)][C], then out bits [2][SUB][TY][C]
	constexpr u32 SUB = kSub, K = kBTZ / SUB;
	const u32 G = pr.G, C = G / 32u;
	const size_t plane_cells = (size_t)G * G;
	const u32 tid = threadIdx.x, lane = tid & 63u;
	const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
	const u32 tz = blockIdx.x / ny, ty = blockIdx.x - tz * ny;
	const int y0 = (int)(ty * kBTY), z0 = (int)pr.lo + (int)(tz * kBTZ);
	constexpr u32 RY = kBTY + 2, RZ = kBTZ + 2, NW = kBThreads / 64u;
	u32 *in_bits = lds;
	u32 *out_bits = lds + RZ * RY * C;
	const u32 X256 = G / 256u; // dwordx4 wave-loads per row: 1, 2 or 4 (the launcher takes this kernel for G >= 256 only)

	auto src_row = [&](u32 row) -> const uint4 * { // image row (zz, yy) -> the cells it packs
		const u32 zz = row / RY, yy = row - zz * RY;
		int gy = y0 + (int)yy - 1, gz = z0 + (int)zz - 1;
		gy = gy < 0 ? gy + (int)G : (gy >= (int)G ? gy - (int)G : gy); // toroidal (power-of-two G)
		if (pr.wrap_full) gz = gz < 0 ? gz + (int)G : (gz >= (int)G ? gz - (int)G : gz);
		else gz = gz < 0 ? 0 : (gz >= (int)pr.nplanes ? (int)pr.nplanes - 1 : gz); // slab: ghosts are adjacent planes
		return reinterpret_cast<const uint4 *>(in + (size_t)gz * plane_cells + (size_t)gy * G) + lane;
	};
	auto pack_issue = [&](u32 row, uint4 (&v)[4]) {
		const uint4 *src = src_row(row);
#pragma unroll
		for (int k = 0; k < 4; k++) v[k] = (u32)k < X256 ? src[(u32)k * 64u] : make_uint4(0, 0, 0, 0);
	};
	auto pack_finish = [&](u32 row, const uint4 (&v)[4]) {
		u32 *dst = in_bits + row * C + (lane >> 3);
#pragma unroll
		for (int k = 0; k < 4; k++)
		{
			u32 n = (v[k].x != 0u ? 1u : 0u) | (v[k].y != 0u ? 2u : 0u) | (v[k].z != 0u ? 4u : 0u) | (v[k].w != 0u ? 8u : 0u);
			n |= (u32)__shfl_down((int)n, 1) << 4; // through the LDS crossbar: DPP moves here (as in ca_unpacked_ballot) measure 2 % slower at 512^3 — they sit on the VALU path the stores wait for
			n |= (u32)__shfl_down((int)n, 2) << 8;
			n |= (u32)__shfl_down((int)n, 4) << 16;
			if ((lane & 7u) == 0 && (u32)k < X256) dst[(u32)k * 8u] = n;
		}
	};
	// row `r` (plane-in-group, y) of output group `grp` -> one u32 per cell
	auto unpack_row = [&](u32 grp, u32 r) {
		const u32 rz = grp * SUB + r / kBTY, ry = r % kBTY;
		const int gy = y0 + (int)ry, gz = z0 + (int)rz;
		if (gy >= (int)G || gz >= (int)pr.hi) return;
		uint4 *drow = reinterpret_cast<uint4 *>(out + (size_t)gz * plane_cells + (size_t)gy * G);
		const u32 *srcb = out_bits + ((grp & 1u) * SUB * kBTY + r) * C + (lane >> 3);
		const u32 sh = (lane & 7u) * 4u;
		for (u32 xc = 0; xc < X256; xc++)
		{
			const u32 w = srcb[xc * 8u] >> sh;
			uint4 v;
			v.x = w & 1u; v.y = (w >> 1) & 1u; v.z = (w >> 2) & 1u; v.w = (w >> 3) & 1u;
			{
				// non-temporal: the output is not read again before the next step, and 512 MiB and more of it would only push the
				// halo rows the neighbour tiles are about to read out of L2 / the Infinity Cache (512^3: 233 -> 217 us per step, same
				// box; non-temporal LOADS on top: 222 — the halo rows are read twice)
				typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
				u32x4_t nv;
				nv.x = v.x; nv.y = v.y; nv.z = v.z; nv.w = v.w;
				__builtin_nontemporal_store(nv, reinterpret_cast<u32x4_t *>(drow) + xc * 64u + lane);
			}
		}
	};

	// ---- prologue: the input planes group 0 needs (zz = 0 .. SUB + 1), two rows in flight per wave
	{
		const u32 nrows = (SUB + 2u) * RY;
		for (u32 row = wave; row < nrows; row += 2u * NW)
		{
			uint4 a[4], b[4];
			pack_issue(row, a);
			const bool two = row + NW < nrows;
			if (two) pack_issue(row + NW, b);
			pack_finish(row, a);
			if (two) pack_finish(row + NW, b);
		}
	}
	__syncthreads();

	FastRules<MAIN, false, false> frules;
	if (FAST) frules = expand_rules<MAIN, false, false>(rules_in);
	const u32 CV = C / 4u;
	for (u32 grp = 0; grp < K; grp++)
	{
		// ---- update group grp: output planes rz = grp * SUB .. + SUB - 1 from image planes rz .. rz + 2
		const u32 nitems = (SUB * kBTY) << cv_shift;
		for (u32 it = tid; it < nitems; it += (u32)kBThreads)
		{
			const u32 cxv = it & (CV - 1u), rrow = it >> cv_shift;
			const u32 ry = rrow % kBTY, rz = grp * SUB + rrow / kBTY;
			const u32 cx0 = cxv * 4u;
			auto seg = [&](u32 zz, u32 yy) {
				SegT<4> sg;
				const u32 *row = in_bits + (zz * RY + yy) * C;
				const uint4 w = *reinterpret_cast<const uint4 *>(row + cx0);
				sg.w[0] = w.x; sg.w[1] = w.y; sg.w[2] = w.z; sg.w[3] = w.w;
				sg.lo = row[cx0 == 0 ? C - 1u : cx0 - 1u];
				sg.hi = row[cx0 + 4u == C ? 0u : cx0 + 4u];
				return sg;
			};
			PlaneRowsT<4> P[3];
#pragma unroll
			for (int dz = 0; dz < 3; dz++)
			{
				P[dz].ym = seg(rz + (u32)dz, ry);
				P[dz].c = seg(rz + (u32)dz, ry + 1u);
				P[dz].yp = seg(rz + (u32)dz, ry + 2u);
			}
			u32 o[4];
			if (FAST) evolve<4, MAIN, false, false>(P[0], P[1], P[2], 0xFFFFFFFFu, frules, o);
			else evolve<4, MAIN, false, false>(P[0], P[1], P[2], 0xFFFFFFFFu, rules_in, o);
			uint4 ov;
			ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
			*reinterpret_cast<uint4 *>(out_bits + ((grp & 1u) * SUB * kBTY + rrow) * C + cx0) = ov;
		}
		__syncthreads();
		// ---- store group grp while the input planes of group grp + 1 (zz = SUB * (grp + 1) + 2 .. + SUB - 1 more) are loaded
		const u32 pack_lo = (SUB * (grp + 1u) + 2u) * RY, pack_hi = grp + 1u < K ? pack_lo + SUB * RY : pack_lo;
		const u32 nun = SUB * kBTY;
		for (u32 i = wave;; i += 2u * NW)
		{
			// two rows of loads in flight per wave, two rows of stores under them
			const u32 p0 = pack_lo + i, p1 = p0 + NW;
			const bool pk0 = p0 < pack_hi, pk1 = p1 < pack_hi, un0 = i < nun, un1 = i + NW < nun;
			if (!pk0 && !un0) break;
			uint4 a[4], b[4];
			if (pk0) pack_issue(p0, a);
			if (pk1) pack_issue(p1, b);
			if (un0) unpack_row(grp, i);
			if (un1) unpack_row(grp, i + NW);
			if (pk0) pack_finish(p0, a);
			if (pk1) pack_finish(p1, b);
		}
		__syncthreads();
	}
}

template <int MAIN, bool FAST, int kBTZ, int kSub>
hipError_t launch_pipe_fzs(const UnpackedLaunch &l, hipStream_t stream, const PackedRuleArgs &prog)
{
	const u32 G = l.pr.G, C = G / 32u;
	const u32 ny = (G + kBTY - 1) / kBTY, nz = (l.pr.hi - l.pr.lo + kBTZ - 1) / kBTZ;
	const size_t lds_bytes = ((size_t)(kBTZ + 2) * (kBTY + 2) + (size_t)2 * kSub * kBTY) * C * sizeof(u32);
	auto kern = ca_unpacked_pipe<MAIN, FAST, kBTZ, kSub>;
	u32 cv_shift = 0;
	while ((1u << cv_shift) < C / 4u) cv_shift++;
	static std::mutex attr_mutex;
	static uint64_t attr_devices = 0;
	{
		int dev = 0;
		hipError_t e = hipGetDevice(&dev);
		if (e != hipSuccess) return e;
		std::lock_guard<std::mutex> lock(attr_mutex);
		if (dev < 0 || dev >= 64 || !(attr_devices >> dev & 1u))
		{
			e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
			if (e != hipSuccess) return e;
			if (dev >= 0 && dev < 64) attr_devices |= 1ull << dev;
		}
	}
	hipLaunchKernelGGL(kern, dim3(ny * nz), dim3(kBThreads), lds_bytes, stream, l.in, l.out, l.pr, ny, cv_shift, prog);
	return hipGetLastError();
}

template <int MAIN, bool FAST, int kBTZ>
hipError_t launch_pipe_fz(const UnpackedLaunch &l, hipStream_t stream, const PackedRuleArgs &prog)
{
	static const int sub_env = getenv("CA3D_UNPACKED_SUB") ? atoi(getenv("CA3D_UNPACKED_SUB")) : 4; // planes per group (tuning)
	if (sub_env == 2) return launch_pipe_fzs<MAIN, FAST, kBTZ, 2>(l, stream, prog);
	if (sub_env == 8) return launch_pipe_fzs<MAIN, FAST, kBTZ, 8>(l, stream, prog);
	return launch_pipe_fzs<MAIN, FAST, kBTZ, 4>(l, stream, prog);
}

template <int MAIN, bool FAST, int kBTZ>
hipError_t launch_ballot_fz(const UnpackedLaunch &l, hipStream_t stream, const PackedRuleArgs &prog)
{
	const u32 G = l.pr.G, C = G / 32u;
	const u32 ny = (G + kBTY - 1) / kBTY, nz = (l.pr.hi - l.pr.lo + kBTZ - 1) / kBTZ;
	const size_t lds_bytes = ((size_t)(kBTZ + 2) * (kBTY + 2) + (size_t)kBTZ * kBTY) * C * sizeof(u32);
	if (lds_bytes > 160u * 1024u) return hipErrorInvalidValue;
	auto kern = ca_unpacked_ballot<MAIN, FAST, kBTZ>;
	u32 cv_shift = 0;
	while ((1u << cv_shift) < C / 4u) cv_shift++;
	// the opt-in to > 64 KiB of dynamic LDS belongs to the function object of the CURRENT device: once per (device,
	// instantiation), under a lock (engines on different devices / threads share this code)
	static std::mutex attr_mutex;
	static uint64_t attr_devices = 0; // bit d: set on device d
	{
		int dev = 0;
		hipError_t e = hipGetDevice(&dev);
		if (e != hipSuccess) return e;
		std::lock_guard<std::mutex> lock(attr_mutex);
		if (dev < 0 || dev >= 64 || !(attr_devices >> dev & 1u))
		{
			e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
			if (e != hipSuccess) return e;
			if (dev >= 0 && dev < 64) attr_devices |= 1ull << dev;
		}
	}
	// dwordx4 loads in flight per wave in the pack phase: 8 (eight rows at once) on rows of one load (G = 256: 37 -> 31.5 us per
	// step), the row-at-a-time form on longer rows (512^3: 262 us against 268 with 8 — the phase is not latency-bound there: all
	// workgroups of the launch are resident at once and run pack / update / unpack in step, so reads and writes never overlap;
	// profiles/r3_e_unpacked_pack_depth.txt). CA3D_UNPACKED_LOADS overrides (0: row at a time).
	static const int pack_env = getenv("CA3D_UNPACKED_LOADS") ? atoi(getenv("CA3D_UNPACKED_LOADS")) : -1;
	const u32 pack_loads = pack_env >= 0 ? (u32)pack_env : (G <= 256u ? 8u : 0u);
	hipLaunchKernelGGL(kern, dim3(ny * nz), dim3(kBThreads), lds_bytes, stream, l.in, l.out, l.pr, ny, cv_shift, prog, pack_loads);
	return hipGetLastError();
}

template <int MAIN, bool FAST>
hipError_t launch_ballot_f(const UnpackedLaunch &l, hipStream_t stream, const PackedRuleArgs &prog)
{
	// 16-plane tiles re-read 1.2x (8-plane tiles 1.33x) but need >= 2 workgroups per CU to keep HBM busy
	const u32 tiles16 = ((l.pr.G + kBTY - 1) / kBTY) * ((l.pr.hi - l.pr.lo + 15u) / 16u);
	// the pipelined form (loads of the next plane group under the stores of this one) where rows are whole dwordx4 wave-loads
	static const int pipe_env = getenv("CA3D_UNPACKED_PIPE") ? atoi(getenv("CA3D_UNPACKED_PIPE")) : 1;
	// Tiles of 32 planes where they fit (round 4): the first group's loads and the last group's stores of a tile have nothing to overlap
	// with, and a tile twice as deep has half as many of them per plane, and 1.13 x instead of 1.19 x the cells read — 512^3: 219 -> 209 us
	// per step (0.61 -> 0.64 of 8 TB/s), what the phase model of DESIGN 4.4 priced. One workgroup per CU then (90 KB of LDS): the
	// range must divide into 32-plane tiles, give every CU one, and the row must be short enough for the image (512^3; 1024^3 keeps 16).
	// CA3D_UNPACKED_TZ = 16 keeps the 16-plane tiles (tuning).
	static const int tz_env = getenv("CA3D_UNPACKED_TZ") ? atoi(getenv("CA3D_UNPACKED_TZ")) : 32;
	const u32 planes = l.pr.hi - l.pr.lo, C32 = l.pr.G / 32u;
	const size_t lds32 = ((size_t)34u * (kBTY + 2) + (size_t)2 * 8u * kBTY) * C32 * sizeof(u32); // (up to 8 planes per group)
	if (pipe_env && tz_env == 32 && l.pr.G >= 512u && planes % 32u == 0 && ((l.pr.G + kBTY - 1) / kBTY) * (planes / 32u) >= 256u && lds32 <= 160u * 1024u)
		return launch_pipe_fz<MAIN, FAST, 32>(l, stream, prog);
	if (pipe_env && l.pr.G >= 512u) // (256^3: 35 us against 31.5 — a tile is 8 planes there, two groups: too little to overlap)
		return tiles16 >= 512u ? launch_pipe_fz<MAIN, FAST, 16>(l, stream, prog) : launch_pipe_fz<MAIN, FAST, 8>(l, stream, prog);
	return tiles16 >= 512u ? launch_ballot_fz<MAIN, FAST, 16>(l, stream, prog) : launch_ballot_fz<MAIN, FAST, 8>(l, stream, prog);
}

template <int MAIN>
hipError_t launch_ballot(const UnpackedLaunch &l, hipStream_t stream, const PackedRuleArgs &prog)
{
	const bool fast = prog.set[0].born.n <= (u32)kFastCubes && prog.set[0].survive.n <= (u32)kFastCubes;
	return fast ? launch_ballot_f<MAIN, true>(l, stream, prog) : launch_ballot_f<MAIN, false>(l, stream, prog);
}

bool ballot_applies(const UnpackedLaunch &l)
{
	const CanonRules &r = *l.rules;
	const u32 G = l.pr.G;
	if (!l.binary_state || !r.unpacked_fast) return false;
	if (G < 128u || (G & (G - 1u))) return false; // torus only for power-of-two G; rows of >= 4 words
	const size_t lds_bytes = ((size_t)(16 + 2) * (kBTY + 2) + (size_t)16 * kBTY) * (G / 32u) * sizeof(u32);
	return lds_bytes <= 160u * 1024u;
}

} // namespace

hipError_t launch_unpacked_step(const UnpackedLaunch &l, hipStream_t stream, const char **kernel_name)
{
	if (l.pr.hi <= l.pr.lo) return hipSuccess;
	if (ballot_applies(l))
	{
		if (kernel_name) *kernel_name = "ca_unpacked_ballot";
		PackedRuleArgs prog{};
		prog.set[0] = l.rules->unpacked_prog; // rule-sets 1 and 2 stay constant-false: the legacy kernel has one rule-set
		switch (l.rules->main)
		{
		case MAIN_VN: return launch_ballot<MAIN_VN>(l, stream, prog);
		case MAIN_VN2D: return launch_ballot<MAIN_VN2D>(l, stream, prog);
		case MAIN_MOORE: return launch_ballot<MAIN_MOORE>(l, stream, prog);
		case MAIN_MOORE2D: return launch_ballot<MAIN_MOORE2D>(l, stream, prog);
		case MAIN_EDGES: return launch_ballot<MAIN_EDGES>(l, stream, prog);
		case MAIN_CORNERS: return launch_ballot<MAIN_CORNERS>(l, stream, prog);
		default: break;
		}
	}
	if (kernel_name) *kernel_name = "ca_unpacked_literal";
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
