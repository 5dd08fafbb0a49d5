// The headline kernel: one packed CA step for the von Neumann neighbourhood with the edges / corners rule-sets
// switched off (the reference's default rule, compute_clustered.wgsl with "27" edge / corner strings) on a
// power-of-two grid. Same data path as ca_packed_class (ca_packed.hip: one uint4 column position per thread, ZR
// planes per thread held in registers, every load issued before the first use), specialised where the general
// kernel spends its instruction budget — measured on MI355X the 512^3 step is a latency chain whose exposed part
// is the instruction stream itself (tools/ubench/stencil_floor.hip: the bare load / store pattern takes 5.4 us,
// the general kernel 6.8 us):
//   * the grid edge is a template parameter: every index, mask and row offset is a shift or an immediate, plane
//     bases are scalar, row offsets are three 32-bit VGPRs;
//   * the words either side of a segment come from the neighbour lanes by DPP moves (no LDS round trip);
//   * the rule is two 8-entry truth tables over the three count planes (survive, born). v_bitop3_b32 evaluates any
//     3-input truth table in ONE instruction but takes the table as an immediate, so the kernel carries all 256
//     of them behind a wave-uniform jump: 3 VALU per 32 cells for the whole rule instead of ~22 through the cube
//     programs of ca_bitslice.inc.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ca3d_internal.h"

namespace ca3d
{
namespace
{
#include "ca_bitslice.inc"

struct VnArgs
{
	u32 lo, hi, nplanes, wrap_full; // PlaneRange without G
	u32 lo2, hi2, runs1;            // second output range; z-runs of the first one
	int zbase;
	u32 lut_s, lut_b; // bit k: a cell with k live von Neumann neighbours survives / is born (k = 0 .. 6)
	u32 nt;           // non-temporal stores
};

// DPP controls (gfx9 encoding)
constexpr int kDppWaveShl1 = 0x130; // lane i <- lane i + 1
constexpr int kDppWaveRol1 = 0x134; // lane i <- lane (i + 1) % 64
constexpr int kDppWaveShr1 = 0x138; // lane i <- lane i - 1
template <int CTRL>
__device__ __forceinline__ u32 dpp_mov(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }

// Word 0 of the lane that starts this lane's row (CV lanes per row), valid in the row's last lane.
template <int CVL>
__device__ __forceinline__ u32 row_first(u32 w0, int lane)
{
	if (CVL == 1) return dpp_mov<0 | (0 << 2) | (2 << 4) | (2 << 6)>(w0);  // quad_perm:[0,0,2,2]
	if (CVL == 2) return dpp_mov<0>(w0);                                     // quad_perm:[0,0,0,0]
	if (CVL == 3) return dpp_mov<0x110 + 7>(w0);                             // row_shr:7
	if (CVL == 4) return dpp_mov<0x110 + 15>(w0);                            // row_shr:15
	if (CVL == 5) return (u32)__shfl((int)w0, lane & 32);                    // 32 lanes per row: no DPP pattern
	return dpp_mov<kDppWaveRol1>(w0);                                        // the row is the wave
}

// o[i] = LUT[p2 p1 p0] for N words. IMM index of v_bitop3 = a << 2 | b << 1 | c.
#define CA3D_L3(n) \
	case n: \
		_Pragma("unroll") for (int i = 0; i < N; i++) o[i] = bitop3<(n)>(p[i][2], p[i][1], p[i][0]); \
		break;
#define CA3D_L3x4(n) CA3D_L3(n) CA3D_L3(n + 1) CA3D_L3(n + 2) CA3D_L3(n + 3)
#define CA3D_L3x16(n) CA3D_L3x4(n) CA3D_L3x4(n + 4) CA3D_L3x4(n + 8) CA3D_L3x4(n + 12)
#define CA3D_L3x64(n) CA3D_L3x16(n) CA3D_L3x16(n + 16) CA3D_L3x16(n + 32) CA3D_L3x16(n + 48)
template <int N>
__device__ __forceinline__ void lut3(u32 lut, const u32 (&p)[N][3], u32 (&o)[N])
{
	switch (lut & 0xFFu)
	{
		CA3D_L3x64(0) CA3D_L3x64(64) CA3D_L3x64(128) CA3D_L3x64(192)
	}
}

// LS / LB >= 0: the survive / born tables are compile-time constants (pre-built specialisation, no dispatch);
// -1: taken from the arguments through lut3's jump.
template <int CVL, int ZR, int LS, int LB>
__global__ __launch_bounds__(256) void ca_packed_vn(const u32 *__restrict__ in, u32 *__restrict__ out, VnArgs a)
{
	constexpr u32 CV = 1u << CVL;       // uint4 per row
	constexpr u32 G = 128u * CV;
	constexpr u32 PLANE = G * CV;       // uint4 per plane
	constexpr u32 TPP = PLANE / 256u;   // 256-thread tiles per plane (>= 2 for CVL >= 1)
	constexpr u32 PLANE_BYTES = PLANE * 16u;

	// XCD-aware block order: XCD k gets the k-th contiguous eighth of the (z-run, tile) space
	const u32 b = blockIdx.x, nb = gridDim.x;
	const u32 v = (nb & 7u) == 0 ? (b & 7u) * (nb >> 3) + (b >> 3) : b;
	const u32 zr = v / TPP, tile = v % TPP;
	const u32 t = tile * 256u + threadIdx.x; // uint4 index inside a plane = y * CV + cxv
	// first output plane of this thread; the last run of an odd range is shifted down to end at `hi` (the overlap
	// is computed twice with identical results) so that the body has no tail guard
	const bool second = zr >= a.runs1; // wave-uniform
	const u32 j0 = second ? min(a.lo2 + (zr - a.runs1) * ZR, a.hi2 - ZR) : min(a.lo + zr * ZR, a.hi - ZR);

	// Scalar plane bases. Own planes j0 .. j0+ZR-1 are always inside the array; only the plane below the run
	// (clamped at plane 0, where it is only ever read masked: global z == 0) and the plane above it (wraps to
	// plane 0 on the full grid, clamps on the last plane of a slab, whose top ghost is never a valid output) need care.
	const char *own = reinterpret_cast<const char *>(in) + (size_t)j0 * PLANE_BYTES;
	const char *below = own - (j0 != 0 ? PLANE_BYTES : 0u);
	const char *above = j0 + ZR < a.nplanes ? own + (size_t)ZR * PLANE_BYTES
	                                        : (a.wrap_full ? reinterpret_cast<const char *>(in) : own + (size_t)(ZR - 1) * PLANE_BYTES);
	const u32 zg0 = (u32)(a.zbase + (int)j0) & (G - 1u); // global z of plane j0

	// row offsets in bytes: y-1 (clamped: masked when y == 0), y, y+1 (wraps to row 0)
	const u32 oc = t * 16u;
	const u32 om = (u32)max((int)oc - (int)(CV * 16u), 0);
	const u32 op = (oc + CV * 16u) & (PLANE_BYTES - 1u);

	// every load of the thread, back to back, in the order the planes are consumed
	uint4 c[ZR + 2], m[ZR], p[ZR];
	c[0] = *reinterpret_cast<const uint4 *>(below + oc);
	c[1] = *reinterpret_cast<const uint4 *>(own + oc);
#pragma unroll
	for (int q = 1; q <= ZR; q++)
	{
		const char *pl = own + (size_t)(q - 1) * PLANE_BYTES;
		m[q - 1] = *reinterpret_cast<const uint4 *>(pl + om);
		p[q - 1] = *reinterpret_cast<const uint4 *>(pl + op);
		c[q + 1] = *reinterpret_cast<const uint4 *>((q < ZR ? pl + PLANE_BYTES : above) + oc);
		__builtin_amdgcn_sched_barrier(0); // keep the issue order: plane q's inputs land before plane q+1's
	}

	const int lane = (int)(threadIdx.x & 63u);
	const u32 cxv = t & (CV - 1u);
	u32 ymask = t < CV ? 0u : 0xFFFFFFFFu;  // row y-1 is dead at y == 0
	u32 lomask = cxv == 0 ? 0u : 0xFFFFFFFFu; // x-1 is dead at x == 0
	asm volatile("" : "+v"(ymask), "+v"(lomask)); // keep them masks (v_and), not per-word selects
	const bool last = cxv == CV - 1u;
	// Only the wave that holds row 0 of a plane has a dead y-1 row and only plane z == 0 a dead z-1 plane: both
	// are handled under wave-uniform branches (the empty asm keeps the compiler from turning them back into
	// per-word selects), so all other waves spend no VALU on boundary masks.
	const bool wave_has_row0 = __builtin_amdgcn_readfirstlane((int)t) < 64;

	u32 cnt[ZR * 4][3], self[ZR * 4];
#pragma unroll
	for (int q = 1; q <= ZR; q++)
	{
		const bool below_dead = ((zg0 + (u32)q - 1u) & (G - 1u)) == 0u; // z-1 == -1 is dropped (compute_clustered.wgsl:104)
		const u32 w[4] = {c[q].x, c[q].y, c[q].z, c[q].w};
		u32 wm[4] = {m[q - 1].x, m[q - 1].y, m[q - 1].z, m[q - 1].w};
		const u32 wp[4] = {p[q - 1].x, p[q - 1].y, p[q - 1].z, p[q - 1].w};
		u32 wb[4] = {c[q - 1].x, c[q - 1].y, c[q - 1].z, c[q - 1].w};
		const u32 wa[4] = {c[q + 1].x, c[q + 1].y, c[q + 1].z, c[q + 1].w};
		if (__builtin_expect(wave_has_row0, 0))
		{
			asm volatile("");
#pragma unroll
			for (int i = 0; i < 4; i++) wm[i] &= ymask;
		}
		if (__builtin_expect(below_dead, 0))
		{
			asm volatile("");
#pragma unroll
			for (int i = 0; i < 4; i++) wb[i] = 0u;
		}
		const u32 lo = dpp_mov<kDppWaveShr1>(w[3]) & lomask;
		const u32 nxt = dpp_mov<kDppWaveShl1>(w[0]);
		// both sources are taken with every lane active (a DPP move cannot read a lane that EXEC has switched
		// off, so the exchange must not sit in a divergent branch), then selected per lane
		const u32 first = row_first<CVL>(w[0], lane);
		const u32 hi = CVL == 6 ? first : (last ? first : nxt);
#pragma unroll
		for (int i = 0; i < 4; i++)
		{
			const u32 l = from_left(w[i], i ? w[i > 0 ? i - 1 : 0] : lo);
			const u32 r = from_right(i < 3 ? w[i < 3 ? i + 1 : 0] : hi, w[i]);
			sum6(l, r, wp[i], wm[i], wa[i], wb[i], cnt[(q - 1) * 4 + i]);
			self[(q - 1) * 4 + i] = w[i];
		}
	}

	u32 S[ZR * 4], B[ZR * 4];
	if (LS >= 0)
	{
#pragma unroll
		for (int i = 0; i < ZR * 4; i++)
		{
			S[i] = bitop3<(LS & 0xFF)>(cnt[i][2], cnt[i][1], cnt[i][0]);
			B[i] = bitop3<(LB & 0xFF)>(cnt[i][2], cnt[i][1], cnt[i][0]);
		}
	}
	else
	{
		lut3<ZR * 4>(a.lut_s, cnt, S);
		lut3<ZR * 4>(a.lut_b, cnt, B);
	}

	typedef u32 u32x4 __attribute__((ext_vector_type(4)));
	char *dst = reinterpret_cast<char *>(out) + (size_t)j0 * PLANE_BYTES + oc;
	u32x4 res[ZR];
#pragma unroll
	for (int q = 0; q < ZR; q++)
		res[q] = u32x4{next_state(self[q * 4], S[q * 4], B[q * 4]), next_state(self[q * 4 + 1], S[q * 4 + 1], B[q * 4 + 1]),
		               next_state(self[q * 4 + 2], S[q * 4 + 2], B[q * 4 + 2]), next_state(self[q * 4 + 3], S[q * 4 + 3], B[q * 4 + 3])};
	if (a.nt)
	{
#pragma unroll
		for (int q = 0; q < ZR; q++) __builtin_nontemporal_store(res[q], reinterpret_cast<u32x4 *>(dst + (size_t)q * PLANE_BYTES));
	}
	else
	{
#pragma unroll
		for (int q = 0; q < ZR; q++) *reinterpret_cast<u32x4 *>(dst + (size_t)q * PLANE_BYTES) = res[q];
	}
}

// Tables with a pre-built specialisation: the reference UI's start-up rule, von Neumann B1,3 / S0-6
// (survive slots 0..6, born slots 1 and 3 of the main rule-set).
constexpr int kDefaultS = 0xFF, kDefaultB = 0x0A; // canonical form: see launch_packed_vn

template <int CVL, int ZR>
hipError_t launch(const PackedLaunch &l, VnArgs a, hipStream_t stream)
{
	a.runs1 = (l.pr.hi - l.pr.lo + ZR - 1u) / ZR;
	constexpr u32 CV = 1u << CVL, TPP = 128u * CV * CV / 256u;
	const u32 blocks = TPP * (a.runs1 + (l.pr.hi2 > l.pr.lo2 ? (l.pr.hi2 - l.pr.lo2 + ZR - 1u) / ZR : 0u));
	if (a.lut_s == (u32)kDefaultS && a.lut_b == (u32)kDefaultB)
		hipLaunchKernelGGL((ca_packed_vn<CVL, ZR, kDefaultS, kDefaultB>), dim3(blocks), dim3(256), 0, stream, l.in, l.out, a);
	else
		hipLaunchKernelGGL((ca_packed_vn<CVL, ZR, -1, -1>), dim3(blocks), dim3(256), 0, stream, l.in, l.out, a);
	return hipGetLastError();
}

int log2_exact(u32 x)
{
	if (!x || (x & (x - 1u))) return -1;
	int s = 0;
	while ((1u << s) < x) s++;
	return s;
}

} // namespace

// The rule reduces to two truth tables over the von Neumann count and the grid is a power of two in [256, 8192].
bool vn_kernel_applies(const CanonRules &r, uint32_t G, int variant)
{
	if ((variant & 0xFF) == 1 || !r.fast || r.main != MAIN_VN) return false;
	for (int s = 1; s < 3; s++)
	{
		const uint32_t reachable = (2u << r.lists.n[s]) - 1u;
		if ((r.onset_born[s] | r.onset_survive[s]) & reachable) return false;
	}
	const int cvl = log2_exact(G / 128u);
	return G % 128u == 0 && cvl >= 1 && cvl <= 6;
}

hipError_t launch_packed_vn(const PackedLaunch &l, hipStream_t stream)
{
	const CanonRules &r = *l.rules;
	const bool two = l.pr.hi2 > l.pr.lo2;
	const u32 G = l.pr.G, planes = l.pr.hi - l.pr.lo + (two ? l.pr.hi2 - l.pr.lo2 : 0u);
	const u32 shortest = two ? (l.pr.hi - l.pr.lo < l.pr.hi2 - l.pr.lo2 ? l.pr.hi - l.pr.lo : l.pr.hi2 - l.pr.lo2) : l.pr.hi - l.pr.lo;
	const int cvl = log2_exact(G / 128u);
	VnArgs a;
	a.lo = l.pr.lo; a.hi = l.pr.hi; a.nplanes = l.pr.nplanes; a.wrap_full = l.pr.wrap_full; a.zbase = l.pr.zbase;
	a.lo2 = l.pr.lo2; a.hi2 = two ? l.pr.hi2 : l.pr.lo2; a.runs1 = 0;
	// Entry 7 of a table is never read (a cell has at most 6 von Neumann neighbours): give it the value that makes
	// the table constant when the other seven agree, so the specialised kernels can drop the evaluation.
	a.lut_s = r.onset_survive[0] & 0x7Fu;
	a.lut_b = r.onset_born[0] & 0x7Fu;
	if (a.lut_s == 0x7Fu) a.lut_s = 0xFFu;
	if (a.lut_b == 0x7Fu) a.lut_b = 0xFFu;
	// Non-temporal stores pay only while both ping-pong buffers sit in the 256 MiB Infinity Cache with room to
	// spare (measured: 6.9 vs 7.5 us per step at 512^3, 58 vs 43 us at 1024^3).
	a.nt = (size_t)l.pr.nplanes * G * (G / 32u) * sizeof(u32) <= (16u << 20) ? 1u : 0u;
	// 2 planes per thread once that still fills the chip (>= 1024 workgroups), else 1
	const u32 tpp = G / 128u * G / 256u;
	const bool deep = shortest >= 2u && (size_t)tpp * ((planes + 1u) / 2u) >= 1024u;
	switch (cvl)
	{
	case 1: return launch<1, 1>(l, a, stream);
	case 2: return deep ? launch<2, 2>(l, a, stream) : launch<2, 1>(l, a, stream);
	case 3: return deep ? launch<3, 2>(l, a, stream) : launch<3, 1>(l, a, stream);
	case 4: return deep ? launch<4, 2>(l, a, stream) : launch<4, 1>(l, a, stream);
	case 5: return deep ? launch<5, 2>(l, a, stream) : launch<5, 1>(l, a, stream);
	case 6: return deep ? launch<6, 2>(l, a, stream) : launch<6, 1>(l, a, stream);
	default: return hipErrorInvalidValue;
	}
}

} // namespace ca3d
