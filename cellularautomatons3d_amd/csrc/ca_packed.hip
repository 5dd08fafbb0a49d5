// Bit-packed CA step for gfx950 (CDNA4): 32 x-adjacent cells per u32, three rule-sets, the asymmetric
// boundary of the reference kernel (shaders/compute_clustered.wgsl:56-272):
//     P(x,y,z) = 0 if any coordinate is -1, else S(x mod G, y mod G, z mod G).
//
// Not a translation of the WGSL (one invocation per word, 32 serial cells, up to 46 scalar lookups per cell):
// every thread owns 128 cells (one dwordx4), neighbour counts are bit-sliced — 32 cells per VALU op through
// carry-save adders built from v_bitop3_b32 / v_alignbit_b32 — and the rule LUT never reaches the GPU: the
// host compiles it into an OR-of-cubes program over the count bit-planes (rules.cpp) that is interpreted with
// wave-uniform control flow.
//
// The 26 neighbours split into the classes Fx Fy Fz (faces), Exy Exz Eyz (edges), Cn (corners); every table
// of main_pathtraced.js:13-85 is a union of them, so the class kernels are templated on <main table, edges
// rule-set live, corners rule-set live>. Arbitrary offset lists (legal through the ABI, never produced by the
// reference host) and grids whose row is not a multiple of 4 words take the generic per-offset kernel.
#include "ca3d_internal.h"

namespace ca3d
{
namespace
{

using u32 = uint32_t;

constexpr u32 TA = 0xF0, TB = 0xCC, TC = 0xAA; // truth-table columns of the three v_bitop3 inputs

template <u32 IMM>
__device__ __forceinline__ u32 bitop3(u32 a, u32 b, u32 c)
{
	return __builtin_amdgcn_bitop3_b32(a, b, c, IMM);
}
__device__ __forceinline__ u32 xor3(u32 a, u32 b, u32 c) { return bitop3<(TA ^ TB ^ TC)>(a, b, c); }
__device__ __forceinline__ u32 maj3(u32 a, u32 b, u32 c) { return bitop3<((TA & TB) | (TA & TC) | (TB & TC))>(a, b, c); }
// bit i of the result = bit i-1 of the row (neighbour at x-1): (w << 1) | (lower >> 31)
__device__ __forceinline__ u32 from_left(u32 w, u32 lower) { return __builtin_amdgcn_alignbit(w, lower, 31); }
// bit i of the result = bit i+1 of the row (neighbour at x+1): (w >> 1) | (upper << 31)
__device__ __forceinline__ u32 from_right(u32 upper, u32 w) { return __builtin_amdgcn_alignbit(upper, w, 1); }

__device__ __forceinline__ void fa(u32 a, u32 b, u32 c, u32 &s, u32 &k) { s = xor3(a, b, c); k = maj3(a, b, c); }
__device__ __forceinline__ void ha(u32 a, u32 b, u32 &s, u32 &k) { s = a ^ b; k = a & b; }

// ---- sums of 1-bit planes (carry-save trees) -------------------------------------------------------------
__device__ __forceinline__ void sum4(u32 a, u32 b, u32 c, u32 d, u32 (&p)[3])
{
	u32 s, k, k2;
	fa(a, b, c, s, k);
	ha(s, d, p[0], k2);
	ha(k, k2, p[1], p[2]);
}

__device__ __forceinline__ void sum6(u32 a, u32 b, u32 c, u32 d, u32 e, u32 f, u32 (&p)[3])
{
	u32 s0, k0, s1, k1, c0;
	fa(a, b, c, s0, k0);
	fa(d, e, f, s1, k1);
	ha(s0, s1, p[0], c0);
	fa(k0, k1, c0, p[1], p[2]);
}

__device__ __forceinline__ void sum8(const u32 (&x)[8], u32 (&p)[4])
{
	u32 s0, k0, s1, k1, s2, k2, k3, t, u, v;
	fa(x[0], x[1], x[2], s0, k0);
	fa(x[3], x[4], x[5], s1, k1);
	ha(x[6], x[7], s2, k2);
	fa(s0, s1, s2, p[0], k3);
	fa(k0, k1, k2, t, u);
	ha(t, k3, p[1], v);
	ha(u, v, p[2], p[3]);
}

__device__ __forceinline__ void sum12(const u32 (&x)[12], u32 (&p)[4])
{
	u32 s0, s1, s2, s3, k0, k1, k2, k3, t, u, v, a, b, c, d, e;
	fa(x[0], x[1], x[2], s0, k0);
	fa(x[3], x[4], x[5], s1, k1);
	fa(x[6], x[7], x[8], s2, k2);
	fa(x[9], x[10], x[11], s3, k3);
	fa(s0, s1, s2, t, u);
	ha(t, s3, p[0], v);
	fa(k0, k1, k2, a, b);
	fa(k3, u, v, c, d);
	ha(a, c, p[1], e);
	fa(b, d, e, p[2], p[3]);
}

// vn (0..6, 3 planes) + corners (0..8, 4 planes) + edges (0..12, 4 planes) -> 0..26, 5 planes
__device__ __forceinline__ void sum_moore(const u32 (&vn)[3], const u32 (&ed)[4], const u32 (&co)[4], u32 (&p)[5])
{
	u32 q0, q1, q2, q3, c;
	ha(vn[0], co[0], q0, c);
	fa(vn[1], co[1], c, q1, c);
	fa(vn[2], co[2], c, q2, c);
	q3 = co[3] ^ c; // <= 14: no carry out of plane 3
	ha(q0, ed[0], p[0], c);
	fa(q1, ed[1], c, p[1], c);
	fa(q2, ed[2], c, p[2], c);
	fa(q3, ed[3], c, p[3], p[4]);
}

// ---- rule program interpreter ----------------------------------------------------------------------------
// OR over cubes of AND over cared planes of (plane == required value); wave-uniform control flow only.
template <int NP>
__device__ __forceinline__ u32 eval_prog(const RuleProg &pr, const u32 *p)
{
	u32 acc = 0;
	for (u32 c = 0; c < pr.n; c++)
	{
		const u32 cube = pr.cubes[c];
		u32 e = 0xFFFFFFFFu;
#pragma unroll
		for (int i = 0; i < NP; i++)
		{
			if (cube & (1u << i))
			{
				const u32 inv = (cube & (0x100u << i)) ? 0u : 0xFFFFFFFFu;
				e = bitop3<(TA & (TB ^ TC))>(e, p[i], inv);
			}
		}
		acc |= e;
	}
	return acc ^ pr.invert;
}

__device__ __forceinline__ u32 next_state(u32 alive, u32 S, u32 B)
{
	return bitop3<((TA & TB) | (~TA & TC)) & 0xFF>(alive, S, B); // alive ? survive : born
}

// ---- plane / row addressing shared by all packed kernels -----------------------------------------------
struct Nbr
{
	const u32 *plane[3]; // z-1, z, z+1 (dead planes point at a valid plane and carry mask 0)
	u32 zmask[3];
	u32 yrow[3]; // y-1, y, y+1 (clamped / wrapped)
	u32 ymask[3];
};

__device__ __forceinline__ Nbr neighbours(const u32 *in, const PlaneRange &pr, u32 C, u32 j, u32 y)
{
	Nbr n;
	const size_t plane_words = (size_t)C * pr.G;
	int zg = (pr.zbase + (int)j) % (int)pr.G; // global z of plane j (zbase may be negative)
	if (zg < 0) zg += (int)pr.G;
	const bool below_dead = zg == 0; // z-1 == -1 is dropped by the >= 0 test (compute_clustered.wgsl:104)
	const u32 jb = below_dead ? j : j - 1;
	const u32 ja = (pr.wrap_full && j + 1 == pr.nplanes) ? 0u : j + 1; // z == G passes `<= G` and wraps to 0
	n.plane[0] = in + jb * plane_words;
	n.plane[1] = in + j * plane_words;
	n.plane[2] = in + ja * plane_words;
	n.zmask[0] = below_dead ? 0u : 0xFFFFFFFFu;
	n.zmask[1] = 0xFFFFFFFFu;
	n.zmask[2] = 0xFFFFFFFFu;
	n.yrow[0] = y == 0 ? 0u : y - 1;
	n.yrow[1] = y;
	n.yrow[2] = (y + 1 == pr.G) ? 0u : y + 1;
	n.ymask[0] = y == 0 ? 0u : 0xFFFFFFFFu;
	n.ymask[1] = 0xFFFFFFFFu;
	n.ymask[2] = 0xFFFFFFFFu;
	return n;
}

// One row segment of 4 words plus the word on either side (x-1 of word 0 is dead at cx0 == 0; x+1 of the last
// word of the row wraps to word 0 of the same row).
struct Seg
{
	u32 w[4];
	u32 lo, hi;
};

template <bool LR>
__device__ __forceinline__ Seg load_seg(const Nbr &n, int dy, int dz, u32 C, u32 cx0)
{
	const u32 *row = n.plane[dz + 1] + (size_t)n.yrow[dy + 1] * C;
	const u32 m = n.zmask[dz + 1] & n.ymask[dy + 1];
	const uint4 v = *reinterpret_cast<const uint4 *>(row + cx0);
	Seg s;
	s.w[0] = v.x & m;
	s.w[1] = v.y & m;
	s.w[2] = v.z & m;
	s.w[3] = v.w & m;
	if (LR)
	{
		const u32 lo = row[cx0 == 0 ? 0u : cx0 - 1];
		const u32 hi = row[cx0 + 4 == C ? 0u : cx0 + 4];
		s.lo = cx0 == 0 ? 0u : (lo & m);
		s.hi = hi & m;
	}
	else
	{
		s.lo = 0;
		s.hi = 0;
	}
	return s;
}

__device__ __forceinline__ u32 seg_l(const Seg &s, int i) { return from_left(s.w[i], i ? s.w[i - 1] : s.lo); }
__device__ __forceinline__ u32 seg_r(const Seg &s, int i) { return from_right(i < 3 ? s.w[i + 1] : s.hi, s.w[i]); }

template <int MAIN>
struct MainPlanes
{
	static constexpr int value = (MAIN == MAIN_VN || MAIN == MAIN_VN2D) ? 3 : (MAIN == MAIN_MOORE ? 5 : 4);
};

// ---------------------------------------------------------------------------------------------- class kernel
template <int MAIN, bool E, bool C_>
__global__ __launch_bounds__(256) void ca_packed_class(const u32 *__restrict__ in, u32 *__restrict__ out,
                                                       PlaneRange pr, u32 CV, int cv_shift, u32 blocks_per_plane,
                                                       PackedRuleArgs rules)
{
	// XCD-aware block order: hardware deals blocks round-robin over the 8 XCDs, so give XCD k the k-th
	// contiguous eighth of the (plane, tile) space: z-neighbour planes then meet in one XCD's L2.
	const u32 b = blockIdx.x, nb = gridDim.x;
	const u32 v = (nb & 7u) == 0 ? (b & 7u) * (nb >> 3) + (b >> 3) : b;
	const u32 pj = v / blocks_per_plane;
	const u32 t = (v - pj * blocks_per_plane) * 256u + threadIdx.x;
	if (t >= pr.G * CV) return;
	u32 y, cxv;
	if (cv_shift >= 0) { y = t >> cv_shift; cxv = t & (CV - 1u); }
	else { y = t / CV; cxv = t - y * CV; }
	const u32 j = pr.lo + pj;
	const u32 C = CV * 4u, cx0 = cxv * 4u;

	const Nbr n = neighbours(in, pr, C, j, y);

	constexpr bool kMainVN = MAIN == MAIN_VN || MAIN == MAIN_MOORE;
	constexpr bool kNeedEdges = E || MAIN == MAIN_MOORE || MAIN == MAIN_EDGES;
	constexpr bool kNeedCorners = C_ || MAIN == MAIN_MOORE || MAIN == MAIN_CORNERS;
	constexpr bool kNeedCenterLR = kMainVN || MAIN == MAIN_VN2D || MAIN == MAIN_MOORE2D;
	constexpr bool kNeedYM = kMainVN || MAIN == MAIN_VN2D || MAIN == MAIN_MOORE2D; // M of (y+-1, z)
	constexpr bool kNeedYLR = kNeedEdges || MAIN == MAIN_MOORE2D;                 // L,R of (y+-1, z)
	constexpr bool kNeedZM = kMainVN;                                              // M of (y, z+-1)
	constexpr bool kNeedZLR = kNeedEdges;                                          // L,R of (y, z+-1)
	constexpr bool kNeedDM = kNeedEdges;                                           // M of the 4 diagonals
	constexpr bool kNeedDLR = kNeedCorners;                                        // L,R of the 4 diagonals

	const Seg c = load_seg<kNeedCenterLR>(n, 0, 0, C, cx0);
	Seg yp{}, ym{}, zp{}, zm{}, pp{}, pm{}, mp{}, mm{};
	if (kNeedYM || kNeedYLR) { yp = load_seg<kNeedYLR>(n, 1, 0, C, cx0); ym = load_seg<kNeedYLR>(n, -1, 0, C, cx0); }
	if (kNeedZM || kNeedZLR) { zp = load_seg<kNeedZLR>(n, 0, 1, C, cx0); zm = load_seg<kNeedZLR>(n, 0, -1, C, cx0); }
	if (kNeedDM || kNeedDLR)
	{
		pp = load_seg<kNeedDLR>(n, 1, 1, C, cx0);
		pm = load_seg<kNeedDLR>(n, 1, -1, C, cx0);
		mp = load_seg<kNeedDLR>(n, -1, 1, C, cx0);
		mm = load_seg<kNeedDLR>(n, -1, -1, C, cx0);
	}

	u32 o[4];
#pragma unroll
	for (int i = 0; i < 4; i++)
	{
		u32 ed[4] = {0, 0, 0, 0}, co[4] = {0, 0, 0, 0};
		if (kNeedEdges)
		{
			const u32 x[12] = {seg_l(yp, i), seg_r(yp, i), seg_l(ym, i), seg_r(ym, i),
			                   seg_l(zp, i), seg_r(zp, i), seg_l(zm, i), seg_r(zm, i),
			                   pp.w[i], pm.w[i], mp.w[i], mm.w[i]};
			sum12(x, ed);
		}
		if (kNeedCorners)
		{
			const u32 x[8] = {seg_l(pp, i), seg_r(pp, i), seg_l(pm, i), seg_r(pm, i),
			                  seg_l(mp, i), seg_r(mp, i), seg_l(mm, i), seg_r(mm, i)};
			sum8(x, co);
		}
		constexpr int NP = MainPlanes<MAIN>::value;
		u32 mn[NP];
		if (MAIN == MAIN_VN)
		{
			u32 p[3];
			sum6(seg_l(c, i), seg_r(c, i), yp.w[i], ym.w[i], zp.w[i], zm.w[i], p);
			mn[0] = p[0]; mn[1] = p[1]; mn[2] = p[2];
		}
		else if (MAIN == MAIN_VN2D)
		{
			u32 p[3];
			sum4(seg_l(c, i), seg_r(c, i), yp.w[i], ym.w[i], p);
			mn[0] = p[0]; mn[1] = p[1]; mn[2] = p[2];
		}
		else if (MAIN == MAIN_MOORE)
		{
			u32 vn[3], p[5];
			sum6(seg_l(c, i), seg_r(c, i), yp.w[i], ym.w[i], zp.w[i], zm.w[i], vn);
			sum_moore(vn, ed, co, p);
			for (int q = 0; q < NP; q++) mn[q] = p[q];
		}
		else if (MAIN == MAIN_MOORE2D)
		{
			const u32 x[8] = {seg_l(c, i), seg_r(c, i), yp.w[i], seg_l(yp, i), seg_r(yp, i),
			                  ym.w[i], seg_l(ym, i), seg_r(ym, i)};
			u32 p[4];
			sum8(x, p);
			for (int q = 0; q < NP; q++) mn[q] = p[q];
		}
		else if (MAIN == MAIN_EDGES)
		{
			for (int q = 0; q < NP; q++) mn[q] = ed[q];
		}
		else
		{
			for (int q = 0; q < NP; q++) mn[q] = co[q];
		}
		u32 S = eval_prog<NP>(rules.set[0].survive, mn);
		u32 B = eval_prog<NP>(rules.set[0].born, mn);
		S |= eval_prog<(E ? 4 : 0)>(rules.set[1].survive, ed);
		B |= eval_prog<(E ? 4 : 0)>(rules.set[1].born, ed);
		S |= eval_prog<(C_ ? 4 : 0)>(rules.set[2].survive, co);
		B |= eval_prog<(C_ ? 4 : 0)>(rules.set[2].born, co);
		o[i] = next_state(c.w[i], S, B);
	}
	uint4 r;
	r.x = o[0]; r.y = o[1]; r.z = o[2]; r.w = o[3];
	*reinterpret_cast<uint4 *>(out + ((size_t)j * pr.G + y) * C + cx0) = r;
}

// -------------------------------------------------------------------------------------------- generic kernel
// One thread per output word; every listed offset is fetched, shifted and ripple-added into a 5-plane counter.
// Handles any row length and any offset list within the 3x3x3 shell (duplicates, (0,0,0)).
__global__ __launch_bounds__(256) void ca_packed_generic(const u32 *__restrict__ in, u32 *__restrict__ out,
                                                         PlaneRange pr, u32 C, OffsetLists lists, PackedRuleArgs rules)
{
	const u32 words_per_plane = C * pr.G;
	const size_t gid = (size_t)blockIdx.x * 256u + threadIdx.x;
	const size_t total = (size_t)(pr.hi - pr.lo) * words_per_plane;
	if (gid >= total) return;
	const u32 pj = (u32)(gid / words_per_plane);
	const u32 rem = (u32)(gid - (size_t)pj * words_per_plane);
	const u32 y = rem / C, cx = rem - y * C;
	const u32 j = pr.lo + pj;
	const Nbr n = neighbours(in, pr, C, j, y);

	u32 S = 0, B = 0;
	for (int s = 0; s < 3; s++)
	{
		u32 p[5] = {0, 0, 0, 0, 0};
		for (u32 i = 0; i < lists.n[s]; i++)
		{
			const u32 code = lists.code[s][i];
			const int dx = (int)(code & 3u) - 1, dy = (int)((code >> 2) & 3u) - 1, dz = (int)((code >> 4) & 3u) - 1;
			const u32 *row = n.plane[dz + 1] + (size_t)n.yrow[dy + 1] * C;
			const u32 m = n.zmask[dz + 1] & n.ymask[dy + 1];
			const u32 w = row[cx];
			u32 v;
			if (dx == 0) v = w;
			else if (dx < 0) v = from_left(w, cx == 0 ? 0u : row[cx - 1]);
			else v = from_right(row[cx + 1 == C ? 0u : cx + 1], w);
			u32 carry = v & m;
#pragma unroll
			for (int q = 0; q < 5; q++) { const u32 t = p[q] & carry; p[q] ^= carry; carry = t; }
		}
		S |= eval_prog<5>(rules.set[s].survive, p);
		B |= eval_prog<5>(rules.set[s].born, p);
	}
	const u32 self = n.plane[1][(size_t)y * C + cx];
	out[((size_t)j * pr.G + y) * C + cx] = next_state(self, S, B);
}

template <int MAIN, bool E, bool C_>
hipError_t launch_class(const PackedLaunch &l, hipStream_t stream)
{
	const u32 C = l.pr.G / 32u, CV = C / 4u;
	const u32 items = l.pr.G * CV;
	const u32 bpp = (items + 255u) / 256u;
	const u32 planes = l.pr.hi - l.pr.lo;
	int shift = -1;
	if ((CV & (CV - 1u)) == 0) { shift = 0; while ((1u << shift) < CV) shift++; }
	hipLaunchKernelGGL((ca_packed_class<MAIN, E, C_>), dim3(bpp * planes), dim3(256), 0, stream, l.in, l.out, l.pr, CV,
	                   shift, bpp, l.rules->prog);
	return hipGetLastError();
}

template <int MAIN>
hipError_t launch_class_ec(const PackedLaunch &l, hipStream_t stream)
{
	const bool e = l.rules->need[1], c = l.rules->need[2];
	if (e && c) return launch_class<MAIN, true, true>(l, stream);
	if (e) return launch_class<MAIN, true, false>(l, stream);
	if (c) return launch_class<MAIN, false, true>(l, stream);
	return launch_class<MAIN, false, false>(l, stream);
}

bool use_class_kernel(const CanonRules &r, uint32_t G, int variant)
{
	return variant != 1 && r.fast && ((G / 32u) % 4u) == 0;
}

} // namespace

const char *packed_kernel_name(const CanonRules &r, uint32_t G, int variant)
{
	if (!use_class_kernel(r, G, variant)) return "ca_packed_generic";
	static const char *names[6][4] = {
	    {"ca_packed_class<vn>", "ca_packed_class<vn,E>", "ca_packed_class<vn,C>", "ca_packed_class<vn,E,C>"},
	    {"ca_packed_class<vn2d>", "ca_packed_class<vn2d,E>", "ca_packed_class<vn2d,C>", "ca_packed_class<vn2d,E,C>"},
	    {"ca_packed_class<moore>", "ca_packed_class<moore,E>", "ca_packed_class<moore,C>", "ca_packed_class<moore,E,C>"},
	    {"ca_packed_class<moore2d>", "ca_packed_class<moore2d,E>", "ca_packed_class<moore2d,C>", "ca_packed_class<moore2d,E,C>"},
	    {"ca_packed_class<edges>", "ca_packed_class<edges,E>", "ca_packed_class<edges,C>", "ca_packed_class<edges,E,C>"},
	    {"ca_packed_class<corners>", "ca_packed_class<corners,E>", "ca_packed_class<corners,C>", "ca_packed_class<corners,E,C>"}};
	return names[r.main][(r.need[1] ? 1 : 0) + (r.need[2] ? 2 : 0)];
}

hipError_t launch_packed_step(const PackedLaunch &l, hipStream_t stream, const char **kernel_name)
{
	const CanonRules &r = *l.rules;
	if (kernel_name) *kernel_name = packed_kernel_name(r, l.pr.G, l.variant);
	if (l.pr.hi <= l.pr.lo) return hipSuccess;
	if (use_class_kernel(r, l.pr.G, l.variant))
	{
		switch (r.main)
		{
		case MAIN_VN: return launch_class_ec<MAIN_VN>(l, stream);
		case MAIN_VN2D: return launch_class_ec<MAIN_VN2D>(l, stream);
		case MAIN_MOORE: return launch_class_ec<MAIN_MOORE>(l, stream);
		case MAIN_MOORE2D: return launch_class_ec<MAIN_MOORE2D>(l, stream);
		case MAIN_EDGES: return launch_class_ec<MAIN_EDGES>(l, stream);
		case MAIN_CORNERS: return launch_class_ec<MAIN_CORNERS>(l, stream);
		default: break;
		}
	}
	const u32 C = l.pr.G / 32u;
	const size_t total = (size_t)(l.pr.hi - l.pr.lo) * C * l.pr.G;
	const u32 blocks = (u32)((total + 255u) / 256u);
	hipLaunchKernelGGL(ca_packed_generic, dim3(blocks), dim3(256), 0, stream, l.in, l.out, l.pr, C, r.lists, r.prog);
	return hipGetLastError();
}

} // namespace ca3d
