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
#include <algorithm>
#include <cstdlib>

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

// Same for the W words a thread owns: the cube is fetched and decoded once (scalar work) per W words. Literal i is
// (plane_i & a_i) ^ b_i with wave-uniform masks — pure mask arithmetic, no per-lane selects (v_cndmask is an
// order of magnitude slower than v_bitop3 on gfx950: tools/ubench/valu_rate.hip).
template <int W, int NP, int NPA>
__device__ __forceinline__ void eval_prog4(const RuleProg &pr, const u32 (&p)[W][NPA], u32 (&acc)[W])
{
	u32 r[W];
#pragma unroll
	for (int w = 0; w < W; w++) r[w] = 0;
	for (u32 c = 0; c < pr.n; c++)
	{
		const u32 cube = pr.cubes[c];
		u32 e[W];
#pragma unroll
		for (int w = 0; w < W; w++) e[w] = 0xFFFFFFFFu;
#pragma unroll
		for (int i = 0; i < NP; i++)
		{
			const u32 care = 0u - ((cube >> i) & 1u);                 // ~0 when the plane matters
			const u32 flip = (0u - ((cube >> (8 + i)) & 1u)) & care;  // ~0 when it must be set
			const u32 bmask = ~flip;                                  // set: 0, clear: ~0, don't care: ~0
#pragma unroll
			for (int w = 0; w < W; w++) e[w] &= bitop3<((TA & TB) ^ TC)>(p[w][i], care, bmask);
		}
#pragma unroll
		for (int w = 0; w < W; w++) r[w] |= e[w];
	}
#pragma unroll
	for (int w = 0; w < W; w++) acc[w] |= r[w] ^ pr.invert;
}

__device__ __forceinline__ u32 next_state(u32 alive, u32 S, u32 B)
{
	return bitop3<((TA & TB) | (~TA & TC)) & 0xFF>(alive, S, B); // alive ? survive : born
}

// ---- register-resident rule programs ----------------------------------------------------------------------
// The interpreter above decodes every cube with scalar instructions each time it runs; inside the kernels that
// becomes the bottleneck (the CU's single scalar unit). Programs of at most kFastCubes cubes — every rule the
// reference UI produces in practice — are instead expanded ONCE per wave into mask pairs: literal i of a cube is
// (plane_i & a_i) ^ b_i with (a, b) = (~0, 0) for "plane set", (~0, ~0) for "plane clear", (0, ~0) for "don't
// care", so a cube is NP bitop3 + one AND-reduction and nothing is decoded in the loops.
constexpr int kFastCubes = 2;

template <int NP>
struct FastProg
{
	u32 n, invert;
	u32 a[kFastCubes][NP > 0 ? NP : 1], b[kFastCubes][NP > 0 ? NP : 1];
};

template <int NP>
__device__ __forceinline__ FastProg<NP> expand_prog(const RuleProg &pr)
{
	FastProg<NP> f;
	f.n = pr.n;
	f.invert = pr.invert;
#pragma unroll
	for (int c = 0; c < kFastCubes; c++)
	{
		const u32 cube = pr.cubes[c];
#pragma unroll
		for (int i = 0; i < NP; i++)
		{
			const bool care = (cube >> i) & 1u, val = (cube >> (8 + i)) & 1u;
			f.a[c][i] = care ? 0xFFFFFFFFu : 0u;
			f.b[c][i] = (care && val) ? 0u : 0xFFFFFFFFu;
		}
	}
	return f;
}

template <int NP>
__device__ __forceinline__ u32 cube_fast(const FastProg<NP> &f, int c, const u32 *p)
{
	u32 lit[NP > 0 ? NP : 1];
#pragma unroll
	for (int i = 0; i < NP; i++) lit[i] = bitop3<((TA & TB) ^ TC)>(p[i], f.a[c][i], f.b[c][i]);
	if (NP == 1) return lit[0];
	if (NP == 2) return lit[0] & lit[1];
	u32 e = bitop3<(TA & TB & TC)>(lit[0], lit[1], lit[2]);
	if (NP == 4) e &= lit[3];
	if (NP == 5) e = bitop3<(TA & TB & TC)>(e, lit[3 < NP ? 3 : 0], lit[4 < NP ? 4 : 0]);
	return e;
}

template <int W, int NP, int NPA>
__device__ __forceinline__ void eval_fast4(const FastProg<NP> &f, const u32 (&p)[W][NPA], u32 (&acc)[W], u32 &constant)
{
	if (NP == 0 || f.n == 0)
	{
		constant |= f.invert; // a program without cubes is the constant `invert`: fold it into one scalar
		return;
	}
	u32 r[W];
#pragma unroll
	for (int w = 0; w < W; w++) r[w] = cube_fast<NP>(f, 0, p[w]);
	if (f.n > 1)
	{
#pragma unroll
		for (int w = 0; w < W; w++) r[w] |= cube_fast<NP>(f, 1, p[w]);
	}
#pragma unroll
	for (int w = 0; w < W; w++) acc[w] |= r[w] ^ f.invert;
}

template <int MAIN>
struct MainPlanes
{
	static constexpr int value = (MAIN == MAIN_VN || MAIN == MAIN_VN2D) ? 3 : (MAIN == MAIN_MOORE ? 5 : 4);
};

// The six programs of a kernel instantiation, expanded into registers.
template <int MAIN, bool E, bool C_>
struct FastRules
{
	FastProg<MainPlanes<MAIN>::value> mb, ms;
	FastProg<(E ? 4 : 0)> eb, es;
	FastProg<(C_ ? 4 : 0)> cb, cs;
};

template <int MAIN, bool E, bool C_>
__device__ __forceinline__ FastRules<MAIN, E, C_> expand_rules(const PackedRuleArgs &r)
{
	FastRules<MAIN, E, C_> f;
	f.mb = expand_prog<MainPlanes<MAIN>::value>(r.set[0].born);
	f.ms = expand_prog<MainPlanes<MAIN>::value>(r.set[0].survive);
	f.eb = expand_prog<(E ? 4 : 0)>(r.set[1].born);
	f.es = expand_prog<(E ? 4 : 0)>(r.set[1].survive);
	f.cb = expand_prog<(C_ ? 4 : 0)>(r.set[2].born);
	f.cs = expand_prog<(C_ ? 4 : 0)>(r.set[2].survive);
	return f;
}

template <int W, int MAIN, bool E, bool C_, int NP>
__device__ __forceinline__ void apply_rules(const PackedRuleArgs &rules, const u32 (&mn)[W][NP], const u32 (&ed)[W][4],
                                            const u32 (&co)[W][4], u32 (&S)[W], u32 (&B)[W])
{
	eval_prog4<W, NP>(rules.set[0].survive, mn, S);
	eval_prog4<W, NP>(rules.set[0].born, mn, B);
	eval_prog4<W, (E ? 4 : 0)>(rules.set[1].survive, ed, S);
	eval_prog4<W, (E ? 4 : 0)>(rules.set[1].born, ed, B);
	eval_prog4<W, (C_ ? 4 : 0)>(rules.set[2].survive, co, S);
	eval_prog4<W, (C_ ? 4 : 0)>(rules.set[2].born, co, B);
}

template <int W, int MAIN, bool E, bool C_, int NP>
__device__ __forceinline__ void apply_rules(const FastRules<MAIN, E, C_> &f, const u32 (&mn)[W][NP], const u32 (&ed)[W][4],
                                            const u32 (&co)[W][4], u32 (&S)[W], u32 (&B)[W])
{
	u32 sc = 0, bc = 0;
	eval_fast4<W, NP>(f.ms, mn, S, sc);
	eval_fast4<W, NP>(f.mb, mn, B, bc);
	eval_fast4<W, (E ? 4 : 0)>(f.es, ed, S, sc);
	eval_fast4<W, (E ? 4 : 0)>(f.eb, ed, B, bc);
	eval_fast4<W, (C_ ? 4 : 0)>(f.cs, co, S, sc);
	eval_fast4<W, (C_ ? 4 : 0)>(f.cb, co, B, bc);
#pragma unroll
	for (int w = 0; w < W; w++) { S[w] |= sc; B[w] |= bc; }
}

// Host side: can every program of these rules be expanded into registers (<= kFastCubes cubes)?
bool rules_fit_fast(const CanonRules &r)
{
	for (int s2 = 0; s2 < 3; s2++)
		if (r.prog.set[s2].born.n > (u32)kFastCubes || r.prog.set[s2].survive.n > (u32)kFastCubes) return false;
	return true;
}

// ---- plane / row addressing for the generic kernel -------------------------------------------------------
struct Nbr
{
	const u32 *plane[3]; // z-1, z, z+1 (dead planes point at a valid plane and carry mask 0)
	u32 zmask[3];
	u32 yrow[3]; // y-1, y, y+1 (clamped / wrapped)
	u32 ymask[3];
};

__device__ __forceinline__ int global_z(const PlaneRange &pr, u32 j)
{
	int zg = pr.zbase + (int)j; // zbase in (-G, G), j <= G + 2*ghost: one conditional fold each way is enough
	if (zg < 0) zg += (int)pr.G;
	if (zg >= (int)pr.G) zg -= (int)pr.G;
	return zg;
}

__device__ __forceinline__ Nbr neighbours(const u32 *in, const PlaneRange &pr, u32 C, u32 j, u32 y)
{
	Nbr n;
	const size_t plane_words = (size_t)C * pr.G;
	const bool below_dead = global_z(pr, j) == 0; // z-1 == -1 is dropped by the >= 0 test (compute_clustered.wgsl:104)
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

// ---------------------------------------------------------------------------------------------- class kernel
// One row segment of 4 words plus the word on either side (x-1 of word 0 is dead at cx0 == 0; x+1 of the last
// word of the row wraps to word 0 of the same row).
template <int W>
struct SegT
{
	u32 w[W];
	u32 lo, hi;
};
using Seg = SegT<4>;

template <int W>
__device__ __forceinline__ u32 seg_l(const SegT<W> &s, int i) { return from_left(s.w[i], i ? s.w[i > 0 ? i - 1 : 0] : s.lo); }
template <int W>
__device__ __forceinline__ u32 seg_r(const SegT<W> &s, int i) { return from_right(i < W - 1 ? s.w[i < W - 1 ? i + 1 : 0] : s.hi, s.w[i]); }

struct TileGeom
{
	u32 CV;              // uint4 per row (C / 4)
	int cv_shift;        // log2(CV) or -1
	u32 tiles_per_plane; // 256-thread tiles per plane
	int tpp_shift;       // log2(tiles_per_plane) or -1
	u32 use_shfl;        // rows sit inside one wave: edge words come from neighbour lanes
};

// Per-thread constants of the (y, cx) position.
struct Pos
{
	u32 off[3];   // word offset of the segment inside a plane for rows y-1, y, y+1
	u32 ymask[3];
	u32 lo_rel, hi_rel; // word offset of the edge words relative to the row start (load path)
	u32 row0[3];        // row starts
	u32 lo_mask;        // 0 at cx0 == 0
	int src_lo, src_hi; // source lanes (shuffle path)
	bool shfl;
};

template <bool LR>
__device__ __forceinline__ Seg load_seg(const u32 *plane, const Pos &ps, int r, u32 zmask)
{
	const u32 m = zmask & ps.ymask[r];
	const uint4 v = *reinterpret_cast<const uint4 *>(plane + ps.off[r]);
	Seg s;
	s.w[0] = v.x & m;
	s.w[1] = v.y & m;
	s.w[2] = v.z & m;
	s.w[3] = v.w & m;
	s.lo = 0;
	s.hi = 0;
	if (LR)
	{
		if (ps.shfl)
		{
			s.lo = (u32)__shfl((int)s.w[3], ps.src_lo) & ps.lo_mask;
			s.hi = (u32)__shfl((int)s.w[0], ps.src_hi);
		}
		else
		{
			s.lo = plane[ps.row0[r] + ps.lo_rel] & m & ps.lo_mask;
			s.hi = plane[ps.row0[r] + ps.hi_rel] & m;
		}
	}
	return s;
}

template <int W>
struct PlaneRowsT
{
	SegT<W> ym, c, yp;
};
using PlaneRows = PlaneRowsT<4>;

// New state of the W words at the centre of P1 given the planes below (P0, already masked by the caller through
// `zmask` when z-1 is dead) and above (P2). Shared by the streaming and the fused kernels.
template <int W, int MAIN, bool E, bool C_, typename RS>
__device__ __forceinline__ void evolve(const PlaneRowsT<W> &P0, const PlaneRowsT<W> &P1, const PlaneRowsT<W> &P2, u32 zmask,
                                       const RS &rules, u32 (&out)[W])
{
	constexpr bool kNeedEdges = E || MAIN == MAIN_MOORE || MAIN == MAIN_EDGES;
	constexpr bool kNeedCorners = C_ || MAIN == MAIN_MOORE || MAIN == MAIN_CORNERS;

	constexpr int NP = MainPlanes<MAIN>::value;
	u32 mn[W][NP], ed[W][4], co[W][4];
#pragma unroll
	for (int i = 0; i < W; i++)
	{
		for (int k = 0; k < 4; k++) { ed[i][k] = 0; co[i][k] = 0; }
		if (kNeedEdges)
		{
			const u32 x[12] = {seg_l(P1.yp, i), seg_r(P1.yp, i), seg_l(P1.ym, i), seg_r(P1.ym, i),
			                   seg_l(P2.c, i), seg_r(P2.c, i), seg_l(P0.c, i) & zmask, seg_r(P0.c, i) & zmask,
			                   P2.yp.w[i], P0.yp.w[i] & zmask, P2.ym.w[i], P0.ym.w[i] & zmask};
			sum12(x, ed[i]);
		}
		if (kNeedCorners)
		{
			const u32 x[8] = {seg_l(P2.yp, i), seg_r(P2.yp, i), seg_l(P0.yp, i) & zmask, seg_r(P0.yp, i) & zmask,
			                  seg_l(P2.ym, i), seg_r(P2.ym, i), seg_l(P0.ym, i) & zmask, seg_r(P0.ym, i) & zmask};
			sum8(x, co[i]);
		}
		if (MAIN == MAIN_VN)
		{
			u32 p[3];
			sum6(seg_l(P1.c, i), seg_r(P1.c, i), P1.yp.w[i], P1.ym.w[i], P2.c.w[i], P0.c.w[i] & zmask, p);
			for (int k = 0; k < 3; k++) mn[i][k] = p[k];
		}
		else if (MAIN == MAIN_VN2D)
		{
			u32 p[3];
			sum4(seg_l(P1.c, i), seg_r(P1.c, i), P1.yp.w[i], P1.ym.w[i], p);
			for (int k = 0; k < 3; k++) mn[i][k] = p[k];
		}
		else if (MAIN == MAIN_MOORE)
		{
			u32 vn[3], p[5];
			sum6(seg_l(P1.c, i), seg_r(P1.c, i), P1.yp.w[i], P1.ym.w[i], P2.c.w[i], P0.c.w[i] & zmask, vn);
			sum_moore(vn, ed[i], co[i], p);
			for (int k = 0; k < NP; k++) mn[i][k] = p[k];
		}
		else if (MAIN == MAIN_MOORE2D)
		{
			const u32 x[8] = {seg_l(P1.c, i), seg_r(P1.c, i), P1.yp.w[i], seg_l(P1.yp, i), seg_r(P1.yp, i),
			                  P1.ym.w[i], seg_l(P1.ym, i), seg_r(P1.ym, i)};
			u32 p[4];
			sum8(x, p);
			for (int k = 0; k < NP; k++) mn[i][k] = p[k];
		}
		else if (MAIN == MAIN_EDGES)
		{
			for (int k = 0; k < NP; k++) mn[i][k] = ed[i][k];
		}
		else
		{
			for (int k = 0; k < NP; k++) mn[i][k] = co[i][k];
		}
	}
	u32 S[W], B[W];
#pragma unroll
	for (int i = 0; i < W; i++) { S[i] = 0; B[i] = 0; }
	apply_rules<W, MAIN, E, C_, NP>(rules, mn, ed, co, S, B);
#pragma unroll
	for (int i = 0; i < W; i++) out[i] = next_state(P1.c.w[i], S[i], B[i]);
}

template <int MAIN, bool E, bool C_, typename RS>
__device__ __forceinline__ uint4 evolve4(const PlaneRows &P0, const PlaneRows &P1, const PlaneRows &P2, u32 zmask, const RS &rules)
{
	u32 o[4];
	evolve<4, MAIN, E, C_>(P0, P1, P2, zmask, rules, o);
	uint4 r;
	r.x = o[0]; r.y = o[1]; r.z = o[2]; r.w = o[3];
	return r;
}

// Each thread owns one dwordx4 column position (y, cx0..cx0+3) and walks ZR consecutive z-planes with the three
// planes it needs held in registers, so a plane's rows are fetched once per ZR outputs instead of three times.
template <int MAIN, bool E, bool C_, int ZR, bool FAST>
__global__ __launch_bounds__(256) void ca_packed_class(const u32 *__restrict__ in, u32 *__restrict__ out,
                                                       PlaneRange pr, TileGeom g, PackedRuleArgs rules_in)
{
	constexpr bool kMainVN = MAIN == MAIN_VN || MAIN == MAIN_MOORE;
	constexpr bool kNeedEdges = E || MAIN == MAIN_MOORE || MAIN == MAIN_EDGES;
	constexpr bool kNeedCorners = C_ || MAIN == MAIN_MOORE || MAIN == MAIN_CORNERS;
	constexpr bool kCenterLR = kMainVN || MAIN == MAIN_VN2D || MAIN == MAIN_MOORE2D || kNeedEdges; // own plane or z+-1 planes
	constexpr bool kYRows = kMainVN || MAIN == MAIN_VN2D || MAIN == MAIN_MOORE2D || kNeedEdges || kNeedCorners;
	constexpr bool kYLR = kNeedEdges || kNeedCorners || MAIN == MAIN_MOORE2D;
	constexpr bool kZNbr = kMainVN || kNeedEdges || kNeedCorners;  // z+-1 planes needed at all
	constexpr bool kZYRows = kNeedEdges || kNeedCorners;           // y+-1 rows of the z+-1 planes (diagonals)

	// XCD-aware block order: hardware deals blocks round-robin over the 8 XCDs; give XCD k the k-th contiguous
	// eighth of the (z-run, tile) space so z-neighbour planes meet in one XCD's L2.
	const u32 b = blockIdx.x, nb = gridDim.x;
	const u32 v = (nb & 7u) == 0 ? (b & 7u) * (nb >> 3) + (b >> 3) : b;
	u32 zr, tile;
	if (g.tpp_shift >= 0) { zr = v >> g.tpp_shift; tile = v & (g.tiles_per_plane - 1u); }
	else { zr = v / g.tiles_per_plane; tile = v - zr * g.tiles_per_plane; }
	const u32 t = tile * 256u + threadIdx.x;
	if (t >= pr.G * g.CV) return;
	u32 y, cxv;
	if (g.cv_shift >= 0) { y = t >> g.cv_shift; cxv = t & (g.CV - 1u); }
	else { y = t / g.CV; cxv = t - y * g.CV; }
	const u32 C = g.CV * 4u, cx0 = cxv * 4u;
	const u32 plane_words = C * pr.G;

	Pos ps;
	{
		const u32 yr[3] = {y == 0 ? 0u : y - 1, y, (y + 1 == pr.G) ? 0u : y + 1};
		for (int r = 0; r < 3; r++) { ps.row0[r] = yr[r] * C; ps.off[r] = ps.row0[r] + cx0; ps.ymask[r] = 0xFFFFFFFFu; }
		ps.ymask[0] = y == 0 ? 0u : 0xFFFFFFFFu;
		ps.lo_rel = cx0 == 0 ? 0u : cx0 - 1;
		ps.hi_rel = cx0 + 4 == C ? 0u : cx0 + 4;
		ps.lo_mask = cx0 == 0 ? 0u : 0xFFFFFFFFu;
		const int lane = (int)(threadIdx.x & 63u);
		ps.src_lo = lane - 1;
		ps.src_hi = cxv + 1 == g.CV ? lane - (int)(g.CV - 1u) : lane + 1;
		ps.shfl = (g.use_shfl & 1u) != 0;
	}

	FastRules<MAIN, E, C_> frules;
	if (FAST) frules = expand_rules<MAIN, E, C_>(rules_in);

	const u32 j0 = pr.lo + zr * ZR;
	// window plane q holds array plane j0 + q - 1 (q = 0 .. ZR+1); out-of-range ends are clamped / wrapped
	PlaneRows win[ZR + 2];
#pragma unroll
	for (int q = 0; q < ZR + 2; q++)
	{
		const bool is_out = q >= 1 && q <= ZR; // planes that are themselves outputs of this thread
		if (!kZNbr && !is_out) continue;
		u32 jq = j0 + (u32)q - 1u;
		if (q == 0 && j0 == 0) jq = 0;                                     // only ever used masked (global z == 0)
		if (jq >= pr.nplanes) jq = (jq == pr.nplanes && pr.wrap_full) ? 0u : pr.nplanes - 1u;
		const u32 *plane = in + (size_t)jq * plane_words;
		win[q].c = load_seg<kCenterLR>(plane, ps, 1, 0xFFFFFFFFu);
		if (kYRows && (is_out || kZYRows))
		{
			win[q].ym = load_seg<kYLR>(plane, ps, 0, 0xFFFFFFFFu);
			win[q].yp = load_seg<kYLR>(plane, ps, 2, 0xFFFFFFFFu);
		}
	}

	int zg = global_z(pr, j0);
#pragma unroll
	for (int q = 1; q <= ZR; q++)
	{
		const u32 j = j0 + (u32)q - 1u;
		if (j >= pr.hi) break;
		const u32 zmask = zg == 0 ? 0u : 0xFFFFFFFFu; // z-1 == -1 is dropped (compute_clustered.wgsl:104)
		zg = zg + 1 == (int)pr.G ? 0 : zg + 1;
		uint4 r;
		if (FAST) r = evolve4<MAIN, E, C_>(win[q - 1], win[q], win[q + 1], zmask, frules);
		else r = evolve4<MAIN, E, C_>(win[q - 1], win[q], win[q + 1], zmask, rules_in);
		typedef u32 u32x4 __attribute__((ext_vector_type(4)));
		if (g.use_shfl & 2u) { u32x4 rv = {r.x, r.y, r.z, r.w}; __builtin_nontemporal_store(rv, reinterpret_cast<u32x4 *>(out + (size_t)j * plane_words + ps.off[1])); }
		else *reinterpret_cast<uint4 *>(out + (size_t)j * plane_words + ps.off[1]) = r;
	}
}

// ---------------------------------------------------------------------------------------------- fused kernel
// Two CA steps per launch (temporal blocking) with NO shared memory and NO barriers: every wavefront is an
// independent worker. A wave owns a strip of ROWS = 64 / LPR consecutive rows x the full x extent (LPR lanes per
// row, W words per lane) and streams along z over a chunk of ZC planes. Per plane it keeps three-plane windows
// of generation 0 (loaded) and generation 1 (after one step) in registers; the z neighbours are the thread's
// own registers and the x / y neighbours come from other lanes of the same wave (ds_bpermute, no LDS memory).
// Two rows on each side of the strip and two planes on each side of the chunk are halo: they go stale one
// layer per step, which is exactly their depth. HBM sees the grid once per two steps, the launch boundary is
// paid once per two steps, and because waves never synchronise their load / compute / store phases drift apart
// and overlap instead of marching in lock-step.
//
// Boundary: rows / planes at global coordinate -1 are dead (forced to zero in every generation); coordinates
// >= G are replicas of coordinate - G and evolve with a dead neighbour below whenever their own coordinate is
// 0 (mod G) — the same rule the slab ghosts use.
struct FusedGeom
{
	u32 LPR, lpr_shift; // lanes per row (C / W), power of two, <= 8
	u32 nstrips;        // strips along y
	u32 nchunks;        // chunks along z
	u32 dbg;            // profiling aid: 1 = skip the compute (memory phases only), 2 = skip the stores too
};

template <int W> struct VecT;
template <> struct VecT<2> { typedef uint2 type; };
template <> struct VecT<4> { typedef uint4 type; };

template <int W>
__device__ __forceinline__ void vec_load(const u32 *p, u32 (&w)[W])
{
	if (W <= 4)
	{
		constexpr int V = W <= 2 ? 2 : 4;
		const typename VecT<V>::type v = *reinterpret_cast<const typename VecT<V>::type *>(p);
		const u32 *e = reinterpret_cast<const u32 *>(&v);
#pragma unroll
		for (int k = 0; k < W; k++) w[k] = e[k];
	}
	else
	{
#pragma unroll
		for (int c = 0; c < W / 4; c++)
		{
			const uint4 v = *reinterpret_cast<const uint4 *>(p + 4 * c);
			w[4 * c] = v.x; w[4 * c + 1] = v.y; w[4 * c + 2] = v.z; w[4 * c + 3] = v.w;
		}
	}
}

template <int W>
__device__ __forceinline__ void vec_store(u32 *p, const u32 (&w)[W])
{
	if (W <= 4)
	{
		constexpr int V = W <= 2 ? 2 : 4;
		typename VecT<V>::type v;
		u32 *e = reinterpret_cast<u32 *>(&v);
#pragma unroll
		for (int k = 0; k < W; k++) e[k] = w[k];
		*reinterpret_cast<typename VecT<V>::type *>(p) = v;
	}
	else
	{
#pragma unroll
		for (int c = 0; c < W / 4; c++)
		{
			uint4 v;
			v.x = w[4 * c]; v.y = w[4 * c + 1]; v.z = w[4 * c + 2]; v.w = w[4 * c + 3];
			*reinterpret_cast<uint4 *>(p + 4 * c) = v;
		}
	}
}

template <int MAIN, bool E, bool C_, int W, int ZC, bool FAST>
__global__ __launch_bounds__(64) void ca_packed_fused(const u32 *__restrict__ in, u32 *__restrict__ out,
                                                      PlaneRange pr, FusedGeom g, PackedRuleArgs rules_in)
{
	constexpr int T = 2;
	constexpr bool kNeedEdges = E || MAIN == MAIN_MOORE || MAIN == MAIN_EDGES;
	constexpr bool kNeedCorners = C_ || MAIN == MAIN_MOORE || MAIN == MAIN_CORNERS;
	constexpr bool kMainVN = MAIN == MAIN_VN || MAIN == MAIN_MOORE;
	constexpr bool kCenterLR = kMainVN || MAIN == MAIN_VN2D || MAIN == MAIN_MOORE2D || kNeedEdges;
	constexpr bool kYLR = kNeedEdges || kNeedCorners || MAIN == MAIN_MOORE2D;

	const u32 b = blockIdx.x, nb = gridDim.x;
	const u32 v = (nb & 7u) == 0 ? (b & 7u) * (nb >> 3) + (b >> 3) : b; // XCD k gets a contiguous z range
	const u32 chunk = v / g.nstrips, strip = v - chunk * g.nstrips;
	const int lane = (int)threadIdx.x;
	const u32 rr = (u32)lane >> g.lpr_shift, cxl = (u32)lane & (g.LPR - 1u);
	const u32 ROWS = 64u >> g.lpr_shift, UR = ROWS - 2u * T;
	const u32 C = g.LPR * W, cx0 = cxl * W;
	const int G = (int)pr.G;
	const u32 plane_words = C * pr.G;

	const int gy = (int)(strip * UR) - T + (int)rr;
	const int gyw = gy < 0 ? 0 : (gy >= G ? gy - G : gy);
	const u32 live_mask = gy >= 0 ? 0xFFFFFFFFu : 0u;
	const u32 ym_mask = gyw == 0 ? 0u : 0xFFFFFFFFu; // y-1 == -1 is dropped
	const int src_ym = rr == 0 ? lane : lane - (int)g.LPR, src_yp = rr + 1u == ROWS ? lane : lane + (int)g.LPR;
	const int src_lo = lane - 1, src_hi = cxl + 1u == g.LPR ? lane - (int)(g.LPR - 1u) : lane + 1;
	const u32 lo_mask = cxl == 0 ? 0u : 0xFFFFFFFFu;
	const size_t row_off = (size_t)gyw * C + cx0;
	const bool store_row = rr >= (u32)T && rr < ROWS - (u32)T && gy >= 0 && gy < G;

	FastRules<MAIN, E, C_> frules;
	if (FAST) frules = expand_rules<MAIN, E, C_>(rules_in);

	// neighbour rows of one plane of one generation, built from the thread's own W words by lane exchange
	auto rows_of = [&](const u32 (&val)[W]) {
		PlaneRowsT<W> P;
#pragma unroll
		for (int k = 0; k < W; k++) P.c.w[k] = val[k];
		P.c.lo = 0; P.c.hi = 0;
		if (kCenterLR || kYLR)
		{
			P.c.lo = (u32)__shfl((int)val[W - 1], src_lo) & lo_mask;
			P.c.hi = (u32)__shfl((int)val[0], src_hi);
		}
#pragma unroll
		for (int k = 0; k < W; k++)
		{
			P.ym.w[k] = (u32)__shfl((int)val[k], src_ym) & ym_mask;
			P.yp.w[k] = (u32)__shfl((int)val[k], src_yp);
		}
		P.ym.lo = P.ym.hi = P.yp.lo = P.yp.hi = 0;
		if (kYLR)
		{
			P.ym.lo = (u32)__shfl((int)P.c.lo, src_ym) & ym_mask;
			P.ym.hi = (u32)__shfl((int)P.c.hi, src_ym) & ym_mask;
			P.yp.lo = (u32)__shfl((int)P.c.lo, src_yp);
			P.yp.hi = (u32)__shfl((int)P.c.hi, src_yp);
		}
		return P;
	};

	const int zc0 = (int)pr.lo + (int)(chunk * ZC); // first output plane of the chunk
	constexpr int NI = ZC + 2 * T;                   // planes streamed: zc0 - 2 .. zc0 + ZC + 1
	// Issue every load of the chunk up front: the data returns in order, so the first planes can be worked on while
	// the rest of the chunk is still in flight (the waits the compiler inserts are counted vmcnt waits).
	u32 raw[NI][W];
#pragma unroll
	for (int i = 0; i < NI; i++)
	{
		const int p = zc0 - T + i;
		int jz;
		if (pr.wrap_full) jz = p < 0 ? 0 : (p >= G ? p - G : p);
		else jz = p < 0 ? 0 : (p >= (int)pr.nplanes ? (int)pr.nplanes - 1 : p);
		vec_load<W>(in + (size_t)jz * plane_words + row_off, raw[i]);
	}
	PlaneRowsT<W> R0[NI], R1[NI]; // generation 0 / 1 rows by stream index (compile-time indices)
#pragma unroll
	for (int i = 0; i < NI; i++)
	{
		// ---- plane p of generation 0
		const int p = zc0 - T + i;
		const bool zl = pr.wrap_full ? p >= 0 : (p >= 0 && p < (int)pr.nplanes);
		u32 val[W];
		const u32 m0 = zl ? live_mask : 0u;
#pragma unroll
		for (int k = 0; k < W; k++) val[k] = raw[i][k] & m0;
		R0[i] = rows_of(val);
		if (g.dbg) { R1[i] = R0[i]; }
		// ---- generation 1 of plane p-1 (stream index i-1)
		if (!g.dbg && i >= 2)
		{
			const int p1 = p - 1;
			bool zl1, zd1;
			if (pr.wrap_full) { zl1 = p1 >= 0; const int w1 = p1 < 0 ? 1 : (p1 >= G ? p1 - G : p1); zd1 = w1 == 0; }
			else { zl1 = p1 >= 0 && p1 < (int)pr.nplanes; zd1 = global_z(pr, (u32)(p1 < 0 ? 0 : p1)) == 0; }
			u32 o[W];
			if (FAST) evolve<W, MAIN, E, C_>(R0[i - 2], R0[i - 1], R0[i], zd1 ? 0u : 0xFFFFFFFFu, frules, o);
			else evolve<W, MAIN, E, C_>(R0[i - 2], R0[i - 1], R0[i], zd1 ? 0u : 0xFFFFFFFFu, rules_in, o);
			const u32 m1 = zl1 ? live_mask : 0u;
#pragma unroll
			for (int k = 0; k < W; k++) o[k] &= m1;
			R1[i - 1] = rows_of(o);
		}
		// ---- generation 2 of plane p-2 (stream index i-2): needs generation 1 of stream indices i-3, i-2, i-1
		if (i >= 4)
		{
			const int p2 = p - 2;
			u32 o[W];
			if (!g.dbg)
			{
				bool zd2;
				if (pr.wrap_full) { const int w2 = p2 >= G ? p2 - G : p2; zd2 = w2 == 0; }
				else { zd2 = global_z(pr, (u32)p2) == 0; }
				if (FAST) evolve<W, MAIN, E, C_>(R1[i - 3], R1[i - 2], R1[i - 1], zd2 ? 0u : 0xFFFFFFFFu, frules, o);
				else evolve<W, MAIN, E, C_>(R1[i - 3], R1[i - 2], R1[i - 1], zd2 ? 0u : 0xFFFFFFFFu, rules_in, o);
			}
			else
			{
#pragma unroll
				for (int k = 0; k < W; k++) o[k] = R1[i - 2].c.w[k];
			}
			if (g.dbg != 2 && store_row && p2 >= (int)pr.lo && p2 < (int)pr.hi && p2 < zc0 + ZC)
				vec_store<W>(out + (size_t)p2 * plane_words + (size_t)gy * C + cx0, o);
		}
	}
}

constexpr int kFuseZC = 12;

// Geometry of the fused kernel for a grid, or false when the grid does not suit it (then single steps are used).
bool fused_geometry(uint32_t G, FusedGeom *g, int *words_per_lane)
{
	const u32 C = G / 32u;
	int W;
	if (C == 8u) W = 2;
	else if (C == 16u) W = 4;
	else if (C == 32u) W = 8;
	else return false; // 256^3, 512^3, 1024^3: 4 lanes per row, 16 rows per wave
	const u32 LPR = C / (u32)W;
	u32 shift = 0;
	while ((1u << shift) < LPR) shift++;
	g->LPR = LPR;
	g->lpr_shift = shift;
	const u32 UR = 64u / LPR - 4u;
	g->nstrips = (G + UR - 1u) / UR;
	g->nchunks = 0;
	g->dbg = 0;
	*words_per_lane = W;
	return true;
}


template <int MAIN, bool E, bool C_, int W>
hipError_t launch_fused_w(const PackedLaunch &l, hipStream_t stream, FusedGeom g)
{
	const u32 planes = l.pr.hi - l.pr.lo;
	g.nchunks = (planes + kFuseZC - 1) / kFuseZC;
	g.dbg = (u32)(l.variant >> 8);
	const dim3 grid(g.nstrips * g.nchunks);
	if (rules_fit_fast(*l.rules))
		hipLaunchKernelGGL((ca_packed_fused<MAIN, E, C_, W, kFuseZC, true>), grid, dim3(64), 0, stream, l.in, l.out, l.pr, g, l.rules->prog);
	else
		hipLaunchKernelGGL((ca_packed_fused<MAIN, E, C_, W, kFuseZC, false>), grid, dim3(64), 0, stream, l.in, l.out, l.pr, g, l.rules->prog);
	return hipGetLastError();
}

template <int MAIN, bool E, bool C_>
hipError_t launch_fused(const PackedLaunch &l, hipStream_t stream)
{
	FusedGeom g;
	int W;
	if (!fused_geometry(l.pr.G, &g, &W)) return hipErrorInvalidValue;
	switch (W)
	{
	case 2: return launch_fused_w<MAIN, E, C_, 2>(l, stream, g);
	case 4: return launch_fused_w<MAIN, E, C_, 4>(l, stream, g);
	default: return launch_fused_w<MAIN, E, C_, 8>(l, stream, g);
	}
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
	const u32 C = l.pr.G / 32u;
	TileGeom g;
	g.CV = C / 4u;
	const u32 items = l.pr.G * g.CV;
	g.tiles_per_plane = (items + 255u) / 256u;
	auto log2_exact = [](u32 x) { int s = -1; if (x && (x & (x - 1u)) == 0) { s = 0; while ((1u << s) < x) s++; } return s; };
	g.cv_shift = log2_exact(g.CV);
	g.tpp_shift = log2_exact(g.tiles_per_plane);
	g.use_shfl = (g.cv_shift >= 0 && g.CV <= 64u) ? 1u : 0u;
	// Non-temporal stores (measured, MI355X): 6.9 vs 7.5 us per step at 512^3, but 58 vs 43 us at 1024^3 — they pay
	// only while both ping-pong buffers sit in the 256 MiB Infinity Cache with room to spare.
	static const int nt_env = [] { const char *e = getenv("CA3D_NT_STORE"); return e ? atoi(e) : -1; }();
	const bool nt = nt_env >= 0 ? nt_env != 0 : (size_t)l.pr.G * l.pr.G * C * sizeof(u32) <= (16u << 20);
	if (nt) g.use_shfl |= 2u; // (sc1 / sc0 sc1 stores were tried too: no gain at either size)
	const u32 planes = l.pr.hi - l.pr.lo;
	// z-run of 4 planes per thread once that still leaves >= 4 workgroups per CU; small grids keep 1 plane per
	// thread so all 256 CUs get work.
	// Planes per thread (measured on MI355X, 512^3 / 1024^3): kernels that read only face neighbours are
	// latency-bound and want many small waves (2 planes: 7.2 us vs 7.9 us with 4 and 9.0 us with 8 at 512^3);
	// kernels that need the diagonal rows re-read 9 rows per plane and amortise them over 4 planes (28 us vs 35 us).
	constexpr bool kDiagonals = E || C_ || MAIN == MAIN_MOORE || MAIN == MAIN_EDGES || MAIN == MAIN_CORNERS;
	constexpr int ZRUN = kDiagonals ? 4 : 2;
	const bool deep = (size_t)g.tiles_per_plane * ((planes + ZRUN - 1u) / ZRUN) >= 1024u;
	const bool fast = rules_fit_fast(*l.rules);
	const dim3 grid_deep(g.tiles_per_plane * ((planes + ZRUN - 1u) / ZRUN)), grid_flat(g.tiles_per_plane * planes);
	if (deep && fast) hipLaunchKernelGGL((ca_packed_class<MAIN, E, C_, ZRUN, true>), grid_deep, dim3(256), 0, stream, l.in, l.out, l.pr, g, l.rules->prog);
	else if (deep) hipLaunchKernelGGL((ca_packed_class<MAIN, E, C_, ZRUN, false>), grid_deep, dim3(256), 0, stream, l.in, l.out, l.pr, g, l.rules->prog);
	else if (fast) hipLaunchKernelGGL((ca_packed_class<MAIN, E, C_, 1, true>), grid_flat, dim3(256), 0, stream, l.in, l.out, l.pr, g, l.rules->prog);
	else hipLaunchKernelGGL((ca_packed_class<MAIN, E, C_, 1, false>), grid_flat, dim3(256), 0, stream, l.in, l.out, l.pr, g, l.rules->prog);
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
	return (variant & 0xFF) != 1 && r.fast && ((G / 32u) % 4u) == 0;
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

int packed_fused_steps(const CanonRules &r, uint32_t G, int variant)
{
	FusedGeom g;
	int W;
	if ((variant & 0xFF) != 0 || !use_class_kernel(r, G, variant & 0xFF)) return 0;
	if (r.main != MAIN_VN && r.main != MAIN_VN2D) return 0; // instantiated for the face neighbourhoods so far
	if (r.need[1] || r.need[2]) return 0;
	return fused_geometry(G, &g, &W) ? 2 : 0;
}

hipError_t launch_packed_fused(const PackedLaunch &l, hipStream_t stream, const char **kernel_name)
{
	if (kernel_name) *kernel_name = "ca_packed_fused<T=2>";
	if (l.pr.hi <= l.pr.lo) return hipSuccess;
	switch (l.rules->main)
	{
	case MAIN_VN: return launch_fused<MAIN_VN, false, false>(l, stream);
	case MAIN_VN2D: return launch_fused<MAIN_VN2D, false, false>(l, stream);
	default: return hipErrorInvalidValue;
	}
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
