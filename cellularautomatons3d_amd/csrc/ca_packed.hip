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

#include "ca_bitslice.inc"

#include "ca_packed_class_kernel.inc"

// launch arguments of the run-time compiled rolling-window kernels (device code: ca_packed_roll_kernel.inc)
struct RollArgs
{
	u32 lo, hi, nplanes, wrap_full;
	u32 lo2, hi2, runs1;
	int zbase;
};

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
		// ---- generation 1 of plane p-1 (stream index i-1)
		if (i >= 2)
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
			bool zd2;
			if (pr.wrap_full) { const int w2 = p2 >= G ? p2 - G : p2; zd2 = w2 == 0; }
			else { zd2 = global_z(pr, (u32)p2) == 0; }
			if (FAST) evolve<W, MAIN, E, C_>(R1[i - 3], R1[i - 2], R1[i - 1], zd2 ? 0u : 0xFFFFFFFFu, frules, o);
			else evolve<W, MAIN, E, C_>(R1[i - 3], R1[i - 2], R1[i - 1], zd2 ? 0u : 0xFFFFFFFFu, rules_in, o);
			if (store_row && p2 >= (int)pr.lo && p2 < (int)pr.hi && p2 < zc0 + ZC)
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
	*words_per_lane = W;
	return true;
}


template <int MAIN, bool E, bool C_, int W>
hipError_t launch_fused_w(const PackedLaunch &l, hipStream_t stream, FusedGeom g)
{
	const u32 planes = l.pr.hi - l.pr.lo;
	g.nchunks = (planes + kFuseZC - 1) / kFuseZC;
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
	const bool nt = (size_t)l.pr.nplanes * l.pr.G * C * sizeof(u32) <= (16u << 20); // per buffer, ghosts included
	if (nt) g.use_shfl |= 2u; // (sc1 / sc0 sc1 stores were tried too: no gain at either size)
	const bool two = l.pr.hi2 > l.pr.lo2;
	const u32 planes1 = l.pr.hi - l.pr.lo, planes2 = two ? l.pr.hi2 - l.pr.lo2 : 0u, planes = planes1 + planes2;
	const u32 shortest = two && planes2 < planes1 ? planes2 : planes1;
	// z-run of 4 planes per thread once that still leaves >= 4 workgroups per CU; small grids keep 1 plane per
	// thread so all 256 CUs get work.
	// Planes per thread (measured on MI355X, 512^3 / 1024^3): kernels that read only face neighbours are
	// latency-bound and want many small waves (2 planes: 7.2 us vs 7.9 us with 4 and 9.0 us with 8 at 512^3);
	// kernels that need the diagonal rows re-read 9 rows per plane and amortise them over 4 planes (28 us vs 35 us).
	constexpr bool kDiagonals = E || C_ || MAIN == MAIN_MOORE || MAIN == MAIN_EDGES || MAIN == MAIN_CORNERS;
	constexpr int ZRUN = kDiagonals ? 4 : 2;
	const bool deep = (size_t)g.tiles_per_plane * ((planes + ZRUN - 1u) / ZRUN) >= 1024u && shortest >= (u32)ZRUN;
	const bool fast = rules_fit_fast(*l.rules);
	const bool p2 = (g.use_shfl & 1u) != 0;
	const ClassJit *jit = l.class_jit;
	const bool jit_ok = jit && jit->main == MAIN && jit->e == E && jit->c == C_ && jit->deep && jit->deep_za && jit->flat;
	const bool np2_jit = !p2 && jit_ok && jit->deep_np2 && jit->flat_np2; // rows of whole uint4, not a power of two of them: the run-time compiled form
	const u32 zr_used = (deep && (p2 || np2_jit)) ? (u32)ZRUN : 1u;
	g.runs1 = (planes1 + zr_used - 1u) / zr_used;
	const u32 runs = g.runs1 + (planes2 + zr_used - 1u) / zr_used;
	const dim3 grid_deep(g.tiles_per_plane * runs), grid_flat(g.tiles_per_plane * runs);
#define CA3D_LAUNCH_CLASS(ZR_, FAST_, P2_, GRID_) \
	hipLaunchKernelGGL((ca_packed_class<MAIN, E, C_, ZR_, FAST_, P2_>), GRID_, dim3(256), 0, stream, l.in, l.out, l.pr, g, l.rules->prog)
	if (const RollJit *rj = l.roll_jit; p2 && rj && rj->cvl == g.cv_shift && rj->main == MAIN && rj->e == E && rj->c == C_)
	{
		// The rolling-window kernel for exactly these rules and this grid (ca_packed_roll_kernel.inc). Z planes per
		// thread: the largest of 8 / 4 / 2 that the shortest range holds and that still gives every SIMD its two waves
		// (a wave alone on a SIMD issues at half rate); launches too small for that keep the one-plane kernels below.
		static const u32 lds_pad = getenv("CA3D_ROLL_LDS") ? (u32)atoi(getenv("CA3D_ROLL_LDS")) : 0u;
		// two words per thread (roll_tile 3): twice the workgroups per plane
		if (l.roll_tile == 3)
			for (int wi = 1; wi >= 0; wi--)
			{
				const u32 Z = wi ? 16u : 8u;
				void *fn = rj->w2[wi];
				if ((l.roll_z && l.roll_z != (int)Z) || shortest < Z || !fn) continue;
				const u32 runs1 = (planes1 + Z - 1u) / Z, nruns = runs1 + (planes2 + Z - 1u) / Z;
				RollArgs a;
				a.lo = l.pr.lo; a.hi = l.pr.hi; a.nplanes = l.pr.nplanes; a.wrap_full = l.pr.wrap_full; a.zbase = l.pr.zbase;
				a.lo2 = l.pr.lo2; a.hi2 = two ? l.pr.hi2 : l.pr.lo2; a.runs1 = runs1;
				const u32 *in = l.in;
				u32 *out = l.out;
				void *args[] = {(void *)&in, (void *)&out, (void *)&a};
				return hipModuleLaunchKernel((hipFunction_t)fn, g.tiles_per_plane * 2u * nruns, 1, 1, 256, 1, 1, 0, stream, args, nullptr);
			}
		// wave tiles (roll_tile 2): one wave per workgroup, Z = 16 or 8
		if (l.roll_tile == 2)
			for (int wi = 1; wi >= 0; wi--)
			{
				const u32 Z = wi ? 16u : 8u;
				void *fn = rj->wtile[wi];
				if ((l.roll_z && l.roll_z != (int)Z) || shortest < Z || !fn) continue;
				const u32 runs1 = (planes1 + Z - 1u) / Z, nruns = runs1 + (planes2 + Z - 1u) / Z;
				RollArgs a;
				a.lo = l.pr.lo; a.hi = l.pr.hi; a.nplanes = l.pr.nplanes; a.wrap_full = l.pr.wrap_full; a.zbase = l.pr.zbase;
				a.lo2 = l.pr.lo2; a.hi2 = two ? l.pr.hi2 : l.pr.lo2; a.runs1 = runs1;
				const u32 *in = l.in;
				u32 *out = l.out;
				void *args[] = {(void *)&in, (void *)&out, (void *)&a};
				return hipModuleLaunchKernel((hipFunction_t)fn, g.tiles_per_plane * 4u * nruns, 1, 1, 64, 1, 1, 0, stream, args, nullptr);
			}
		// looped forms (15 / 30 planes per thread): forced by roll_z 15 / 30
		for (int li = 1; li >= 0; li--)
		{
			const u32 Z = li ? 30u : 15u;
			void *fn = rj->loop[l.roll_tile == 1 ? 1 : 0][li];
			if (l.roll_z != (int)Z || shortest < Z || !fn) continue;
			const u32 runs1 = (planes1 + Z - 1u) / Z, nruns = runs1 + (planes2 + Z - 1u) / Z;
			RollArgs a;
			a.lo = l.pr.lo; a.hi = l.pr.hi; a.nplanes = l.pr.nplanes; a.wrap_full = l.pr.wrap_full; a.zbase = l.pr.zbase;
			a.lo2 = l.pr.lo2; a.hi2 = two ? l.pr.hi2 : l.pr.lo2; a.runs1 = runs1;
			const u32 *in = l.in;
			u32 *out = l.out;
			void *args[] = {(void *)&in, (void *)&out, (void *)&a};
			return hipModuleLaunchKernel((hipFunction_t)fn, g.tiles_per_plane * nruns, 1, 1, 256, 1, 1, 0, stream, args, nullptr);
		}
		if (rj->tile_x && l.roll_tile == 1 && l.roll_z == 0 && shortest >= (u32)rj->zx)
		{
			const u32 Z = (u32)rj->zx;
			const u32 runs1 = (planes1 + Z - 1u) / Z, nruns = runs1 + (planes2 + Z - 1u) / Z;
			RollArgs a;
			a.lo = l.pr.lo; a.hi = l.pr.hi; a.nplanes = l.pr.nplanes; a.wrap_full = l.pr.wrap_full; a.zbase = l.pr.zbase;
			a.lo2 = l.pr.lo2; a.hi2 = two ? l.pr.hi2 : l.pr.lo2; a.runs1 = runs1;
			const u32 *in = l.in;
			u32 *out = l.out;
			void *args[] = {(void *)&in, (void *)&out, (void *)&a};
			return hipModuleLaunchKernel((hipFunction_t)rj->tile_x, g.tiles_per_plane * nruns, 1, 1, 256, 1, 1, 0, stream, args, nullptr);
		}
		for (int zi = l.roll_tile == 1 ? 3 : 2; zi >= 0; zi--)
		{
			const u32 Z = 2u << zi;
			const u32 runs1 = (planes1 + Z - 1u) / Z, nruns = runs1 + (planes2 + Z - 1u) / Z;
			void *fn = l.roll_tile == 1 ? rj->tile[zi] : rj->z[zi];
			if (shortest < Z || !fn) continue;
			// automatic choice: only launches of more than one resident generation (measured at 512^3, 2048 waves: the plain class
			// kernel 11.5 us, the rolling one 12.0; at 1024^3, 16384 waves: 80 vs 72). The tile form pays at 16 planes per thread
			// (two halo planes per 16 instead of per 8, and the LDS exchange replaces two of three row shifts: 1024^3 66.3 vs 68.8 us,
			// 2048^3 535 vs 574); at 8 planes its barrier per plane eats the saving (70.9 vs 68.8): there every thread shifts its own rows
			const bool tile_form = l.roll_tile == 1 && (l.roll_z || Z == 16u); // a forced depth (tests, tuning) takes the form roll_tile names
			if (!tile_form) fn = Z <= 8u ? rj->z[zi] : nullptr;
			if (!fn) continue;
			if (l.roll_z ? l.roll_z != (int)Z : (size_t)g.tiles_per_plane * nruns * 4u < 4096u) continue;
			RollArgs a;
			a.lo = l.pr.lo; a.hi = l.pr.hi; a.nplanes = l.pr.nplanes; a.wrap_full = l.pr.wrap_full; a.zbase = l.pr.zbase;
			a.lo2 = l.pr.lo2; a.hi2 = two ? l.pr.hi2 : l.pr.lo2; a.runs1 = runs1;
			const u32 *in = l.in;
			u32 *out = l.out;
			void *args[] = {(void *)&in, (void *)&out, (void *)&a};
			return hipModuleLaunchKernel((hipFunction_t)fn, g.tiles_per_plane * nruns, 1, 1, 256, 1, 1, tile_form ? 0u : lds_pad, stream, args, nullptr);
		}
	}
	if (np2_jit)
	{
		const u32 *in = l.in;
		u32 *out = l.out;
		PlaneRange pr = l.pr;
		PackedRuleArgs prog = l.rules->prog;
		void *args[] = {(void *)&in, (void *)&out, (void *)&pr, (void *)&g, (void *)&prog};
		return hipModuleLaunchKernel((hipFunction_t)(deep ? jit->deep_np2 : jit->flat_np2), grid_deep.x, 1, 1, 256, 1, 1, 0, stream, args, nullptr);
	}
	if (p2 && jit_ok)
	{
		// the run-time compiled kernel for exactly these rules (truth tables baked in: ca_jit.cpp). z-aligned: every
		// z-run starts on a global plane that is a multiple of ZRUN, so plane 0 is never in the middle of a run.
		auto aligned = [&](u32 lo, u32 hi) {
			const u32 zg = (u32)((l.pr.zbase + (int)lo) % (int)l.pr.G + (int)l.pr.G) % l.pr.G;
			return zg % (u32)ZRUN == 0 && (hi - lo) % (u32)ZRUN == 0;
		};
		const bool za = l.pr.G % (u32)ZRUN == 0 && aligned(l.pr.lo, l.pr.hi) && (!two || aligned(l.pr.lo2, l.pr.hi2));
		const u32 *in = l.in;
		u32 *out = l.out;
		PlaneRange pr = l.pr;
		PackedRuleArgs prog = l.rules->prog;
		void *args[] = {(void *)&in, (void *)&out, (void *)&pr, (void *)&g, (void *)&prog};
		void *fn = deep ? (za ? jit->deep_za : jit->deep) : jit->flat;
		return hipModuleLaunchKernel((hipFunction_t)fn, grid_deep.x, 1, 1, 256, 1, 1, 0, stream, args, nullptr);
	}
	if (p2)
	{
		if (deep && fast) CA3D_LAUNCH_CLASS(ZRUN, true, true, grid_deep);
		else if (deep) CA3D_LAUNCH_CLASS(ZRUN, false, true, grid_deep);
		else if (fast) CA3D_LAUNCH_CLASS(1, true, true, grid_flat);
		else CA3D_LAUNCH_CLASS(1, false, true, grid_flat);
	}
	else
	{
		// non-power-of-two grids (G = 384, 640, ...): one variant per rule shape is enough
		if (fast) CA3D_LAUNCH_CLASS(1, true, false, grid_flat);
		else CA3D_LAUNCH_CLASS(1, false, false, grid_flat);
	}
#undef CA3D_LAUNCH_CLASS
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

} // namespace

struct RowsArgs // ca_packed_rows_kernel.inc
{
	u32 runs1, nt;
};

bool roll_kernel_applies(const CanonRules &r, uint32_t G, int variant)
{
	if (!use_class_kernel(r, G, variant) || vn_kernel_applies(r, G, variant)) return false;
	const uint32_t cv = G / 128u;
	if (G % 128u || (cv & (cv - 1u)) || cv < 2u || cv > 64u) return false; // power-of-two rows of 2 .. 64 uint4: a row sits inside one wave
	// rules whose counts come from the thread's own plane only gain nothing from a z window
	const bool diagonals_or_z = r.need[1] || r.need[2] || r.main == MAIN_MOORE || r.main == MAIN_EDGES || r.main == MAIN_CORNERS || r.main == MAIN_VN;
	return diagonals_or_z;
}

// Rows of whole uint4, not a power of two of them (384, 640, 768, 896), and a rule whose counts reach into the planes above and
// below (every rule but the 2D neighbourhoods): the rolling-window kernel's whole-rows-per-wave form (ca_packed_roll_kernel.inc,
// roll_step_np2). Measured against the kernels these grids had (us per step, MI355X, profiles/r4_zz_roll_np2.txt): clustered
// 384 8.69 -> 8.88, 640 29.6 -> 26.4, 768 51.4 -> 39.6, 896 69.0 -> 53.7; start-up rule 5.43 -> 4.66, 11.9 -> 11.4, 20.5 -> 18.0,
// 32.6 -> 27.0. CA3D_ROLL_NP2 = 0 keeps the earlier kernels (rows / class np2; tuning).
bool roll_np2_applies(const CanonRules &r, uint32_t G, int variant)
{
	static const int env = getenv("CA3D_ROLL_NP2") ? atoi(getenv("CA3D_ROLL_NP2")) : 1;
	if (!env || variant == 1 || !r.fast || G % 128u) return false;
	const uint32_t cv = G / 128u;
	if ((cv & (cv - 1u)) == 0 || cv < 3u || cv > 7u) return false;
	return r.need[1] || r.need[2] || r.main == MAIN_MOORE || r.main == MAIN_EDGES || r.main == MAIN_CORNERS || r.main == MAIN_VN;
}

int class_zrun(const CanonRules &r)
{
	const bool diagonals = r.need[1] || r.need[2] || r.main == MAIN_MOORE || r.main == MAIN_EDGES || r.main == MAIN_CORNERS;
	return diagonals ? 4 : 2; // must match ZRUN of launch_class
}

// Grids the uint4 kernels serve without their compile-time-rule forms (G % 128 == 0 but not a power of two: cube programs, extra
// edge loads) or not at all (rows that are not whole uint4: ca_packed_generic): the rows kernel (ca_packed_rows_kernel.inc, run-time
// compiled for the grid and the rule) takes them.
bool rows_kernel_applies(const CanonRules &r, uint32_t G, int variant)
{
	if (variant == 1 || !r.fast || G % 32u || G < 32u || G > 2048u) return false;
	const uint32_t cv = G / 128u;
	const bool p2_uint4 = G % 128u == 0 && (cv & (cv - 1u)) == 0; // vn / class (jit) / rolling kernels
	if (p2_uint4) return false;
	// Rows of whole uint4 that are not a power of two of them (384, 640, 768, 896): the class kernel's run-time compiled form serves
	// them as well (ca3d_jit_class_*_np2: uint4 loads, two extra dword loads per row for the edge words). Measured (us per step,
	// class np2 / rows with its z-run chosen by grid; profiles/r4_s_rows_vs_class_np2.txt): start-up rule 384 5.2 / 4.7, 640 13.5 / 12.2,
	// 768 22.4 / 25.1, 896 30.7 / 32.0; clustered 384 12.7 / 8.5, 640 35.9 / 30.2, 768 54.5 / 50.8, 896 81.3 / 70.9 — face-only rules on
	// the largest of these grids stay with the uint4 loads, everything else (rules with diagonal classes: nine rows per plane, the edge
	// loads triple) takes the rows kernel. CA3D_ROWS_NP2 = 0 / 1 forces one of them (tuning).
	static const int np2_env = getenv("CA3D_ROWS_NP2") ? atoi(getenv("CA3D_ROWS_NP2")) : -1;
	if (G % 128u == 0) return np2_env >= 0 ? np2_env != 0 : (class_zrun(r) == 4 || G < 768u);
	return true;
}

bool use_class_kernel(const CanonRules &r, uint32_t G, int variant)
{
	return variant != 1 && r.fast && ((G / 32u) % 4u) == 0;
}


const char *packed_kernel_name(const CanonRules &r, uint32_t G, int variant)
{
	if (vn_kernel_applies(r, G, variant)) return "ca_packed_vn";
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
	if (variant != 0 || !use_class_kernel(r, G, variant)) return 0;
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
	if (l.pr.hi <= l.pr.lo)
	{
		if (l.pr.hi2 <= l.pr.lo2) return hipSuccess;
		PackedLaunch only = l; // only the second range has planes
		only.pr.lo = l.pr.lo2; only.pr.hi = l.pr.hi2; only.pr.lo2 = only.pr.hi2 = 0;
		return launch_packed_step(only, stream, nullptr);
	}
	if (vn_kernel_applies(r, l.pr.G, l.variant)) return launch_packed_vn(l, stream);
	if (const RollJit *rj = l.roll_jit; rj && rj->cv_np2 > 0 && (u32)rj->cv_np2 * 128u == l.pr.G && rj->main == (int)r.main && rj->e == r.need[1] && rj->c == r.need[2])
	{
		// rows of 3 / 5 / 6 / 7 uint4: the rolling-window kernel with whole rows per wave (roll_step_np2). Z = the deepest of 8 / 4 / 2 planes
		// per thread that the shortest range holds and that still leaves two waves for every SIMD; ranges shorter than two planes
		// (a slab's edge phase) fall through to the kernels below.
		const bool two = l.pr.hi2 > l.pr.lo2;
		const u32 G = l.pr.G, CV = G / 128u, R = 64u / CV, bpp = ((G + R - 1u) / R + 3u) / 4u;
		const u32 planes1 = l.pr.hi - l.pr.lo, planes2 = two ? l.pr.hi2 - l.pr.lo2 : 0u;
		const u32 shortest = two ? (planes1 < planes2 ? planes1 : planes2) : planes1;
		for (int zi = 2; zi >= 0; zi--)
		{
			const u32 Z = 2u << zi;
			if (shortest < Z || !rj->np2[zi]) continue;
			const u32 runs1 = (planes1 + Z - 1u) / Z, nruns = runs1 + (planes2 + Z - 1u) / Z;
			if (l.roll_z ? l.roll_z != (int)Z : (zi > 0 && (size_t)bpp * nruns < 512u)) continue; // option roll_z forces a depth (tests, tuning)
			RollArgs a;
			a.lo = l.pr.lo; a.hi = l.pr.hi; a.nplanes = l.pr.nplanes; a.wrap_full = l.pr.wrap_full; a.zbase = l.pr.zbase;
			a.lo2 = l.pr.lo2; a.hi2 = two ? l.pr.hi2 : l.pr.lo2; a.runs1 = runs1;
			const u32 *in = l.in;
			u32 *out = l.out;
			void *args[] = {(void *)&in, (void *)&out, (void *)&a};
			if (kernel_name) *kernel_name = "ca_packed_roll_np2(jit)";
			return hipModuleLaunchKernel((hipFunction_t)rj->np2[zi], bpp * nruns, 1, 1, 256, 1, 1, 0, stream, args, nullptr);
		}
	}
	if (const RowsJit *rj = l.rows_jit; rj && rj->G == l.pr.G && rj->main == (int)r.main && rj->e == r.need[1] && rj->c == r.need[2] && rj->deep && rj->flat)
	{
		const bool two = l.pr.hi2 > l.pr.lo2;
		const u32 G = l.pr.G, C = G / 32u, R = C >= 64u ? 1u : 64u / C, bpp = ((G + R - 1u) / R + 3u) / 4u;
		const u32 planes1 = l.pr.hi - l.pr.lo, planes2 = two ? l.pr.hi2 - l.pr.lo2 : 0u;
		const u32 shortest = two ? (planes1 < planes2 ? planes1 : planes2) : planes1;
		const u32 Z = shortest >= (u32)rj->zrun && (size_t)bpp * ((planes1 + planes2) / (u32)rj->zrun) >= 1024u ? (u32)rj->zrun : 1u; // deep only while it fills the chip
		RowsArgs a;
		a.runs1 = (planes1 + Z - 1u) / Z;
		a.nt = (size_t)l.pr.nplanes * G * C * sizeof(u32) <= (16u << 20) ? 1u : 0u; // as ca_packed_vn: while both buffers sit in the Infinity Cache with room to spare
		const u32 nruns = a.runs1 + (planes2 + Z - 1u) / Z;
		const u32 *in = l.in;
		u32 *out = l.out;
		PlaneRange pr = l.pr;
		if (!two) pr.lo2 = pr.hi2 = 0;
		void *args[] = {(void *)&in, (void *)&out, (void *)&pr, (void *)&a};
		if (kernel_name) *kernel_name = "ca_packed_rows(jit)";
		return hipModuleLaunchKernel((hipFunction_t)(Z > 1u ? rj->deep : rj->flat), bpp * nruns, 1, 1, 256, 1, 1, 0, stream, args, nullptr);
	}
	if (l.pr.hi2 > l.pr.lo2 && !use_class_kernel(r, l.pr.G, l.variant))
	{
		// the generic kernel takes one range per launch
		PackedLaunch a = l, b = l;
		a.pr.lo2 = a.pr.hi2 = 0;
		b.pr.lo = l.pr.lo2; b.pr.hi = l.pr.hi2; b.pr.lo2 = b.pr.hi2 = 0;
		hipError_t e = launch_packed_step(a, stream, nullptr);
		return e != hipSuccess ? e : launch_packed_step(b, stream, nullptr);
	}
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
