// The converged frame of a dense packed volume as a RAY STREAM (gfx950): the same frame as render.hip's plain kernel, bit for
// bit, drawn by four lean passes instead of one kernel that carries a sample from its view ray to its colour.
//
//   ca_stream_walk<primary>   persistent; lanes take sample jobs of the volume's screen rectangle from queues of 64-job chunks,
//                             form the view ray (shade_sample up to its walk) and walk it; answer per job: hit distance | none
//   ca_stream_shadow_rays     one lane per job: shade_sample with the answer looked up, up to the shadow ray; lit jobs leave the
//                             point the shadow ray starts from
//   ca_stream_walk<shadow>    persistent, the same stepping loop over the shadow rays; answer per lit job: occluded or not
//   ca_stream_resolve         one lane per pixel: shade_sample with both answers looked up, samples summed in order, three outputs
//
// Why: in one kernel (ca_render_packed_sched) the stepping loop was ~110 vector instructions per cell at 30 of 64 lanes, compiled
// at the 128-register edge the shading code sets (4 waves per SIMD), and a third of it was the slab test of the one or two lanes
// that stood on a live cell. Here the stepping loop stands alone: ~17 registers of state, 8 waves per SIMD, a lane that finishes is
// refilled from the queue at once (no tile to wait for), and the live-cell test is a conservative interval filter on the walk's own
// boundary times (14 instructions) that calls the reference's slab test only when the two disagree within rounding — so the cells
// visited, the hits found and the distances reported are those of the plain walk (pathtraced_fragment_clustered.wgsl:212-225 for
// the test, 682-741 / 635-680 for what the walks replace). Everything per-sample and expensive (view ray, shading gate,
// Cook-Torrance) runs in the two one-lane-per-job passes at full lane occupancy.
//
// Built with -ffp-contract=off like render.hip: every pass instantiates the same shade_sample_with, and the walks' boundary times are
// accumulated by the same float operations as render_device.inc's walk().
#include <hip/hip_fp16.h>

#include <cstdio>
#include <vector>

#include "ca3d_internal.h"

namespace ca3d
{
namespace
{

#include "render_device.inc"

constexpr u32 kNoHit = 0xFFFFFFFFu; // a hit distance is never NaN (the slab test rejects NaN)
constexpr u32 kUnanswered = 0x5A5A5A5Au; // check build: what the answer arrays are filled with before the passes (a distance of 1.5e16)
enum : unsigned char { kOcclNone = 0, kOcclHit = 1, kOcclPending = 2 };

#ifndef CA3D_STREAM_WAVES
#define CA3D_STREAM_WAVES 8
#endif
constexpr int kStreamWaves = CA3D_STREAM_WAVES; // waves per SIMD the walk kernel is compiled for
constexpr int kStatSlots = 64;
#ifndef CA3D_STREAM_INNER
#define CA3D_STREAM_INNER 1
#endif
constexpr int kInner = CA3D_STREAM_INNER; // cells a lane may advance per read of the volume (stepping loop; 3 measured SLOWER: below)

struct StreamParams
{
	RenderParams R;
	u32 *hit;            // per job: the view ray's answer (float bits of the hit cube's slab entry, kNoHit)
	unsigned char *occl; // per job: kOcclPending (lit, shadow ray waiting) -> kOcclHit / kOcclNone
	float4 *rays;        // per lit job: the point the shadow ray starts from
	u32 *ctl;            // [2] filter / slab-test contradictions (check build); [16 + 4 * slot + {0 shadow rays, 1 primary visits, 2 shadow
	                     // visits}] statistics, kStatSlots slots; [kQueueWords + 32 * (8 * pass + q)] head of queue q of walk pass `pass`
	u32 chunks, chunks_x; // chunks of 64 jobs = pixel blocks (4 x 4 pixels at 4 samples, 8 x 8 at one) of the rectangle, row-major;
	                      // jobs of chunk c: [64 c, 64 c + 64), pixel-major, a pixel's samples next to each other
	const u32 *volume;   // what the walks read: R.cells, or the bricked copy of it
	u32 lg, lc;          // log2 G, log2 cols (power-of-two grids)
	u32 nb;              // bricks per edge, G / 8
	u32 lnbp;            // packed cell positions (kBricksPacked): bits of a brick coordinate, log2 nb
	u32 probe_mask;      // timing probes (kProbe*): 0
	int tail_batch;      // ca_stream_walk2: 1 = the batched stepping loop once a wave cannot refill any more and in the drain launch
	int refill2;         // ca_stream_walk2: idle lanes at which a wave leaves the stepping loop to pop prepared rays (tuning: CA3D_STREAM_POP)
	int refill;          // lanes without a ray at which a wave leaves the stepping loop to take new jobs (tuning: CA3D_STREAM_REFILL)
	u32 lb;              // log2 of a chunk's pixel-block edge; jobs per chunk = spp << (2 lb)
	u32 qmap;            // 0: queue q owns a contiguous eighth of the chunks (a band of the image); 1: every eighth chunk
	u32 skip_ok;         // 1: sparse (scattered) volumes are drawn by the stream passes too — ca_stream_walk2<.., SKIP = true> (round 5, late)
	u32 check;           // diagnostics build: answers were pre-set to kUnanswered and whoever looks one up counts those still unset in ctl[3]
	unsigned long long *trace; // diagnostics (CA3D_STREAM_TRACE=<file>): 8 words per wave and walk pass — start, end (s_memrealtime, 100 MHz),
	                           // refill rounds, stepping iterations, chunks taken, ticks spent refilling, jobs started, 0
};
constexpr u32 kQueueWords = 512, kNoChunk = 0xFFFFFFFFu;

// Which frames the stream passes draw: dense volumes; and, with skip_ok, sparse volumes whose live cells are scattered (the block-skipping
// walk inside ca_stream_walk2<.., true>). A sparse volume with a small live box stays with render.hip's spread kernel.
__device__ __forceinline__ bool stream_sparse_frame(const StreamParams &S) { return S.skip_ok && occ_skip_enabled(S.R) && !live_box_small(S.R); }
__device__ __forceinline__ bool stream_owns_frame(const StreamParams &S) { return !occ_skip_enabled(S.R) || stream_sparse_frame(S); }

__device__ __forceinline__ u32 job_shift(const RenderParams &P) { return P.spp == 4u ? 2u : 0u; }

// pixel of job r of chunk c
__device__ __forceinline__ void job_pixel(const StreamParams &S, u32 c, u32 r, u32 &px, u32 &py, u32 &k)
{
	const RenderParams &P = S.R;
	const u32 sh = job_shift(P), q = r >> sh, lb = S.lb;
	k = r & ((1u << sh) - 1u);
	px = P.rx0 + ((c % S.chunks_x) << lb) + (q & ((1u << lb) - 1u));
	py = P.ry0 + ((c / S.chunks_x) << lb) + (q >> lb);
}

__device__ __forceinline__ void sample_uv(const RenderParams &P, u32 px, u32 py, u32 k, float &vu, float &vv)
{
	// the plain kernel's sample positions (ca_render_packed)
	const float ox = P.spp == 1u ? 0.5f : ((k & 1u) ? 0.75f : 0.25f);
	const float oy = P.spp == 1u ? 0.5f : ((k & 2u) ? 0.75f : 0.25f);
	vu = ((float)px + ox) / (float)P.W;
	vv = 1.0f - ((float)py + oy) / (float)P.H;
}

// ---- tracers of the three non-walking uses of shade_sample_with -------------------------------------------------------------
struct CapturePrimary // first pass: what the view ray's walk is given
{
	static constexpr bool kSkipBox = false, kStopAfterPrimary = true, kStopAfterShadow = false;
	bool wanted = false;
	v3 enter, dir;
	float len;
	__device__ __forceinline__ bool primary(const RenderParams &, v3, v3, v3 e, v3 d, float l, v3, float &)
	{
		wanted = true; enter = e; dir = d; len = l;
		return false;
	}
	__device__ __forceinline__ bool shadow(const RenderParams &, v3, v3, float, v3, int, int, int) { return false; }
};

struct LookupPrimaryRecordShadow // second pass
{
	static constexpr bool kSkipBox = false, kStopAfterPrimary = false, kStopAfterShadow = true;
	u32 answer;
	bool lit = false, unanswered = false;
	v3 p;
	__device__ __forceinline__ bool primary(const RenderParams &, v3, v3, v3, v3, float, v3, float &tnear)
	{
		unanswered = answer == kUnanswered;
		if (answer == kNoHit) return false;
		tnear = __uint_as_float(answer);
		return true;
	}
	__device__ __forceinline__ bool shadow(const RenderParams &, v3 from, v3, float, v3, int, int, int)
	{
		lit = true; p = from;
		return false;
	}
};

struct LookupBoth // last pass
{
	static constexpr bool kSkipBox = false, kStopAfterPrimary = false, kStopAfterShadow = false;
	u32 answer;
	const unsigned char *occl;
	bool unanswered = false;
	__device__ __forceinline__ bool primary(const RenderParams &, v3, v3, v3, v3, float, v3, float &tnear)
	{
		if (answer == kNoHit) return false;
		tnear = __uint_as_float(answer);
		return true;
	}
	__device__ __forceinline__ bool shadow(const RenderParams &, v3, v3, float, v3, int, int, int)
	{
		unanswered = *occl != kOcclHit && *occl != kOcclNone;
		return *occl == kOcclHit;
	}
};

// ---- the stepping loop ------------------------------------------------------------------------------------------------------
// Per-lane walk state: the Amanatides-Woo walk of render_device.inc's walk(), same float operations in the same order.
struct Walker
{
	float tx, ty, tz, dx, dy, dz, t, tmax;
	int ix, iy, iz, sx, sy, sz;
	int wkey;
	u32 word;
	float eps_b; // constant term of the filter's error bound (below); +inf: always ask the slab test
	u32 pos, neg, edge; // kBricksPacked: the cell as ONE word (PackGeom below) instead of ix .. sz
};

// Where the walk reads the volume: the packed state itself (row-major words of 32 x-cells; power-of-two grids index it by shifts) or a
// BRICKED copy (render_frame.hip: 8 x 8 x 8 cells = 64 contiguous bytes, word = 8 x-cells x 4 y-rows of one z-plane). The walk steps
// from cell to neighbouring cell, and in the row-major layout every step in y leaves the 64-byte row segment and every step in z the
// 32 KiB plane: the 64 lanes of a wave, a few cells apart, read as many cache lines as there are lanes, and the kernels measured bound
// by exactly that — the vector L1 taking a line per cycle (137 M accesses in 0.96 M cycles per CU x 256 CUs), insensitive to the waves
// per SIMD (4 or 8) and to ten fewer instructions per cell. In bricks seven of eight steps along any axis stay inside the 64 bytes.
// kBricksRead: bricks, and the walker reads its cell's word at every cell instead of keeping the last word in a register (no key to
// compare, no branch around the read; the default). Measured on top of each other (1080p / 4K, 4 samples, ms per frame): rows 0.88 /
// 2.32, bricks 0.80 / 1.98, bricks + a read per cell 0.785 / 1.94; a whole 8 x 8 z-slice per lane in two registers 0.805 / 2.11.
// kBricksReadAny: the same over the bricks of a grid that is not a power of two (96, 160, ..., 992): G / 8 bricks per edge, padded to a power
// of two in the address (render_frame.hip, ca_brick_volume_any), so the brick index is shifts there too.
// kProbe*: timing probes of the stepping loop (CA3D_STREAM_PROBE=1..4; the frame is garbage: no cell is ever taken for live, every walk runs to
// the end of the volume, so the four differ ONLY in what the read costs): no read at all / every lane reads word 0 (one cache line per
// wave-level read) / word key & 1023 (a 4 KiB footprint, as many lines per read as the real walk) / the real word.
enum { kRowsP2 = 0, kRowsAny = 1, kBricks = 2, kBricksRead = 3, kBricksReadAny = 4, kProbeNone = 5, kProbeSame = 6, kProbeSmall = 7, kProbeFull = 8, kBricksPacked = 9 };

template <int LAYOUT>
__device__ __forceinline__ int word_key(const StreamParams &S, int ix, int iy, int iz)
{
	if (LAYOUT == kBricksReadAny)
	{
		const u32 b = (((((u32)iz >> 3) << S.lnbp) + ((u32)iy >> 3)) << S.lnbp) + ((u32)ix >> 3); // (bricks per edge padded to a power of two in the address)
		return (int)((b << 4) + (((u32)iz & 7u) << 1) + (((u32)iy & 7u) >> 2));
	}
	if (LAYOUT == kBricks || LAYOUT == kBricksRead || LAYOUT >= kProbeNone)
	{
		const u32 lnb = S.lg - 3u;
		const u32 b = (((((u32)iz >> 3) << lnb) + ((u32)iy >> 3)) << lnb) + ((u32)ix >> 3);
		return (int)((b << 4) + (((u32)iz & 7u) << 1) + (((u32)iy & 7u) >> 2));
	}
	if (LAYOUT == kRowsP2) return (int)(((((u32)iz << S.lg) + (u32)iy) << S.lc) + ((u32)ix >> 5));
	return (ix >> 5) + (iy + iz * (int)S.R.G) * (int)S.R.cols;
}
template <int LAYOUT>
__device__ __forceinline__ u32 word_bit(int ix, int iy)
{
	return (LAYOUT == kBricks || LAYOUT == kBricksRead || LAYOUT == kBricksReadAny || LAYOUT >= kProbeNone) ? (((u32)ix & 7u) | (((u32)iy & 3u) << 3)) : ((u32)ix & 31u);
}

// walk_begin of render.hip / the head of walk(): first cell, boundary times, increments
__device__ __forceinline__ void walker_begin(const RenderParams &P, Walker &w, v3 start, v3 dir, float t0, float tmax, float *ctx, int stride)
{
	const int G = (int)P.G;
	const float cs = 1.0f / (float)P.G;
	const v3 p = start + dir * t0;
	int ix = (int)floorf(to_cells(P, p.x)), iy = (int)floorf(to_cells(P, p.y)), iz = (int)floorf(to_cells(P, p.z));
	ix = min(max(ix, 0), G - 1);
	iy = min(max(iy, 0), G - 1);
	iz = min(max(iz, 0), G - 1);
	const int sx = dir.x > 0.0f ? 1 : -1, sy = dir.y > 0.0f ? 1 : -1, sz = dir.z > 0.0f ? 1 : -1;
	const float big = 3.0e38f;
	w.tx = dir.x != 0.0f ? (((float)(ix + (sx > 0 ? 1 : 0)) * cs - kHalf) - start.x) / dir.x : big;
	w.ty = dir.y != 0.0f ? (((float)(iy + (sy > 0 ? 1 : 0)) * cs - kHalf) - start.y) / dir.y : big;
	w.tz = dir.z != 0.0f ? (((float)(iz + (sz > 0 ? 1 : 0)) * cs - kHalf) - start.z) / dir.z : big;
	w.dx = dir.x != 0.0f ? cs / fabsf(dir.x) : big;
	w.dy = dir.y != 0.0f ? cs / fabsf(dir.y) : big;
	w.dz = dir.z != 0.0f ? cs / fabsf(dir.z) : big;
	w.t = t0;
	w.tmax = tmax;
	w.ix = ix; w.iy = iy; w.iz = iz;
	w.sx = sx; w.sy = sy; w.sz = sz;
	w.word = 0;
	w.wkey = -1;
	// Error bound of the interval filter (walk_cell below). The filter reads the hit cube's slab times off the walk's boundary times:
	// along axis a the cell is left at ta (accumulated: a division, then k <= t / da + 1 additions of da, each rounded) and the cube of
	// half-size h spans [ta - k1 da, ta - k0 da], k0 = 1/2 - h / cs. Against the real slab times that is off by at most
	// 2^-24 (t^2 / da + 3 t + 2 da) per axis; the reference's slab test ((c -+ h) - o) * (1 / d) is itself off by 2^-24 (G da + 2 t).
	// With sum 1 / da = G sum |d| <= 1.74 G:  eps(t) = 2^-21 G t^2 + 2^-18 t + 2^-20 (G + 8) (dx + dy + dz) covers the sum of both
	// several times over, and a zero direction component (its slab times are +-inf in the reference's test) makes it infinite.
	const bool axis_parallel = dir.x == 0.0f || dir.y == 0.0f || dir.z == 0.0f;
	w.eps_b = axis_parallel ? __builtin_inff() : 9.5367431640625e-7f * ((float)P.G + 8.0f) * (w.dx + w.dy + w.dz);
	// what the slab test reads: ray origin and 1 / direction — in LDS, not in registers: only a lane whose filter cannot decide
	// and a lane that reports a hit look at them
	ctx[0 * stride] = start.x; ctx[1 * stride] = start.y; ctx[2 * stride] = start.z;
	ctx[3 * stride] = 1.0f / dir.x; ctx[4 * stride] = 1.0f / dir.y; ctx[5 * stride] = 1.0f / dir.z;
}

// the reference's slab test of the cube in the walker's cell (ray_cube_inv: bit-identical to ray_cube)
template <bool SHADOW>
__device__ __forceinline__ bool slab_test_at(const RenderParams &P, int ix, int iy, int iz, v3 vhalf, const float *ctx, int stride, float &tn)
{
	const v3 start = V(ctx[0 * stride], ctx[1 * stride], ctx[2 * stride]);
	const v3 inv = V(ctx[3 * stride], ctx[4 * stride], ctx[5 * stride]);
	float tf;
	ray_cube_inv(start, inv, cell_origin(1.0f / (float)P.G, ix, iy, iz), vhalf, tn, tf);
	return SHADOW ? (tn <= tf && tn >= 0.0f) /* :668 */ : (tf >= 0.0f && tn <= tf) /* :722-729 */;
}
template <bool SHADOW>
__device__ __forceinline__ bool slab_test(const RenderParams &P, const Walker &w, v3 vhalf, const float *ctx, int stride, float &tn)
{
	return slab_test_at<SHADOW>(P, w.ix, w.iy, w.iz, vhalf, ctx, stride, tn);
}

template <bool SHADOW, int LAYOUT, bool CHECK>
__device__ __forceinline__ int walk_cell_word(const StreamParams &S, Walker &w, u32 cur, bool &exempt, v3 vhalf, float k0, float k1, float eps_a,
                                              const float *ctx, int stride);

// One cell of the walk. Returns 0 keep walking, 1 hit, 2 the ray left the volume or ran out of range.
//   k0, k1: the cube's slab offsets in units of the cell's crossing time (walker_begin); eps_a = 2^-21 G
//   LOAD: the lane may read the volume (the first pass of an iteration of the stepping loop); false: the caller has made sure the
//   cell lies in what the lane holds (key == w.wkey)
template <bool SHADOW, int LAYOUT, bool CHECK, bool LOAD>
__device__ __forceinline__ int walk_cell(const StreamParams &S, Walker &w, int key, bool &exempt, v3 vhalf, float k0, float k1, float eps_a,
                                         const float *ctx, int stride)
{
	const RenderParams &P = S.R;
	// (a 32-bit byte offset from the scalar base: one shift instead of a sign extension and a 64-bit add per visit)
	u32 cur;
	if (LAYOUT >= kProbeNone)
	{
		const u32 k2 = LAYOUT == kProbeSame ? 0u : (LAYOUT == kProbeSmall ? ((u32)key & 1023u) : (u32)key);
		cur = LAYOUT == kProbeNone ? (u32)key : *reinterpret_cast<const u32 *>(reinterpret_cast<const char *>(S.volume) + (k2 << 2));
		cur &= S.probe_mask; // 0 at run time: the compiler cannot drop the read, and no cell is live
	}
	else if (LAYOUT == kBricksRead || LAYOUT == kBricksReadAny) cur = *reinterpret_cast<const u32 *>(reinterpret_cast<const char *>(S.volume) + ((u32)key << 2));
	else
	{
		if (LOAD && key != w.wkey) { w.word = *reinterpret_cast<const u32 *>(reinterpret_cast<const char *>(S.volume) + ((u32)key << 2)); w.wkey = key; }
		cur = w.word;
	}
	return walk_cell_word<SHADOW, LAYOUT, CHECK>(S, w, cur, exempt, vhalf, k0, k1, eps_a, ctx, stride);
}

// walk_cell behind its read: `cur` is the word of the walker's cell
template <bool SHADOW, int LAYOUT, bool CHECK>
__device__ __forceinline__ int walk_cell_word(const StreamParams &S, Walker &w, u32 cur, bool &exempt, v3 vhalf, float k0, float k1, float eps_a,
                                              const float *ctx, int stride)
{
	const RenderParams &P = S.R;
	if (((cur >> word_bit<LAYOUT>(w.ix, w.iy)) & 1u) && !exempt)
	{
		const float tn = fmaxf(fmaxf(__builtin_fmaf(-k1, w.dx, w.tx), __builtin_fmaf(-k1, w.dy, w.ty)), __builtin_fmaf(-k1, w.dz, w.tz));
		const float tf = fminf(fminf(__builtin_fmaf(-k0, w.dx, w.tx), __builtin_fmaf(-k0, w.dy, w.ty)), __builtin_fmaf(-k0, w.dz, w.tz));
		const float gap = tf - tn, lead = SHADOW ? tn : tf;
		const float eps = __builtin_fmaf(__builtin_fmaf(eps_a, w.t, 3.814697265625e-6f), w.t, w.eps_b);
		const bool yes = gap > eps && lead > eps, no = gap < -eps || lead < -eps;
		if (CHECK)
		{
			float e;
			const bool truth = slab_test<SHADOW>(P, w, vhalf, ctx, stride, e);
			if ((yes && !truth) || (no && truth)) atomicAdd(&S.ctl[2], 1u);
			if (truth) return 1;
		}
		else
		{
			if (yes) return 1;
			if (!no)
			{
				float e;
				if (slab_test<SHADOW>(P, w, vhalf, ctx, stride, e)) return 1;
			}
		}
	}
	// advance along the axis whose boundary comes first (walk(): mx = tx <= ty && tx <= tz, my = !mx && ty <= tz)
	const float t = fminf(fminf(w.tx, w.ty), w.tz);
	const bool mx = w.tx == t, my = !mx && w.ty == t, mz = !mx && !my;
	w.t = t;
	w.tx += mx ? w.dx : 0.0f;
	w.ty += my ? w.dy : 0.0f;
	w.tz += mz ? w.dz : 0.0f;
	w.ix += mx ? w.sx : 0;
	w.iy += my ? w.sy : 0;
	w.iz += mz ? w.sz : 0;
	exempt = false; // the walk is monotone: once it has left the cell it started in it does not come back
	// (one v_max3_u32; a negative coordinate is a huge unsigned one. `ix | iy | iz` >= G, the first form, is the same test on power-of-two grids
	// only: at G = 96 the cell (64, 32, z) reads as outside — a tenth of the pixels of a 96^3 frame were wrong until round 5 tested one)
	const bool outside = max(max((u32)w.ix, (u32)w.iy), (u32)w.iz) >= P.G;
	return (outside || t >= w.tmax) ? 2 : 0;
}

// ---- the batched form of the stepping loop -----------------------------------------------------------------------------------
// The walk's next cell depends on arithmetic only (the boundary times), never on what the volume holds: a lane can run kBatch cells
// ahead, ask for all their words at once and test them in order when they arrive — ONE memory round trip per kBatch cells instead
// of one per cell (the literal frame's batched march, render_frame.hip, for a walk whose positions come from a DDA). A batch:
//   forward   kBatch x { key and bit of the cell, read issued, advance } from a saved copy of the boundary state; cells past the walk's
//             end are masked;
//   test      the live bits of the valid cells (the shadow ray's start cell exempt), a mask per lane;
//   replay    lanes with a live cell (one in eight batches at the bench scene's density) go back to the saved state and advance to each
//             live cell in turn for the interval filter / slab test — a hit ends the walk there, with the walker standing on the
//             hit cell as the retire step expects; a lane without a hit ends where the forward pass ended.
// Cells visited, hits and the state a hit leaves are those of the cell-by-cell loop (the same advance, the same filter).
// Measured (CA3D_STREAM_BRICKS=3 selects it; bit-identical frames, the check mode passes): 1080p 4 spp 0.888 against 0.900 ms, one
// sample 0.529 against 0.549, 4K 2.27 against 2.14 — a quarter of the memory round trips and no faster: like every other probe of
// the stepping loop (see the cell-by-cell loop below) it says the walks are not waiting for the volume. Kept as an option.
#ifndef CA3D_STREAM_BATCH
#define CA3D_STREAM_BATCH 4
#endif
constexpr int kBatch = CA3D_STREAM_BATCH;

// advance along the axis whose boundary comes first; true: the walk is over (left the volume or ran out of range)
__device__ __forceinline__ bool walk_advance(const RenderParams &P, Walker &w)
{
	const float t = fminf(fminf(w.tx, w.ty), w.tz);
	const bool mx = w.tx == t, my = !mx && w.ty == t, mz = !mx && !my;
	w.t = t;
	w.tx += mx ? w.dx : 0.0f;
	w.ty += my ? w.dy : 0.0f;
	w.tz += mz ? w.dz : 0.0f;
	w.ix += mx ? w.sx : 0;
	w.iy += my ? w.sy : 0;
	w.iz += mz ? w.sz : 0;
	const bool outside = max(max((u32)w.ix, (u32)w.iy), (u32)w.iz) >= P.G;
	return outside || t >= w.tmax;
}

// the live cell the walker stands on: does the ray meet its visible cube? (interval filter, slab test inside its error band)
template <bool SHADOW, bool CHECK>
__device__ __forceinline__ bool walk_hit_test(const StreamParams &S, const Walker &w, v3 vhalf, float k0, float k1, float eps_a, const float *ctx, int stride)
{
	const RenderParams &P = S.R;
	const float tn = fmaxf(fmaxf(__builtin_fmaf(-k1, w.dx, w.tx), __builtin_fmaf(-k1, w.dy, w.ty)), __builtin_fmaf(-k1, w.dz, w.tz));
	const float tf = fminf(fminf(__builtin_fmaf(-k0, w.dx, w.tx), __builtin_fmaf(-k0, w.dy, w.ty)), __builtin_fmaf(-k0, w.dz, w.tz));
	const float gap = tf - tn, lead = SHADOW ? tn : tf;
	const float eps = __builtin_fmaf(__builtin_fmaf(eps_a, w.t, 3.814697265625e-6f), w.t, w.eps_b);
	const bool yes = gap > eps && lead > eps, no = gap < -eps || lead < -eps;
	float e;
	if (CHECK)
	{
		const bool truth = slab_test<SHADOW>(P, w, vhalf, ctx, stride, e);
		if ((yes && !truth) || (no && truth)) atomicAdd(&S.ctl[2], 1u);
		return truth;
	}
	if (yes) return true;
	if (no) return false;
	return slab_test<SHADOW>(P, w, vhalf, ctx, stride, e);
}

// One batch for a lane with a ray. Returns the cells it visited; term: 0 keep walking, 1 hit (the walker stands on the cell), 2 over.
template <bool SHADOW, int LAYOUT, bool CHECK>
__device__ __forceinline__ u32 walk_batch(const StreamParams &S, Walker &w, bool &exempt, int &term, v3 vhalf, float k0, float k1, float eps_a,
                                          const float *ctx, int stride)
{
	const RenderParams &P = S.R;
	const float stx = w.tx, sty = w.ty, stz = w.tz, st = w.t;
	const int six = w.ix, siy = w.iy, siz = w.iz;
	u32 word[kBatch], nvalid = 0;
	unsigned long long pos = 0; // 6 bits per cell — bit position, 32 for a cell past the walk's end
	bool over = false;
#pragma unroll
	for (int k = 0; k < kBatch; k++)
	{
		const int key = over ? 0 : word_key<LAYOUT>(S, w.ix, w.iy, w.iz);
		word[k] = *reinterpret_cast<const u32 *>(reinterpret_cast<const char *>(S.volume) + ((u32)key << 2));
		pos |= (unsigned long long)(over ? 32u : word_bit<LAYOUT>(w.ix, w.iy)) << (6 * k);
		nvalid += over ? 0u : 1u;
		over = walk_advance(P, w) || over;
	}
	u32 live = 0;
#pragma unroll
	for (int k = 0; k < kBatch; k++)
	{
		const u32 b = (u32)(pos >> (6 * k)) & 63u;
		live |= (b < 32u ? (word[k] >> b) & 1u : 0u) << k;
	}
	if (exempt) live &= ~1u; // the cell the shadow ray starts in: only ever the first cell of a walk
	exempt = false;
	u32 visited = nvalid;
	bool hit = false;
	if (live)
	{
		w.tx = stx; w.ty = sty; w.tz = stz; w.t = st; w.ix = six; w.iy = siy; w.iz = siz;
#pragma unroll
		for (int k = 0; k < kBatch; k++)
		{
			if (!hit)
			{
				if ((live >> k) & 1u)
				{
					if (walk_hit_test<SHADOW, CHECK>(S, w, vhalf, k0, k1, eps_a, ctx, stride)) { hit = true; visited = (u32)k + 1u; }
				}
				if (!hit) walk_advance(P, w);
			}
		}
	}
	term = hit ? 1 : (over ? 2 : 0);
	return visited;
}

// ---- packed cell positions (kBricksPacked; round 5, late) ------------------------------------------------------------------------
// The ledger of the stepping loop (profiles/r5_render_walk_ledger.txt) prices an empty cell at 44 VALU: 10 for the bricked word's key,
// 6 for the bit, 23 for the advance (three cell coordinates and three signs next to the three boundary times). Here a walker carries
// its cell as ONE word laid out like the bricked volume's address, pos = (word key << 5) | bit in the word:
//     bits 0-2 x & 7 | 3-5 y & 7 | 6-8 z & 7 | then x >> 3, y >> 3, z >> 3 in lnb bits each        (9 + 3 lnb <= 30 bits: G <= 1024)
// so the key is pos >> 5 and the bit pos & 31 — no arithmetic — and a step of +-1 along one axis is an add on that axis's scattered
// bits ("dilated integer" add): with m the axis's mask, (pos | ~m) fills the gaps with ones so that a carry runs through them,
// + (m & neg) adds the dilated +1 (the mask's lowest bit) or -1 (the whole mask), and v_bfi puts the axis back among the others.
// `neg` = the masks of the axes the ray descends along | 0x49 (the three lowest bits of the masks), `edge` = the last coordinate inside
// the grid along each axis in the ray's direction (G - 1 ascending, 0 descending), dilated: the walk leaves the grid exactly when the
// moving axis stands on its edge before the step. Boundary times, the order of the float operations and the interval filter are
// walk_cell's: the cells visited and the answers are the same, bit for bit (the coordinates are unpacked for the slab test).
struct PackGeom
{
	u32 mx, my, mz; // the axes' masks
	u32 ex, ey, ez; // dilated G - 1 per axis
	u32 lnb;
};
__device__ __forceinline__ PackGeom pack_geom(const StreamParams &S)
{
	PackGeom g;
	g.lnb = S.lnbp;
	const u32 nbm = (1u << g.lnb) - 1u, top = S.nb - 1u;
	g.mx = 7u | (nbm << 9);
	g.my = (7u << 3) | (nbm << (9u + g.lnb));
	g.mz = (7u << 6) | (nbm << (9u + 2u * g.lnb));
	g.ex = 7u | (top << 9);
	g.ey = (7u << 3) | (top << (9u + g.lnb));
	g.ez = (7u << 6) | (top << (9u + 2u * g.lnb));
	// (opaque scalars: left as fields of a struct the compiler turns `mx ? g.mx : my ? g.my : g.mz` into an indexed read of the struct —
	// a scratch load per cell)
	// (in vector registers: v_cndmask takes one scalar operand, the select of the moving axis's mask would pay a v_mov per scalar)
	asm volatile("" : "+v"(g.mx), "+v"(g.my), "+v"(g.mz));
	return g;
}
__device__ __forceinline__ u32 pack_cell(const PackGeom &g, int ix, int iy, int iz)
{
	return ((u32)ix & 7u) | (((u32)iy & 7u) << 3) | (((u32)iz & 7u) << 6) | (((u32)ix >> 3) << 9) | (((u32)iy >> 3) << (9u + g.lnb)) | (((u32)iz >> 3) << (9u + 2u * g.lnb));
}
__device__ __forceinline__ void unpack_cell(const PackGeom &g, u32 p, int &ix, int &iy, int &iz)
{
	const u32 nbm = (1u << g.lnb) - 1u;
	ix = (int)((p & 7u) | (((p >> 9) & nbm) << 3));
	iy = (int)(((p >> 3) & 7u) | (((p >> (9u + g.lnb)) & nbm) << 3));
	iz = (int)(((p >> 6) & 7u) | (((p >> (9u + 2u * g.lnb)) & nbm) << 3));
}
// walk_advance on a packed position
__device__ __forceinline__ bool walk_advance_p(const PackGeom &g, Walker &w)
{
	const float t = fminf(fminf(w.tx, w.ty), w.tz);
	const bool mx = w.tx == t, my = !mx && w.ty == t, mz = !mx && !my;
	w.t = t;
	w.tx += mx ? w.dx : 0.0f;
	w.ty += my ? w.dy : 0.0f;
	w.tz += mz ? w.dz : 0.0f;
	const u32 gmx = g.mx, gmy = g.my, gmz = g.mz;
	const u32 m = mx ? gmx : (my ? gmy : gmz);
	const bool leaving = ((w.pos ^ w.edge) & m) == 0u; // the moving axis stands on the last cell inside the grid
	const u32 q = (w.pos | ~m) + (m & w.neg);
	w.pos = (q & m) | (w.pos & ~m);
	return leaving || t >= w.tmax;
}
// walk_hit_test on a packed position
template <bool SHADOW, bool CHECK>
__device__ __forceinline__ bool walk_hit_test_p(const StreamParams &S, const PackGeom &g, const Walker &w, v3 vhalf, float k0, float k1, float eps_a, const float *ctx, int stride)
{
	const RenderParams &P = S.R;
	const float tn = fmaxf(fmaxf(__builtin_fmaf(-k1, w.dx, w.tx), __builtin_fmaf(-k1, w.dy, w.ty)), __builtin_fmaf(-k1, w.dz, w.tz));
	const float tf = fminf(fminf(__builtin_fmaf(-k0, w.dx, w.tx), __builtin_fmaf(-k0, w.dy, w.ty)), __builtin_fmaf(-k0, w.dz, w.tz));
	const float gap = tf - tn, lead = SHADOW ? tn : tf;
	const float eps = __builtin_fmaf(__builtin_fmaf(eps_a, w.t, 3.814697265625e-6f), w.t, w.eps_b);
	const bool yes = gap > eps && lead > eps, no = gap < -eps || lead < -eps;
	float e;
	if (CHECK)
	{
		int ix, iy, iz;
		unpack_cell(g, w.pos, ix, iy, iz);
		const bool truth = slab_test_at<SHADOW>(P, ix, iy, iz, vhalf, ctx, stride, e);
		if ((yes && !truth) || (no && truth)) atomicAdd(&S.ctl[2], 1u);
		return truth;
	}
	if (yes) return true;
	if (no) return false;
	int ix, iy, iz;
	unpack_cell(g, w.pos, ix, iy, iz);
	return slab_test_at<SHADOW>(P, ix, iy, iz, vhalf, ctx, stride, e);
}
// walk_batch on a packed position (no exempt cell: a shadow ray that starts in its own cell is advanced past it when it is set up)
template <bool SHADOW, bool CHECK>
__device__ __forceinline__ u32 walk_batch_p(const StreamParams &S, const PackGeom &g, Walker &w, int &term, v3 vhalf, float k0, float k1, float eps_a,
                                            const float *ctx, int stride)
{
	const float stx = w.tx, sty = w.ty, stz = w.tz, st = w.t;
	const u32 spos = w.pos;
	u32 word[kBatch], nvalid = 0;
	unsigned long long pos = 0; // 6 bits per cell — bit position, 32 for a cell past the walk's end
	bool over = false;
#pragma unroll
	for (int k = 0; k < kBatch; k++)
	{
		const u32 off = over ? 0u : ((w.pos >> 3) & ~3u);
		word[k] = *reinterpret_cast<const u32 *>(reinterpret_cast<const char *>(S.volume) + off);
		pos |= (unsigned long long)(over ? 32u : (w.pos & 31u)) << (6 * k);
		nvalid += over ? 0u : 1u;
		over = walk_advance_p(g, w) || over;
	}
	u32 live = 0;
#pragma unroll
	for (int k = 0; k < kBatch; k++)
	{
		const u32 b = (u32)(pos >> (6 * k)) & 63u;
		live |= (b < 32u ? (word[k] >> b) & 1u : 0u) << k;
	}
	u32 visited = nvalid;
	bool hit = false;
	if (live)
	{
		w.tx = stx; w.ty = sty; w.tz = stz; w.t = st; w.pos = spos;
#pragma unroll
		for (int k = 0; k < kBatch; k++)
		{
			if (!hit)
			{
				if ((live >> k) & 1u)
				{
					if (walk_hit_test_p<SHADOW, CHECK>(S, g, w, vhalf, k0, k1, eps_a, ctx, stride)) { hit = true; visited = (u32)k + 1u; }
				}
				if (!hit) walk_advance_p(g, w);
			}
		}
	}
	term = hit ? 1 : (over ? 2 : 0);
	return visited;
}

// Job source of the walk passes. The rectangle's jobs come in chunks (a pixel block x its samples, a power of two of jobs). Chunks
// are handed to WORKGROUPS — eight queues, one per XCD (workgroups go to the XCDs round-robin), a load before the atomic so that an
// empty queue costs none, the first chunk of a workgroup its own (thousands of pullers asking at once queue for tens of
// microseconds behind a word: ~88 dequeues per us) — and inside a workgroup the jobs are handed to LANES through a ticket counter in
// LDS: a wave whose idle lanes reach the refill threshold draws that many tickets; ticket t is job t % per of the workgroup's
// (t / per)-th chunk. The wave that draws the first ticket of chunk k fetches chunk k + 1's id from the global queue and posts it in a
// ring of slots, so the id is there long before chunk k runs out; a lane whose ticket lands in a chunk not yet posted waits for it.
// When the queues run dry the id posted is "none", and so is every id after it.
// Per-wave chunk ownership (the first form of this kernel) left waves with one or with two heavy chunks of 256 jobs: the launch
// lasted twice the average wave's life. Chunks of 64 jobs handed to waves through the global queues cost more in atomics than they
// balanced (1.55 ms against 0.90).
constexpr int kWalkThreads = 512, kWalkWaves = kWalkThreads / 64, kSlots = 8;

template <bool SHADOW, int LAYOUT, bool CHECK, bool BATCHED>
__global__ __launch_bounds__(kWalkThreads, kStreamWaves) void ca_stream_walk(StreamParams S)
{
	const RenderParams &P = S.R;
	if (occ_skip_enabled(P)) return; // a sparse volume: the skipping kernels of render.hip draw the frame
	__shared__ float ctx_lds[6][kWalkThreads];
	__shared__ u32 q_ticket, q_id[kSlots], q_seq[kSlots], q_read[kSlots];
	const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
	float *ctx = &ctx_lds[0][tid];
	constexpr int stride = kWalkThreads;
	const float cs = 1.0f / (float)P.G;
	const float vis = cs * P.u[U_CELLSIZE] * 0.5f;
	const v3 vhalf = V(vis, vis, vis);
	const float k0 = 0.5f - 0.5f * fabsf(P.u[U_CELLSIZE]), k1 = 1.0f - k0;
	const float eps_a = 4.76837158203125e-7f * (float)P.G;
	const u32 qown = blockIdx.x & 7u;
	u32 qcur = qown, qtried = 0;
	const u32 lper = 2u * S.lb + job_shift(P), per = 1u << lper; // jobs of a chunk
	auto q_lo = [&](u32 q) { return S.qmap ? q : (u32)(((unsigned long long)S.chunks * q) >> 3); };
	auto q_len = [&](u32 q) { return S.qmap ? (S.chunks + 7u - q) >> 3 : q_lo(q + 1u) - q_lo(q); };
	auto q_chunk = [&](u32 q, u32 pos) { return S.qmap ? pos * 8u + q : q_lo(q) + pos; };
	auto q_static = [&](u32 q) { return (gridDim.x + 7u - q) >> 3; }; // workgroups whose own chunk comes out of queue q
	auto next_chunk = [&]() -> u32 { // wave-uniform
		while (qtried < 8u)
		{
			u32 *head = S.ctl + kQueueWords + 32u * ((SHADOW ? 8u : 0u) + qcur);
			const u32 len = q_len(qcur), st = q_static(qcur);
			u32 pos = kNoChunk;
			if (lane == 0)
			{
				if (st + __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < len) pos = st + atomicAdd(head, 1u);
			}
			pos = (u32)__builtin_amdgcn_readfirstlane((int)pos);
			if (pos < len) return q_chunk(qcur, pos);
			qcur = (qcur + 1u) & 7u;
			qtried++;
		}
		return kNoChunk;
	};
	if (wave == 0)
	{
		const u32 idx = blockIdx.x >> 3;
		const u32 c0 = idx < q_len(qown) ? q_chunk(qown, idx) : next_chunk();
		if (lane == 0) { q_ticket = 0; q_id[0] = c0; q_seq[0] = 1u; }
		if (lane > 0 && lane < kSlots) q_seq[lane] = 0u;
		if (lane < kSlots) q_read[lane] = 0u;
	}
	__syncthreads();
	bool more = true;
	int job = -1, term = 0;
	bool exempt = false;
	Walker w;
	u32 visits = 0; // wave-uniform
	const unsigned long long tr_t0 = S.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
	unsigned long long tr_refill = 0;
	u32 tr_rounds = 0, tr_chunks = 0, tr_jobs = 0;
	const u32 tr_iters = 0; // (not counted: a counter in the stepping loop costs every frame an instruction per cell visit)
	for (;;)
	{
		// retire the lanes whose walk is over
		if (job >= 0 && term != 0)
		{
			if (SHADOW) S.occl[job] = term == 1 ? kOcclHit : kOcclNone;
			else
			{
				u32 out = kNoHit;
				if (term == 1)
				{
					float tn;
					slab_test<false>(P, w, vhalf, ctx, stride, tn); // the distance the plain walk reports: the slab test's own
					out = __float_as_uint(tn);
				}
				S.hit[job] = out;
			}
			job = -1;
			term = 0;
		}
		// refill
		const unsigned long long tr_r0 = S.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
		for (;;)
		{
			const unsigned long long idle = __ballot(job < 0);
			const int nidle = __popcll(idle);
			if (!more || nidle < S.refill) break;
			tr_rounds++;
			u32 base = 0;
			if (lane == 0) base = atomicAdd(&q_ticket, (u32)nidle);
			base = (u32)__builtin_amdgcn_readfirstlane((int)base);
			// the first ticket of a chunk is among these: post the next chunk's id (at most one chunk starts in a draw: nidle <= 64 <= per)
			const u32 kfirst = (base + per - 1u) >> lper;
			if ((kfirst << lper) < base + (u32)nidle)
			{
				// ... unless chunk kfirst itself is "none": the queues ran dry, and everything after the first "none" must be "none" too —
				// a wave that meets one stops drawing, so a real chunk posted behind it would never be drawn (two posters racing for the
				// last chunks of the queues did exactly that: the one for chunk k got nothing, the one for k + 1 the last chunk). Chunk
				// kfirst was asked for a whole chunk ago: it is there.
				u32 prev = 0;
				if (lane == 0)
				{
					while (__hip_atomic_load(&q_seq[kfirst % kSlots], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != kfirst + 1u) __builtin_amdgcn_s_sleep(1);
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
					prev = q_id[kfirst % kSlots];
				}
				prev = (u32)__builtin_amdgcn_readfirstlane((int)prev);
				const u32 c = prev == kNoChunk ? kNoChunk : next_chunk();
				tr_chunks++;
				if (lane == 0)
				{
					// the slot's last tenant (chunk kfirst + 1 - kSlots) must have been read by every lane that drew one of its tickets:
					// q_read counts the tickets read per slot, all of its tenants together (they were all drawn before this draw, and a
					// drawer reads at once: the wait is over before it begins, but it is what makes the ring safe)
					const u32 m = kfirst + 1u;
					while (__hip_atomic_load(&q_read[m % kSlots], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != (m / kSlots) << lper) __builtin_amdgcn_s_sleep(1);
					q_id[(kfirst + 1u) % kSlots] = c;
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
					__hip_atomic_store(&q_seq[(kfirst + 1u) % kSlots], kfirst + 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			}
			bool dry = false;
			if (job < 0)
			{
				const u32 t = base + (u32)__popcll(idle & ((1ull << lane) - 1ull));
				const u32 k = t >> lper, cand = t & (per - 1u);
				while (__hip_atomic_load(&q_seq[k % kSlots], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != k + 1u) __builtin_amdgcn_s_sleep(2);
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				const u32 chunk = q_id[k % kSlots];
				if (chunk == kNoChunk) dry = true;
				else
				{
					const u32 j = chunk * per + cand;
					u32 px, py, ks;
					job_pixel(S, chunk, cand, px, py, ks);
					if (!SHADOW)
					{
						if (px < P.W && py < P.row1)
						{
							float vu, vv;
							sample_uv(P, px, py, ks, vu, vv);
							CapturePrimary tr;
							shade_sample_with(P, vu, vv, tr);
							if (tr.wanted)
							{
								if (0.0f >= tr.len) S.hit[j] = kNoHit; // walk(): `if (t >= tmax) return false` before the first cell
								else
								{
									walker_begin(P, w, tr.enter, tr.dir, 0.0f, tr.len, ctx, stride);
									job = (int)j;
								}
							}
						}
					}
					else if (S.occl[j] == kOcclPending)
					{
						// shade_sample_with between the shading gate and the shadow walk
						const float4 r = S.rays[j];
						const v3 p = V(r.x, r.y, r.z);
						const v3 light_pos = V(P.u[U_LIGHT], P.u[U_LIGHT + 1], P.u[U_LIGHT + 2]);
						const v3 ldir = norm3(light_pos - p);
						float vn, vf;
						ray_cube(p, ldir, V(0.0f, 0.0f, 0.0f), V(kHalf, kHalf, kHalf), vn, vf);
						const v3 vexit = p + ldir * vf;
						const v3 sseg = vexit - p;
						const float slen = len3(sseg);
						if (0.0025f >= slen) S.occl[j] = kOcclNone;
						else
						{
							walker_begin(P, w, p, norm3(sseg), 0.0025f, slen, ctx, stride);
							const int cx = (int)floorf(to_cells(P, p.x)), cy = (int)floorf(to_cells(P, p.y)), cz = (int)floorf(to_cells(P, p.z));
							exempt = w.ix == cx && w.iy == cy && w.iz == cz; // any(cell != startCell) :664
							job = (int)j;
						}
					}
				}
			}
			if (__ballot(dry) != 0ull) more = false; // the queues are empty (every later chunk is posted as "none" too)
			if (lane == 0)
			{
				// tickets read, per chunk (a draw spans at most two)
				const u32 klo = base >> lper, nlo = min((u32)nidle, ((klo + 1u) << lper) - base);
				atomicAdd(&q_read[klo % kSlots], nlo);
				if ((u32)nidle > nlo) atomicAdd(&q_read[(klo + 1u) % kSlots], (u32)nidle - nlo);
			}
		}
		int walking = __popcll(__ballot(job >= 0));
		if (S.trace) { tr_refill += __builtin_amdgcn_s_memrealtime() - tr_r0; tr_jobs += (u32)walking; }
		if (walking == 0)
		{
			if (!more) break;
			continue;
		}
		// (readfirstlane: the loop's bookkeeping — counts, threshold, exit — is wave-uniform and the compiler is told so: scalar
		// registers and a scalar branch instead of an exec-masked loop with its per-iteration v_cndmask / v_cmp of uniform values)
		const int leave_at = __builtin_amdgcn_readfirstlane(more ? 64 - S.refill : 0);
		if (BATCHED)
		{
			do
			{
				u32 n = 0;
				if (job >= 0 && term == 0) n = walk_batch<SHADOW, LAYOUT, CHECK>(S, w, exempt, term, vhalf, k0, k1, eps_a, ctx, stride);
				// cells visited by the wave: n is 0 .. kBatch per lane
#pragma unroll
				for (int k = 0; k < kBatch; k++) visits += (u32)__builtin_amdgcn_readfirstlane(__popcll(__ballot(n > (u32)k)));
				walking = __builtin_amdgcn_readfirstlane(__popcll(__ballot(job >= 0 && term == 0)));
			} while (walking > leave_at);
		}
		else
		// An iteration = one read of the volume per lane that needs one, then up to kInner cells (kInner > 1: the lanes whose next cell
		// lies in what they hold go on without reading). What the loop is bound by was probed from every side (1080p, 4 samples, dense
		// scene, kernel ms): 8 -> 4 waves per SIMD 0.87 -> 0.90; ten instructions fewer per cell (scalar bookkeeping, 32-bit offsets) no
		// change; + 16 dependent VALU or + 16 SALU per cell + 8-9 %; bricks instead of rows (a quarter of the cache lines per read) - 1 %,
		// a whole 8 x 8 z-slice per lane (40 % fewer reads) - 4 %; kInner = 3 (a third fewer wave-level reads, more passes at low lane
		// occupancy) + 13 %. PMC: the waves sit in s_waitcnt for 64-72 % of their cycles and issue VALU in 10-13 %. A latency chain per
		// wave — cell -> key -> read -> test -> advance — that more waves do not hide (they queue behind the same vector L1) and that
		// fewer instructions do not shorten much: the per-cell cost is the chain. Two more forms were built on top, both bit-identical,
		// neither faster: the batched loop below (four cells per memory round trip) and ray set-up moved out of the walkers into
		// one-lane-per-job passes that leave 64-byte ray records (the walkers' refill was 40 % of their vector instructions: 926 static
		// VALU and 25 divisions per round, against ~55 per cell) — 0.93 ms against 0.87 at 1080p, 2.43 against 2.14 at 4K: the record
		// traffic cost more than the instructions it removed, the per-wave timeline (tools/stream_trace.py) stayed what it was — every
		// wave alive for the first 45 % of a launch, mean life 225 of 369 us, 11 refill rounds of ~3 us of latency each.
		do
		{
			int key = word_key<LAYOUT>(S, w.ix, w.iy, w.iz);
			visits += (u32)walking;
			if (job >= 0 && term == 0) term = walk_cell<SHADOW, LAYOUT, CHECK, true>(S, w, key, exempt, vhalf, k0, k1, eps_a, ctx, stride);
#pragma unroll
			for (int k = 1; k < kInner; k++)
			{
				key = word_key<LAYOUT>(S, w.ix, w.iy, w.iz);
				const bool go = job >= 0 && term == 0 && key == w.wkey;
				const int n = __builtin_amdgcn_readfirstlane(__popcll(__ballot(go)));
				if (n == 0) break;
				visits += (u32)n;
				if (go) term = walk_cell<SHADOW, LAYOUT, CHECK, false>(S, w, key, exempt, vhalf, k0, k1, eps_a, ctx, stride);
			}
			walking = __builtin_amdgcn_readfirstlane(__popcll(__ballot(job >= 0 && term == 0)));
		} while (walking > leave_at);
	}
	if (S.trace && lane == 0)
	{
		unsigned long long *t = S.trace + 8ull * ((SHADOW ? gridDim.x * (u32)kWalkWaves : 0u) + blockIdx.x * (u32)kWalkWaves + (u32)wave);
		t[0] = tr_t0; t[1] = __builtin_amdgcn_s_memrealtime(); t[2] = tr_rounds; t[3] = tr_iters; t[4] = tr_chunks; t[5] = tr_refill; t[6] = tr_jobs; t[7] = 0;
	}
	if (lane == 0 && visits) atomicAdd(&S.ctl[16 + 4 * ((blockIdx.x * (u32)kWalkWaves + (u32)wave) % kStatSlots) + (SHADOW ? 2 : 1)], visits);
}

// ---- second form of the walk passes (round 5; the default) -----------------------------------------------------------------------
// What ca_stream_walk's time is made of was measured this round by taking the volume out of its stepping loop (CA3D_STREAM_PROBE: no
// read / every lane the same word / a 4 KiB footprint / the real words — 2.34 / 1.91 / 2.40 / 2.39 ms for the same 967 M cell visits
// with no cell ever live): with its lanes full and eight waves on a SIMD the loop does not wait for memory at all — it is bound by
// instruction issue (~65 instructions per wave and cell, 62 % of the VALU issue slots) at 0.42 visits per ns. The real frame's
// 123 M visits would take 0.29 ms at that rate; they took 0.69. The rest is what surrounds the loop:
//   (a) a refill round set ~24 new rays up with the other 40 lanes of the wave idle (~900 instructions, 25 divisions), and three
//       of four tickets of the shadow pass were samples without a shadow ray;
//   (b) the loop ran between refills with up to 24 lanes empty (lanes active 0.46 / 0.43);
//   (c) THE TAIL: one ray in twelve of the bench scene meets nothing and crosses the whole volume, 120-200 cells where the average
//       walk ends after 9. When the queues run dry every wave is left with a handful of them and steps on for a hundred and more
//       iterations at 5 of 64 lanes, alone or nearly alone on its SIMD — where an iteration is a dependent chain of ~1 200 cycles that
//       nothing overlaps (tools/stream_trace.py: every wave alive for the first 55 % of a launch, mean life 0.68 of it).
// This form:
//   * sets rays up 64 AT A TIME — every lane, walking or not, takes a job and forms its ray (the walkers in flight stay in their
//     registers) — and the prepared walk states go into a 64-slot queue of the wave in LDS (16 words per ray); a lane whose walk ends
//     POPS a prepared state (16 LDS reads instead of the set-up), so the loop is left when 16 lanes are idle instead of 24;
//   * the shadow pass first gathers the ids of samples that HAVE a shadow ray (one 16-bit read of two flags per ticket, compacted
//     into an id list in LDS) and only then sets 64 of them up, again on every lane;
//   * tickets come straight off eight global counters (one per XCD, 64 per draw, a wave's first draw static): ticket -> chunk ->
//     pixel is arithmetic, and the balance between waves is 64 jobs fine instead of a 256-job chunk per workgroup;
//   * once a wave cannot refill any more it switches to the BATCHED stepping loop (walk_batch: four cells per memory round trip and
//     per pass through the loop's bookkeeping): in the tail the chain per cell is what costs, and this is the one place where the
//     batched loop pays (as the loop of the whole launch it measured no faster, above).
// Same arithmetic per ray and cell, same answers: the frame is ca_stream_walk's and the plain kernel's bit for bit
// (tests/test_gpu_render.py). Measured, dense bench scene, ms per frame, round 4's form -> this one (same box, same run):
// 1080p 4 spp 0.792 -> 0.729, 3840 x 2160 1.946 -> 1.847, 1080p 1 spp 0.480 -> 0.440; lanes active in the walks 0.46 / 0.43 ->
// 0.56 / 0.52 (the batched tail replays cells), vector instructions of the two walks 277 M -> 247 M. Without the batched tail 0.800: the tail is where the gain is.
// Five waves per SIMD (88 / 81 registers, no scratch; six waves spill 64 bytes and measure 0.757; four 0.741).
// Built on top and REMOVED again, both bit-identical and slower: a branch-free cell step (every lane computes filter and advance,
// selects keep what must not change: 3 branches per cell instead of 8, but 9 % more vector instructions — 0.914 against 0.824);
// handing the walkers of a wave that cannot refill to a second, compacting "drain" launch through a pool in memory (the first
// launches shrink to 0.72 of their time and lanes active rise to 0.74, but the drain launch itself — 100-160 k long rays, 64 to a
// wave, one or two waves per SIMD — takes 200-290 us at any grid size from 64 to 1 280 workgroups: long rays x a lone wave's chain per
// cell is the critical path of the tail whoever walks them; 1.07-1.24 ms per frame).
#ifndef CA3D_STREAM2_WAVES
#define CA3D_STREAM2_WAVES 5
#endif
constexpr int kW2Threads = 256, kW2Waves = kW2Threads / 64, kW2PerSimd = CA3D_STREAM2_WAVES, kIdCap = 192, kRecWords = 16;
// walker_begin without the LDS writes: the walk state in `w`, what the slab test reads in c6
__device__ __forceinline__ void walker_prepare(const RenderParams &P, Walker &w, bool &axis_parallel, float c6[6], v3 start, v3 dir, float t0, float tmax)
{
	const int G = (int)P.G;
	const float cs = 1.0f / (float)P.G;
	const v3 p = start + dir * t0;
	int ix = (int)floorf(to_cells(P, p.x)), iy = (int)floorf(to_cells(P, p.y)), iz = (int)floorf(to_cells(P, p.z));
	ix = min(max(ix, 0), G - 1);
	iy = min(max(iy, 0), G - 1);
	iz = min(max(iz, 0), G - 1);
	const int sx = dir.x > 0.0f ? 1 : -1, sy = dir.y > 0.0f ? 1 : -1, sz = dir.z > 0.0f ? 1 : -1;
	const float big = 3.0e38f;
	w.tx = dir.x != 0.0f ? (((float)(ix + (sx > 0 ? 1 : 0)) * cs - kHalf) - start.x) / dir.x : big;
	w.ty = dir.y != 0.0f ? (((float)(iy + (sy > 0 ? 1 : 0)) * cs - kHalf) - start.y) / dir.y : big;
	w.tz = dir.z != 0.0f ? (((float)(iz + (sz > 0 ? 1 : 0)) * cs - kHalf) - start.z) / dir.z : big;
	w.dx = dir.x != 0.0f ? cs / fabsf(dir.x) : big;
	w.dy = dir.y != 0.0f ? cs / fabsf(dir.y) : big;
	w.dz = dir.z != 0.0f ? cs / fabsf(dir.z) : big;
	w.t = t0;
	w.tmax = tmax;
	w.ix = ix; w.iy = iy; w.iz = iz;
	w.sx = sx; w.sy = sy; w.sz = sz;
	w.word = 0;
	w.wkey = -1;
	axis_parallel = dir.x == 0.0f || dir.y == 0.0f || dir.z == 0.0f; // (walker_begin: the filter's bound is infinite then)
	w.eps_b = 0.0f;
	c6[0] = start.x; c6[1] = start.y; c6[2] = start.z;
	c6[3] = 1.0f / dir.x; c6[4] = 1.0f / dir.y; c6[5] = 1.0f / dir.z;
}

// SKIP (round 5, late): the walk of a SPARSE volume — walk<.., true> of render_device.inc, cell for cell: clipped to the live box at set-up
// (live_box_clip), and a walker that enters a block the occupancy bits call empty (coarse 128 x 32 x 32, then fine 32 x 8 x 8; the last
// occupied block remembered) jumps to the block's exit and re-seeds its boundary times in closed form (block_jump) — which needs the
// ray's direction next to its origin and reciprocal (nine floats of context per lane instead of six, twenty words per queued ray instead
// of sixteen: its start time too, which the clip may have moved). A pass of the stepping loop is a jump or a cell for a lane, and
// counts as a visit either way, as in walk(). Cells are read cell by cell from the bricked copy; no batched tail.
template <bool SHADOW, int LAYOUT, bool CHECK, bool SKIP = false>
__global__ __launch_bounds__(kW2Threads, kW2PerSimd) void ca_stream_walk2(StreamParams S)
{
	const RenderParams &P = S.R;
	if (SKIP ? !stream_sparse_frame(S) : occ_skip_enabled(P)) return; // (a sparse volume with a small live box: render.hip's spread kernel draws the frame)
	constexpr int kCtx = SKIP ? 9 : 6, kRec = SKIP ? kRecWords + 4 : kRecWords;
	static_assert(!SKIP || (LAYOUT == kBricksRead || LAYOUT == kBricksReadAny), "the skipping walk reads bricks cell by cell");
	__shared__ float ctx_lds[kCtx][kW2Threads];
	__shared__ u32 rec_lds[kRec][kW2Threads]; // a wave's queue: rec_lds[field][64 * wave + slot]
	__shared__ u32 ids_lds[kW2Waves][kIdCap];      // a wave's list of job ids waiting for their set-up
	const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
	float *ctx = &ctx_lds[0][tid];
	constexpr int stride = kW2Threads;
	u32 *rec = &rec_lds[0][wave * 64];
	u32 *ids = ids_lds[wave];
	const float cs = 1.0f / (float)P.G;
	const float vis = cs * P.u[U_CELLSIZE] * 0.5f;
	const v3 vhalf = V(vis, vis, vis);
	const float k0 = 0.5f - 0.5f * fabsf(P.u[U_CELLSIZE]), k1 = 1.0f - k0;
	const float eps_a = 4.76837158203125e-7f * (float)P.G;
	const float eps_c = 9.5367431640625e-7f * ((float)P.G + 8.0f);
	constexpr bool PACKED = LAYOUT == kBricksPacked;
	const PackGeom pg = pack_geom(S);
	const u32 qown = blockIdx.x & 7u;
	u32 qcur = qown, qtried = 0;
	const u32 lper = 2u * S.lb + job_shift(P);   // log2 jobs of a chunk
	constexpr u32 ltj = SHADOW ? 1u : 0u;        // log2 jobs of a ticket: the shadow pass reads the flags of two jobs at once
	const u32 ltper = lper - ltj, tper = 1u << ltper; // tickets of a chunk (>= 64: the launcher has checked)
	auto q_lo = [&](u32 q) { return S.qmap ? q : (u32)(((unsigned long long)S.chunks * q) >> 3); };
	auto q_len = [&](u32 q) { return S.qmap ? (S.chunks + 7u - q) >> 3 : q_lo(q + 1u) - q_lo(q); };
	auto q_chunk = [&](u32 q, u32 pos) { return S.qmap ? pos * 8u + q : q_lo(q) + pos; };
	auto q_static = [&](u32 q) { return (gridDim.x + 7u - q) >> 3; };
	// Job source: tickets straight off eight GLOBAL counters (one per XCD: workgroups go to the XCDs round-robin, queue q owns every
	// eighth chunk), 64 per draw, one per lane; ticket t of queue q is job (t mod tickets-per-chunk) of the queue's (t / tickets-per-
	// chunk)-th chunk — arithmetic, no ring of chunk ids. ca_stream_walk hands whole chunks to workgroups and tickets to lanes through
	// LDS: one global atomic per 256 jobs, but a workgroup that draws its last chunk just before the queues run dry ends a chunk's
	// worth (12-25 us, a tenth of the launch) after one that did not, and chunks differ by a factor of three in cost: wave lives
	// spread over the last 45 % of a launch. At one atomic per 64 tickets the eight counters take 16 draws per us each (they sustain
	// ~88), and the granularity of the balance is a quarter of a chunk per WAVE. A wave's first draw is static (its index in its
	// queue): thousands of first draws at once would queue behind the counters for microseconds.
	bool first_draw = true;
	const u32 wave_in_queue = (blockIdx.x >> 3) * (u32)kW2Waves + (u32)wave; // this wave among those whose own queue is qown
	// true: no queue has a ticket left (wave-uniform); else chunk / cand are this lane's
	auto draw = [&](u32 &chunk, u32 &cand) -> bool {
		for (;;)
		{
			if (qtried >= 8u) return true;
			u32 *head = S.ctl + kQueueWords + 32u * ((SHADOW ? 8u : 0u) + qcur);
			const u32 qtickets = q_len(qcur) << ltper;                    // (a multiple of 64: tickets per chunk >= 64)
			const u32 nstatic = q_static(qcur) * (u32)kW2Waves * 64u;     // tickets handed out statically from this queue
			u32 base = 0;
			if (first_draw) base = wave_in_queue * 64u;
			else
			{
				if (lane == 0) base = nstatic + atomicAdd(head, 64u);
				base = (u32)__builtin_amdgcn_readfirstlane((int)base);
			}
			first_draw = false;
			if (base >= qtickets) { qcur = (qcur + 1u) & 7u; qtried++; continue; } // dry: the next XCD's queue
			const u32 t = base + (u32)lane;
			chunk = q_chunk(qcur, t >> ltper);
			cand = t & (tper - 1u);
			return false;
		}
	};
	bool more = true;
	int job = -1, term = 0;
	bool exempt = false;
	Walker w;
	w.tx = w.ty = w.tz = w.dx = w.dy = w.dz = w.t = w.tmax = w.eps_b = 0.0f;
	w.ix = w.iy = w.iz = 0; w.sx = w.sy = w.sz = 1; w.wkey = -1; w.word = 0;
	w.pos = 0u; w.neg = 0x49u; w.edge = 0u;
	u32 hpos = 0u;               // kBricksPacked: the cell a walk ended its hit on
	int okey = -1;               // SKIP: the block the walker was last found in an occupied block of (walk(): okey)
	u32 visits = 0;              // wave-uniform
	u32 idcount = 0;             // wave-uniform: ids waiting in ids[]
	u32 qhead = 0, qavail = 0;   // wave-uniform: prepared rays rec[.][qhead .. qhead + qavail)
	const int pop_at = S.refill2; // idle lanes at which the stepping loop is left for a pop
	const unsigned long long tr_t0 = S.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
	unsigned long long tr_refill = 0;
	u32 tr_rounds = 0, tr_jobs = 0;
	for (;;)
	{
		// retire the lanes whose walk is over (ca_stream_walk's)
		if (job >= 0 && term != 0)
		{
			if (SHADOW) S.occl[job] = term == 1 ? kOcclHit : kOcclNone;
			else
			{
				u32 out = kNoHit;
				if (term == 1)
				{
					float tn;
					int hx = w.ix, hy = w.iy, hz = w.iz;
					if (PACKED) unpack_cell(pg, hpos, hx, hy, hz);
					slab_test_at<false>(P, hx, hy, hz, vhalf, ctx, stride, tn);
					out = __float_as_uint(tn);
				}
				S.hit[job] = out;
			}
			job = -1;
			term = 0;
		}
		const unsigned long long tr_r0 = S.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
		for (;;)
		{
			const unsigned long long idle = __ballot(job < 0);
			const u32 nidle = (u32)__popcll(idle);
			if (nidle == 0u) break;
			if (qavail == 0u)
			{
				if (!more && idcount == 0u) break;
				// ---- fill, stage A: ids of jobs that want a ray, until there are 64 of them (or the queues are dry)
				while (idcount < 64u && more)
				{
					u32 chunk = 0, cand = 0;
					const bool dry = draw(chunk, cand); // wave-uniform
					if (dry) { more = false; break; }
					if (!SHADOW)
					{
						const u32 j = (chunk << lper) + cand;
						bool valid = false;
						if (!dry)
						{
							u32 px, py, ks;
							job_pixel(S, chunk, cand, px, py, ks);
							valid = px < P.W && py < P.row1;
						}
						const unsigned long long m = __ballot(valid);
						if (valid) ids[idcount + (u32)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u))] = j;
						idcount += (u32)__popcll(m);
					}
					else
					{
						const u32 j0 = (chunk << lper) + (cand << 1);
						const u32 flags = *reinterpret_cast<const unsigned short *>(S.occl + j0);
#pragma unroll
						for (u32 q = 0; q < 2u; q++)
						{
							const bool valid = ((flags >> (8u * q)) & 0xFFu) == (u32)kOcclPending;
							const unsigned long long m = __ballot(valid);
							if (valid) ids[idcount + (u32)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u))] = j0 + q;
							idcount += (u32)__popcll(m);
						}
					}
				}
				if (idcount == 0u) break; // dry, and nothing was waiting
				// ---- fill, stage B: the set-up of up to 64 of them, one per lane, every lane
				tr_rounds++;
				const u32 n = min(idcount, 64u);
				bool pushed = false, ap = false, ex = false;
				v3 sdir = V(0.0f, 0.0f, 0.0f); // SKIP: the ray's direction (block_jump divides by it)
				Walker nw;
				float c6[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
				u32 nj = 0;
				nw.tx = nw.ty = nw.tz = nw.dx = nw.dy = nw.dz = nw.t = nw.tmax = nw.eps_b = 0.0f;
				nw.ix = nw.iy = nw.iz = 0; nw.sx = nw.sy = nw.sz = 1; nw.wkey = -1; nw.word = 0;
				if ((u32)lane < n)
				{
					const u32 j = ids[lane];
					nj = j;
					if (!SHADOW)
					{
						u32 px, py, ks;
						job_pixel(S, j >> lper, j & ((1u << lper) - 1u), px, py, ks);
						float vu, vv;
						sample_uv(P, px, py, ks, vu, vv);
						CapturePrimary tr;
						shade_sample_with(P, vu, vv, tr);
						if (tr.wanted)
						{
							if (0.0f >= tr.len) S.hit[j] = kNoHit; // walk(): `if (t >= tmax) return false` before the first cell
							else
							{
								walker_prepare(P, nw, ap, c6, tr.enter, tr.dir, 0.0f, tr.len);
								pushed = true;
								if (SKIP)
								{
									sdir = tr.dir;
									// walk(): the live-box clip in front of the loop — false: the ray stays outside the box, no walk; it may move the start and the end
									if (P.live_box && !live_box_clip(P, tr.enter, tr.dir, V(c6[3], c6[4], c6[5]), 0.0f, nw.tmax, nw.ix, nw.iy, nw.iz, nw.t, nw.tx, nw.ty, nw.tz)) pushed = false;
									else if (nw.t >= nw.tmax) pushed = false; // (the loop's first `t >= tmax`)
									if (!pushed) S.hit[j] = kNoHit;
								}
							}
						}
					}
					else
					{
						// shade_sample_with between the shading gate and the shadow walk
						const float4 r = S.rays[j];
						const v3 p = V(r.x, r.y, r.z);
						const v3 light_pos = V(P.u[U_LIGHT], P.u[U_LIGHT + 1], P.u[U_LIGHT + 2]);
						const v3 ldir = norm3(light_pos - p);
						float vn, vf;
						ray_cube(p, ldir, V(0.0f, 0.0f, 0.0f), V(kHalf, kHalf, kHalf), vn, vf);
						const v3 vexit = p + ldir * vf;
						const v3 sseg = vexit - p;
						const float slen = len3(sseg);
						if (0.0025f >= slen) S.occl[j] = kOcclNone;
						else
						{
							const v3 sd = norm3(sseg);
							walker_prepare(P, nw, ap, c6, p, sd, 0.0025f, slen);
							pushed = true;
							if (SKIP)
							{
								sdir = sd;
								if (P.live_box && !live_box_clip(P, p, sd, V(c6[3], c6[4], c6[5]), 0.0025f, nw.tmax, nw.ix, nw.iy, nw.iz, nw.t, nw.tx, nw.ty, nw.tz)) pushed = false;
								else if (nw.t >= nw.tmax) pushed = false;
								if (!pushed) S.occl[j] = kOcclNone;
							}
							const int cx = (int)floorf(to_cells(P, p.x)), cy = (int)floorf(to_cells(P, p.y)), cz = (int)floorf(to_cells(P, p.z));
							ex = nw.ix == cx && nw.iy == cy && nw.iz == cz; // any(cell != startCell) :664
						}
					}
				}
				// the ids not taken move to the front of the list (at most 127 of them: two per lane; reads before writes)
				if (idcount > 64u)
				{
					const u32 left = idcount - 64u;
					u32 keep[2];
#pragma unroll
					for (u32 q = 0; q < 2u; q++) keep[q] = (u32)lane + 64u * q < left ? ids[64u + (u32)lane + 64u * q] : 0u;
#pragma unroll
					for (u32 q = 0; q < 2u; q++)
						if ((u32)lane + 64u * q < left) ids[(u32)lane + 64u * q] = keep[q];
				}
				idcount -= n;
				u32 ppos = 0u, pflags = 0u;
				if (PACKED)
				{
					ppos = pack_cell(pg, nw.ix, nw.iy, nw.iz);
					pflags = (nw.sx > 0 ? 1u : 0u) | (nw.sy > 0 ? 2u : 0u) | (nw.sz > 0 ? 4u : 0u) | (ap ? 8u : 0u);
					if (SHADOW)
					{
						// The cell a shadow ray starts in is visited and never tested (any(cell != startCell), :664): walk past it HERE, on every
						// lane at once, instead of carrying an "exempt" flag through every pass of the stepping loop.
						const bool first = pushed && ex;
						if (first)
						{
							nw.pos = ppos;
							nw.neg = ((pflags & 1u) ? 0u : pg.mx) | ((pflags & 2u) ? 0u : pg.my) | ((pflags & 4u) ? 0u : pg.mz) | 0x49u;
							nw.edge = ((pflags & 1u) ? pg.ex : 0u) | ((pflags & 2u) ? pg.ey : 0u) | ((pflags & 4u) ? pg.ez : 0u);
							if (walk_advance_p(pg, nw))
							{
								S.occl[nj] = kOcclNone; // the walk ended on its first cell
								pushed = false;
							}
							ppos = nw.pos;
						}
						visits += (u32)__popcll(__ballot(first));
					}
				}
				// prepared rays into the wave's queue, packed
				{
					const unsigned long long m = __ballot(pushed);
					if (pushed)
					{
						const u32 slot = (u32)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
						u32 *r = rec + slot;
						r[0 * kW2Threads] = __float_as_uint(nw.tx); r[1 * kW2Threads] = __float_as_uint(nw.ty); r[2 * kW2Threads] = __float_as_uint(nw.tz);
						r[3 * kW2Threads] = __float_as_uint(nw.dx); r[4 * kW2Threads] = __float_as_uint(nw.dy); r[5 * kW2Threads] = __float_as_uint(nw.dz);
						r[6 * kW2Threads] = __float_as_uint(nw.tmax);
						if (PACKED)
						{
							r[7 * kW2Threads] = ppos;
							r[8 * kW2Threads] = __float_as_uint(nw.t);
							r[9 * kW2Threads] = nj | (pflags << 28); // (job ids stay below 2^28: the launcher has checked)
						}
						else
						{
							r[7 * kW2Threads] = (u32)nw.ix | ((u32)nw.iy << 16);
							r[8 * kW2Threads] = (u32)nw.iz | (nw.sx > 0 ? 1u << 16 : 0u) | (nw.sy > 0 ? 1u << 17 : 0u) | (nw.sz > 0 ? 1u << 18 : 0u) | (ap ? 1u << 19 : 0u) | (ex ? 1u << 20 : 0u);
							r[9 * kW2Threads] = nj;
						}
#pragma unroll
						for (int q = 0; q < 6; q++) r[(10 + q) * kW2Threads] = __float_as_uint(c6[q]);
						if (SKIP)
						{
							r[16 * kW2Threads] = __float_as_uint(nw.t);
							r[17 * kW2Threads] = __float_as_uint(sdir.x); r[18 * kW2Threads] = __float_as_uint(sdir.y); r[19 * kW2Threads] = __float_as_uint(sdir.z);
						}
					}
					qavail = (u32)__popcll(m);
					qhead = 0u;
				}
				if (qavail == 0u) continue; // none of the 64 needed a walk: the next 64
			}
			// ---- pop: the idle lanes take prepared rays
			const u32 take = min(nidle, qavail);
			if (job < 0)
			{
				const u32 rk = (u32)__builtin_amdgcn_mbcnt_hi((u32)(idle >> 32), __builtin_amdgcn_mbcnt_lo((u32)idle, 0u));
				if (rk < take)
				{
					const u32 *r = rec + qhead + rk;
					w.tx = __uint_as_float(r[0 * kW2Threads]); w.ty = __uint_as_float(r[1 * kW2Threads]); w.tz = __uint_as_float(r[2 * kW2Threads]);
					w.dx = __uint_as_float(r[3 * kW2Threads]); w.dy = __uint_as_float(r[4 * kW2Threads]); w.dz = __uint_as_float(r[5 * kW2Threads]);
					w.tmax = __uint_as_float(r[6 * kW2Threads]);
					const u32 c0 = r[7 * kW2Threads], c1 = r[8 * kW2Threads], c2 = r[9 * kW2Threads];
					if (PACKED)
					{
						const bool ux = (c2 >> 28) & 1u, uy = (c2 >> 29) & 1u, uz = (c2 >> 30) & 1u; // the ray ascends along x / y / z
						w.pos = c0;
						w.neg = (ux ? 0u : pg.mx) | (uy ? 0u : pg.my) | (uz ? 0u : pg.mz) | 0x49u;
						w.edge = (ux ? pg.ex : 0u) | (uy ? pg.ey : 0u) | (uz ? pg.ez : 0u);
						w.eps_b = (c2 >> 31) ? __builtin_inff() : eps_c * (w.dx + w.dy + w.dz);
						w.t = __uint_as_float(c1);
						exempt = false;
						job = (int)(c2 & 0x0FFFFFFFu);
					}
					else
					{
						w.ix = (int)(c0 & 0xFFFFu); w.iy = (int)(c0 >> 16); w.iz = (int)(c1 & 0xFFFFu);
						w.sx = (c1 >> 16) & 1u ? 1 : -1; w.sy = (c1 >> 17) & 1u ? 1 : -1; w.sz = (c1 >> 18) & 1u ? 1 : -1;
						w.eps_b = (c1 >> 19) & 1u ? __builtin_inff() : eps_c * (w.dx + w.dy + w.dz);
						exempt = (c1 >> 20) & 1u;
						w.t = SHADOW ? 0.0025f : 0.0f;
						job = (int)c2;
					}
					w.word = 0;
					w.wkey = -1;
					term = 0;
#pragma unroll
					for (int q = 0; q < 6; q++) ctx[q * stride] = __uint_as_float(r[(10 + q) * kW2Threads]);
					if (SKIP)
					{
						w.t = __uint_as_float(r[16 * kW2Threads]);
#pragma unroll
						for (int q = 0; q < 3; q++) ctx[(6 + q) * stride] = __uint_as_float(r[(17 + q) * kW2Threads]);
						okey = -1;
					}
				}
			}
			qhead += take;
			qavail -= take;
			if (qavail != 0u || take == nidle) break; // every idle lane has a ray, or rays are left over: walk
		}
		int walking = __popcll(__ballot(job >= 0));
		if (S.trace) { tr_refill += __builtin_amdgcn_s_memrealtime() - tr_r0; tr_jobs += (u32)walking; }
		if (walking == 0)
		{
			if (!more && idcount == 0u && qavail == 0u) break;
			continue;
		}
		const bool refillable = more || idcount != 0u || qavail != 0u;
		const int leave_at = __builtin_amdgcn_readfirstlane(refillable ? 64 - pop_at : 0);
		if (!SKIP && (S.tail_batch == 2 || (S.tail_batch && !refillable)))
		{
			// nothing left to refill from: few waves are left on the chip, their rays are the long ones and lie
			// all over the volume — an iteration is one memory round trip that nothing hides. Four cells per round trip (walk_batch).
			do
			{
				u32 n = 0;
				if (job >= 0 && term == 0)
				{
					if (PACKED)
					{
						n = walk_batch_p<SHADOW, CHECK>(S, pg, w, term, vhalf, k0, k1, eps_a, ctx, stride);
						if (term == 1) hpos = w.pos;
					}
					else n = walk_batch<SHADOW, LAYOUT, CHECK>(S, w, exempt, term, vhalf, k0, k1, eps_a, ctx, stride);
				}
#pragma unroll
				for (int k = 0; k < kBatch; k++) visits += (u32)__builtin_amdgcn_readfirstlane(__popcll(__ballot(n > (u32)k)));
				walking = __builtin_amdgcn_readfirstlane(__popcll(__ballot(job >= 0 && term == 0)));
			} while (walking > leave_at);
		}
		else if (PACKED)
		{
			// (one state word per lane — 0 walking, 1 hit, 2 over, 3 no ray — so that "who still walks" is ONE compare whose mask is both the
			// loop body's exec mask and the count the loop ends on)
			// A lane that finds its hit advances like the others (its walk state is not needed again; the hit cell is kept in hpos): the
			// body has ONE nested exec region, the live cells' filter, instead of a second one around the advance.
			// The NEXT cell's word is asked for before this cell is looked at: which cell comes next depends on the boundary times only, so
			// the advance is split — axis, next position, read issued; then this cell's bit and, rarely, its filter (on the boundary times
			// of THIS cell, not yet touched); then the times are moved on. A wave's pass through the loop was read -> ~800 cycles -> 45
			// instructions (per-wave trace: 857 cycles a pass, five waves to a SIMD: 171 cycles per pass and SIMD against ~105 of pure
			// issue); now the read of cell k + 1 is in flight while cell k is tested and the other four waves issue.
			int st = (job >= 0 && term == 0) ? 0 : 3;
			const char *vol = reinterpret_cast<const char *>(S.volume);
			u32 cur = 0u;
			if (st == 0) cur = *reinterpret_cast<const u32 *>(vol + ((w.pos >> 3) & ~3u));
			do
			{
				visits += (u32)walking;
				if (st == 0)
				{
					const float tmin = fminf(fminf(w.tx, w.ty), w.tz);
					const bool mx = w.tx == tmin, my = !mx && w.ty == tmin, mz = !mx && !my;
					const u32 gmx = pg.mx, gmy = pg.my, gmz = pg.mz;
					const u32 m = mx ? gmx : (my ? gmy : gmz);
					const bool leaving = ((w.pos ^ w.edge) & m) == 0u; // the moving axis stands on the last cell inside the grid
					const u32 q = (w.pos | ~m) + (m & w.neg);
					const u32 npos = (q & m) | (w.pos & ~m); // (a lane that leaves wraps to the far side of the grid: a valid address, never used)
					const u32 nxt = *reinterpret_cast<const u32 *>(vol + ((npos >> 3) & ~3u));
					bool hit = false;
					if (__builtin_amdgcn_ubfe(cur, w.pos, 1u) != 0u) // (the shift takes bits 0-4 of pos: the bit in the word)
					{
						if (walk_hit_test_p<SHADOW, CHECK>(S, pg, w, vhalf, k0, k1, eps_a, ctx, stride)) { hit = true; hpos = w.pos; }
					}
					w.t = tmin;
					w.tx += mx ? w.dx : 0.0f;
					w.ty += my ? w.dy : 0.0f;
					w.tz += mz ? w.dz : 0.0f;
					w.pos = npos;
					cur = nxt;
					st = hit ? 1 : ((leaving || tmin >= w.tmax) ? 2 : 0);
				}
				walking = __builtin_popcountll(__builtin_amdgcn_ballot_w64(st == 0));
			} while (walking > leave_at);
			if (st != 3) term = st;
		}
		else if (SKIP)
		{
			do
			{
				visits += (u32)walking;
				if (job >= 0 && term == 0)
				{
					// walk(): the block of the cell — coarse bits first, then fine — unless it is the block last found occupied. The three reads a
					// visit may need (coarse word, fine word, the cell's own word: every address follows from the cell alone) are issued TOGETHER:
					// one memory round trip per visit instead of up to three dependent ones — a sparse volume's walks are few and long, and a
					// lone wave's chain of round trips is what the frame waits for (82 k shadow rays of 122 visits each took 0.65 ms).
					int empty = 0;
					const int bk = (w.ix >> 5) + ((w.iy >> 3) + (w.iz >> 3) * ((int)P.G >> 3)) * (int)P.cols;
					const int key = word_key<LAYOUT>(S, w.ix, w.iy, w.iz);
					const u32 cur = *reinterpret_cast<const u32 *>(reinterpret_cast<const char *>(S.volume) + ((u32)key << 2));
					if (bk != okey)
					{
						const int ck = (w.ix >> 7) + ((w.iy >> 5) + (w.iz >> 5) * ((int)P.G >> 5)) * ((int)P.cols >> 2);
						const unsigned long long fw = P.occ[bk >> 6];
						const unsigned long long cw = P.occ_coarse ? P.occ[P.occ_words + 1u + (u32)(ck >> 6)] : ~0ull;
						if (!((cw >> (ck & 63)) & 1ull)) empty = 1;       // (coarse_occupied)
						else if (!((fw >> (bk & 63)) & 1ull)) empty = 2;  // (block_occupied)
						else okey = bk;
					}
					if (empty)
					{
						const v3 start = V(ctx[0 * stride], ctx[1 * stride], ctx[2 * stride]), inv = V(ctx[3 * stride], ctx[4 * stride], ctx[5 * stride]), dir = V(ctx[6 * stride], ctx[7 * stride], ctx[8 * stride]);
						const bool on = empty == 1 ? block_jump<7, 5, 5>(P, start, dir, inv, w.tmax, w.ix, w.iy, w.iz, w.t, w.tx, w.ty, w.tz)
						                           : block_jump<5, 3, 3>(P, start, dir, inv, w.tmax, w.ix, w.iy, w.iz, w.t, w.tx, w.ty, w.tz);
						if (!on) term = 2;
						exempt = false; // (the walk has left the cell it started in)
					}
					else term = walk_cell_word<SHADOW, LAYOUT, CHECK>(S, w, cur, exempt, vhalf, k0, k1, eps_a, ctx, stride);
				}
				walking = __builtin_amdgcn_readfirstlane(__popcll(__ballot(job >= 0 && term == 0)));
			} while (walking > leave_at);
		}
		else
		{
			do
			{
				const int key = word_key<LAYOUT>(S, w.ix, w.iy, w.iz);
				visits += (u32)walking;
				if (job >= 0 && term == 0) term = walk_cell<SHADOW, LAYOUT, CHECK, true>(S, w, key, exempt, vhalf, k0, k1, eps_a, ctx, stride);
				walking = __builtin_amdgcn_readfirstlane(__popcll(__ballot(job >= 0 && term == 0)));
			} while (walking > leave_at);
		}
	}
	if (S.trace && lane == 0)
	{
		unsigned long long *t = S.trace + 8ull * ((SHADOW ? gridDim.x * (u32)kW2Waves : 0u) + blockIdx.x * (u32)kW2Waves + (u32)wave);
		t[0] = tr_t0; t[1] = __builtin_amdgcn_s_memrealtime(); t[2] = tr_rounds; t[3] = 0; t[4] = 0; t[5] = tr_refill; t[6] = tr_jobs; t[7] = 0;
	}
	if (lane == 0 && visits) atomicAdd(&S.ctl[16 + 4 * ((blockIdx.x * (u32)kW2Waves + (u32)wave) % kStatSlots) + (SHADOW ? 2 : 1)], visits);
}

// Second pass: every job of the rectangle, one lane each.
__global__ __launch_bounds__(256) void ca_stream_shadow_rays(StreamParams S)
{
	const RenderParams &P = S.R;
	if (!stream_owns_frame(S)) return;
	const u32 j = blockIdx.x * 256u + threadIdx.x;
	const u32 lj = 2u * S.lb + job_shift(P); // log2 jobs per chunk
	const u32 c = j >> lj;
	bool lit = false;
	if (c < S.chunks)
	{
		u32 px, py, k;
		job_pixel(S, c, j & ((1u << lj) - 1u), px, py, k);
		unsigned char flag = kOcclNone;
		if (px < P.W && py < P.row1)
		{
			float vu, vv;
			sample_uv(P, px, py, k, vu, vv);
			LookupPrimaryRecordShadow tr;
			tr.answer = S.hit[j]; // (only read by primary(): a view ray that misses the volume never asks)
			shade_sample_with(P, vu, vv, tr);
			if (S.check && tr.unanswered) { atomicAdd(&S.ctl[3], 1u); atomicMax(&S.ctl[4], ~j); atomicMax(&S.ctl[5], j); }
			if (tr.lit)
			{
				S.rays[j] = make_float4(tr.p.x, tr.p.y, tr.p.z, 0.0f);
				flag = kOcclPending;
				lit = true;
			}
		}
		S.occl[j] = flag;
	}
	const u32 n = (u32)__popcll(__ballot(lit));
	if ((threadIdx.x & 63u) == 0u && n) atomicAdd(&S.ctl[16 + 4 * (blockIdx.x % kStatSlots) + 0], n);
}

// Last pass: one lane per pixel of the rectangle, chunk by chunk; the plain kernel's sums and outputs.
__global__ __launch_bounds__(256) void ca_stream_resolve(StreamParams S)
{
	const RenderParams &P = S.R;
	if (!stream_owns_frame(S)) return;
	if (blockIdx.x == 0 && threadIdx.x < 3u && P.counters)
	{
		// the frame's statistics: the slots the passes before this one added to (stream order: they are complete)
		unsigned long long sum = 0;
		for (int s = 0; s < kStatSlots; s++) sum += S.ctl[16 + 4 * s + threadIdx.x];
		if (sum) atomicAdd(&P.counters[threadIdx.x], sum);
	}
	const u32 sh = job_shift(P);
	const u32 n = blockIdx.x * 256u + threadIdx.x; // pixel slot: its first job is n << sh
	const u32 lj = 2u * S.lb + sh;
	const u32 c = (n << sh) >> lj;
	if (c >= S.chunks) return;
	u32 px, py, k0;
	job_pixel(S, c, (n << sh) & ((1u << lj) - 1u), px, py, k0);
	if (px >= P.W || py >= P.row1) return;
	float r = 0.0f, g = 0.0f, b = 0.0f, a = 0.0f, d0 = 0.0f;
	for (u32 k = 0; k < P.spp; k++)
	{
		const u32 j = (n << sh) + k;
		float vu, vv;
		sample_uv(P, px, py, k, vu, vv);
		LookupBoth tr;
		tr.answer = S.hit[j];
		tr.occl = S.occl + j;
		const Sample s = shade_sample_with(P, vu, vv, tr);
		if (S.check && tr.unanswered) atomicAdd(&S.ctl[3], 1u);
		r += s.r; g += s.g; b += s.b; a += s.a;
		if (k == 0) d0 = s.depth;
	}
	const float inv = 1.0f / (float)P.spp;
	r *= inv; g *= inv; b *= inv; a *= inv;
	const size_t i = (size_t)py * P.W + px;
	if (P.light)
	{
		const __half2 rg = __floats2half2_rn(r, g), ba = __floats2half2_rn(b, 1.0f);
		uint2 v;
		v.x = *reinterpret_cast<const u32 *>(&rg);
		v.y = *reinterpret_cast<const u32 *>(&ba);
		P.light[i] = v;
	}
	if (P.depth)
	{
		const __half2 d = __floats2half2_rn(d0, 1.0f);
		P.depth[i] = *reinterpret_cast<const u32 *>(&d);
	}
	if (P.presentation)
	{
		const float ig = 1.0f / P.u[U_GAMMA];
		P.presentation[i] = unorm8(powf(r, ig)) | (unorm8(powf(g, ig)) << 8) | (unorm8(powf(b, ig)) << 16) | (unorm8(a) << 24);
	}
}

// (S.skip_ok: the walks of a scattered sparse volume too — each of the two kernels of a pass returns at once on the frames of the other)
template <int P2, bool CHECK>
void launch_walks2(const StreamParams &S, u32 wgs2, u32 job_blocks, hipStream_t stream)
{
	constexpr int PS = kBricksReadAny; // the skipping walk reads bricks cell by cell (not the packed positions); the brick index by the padded bit count — every grid's
	hipLaunchKernelGGL((ca_stream_walk2<false, P2, CHECK>), dim3(wgs2), dim3(kW2Threads), 0, stream, S);
	if (S.skip_ok) hipLaunchKernelGGL((ca_stream_walk2<false, PS, CHECK, true>), dim3(wgs2), dim3(kW2Threads), 0, stream, S);
	hipLaunchKernelGGL(ca_stream_shadow_rays, dim3(job_blocks), dim3(256), 0, stream, S);
	hipLaunchKernelGGL((ca_stream_walk2<true, P2, CHECK>), dim3(wgs2), dim3(kW2Threads), 0, stream, S);
	if (S.skip_ok) hipLaunchKernelGGL((ca_stream_walk2<true, PS, CHECK, true>), dim3(wgs2), dim3(kW2Threads), 0, stream, S);
}

template <int P2, bool CHECK, bool BATCHED = false>
void launch_walks(const StreamParams &S, u32 wgs, u32 job_blocks, hipStream_t stream)
{
	hipLaunchKernelGGL((ca_stream_walk<false, P2, CHECK, BATCHED>), dim3(wgs), dim3(kWalkThreads), 0, stream, S);
	hipLaunchKernelGGL(ca_stream_shadow_rays, dim3(job_blocks), dim3(256), 0, stream, S);
	hipLaunchKernelGGL((ca_stream_walk<true, P2, CHECK, BATCHED>), dim3(wgs), dim3(kWalkThreads), 0, stream, S);
}

} // namespace

size_t stream_scratch_bytes(uint32_t W, uint32_t H, uint32_t spp, size_t *hit_off, size_t *occl_off, size_t *rays_off)
{
	// the rectangle is aligned to 32 x 16 pixels and clipped to the frame: at most the padded frame
	const size_t jobs = (size_t)((W + 31u) / 32u * 32u) * ((H + 15u) / 16u * 16u) * spp;
	size_t off = 4096; // control words + statistics slots
	*hit_off = off; off += jobs * 4u;
	*occl_off = off; off += (jobs + 255u) / 256u * 256u;
	*rays_off = off; off += jobs * 16u;
	return off;
}

// The dense-volume part of a frame over the rectangle P.rx0 .. P.ry1 (render.hip's volume_rect), on `stream`. The caller has
// cleared P.counters and, when there is one, run the occupancy pass (the passes here test its count on the device).
hipError_t launch_render_stream(const void *params, void *scratch, uint32_t W, uint32_t H, bool check, uint32_t *bricks, bool bricks_valid, hipStream_t stream, bool *bricks_built, hipEvent_t before_resolve, int walk_share_pct, bool *sparse_too)
{
	StreamParams S;
	S.R = *static_cast<const RenderParams *>(params);
	S.skip_ok = 0u;
	if (sparse_too) *sparse_too = false;
	const RenderParams &P = S.R;
	size_t hit_off, occl_off, rays_off;
	stream_scratch_bytes(W, H, P.spp, &hit_off, &occl_off, &rays_off);
	char *base = static_cast<char *>(scratch);
	S.ctl = reinterpret_cast<u32 *>(base);
	S.hit = reinterpret_cast<u32 *>(base + hit_off);
	S.occl = reinterpret_cast<unsigned char *>(base + occl_off);
	S.rays = reinterpret_cast<float4 *>(base + rays_off);
	static const int lb_env = getenv("CA3D_STREAM_LB") ? atoi(getenv("CA3D_STREAM_LB")) : 0;
	static const int qmap_env = getenv("CA3D_STREAM_QMAP") ? atoi(getenv("CA3D_STREAM_QMAP")) : 0;
	S.lb = lb_env >= 1 && lb_env <= 4 ? (u32)lb_env : (P.spp == 4u ? 3u : 4u); // pixel block of a chunk (the rectangle is aligned to 32 x 16)
	S.qmap = qmap_env == 1 || !getenv("CA3D_STREAM_QMAP") ? 1u : 0u;
	const u32 edge = 1u << S.lb;
	S.chunks_x = (P.rx1 - P.rx0) / edge;
	S.chunks = S.chunks_x * ((P.ry1 - P.ry0) / edge);
	const u32 per = P.spp << (2u * S.lb);
	const bool p2 = (P.G & (P.G - 1u)) == 0u;
	S.lg = S.lc = 0;
	S.nb = P.G >> 3;
	S.lnbp = 0;
	while ((1u << S.lnbp) < S.nb) S.lnbp++;
	if (p2)
	{
		while ((1u << S.lg) < P.G) S.lg++;
		S.lc = S.lg - 5u;
	}
	static const int refill_env = getenv("CA3D_STREAM_REFILL") ? atoi(getenv("CA3D_STREAM_REFILL")) : 0;
	S.refill = refill_env >= 1 && refill_env <= 64 ? refill_env : 24;
	static const char *trace_path = getenv("CA3D_STREAM_TRACE");
	static unsigned long long *trace_buf = nullptr;
	const size_t trace_words = 8u * 2u * 4u * 256u * 8u * 2u; // two walk passes, up to 16 384 waves each
	if (trace_path && !trace_buf && hipMalloc((void **)&trace_buf, trace_words * 8u) != hipSuccess) trace_buf = nullptr;
	S.trace = trace_path ? trace_buf : nullptr;
	if (S.trace) hipMemsetAsync(S.trace, 0, trace_words * 8u, stream);
	S.check = check ? 1u : 0u;
	if (check)
	{
		// every answer a pass looks up must have been given in THIS frame: a job the queues lost would otherwise show last frame's
		hipMemsetAsync(S.hit, 0x5A, (size_t)S.chunks * per * 4u, stream);
		hipMemsetAsync(S.occl, 0x5A, (size_t)S.chunks * per, stream);
	}
	hipError_t e = hipMemsetAsync(S.ctl, 0, 4096, stream);
	if (e != hipSuccess) return e;
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	const u32 wgs = min(S.chunks, (u32)cus * (u32)kStreamWaves * 4u / (u32)kWalkWaves); // persistent: what the chip holds at kStreamWaves per SIMD
	const u32 job_blocks = (S.chunks * per + 255u) / 256u;
	static const int bricks_env = getenv("CA3D_STREAM_BRICKS") ? atoi(getenv("CA3D_STREAM_BRICKS")) : 1; // tuning: 0 rows, 1 default, 2 bricks with a cached word, 3 batched loop
	unsigned long long trace_waves = (unsigned long long)wgs * (unsigned)kWalkWaves;
	const bool bricked = bricks && bricks_env && frame_bricks_applies(P.G);
	S.volume = bricked ? bricks : P.cells;
	if (bricked)
	{
		if (!bricks_valid) // (the engine says the bricks are those of this very state: a frame of the same state was drawn before)
		{
			hipError_t eb = launch_brick_volume(P.cells, bricks, P.G, stream); // one pass over the state (~10 us at 512^3, 0.66 ms at 2048^3)
			if (eb != hipSuccess) return eb;
			if (bricks_built) *bricks_built = true;
		}
		// which form of the walk passes: 2 (default) sets rays up 64 at a time and queues them in LDS (ca_stream_walk2); 1: round 4's form
		static const int form_env = getenv("CA3D_STREAM_FORM") ? atoi(getenv("CA3D_STREAM_FORM")) : 2;
		static const int pop_env = getenv("CA3D_STREAM_POP") ? atoi(getenv("CA3D_STREAM_POP")) : 0;
		S.refill2 = pop_env >= 1 && pop_env <= 64 ? pop_env : 16;
		static const int tb_env = getenv("CA3D_STREAM_TAIL_BATCH") ? atoi(getenv("CA3D_STREAM_TAIL_BATCH")) : 1;
		S.tail_batch = tb_env == 2 ? 2 : (tb_env ? 1 : 0); // (2, tuning: the batched loop from the first cell on)
		// A persistent walk launch asks for the wave slots of the whole chip — unless other frames are in flight beside this one (ca3d_api.cpp,
		// FrameLane): then for its share of them, so that the frames' walks run side by side from their first workgroup on instead of one
		// launch filling the chip and the next one seeping into its tail (tools/sweep_stream_wgs.sh; CA3D_STREAM_WGS_PCT overrides: tuning).
		static const int wgs_env = getenv("CA3D_STREAM_WGS_PCT") ? atoi(getenv("CA3D_STREAM_WGS_PCT")) : 0;
		const u32 pct = (u32)(wgs_env >= 10 && wgs_env <= 100 ? wgs_env : (walk_share_pct >= 10 && walk_share_pct <= 100 ? walk_share_pct : 100));
		const u32 wgs2 = max(8u, min(S.chunks, (u32)cus * (u32)kW2PerSimd * 4u / (u32)kW2Waves * pct / 100u));
		const bool form2 = form_env == 2 && per >= 128u && bricks_env == 1; // (a chunk holds at least 64 tickets of two jobs)
		// scattered sparse volumes through the stream passes as well (ca_stream_walk2<.., SKIP>; CA3D_STREAM_SKIP=0: render.hip's scheduled kernel, tuning / A-B)
		static const int skip_env = getenv("CA3D_STREAM_SKIP") ? atoi(getenv("CA3D_STREAM_SKIP")) : 1;
		S.skip_ok = form2 && skip_env && P.occ && !getenv("CA3D_STREAM_PROBE") ? 1u : 0u;
		if (sparse_too) *sparse_too = S.skip_ok != 0u;
		if (form2) trace_waves = (unsigned long long)wgs2 * (unsigned)kW2Waves;
		static const int probe_env = getenv("CA3D_STREAM_PROBE") ? atoi(getenv("CA3D_STREAM_PROBE")) : 0;
		static const int packed_env = getenv("CA3D_STREAM_PACKED") ? atoi(getenv("CA3D_STREAM_PACKED")) : 1; // tuning: 0 = cell coordinates as three integers (kBricksRead)
		S.probe_mask = 0u;
		if (probe_env >= 1 && probe_env <= 4 && p2)
		{
			if (probe_env == 1) launch_walks<kProbeNone, false>(S, wgs, job_blocks, stream);
			else if (probe_env == 2) launch_walks<kProbeSame, false>(S, wgs, job_blocks, stream);
			else if (probe_env == 3) launch_walks<kProbeSmall, false>(S, wgs, job_blocks, stream);
			else launch_walks<kProbeFull, false>(S, wgs, job_blocks, stream);
		}
		else if (form2 && packed_env && P.G <= 1024u && (unsigned long long)S.chunks * per < (1ull << 28)) { if (check) launch_walks2<kBricksPacked, true>(S, wgs2, job_blocks, stream); else launch_walks2<kBricksPacked, false>(S, wgs2, job_blocks, stream); }
		else if (form2 && !p2) { if (check) launch_walks2<kBricksReadAny, true>(S, wgs2, job_blocks, stream); else launch_walks2<kBricksReadAny, false>(S, wgs2, job_blocks, stream); }
		else if (form2) { if (check) launch_walks2<kBricksRead, true>(S, wgs2, job_blocks, stream); else launch_walks2<kBricksRead, false>(S, wgs2, job_blocks, stream); }
		else if (!p2) { if (check) launch_walks<kBricksReadAny, true>(S, wgs, job_blocks, stream); else launch_walks<kBricksReadAny, false>(S, wgs, job_blocks, stream); }
		else if (bricks_env == 3) { if (check) launch_walks<kBricks, true, true>(S, wgs, job_blocks, stream); else launch_walks<kBricks, false, true>(S, wgs, job_blocks, stream); }
		else if (bricks_env == 2) { if (check) launch_walks<kBricks, true>(S, wgs, job_blocks, stream); else launch_walks<kBricks, false>(S, wgs, job_blocks, stream); }
		else if (check) launch_walks<kBricksRead, true>(S, wgs, job_blocks, stream);
		else launch_walks<kBricksRead, false>(S, wgs, job_blocks, stream);
	}
	else if (p2) { if (check) launch_walks<kRowsP2, true>(S, wgs, job_blocks, stream); else launch_walks<kRowsP2, false>(S, wgs, job_blocks, stream); }
	else { if (check) launch_walks<kRowsAny, true>(S, wgs, job_blocks, stream); else launch_walks<kRowsAny, false>(S, wgs, job_blocks, stream); }
	if (before_resolve) // (two frames in flight: the frame before this one must have written the shared presentation surface first)
	{
		hipError_t ew = hipStreamWaitEvent(stream, before_resolve, 0);
		if (ew != hipSuccess) return ew;
	}
	hipLaunchKernelGGL(ca_stream_resolve, dim3((S.chunks * (per / P.spp) + 255u) / 256u), dim3(256), 0, stream, S);
	if (S.trace)
	{
		// the last frame wins; [0] = waves per pass
		std::vector<unsigned long long> t(trace_words);
		if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(t.data(), S.trace, trace_words * 8u, hipMemcpyDeviceToHost) == hipSuccess)
			if (FILE *f = fopen(trace_path, "wb"))
			{
				const unsigned long long n = trace_waves;
				fwrite(&n, 8, 1, f);
				fwrite(t.data(), 8, (size_t)n * 16u, f);
				fclose(f);
			}
	}
	return hipGetLastError();
}

} // namespace ca3d
