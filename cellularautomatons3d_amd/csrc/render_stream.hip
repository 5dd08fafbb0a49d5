// The converged frame of a dense packed volume as a RAY STREAM (gfx950): the same frame as render.hip's plain kernel, bit for
// bit, drawn by four lean passes instead of one kernel that carries a sample from its view ray to its colour.
//
//   ca_stream_walk<primary>   persistent; lanes take sample jobs of the volume's screen rectangle from a queue of 8 x 8-pixel tiles,
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

#include "ca3d_internal.h"

namespace ca3d
{
namespace
{

#include "render_device.inc"

constexpr u32 kNoHit = 0xFFFFFFFFu; // a hit distance is never NaN (the slab test rejects NaN)
enum : unsigned char { kOcclNone = 0, kOcclHit = 1, kOcclPending = 2 };

#ifndef CA3D_STREAM_WAVES
#define CA3D_STREAM_WAVES 8
#endif
#ifndef CA3D_STREAM_REFILL
#define CA3D_STREAM_REFILL 24
#endif
constexpr int kStreamWaves = CA3D_STREAM_WAVES; // waves per SIMD the walk kernel is compiled for
constexpr int kRefillAt = CA3D_STREAM_REFILL;   // lanes without a ray at which a wave leaves the stepping loop to take new jobs
constexpr int kStatSlots = 64;

struct StreamParams
{
	RenderParams R;
	u32 *hit;            // per job: the view ray's answer (float bits of the hit cube's slab entry, kNoHit)
	unsigned char *occl; // per job: kOcclPending (lit, shadow ray waiting) -> kOcclHit / kOcclNone
	float4 *rays;        // per lit job: the point the shadow ray starts from
	u32 *ctl;            // [0] primary tile queue, [1] shadow tile queue, [2] filter / slab-test contradictions (check build),
	                     // [16 + 4 * slot + {0 shadow rays, 1 primary visits, 2 shadow visits}] statistics, kStatSlots slots
	u32 tiles, tiles_x;  // 8 x 8-pixel tiles of the rectangle; jobs of tile i: [i * 64 * spp, (i + 1) * 64 * spp), pixel-major
	u32 lg, lc;          // log2 G, log2 cols (power-of-two grids)
};

__device__ __forceinline__ u32 job_shift(const RenderParams &P) { return P.spp == 4u ? 2u : 0u; }

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
	bool lit = false;
	v3 p;
	__device__ __forceinline__ bool primary(const RenderParams &, v3, v3, v3, v3, float, v3, float &tnear)
	{
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
	__device__ __forceinline__ bool primary(const RenderParams &, v3, v3, v3, v3, float, v3, float &tnear)
	{
		if (answer == kNoHit) return false;
		tnear = __uint_as_float(answer);
		return true;
	}
	__device__ __forceinline__ bool shadow(const RenderParams &, v3, v3, float, v3, int, int, int) { return *occl == kOcclHit; }
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
};

template <bool P2>
__device__ __forceinline__ int word_key(const StreamParams &S, int ix, int iy, int iz)
{
	if (P2) return (int)(((((u32)iz << S.lg) + (u32)iy) << S.lc) + ((u32)ix >> 5));
	return (ix >> 5) + (iy + iz * (int)S.R.G) * (int)S.R.cols;
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
__device__ __forceinline__ bool slab_test(const RenderParams &P, const Walker &w, v3 vhalf, const float *ctx, int stride, float &tn)
{
	const v3 start = V(ctx[0 * stride], ctx[1 * stride], ctx[2 * stride]);
	const v3 inv = V(ctx[3 * stride], ctx[4 * stride], ctx[5 * stride]);
	float tf;
	ray_cube_inv(start, inv, cell_origin(1.0f / (float)P.G, w.ix, w.iy, w.iz), vhalf, tn, tf);
	return SHADOW ? (tn <= tf && tn >= 0.0f) /* :668 */ : (tf >= 0.0f && tn <= tf) /* :722-729 */;
}

// One cell of the walk. Returns 0 keep walking, 1 hit, 2 the ray left the volume or ran out of range.
//   k0, k1: the cube's slab offsets in units of the cell's crossing time (walker_begin); eps_a = 2^-21 G
template <bool SHADOW, bool P2, bool CHECK>
__device__ __forceinline__ int walk_cell(const StreamParams &S, Walker &w, bool &exempt, v3 vhalf, float k0, float k1, float eps_a,
                                         const float *ctx, int stride)
{
	const RenderParams &P = S.R;
	const int key = word_key<P2>(S, w.ix, w.iy, w.iz);
	if (key != w.wkey) { w.word = P.cells[key]; w.wkey = key; }
	if (((w.word >> (w.ix & 31)) & 1u) && !exempt)
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
	const bool outside = (u32)(w.ix | w.iy | w.iz) >= P.G; // a negative coordinate sets the top bit
	return (outside || t >= w.tmax) ? 2 : 0;
}

template <bool SHADOW, bool P2, bool CHECK>
__global__ __launch_bounds__(256, kStreamWaves) void ca_stream_walk(StreamParams S)
{
	const RenderParams &P = S.R;
	if (occ_skip_enabled(P)) return; // a sparse volume: the skipping kernels of render.hip draw the frame
	__shared__ float ctx_lds[6][256];
	const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
	float *ctx = &ctx_lds[0][tid];
	constexpr int stride = 256;
	const float cs = 1.0f / (float)P.G;
	const float vis = cs * P.u[U_CELLSIZE] * 0.5f;
	const v3 vhalf = V(vis, vis, vis);
	const float k0 = 0.5f - 0.5f * fabsf(P.u[U_CELLSIZE]), k1 = 1.0f - k0;
	const float eps_a = 4.76837158203125e-7f * (float)P.G;
	const u32 sh = job_shift(P), per = 64u << sh; // jobs of a tile
	const u32 nwaves = gridDim.x * 4u;
	u32 tile = blockIdx.x * 4u + (u32)wave; // the first tile is the wave's own; the queue hands out the rest
	bool more = tile < S.tiles;
	u32 jn = 0;                             // next job of the tile nobody has taken
	int job = -1, term = 0;
	bool exempt = false;
	Walker w;
	u32 visits = 0; // wave-uniform
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
		for (;;)
		{
			const unsigned long long idle = __ballot(job < 0);
			const int nidle = __popcll(idle);
			if (!more || nidle < kRefillAt) break;
			const u32 cand = jn + (u32)__popcll(idle & ((1ull << lane) - 1ull));
			if (job < 0 && cand < per)
			{
				const u32 j = tile * per + cand;
				const u32 q = cand >> sh, k = cand & ((1u << sh) - 1u);
				const u32 px = P.rx0 + (tile % S.tiles_x) * 8u + (q & 7u), py = P.ry0 + (tile / S.tiles_x) * 8u + (q >> 3);
				if (!SHADOW)
				{
					if (px < P.W && py < P.row1)
					{
						float vu, vv;
						sample_uv(P, px, py, k, vu, vv);
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
			jn += (u32)nidle;
			if (jn >= per)
			{
				u32 nt = 0;
				if (lane == 0) nt = nwaves + atomicAdd(&S.ctl[SHADOW ? 1 : 0], 1u);
				tile = (u32)__builtin_amdgcn_readfirstlane((int)nt);
				more = tile < S.tiles;
				jn = 0;
			}
		}
		int walking = __popcll(__ballot(job >= 0));
		if (walking == 0)
		{
			if (!more) break;
			continue;
		}
		const int leave_at = more ? 64 - kRefillAt : 0;
		do
		{
			visits += (u32)walking;
			if (job >= 0 && term == 0) term = walk_cell<SHADOW, P2, CHECK>(S, w, exempt, vhalf, k0, k1, eps_a, ctx, stride);
			walking = __popcll(__ballot(job >= 0 && term == 0));
		} while (walking > leave_at);
	}
	if (lane == 0 && visits) atomicAdd(&S.ctl[16 + 4 * ((blockIdx.x * 4u + (u32)wave) % kStatSlots) + (SHADOW ? 2 : 1)], visits);
}

// Second pass: every job of the rectangle, one lane each.
__global__ __launch_bounds__(256) void ca_stream_shadow_rays(StreamParams S)
{
	const RenderParams &P = S.R;
	if (occ_skip_enabled(P)) return;
	const u32 sh = job_shift(P), per = 64u << sh;
	const u32 j = blockIdx.x * 256u + threadIdx.x;
	const u32 tile = j / per, cand = j % per;
	bool lit = false;
	if (tile < S.tiles)
	{
		const u32 q = cand >> sh, k = cand & ((1u << sh) - 1u);
		const u32 px = P.rx0 + (tile % S.tiles_x) * 8u + (q & 7u), py = P.ry0 + (tile / S.tiles_x) * 8u + (q >> 3);
		unsigned char flag = kOcclNone;
		if (px < P.W && py < P.row1)
		{
			float vu, vv;
			sample_uv(P, px, py, k, vu, vv);
			LookupPrimaryRecordShadow tr;
			tr.answer = S.hit[j]; // (only read by primary(): a view ray that misses the volume never asks)
			shade_sample_with(P, vu, vv, tr);
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

// Last pass: one lane per pixel of the rectangle, tile by tile (a wave = one 8 x 8 tile); the plain kernel's sums and outputs.
__global__ __launch_bounds__(256) void ca_stream_resolve(StreamParams S)
{
	const RenderParams &P = S.R;
	if (occ_skip_enabled(P)) return;
	if (blockIdx.x == 0 && threadIdx.x < 3u && P.counters)
	{
		// the frame's statistics: the slots the passes before this one added to (stream order: they are complete)
		unsigned long long sum = 0;
		for (int s = 0; s < kStatSlots; s++) sum += S.ctl[16 + 4 * s + threadIdx.x];
		if (sum) atomicAdd(&P.counters[threadIdx.x], sum);
	}
	const u32 sh = job_shift(P);
	const u32 n = blockIdx.x * 256u + threadIdx.x; // pixel slot: tile * 64 + q
	const u32 tile = n >> 6, q = n & 63u;
	if (tile >= S.tiles) return;
	const u32 px = P.rx0 + (tile % S.tiles_x) * 8u + (q & 7u), py = P.ry0 + (tile / S.tiles_x) * 8u + (q >> 3);
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

template <bool P2, bool CHECK>
void launch_walks(const StreamParams &S, u32 wgs, u32 job_blocks, hipStream_t stream)
{
	hipLaunchKernelGGL((ca_stream_walk<false, P2, CHECK>), dim3(wgs), dim3(256), 0, stream, S);
	hipLaunchKernelGGL(ca_stream_shadow_rays, dim3(job_blocks), dim3(256), 0, stream, S);
	hipLaunchKernelGGL((ca_stream_walk<true, P2, CHECK>), dim3(wgs), dim3(256), 0, stream, S);
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
hipError_t launch_render_stream(const void *params, void *scratch, uint32_t W, uint32_t H, bool check, hipStream_t stream)
{
	StreamParams S;
	S.R = *static_cast<const RenderParams *>(params);
	const RenderParams &P = S.R;
	size_t hit_off, occl_off, rays_off;
	stream_scratch_bytes(W, H, P.spp, &hit_off, &occl_off, &rays_off);
	char *base = static_cast<char *>(scratch);
	S.ctl = reinterpret_cast<u32 *>(base);
	S.hit = reinterpret_cast<u32 *>(base + hit_off);
	S.occl = reinterpret_cast<unsigned char *>(base + occl_off);
	S.rays = reinterpret_cast<float4 *>(base + rays_off);
	S.tiles_x = (P.rx1 - P.rx0) / 8u;
	S.tiles = S.tiles_x * ((P.ry1 - P.ry0) / 8u);
	const bool p2 = (P.G & (P.G - 1u)) == 0u;
	S.lg = S.lc = 0;
	if (p2)
	{
		while ((1u << S.lg) < P.G) S.lg++;
		S.lc = S.lg - 5u;
	}
	hipError_t e = hipMemsetAsync(S.ctl, 0, 4096, stream);
	if (e != hipSuccess) return e;
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	const u32 wgs = min((S.tiles + 3u) / 4u, (u32)cus * (u32)kStreamWaves);
	const u32 jobs = S.tiles * 64u * P.spp;
	const u32 job_blocks = (jobs + 255u) / 256u;
	if (p2) { if (check) launch_walks<true, true>(S, wgs, job_blocks, stream); else launch_walks<true, false>(S, wgs, job_blocks, stream); }
	else { if (check) launch_walks<false, true>(S, wgs, job_blocks, stream); else launch_walks<false, false>(S, wgs, job_blocks, stream); }
	hipLaunchKernelGGL(ca_stream_resolve, dim3((S.tiles * 64u + 255u) / 256u), dim3(256), 0, stream, S);
	return hipGetLastError();
}

} // namespace ca3d
