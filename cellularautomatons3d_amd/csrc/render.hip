// Per-pixel volume renderer over the bit-packed state for gfx950 — the frame the reference's fragment shader
// (shaders/pathtraced_fragment_clustered.wgsl) converges to under a static camera (SURVEY 8(a) rows R1-R12,
// R-par): an exact DDA cell walk replaces the jittered fixed-step march (682-741) and the jittered shadow march
// (635-680); visible-cube slab tests, shading gate, Cook-Torrance BRDF, clamp of the temporal blend, light
// gizmo, depth overlay and gamma follow the shader line by line. One thread per pixel, spp sub-samples in a
// loop, three outputs: presentation RGBA8, light RGBA16F, depth RG16F (885-887).
//
// Built with -ffp-contract=off: the parity target is a plain-float CPU restatement (oracle/render_oracle.c).
#include <hip/hip_fp16.h>

#include "ca3d_internal.h"

namespace ca3d
{
namespace
{

#include "render_device.inc"


// SKIP: the variant with empty-space skipping. Both variants are launched for every frame and the one that does not
// match the frame's occupancy (occ_skip_enabled) returns at once: the choice is made on the device, without a host
// round trip, and the variant that runs carries none of the other's code in its loops.
template <bool SKIP>
__global__ __launch_bounds__(256, 4) void ca_render_packed(RenderParams P)
{
	if ((!P.legacy && occ_skip_enabled(P)) != SKIP) return;
	const bool whole_frame = SKIP && live_box_small(P); // the scheduled kernel has left the frame to the plain ones
	if (whole_frame && P.spread) return;               // ... to ca_render_packed_spread
	if (P.outside_only && !whole_frame && blockIdx.x * 16u >= P.rx0 && blockIdx.x * 16u < P.rx1 && P.row0 + blockIdx.y * 16u >= P.ry0 && P.row0 + blockIdx.y * 16u < P.ry1) return;
	const u32 px = blockIdx.x * 16u + (threadIdx.x & 15u);
	const u32 py = P.row0 + blockIdx.y * 16u + (threadIdx.x >> 4);
	if (px >= P.W || py >= P.row1) return;
	float r = 0.0f, g = 0.0f, b = 0.0f, a = 0.0f, d0 = 0.0f;
	u32 shadow = 0, pvis = 0, svis = 0;
	for (u32 k = 0; k < P.spp; k++)
	{
		const float ox = P.spp == 1u ? 0.5f : ((k & 1u) ? 0.75f : 0.25f);
		const float oy = P.spp == 1u ? 0.5f : ((k & 2u) ? 0.75f : 0.25f);
		const float vu = ((float)px + ox) / (float)P.W, vv = 1.0f - ((float)py + oy) / (float)P.H;
		const Sample s = shade_sample<SKIP>(P, vu, vv, pvis, svis);
		r += s.r; g += s.g; b += s.b; a += s.a;
		if (k == 0) d0 = s.depth;
		shadow += s.shadow_ray;
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
		const float ig = P.legacy ? 1.0f / 2.2f : 1.0f / P.u[U_GAMMA]; // legacy gamma is the constant 2.2 (:704)
		P.presentation[i] = unorm8(powf(r, ig)) | (unorm8(powf(g, ig)) << 8) | (unorm8(powf(b, ig)) << 16) | (unorm8(a) << 24);
	}
	// (a word takes ~90 atomics per us: waves with nothing to add — every sky tile — must not queue up behind it)
	if (P.counters && (shadow | pvis | svis) != 0u)
	{
		atomicAdd(&P.counters[0], (unsigned long long)shadow);
		atomicAdd(&P.counters[1], (unsigned long long)pvis);
		atomicAdd(&P.counters[2], (unsigned long long)svis);
	}
}

// The frame of a sparse volume with a SMALL live box (live_box_small): every ray outside the box's silhouette is answered at once,
// the rays inside it — a few thousand pixels next to each other — are all the walking there is. One lane per SAMPLE instead of
// per pixel (8 x 8-pixel tiles at 4 samples, the samples of a pixel summed in order through LDS) and the tiles dealt round-robin
// to a persistent launch: the same walks land on four times as many lanes spread over every CU, instead of some forty workgroups
// of the plain kernel walking four samples one after the other.
template <bool SKIP>
__global__ __launch_bounds__(256, 4) void ca_render_packed_spread(RenderParams P)
{
	if ((!P.legacy && occ_skip_enabled(P)) != SKIP) return;
	if (!live_box_small(P)) return;
	__shared__ float res[6][256];
	const u32 tid = threadIdx.x, spp = P.spp; // 1 or 4 (ca3d_render)
	const u32 tw = spp == 1u ? 16u : 8u, th = tw; // pixels of a tile: 256 / spp
	const u32 ntx = (P.W + tw - 1u) / tw, nty = (P.row1 - P.row0 + th - 1u) / th;
	const u32 p = tid / spp, k = tid % spp;
	u32 shadow = 0, pvis = 0, svis = 0;
	const float inv = 1.0f / (float)spp;
	const float ig = P.legacy ? 1.0f / 2.2f : 1.0f / P.u[U_GAMMA];
	// bg: only the tiles something can be seen in — the live box's rectangle and the light gizmo's (16-aligned: a tile lies inside or
	// outside as a whole); ca_render_background fills the others (same test there)
	PixRect live_rect;
	const bool rects = P.bg != 0u && live_rect_load(P, live_rect);
	for (u32 t = blockIdx.x; t < ntx * nty; t += gridDim.x)
	{
		if (rects)
		{
			const u32 ox = ((t % ntx) * tw) & ~15u, oy = P.row0 + (((t / ntx) * th) & ~15u);
			if (!in_pix_rect(live_rect, ox, oy) && !(ox >= P.gx0 && ox < P.gx1 && oy >= P.gy0 && oy < P.gy1)) continue;
		}
		const u32 px = (t % ntx) * tw + p % tw, py = P.row0 + (t / ntx) * th + p / tw;
		const bool live = px < P.W && py < P.row1;
		Sample s{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0u};
		if (live)
		{
			const float ox = spp == 1u ? 0.5f : ((k & 1u) ? 0.75f : 0.25f);
			const float oy = spp == 1u ? 0.5f : ((k & 2u) ? 0.75f : 0.25f);
			const float vu = ((float)px + ox) / (float)P.W, vv = 1.0f - ((float)py + oy) / (float)P.H;
			s = shade_sample<SKIP>(P, vu, vv, pvis, svis);
			shadow += s.shadow_ray;
		}
		res[0][tid] = s.r; res[1][tid] = s.g; res[2][tid] = s.b; res[3][tid] = s.a; res[4][tid] = s.depth;
		__syncthreads();
		if (live && k == 0u)
		{
			float r = 0.0f, g = 0.0f, b = 0.0f, a = 0.0f; // samples in order: the plain kernel's sums
			for (u32 kk = 0; kk < spp; kk++) { r += res[0][tid + kk]; g += res[1][tid + kk]; b += res[2][tid + kk]; a += res[3][tid + kk]; }
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
				const __half2 d = __floats2half2_rn(res[4][tid], 1.0f);
				P.depth[i] = *reinterpret_cast<const u32 *>(&d);
			}
			if (P.presentation) P.presentation[i] = unorm8(powf(r, ig)) | (unorm8(powf(g, ig)) << 8) | (unorm8(powf(b, ig)) << 16) | (unorm8(a) << 24);
		}
		__syncthreads();
	}
	if (P.counters && (shadow | pvis | svis) != 0u)
	{
		atomicAdd(&P.counters[0], (unsigned long long)shadow);
		atomicAdd(&P.counters[1], (unsigned long long)pvis);
		atomicAdd(&P.counters[2], (unsigned long long)svis);
	}
}

// Round 5: the pixels nothing can be seen in (RenderParams::bg). One lane per pixel, 16 x 16 tiles over the band. Which kernel owns a
// tile is decided from the same words on every side (the volume's rectangle from the host, the occupancy count and the live box's
// rectangle on the device): dense or scattered volume — the tiles of the volume's rectangle belong to the stream passes / the
// scheduled kernel, this kernel takes the rest; small live box — the tiles of the live box's and the gizmo's rectangles belong to
// ca_render_packed_spread, this kernel takes the rest. What it writes is what the plain kernel writes there, bit for bit:
//   outside the volume's rectangle   every view ray misses the volume: colour 0, alpha 1, depth 0 (the gizmo's tiles are traced: sample by sample,
//                                    as the plain kernel does — there is no volume to walk, only the gizmo's slab test)
//   inside it, outside the live box's a missed sample's colour 0, alpha 1, and the depth of sample 0 — the far side of the volume —
//                                    by the same float operations (shade_sample_with, MissTracer)
__global__ void ca_live_rect(RenderParams P)
{
	if (threadIdx.x != 0u) return;
	PixRect r;
	const bool ok = live_box_pix_rect(P, r);
	u32 *w = reinterpret_cast<u32 *>(P.counters + 4);
	w[1] = r.x0 | (r.x1 << 16); // (targets are at most 16 384 pixels wide and high: ca3d_render)
	w[2] = r.y0 | (r.y1 << 16);
	w[0] = ok ? 1u : 2u;
}

__global__ __launch_bounds__(256) void ca_render_background(RenderParams P)
{
	const bool skip = !P.legacy && occ_skip_enabled(P);
	const bool small = skip && live_box_small(P);
	const u32 tx = blockIdx.x * 16u, ty = P.row0 + blockIdx.y * 16u;
	const bool in_vol = tx >= P.rx0 && tx < P.rx1 && ty >= P.ry0 && ty < P.ry1;
	const bool in_giz = tx >= P.gx0 && tx < P.gx1 && ty >= P.gy0 && ty < P.gy1;
	bool far_side = false;
	if (!small)
	{
		if (in_vol) return;
	}
	else
	{
		PixRect live_rect;
		if (!live_rect_load(P, live_rect) || in_giz || in_pix_rect(live_rect, tx, ty)) return; // the spread kernel's tile
		far_side = in_vol;
	}
	const u32 px = tx + (threadIdx.x & 15u), py = ty + (threadIdx.x >> 4);
	if (px >= P.W || py >= P.row1) return;
	float r = 0.0f, g = 0.0f, b = 0.0f, a = 1.0f, d0 = 0.0f;
	if (in_giz)
	{
		r = g = b = a = 0.0f;
		u32 pvis = 0, svis = 0;
		for (u32 k = 0; k < P.spp; k++)
		{
			const float ox = P.spp == 1u ? 0.5f : ((k & 1u) ? 0.75f : 0.25f);
			const float oy = P.spp == 1u ? 0.5f : ((k & 2u) ? 0.75f : 0.25f);
			const float vu = ((float)px + ox) / (float)P.W, vv = 1.0f - ((float)py + oy) / (float)P.H;
			const Sample s = shade_sample<false>(P, vu, vv, pvis, svis); // (outside the volume's rectangle: nothing to walk, nothing to skip)
			r += s.r; g += s.g; b += s.b; a += s.a;
			if (k == 0) d0 = s.depth;
		}
		const float inv = 1.0f / (float)P.spp;
		r *= inv; g *= inv; b *= inv; a *= inv;
	}
	else if (far_side)
	{
		const float ox = P.spp == 1u ? 0.5f : 0.25f, oy = ox;
		const float vu = ((float)px + ox) / (float)P.W, vv = 1.0f - ((float)py + oy) / (float)P.H;
		MissTracer tr;
		d0 = shade_sample_with(P, vu, vv, tr).depth;
	}
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
		// (pow(+0, y) is +0 for every y > 0: the three powf of a black pixel — 300 of this kernel's 375 instructions — are skipped)
		if (r == 0.0f && g == 0.0f && b == 0.0f && ig > 0.0f) P.presentation[i] = unorm8(a) << 24;
		else P.presentation[i] = unorm8(powf(r, ig)) | (unorm8(powf(g, ig)) << 8) | (unorm8(powf(b, ig)) << 16) | (unorm8(a) << 24);
	}
}

// ================================================================================================ scheduled form
// The same frame as ca_render_packed, bit for bit, with the rays of a wave scheduled dynamically. In the plain
// kernel a lane walks its pixel's samples one after the other and the wave waits for its longest walk every time:
// walk lengths are roughly geometric, so only ~27 % of the lanes are doing work on average (PMC VALUUtilization on
// the bench scene). Here a wave owns the 64 x spp sample jobs of its 64 pixels; a lane whose walk ends takes the next
// job ("while-while" scheduling): the walk loop is left as soon as half of the walkers have finished (measured:
// leaving at 1/4, 1/2, 3/4, 7/8 finished gives 1.94, 1.78, 1.85, 2.01 ms per 1080p 4 spp frame — the per-ray set-up
// and shading code is as expensive as the walk itself, so it must not run with too few lanes either), finished
// rays are shaded / turned into shadow rays / replaced by fresh primary rays, and everybody walks again. Primary and
// shadow walks share one step function, so lanes in either phase step together. Results go to LDS per (sample,
// pixel) and each lane then sums ITS pixel's samples in sample order, which keeps the float sums identical to
// the plain kernel's.
struct RayState
{
	int job;   // -1: idle
	int phase; // 1: primary walk, 2: shadow walk
	// Amanatides-Woo walk
	v3 start, dir, inv; // inv = 1 / dir, formed once per ray (ray_cube_inv)
	float tmax, t, tx, ty, tz, dx, dy, dz;
	int ix, iy, iz, guard, wkey;
	u32 word;
	int cx, cy, cz; // shadow walk: the cell the ray starts in (exempt from the hit test); also the shaded cell
	// The sample's context — view ray, screen u, the distance at which the view ray leaves the volume, the depth — is NOT
	// here: nothing in the stepping loop reads it, and nine more live registers put the kernel over the 128 that four waves
	// per SIMD allow (it spilled 60 bytes per lane). It waits in LDS (SampleCtx) between the three stages of a sample. The
	// shaded point of the shadow stage is `start` (the shadow ray starts there).
};

// Per-lane sample context in LDS: ctx[k][thread] (conflict-free: consecutive lanes, consecutive words)
constexpr int kCtxRay = 0, kCtxVu = 3, kCtxTf = 4, kCtxDepth = 5, kCtxWords = 6;
struct SampleCtx
{
	float *base; // &ctx[0][thread]
	int stride;  // threads per block
	__device__ __forceinline__ float get(int k) const { return base[k * stride]; }
	__device__ __forceinline__ void set(int k, float v) const { base[k * stride] = v; }
	__device__ __forceinline__ v3 ray() const { return V(get(kCtxRay), get(kCtxRay + 1), get(kCtxRay + 2)); }
};

__device__ __forceinline__ void walk_begin(const RenderParams &P, RayState &w, v3 start, v3 dir, float t0, float tmax, bool clip)
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
	w.start = start;
	w.dir = dir;
	w.inv = V(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
	w.tmax = tmax;
	w.t = t0;
	w.ix = ix; w.iy = iy; w.iz = iz;
	w.word = 0;
	w.wkey = -1;
	w.guard = 0;
	// sparse-volume variant: the walk clipped to the live box (render_device.inc, live_box_clip — what walk() does); a ray that stays
	// outside it ends at its first step, before it visits a cell
	if (clip && P.live_box && !live_box_clip(P, start, dir, w.inv, t0, w.tmax, w.ix, w.iy, w.iz, w.t, w.tx, w.ty, w.tz)) w.tmax = -1.0f;
}

// One cell of `walk` above. 0: keep walking, 1: hit (tnear_out), 2: the ray left the volume / ran out of range.
template <bool SKIP>
__device__ __forceinline__ int walk_step(const RenderParams &P, RayState &w, v3 half, bool shadow, float &tnear_out, u32 &visits)
{
	const int G = (int)P.G;
	const float cs = 1.0f / (float)P.G;
	if (w.guard >= 3 * G + 3) return 2;
	w.guard++;
	if (w.t >= w.tmax) return 2;
	visits++;
	if (SKIP)
	{
		if (P.occ_coarse && !coarse_occupied(P, w.ix, w.iy, w.iz))
			return block_jump<7, 5, 5>(P, w.start, w.dir, w.inv, w.tmax, w.ix, w.iy, w.iz, w.t, w.tx, w.ty, w.tz) ? 0 : 2;
		if (!block_occupied(P, w.ix, w.iy, w.iz))
			return block_jump<5, 3, 3>(P, w.start, w.dir, w.inv, w.tmax, w.ix, w.iy, w.iz, w.t, w.tx, w.ty, w.tz) ? 0 : 2;
	}
	bool alive;
	if (P.legacy) alive = P.cells[(size_t)w.ix + ((size_t)w.iy + (size_t)w.iz * G) * G] == 1u;
	else
	{
		const int key = (w.ix >> 5) + (w.iy + w.iz * G) * (int)P.cols;
		if (key != w.wkey) { w.word = P.cells[key]; w.wkey = key; }
		alive = (w.word >> (w.ix & 31)) & 1u;
	}
	if (alive)
	{
		if (!(shadow && w.ix == w.cx && w.iy == w.cy && w.iz == w.cz))
		{
			float tn, tf;
			ray_cube_inv(w.start, w.inv, cell_origin(cs, w.ix, w.iy, w.iz), half, tn, tf);
			if (shadow ? (tn <= tf && tn >= 0.0f) : (tf >= 0.0f && tn <= tf))
			{
				tnear_out = tn;
				return 1;
			}
		}
	}
	const int sx = w.dir.x > 0.0f ? 1 : -1, sy = w.dir.y > 0.0f ? 1 : -1, sz = w.dir.z > 0.0f ? 1 : -1;
	const bool mx = w.tx <= w.ty && w.tx <= w.tz, my = !mx && w.ty <= w.tz, mz = !mx && !my;
	w.t = mx ? w.tx : (my ? w.ty : w.tz);
	w.tx += mx ? w.dx : 0.0f;
	w.ty += my ? w.dy : 0.0f;
	w.tz += mz ? w.dz : 0.0f;
	w.ix += mx ? sx : 0;
	w.iy += my ? sy : 0;
	w.iz += mz ? sz : 0;
	if ((u32)w.ix >= (u32)G || (u32)w.iy >= (u32)G || (u32)w.iz >= (u32)G) return 2;
	return 0;
}

// walk_step in three pieces, for the scheduled kernel's parking loop (below): the probe of the current cell (everything up
// to "is it alive and not the exempt start cell"), the hit test of a live cell, and the advance to the next cell.
// walk_probe: 0 dead cell (advance), 2 walk over, 3 live cell (hit test pending, nothing advanced), 4 jumped over an empty block (no advance)
template <bool SKIP>
__device__ __forceinline__ int walk_probe(const RenderParams &P, RayState &w, bool shadow, u32 &visits)
{
	const int G = (int)P.G;
	if (w.guard >= 3 * G + 3) return 2;
	w.guard++;
	if (w.t >= w.tmax) return 2;
	visits++;
	if (SKIP)
	{
		if (P.occ_coarse && !coarse_occupied(P, w.ix, w.iy, w.iz))
			return block_jump<7, 5, 5>(P, w.start, w.dir, w.inv, w.tmax, w.ix, w.iy, w.iz, w.t, w.tx, w.ty, w.tz) ? 4 : 2;
		if (!block_occupied(P, w.ix, w.iy, w.iz))
			return block_jump<5, 3, 3>(P, w.start, w.dir, w.inv, w.tmax, w.ix, w.iy, w.iz, w.t, w.tx, w.ty, w.tz) ? 4 : 2;
	}
	bool alive;
	if (P.legacy) alive = P.cells[(size_t)w.ix + ((size_t)w.iy + (size_t)w.iz * G) * G] == 1u;
	else
	{
		const int key = (w.ix >> 5) + (w.iy + w.iz * G) * (int)P.cols;
		if (key != w.wkey) { w.word = P.cells[key]; w.wkey = key; }
		alive = (w.word >> (w.ix & 31)) & 1u;
	}
	return alive && !(shadow && w.ix == w.cx && w.iy == w.cy && w.iz == w.cz) ? 3 : 0;
}
__device__ __forceinline__ bool walk_hit(const RenderParams &P, const RayState &w, v3 half, bool shadow, float &tnear_out)
{
	float tn, tf;
	ray_cube_inv(w.start, w.inv, cell_origin(1.0f / (float)P.G, w.ix, w.iy, w.iz), half, tn, tf);
	if (shadow ? (tn <= tf && tn >= 0.0f) : (tf >= 0.0f && tn <= tf))
	{
		tnear_out = tn;
		return true;
	}
	return false;
}
__device__ __forceinline__ int walk_advance(const RenderParams &P, RayState &w)
{
	const int G = (int)P.G;
	const int sx = w.dir.x > 0.0f ? 1 : -1, sy = w.dir.y > 0.0f ? 1 : -1, sz = w.dir.z > 0.0f ? 1 : -1;
	const bool mx = w.tx <= w.ty && w.tx <= w.tz, my = !mx && w.ty <= w.tz, mz = !mx && !my;
	w.t = mx ? w.tx : (my ? w.ty : w.tz);
	w.tx += mx ? w.dx : 0.0f;
	w.ty += my ? w.dy : 0.0f;
	w.tz += mz ? w.dz : 0.0f;
	w.ix += mx ? sx : 0;
	w.iy += my ? sy : 0;
	w.iz += mz ? sz : 0;
	return ((u32)w.ix >= (u32)G || (u32)w.iy >= (u32)G || (u32)w.iz >= (u32)G) ? 2 : 0;
}

// The tail every sample goes through (shade_sample's last two blocks): light gizmo, show-depth split.
__device__ __forceinline__ void sample_tail(const RenderParams &P, v3 ray, float vu, Sample &s)
{
	const float *u = P.u;
	const v3 cam = V(u[U_VIEW + 12], u[U_VIEW + 13], u[U_VIEW + 14]);
	const v3 light_pos = V(u[U_LIGHT], u[U_LIGHT + 1], u[U_LIGHT + 2]);
	float ln, lf;
	ray_cube(cam, ray, light_pos, V(0.005f, 0.005f, 0.005f), ln, lf);
	if (ln <= lf && lf >= 0.0f && s.r == 0.0f && s.g == 0.0f && s.b == 0.0f) { s.r = s.g = s.b = 1.0f; s.a = 1.0f; }
	if (u[U_SHOWDEPTH] == 1.0f && vu < 0.5f) { s.r = s.depth; s.g = 0.0f; s.b = 0.0f; s.a = 1.0f; }
}

__device__ __forceinline__ void sample_clamp(Sample &s)
{
	s.r = fminf(fmaxf(s.r, 0.0f), 1.0f);
	s.g = fminf(fmaxf(s.g, 0.0f), 1.0f);
	s.b = fminf(fmaxf(s.b, 0.0f), 1.0f);
	s.a = fminf(fmaxf(s.a, 0.0f), 1.0f);
}

// shade_sample up to the primary walk. true: the sample is complete (the view ray misses the volume).
// skip_box: the sparse-volume variant — a view ray that misses the box of the occupied blocks goes straight to the "no hit" branch
__device__ bool sample_begin(const RenderParams &P, RayState &st, const SampleCtx &ctx, float vu, float vv, Sample &s, bool skip_box)
{
	const float *u = P.u;
	const float *view = u + U_VIEW;
	s = Sample{0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0u};
	const v3 cam = V(view[12], view[13], view[14]);
	const float r = u[U_WINDOW] / u[U_WINDOW + 1];
	const v3 rl = norm3(V((vu - 0.5f) * r, vv - 0.5f, -(0.5f * P.cot_half_fov)));
	const v3 ray = V(view[0] * rl.x + view[4] * rl.y + view[8] * rl.z, view[1] * rl.x + view[5] * rl.y + view[9] * rl.z,
	                 view[2] * rl.x + view[6] * rl.y + view[10] * rl.z);
	const v3 half = V(kHalf, kHalf, kHalf);
	float tn, tf;
	ray_cube(cam, ray, V(0.0f, 0.0f, 0.0f), half, tn, tf);
	const float cam_dist = sd_box(cam, half);
	if (tn <= tf && tf >= 0.0f)
	{
		v3 enter = cam;
		const v3 exitp = cam + ray * tf;
		if (cam_dist >= 0.0f) enter = cam + ray * tn;
		const v3 seg = exitp - enter;
		walk_begin(P, st, enter, norm3(seg), 0.0f, len3(seg), skip_box);
		if (skip_box && misses_live_box(P, cam, ray)) st.tmax = -1.0f; // the walk ends at its first step, before it visits a cell: no hit
		ctx.set(kCtxRay, ray.x); ctx.set(kCtxRay + 1, ray.y); ctx.set(kCtxRay + 2, ray.z);
		ctx.set(kCtxVu, vu);
		ctx.set(kCtxTf, tf);
		st.phase = 1;
		return false;
	}
	sample_tail(P, ray, vu, s);
	return true;
}

// shade_sample between the primary walk and the shadow walk. true: complete (nothing to light at the end point).
__device__ bool sample_after_primary(const RenderParams &P, RayState &st, const SampleCtx &ctx, bool hit, float tnear, Sample &s, bool skip_box)
{
	const float *u = P.u;
	const v3 ray = ctx.ray();
	s = Sample{0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0u};
	const v3 cam = V(u[U_VIEW + 12], u[U_VIEW + 13], u[U_VIEW + 14]);
	const v3 half = V(kHalf, kHalf, kHalf);
	const float cs = 1.0f / (float)P.G;
	const float vis = cs * u[U_CELLSIZE] * 0.5f;
	const v3 vhalf = V(vis, vis, vis);
	const v3 exitp = cam + ray * ctx.get(kCtxTf);
	const v3 final_point = hit ? st.start + st.dir * tnear : exitp;
	s.depth = len3(final_point - cam);
	const v3 p = cam + ray * s.depth;
	const v3 f = V(floorf(to_cells(P, p.x)), floorf(to_cells(P, p.y)), floorf(to_cells(P, p.z)));
	const v3 origin = V(f.x * cs + cs * 0.5f - kHalf, f.y * cs + cs * 0.5f - kHalf, f.z * cs + cs * 0.5f - kHalf);
	const int cx = (int)f.x, cy = (int)f.y, cz = (int)f.z;
	const u32 state = cell_state(P, (u32)cx, (u32)cy, (u32)cz);
	const float dist = sd_box(p - origin, vhalf);
	if (state == 1u && !(dist > 0.001f))
	{
		ctx.set(kCtxDepth, s.depth);
		const v3 light_pos = V(u[U_LIGHT], u[U_LIGHT + 1], u[U_LIGHT + 2]);
		const v3 ldir = norm3(light_pos - p);
		float vn, vf;
		ray_cube(p, ldir, V(0.0f, 0.0f, 0.0f), half, vn, vf);
		const v3 vexit = p + ldir * vf;
		const v3 sseg = vexit - p;
		walk_begin(P, st, p, norm3(sseg), 0.0025f, len3(sseg), skip_box);
		st.cx = cx; st.cy = cy; st.cz = cz; // (the shaded point is st.start from here on)
		st.phase = 2;
		return false;
	}
	sample_clamp(s);
	sample_tail(P, ray, ctx.get(kCtxVu), s);
	return true;
}

// shade_sample after the shadow walk: lighting of the point the shadow ray started from (st.start) in cell (cx, cy, cz).
__device__ void sample_after_shadow(const RenderParams &P, const RayState &st, const SampleCtx &ctx, bool occluded, Sample &s)
{
	const float *u = P.u;
	s = Sample{0.0f, 0.0f, 0.0f, 1.0f, ctx.get(kCtxDepth), 0u};
	const v3 cam = V(u[U_VIEW + 12], u[U_VIEW + 13], u[U_VIEW + 14]);
	const float cs = 1.0f / (float)P.G;
	const v3 p = st.start;
	const int cx = st.cx, cy = st.cy, cz = st.cz;
	const v3 f = V((float)cx, (float)cy, (float)cz);
	const v3 origin = V(f.x * cs + cs * 0.5f - kHalf, f.y * cs + cs * 0.5f - kHalf, f.z * cs + cs * 0.5f - kHalf);
	const v3 light_pos = V(u[U_LIGHT], u[U_LIGHT + 1], u[U_LIGHT + 2]);
	const float occ = occluded ? (P.legacy ? 0.095f : kOcclusion) : 1.0f;
	if (P.legacy)
	{
		const v3 N = face_normal(p, origin);
		const float Gf = (float)P.G;
		const v3 colr = V(f.x / Gf, f.y / Gf, 1.0f - f.x / Gf);
		const v3 view_dir = norm3(p - cam);
		const float dl = len3(light_pos - p), dc = len3(cam - p);
		const float fl = fmaxf(1.0f, powf(dl, 2.0f)), fc = fmaxf(1.0f, powf(dc, 2.0f));
		const float incident = u[U_LIGHT + 3] / fl;
		const v3 inc_dir = norm3(p - light_pos);
		const float ndi = dot3(N, inc_dir);
		const v3 refl = V(inc_dir.x - 2.0f * ndi * N.x, inc_dir.y - 2.0f * ndi * N.y, inc_dir.z - 2.0f * ndi * N.z);
		const float reflected = incident * dot3(refl, V(-view_dir.x, -view_dir.y, -view_dir.z));
		s.r = occ * ((colr.x * reflected + incident * colr.x) / fc);
		s.g = occ * ((colr.y * reflected + incident * colr.y) / fc);
		s.b = occ * ((colr.z * reflected + incident * colr.z) / fc);
		s.a = occ;
		s.shadow_ray = 1u;
	}
	else
	{
		const v3 N = face_normal(p, origin);
		const float Gf = (float)P.G;
		const float cxn = (float)(u32)cx / Gf, cyn = (float)(u32)cy / Gf;
		v3 albedo = V(cxn, cyn, 1.0f - cxn);
		if (u[U_MATERIALCOLOR] != 0.0f || u[U_MATERIALCOLOR + 1] != 0.0f || u[U_MATERIALCOLOR + 2] != 0.0f)
			albedo = V(u[U_MATERIALCOLOR], u[U_MATERIALCOLOR + 1], u[U_MATERIALCOLOR + 2]);
		const v3 Vd = norm3(cam - p);
		const v3 L = norm3(light_pos - p);
		const v3 F0 = V(u[U_REFLECTIVITY], u[U_REFLECTIVITY + 1], u[U_REFLECTIVITY + 2]);
		const v3 brdf = surface_brdf(L, Vd, N, u[U_ROUGHNESS], albedo, F0);
		const float mag = u[U_LIGHT + 3];
		const float LoN = dot3(L, N);
		s.r = occ * fmaxf(0.0f, brdf.x * mag * LoN);
		s.g = occ * fmaxf(0.0f, brdf.y * mag * LoN);
		s.b = occ * fmaxf(0.0f, brdf.z * mag * LoN);
		s.shadow_ray = 1u;
	}
	sample_clamp(s);
	sample_tail(P, ctx.ray(), ctx.get(kCtxVu), s);
}

constexpr int kSchedWaves = 4; // waves per SIMD the scheduled kernel is compiled for (128 VGPRs; measured 3 / 4 / 5 / 6: 1.90 / 1.77 / 1.96 / 2.01 ms)
constexpr int kSchedChunk = 4; // samples per pixel scheduled together (LDS: 4 x 6 x 256 floats = 24 KiB per block)
#ifndef CA3D_SCHED_LEAVE_DIV
#define CA3D_SCHED_LEAVE_DIV 2
#endif
constexpr int kSchedLeaveDiv = CA3D_SCHED_LEAVE_DIV;
// RenderParams::park > 0: lanes that stand on a live cell park until that many wait for the hit test (0: test at once); CA3D_RENDER_PARK // the walk loop is left when fewer than 1/N of its walkers are still walking

// PPW pixels per wave tile (64: 16 x 4, 256: 32 x 8), NK samples per pixel scheduled together: a wave has PPW x NK jobs
// in flight per chunk, and 4 x PPW x NK x 6 floats of LDS per block hold the results. More jobs per wave = more refills
// before the tail: one sample per pixel (the interactive case) takes 256 pixels per wave, four samples 64.
// The launch is PERSISTENT: as many workgroups as fit on the chip at WPE waves per SIMD, each wave takes tiles from a
// queue until it is empty. A frame's tiles differ in cost by two orders of magnitude (sky: nothing to walk; the
// heaviest tiles of the bench scene: 30 000 cell visits) and a wave per tile in dispatch order left the chip a third full
// on average — 350 us to ramp up, a 700 us tail behind the last heavy tiles (tools/render_trace.py). Only the tiles
// inside the volume's screen rectangle are queued (one queue word sustains ~90 dequeues per us: plenty for tiles that
// walk, not for thousands of empty ones); the plain kernel renders the rest.
template <bool SKIP, int PPW, int NK, int WPE>
__global__ __launch_bounds__(256, WPE) void ca_render_packed_sched(RenderParams P)
{
	if ((!P.legacy && occ_skip_enabled(P)) != SKIP) return;
	if (SKIP && live_box_small(P)) return; // the plain kernel renders the whole frame (live_box_small)
	constexpr int TW = PPW == 64 ? 16 : 32, RW = PPW / TW, PPL = PPW / 64; // tile width, rows, pixels per lane
	__shared__ float res[NK][6][4 * PPW];
	__shared__ float ctx_lds[kCtxWords][256];
	const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const SampleCtx ctx{&ctx_lds[0][tid], 256};
	const float cs = 1.0f / (float)P.G;
	const float vis = cs * P.u[U_CELLSIZE] * 0.5f;
	const v3 vhalf = V(vis, vis, vis);
	const float inv = 1.0f / (float)P.spp;
	const float ig = P.legacy ? 1.0f / 2.2f : 1.0f / P.u[U_GAMMA];
	const u32 rtw = (P.rx1 - P.rx0) / (u32)TW, ntiles = rtw * ((P.ry1 - P.ry0) / (u32)RW); // wave tiles of the rectangle
	u32 shadow = 0, pvis = 0, svis = 0;
	for (;;)
	{
		u32 tile = 0;
		if (lane == 0) tile = (u32)atomicAdd(&P.counters[3], 1ull);
		tile = (u32)__builtin_amdgcn_readfirstlane((int)tile);
		if (tile >= ntiles) break;
		const unsigned long long trace_t0 = P.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
		const u32 vis0 = pvis + svis;
		const u32 x0 = P.rx0 + (tile % rtw) * (u32)TW, y0 = P.ry0 + (tile / rtw) * (u32)RW;
		float r[PPL], g[PPL], b[PPL], a[PPL], d0[PPL];
#pragma unroll
		for (int i = 0; i < PPL; i++) { r[i] = 0.0f; g[i] = 0.0f; b[i] = 0.0f; a[i] = 0.0f; d0[i] = 0.0f; }
		for (u32 k0 = 0; k0 < P.spp; k0 += (u32)NK)
		{
			const int nk = (int)min((u32)NK, P.spp - k0), total = PPW * nk;
			int next = 0; // wave-uniform: first job nobody has taken
			RayState st;
			st.job = -1;
			st.phase = 0;
			auto complete = [&](const Sample &s) {
				const int slot = wave * PPW + st.job % PPW, kk = st.job / PPW;
				res[kk][0][slot] = s.r; res[kk][1][slot] = s.g; res[kk][2][slot] = s.b; res[kk][3][slot] = s.a;
				res[kk][4][slot] = s.depth; res[kk][5][slot] = (float)s.shadow_ray;
				st.job = -1;
			};
			for (;;)
			{
				// idle lanes take the next jobs, in lane order
				const unsigned long long idle = __ballot(st.job < 0);
				if (idle != 0ull && next < total)
				{
					const int j = next + __popcll(idle & ((1ull << lane) - 1ull));
					if (st.job < 0 && j < total)
					{
						st.job = j;
						const int lp = j % PPW;
						const u32 k = k0 + (u32)(j / PPW);
						const u32 jx = x0 + (u32)(lp % TW), jy = y0 + (u32)(lp / TW);
						Sample s{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0u};
						bool done = true;
						if (jx < P.W && jy < P.row1)
						{
							const float ox = P.spp == 1u ? 0.5f : ((k & 1u) ? 0.75f : 0.25f);
							const float oy = P.spp == 1u ? 0.5f : ((k & 2u) ? 0.75f : 0.25f);
							const float vu = ((float)jx + ox) / (float)P.W, vv = 1.0f - ((float)jy + oy) / (float)P.H;
							done = sample_begin(P, st, ctx, vu, vv, s, SKIP);
						}
						if (done) complete(s);
					}
					next += __popcll(idle);
				}
				const int entry = __popcll(__ballot(st.job >= 0));
				if (entry == 0)
				{
					if (next >= total) break;
					continue;
				}
				// walk until half of the walkers have finished (the tail of a chunk runs to the end)
				const int leave_below = (next < total || entry > 16) ? max(entry / kSchedLeaveDiv, 1) : 1;
				int term = 0;
				float tnear = 0.0f;
				const int park = (int)P.park;
				if (park > 0)
				{
					// Parking: at the bench scene's density a walker stands on a live cell in 3 % of its steps, so with ~30 walkers
					// some lane needs the 45-instruction hit test in six iterations out of ten — and it ran for that one lane.
					// A lane that finds a live cell parks instead; the hit tests run together once kSchedPark lanes wait (or nobody
					// is left stepping, or the loop is about to be left). Every ray still sees the same cells and tests in the
					// same order: the frame is unchanged.
					bool parked = false;
					for (;;)
					{
						const bool shadow = st.phase == 2;
						bool advance = false;
						if (st.job >= 0 && term == 0 && !parked)
						{
							const int pr = walk_probe<SKIP>(P, st, shadow, shadow ? svis : pvis);
							if (pr == 2) term = 2;
							else if (pr == 3) parked = true;
							else advance = pr == 0;
						}
						const int npark = __popcll(__ballot(parked));
						const int stepping = __popcll(__ballot(st.job >= 0 && term == 0 && !parked));
						const int walking = npark + stepping;
						const bool leaving = walking == 0 || walking < leave_below;
						if (npark > 0 && (npark >= park || stepping == 0 || leaving))
						{
							if (parked)
							{
								if (walk_hit(P, st, vhalf, shadow, tnear)) term = 1;
								else advance = true;
								parked = false;
							}
						}
						if (advance) term = walk_advance(P, st);
						const int still = __popcll(__ballot(st.job >= 0 && term == 0));
						if (still == 0 || still < leave_below) // (nobody is parked here: a wave that leaves has just run its hit tests)
						{
							if (__ballot(parked) == 0ull) break;
						}
					}
				}
				else
				for (;;)
				{
					if (st.job >= 0 && term == 0) term = walk_step<SKIP>(P, st, vhalf, st.phase == 2, tnear, st.phase == 2 ? svis : pvis);
					const int walking = __popcll(__ballot(st.job >= 0 && term == 0));
					if (walking == 0 || walking < leave_below) break;
				}
				if (st.job >= 0 && term != 0)
				{
					Sample s;
					bool done = true;
					if (st.phase == 1) done = sample_after_primary(P, st, ctx, term == 1, tnear, s, SKIP);
					else sample_after_shadow(P, st, ctx, term == 1, s);
					if (done) complete(s);
				}
			}
			// this lane's own pixels, samples in order: the same float sums as the plain kernel
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int i = 0; i < PPL; i++)
			{
				const int slot = wave * PPW + i * 64 + lane;
				for (int kk = 0; kk < nk; kk++)
				{
					r[i] += res[kk][0][slot]; g[i] += res[kk][1][slot]; b[i] += res[kk][2][slot]; a[i] += res[kk][3][slot];
					if (k0 == 0 && kk == 0) d0[i] = res[kk][4][slot];
					shadow += (u32)res[kk][5][slot];
				}
			}
			__builtin_amdgcn_wave_barrier();
		}
#pragma unroll
		for (int i = 0; i < PPL; i++)
		{
			const int lp = i * 64 + lane;
			const u32 px = x0 + (u32)(lp % TW), py = y0 + (u32)(lp / TW);
			if (px >= P.W || py >= P.row1) continue;
			const float rr = r[i] * inv, gg = g[i] * inv, bb = b[i] * inv, aa = a[i] * inv;
			const size_t idx = (size_t)py * P.W + px;
			if (P.light)
			{
				const __half2 rg = __floats2half2_rn(rr, gg), ba = __floats2half2_rn(bb, 1.0f);
				uint2 v;
				v.x = *reinterpret_cast<const u32 *>(&rg);
				v.y = *reinterpret_cast<const u32 *>(&ba);
				P.light[idx] = v;
			}
			if (P.depth)
			{
				const __half2 d = __floats2half2_rn(d0[i], 1.0f);
				P.depth[idx] = *reinterpret_cast<const u32 *>(&d);
			}
			if (P.presentation)
				P.presentation[idx] = unorm8(powf(rr, ig)) | (unorm8(powf(gg, ig)) << 8) | (unorm8(powf(bb, ig)) << 16) | (unorm8(aa) << 24);
		}
		if (P.trace)
		{
			u32 tv = pvis + svis - vis0;
			for (int o = 32; o > 0; o >>= 1) tv += __shfl_xor(tv, o);
			if (lane == 0)
			{
				unsigned long long *t = P.counters + 8 + 4 * (size_t)tile;
				t[0] = trace_t0;
				t[1] = __builtin_amdgcn_s_memrealtime();
				t[2] = (unsigned long long)__builtin_amdgcn_s_getreg((15 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32); // HW_ID (wave slot, SIMD, CU, SH, SE), XCC_ID
				t[3] = tv;
			}
		}
	}
	atomicAdd(&P.counters[0], (unsigned long long)shadow);
	atomicAdd(&P.counters[1], (unsigned long long)pvis);
	atomicAdd(&P.counters[2], (unsigned long long)svis);
}

// ================================================================================================ occupancy
// occ: one bit per block of 32 x 8 x 8 cells (index wx + cols * (by + G/8 * bz)), set when any cell of the block is
// alive; the 64-bit word after the bits accumulates their count. Rebuilt from the current state before every frame
// (one read of the volume: ~5 us at 512^3).
// live_box (nullable): the corner cells of the occupied blocks, reduced to six maxima (RenderParams::live_box) — per wave by
// shuffles, per workgroup in LDS, then at most six atomics per workgroup.
__global__ __launch_bounds__(256) void ca_occupancy(const u32 *__restrict__ cells, unsigned long long *__restrict__ occ, u32 G, u32 cols,
                                                     u32 nblocks, u32 occ_words, u32 *__restrict__ live_box)
{
	__shared__ u32 wg_box[6];
	if (threadIdx.x < 6u) wg_box[threadIdx.x] = 0u;
	const u32 bk = blockIdx.x * 256u + threadIdx.x; // consecutive lanes = consecutive wx
	u32 any = 0;
	u32 box[6] = {0, 0, 0, 0, 0, 0};
	if (bk < nblocks)
	{
		const u32 nb = G >> 3;
		const u32 wx = bk % cols, byz = bk / cols, by = byz % nb, bz = byz / nb;
		for (u32 dz = 0; dz < 8; dz++)
			for (u32 dy = 0; dy < 8; dy++) any |= cells[wx + ((size_t)(by * 8 + dy) + (size_t)(bz * 8 + dz) * G) * cols];
		box[0] = (wx + 1u) * 32u; box[1] = ~(wx * 32u);
		box[2] = (by + 1u) * 8u; box[3] = ~(by * 8u);
		box[4] = (bz + 1u) * 8u; box[5] = ~(bz * 8u);
	}
	const unsigned long long m = __ballot(any != 0);
	if ((threadIdx.x & 63u) == 0 && (bk >> 6) < occ_words)
	{
		occ[bk >> 6] = m;
		if (m) atomicAdd(&occ[occ_words], (unsigned long long)__popcll(m));
	}
	if (!live_box) return;
	__syncthreads();
	if (m)
	{
#pragma unroll
		for (int i = 0; i < 6; i++)
		{
			u32 v = any ? box[i] : 0u;
			for (int o = 32; o > 0; o >>= 1) v = max(v, (u32)__shfl_xor((int)v, o));
			if ((threadIdx.x & 63u) == 0) atomicMax(&wg_box[i], v);
		}
	}
	__syncthreads();
	if (threadIdx.x < 6u && wg_box[threadIdx.x]) atomicMax(&live_box[threadIdx.x], wg_box[threadIdx.x]);
}

// Second level from the first: one bit per 4 x 4 x 4 fine blocks (128 x 32 x 32 cells), stored after the count word.
__global__ __launch_bounds__(256) void ca_occupancy_coarse(unsigned long long *__restrict__ occ, u32 G, u32 cols, u32 ncoarse, u32 occ_words)
{
	const u32 ck = blockIdx.x * 256u + threadIdx.x; // cwx + (cols / 4) * (cby + (G / 32) * cbz)
	u32 any = 0;
	if (ck < ncoarse)
	{
		const u32 ccols = cols >> 2, nb = G >> 3, cnb = G >> 5;
		const u32 cwx = ck % ccols, cbyz = ck / ccols, cby = cbyz % cnb, cbz = cbyz / cnb;
		for (u32 dz = 0; dz < 4; dz++)
			for (u32 dy = 0; dy < 4; dy++)
			{
				const u32 bk = cwx * 4u + cols * ((cby * 4u + dy) + nb * (cbz * 4u + dz)); // 4 consecutive bits: cols % 4 == 0
				any |= (u32)((occ[bk >> 6] >> (bk & 63u)) & 0xFull);
			}
	}
	const unsigned long long m = __ballot(any != 0);
	if ((threadIdx.x & 63u) == 0 && ck < ((ncoarse + 63u) & ~63u)) occ[occ_words + 1u + (ck >> 6)] = m;
}

// ================================================================================================ literal frame
// (FrameParams, the jitter hash, the history look-ups: render_device.inc)
__global__ __launch_bounds__(256) void ca_render_frame_packed(FrameParams F)
{
	const RenderParams &P = F.base;
	const u32 px = blockIdx.x * 16u + (threadIdx.x & 15u);
	const u32 py = blockIdx.y * 16u + (threadIdx.x >> 4);
	if (px >= P.W || py >= P.H) return;
	const float *u = P.u;
	const float *view = u + U_VIEW;
	const float vu = ((float)px + 0.5f) / (float)P.W, vv = 1.0f - ((float)py + 0.5f) / (float)P.H;
	float out[4] = {0.0f, 0.0f, 0.0f, 1.0f};
	float mixed_depth = 0.0f;
	u32 shadow = 0, pvis = 0, svis = 0;
	const v3 cam = V(view[12], view[13], view[14]);
	const float ar = u[U_WINDOW] / u[U_WINDOW + 1];
	const v3 rl = norm3(V((vu - 0.5f) * ar, vv - 0.5f, -(0.5f * P.cot_half_fov)));
	const v3 ray = V(view[0] * rl.x + view[4] * rl.y + view[8] * rl.z, view[1] * rl.x + view[5] * rl.y + view[9] * rl.z,
	                 view[2] * rl.x + view[6] * rl.y + view[10] * rl.z);
	const v3 half = V(kHalf, kHalf, kHalf);
	const float cs = 1.0f / (float)P.G;
	const float vis = cs * u[U_CELLSIZE] * 0.5f;
	const v3 vhalf = V(vis, vis, vis);
	float tn, tf;
	ray_cube(cam, ray, V(0.0f, 0.0f, 0.0f), half, tn, tf);
	const float cam_dist = sd_box(cam, half);
	if (tn <= tf && tf >= 0.0f)
	{
		v3 enter = cam;
		const v3 exitp = cam + ray * tf;
		if (cam_dist >= 0.0f) enter = cam + ray * tn;
		// rayMarchDepth :682-741
		v3 final_point = exitp;
		{
			const v3 seg = exitp - enter;
			const v3 dir = norm3(seg);
			const float march = len3(seg);
			const float step = march / u[U_DEPTHSAMPLES];
			const float rnd = n1rand(P, vu, vv);
			float depth = step * rnd + 0.01f;
			for (int guard = 0; depth < march && guard < 100000; guard++)
			{
				pvis++;
				const v3 sp = enter + dir * depth;
				const v3 cc = V(floorf(to_cells(P, sp.x)), floorf(to_cells(P, sp.y)), floorf(to_cells(P, sp.z)));
				const v3 origin = V(cc.x * cs + cs * 0.5f - kHalf, cc.y * cs + cs * 0.5f - kHalf, cc.z * cs + cs * 0.5f - kHalf);
				if (cell_state(P, f2u(cc.x), f2u(cc.y), f2u(cc.z)) != 0u)
				{
					float a, b;
					ray_cube(enter, dir, origin, vhalf, a, b);
					if (b >= 0.0f && a <= b) { final_point = enter + dir * a; break; }
				}
				depth += step;
			}
		}
		float uvx, uvy;
		reprojected_uv(P, final_point, uvx, uvy);
		float pdr = 0.0f;
		{
			size_t ti;
			if (F.prev_depth && texel_xy(P, uvx * u[U_WINDOW], uvy * u[U_WINDOW + 1], ti))
			{
				const u32 raw = F.prev_depth[ti];
				pdr = __half2float(__ushort_as_half((unsigned short)(raw & 0xFFFFu)));
			}
		}
		// estimateLikelyDepth :743-798
		const float *pview = u + U_PREVVIEW;
		const v3 pcam = V(pview[12], pview[13], pview[14]);
		{
			const float current = len3(final_point - cam);
			const v3 view_ray = norm3(ray);
			const v3 view_ray2 = norm3(final_point - pcam);
			const v3 reproj_point = pcam + view_ray2 * pdr;
			mixed_depth = current;
			const CellU rc = cell_u(P, reproj_point), cc = cell_u(P, final_point);
			if (cell_state(P, rc.x, rc.y, rc.z) == 1u && cc.idx != rc.idx && pdr < current)
			{
				float a, b;
				ray_cube(cam, view_ray, rc.origin, vhalf, a, b);
				if (a <= b && a >= 0.0f) mixed_depth = a;
			}
		}
		const v3 p = cam + ray * mixed_depth;
		reprojected_uv(P, p, uvx, uvy);
		// calculateLightingAndOcclusionAt :379-427
		float col[3] = {0.0f, 0.0f, 0.0f};
		{
			const CellU cell = cell_u(P, p);
			const u32 st = cell_state(P, cell.x, cell.y, cell.z);
			const float dist = sd_box(p - cell.origin, vhalf);
			if (st == 1u && !(dist > 0.001f))
			{
				const v3 light_pos = V(u[U_LIGHT], u[U_LIGHT + 1], u[U_LIGHT + 2]);
				const v3 ldir = norm3(light_pos - p);
				const float rnd = n1rand(P, vu, vv);
				float vn, vf;
				ray_cube(p, ldir, V(0.0f, 0.0f, 0.0f), half, vn, vf);
				const v3 vexit = p + ldir * vf;
				// rayMarchShadow :635-680
				float occ = 1.0f;
				{
					const v3 seg = vexit - p;
					const v3 dir = norm3(seg);
					const float march = len3(seg);
					const float step = fmaxf(cs * u[U_CELLSIZE], march / u[U_SHADOWSAMPLES]);
					float depth = step * rnd + 0.0025f;
					for (int guard = 0; depth < march && guard < 100000; guard++)
					{
						svis++;
						const v3 sp = p + dir * depth;
						const v3 cc = V(floorf(to_cells(P, sp.x)), floorf(to_cells(P, sp.y)), floorf(to_cells(P, sp.z)));
						const u32 ux = f2u(cc.x), uy = f2u(cc.y), uz = f2u(cc.z);
						const u32 st2 = cell_state(P, ux, uy, uz);
						const v3 origin = V(cc.x * cs + cs * 0.5f - kHalf, cc.y * cs + cs * 0.5f - kHalf, cc.z * cs + cs * 0.5f - kHalf);
						if ((ux != cell.x || uy != cell.y || uz != cell.z) && st2 == 1u)
						{
							float a, b;
							ray_cube(p, dir, origin, vhalf, a, b);
							if (a <= b && a >= 0.0f) { occ = kOcclusion; break; }
						}
						depth += step;
					}
				}
				shadow = 1u;
				const v3 N = face_normal(p, cell.origin);
				const float Gf = (float)P.G;
				const float cxn = (float)cell.x / Gf, cyn = (float)cell.y / Gf;
				v3 albedo = V(cxn, cyn, 1.0f - cxn);
				if (u[U_MATERIALCOLOR] != 0.0f || u[U_MATERIALCOLOR + 1] != 0.0f || u[U_MATERIALCOLOR + 2] != 0.0f)
					albedo = V(u[U_MATERIALCOLOR], u[U_MATERIALCOLOR + 1], u[U_MATERIALCOLOR + 2]);
				const v3 Vd = norm3(cam - p);
				const v3 L = norm3(light_pos - p);
				const v3 F0 = V(u[U_REFLECTIVITY], u[U_REFLECTIVITY + 1], u[U_REFLECTIVITY + 2]);
				const v3 brdf = surface_brdf(L, Vd, N, u[U_ROUGHNESS], albedo, F0);
				const float mag = u[U_LIGHT + 3];
				const float LoN = dot3(L, N);
				col[0] = occ * fmaxf(0.0f, brdf.x * mag * LoN);
				col[1] = occ * fmaxf(0.0f, brdf.y * mag * LoN);
				col[2] = occ * fmaxf(0.0f, brdf.z * mag * LoN);
			}
		}
		// mixWithReprojectedColor :429-471
		{
			float pc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
			size_t ti;
			if (F.prev_light && texel_xy(P, uvx * u[U_WINDOW], uvy * u[U_WINDOW + 1], ti))
			{
				const uint2 raw = F.prev_light[ti];
				pc[0] = __half2float(__ushort_as_half((unsigned short)(raw.x & 0xFFFFu)));
				pc[1] = __half2float(__ushort_as_half((unsigned short)(raw.x >> 16)));
				pc[2] = __half2float(__ushort_as_half((unsigned short)(raw.y & 0xFFFFu)));
				pc[3] = __half2float(__ushort_as_half((unsigned short)(raw.y >> 16)));
			}
			const v3 rdir = norm3(p - pcam);
			const v3 rpoint = pcam + rdir * pdr;
			const CellU rcell = cell_u(P, rpoint), ccell = cell_u(P, p);
			const bool outside = uvx < 0.0f || uvx > 1.0f || uvy < 0.0f || uvy > 1.0f;
			if (outside || ccell.idx != rcell.idx) { out[0] = col[0]; out[1] = col[1]; out[2] = col[2]; out[3] = 1.0f; }
			else
			{
				const float a = u[U_TEMPORALALPHA];
				const float cur[4] = {col[0], col[1], col[2], 1.0f};
				for (int k = 0; k < 4; k++) out[k] = fminf(fmaxf(pc[k] * (1.0f - a) + cur[k] * a, 0.0f), 1.0f);
			}
		}
	}
	{
		const v3 light_pos = V(u[U_LIGHT], u[U_LIGHT + 1], u[U_LIGHT + 2]);
		float ln, lf;
		ray_cube(cam, ray, light_pos, V(0.005f, 0.005f, 0.005f), ln, lf);
		if (ln <= lf && lf >= 0.0f && out[0] == 0.0f && out[1] == 0.0f && out[2] == 0.0f) { out[0] = out[1] = out[2] = out[3] = 1.0f; }
	}
	if (u[U_SHOWDEPTH] == 1.0f && vu < 0.5f) { out[0] = mixed_depth; out[1] = 0.0f; out[2] = 0.0f; out[3] = 1.0f; }
	const size_t i = (size_t)py * P.W + px;
	if (P.light)
	{
		const __half2 rg = __floats2half2_rn(out[0], out[1]), ba = __floats2half2_rn(out[2], 1.0f);
		uint2 v;
		v.x = *reinterpret_cast<const u32 *>(&rg);
		v.y = *reinterpret_cast<const u32 *>(&ba);
		P.light[i] = v;
	}
	if (P.depth)
	{
		const __half2 d = __floats2half2_rn(mixed_depth, 1.0f);
		P.depth[i] = *reinterpret_cast<const u32 *>(&d);
	}
	if (P.presentation)
	{
		const float ig = 1.0f / u[U_GAMMA];
		P.presentation[i] = unorm8(powf(out[0], ig)) | (unorm8(powf(out[1], ig)) << 8) | (unorm8(powf(out[2], ig)) << 16) | (unorm8(out[3]) << 24);
	}
	if (P.counters)
	{
		atomicAdd(&P.counters[0], (unsigned long long)shadow);
		atomicAdd(&P.counters[1], (unsigned long long)pvis);
		atomicAdd(&P.counters[2], (unsigned long long)svis);
	}
}

} // namespace

// Screen rectangle of the volume (the cube [-0.5, 0.5]^3) in pixels, from the same camera model as sample_begin: the 8
// corners projected, two pixels of margin, x aligned to 32 and y to 16, clipped to the band. A corner beside or behind
// the camera: the whole band. Only a hint for WHERE the scheduled kernel runs — the plain kernel renders the rest and both
// produce the same pixels, so a rectangle that is too small costs time, never correctness.
static void volume_rect(RenderParams &P)
{
	const float *v = P.u + U_VIEW;
	const double m[3][3] = {{v[0], v[4], v[8]}, {v[1], v[5], v[9]}, {v[2], v[6], v[10]}}; // world dir = m * camera dir
	const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
	                   m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
	const double aspect = (double)P.u[U_WINDOW] / (double)P.u[U_WINDOW + 1];
	double lo_x = 1e30, hi_x = -1e30, lo_y = 1e30, hi_y = -1e30;
	bool whole = !(fabs(det) > 1e-12) || !(aspect > 0.0);
	for (int c = 0; c < 8 && !whole; c++)
	{
		const double d[3] = {(c & 1 ? 0.5 : -0.5) - v[12], (c & 2 ? 0.5 : -0.5) - v[13], (c & 4 ? 0.5 : -0.5) - v[14]};
		double q[3]; // camera-space position: m^-1 * d (Cramer)
		for (int k = 0; k < 3; k++)
		{
			double t[3][3];
			for (int i = 0; i < 3; i++)
				for (int j = 0; j < 3; j++) t[i][j] = j == k ? d[i] : m[i][j];
			q[k] = (t[0][0] * (t[1][1] * t[2][2] - t[1][2] * t[2][1]) - t[0][1] * (t[1][0] * t[2][2] - t[1][2] * t[2][0]) +
			        t[0][2] * (t[1][0] * t[2][1] - t[1][1] * t[2][0])) / det;
		}
		if (!(q[2] < -1e-3)) { whole = true; break; } // the view ray looks down -z
		const double vu = 0.5 - 0.5 * P.cot_half_fov * (q[0] / q[2]) / aspect, vv = 0.5 - 0.5 * P.cot_half_fov * (q[1] / q[2]);
		const double x = vu * P.W, y = (1.0 - vv) * P.H;
		lo_x = fmin(lo_x, x); hi_x = fmax(hi_x, x); lo_y = fmin(lo_y, y); hi_y = fmax(hi_y, y);
	}
	if (whole) { lo_x = 0; hi_x = P.W; lo_y = 0; hi_y = P.H; }
	const double x0 = fmax(0.0, floor(lo_x) - 2.0), x1 = fmin((double)P.W, ceil(hi_x) + 2.0);
	const double y0 = fmax((double)P.row0, floor(lo_y) - 2.0), y1 = fmin((double)P.row1, ceil(hi_y) + 2.0);
	if (!(x1 > x0) || !(y1 > y0)) return; // the volume is off screen (or outside the band): nothing to schedule
	P.rx0 = ((u32)x0 / 32u) * 32u;
	P.rx1 = (((u32)x1 + 31u) / 32u) * 32u;
	P.ry0 = P.row0 + (((u32)y0 - P.row0) / 16u) * 16u;
	P.ry1 = P.row0 + (((u32)y1 - P.row0 + 15u) / 16u) * 16u;
}

hipError_t launch_render(const RenderLaunch &l, hipStream_t stream)
{
	RenderParams P;
	P.cells = l.cells;
	P.G = l.G;
	P.cols = l.G / 32u;
	P.W = l.W;
	P.H = l.H;
	P.spp = l.spp;
	P.cot_half_fov = (float)(1.0 / tan(37.5 * 3.14159265359 / 180.0)); // COT_HALF_FOV :70
	for (int i = 0; i < U_LIVE; i++) P.u[i] = l.uniforms[i];
	P.presentation = l.presentation;
	P.light = reinterpret_cast<uint2 *>(l.light);
	P.depth = l.depth;
	P.counters = l.counters;
	P.legacy = l.legacy ? 1u : 0u;
	P.indirect = l.indirect && !l.legacy ? 1u : 0u;
	P.trace = l.trace ? 1u : 0u;
	P.occ = nullptr;
	P.live_box = nullptr;
	P.spread = 0;
	static const int park_env = getenv("CA3D_RENDER_PARK") ? atoi(getenv("CA3D_RENDER_PARK")) : 0; // tuning: see RenderParams::park
	P.park = (u32)(park_env > 0 ? park_env : 0);
	P.occ_words = 0;
	P.occ_coarse = 0;
	if (!l.legacy && l.mode != 1 && l.occ)
	{
		const u32 nblocks = P.cols * (l.G >> 3) * (l.G >> 3);
		P.occ_words = (nblocks + 63u) / 64u;
		P.occ_coarse = l.G % 128u == 0 ? 1u : 0u;
		// the box of the occupied blocks: six words behind the coarse bits (kept with the bits: l.occ_valid)
		const size_t coarse_words = (nblocks / 64u + 63u) / 64u;
		u32 *box = reinterpret_cast<u32 *>(l.occ + P.occ_words + 1u + coarse_words);
		if (!l.occ_valid) // the engine says the bits are those of this very state (a frame of the same state was drawn before)
		{
			hipError_t e = hipMemsetAsync(l.occ + P.occ_words, 0, sizeof(unsigned long long), stream);
			if (e == hipSuccess) e = hipMemsetAsync(box, 0, 6u * sizeof(u32), stream);
			if (e != hipSuccess) return e;
			hipLaunchKernelGGL(ca_occupancy, dim3((nblocks + 255u) / 256u), dim3(256), 0, stream, l.cells, l.occ, l.G, P.cols, nblocks, P.occ_words, box);
			if (l.occ_built) *l.occ_built = true;
			if (P.occ_coarse)
				hipLaunchKernelGGL(ca_occupancy_coarse, dim3((nblocks / 64u + 255u) / 256u), dim3(256), 0, stream, l.occ, l.G, P.cols, nblocks / 64u, P.occ_words);
		}
		P.live_box = box;
		P.occ = l.occ;
	}
	P.row0 = l.row0;
	P.row1 = l.row1 ? l.row1 : l.H;
	P.rx0 = P.rx1 = P.ry0 = P.ry1 = P.outside_only = 0;
	P.bg = P.gx0 = P.gx1 = P.gy0 = P.gy1 = 0;
	const dim3 grid((l.W + 15u) / 16u, (l.mode == 1 ? l.H + 15u : P.row1 - P.row0 + 15u) / 16u);
	if (l.mode == 1)
	{
		FrameParams F;
		F.base = P;
		F.prev_light = reinterpret_cast<const uint2 *>(l.prev_light);
		F.prev_depth = l.prev_depth;
		if (l.bricks && frame_bricks_applies(l.G))
		{
			hipError_t e = launch_render_frame_bricks(&F, l.bricks, l.bricks_valid, stream, l.bricks_built);
			if (e != hipSuccess) return e;
		}
		else hipLaunchKernelGGL(ca_render_frame_packed, grid, dim3(256), 0, stream, F);
	}
	else if (l.sched && !P.indirect && l.counters)
	{
		// the scheduled kernel on the volume's screen rectangle (persistent: as many workgroups as the chip holds at 4 waves
		// per SIMD, fewer when the rectangle has fewer wave tiles), the plain kernel on the tiles around it
		volume_rect(P);
		const bool one = l.spp == 1; // one sample per pixel: 256 pixels per wave tile (32 x 8), else 64 (16 x 4)
		const u32 tiles = ((P.rx1 - P.rx0) / (one ? 32u : 16u)) * ((P.ry1 - P.ry0) / (one ? 8u : 4u));
		const bool around = tiles == 0 || P.rx0 > 0 || P.rx1 < l.W || P.ry0 > P.row0 || P.ry1 < P.row1; // tiles outside the rectangle exist
		// Everything but the dense-volume scheduled kernel — the plain kernel on the tiles around the rectangle (view rays that miss the
		// volume), the sparse-volume variants (which return at once on a dense volume, as the dense ones do on a sparse one: each
		// tests the occupancy count before it touches the tile queue) — goes to a second stream when there is one: it then runs
		// BESIDE the scheduled launch, which is persistent and ends in a tail with most CUs idle, and joins this stream before the
		// frame is done. The kernels write disjoint pixels. The fork sits behind the counters' memset and the occupancy pass.
		// Pixels nothing can be seen in are filled, not traced (ca_render_background) — unless the depth overlay colours them (:880-883),
		// the volume is the legacy one, or the light gizmo's cube has a corner beside or behind the camera (no rectangle for it).
		static const bool bg_off = getenv("CA3D_RENDER_BG") && atoi(getenv("CA3D_RENDER_BG")) == 0; // tuning / A-B: trace every pixel
		if (!bg_off && !P.legacy && !(P.u[U_SHOWDEPTH] == 1.0f))
		{
			const double lo[3] = {(double)P.u[U_LIGHT] - 0.005, (double)P.u[U_LIGHT + 1] - 0.005, (double)P.u[U_LIGHT + 2] - 0.005};
			const double hi[3] = {(double)P.u[U_LIGHT] + 0.005, (double)P.u[U_LIGHT + 1] + 0.005, (double)P.u[U_LIGHT + 2] + 0.005};
			PixRect gr;
			if (box_pix_rect(P.u, P.cot_half_fov, P.W, P.H, P.row0, P.row1, lo, hi, gr))
			{
				P.bg = 1u;
				P.gx0 = gr.x0; P.gx1 = gr.x1; P.gy0 = gr.y0; P.gy1 = gr.y1;
			}
		}
		// the live box's screen rectangle, once per frame, for the kernels that own or skip tiles by it (behind the occupancy pass, before the fork)
		if (P.bg && P.occ && P.live_box) hipLaunchKernelGGL(ca_live_rect, dim3(1), dim3(64), 0, stream, P);
		const bool beside = tiles && l.aux && l.ev_fork && l.ev_join;
		hipStream_t side = beside ? l.aux : stream;
		if (beside)
		{
			hipError_t e = hipEventRecord(l.ev_fork, stream);
			if (e == hipSuccess) e = hipStreamWaitEvent(l.aux, l.ev_fork, 0);
			if (e == hipSuccess && l.after) e = hipStreamWaitEvent(l.aux, l.after, 0); // the side kernels write pixels: after the frame before
			if (e != hipSuccess) return e;
		}

		if (tiles)
		{
			int dev = 0, cus = 256;
			if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
			const u32 wgs = min((tiles + 3u) / 4u, (u32)cus * (u32)kSchedWaves);
			// the dense-volume kernel: the ray-stream passes (render_stream.hip) when the engine has given them scratch, else the
			// in-wave scheduled kernel (legacy volumes, traces, option render_stream 0)
			const bool streamed = l.stream_scratch && !P.legacy && !P.trace;
			bool stream_sparse = false; // the stream passes draw the scattered sparse volume too (ca_stream_walk2<.., SKIP>): no scheduled launch for it
			if (streamed)
			{
				hipError_t e = launch_render_stream(&P, l.stream_scratch, l.W, l.H, l.stream_check, l.bricks, l.bricks_valid, stream, l.bricks_built, l.after, l.walk_share_pct, &stream_sparse);
				if (e != hipSuccess) return e;
			}
			if (one)
			{
				if (!streamed) hipLaunchKernelGGL((ca_render_packed_sched<false, 256, 1, kSchedWaves>), dim3(wgs), dim3(256), 0, stream, P);
				if (P.occ && !stream_sparse) hipLaunchKernelGGL((ca_render_packed_sched<true, 256, 1, kSchedWaves>), dim3(wgs), dim3(256), 0, side, P);
			}
			else
			{
				if (!streamed) hipLaunchKernelGGL((ca_render_packed_sched<false, 64, kSchedChunk, kSchedWaves>), dim3(wgs), dim3(256), 0, stream, P);
				if (P.occ && !stream_sparse) hipLaunchKernelGGL((ca_render_packed_sched<true, 64, kSchedChunk, kSchedWaves>), dim3(wgs), dim3(256), 0, side, P);
			}
		}
		if (!beside && l.after)
		{
			// two frames in flight, one stream per frame: the side kernels below write pixels — after the frame before (the stream passes'
			// resolve has waited for the same event; a frame without stream passes has not)
			hipError_t e = hipStreamWaitEvent(stream, l.after, 0);
			if (e != hipSuccess) return e;
		}
		P.outside_only = tiles ? 1u : 0u;
		if (P.bg) hipLaunchKernelGGL(ca_render_background, grid, dim3(256), 0, side, P);
		else if (around) hipLaunchKernelGGL(ca_render_packed<false>, grid, dim3(256), 0, side, P);
		if (P.occ && P.live_box)
		{
			// a sparse volume with a small live box: its frame belongs to the spread kernel (a persistent launch: it costs a dense frame
			// a thousand workgroups that return at once)
			int dev = 0, cus = 256;
			if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
			P.spread = 1u;
			hipLaunchKernelGGL(ca_render_packed_spread<true>, dim3((u32)cus * 4u), dim3(256), 0, side, P);
		}
		if (P.occ && around && !P.bg) hipLaunchKernelGGL(ca_render_packed<true>, grid, dim3(256), 0, side, P);
		if (beside)
		{
			hipError_t e = hipEventRecord(l.ev_join, l.aux);
			if (e == hipSuccess) e = hipStreamWaitEvent(stream, l.ev_join, 0);
			if (e != hipSuccess) return e;
		}
	}
	else
	{
		hipLaunchKernelGGL(ca_render_packed<false>, grid, dim3(256), 0, stream, P);
		if (P.occ) hipLaunchKernelGGL(ca_render_packed<true>, grid, dim3(256), 0, stream, P);
	}
	return hipGetLastError();
}

} // namespace ca3d
