// The reference's OWN frame (ca3d_set_option("render_mode", 1); pathtraced_fragment_clustered.wgsl:800-890 — one jittered fixed-step
// sample per pixel, history look-ups, depth repair, temporal blend) re-designed for gfx950. render.hip keeps the first form of this
// mode, ca_render_frame_packed, which follows fragment_main statement by statement: every march sample there is a dependent 4-byte
// read of the row-major packed volume, 64 lanes reading 64 different cache lines, and the kernel measured 1.21 ms per 1080p frame
// with the SIMDs issuing in 6 % of their cycles and the waves waiting for memory in 80 % of theirs (profiles/r4_a_pmc_render_literal_before.json:
// 17 M look-ups, 11 M L2 requests at 1 100 cycles each, L2 hit rate 0.61). A fixed-step march has no address dependence — every
// sample position is known before the first read — so this form
//   * reads the volume from a BRICKED copy (ca_brick_volume: 8 x 8 x 8 cells = 64 contiguous bytes, rebuilt per frame: one pass over
//     the state) — the samples a wave takes at one march index lie within a step length of each other along the ray and a pixel
//     apart across it: a handful of bricks instead of one cache line per lane;
//   * marches in batches of 8 samples: 8 positions, 8 addresses, 8 reads in flight, then the live samples are slab-tested in march
//     order (first hit wins) — a memory round trip per 8 samples instead of per sample, and the 45-instruction slab test runs for
//     the two or three live samples of a batch instead of inside every iteration;
//   * hands 16 x 16-pixel tiles to the XCDs in contiguous bands (an XCD's L2 then holds the eighth of the volume its rays cross);
//   * evaluates the jitter hash once per pixel (the shader calls it twice with the same arguments) and forms 1 / direction once per
//     march (ray_cube_inv: bit-identical to the shader's slab test).
// The arithmetic of every sample — positions by repeated addition of the step, floor, the u32 conversions, the slab test, shading,
// blend — is the first form's, operation for operation: tests/test_gpu_render.py compares the two frames for equality.
// Every grid the UI offers (multiples of 32, main_pathtraced.js:675-693): power-of-two grids index the bricks by shifts and wrap by masks
// (the modulo of :268-290); the others (96, 160, ..., 992: bricks of 8^3 divide every multiple of 32) by 24-bit multiply-adds and a
// compare (a coordinate >= G — only the sample on a + face and a reprojected point outside the volume — takes a real modulo).
#include <hip/hip_fp16.h>

#include "ca3d_internal.h"

namespace ca3d
{
namespace
{

#include "render_device.inc"

// packed state -> bricks. Brick (bx, by, bz) covers cells [8 bx, 8 bx + 8) x ...; its 16 words: word w = 2 (z & 7) + ((y & 7) >> 2),
// bit (x & 7) + 8 (y & 3). A workgroup takes one (by, bz): 8 planes x 8 rows of the state, full x extent — C = G / 32 words per row,
// 64 rows — into LDS with coalesced reads (a row is contiguous), then writes the 4 C bricks of that row of bricks, 16 bytes per lane,
// consecutive lanes consecutive addresses (a brick row is contiguous: 4 C x 64 bytes). The first form — one thread per brick word,
// four strided source reads each — ran at 1 TB/s (30 us at 512^3: a seventh of the literal frame); this one moves the same 2 x 16 MiB
// at the copy rate.
constexpr u32 kBrickMaxC = 64u; // rows of up to 64 words: grids up to 2048

__global__ __launch_bounds__(256) void ca_brick_volume(const u32 *__restrict__ cells, u32 *__restrict__ bricks, u32 lg)
{
	__shared__ u32 src[64u * kBrickMaxC]; // [z & 7][y & 7][word]
	const u32 lc = lg - 5u, C = 1u << lc, lnb = lg - 3u;
	const u32 by = blockIdx.x & ((1u << lnb) - 1u), bz = blockIdx.x >> lnb;
	const u32 words = 64u << lc; // of the slab
	for (u32 i = threadIdx.x; i < words; i += 256u)
	{
		const u32 xw = i & (C - 1u), r = i >> lc; // r = 8 z + y
		src[i] = cells[xw + ((((size_t)((bz << 3) + (r >> 3)) << lg) + ((by << 3) + (r & 7u))) << lc)];
	}
	__syncthreads();
	// output: bricks bx = 0 .. 4 C - 1 of this brick row, 16 words each; a lane writes 4 consecutive words (one uint4)
	uint4 *dst = reinterpret_cast<uint4 *>(bricks + ((size_t)blockIdx.x << (lc + 2u + 4u)));
	const u32 quads = (4u << lc) * 4u; // uint4 per brick row
	for (u32 q = threadIdx.x; q < quads; q += 256u)
	{
		const u32 bx = q >> 2, w0 = (q & 3u) << 2; // words w0 .. w0 + 3 of brick bx
		const u32 xw = bx >> 2, sh = (bx & 3u) << 3;
		u32 o[4];
#pragma unroll
		for (u32 k = 0; k < 4u; k++)
		{
			const u32 w = w0 + k, z = w >> 1, y0 = (w & 1u) << 2;
			u32 v = 0;
#pragma unroll
			for (u32 r = 0; r < 4u; r++) v |= ((src[(((z << 3) + y0 + r) << lc) + xw] >> sh) & 0xFFu) << (8u * r);
			o[k] = v;
		}
		dst[q] = make_uint4(o[0], o[1], o[2], o[3]);
	}
}

// the same for any G that is a multiple of 32 (rows of C = G / 32 words, G / 8 bricks per edge)
// Bricks per edge are padded to a power of two in the ADDRESS (brick (bx, by, bz) at ((bz << lnbp) + by << lnbp) + bx, lnbp = ceil log2 (G / 8)):
// the brick index is then shifts and the stream walks carry a cell as one packed word (render_stream.hip, kBricksPacked) on these grids
// too. The padding is address space, never written and never read for a cell of the grid (992^3: 128^3 instead of 124^3 bricks).
__global__ __launch_bounds__(256) void ca_brick_volume_any(const u32 *__restrict__ cells, u32 *__restrict__ bricks, u32 G, u32 lnbp)
{
	__shared__ u32 src[64u * kBrickMaxC]; // [z & 7][y & 7][word]
	const u32 C = G >> 5, nb = G >> 3;
	const u32 by = blockIdx.x % nb, bz = blockIdx.x / nb;
	const u32 words = 64u * C;
	for (u32 i = threadIdx.x; i < words; i += 256u)
	{
		const u32 r = i / C, xw = i - r * C; // r = 8 z + y
		src[i] = cells[xw + ((size_t)((bz << 3) + (r >> 3)) * G + ((by << 3) + (r & 7u))) * C];
	}
	__syncthreads();
	uint4 *dst = reinterpret_cast<uint4 *>(bricks + ((size_t)((bz << lnbp) + by) << (lnbp + 4u)));
	const u32 quads = nb * 4u;
	for (u32 q = threadIdx.x; q < quads; q += 256u)
	{
		const u32 bx = q >> 2, w0 = (q & 3u) << 2;
		const u32 xw = bx >> 2, sh = (bx & 3u) << 3;
		u32 o[4];
#pragma unroll
		for (u32 k = 0; k < 4u; k++)
		{
			const u32 w = w0 + k, z = w >> 1, y0 = (w & 1u) << 2;
			u32 v = 0;
#pragma unroll
			for (u32 r = 0; r < 4u; r++) v |= ((src[((z << 3) + y0 + r) * C + xw] >> sh) & 0xFFu) << (8u * r);
			o[k] = v;
		}
		dst[q] = make_uint4(o[0], o[1], o[2], o[3]);
	}
}

template <bool P2>
struct BrickVolume
{
	const u32 *bricks;
	u32 lg; // log2 G (P2); (!P2) bits of a brick coordinate in the address, ceil log2 (G / 8)
	u32 G;  // (!P2)
	// word index and bit of cell (x, y, z), every coordinate modulo the grid (:268-290)
	__device__ __forceinline__ u32 word_of(u32 x, u32 y, u32 z, u32 &bit) const
	{
		if (P2)
		{
			const u32 m = (1u << lg) - 1u, lnb = lg - 3u;
			x &= m; y &= m; z &= m;
			bit = (x & 7u) | ((y & 3u) << 3);
			return ((((((z >> 3) << lnb) + (y >> 3)) << lnb) + (x >> 3)) << 4) + ((z & 7u) << 1) + ((y & 7u) >> 2);
		}
		if (__builtin_expect(x >= G, 0)) x %= G;
		if (__builtin_expect(y >= G, 0)) y %= G;
		if (__builtin_expect(z >= G, 0)) z %= G;
		bit = (x & 7u) | ((y & 3u) << 3);
		return ((((((z >> 3) << lg) + (y >> 3)) << lg) + (x >> 3)) << 4) + ((z & 7u) << 1) + ((y & 7u) >> 2);
	}
	__device__ __forceinline__ u32 state(u32 x, u32 y, u32 z) const
	{
		u32 bit;
		const u32 w = word_of(x, y, z, bit);
		return (bricks[w] >> bit) & 1u;
	}
};

struct FrameBricks
{
	FrameParams F;
	const u32 *bricks;
	u32 lg, tiles_x, tiles;
};

constexpr int kBatch = 8;

// rayMarchDepth (:682-741, SHADOW false) and rayMarchShadow (:635-680, SHADOW true): samples at depth0, depth0 + step, ... (repeated
// addition, as the shader's `depth += step`) while depth < march; a live cell whose visible cube the ray meets ends the march.
// Returns true on such a hit (a_out: the cube's slab entry). `visits` counts the samples taken, the hit one included.
template <bool SHADOW, bool P2>
__device__ __forceinline__ bool march_batched(const RenderParams &P, const BrickVolume<P2> &vol, v3 from, v3 dir, float march, float step, float depth0, v3 vhalf,
                                              u32 ex, u32 ey, u32 ez, u32 &visits, float &a_out)
{
	const float cs = 1.0f / (float)P.G;
	const v3 inv = V(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z); // what the shader's slab test forms first, once per march
	float depth = depth0;
	int guard = 0;
	while (depth < march && guard < 100000)
	{
		// the batch's samples: cells, words in flight
		u32 word[kBatch], bitpos[kBatch];
		u32 nvalid = 0;
		float d = depth;
#pragma unroll
		for (int i = 0; i < kBatch; i++)
		{
			const bool valid = d < march && guard + i < 100000;
			nvalid += valid ? 1u : 0u;
			const v3 sp = from + dir * d;
			const u32 x = f2u(floorf(to_cells(P, sp.x))), y = f2u(floorf(to_cells(P, sp.y))), z = f2u(floorf(to_cells(P, sp.z)));
			u32 bit;
			const u32 w = vol.word_of(x, y, z, bit);
			word[i] = vol.bricks[w]; // (a sample past the march's end reads some word of the volume: in range, not used)
			bitpos[i] = valid ? bit : 32u;
			d += step;
		}
		u32 live = 0; // bit i: sample i is valid and its cell alive
#pragma unroll
		for (int i = 0; i < kBatch; i++) live |= (bitpos[i] < 32u ? (word[i] >> bitpos[i]) & 1u : 0u) << i;
		// live samples in march order
		while (live)
		{
			const int i = __ffs((int)live) - 1;
			live &= live - 1u;
			float di = depth;
			for (int j = 0; j < i; j++) di += step; // the i-th sample's depth, by the additions that led to it
			const v3 sp = from + dir * di;
			const v3 cc = V(floorf(to_cells(P, sp.x)), floorf(to_cells(P, sp.y)), floorf(to_cells(P, sp.z)));
			if (SHADOW && f2u(cc.x) == ex && f2u(cc.y) == ey && f2u(cc.z) == ez) continue; // any(cell != startCell) :664
			const v3 origin = V(cc.x * cs + cs * 0.5f - kHalf, cc.y * cs + cs * 0.5f - kHalf, cc.z * cs + cs * 0.5f - kHalf);
			float a, b;
			ray_cube_inv(from, inv, origin, vhalf, a, b);
			if (SHADOW ? (a <= b && a >= 0.0f) : (b >= 0.0f && a <= b))
			{
				visits += (u32)i + 1u;
				a_out = a;
				return true;
			}
		}
		visits += nvalid;
#pragma unroll
		for (int i = 0; i < kBatch; i++) depth += step;
		guard += kBatch;
	}
	return false;
}

template <bool P2>
__global__ __launch_bounds__(256, 4) void ca_render_frame_bricks(FrameBricks B)
{
	const FrameParams &F = B.F;
	const RenderParams &P = F.base;
	// tile order: consecutive workgroups go to consecutive XCDs; XCD k takes the k-th eighth of the tiles (rows of the image)
	const u32 per = (B.tiles + 7u) >> 3;
	const u32 tile = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
	if (tile >= B.tiles) return;
	const u32 px = (tile % B.tiles_x) * 16u + (threadIdx.x & 15u);
	const u32 py = (tile / B.tiles_x) * 16u + (threadIdx.x >> 4);
	if (px >= P.W || py >= P.H) return;
	const BrickVolume<P2> vol{B.bricks, B.lg, P.G};
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
	if (tn <= tf && tf >= 0.0f) // :822
	{
		const float rnd = n1rand(P, vu, vv); // both marches jitter by the same hash value (:692, :645)
		v3 enter = cam;
		const v3 exitp = cam + ray * tf;
		if (cam_dist >= 0.0f) enter = cam + ray * tn;
		// primary march :682-741
		v3 final_point = exitp;
		{
			const v3 seg = exitp - enter;
			const v3 dir = norm3(seg);
			const float march = len3(seg);
			const float step = march / u[U_DEPTHSAMPLES];
			float a = 0.0f;
			if (march_batched<false, P2>(P, vol, enter, dir, march, step, step * rnd + 0.01f, vhalf, 0, 0, 0, pvis, a)) final_point = enter + dir * a;
		}
		float uvx, uvy;
		reprojected_uv(P, final_point, uvx, uvy);
		float pdr = 0.0f;
		{
			size_t ti;
			if (F.prev_depth && texel_xy(P, uvx * u[U_WINDOW], uvy * u[U_WINDOW + 1], ti))
				pdr = __half2float(__ushort_as_half((unsigned short)(F.prev_depth[ti] & 0xFFFFu)));
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
			if (vol.state(rc.x, rc.y, rc.z) == 1u && cc.idx != rc.idx && pdr < current)
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
		const CellU cell = cell_u(P, p);
		{
			const u32 st = vol.state(cell.x, cell.y, cell.z);
			const float dist = sd_box(p - cell.origin, vhalf);
			if (st == 1u && !(dist > 0.001f))
			{
				const v3 light_pos = V(u[U_LIGHT], u[U_LIGHT + 1], u[U_LIGHT + 2]);
				const v3 ldir = norm3(light_pos - p);
				float vn, vf;
				ray_cube(p, ldir, V(0.0f, 0.0f, 0.0f), half, vn, vf);
				const v3 vexit = p + ldir * vf;
				// shadow march :635-680
				float occ = 1.0f;
				{
					const v3 seg = vexit - p;
					const v3 dir = norm3(seg);
					const float march = len3(seg);
					const float step = fmaxf(cs * u[U_CELLSIZE], march / u[U_SHADOWSAMPLES]);
					float a = 0.0f;
					if (march_batched<true, P2>(P, vol, p, dir, march, step, step * rnd + 0.0025f, vhalf, cell.x, cell.y, cell.z, svis, a)) occ = kOcclusion;
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
			const CellU rcell = cell_u(P, rpoint);
			const bool outside = uvx < 0.0f || uvx > 1.0f || uvy < 0.0f || uvy > 1.0f;
			if (outside || cell.idx != rcell.idx) { out[0] = col[0]; out[1] = col[1]; out[2] = col[2]; out[3] = 1.0f; }
			else
			{
				const float al = u[U_TEMPORALALPHA];
				const float cur[4] = {col[0], col[1], col[2], 1.0f};
				for (int k = 0; k < 4; k++) out[k] = fminf(fmaxf(pc[k] * (1.0f - al) + cur[k] * al, 0.0f), 1.0f);
			}
		}
	}
	{
		// light gizmo :866-874
		const v3 light_pos = V(u[U_LIGHT], u[U_LIGHT + 1], u[U_LIGHT + 2]);
		float ln, lf;
		ray_cube(cam, ray, light_pos, V(0.005f, 0.005f, 0.005f), ln, lf);
		if (ln <= lf && lf >= 0.0f && out[0] == 0.0f && out[1] == 0.0f && out[2] == 0.0f) { out[0] = out[1] = out[2] = out[3] = 1.0f; }
	}
	if (u[U_SHOWDEPTH] == 1.0f && vu < 0.5f) { out[0] = mixed_depth; out[1] = 0.0f; out[2] = 0.0f; out[3] = 1.0f; } // :880-883
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
	// statistics: a workgroup's sums, one atomic per counter and workgroup that has something to add
	if (P.counters)
	{
		__shared__ u32 sums[3];
		if (threadIdx.x < 3u) sums[threadIdx.x] = 0u;
		__syncthreads();
		u32 v[3] = {shadow, pvis, svis};
#pragma unroll
		for (int k = 0; k < 3; k++)
		{
			u32 x = v[k];
			for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
			if ((threadIdx.x & 63u) == 0u && x) atomicAdd(&sums[k], x);
		}
		__syncthreads();
		if (threadIdx.x < 3u && sums[threadIdx.x]) atomicAdd(&P.counters[threadIdx.x], (unsigned long long)sums[threadIdx.x]);
	}
}

} // namespace

static u32 brick_bits(uint32_t G) // bits of a brick coordinate in the address
{
	u32 l = 0;
	while ((1u << l) < (G >> 3)) l++;
	return l;
}
size_t frame_bricks_bytes(uint32_t G) { return (size_t)64u << (3u * brick_bits(G)); } // (a power-of-two grid: G^3 / 8)

bool frame_bricks_applies(uint32_t G) { return G >= 32u && G <= 2048u && (G & 31u) == 0u; } // (ca_brick_volume stages rows of up to 64 words)

// The bricked copy of a packed state: one pass, on `stream`.
hipError_t launch_brick_volume(const uint32_t *cells, uint32_t *bricks, uint32_t G, hipStream_t stream)
{
	const u32 nb = G >> 3;
	if ((G & (G - 1u)) == 0u)
	{
		u32 lg = 0;
		while ((1u << lg) < G) lg++;
		hipLaunchKernelGGL(ca_brick_volume, dim3(nb * nb), dim3(256), 0, stream, cells, bricks, lg);
	}
	else hipLaunchKernelGGL(ca_brick_volume_any, dim3(nb * nb), dim3(256), 0, stream, cells, bricks, G, brick_bits(G));
	return hipGetLastError();
}

// One literal frame over the bricked copy of `cells` (rebuilt here): `frame_params` is render.hip's FrameParams.
hipError_t launch_render_frame_bricks(const void *frame_params, uint32_t *bricks, bool bricks_valid, hipStream_t stream, bool *bricks_built)
{
	FrameBricks B;
	B.F = *static_cast<const FrameParams *>(frame_params);
	const RenderParams &P = B.F.base;
	B.bricks = bricks;
	B.lg = 0;
	while ((1u << B.lg) < P.G) B.lg++;
	if ((P.G & (P.G - 1u)) != 0u) B.lg = brick_bits(P.G); // (BrickVolume<false> reads it as the bits of a brick coordinate)
	if (!bricks_valid)
	{
		hipError_t eb = launch_brick_volume(P.cells, bricks, P.G, stream);
		if (eb != hipSuccess) return eb;
		if (bricks_built) *bricks_built = true;
	}
	B.tiles_x = (P.W + 15u) / 16u;
	B.tiles = B.tiles_x * ((P.H + 15u) / 16u);
	const u32 per = (B.tiles + 7u) >> 3;
	if ((P.G & (P.G - 1u)) == 0u) hipLaunchKernelGGL(ca_render_frame_bricks<true>, dim3(per * 8u), dim3(256), 0, stream, B);
	else hipLaunchKernelGGL(ca_render_frame_bricks<false>, dim3(per * 8u), dim3(256), 0, stream, B);
	return hipGetLastError();
}

} // namespace ca3d
