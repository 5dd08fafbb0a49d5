/*
 * render_oracle.c — CPU ORACLE for the volume renderer. TEST INFRASTRUCTURE ONLY (see ca_oracle.c header).
 *
 * float32 restatement of shaders/pathtraced_fragment_clustered.wgsl. The reference shades one jittered,
 * fixed-step ray per pixel per frame and converges through a temporal EMA; its jitter hash
 * (fract(sin(x) * 43758.5453), lines 172-180) is implementation-dependent, so a single reference frame is not
 * a defined target. The target is the frame the reference converges to under a static camera (SURVEY 8(a)
 * row R-par):
 *   - primary visibility = the first alive cell along the ray whose visible (shrunken) cube passes the
 *     reference's own slab test (rayMarchDepth, 682-741: `tFar >= 0 && tNear <= tFar`), hit = start + dir*tNear;
 *     an exact cell walk replaces the jittered fixed-step march, which only approximates it;
 *   - depth = |hit - camera| (estimateLikelyDepth, 743-798, only repairs over-stepping, which an exact walk
 *     does not do);
 *   - shading gate, Cook-Torrance BRDF and light/albedo model exactly as 379-427, 537-633;
 *   - shadow = first alive cell other than the start cell whose visible cube is hit with tNear >= 0 on the way
 *     to the volume exit, walking every cell from the reference's minimum start offset 0.0025 (635-680);
 *   - colour = clamp(c, 0, 1): the fixed point of `clamp(mix(prev, cur, alpha), 0, 1)` (429-471);
 *   - light gizmo, depth overlay, gamma and the three outputs as fragment_main 866-889.
 * Pinning: no reference frame can be produced offline (no WebGPU); the uniform block that drives this code IS
 * pinned by the values captured from the reference host (tests/golden/reference_host.json).
 *
 * All arithmetic is float; build with -ffp-contract=off so no FMA contraction differs from the HIP build.
 * ca3d_oracle_render_bruteforce is an independent visibility check for small grids (tests every alive cell).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float length3(v3 a) { return sqrtf(dot(a, a)); }
static inline v3 normalize3(v3 a) { float l = length3(a); return V(a.x / l, a.y / l, a.z / l); }
static inline float minf(float a, float b) { return a < b ? a : b; }
static inline float maxf(float a, float b) { return a > b ? a : b; }
static inline float clampf(float x, float lo, float hi) { return minf(maxf(x, lo), hi); }

/* uniform block, float indices (MemoryManager order; pathtraced_fragment_clustered.wgsl:17-34) */
enum { U_LIGHT = 0, U_VIEW = 4, U_PROJVIEWINV = 20, U_PREVVIEW = 36, U_PREVPROJVIEWINV = 52, U_WINDOW = 68,
       U_TIME = 70, U_DEPTHSAMPLES = 71, U_SHADOWSAMPLES = 72, U_CELLSIZE = 73, U_SHOWDEPTH = 74,
       U_TEMPORALALPHA = 75, U_REFLECTIVITY = 76, U_ROUGHNESS = 79, U_MATERIALCOLOR = 80, U_GAMMA = 83 };

static const float PI_F = 3.14159265359f;
#define HALF_CUBE_SIZE 0.5f
#define OCCLUSION_FACTOR 0.0095f

typedef struct
{
	const uint32_t *cells;
	uint32_t G;
	const float *u;
	float cot_half_fov;
	int legacy; /* one u32 per cell + shaders/pathtraced_fragment.wgsl shading */
	int indirect; /* add calculateIndirectLighting (:307-377; the reference leaves its call commented out at :424) */
} Ctx;

/* :268-290 (modulo wrap of every coordinate) */
static inline uint32_t cell_state(const Ctx *c, uint32_t x, uint32_t y, uint32_t z)
{
	const uint32_t G = c->G, cols = G / 32u;
	if (c->legacy) /* pathtraced_fragment.wgsl:157-168, 201: no wrap; out-of-range coordinates read as dead */
		return (x >= G || y >= G || z >= G) ? 0u : (c->cells[(size_t)x + (size_t)y * G + (size_t)z * G * G] == 1u ? 1u : 0u);
	const uint32_t idx = ((x / 32u) % cols) + (y % G) * cols + (z % G) * cols * G;
	return (c->cells[idx] >> (x % 32u)) & 1u;
}

/* :188-197 */
static v3 get_ray(const Ctx *c, float u, float v)
{
	const float r = c->u[U_WINDOW] / c->u[U_WINDOW + 1];
	float x = (u - 0.5f) * r, y = v - 0.5f;
	const float z = 0.5f * c->cot_half_fov;
	return normalize3(V(x, y, -z));
}

/* mat4x4f (column-major) * vec4(v, 0) */
static v3 mat_dir(const float *m, v3 v)
{
	return V(m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z, m[2] * v.x + m[6] * v.y + m[10] * v.z);
}

/* :212-225 */
static void ray_cube(v3 o, v3 d, v3 center, v3 half, float *tnear, float *tfar)
{
	const v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
	const v3 tmin = mul(sub(sub(center, half), o), inv);
	const v3 tmax = mul(sub(add(center, half), o), inv);
	const v3 t1 = V(minf(tmin.x, tmax.x), minf(tmin.y, tmax.y), minf(tmin.z, tmax.z));
	const v3 t2 = V(maxf(tmin.x, tmax.x), maxf(tmin.y, tmax.y), maxf(tmin.z, tmax.z));
	*tnear = maxf(maxf(t1.x, t1.y), t1.z);
	*tfar = minf(minf(t2.x, t2.y), t2.z);
}

/* :182-186 */
static float sd_box(v3 p, v3 b)
{
	const v3 q = V(fabsf(p.x) - b.x, fabsf(p.y) - b.y, fabsf(p.z) - b.z);
	const v3 m = V(maxf(q.x, 0.0f), maxf(q.y, 0.0f), maxf(q.z, 0.0f));
	return length3(m) + minf(maxf(q.x, maxf(q.y, q.z)), 0.0f);
}

/* :227-254 */
static v3 face_normal(v3 p, v3 origin)
{
	const v3 d = sub(p, origin);
	const v3 a = V(fabsf(d.x), fabsf(d.y), fabsf(d.z));
	const float m = maxf(maxf(a.x, a.y), a.z);
	v3 n;
	if (a.x == m) n = V(d.x, 0, 0);
	else if (a.y == m) n = V(0, d.y, 0);
	else n = V(0, 0, d.z);
	return normalize3(n);
}

typedef struct { v3 origin; int32_t cx, cy, cz; } Cell;

/* :292-304 — cellCoords = floor((p + 0.5) / cellSize) with cellSize = 1 / G */
static Cell cell_from_point(const Ctx *c, v3 p)
{
	const float cs = 1.0f / (float)c->G; /* FULL_CUBE_SIZE / uGridSize */
	const v3 f = V(floorf((p.x + HALF_CUBE_SIZE) / cs), floorf((p.y + HALF_CUBE_SIZE) / cs), floorf((p.z + HALF_CUBE_SIZE) / cs));
	Cell r;
	r.origin = V(f.x * cs + cs * 0.5f - HALF_CUBE_SIZE, f.y * cs + cs * 0.5f - HALF_CUBE_SIZE, f.z * cs + cs * 0.5f - HALF_CUBE_SIZE);
	r.cx = (int32_t)f.x; r.cy = (int32_t)f.y; r.cz = (int32_t)f.z;
	return r;
}

static v3 cell_origin(const Ctx *c, int32_t x, int32_t y, int32_t z)
{
	const float cs = 1.0f / (float)c->G;
	return V((float)x * cs + cs * 0.5f - HALF_CUBE_SIZE, (float)y * cs + cs * 0.5f - HALF_CUBE_SIZE, (float)z * cs + cs * 0.5f - HALF_CUBE_SIZE);
}

/* :537-592 */
static v3 surface_brdf(v3 L, v3 Vd, v3 N, float roughness, v3 albedo, v3 F0)
{
	const v3 H = normalize3(add(L, Vd));
	const v3 fL = V(albedo.x / PI_F, albedo.y / PI_F, albedo.z / PI_F);
	const float a2 = roughness * roughness;
	const float NoH = dot(N, H);
	const float NoH2 = NoH * NoH;
	const float f = NoH2 * (a2 - 1.0f) + 1.0f;
	const float D = a2 / (PI_F * f * f);
	const float n = roughness + 1.0f;
	const float k = (n * n) / 8.0f;
	const float NoV = maxf(0.0f, dot(N, Vd));
	const float gv = NoV / (NoV * (1.0f - k) + k);
	const float NoL = maxf(0.0f, dot(N, L));
	const float gl = NoL / (NoL * (1.0f - k) + k);
	const float Gm = gv * gl;
	const float p = powf(1.0f - dot(H, Vd), 5.0f);
	const v3 F = V(F0.x + (1.0f - F0.x) * p, F0.y + (1.0f - F0.y) * p, F0.z + (1.0f - F0.z) * p);
	const float denom = 4.0f * dot(Vd, N) * dot(L, N);
	return V(fL.x + (D * Gm * F.x) / denom, fL.y + (D * Gm * F.y) / denom, fL.z + (D * Gm * F.z) / denom);
}

/* :594-633 calculateLightingAt(samplePoint, cellOrigin, cellCoords, eyePos, incidentLight, incidentLightPos); the cell's x and
 * y coordinates enter as u32 (albedo = coords / G) */
static v3 lighting_from(const Ctx *c, v3 p, v3 origin, uint32_t cx, uint32_t cy, v3 eye, v3 incident, v3 lightPos)
{
	const float *u = c->u;
	const v3 N = face_normal(p, origin);
	const float Gf = (float)c->G;
	const float cxn = (float)cx / Gf, cyn = (float)cy / Gf;
	v3 albedo = V(cxn, cyn, 1.0f - cxn);
	if (u[U_MATERIALCOLOR] != 0.0f || u[U_MATERIALCOLOR + 1] != 0.0f || u[U_MATERIALCOLOR + 2] != 0.0f)
		albedo = V(u[U_MATERIALCOLOR], u[U_MATERIALCOLOR + 1], u[U_MATERIALCOLOR + 2]);
	const v3 Vd = normalize3(sub(eye, p));
	const v3 L = normalize3(sub(lightPos, p));
	const v3 F0 = V(u[U_REFLECTIVITY], u[U_REFLECTIVITY + 1], u[U_REFLECTIVITY + 2]);
	const v3 brdf = surface_brdf(L, Vd, N, u[U_ROUGHNESS], albedo, F0);
	const float LoN = dot(L, N);
	return V(maxf(0.0f, brdf.x * incident.x * LoN), maxf(0.0f, brdf.y * incident.y * LoN), maxf(0.0f, brdf.z * incident.z * LoN));
}

static v3 lighting_at(const Ctx *c, v3 p, Cell cell, v3 eye)
{
	const float *u = c->u;
	const float mag = u[U_LIGHT + 3];
	return lighting_from(c, p, cell.origin, (uint32_t)cell.cx, (uint32_t)cell.cy, eye, V(mag, mag, mag), V(u[U_LIGHT], u[U_LIGHT + 1], u[U_LIGHT + 2]));
}

/* Exact cell walk from `start` along unit `dir` over parametric range (t0, tmax): calls visit(cell) in
 * traversal order until it returns non-zero. Cells outside [0, G) are never visited. */
typedef int (*visit_fn)(void *ud, int32_t x, int32_t y, int32_t z);

static void walk_cells(const Ctx *c, v3 start, v3 dir, float t0, float tmax, visit_fn visit, void *ud)
{
	const int32_t G = (int32_t)c->G;
	const float cs = 1.0f / (float)c->G;
	const v3 p = add(start, scale(dir, t0));
	int32_t ix = (int32_t)floorf((p.x + HALF_CUBE_SIZE) / cs), iy = (int32_t)floorf((p.y + HALF_CUBE_SIZE) / cs), iz = (int32_t)floorf((p.z + HALF_CUBE_SIZE) / cs);
	if (ix < 0) ix = 0; if (ix >= G) ix = G - 1;
	if (iy < 0) iy = 0; if (iy >= G) iy = G - 1;
	if (iz < 0) iz = 0; if (iz >= G) iz = G - 1;
	const int32_t sx = dir.x > 0 ? 1 : -1, sy = dir.y > 0 ? 1 : -1, sz = dir.z > 0 ? 1 : -1;
	const float big = 3.0e38f;
	/* t of the next boundary crossing per axis, measured from `start` */
	float tx = dir.x != 0.0f ? (((float)(ix + (sx > 0 ? 1 : 0)) * cs - HALF_CUBE_SIZE) - start.x) / dir.x : big;
	float ty = dir.y != 0.0f ? (((float)(iy + (sy > 0 ? 1 : 0)) * cs - HALF_CUBE_SIZE) - start.y) / dir.y : big;
	float tz = dir.z != 0.0f ? (((float)(iz + (sz > 0 ? 1 : 0)) * cs - HALF_CUBE_SIZE) - start.z) / dir.z : big;
	const float dx = dir.x != 0.0f ? cs / fabsf(dir.x) : big, dy = dir.y != 0.0f ? cs / fabsf(dir.y) : big, dz = dir.z != 0.0f ? cs / fabsf(dir.z) : big;
	float t = t0;
	for (int32_t guard = 0; guard < 3 * G + 3; guard++)
	{
		if (t >= tmax) return;
		if (visit(ud, ix, iy, iz)) return;
		if (tx <= ty && tx <= tz) { t = tx; tx += dx; ix += sx; if (ix < 0 || ix >= G) return; }
		else if (ty <= tz) { t = ty; ty += dy; iy += sy; if (iy < 0 || iy >= G) return; }
		else { t = tz; tz += dz; iz += sz; if (iz < 0 || iz >= G) return; }
	}
}

typedef struct { const Ctx *c; v3 start, dir, half; int hit; float tnear; } PrimaryUD;

static int primary_visit(void *ud_, int32_t x, int32_t y, int32_t z)
{
	PrimaryUD *ud = (PrimaryUD *)ud_;
	if (!cell_state(ud->c, (uint32_t)x, (uint32_t)y, (uint32_t)z)) return 0;
	float tn, tf;
	ray_cube(ud->start, ud->dir, cell_origin(ud->c, x, y, z), ud->half, &tn, &tf);
	if (tf >= 0.0f && tn <= tf) { ud->hit = 1; ud->tnear = tn; return 1; } /* :722-729 */
	return 0;
}

typedef struct { const Ctx *c; v3 start, dir, half; int32_t sx, sy, sz; int occluded; } ShadowUD;

static int shadow_visit(void *ud_, int32_t x, int32_t y, int32_t z)
{
	ShadowUD *ud = (ShadowUD *)ud_;
	if (x == ud->sx && y == ud->sy && z == ud->sz) return 0; /* any(cell != startCell) :664 */
	if (!cell_state(ud->c, (uint32_t)x, (uint32_t)y, (uint32_t)z)) return 0;
	float tn, tf;
	ray_cube(ud->start, ud->dir, cell_origin(ud->c, x, y, z), ud->half, &tn, &tf);
	if (tn <= tf && tn >= 0.0f) { ud->occluded = 1; return 1; } /* :668 */
	return 0;
}

/* :307-377 calculateIndirectLighting: light reflected once off the four neighbour cells in the layer the surface normal
 * points into (layer tables :117-169). Each live neighbour is hit along the INTEGER offset direction (:357), lit from the
 * light source — with its own shadow test, here the exact walk like every shadow ray of the converged frame — and that
 * reflected light is then the incident light of the sample point, coming from the hit point. */
static v3 indirect_lighting(const Ctx *c, v3 p, Cell cell, v3 cam)
{
	static const int32_t layers[6][4][3] = {
	    { { -1, 1, 0 }, { -1, -1, 0 }, { -1, 0, 1 }, { -1, 0, -1 } }, /* left   (normal.x < 0) */
	    { { 1, 1, 0 }, { 1, -1, 0 }, { 1, 0, 1 }, { 1, 0, -1 } },     /* right  (normal.x > 0) */
	    { { -1, -1, 0 }, { 1, -1, 0 }, { 0, -1, 1 }, { 0, -1, -1 } }, /* bottom (normal.y < 0) */
	    { { -1, 1, 0 }, { 1, 1, 0 }, { 0, 1, 1 }, { 0, 1, -1 } },     /* top    (normal.y > 0) */
	    { { 0, 1, -1 }, { 0, -1, -1 }, { -1, 0, -1 }, { 1, 0, -1 } }, /* back   (normal.z < 0) */
	    { { 0, 1, 1 }, { 0, -1, 1 }, { -1, 0, 1 }, { 1, 0, 1 } } };   /* front  (normal.z > 0) */
	const float *u = c->u;
	const v3 N = face_normal(p, cell.origin);
	int layer;
	if (N.x < 0) layer = 0; else if (N.x > 0) layer = 1; else if (N.y < 0) layer = 2; else if (N.y > 0) layer = 3; else if (N.z < 0) layer = 4; else if (N.z > 0) layer = 5; else return V(0, 0, 0);
	const float cs = 1.0f / (float)c->G;
	const float vis = cs * u[U_CELLSIZE] * 0.5f;
	const v3 half = V(HALF_CUBE_SIZE, HALF_CUBE_SIZE, HALF_CUBE_SIZE);
	const v3 lightPos = V(u[U_LIGHT], u[U_LIGHT + 1], u[U_LIGHT + 2]);
	const float mag = u[U_LIGHT + 3];
	v3 sum = V(0, 0, 0);
	for (int i = 0; i < 4; i++)
	{
		const int32_t *o = layers[layer][i];
		const uint32_t nx = (uint32_t)(cell.cx + o[0]), ny = (uint32_t)(cell.cy + o[1]), nz = (uint32_t)(cell.cz + o[2]); /* vec3u(vec3i) :345 */
		if (!cell_state(c, nx, ny, nz)) continue;
		const v3 norigin = V((float)nx * cs + cs * 0.5f - HALF_CUBE_SIZE, (float)ny * cs + cs * 0.5f - HALF_CUBE_SIZE, (float)nz * cs + cs * 0.5f - HALF_CUBE_SIZE);
		const v3 ndir = V((float)o[0], (float)o[1], (float)o[2]);
		float tn, tf;
		ray_cube(p, ndir, norigin, V(vis, vis, vis), &tn, &tf);
		if (!(tn <= tf && tf >= 0.0f)) continue;
		const v3 np = add(p, scale(ndir, tn));
		const v3 ldir = normalize3(sub(lightPos, np));
		float vn, vf;
		ray_cube(np, ldir, V(0, 0, 0), half, &vn, &vf);
		const v3 sseg = sub(add(np, scale(ldir, vf)), np);
		ShadowUD su = { c, np, normalize3(sseg), V(vis, vis, vis), (int32_t)nx, (int32_t)ny, (int32_t)nz, 0 };
		walk_cells(c, np, su.dir, 0.0025f, length3(sseg), shadow_visit, &su);
		const float occ = su.occluded ? OCCLUSION_FACTOR : 1.0f;
		const v3 refl = scale(lighting_from(c, np, norigin, nx, ny, p, V(mag, mag, mag), lightPos), occ);
		sum = add(sum, lighting_from(c, p, cell.origin, (uint32_t)cell.cx, (uint32_t)cell.cy, cam, refl, np));
	}
	return sum;
}

typedef struct { float r, g, b, a, depth; int shadow_ray; } Sample;

static Sample shade_sample(const Ctx *c, float vu, float vv)
{
	const float *u = c->u;
	const float *view = u + U_VIEW;
	Sample s = { 0, 0, 0, 1, 0, 0 };
	const v3 cam = V(view[12], view[13], view[14]);
	const v3 ray = mat_dir(view, get_ray(c, vu, vv));
	const v3 half = V(HALF_CUBE_SIZE, HALF_CUBE_SIZE, HALF_CUBE_SIZE);
	float tn, tf;
	ray_cube(cam, ray, V(0, 0, 0), half, &tn, &tf);
	const float cam_dist = sd_box(cam, half);
	if (tn <= tf && tf >= 0.0f) /* :822 */
	{
		v3 enter = cam;
		const v3 exitp = add(cam, scale(ray, tf));
		if (cam_dist >= 0.0f) enter = add(cam, scale(ray, tn));
		/* rayMarchDepth(enter, exit): exact walk */
		const v3 seg = sub(exitp, enter);
		const v3 dir = normalize3(seg);
		const float depth_len = length3(seg);
		const float cs = 1.0f / (float)c->G;
		const float vis = cs * u[U_CELLSIZE] * 0.5f;
		PrimaryUD pu = { c, enter, dir, V(vis, vis, vis), 0, 0.0f };
		walk_cells(c, enter, dir, 0.0f, depth_len, primary_visit, &pu);
		const v3 final_point = pu.hit ? add(enter, scale(dir, pu.tnear)) : exitp;
		s.depth = length3(sub(final_point, cam)); /* :762, 774 */
		const v3 p = add(cam, scale(ray, s.depth)); /* moreAccurateSamplePoint :840 */
		/* calculateLightingAndOcclusionAt :379-427 */
		const Cell cell = cell_from_point(c, p);
		const uint32_t st = cell_state(c, (uint32_t)cell.cx, (uint32_t)cell.cy, (uint32_t)cell.cz);
		const float dist = sd_box(sub(p, cell.origin), V(vis, vis, vis));
		if (st == 1u && !(dist > 0.001f))
		{
			const v3 lightPos = V(u[U_LIGHT], u[U_LIGHT + 1], u[U_LIGHT + 2]);
			const v3 ldir = normalize3(sub(lightPos, p));
			float vn, vf;
			ray_cube(p, ldir, V(0, 0, 0), half, &vn, &vf);
			const v3 vexit = add(p, scale(ldir, vf));
			const v3 sseg = sub(vexit, p);
			const v3 sdir = normalize3(sseg);
			const float slen = length3(sseg);
			ShadowUD su = { c, p, sdir, V(vis, vis, vis), cell.cx, cell.cy, cell.cz, 0 };
			walk_cells(c, p, sdir, 0.0025f, slen, shadow_visit, &su);
			const float occ = su.occluded ? (c->legacy ? 0.095f : OCCLUSION_FACTOR) : 1.0f;
			if (c->legacy)
			{
				/* calculateLigtingAt, pathtraced_fragment.wgsl:338-365 */
				const v3 N = face_normal(p, cell.origin);
				const float Gf = (float)c->G;
				const float fx = floorf((p.x + HALF_CUBE_SIZE) / cs), fy = floorf((p.y + HALF_CUBE_SIZE) / cs);
				const v3 colr = V(fx / Gf, fy / Gf, 1.0f - fx / Gf);
				const v3 view_dir = normalize3(sub(p, cam));
				const float dl = length3(sub(lightPos, p)), dc = length3(sub(cam, p));
				const float fl = maxf(1.0f, powf(dl, 2.0f)), fc = maxf(1.0f, powf(dc, 2.0f));
				const float incident = u[U_LIGHT + 3] / fl;
				const v3 inc_dir = normalize3(sub(p, lightPos));
				const float ndi = dot(N, inc_dir);
				const v3 refl = V(inc_dir.x - 2.0f * ndi * N.x, inc_dir.y - 2.0f * ndi * N.y, inc_dir.z - 2.0f * ndi * N.z);
				const float reflected = incident * dot(refl, V(-view_dir.x, -view_dir.y, -view_dir.z));
				s.r = occ * ((colr.x * reflected + incident * colr.x) / fc);
				s.g = occ * ((colr.y * reflected + incident * colr.y) / fc);
				s.b = occ * ((colr.z * reflected + incident * colr.z) / fc);
				s.a = occ;
			}
			else
			{
				const v3 lit = lighting_at(c, p, cell, cam);
				s.r = occ * lit.x; s.g = occ * lit.y; s.b = occ * lit.z;
				if (c->indirect) /* the term of :424 */
				{
					const v3 ind = indirect_lighting(c, p, cell, cam);
					s.r += ind.x; s.g += ind.y; s.b += ind.z;
				}
			}
			s.shadow_ray = 1;
		}
		/* temporal limit of clamp(mix(prev, cur, alpha), 0, 1) :468 */
		s.r = clampf(s.r, 0.0f, 1.0f); s.g = clampf(s.g, 0.0f, 1.0f); s.b = clampf(s.b, 0.0f, 1.0f); s.a = clampf(s.a, 0.0f, 1.0f);
	}
	/* light gizmo :866-874 */
	{
		const v3 lightPos = V(u[U_LIGHT], u[U_LIGHT + 1], u[U_LIGHT + 2]);
		float ln, lf;
		ray_cube(cam, ray, lightPos, V(0.005f, 0.005f, 0.005f), &ln, &lf);
		if (ln <= lf && lf >= 0.0f && s.r == 0.0f && s.g == 0.0f && s.b == 0.0f) { s.r = s.g = s.b = 1.0f; s.a = 1.0f; }
	}
	if (u[U_SHOWDEPTH] == 1.0f && vu < 0.5f) { s.r = s.depth; s.g = 0; s.b = 0; s.a = 1; } /* :880-883 */
	return s;
}

static const float kSub4[4][2] = { { 0.25f, 0.25f }, { 0.75f, 0.25f }, { 0.25f, 0.75f }, { 0.75f, 0.75f } };

/* Outputs as float: light[W*H*4] (linear rgb, a), depth[W*H*2] (depth of sub-sample 0, 1), presentation[W*H*4]
 * (pow(rgb, 1/gamma), a). spp is 1 (pixel centre) or 4 (2x2 stratified). Returns shadow rays traced, or <0. */
static int64_t render_impl(const uint32_t *cells, uint32_t G, const float *uniforms, uint32_t W, uint32_t H, uint32_t spp,
                           float *light, float *depth, float *presentation, uint32_t y_begin, uint32_t y_end, int legacy, int indirect)
{
	if (!cells || !uniforms || G == 0 || (!legacy && (G % 32u)) || (spp != 1 && spp != 4)) return -1;
	Ctx c = { cells, G, uniforms, (float)(1.0 / tan(37.5 * 3.14159265359 / 180.0)), legacy, indirect };
	int64_t shadow = 0;
	if (y_end > H) y_end = H;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : shadow)
#endif
	for (uint32_t py = y_begin; py < y_end; py++)
		for (uint32_t px = 0; px < W; px++)
		{
			float r = 0, g = 0, b = 0, a = 0, d0 = 0;
			for (uint32_t k = 0; k < spp; k++)
			{
				const float ox = spp == 1 ? 0.5f : kSub4[k][0], oy = spp == 1 ? 0.5f : kSub4[k][1];
				const float vu = ((float)px + ox) / (float)W, vv = 1.0f - ((float)py + oy) / (float)H;
				const Sample s = shade_sample(&c, vu, vv);
				r += s.r; g += s.g; b += s.b; a += s.a;
				if (k == 0) d0 = s.depth;
				shadow += s.shadow_ray;
			}
			const float inv = 1.0f / (float)spp;
			r *= inv; g *= inv; b *= inv; a *= inv;
			const size_t i = (size_t)py * W + px;
			if (light) { light[4 * i] = r; light[4 * i + 1] = g; light[4 * i + 2] = b; light[4 * i + 3] = 1.0f; }
			if (depth) { depth[2 * i] = d0; depth[2 * i + 1] = 1.0f; }
			if (presentation)
			{
				const float ig = legacy ? 1.0f / 2.2f : 1.0f / uniforms[U_GAMMA]; /* legacy: constant 2.2 (:704) */
				presentation[4 * i] = powf(r, ig); presentation[4 * i + 1] = powf(g, ig); presentation[4 * i + 2] = powf(b, ig);
				presentation[4 * i + 3] = a;
			}
		}
	return shadow;
}

int64_t ca3d_oracle_render(const uint32_t *cells, uint32_t G, const float *uniforms, uint32_t W, uint32_t H, uint32_t spp,
                           float *light, float *depth, float *presentation, uint32_t y_begin, uint32_t y_end)
{
	return render_impl(cells, G, uniforms, W, H, spp, light, depth, presentation, y_begin, y_end, 0, 0);
}

/* the same frame with the reference's one-bounce neighbour lighting switched on (:307-377, call site :424) */
int64_t ca3d_oracle_render_indirect(const uint32_t *cells, uint32_t G, const float *uniforms, uint32_t W, uint32_t H, uint32_t spp,
                                    float *light, float *depth, float *presentation, uint32_t y_begin, uint32_t y_end)
{
	return render_impl(cells, G, uniforms, W, H, spp, light, depth, presentation, y_begin, y_end, 0, 1);
}

/* The legacy renderer (shaders/pathtraced_fragment.wgsl): same converged-frame definition over the one-u32-per-cell
 * volume with its own shading, OCCLUSION_FACTOR 0.095 and gamma 2.2. */
int64_t ca3d_oracle_render_legacy(const uint32_t *cells, uint32_t G, const float *uniforms, uint32_t W, uint32_t H, uint32_t spp,
                                  float *light, float *depth, float *presentation, uint32_t y_begin, uint32_t y_end)
{
	return render_impl(cells, G, uniforms, W, H, spp, light, depth, presentation, y_begin, y_end, 1, 0);
}

/* Independent visibility check for small grids: nearest passing visible cube among ALL alive cells. Returns the
 * depth (|hit - cam|) the exact walk must reproduce for the pixel-centre ray, or -1 when the ray misses the
 * volume; `hit_cell` gets the linear cell id or -1. */
float ca3d_oracle_primary_bruteforce(const uint32_t *cells, uint32_t G, const float *uniforms, uint32_t W, uint32_t H,
                                     uint32_t px, uint32_t py, int64_t *hit_cell)
{
	Ctx c = { cells, G, uniforms, (float)(1.0 / tan(37.5 * 3.14159265359 / 180.0)), 0, 0 };
	const float *view = uniforms + U_VIEW;
	const v3 cam = V(view[12], view[13], view[14]);
	const float vu = ((float)px + 0.5f) / (float)W, vv = 1.0f - ((float)py + 0.5f) / (float)H;
	const v3 ray = mat_dir(view, get_ray(&c, vu, vv));
	const v3 half = V(HALF_CUBE_SIZE, HALF_CUBE_SIZE, HALF_CUBE_SIZE);
	float tn, tf;
	ray_cube(cam, ray, V(0, 0, 0), half, &tn, &tf);
	*hit_cell = -1;
	if (!(tn <= tf && tf >= 0.0f)) return -1.0f;
	v3 enter = cam;
	const v3 exitp = add(cam, scale(ray, tf));
	if (sd_box(cam, half) >= 0.0f) enter = add(cam, scale(ray, tn));
	const v3 dir = normalize3(sub(exitp, enter));
	const float cs = 1.0f / (float)G, vis = cs * uniforms[U_CELLSIZE] * 0.5f;
	float best = 3.0e38f;
	for (uint32_t z = 0; z < G; z++)
		for (uint32_t y = 0; y < G; y++)
			for (uint32_t x = 0; x < G; x++)
			{
				if (!cell_state(&c, x, y, z)) continue;
				float a, b;
				ray_cube(enter, dir, cell_origin(&c, (int32_t)x, (int32_t)y, (int32_t)z), V(vis, vis, vis), &a, &b);
				if (b >= 0.0f && a <= b && a < best) { best = a; *hit_cell = (int64_t)x + (int64_t)y * G + (int64_t)z * G * G; }
			}
	const v3 fp = *hit_cell >= 0 ? add(enter, scale(dir, best)) : exitp;
	return length3(sub(fp, cam));
}

/* ================================================================================================ literal frame
 * ONE frame exactly as fragment_main (800-890) produces it: jittered fixed-step primary march (682-741), history
 * look-ups and depth repair (473-487, 743-798), jittered shadow march (635-680), temporal blend (429-471). The
 * jitter hash fract(sin(x) * 43758.5453) (172-180) is evaluated with sin in double precision and rounded to f32 —
 * one concrete choice for a function WGSL leaves implementation-defined. History texel fetches outside the
 * target, and NaN coordinates (first frame: previous matrices are zero), read (0,0,0,0), like a robust
 * textureLoad. prev_light / prev_depth hold what the previous frame wrote, already rounded to binary16.
 */
static inline float fractf_(float x) { return x - floorf(x); }

static float n1rand(const Ctx *c, float u, float v)
{
	const float t = 0.07f * fractf_(c->u[U_TIME]);
	const float d = (t + u) * 12.9898f + (t + v) * 78.233f; /* dot(n, vec2(12.9898, 78.233)) */
	const float s = (float)sin((double)d);
	return fractf_(s * 43758.5453f);
}

static inline uint32_t f2u(float f) { return !(f >= 0.0f) ? 0u : (f >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)f); }

static void mat_point(const float *m, v3 p, float out[4])
{
	for (int r = 0; r < 4; r++) out[r] = m[r] * p.x + m[4 + r] * p.y + m[8 + r] * p.z + m[12 + r];
}

/* :473-487 */
static void reprojected_uv(const Ctx *c, v3 p, float uv[2])
{
	float q[4];
	mat_point(c->u + U_PREVPROJVIEWINV, p, q);
	const float cx = q[0] / q[3], cy = q[1] / q[3];
	uv[0] = cx * 0.5f + 0.5f;
	uv[1] = -cy * 0.5f + 0.5f;
}

static void texel(const float *tex, int comps, uint32_t W, uint32_t H, float fx, float fy, float *out)
{
	/* vec2i(uv * windowSize): truncation; NaN and out-of-range coordinates fetch zeros */
	for (int k = 0; k < comps; k++) out[k] = 0.0f;
	if (!(fx == fx) || !(fy == fy) || !tex) return;
	if (fx <= -1.0f || fy <= -1.0f || fx >= (float)W || fy >= (float)H) return;
	const int x = (int)fx, y = (int)fy;
	if (x < 0 || y < 0 || x >= (int)W || y >= (int)H) return;
	for (int k = 0; k < comps; k++) out[k] = tex[((size_t)y * W + x) * comps + k];
}

typedef struct { v3 final_point, farthest; } MarchOut;

/* :682-741 */
static MarchOut ray_march_depth(const Ctx *c, v3 start, v3 end, float vu, float vv, float steps)
{
	MarchOut o;
	o.farthest = end;
	const v3 seg = sub(end, start);
	const v3 dir = normalize3(seg);
	const float march = length3(seg);
	const float step = march / steps;
	const float rnd = n1rand(c, vu, vv);
	float depth = step * rnd + 0.01f;
	const float cs = 1.0f / (float)c->G;
	const float vis = cs * c->u[U_CELLSIZE] * 0.5f;
	for (int guard = 0; depth < march && guard < 100000; guard++)
	{
		const v3 sp = add(start, scale(dir, depth));
		const v3 cc = V(floorf((sp.x + HALF_CUBE_SIZE) / cs), floorf((sp.y + HALF_CUBE_SIZE) / cs), floorf((sp.z + HALF_CUBE_SIZE) / cs));
		const v3 origin = V(cc.x * cs + cs * 0.5f - HALF_CUBE_SIZE, cc.y * cs + cs * 0.5f - HALF_CUBE_SIZE, cc.z * cs + cs * 0.5f - HALF_CUBE_SIZE);
		if (cell_state(c, f2u(cc.x), f2u(cc.y), f2u(cc.z)) != 0u)
		{
			float tn, tf;
			ray_cube(start, dir, origin, V(vis, vis, vis), &tn, &tf);
			if (tf >= 0.0f && tn <= tf) { o.final_point = add(start, scale(dir, tn)); return o; }
		}
		depth += step;
	}
	o.final_point = end;
	return o;
}

/* :635-680 */
static float ray_march_shadow(const Ctx *c, v3 start, v3 end, uint32_t sx, uint32_t sy, uint32_t sz, float rnd, float steps)
{
	const v3 seg = sub(end, start);
	const v3 dir = normalize3(seg);
	const float march = length3(seg);
	const float cs = 1.0f / (float)c->G;
	const float vis = cs * c->u[U_CELLSIZE] * 0.5f;
	const float step = maxf(cs * c->u[U_CELLSIZE], march / steps);
	float depth = step * rnd + 0.0025f;
	for (int guard = 0; depth < march && guard < 100000; guard++)
	{
		const v3 sp = add(start, scale(dir, depth));
		const v3 cc = V(floorf((sp.x + HALF_CUBE_SIZE) / cs), floorf((sp.y + HALF_CUBE_SIZE) / cs), floorf((sp.z + HALF_CUBE_SIZE) / cs));
		const uint32_t ux = f2u(cc.x), uy = f2u(cc.y), uz = f2u(cc.z);
		const uint32_t st = cell_state(c, ux, uy, uz);
		const v3 origin = V(cc.x * cs + cs * 0.5f - HALF_CUBE_SIZE, cc.y * cs + cs * 0.5f - HALF_CUBE_SIZE, cc.z * cs + cs * 0.5f - HALF_CUBE_SIZE);
		if ((ux != sx || uy != sy || uz != sz) && st == 1u)
		{
			float tn, tf;
			ray_cube(start, dir, origin, V(vis, vis, vis), &tn, &tf);
			if (tn <= tf && tn >= 0.0f) return OCCLUSION_FACTOR;
		}
		depth += step;
	}
	return 1.0f;
}

typedef struct { v3 origin; uint32_t x, y, z, idx; } CellU;

/* :292-304 with vec3u conversion and getCellIdx (258-266) */
static CellU cell_u(const Ctx *c, v3 p)
{
	const float cs = 1.0f / (float)c->G;
	const v3 f = V(floorf((p.x + HALF_CUBE_SIZE) / cs), floorf((p.y + HALF_CUBE_SIZE) / cs), floorf((p.z + HALF_CUBE_SIZE) / cs));
	CellU r;
	r.origin = V(f.x * cs + cs * 0.5f - HALF_CUBE_SIZE, f.y * cs + cs * 0.5f - HALF_CUBE_SIZE, f.z * cs + cs * 0.5f - HALF_CUBE_SIZE);
	r.x = f2u(f.x); r.y = f2u(f.y); r.z = f2u(f.z);
	r.idx = r.x + r.y * c->G + r.z * (uint32_t)((float)c->G * (float)c->G);
	return r;
}

/* `flags` (optional) records which branches this pixel took: 1 the view ray meets the volume (:822); 2 the depth repair of
 * estimateLikelyDepth replaced the marched depth (:779-786); 4 reprojected uv outside [0,1]^2 -> current sample unblended (:450-454);
 * 8 cell identity differs -> unblended (:456-459); 16 blended with the history (:469); 32 the pixel is lit (colour != 0 before the
 * blend). Tests use them to show that a moving-camera sequence really exercises every branch of R6 / R10. */
static void frame_pixel(const Ctx *c, uint32_t W, uint32_t H, uint32_t px, uint32_t py, const float *prev_light,
                        const float *prev_depth, float out_rgba[4], float *out_depth, uint8_t *flags)
{
	uint8_t fl = 0;
	const float *u = c->u;
	const float *view = u + U_VIEW;
	const float vu = ((float)px + 0.5f) / (float)W, vv = 1.0f - ((float)py + 0.5f) / (float)H;
	float out[4] = { 0, 0, 0, 1 };
	float mixed_depth = 0.0f;
	const v3 cam = V(view[12], view[13], view[14]);
	const v3 ray = mat_dir(view, get_ray(c, vu, vv));
	const v3 half = V(HALF_CUBE_SIZE, HALF_CUBE_SIZE, HALF_CUBE_SIZE);
	float tn, tf;
	ray_cube(cam, ray, V(0, 0, 0), half, &tn, &tf);
	const float cam_dist = sd_box(cam, half);
	if (tn <= tf && tf >= 0.0f)
	{
		fl |= 1;
		v3 enter = cam;
		const v3 exitp = add(cam, scale(ray, tf));
		if (cam_dist >= 0.0f) enter = add(cam, scale(ray, tn));
		const MarchOut mo = ray_march_depth(c, enter, exitp, vu, vv, u[U_DEPTHSAMPLES]);
		float uvr[2];
		reprojected_uv(c, mo.final_point, uvr);
		float pd[2], pdr[2];
		texel(prev_depth, 2, W, H, vu * u[U_WINDOW], (1.0f - vv) * u[U_WINDOW + 1], pd);
		texel(prev_depth, 2, W, H, uvr[0] * u[U_WINDOW], uvr[1] * u[U_WINDOW + 1], pdr);
		/* estimateLikelyDepth :743-798 */
		float likely;
		{
			const float *pview = u + U_PREVVIEW;
			const v3 pcam = V(pview[12], pview[13], pview[14]);
			const float current = length3(sub(mo.final_point, cam));
			const v3 view_ray = normalize3(ray);
			const v3 view_ray2 = normalize3(sub(mo.final_point, pcam));
			const v3 reproj_point = add(pcam, scale(view_ray2, pdr[0]));
			likely = current;
			const float cs = 1.0f / (float)c->G, vis = cs * u[U_CELLSIZE] * 0.5f;
			const CellU rc = cell_u(c, reproj_point), cc = cell_u(c, mo.final_point);
			if (cell_state(c, rc.x, rc.y, rc.z) == 1u && cc.idx != rc.idx && pdr[0] < current)
			{
				float a, b;
				ray_cube(cam, view_ray, rc.origin, V(vis, vis, vis), &a, &b);
				if (a <= b && a >= 0.0f) { likely = a; fl |= 2; }
			}
			(void)pd;
		}
		mixed_depth = likely;
		const v3 p = add(cam, scale(ray, mixed_depth));
		reprojected_uv(c, p, uvr);
		/* calculateLightingAndOcclusionAt :379-427 */
		float col[3] = { 0, 0, 0 };
		{
			const float cs = 1.0f / (float)c->G, vis = cs * u[U_CELLSIZE] * 0.5f;
			const CellU cell = cell_u(c, p);
			const uint32_t st = cell_state(c, cell.x, cell.y, cell.z);
			const float dist = sd_box(sub(p, cell.origin), V(vis, vis, vis));
			if (st == 1u && !(dist > 0.001f))
			{
				const v3 lightPos = V(u[U_LIGHT], u[U_LIGHT + 1], u[U_LIGHT + 2]);
				const v3 ldir = normalize3(sub(lightPos, p));
				const float rnd = n1rand(c, vu, vv);
				float vn, vf;
				ray_cube(p, ldir, V(0, 0, 0), half, &vn, &vf);
				const v3 vexit = add(p, scale(ldir, vf));
				const float occ = ray_march_shadow(c, p, vexit, cell.x, cell.y, cell.z, rnd, u[U_SHADOWSAMPLES]);
				Cell ci; ci.origin = cell.origin; ci.cx = (int32_t)cell.x; ci.cy = (int32_t)cell.y; ci.cz = (int32_t)cell.z;
				const v3 lit = lighting_at(c, p, ci, cam);
				col[0] = occ * lit.x; col[1] = occ * lit.y; col[2] = occ * lit.z;
				if (col[0] != 0.0f || col[1] != 0.0f || col[2] != 0.0f) fl |= 32;
			}
		}
		/* mixWithReprojectedColor :429-471 */
		{
			float pc[4];
			texel(prev_light, 4, W, H, uvr[0] * u[U_WINDOW], uvr[1] * u[U_WINDOW + 1], pc);
			const float *pview = u + U_PREVVIEW;
			const v3 pcam = V(pview[12], pview[13], pview[14]);
			const v3 rdir = normalize3(sub(p, pcam));
			const v3 rpoint = add(pcam, scale(rdir, pdr[0]));
			const CellU rcell = cell_u(c, rpoint), ccell = cell_u(c, p);
			const int outside = uvr[0] < 0.0f || uvr[0] > 1.0f || uvr[1] < 0.0f || uvr[1] > 1.0f;
			if (outside || ccell.idx != rcell.idx) { out[0] = col[0]; out[1] = col[1]; out[2] = col[2]; out[3] = 1.0f; fl |= outside ? 4 : 8; }
			else
			{
				fl |= 16;
				const float a = u[U_TEMPORALALPHA];
				const float cur[4] = { col[0], col[1], col[2], 1.0f };
				for (int k = 0; k < 4; k++) out[k] = clampf(pc[k] * (1.0f - a) + cur[k] * a, 0.0f, 1.0f); /* mix(x, y, a) = x*(1-a) + y*a */
			}
		}
	}
	{
		const v3 lightPos = V(u[U_LIGHT], u[U_LIGHT + 1], u[U_LIGHT + 2]);
		float ln, lf;
		ray_cube(cam, ray, lightPos, V(0.005f, 0.005f, 0.005f), &ln, &lf);
		if (ln <= lf && lf >= 0.0f && out[0] == 0.0f && out[1] == 0.0f && out[2] == 0.0f) { out[0] = out[1] = out[2] = out[3] = 1.0f; }
	}
	if (u[U_SHOWDEPTH] == 1.0f && vu < 0.5f) { out[0] = mixed_depth; out[1] = 0; out[2] = 0; out[3] = 1; }
	for (int k = 0; k < 4; k++) out_rgba[k] = out[k];
	*out_depth = mixed_depth;
	if (flags) *flags = fl;
}

int ca3d_oracle_render_frame_branches(const uint32_t *cells, uint32_t G, const float *uniforms, uint32_t W, uint32_t H,
                                      const float *prev_light, const float *prev_depth, float *light, float *depth, float *presentation,
                                      uint8_t *branches);

int ca3d_oracle_render_frame(const uint32_t *cells, uint32_t G, const float *uniforms, uint32_t W, uint32_t H,
                             const float *prev_light, const float *prev_depth, float *light, float *depth, float *presentation)
{
	return ca3d_oracle_render_frame_branches(cells, G, uniforms, W, H, prev_light, prev_depth, light, depth, presentation, NULL);
}

/* the same frame; branches[H * W] (optional) receives frame_pixel's flags */
int ca3d_oracle_render_frame_branches(const uint32_t *cells, uint32_t G, const float *uniforms, uint32_t W, uint32_t H,
                                      const float *prev_light, const float *prev_depth, float *light, float *depth, float *presentation,
                                      uint8_t *branches)
{
	if (!cells || !uniforms || G == 0 || (G % 32u)) return -1;
	Ctx c = { cells, G, uniforms, (float)(1.0 / tan(37.5 * 3.14159265359 / 180.0)), 0, 0 };
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
	for (uint32_t py = 0; py < H; py++)
		for (uint32_t px = 0; px < W; px++)
		{
			float o[4], d;
			const size_t i = (size_t)py * W + px;
			frame_pixel(&c, W, H, px, py, prev_light, prev_depth, o, &d, branches ? branches + i : NULL);
			if (light) { light[4 * i] = o[0]; light[4 * i + 1] = o[1]; light[4 * i + 2] = o[2]; light[4 * i + 3] = 1.0f; }
			if (depth) { depth[2 * i] = d; depth[2 * i + 1] = 1.0f; }
			if (presentation)
			{
				const float ig = 1.0f / uniforms[U_GAMMA];
				presentation[4 * i] = powf(o[0], ig); presentation[4 * i + 1] = powf(o[1], ig); presentation[4 * i + 2] = powf(o[2], ig);
				presentation[4 * i + 3] = o[3];
			}
		}
	return 0;
}
