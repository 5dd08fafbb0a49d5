/*
 * ca_oracle.c — CPU ORACLE for the cellular-automaton step. TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it. The shipped library (cellularautomatons3d_amd/csrc) does not link, call or fall
 * back to anything in here.
 *
 * Pinning status: the reference holds no tests, golden vectors or fixtures for its WGSL kernels, and no WebGPU
 * implementation exists offline, so the kernels cannot be executed. The HOST-side surface (rule LUTs, offset
 * tables, packed layout, seeds, dispatch) is pinned by tests/golden/reference_host.json, produced by running
 * the reference's own JavaScript (tests/golden/capture_reference_host.js). The KERNEL semantics below are a
 * line-by-line restatement, cross-checked by analytic known answers (tests/test_oracle_kat.py) — for the kernel
 * itself parity is "restated, not executed": see DESIGN.md §Oracle.
 *
 * Three entry points:
 *   ca3d_oracle_packed_step_literal  — one invocation per output word, one loop iteration per bit, exactly as
 *                                      shaders/compute_clustered.wgsl:56-272 does it (slow; the anchor).
 *   ca3d_oracle_packed_step_fast     — word-parallel form of the same function (per-offset shifted row adds
 *                                      into bit-sliced counters); proven equal to the literal form by the CPU
 *                                      tests on random grids, then used at 256^3+ sizes and as cpu_baseline.
 *   ca3d_oracle_unpacked_step        — shaders/compute.wgsl:17-47, 49-53, 101, 160-174 (one u32 per cell).
 *
 * ca3d_oracle_packed_step_planes is the fast form on an array of z-planes with ghosts, so the multi-GPU Z-slab
 * decomposition can be checked on CPU.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CA3D_LUT_LEN 81
#define CA3D_SET_STRIDE 27

/* ------------------------------------------------------------------------------------------------ literal */

/* compute_clustered.wgsl:56-66 getClusterIdxFromGridCoordinates (u32 arithmetic, modulo wraps >= G to 0). */
static inline uint32_t lit_cluster_idx(uint32_t G, uint32_t x, uint32_t y, uint32_t z)
{
	const uint32_t cols = G / 32u;
	const uint32_t rows = G;
	const uint32_t depth = G;
	const uint32_t layer = cols * G;
	const uint32_t xw = x / 32u;
	return (xw % cols) + (y % rows) * cols + (z % depth) * layer;
}

/* compute_clustered.wgsl:79-86 getCellState: (word & masks[x % 32]) > 0. */
static inline uint32_t lit_cell_state(const uint32_t *in, uint32_t G, uint32_t x, uint32_t y, uint32_t z)
{
	const uint32_t w = in[lit_cluster_idx(G, x, y, z)];
	return (w & (1u << (x % 32u))) > 0u ? 1u : 0u;
}

/* compute_clustered.wgsl:88-111 (and the two copies at 115-138, 140-163): runtime-length xyz-triple list,
 * neighbour counted iff all(n >= 0) && all(n <= G)  — inclusive upper bound, line 104/131/156. */
static uint32_t lit_count(const uint32_t *in, uint32_t G, int32_t cx, int32_t cy, int32_t cz,
                          const int32_t *offs, uint32_t n_i32)
{
	uint32_t count = 0;
	const int32_t Gi = (int32_t)G;
	for (uint32_t i = 0; i + 2 < n_i32; i += 3)
	{
		const int32_t nx = cx + offs[i], ny = cy + offs[i + 1], nz = cz + offs[i + 2];
		if (nx >= 0 && ny >= 0 && nz >= 0 && nx <= Gi && ny <= Gi && nz <= Gi)
		{
			count += lit_cell_state(in, G, (uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
		}
	}
	return count;
}

/* compute_clustered.wgsl:165-190 + 208-211: stateLUT = [born, survive]; value = LUT[state][count + offset]. */
static inline uint32_t lit_next(uint32_t state, uint32_t count, uint32_t offset,
                                const uint32_t *born, const uint32_t *survive)
{
	const uint32_t *lut = state ? survive : born;
	uint32_t idx = count + offset;
	if (idx >= CA3D_LUT_LEN) idx = CA3D_LUT_LEN - 1; /* unreachable for offset lists of <= 26 entries */
	return lut[idx];
}

/* compute_clustered.wgsl:192-247 updateU32Cluster, dispatched over (G/32, G, G) invocations (267-272 and
 * main_pathtraced.js:1805-1806). */
int ca3d_oracle_packed_step_literal(uint32_t G, const uint32_t *in, uint32_t *out,
                                    const int32_t *main_offs, uint32_t n_main,
                                    const int32_t *edge_offs, uint32_t n_edge,
                                    const int32_t *corner_offs, uint32_t n_corner,
                                    const uint32_t *survive, const uint32_t *born)
{
	if (G == 0 || (G % 32u) != 0) return -1;
	const uint32_t C = G / 32u;
	for (uint32_t z = 0; z < G; z++)
		for (uint32_t y = 0; y < G; y++)
			for (uint32_t ix = 0; ix < C; ix++)
			{
				const uint32_t cluster = (ix % C) + (y % G) * C + (z % G) * C * G; /* :68-77 */
				uint32_t word = in[cluster];                                        /* :204 */
				for (uint32_t i = 0; i < 32; i++)
				{
					const uint32_t x = i + ix * 32u;
					const uint32_t cur = lit_cell_state(in, G, x, y, z);
					const uint32_t n0 = lit_count(in, G, (int32_t)x, (int32_t)y, (int32_t)z, main_offs, n_main);
					const uint32_t n1 = lit_count(in, G, (int32_t)x, (int32_t)y, (int32_t)z, edge_offs, n_edge);
					const uint32_t n2 = lit_count(in, G, (int32_t)x, (int32_t)y, (int32_t)z, corner_offs, n_corner);
					const uint32_t r0 = lit_next(cur, n0, 0, born, survive);
					const uint32_t r1 = lit_next(cur, n1, 27, born, survive);
					const uint32_t r2 = lit_next(cur, n2, 54, born, survive);
					const uint32_t alive = (r0 == 1u) || (r1 == 1u) || (r2 == 1u); /* any(v == vec3u(1)) :232 */
					const uint32_t m = 1u << i;
					if (alive) word |= m; else word &= ~m;                           /* :234-244 */
				}
				out[cluster] = word;                                                 /* :247 */
			}
	return 0;
}

/* ------------------------------------------------------------------------------------------------ fast */

/*
 * Word-parallel form. Equivalent padded-grid reading of lines 56-66 + 101-107:
 *   P(x,y,z) = 0 if any coordinate is -1, else S(x mod G, y mod G, z mod G).
 *
 * The state is given as an array of `nplanes` consecutive z-planes; plane j holds global z = (zbase + j) mod G
 * (zbase may be negative: ghost planes of a Z-slab). Output planes [lo, hi) are written into `out` (same
 * shape as `in`). z-neighbour rules, which reproduce P for any slab with ghosts:
 *   dz = -1: dead if the plane's own global z is 0 (z-1 == -1 is dropped), else plane j-1;
 *   dz = +1: plane j+1, or plane 0 when wrap_full (the whole grid in one array: z == G wraps to 0).
 * In a slab the ghost copy of global plane 0 that sits above plane G-1 therefore evolves with a dead plane
 * below it, exactly like the real plane 0 it mirrors.
 */
static inline void add_bit(uint32_t *p, uint32_t v)
{
	/* ripple-carry increment of a 5-plane bit-sliced counter by the 1-bit plane v */
	uint32_t c = v;
	for (int i = 0; i < 5; i++) { const uint32_t t = p[i] & c; p[i] ^= c; c = t; }
}

static inline int64_t mod_floor(int64_t a, int64_t m) { int64_t r = a % m; return r < 0 ? r + m : r; }

static int fast_row(uint32_t G, uint32_t C, const uint32_t *in, int64_t zbase, uint32_t nplanes, int wrap_full,
                    uint32_t j, uint32_t y,
                    const int32_t *const lists[3], const uint32_t nlist[3],
                    const uint32_t *survive, const uint32_t *born, uint32_t *out_row, uint32_t *cnt /* [3][5][C] */)
{
	memset(cnt, 0, sizeof(uint32_t) * 15u * C);
	const int z_is_zero = mod_floor(zbase + (int64_t)j, (int64_t)G) == 0;
	for (int s = 0; s < 3; s++)
	{
		const int32_t *offs = lists[s];
		for (uint32_t i = 0; i + 2 < nlist[s]; i += 3)
		{
			const int32_t dx = offs[i], dy = offs[i + 1], dz = offs[i + 2];
			if (dx < -1 || dx > 1 || dy < -1 || dy > 1 || dz < -1 || dz > 1) return -3;
			int32_t yy = (int32_t)y + dy;
			if (yy < 0) continue;
			if (yy == (int32_t)G) yy = 0;
			int64_t jj = (int64_t)j + dz;
			if (dz == -1 && z_is_zero) continue;
			if (dz == 1 && wrap_full && jj == (int64_t)nplanes) jj = 0;
			if (jj < 0 || jj >= (int64_t)nplanes) return -4; /* caller asked for a plane whose neighbour is absent */
			const uint32_t *row = in + ((size_t)jj * G + (size_t)yy) * C;
			for (uint32_t cx = 0; cx < C; cx++)
			{
				uint32_t v;
				if (dx == 0) v = row[cx];
				else if (dx == -1) v = (row[cx] << 1) | (cx > 0 ? row[cx - 1] >> 31 : 0u);
				else v = (row[cx] >> 1) | ((cx + 1 < C ? row[cx + 1] : row[0]) << 31);
				uint32_t pl[5];
				for (int k = 0; k < 5; k++) pl[k] = cnt[(s * 5 + k) * C + cx];
				add_bit(pl, v);
				for (int k = 0; k < 5; k++) cnt[(s * 5 + k) * C + cx] = pl[k];
			}
		}
	}
	const uint32_t *self = in + ((size_t)j * G + y) * C;
	for (uint32_t cx = 0; cx < C; cx++)
	{
		uint32_t B = 0, S = 0;
		for (int s = 0; s < 3; s++)
		{
			for (uint32_t k = 0; k < CA3D_SET_STRIDE; k++)
			{
				const int b = born[k + 27u * s] == 1u, v = survive[k + 27u * s] == 1u;
				if (!b && !v) continue;
				uint32_t eq = 0xFFFFFFFFu;
				for (int q = 0; q < 5; q++)
				{
					const uint32_t pl = cnt[(s * 5 + q) * C + cx];
					eq &= ((k >> q) & 1u) ? pl : ~pl;
				}
				if (b) B |= eq;
				if (v) S |= eq;
			}
		}
		const uint32_t a = self[cx];
		out_row[cx] = (a & S) | (~a & B);
	}
	return 0;
}

int ca3d_oracle_packed_step_planes(uint32_t G, const uint32_t *in, uint32_t *out,
                                   int64_t zbase, uint32_t nplanes, uint32_t lo, uint32_t hi, int wrap_full,
                                   const int32_t *main_offs, uint32_t n_main,
                                   const int32_t *edge_offs, uint32_t n_edge,
                                   const int32_t *corner_offs, uint32_t n_corner,
                                   const uint32_t *survive, const uint32_t *born, int nthreads)
{
	if (G == 0 || (G % 32u) != 0 || nplanes == 0 || lo > hi || hi > nplanes) return -1;
	if ((n_main % 3u) || (n_edge % 3u) || (n_corner % 3u)) return -1;
	if (wrap_full && !(zbase == 0 && nplanes == G)) return -2;
	const uint32_t C = G / 32u;
	const int32_t *lists[3] = { main_offs, edge_offs, corner_offs };
	const uint32_t nlist[3] = { n_main, n_edge, n_corner };
	int err = 0;
	(void)nthreads;
#ifdef _OPENMP
	if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
	{
		uint32_t *scratch = (uint32_t *)malloc(sizeof(uint32_t) * 15u * C);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
		for (int64_t r = (int64_t)lo * G; r < (int64_t)hi * G; r++)
		{
			const uint32_t j = (uint32_t)(r / G), y = (uint32_t)(r % G);
			int rc = fast_row(G, C, in, zbase, nplanes, wrap_full, j, y, lists, nlist, survive, born,
			                  out + ((size_t)j * G + y) * C, scratch);
			if (rc) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
				err = rc;
			}
		}
		free(scratch);
	}
	return err;
}

int ca3d_oracle_packed_step_fast(uint32_t G, const uint32_t *in, uint32_t *out,
                                 const int32_t *main_offs, uint32_t n_main,
                                 const int32_t *edge_offs, uint32_t n_edge,
                                 const int32_t *corner_offs, uint32_t n_corner,
                                 const uint32_t *survive, const uint32_t *born, int nthreads)
{
	return ca3d_oracle_packed_step_planes(G, in, out, 0, G, 0, G, 1, main_offs, n_main, edge_offs, n_edge,
	                                      corner_offs, n_corner, survive, born, nthreads);
}

/* n successive steps with the reference's ping-pong (main_pathtraced.js:1580-1609, 1800-1808): step k reads
 * buf[k % 2] and writes buf[(k+1) % 2]; both buffers start with the same data (1361-1362). Returns the index
 * of the buffer holding the final state. */
int ca3d_oracle_packed_run(uint32_t G, uint32_t *buf0, uint32_t *buf1, uint32_t steps, uint32_t first_step,
                           const int32_t *main_offs, uint32_t n_main,
                           const int32_t *edge_offs, uint32_t n_edge,
                           const int32_t *corner_offs, uint32_t n_corner,
                           const uint32_t *survive, const uint32_t *born, int nthreads)
{
	uint32_t *b[2] = { buf0, buf1 };
	uint32_t k = first_step;
	for (uint32_t s = 0; s < steps; s++, k++)
	{
		int rc = ca3d_oracle_packed_step_fast(G, b[k % 2], b[(k + 1) % 2], main_offs, n_main, edge_offs, n_edge,
		                                      corner_offs, n_corner, survive, born, nthreads);
		if (rc) return rc;
	}
	return (int)(k % 2);
}

/* ------------------------------------------------------------------------------------------------ unpacked */

/* compute.wgsl:17-28 getCellIdx on vec3u coordinates: u32 modulo; -1 arrives as 0xFFFFFFFF (line 42). */
static inline uint32_t leg_idx(uint32_t G, uint32_t x, uint32_t y, uint32_t z)
{
	const uint32_t layer = (uint32_t)((float)G * (float)G); /* u32(uGridSize.x * uGridSize.y): f32 product */
	return (x % G) + (y % G) * G + (z % G) * layer;
}

/* compute.wgsl:30-47, 49-53, 101, 160-174; dispatch ceil(G/4)^3 workgroups of 4x4x4 (G % 4 == 0 here). `in`
 * may be a slab with ghosts: plane index = z - zbase where zbase is the global z of in's first plane and the
 * caller guarantees every touched plane is present (full grid: zbase = 0, planes = G). */
int ca3d_oracle_unpacked_step(uint32_t G, const uint32_t *in, uint32_t *out,
                              const int32_t *offs, uint32_t n_offs,
                              const uint32_t *survive, uint32_t n_survive,
                              const uint32_t *born, uint32_t n_born, int nthreads)
{
	if (G == 0) return -1;
	(void)nthreads;
#ifdef _OPENMP
	if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static) collapse(2)
#endif
	for (uint32_t z = 0; z < G; z++)
		for (uint32_t y = 0; y < G; y++)
			for (uint32_t x = 0; x < G; x++)
			{
				const uint32_t idx = leg_idx(G, x, y, z);
				uint32_t count = 0;
				for (uint32_t i = 0; i + 2 < n_offs; i += 3)
				{
					const uint32_t nx = (uint32_t)((int32_t)x + offs[i]);
					const uint32_t ny = (uint32_t)((int32_t)y + offs[i + 1]);
					const uint32_t nz = (uint32_t)((int32_t)z + offs[i + 2]);
					count += in[leg_idx(G, nx, ny, nz)];
				}
				const uint32_t st = in[idx];
				const uint32_t sv = survive[count < n_survive ? count : n_survive - 1];
				const uint32_t bv = born[count < n_born ? count : n_born - 1];
				uint32_t o;
				if (st == 1u && sv > 0u) o = 1u;
				else if (st == 0u && bv > 0u) o = 1u;
				else o = 0u;
				out[idx] = o;
			}
	return 0;
}

/* compute.wgsl on an array of z-planes with ghosts (power-of-two G only, where the kernel is a true torus):
 * the z-neighbour of local plane j is j+dz, physically adjacent; x and y wrap as in the full grid. */
int ca3d_oracle_unpacked_step_planes(uint32_t G, const uint32_t *in, uint32_t *out, uint32_t nplanes,
                                     uint32_t lo, uint32_t hi, const int32_t *offs, uint32_t n_offs,
                                     const uint32_t *survive, uint32_t n_survive,
                                     const uint32_t *born, uint32_t n_born)
{
	if (G == 0 || (G & (G - 1u)) || lo > hi || hi > nplanes) return -1;
	const size_t plane = (size_t)G * G;
	for (uint32_t j = lo; j < hi; j++)
		for (uint32_t y = 0; y < G; y++)
			for (uint32_t x = 0; x < G; x++)
			{
				uint32_t count = 0;
				for (uint32_t i = 0; i + 2 < n_offs; i += 3)
				{
					const uint32_t nx = (uint32_t)((int32_t)x + offs[i]) % G;
					const uint32_t ny = (uint32_t)((int32_t)y + offs[i + 1]) % G;
					const int64_t nj = (int64_t)j + offs[i + 2];
					if (nj < 0 || nj >= (int64_t)nplanes) return -4;
					count += in[(size_t)nj * plane + (size_t)ny * G + nx];
				}
				const size_t idx = (size_t)j * plane + (size_t)y * G + x;
				const uint32_t st = in[idx];
				const uint32_t sv = survive[count < n_survive ? count : n_survive - 1];
				const uint32_t bv = born[count < n_born ? count : n_born - 1];
				out[idx] = (st == 1u && sv > 0u) ? 1u : ((st == 0u && bv > 0u) ? 1u : 0u);
			}
	return 0;
}

/* ------------------------------------------------------------------------------------------------ helpers */

uint32_t ca3d_oracle_fnv1a32(const uint8_t *bytes, size_t n)
{
	uint32_t h = 2166136261u;
	for (size_t i = 0; i < n; i++) { h ^= bytes[i]; h *= 16777619u; }
	return h;
}

uint64_t ca3d_oracle_popcount(const uint32_t *words, size_t n)
{
	uint64_t c = 0;
	for (size_t i = 0; i < n; i++) c += (uint64_t)__builtin_popcount(words[i]);
	return c;
}

/* Counter-based fill used by tests and bench for reproducible synthetic grids:
 * word[i] = mix32(seed, i); `and_rounds` extra hashed words are AND-ed in to thin the density to 2^-(1+r). */
static inline uint32_t mix32(uint32_t seed, uint32_t i, uint32_t round)
{
	uint32_t x = i * 0x9E3779B9u + seed + round * 0x85EBCA6Bu;
	x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
	return x;
}

void ca3d_oracle_fill(uint32_t *words, size_t n, uint32_t seed, uint32_t and_rounds)
{
	for (size_t i = 0; i < n; i++)
	{
		uint32_t w = mix32(seed, (uint32_t)i, 0);
		for (uint32_t r = 1; r <= and_rounds; r++) w &= mix32(seed, (uint32_t)i, r);
		words[i] = w;
	}
}
