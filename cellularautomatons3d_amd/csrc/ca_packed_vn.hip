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
//     3-input truth table in ONE instruction but takes the table as an immediate: 3 VALU per 32 cells for the whole
//     rule instead of ~22 through the cube programs of ca_bitslice.inc. The tables are compile-time constants in the
//     kernels ca_jit.cpp builds per rule at run time (and in the pre-built specialisation of the start-up rule);
//     the fallback carries all 256 tables behind a wave-uniform jump.
// This file: the ahead-of-time instantiations and the launcher. The device code is ca_packed_vn_kernel.inc.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ca3d_internal.h"

namespace ca3d
{
namespace
{
#include "ca_bitops.inc"

#include "ca_packed_vn_kernel.inc"

// Tables with a pre-built specialisation: the reference UI's start-up rule, von Neumann B1,3 / S0-6
// (survive slots 0..6, born slots 1 and 3 of the main rule-set).
constexpr int kDefaultS = 0xFF, kDefaultB = 0x0A; // canonical form: see launch_packed_vn

template <int CVL, int ZR>
hipError_t launch(const PackedLaunch &l, VnArgs a, hipStream_t stream)
{
	a.runs1 = (l.pr.hi - l.pr.lo + ZR - 1u) / ZR;
	constexpr u32 CV = 1u << CVL, TPP = 128u * CV * CV / 256u;
	const u32 blocks = TPP * (a.runs1 + (l.pr.hi2 > l.pr.lo2 ? (l.pr.hi2 - l.pr.lo2 + ZR - 1u) / ZR : 0u));
	// z-fastest block order (ca_packed_vn_kernel.inc; one output range, runs divisible by the XCDs), CA3D_VN_ORDER=1. Measured at
	// 2048^3: the fabric reads of a step fall from 1.29 x to 1.01 x the state (FETCH_SIZE) and the step is no faster (419 vs 414 us,
	// same box), so it stays off: what bounds that grid is not the re-read of neighbour planes.
	static const int order_env = getenv("CA3D_VN_ORDER") ? atoi(getenv("CA3D_VN_ORDER")) : 0;
	const bool one_range = !(l.pr.hi2 > l.pr.lo2) && (l.pr.hi - l.pr.lo) % ZR == 0;
	a.zfast = one_range && a.runs1 % 8u == 0 && order_env == 1 ? a.runs1 / 8u : 0u;
	const VnJit *jit = l.vn_jit;
	if (a.lut_s == (u32)kDefaultS && a.lut_b == (u32)kDefaultB)
		hipLaunchKernelGGL((ca_packed_vn<CVL, ZR, kDefaultS, kDefaultB>), dim3(blocks), dim3(256), 0, stream, l.in, l.out, a);
	else if (void *fn = ZR == 1 ? (jit ? jit->zr1 : nullptr) : ZR == 2 ? (jit ? jit->zr2 : nullptr) : ZR == 4 ? (jit ? jit->zr4 : nullptr) : (jit ? jit->zr8 : nullptr);
	         fn && jit->cvl == CVL && jit->lut_s == a.lut_s && jit->lut_b == a.lut_b)
	{
		// the run-time compiled specialisation for exactly these tables (ca_jit.cpp): same code shape as the branch above
		const u32 *in = l.in;
		u32 *out = l.out;
		void *args[] = {(void *)&in, (void *)&out, (void *)&a};
		return hipModuleLaunchKernel((hipFunction_t)fn, blocks, 1, 1, 256, 1, 1, 0, stream, args, nullptr);
	}
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

// The rule reduces to two truth tables over the von Neumann count ...
bool vn_rule_applies(const CanonRules &r, int variant)
{
	if (variant == 1 || !r.fast || r.main != MAIN_VN) return false;
	for (int s = 1; s < 3; s++)
	{
		const uint32_t reachable = (2u << r.lists.n[s]) - 1u;
		if ((r.onset_born[s] | r.onset_survive[s]) & reachable) return false;
	}
	return true;
}

// ... and the grid is a power of two in [256, 8192].
bool vn_kernel_applies(const CanonRules &r, uint32_t G, int variant)
{
	if (!vn_rule_applies(r, variant)) return false;
	const int cvl = log2_exact(G / 128u);
	return G % 128u == 0 && cvl >= 1 && cvl <= 6;
}

void vn_tables(const CanonRules &r, uint32_t *lut_s, uint32_t *lut_b)
{
	// Entry 7 of a table is never read (a cell has at most 6 von Neumann neighbours): give it the value the other odd entries share,
	// when they share one — a table whose seven entries agree becomes constant (the specialised kernels drop the evaluation), and one that
	// answers every odd count alike stays so with the entry in (vn_next drops the carry out of the count's plane 0 then: ca_bitops.inc,
	// vn_same_for_parity).
	*lut_s = r.onset_survive[0] & 0x7Fu;
	*lut_b = r.onset_born[0] & 0x7Fu;
	if ((*lut_s & 0x2Au) == 0x2Au) *lut_s |= 0x80u;
	if ((*lut_b & 0x2Au) == 0x2Au) *lut_b |= 0x80u;
}

bool vn_tables_prebuilt(uint32_t lut_s, uint32_t lut_b) { return lut_s == (u32)kDefaultS && lut_b == (u32)kDefaultB; }

int vn_grid_log2(uint32_t G) { return log2_exact(G / 128u); }

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
	vn_tables(r, &a.lut_s, &a.lut_b);
	// Non-temporal stores pay only while both ping-pong buffers sit in the 256 MiB Infinity Cache with room to
	// spare (measured: 6.9 vs 7.5 us per step at 512^3, 58 vs 43 us at 1024^3).
	a.nt = (size_t)l.pr.nplanes * G * (G / 32u) * sizeof(u32) <= (16u << 20) ? 1u : 0u;
	// 2 planes per thread once that still fills the chip (>= 1024 workgroups), else 1; 4 planes per thread on grids past the
	// Infinity Cache (2048^3 and up: 412 vs 433 us per step — fewer re-reads of the neighbour planes through L2; at 512^3 the deeper
	// run is slower, 6.7 vs 5.8 us; non-temporal stores change nothing there: 430 vs 433). CA3D_VN_ZR=2 keeps 2.
	static const int zr_env = getenv("CA3D_VN_ZR") ? atoi(getenv("CA3D_VN_ZR")) : 0;
	const u32 tpp = G / 128u * G / 256u;
	const bool deep = shortest >= 2u && (size_t)tpp * ((planes + 1u) / 2u) >= 1024u;
	const bool deep4 = zr_env != 2 && cvl >= 4 && shortest >= 4u && (size_t)tpp * ((planes + 3u) / 4u) >= 4096u;
	// 8 planes per thread on 2048^3 and up (round 5): 10 plane streams per thread for 8 planes instead of 6 for 4 — 381 against 411 us per
	// step at 2048^3, three processes each, alternating (0.70 against 0.65 of the HBM figure; 26 x 4 registers of rows: four waves per
	// SIMD, which a stream of this size does not miss); 16 planes per thread (two waves per SIMD): 404-417. Different PROCESSES differ
	// by +-5 % at this size (391 and 410 us were measured for the same code in one sweep): compare alternating runs, not single ones.
	// CA3D_VN_ZR=4 / 2 keep the shallower runs. (A table pair without a run-time compiled zr8 entry: 4 planes.)
	const bool prebuilt = a.lut_s == (u32)kDefaultS && a.lut_b == (u32)kDefaultB;
	const bool has8 = prebuilt || (l.vn_jit && l.vn_jit->zr8 && l.vn_jit->cvl == cvl && l.vn_jit->lut_s == a.lut_s && l.vn_jit->lut_b == a.lut_b);
	const bool deep8 = zr_env != 2 && zr_env != 4 && cvl >= 4 && has8 && shortest >= 8u && (size_t)tpp * ((planes + 7u) / 8u) >= 4096u;
	if (deep8) return cvl == 4 ? launch<4, 8>(l, a, stream) : cvl == 5 ? launch<5, 8>(l, a, stream) : launch<6, 8>(l, a, stream);
	switch (cvl)
	{
	case 1: return launch<1, 1>(l, a, stream);
	case 2: return deep ? launch<2, 2>(l, a, stream) : launch<2, 1>(l, a, stream);
	case 3: return deep ? launch<3, 2>(l, a, stream) : launch<3, 1>(l, a, stream);
	case 4: return deep4 ? launch<4, 4>(l, a, stream) : deep ? launch<4, 2>(l, a, stream) : launch<4, 1>(l, a, stream);
	case 5: return deep4 ? launch<5, 4>(l, a, stream) : deep ? launch<5, 2>(l, a, stream) : launch<5, 1>(l, a, stream);
	case 6: return deep4 ? launch<6, 4>(l, a, stream) : deep ? launch<6, 2>(l, a, stream) : launch<6, 1>(l, a, stream);
	default: return hipErrorInvalidValue;
	}
}

} // namespace ca3d
