// Rule canonicalisation: the reference's five rule buffers (main_pathtraced.js:1330-1369) -> what the HIP
// kernels consume. Offset lists become either a named neighbour-class union (fast kernels) or a packed
// offset-code list (generic kernel); each 27-slot LUT slice becomes a minimised OR-of-cubes program over the
// bit-sliced neighbour count (compute_clustered.wgsl:165-232 restated as boolean functions).
#include <algorithm>
#include <array>
#include <set>
#include <vector>

#include "ca3d_internal.h"

namespace ca3d
{

namespace
{

int bits_for(uint32_t max_count)
{
	int b = 0;
	while ((1u << b) <= max_count) b++;
	return b; // 0 for max_count == 0
}

struct Imp
{
	uint32_t val, care; // care bit set = literal present
	bool operator<(const Imp &o) const { return care != o.care ? care < o.care : val < o.val; }
	bool operator==(const Imp &o) const { return care == o.care && val == o.val; }
};

// Quine-McCluskey over <= 5 variables. `on` and `dc` are minterm masks (bit k = count k).
std::vector<Imp> qm_cover(uint32_t on, uint32_t dc, int nvars)
{
	std::vector<Imp> primes;
	if (on == 0) return primes;
	const uint32_t full = (nvars == 0) ? 0u : ((1u << nvars) - 1u);
	std::set<Imp> cur;
	for (uint32_t k = 0; k < (1u << nvars); k++)
		if (((on | dc) >> k) & 1u) cur.insert(Imp{k, full});
	while (!cur.empty())
	{
		std::set<Imp> next;
		std::set<Imp> used;
		std::vector<Imp> v(cur.begin(), cur.end());
		for (size_t i = 0; i < v.size(); i++)
			for (size_t j = i + 1; j < v.size(); j++)
			{
				if (v[i].care != v[j].care) continue;
				const uint32_t diff = (v[i].val ^ v[j].val) & v[i].care;
				if (diff == 0 || (diff & (diff - 1))) continue; // differ in exactly one cared bit
				next.insert(Imp{v[i].val & ~diff, v[i].care & ~diff});
				used.insert(v[i]);
				used.insert(v[j]);
			}
		for (const Imp &p : v)
			if (!used.count(p)) primes.push_back(p);
		cur.swap(next);
	}
	// greedy cover of the on-set (don't-cares need no cover)
	auto covers = [&](const Imp &p, uint32_t k) { return ((k ^ p.val) & p.care) == 0; };
	std::vector<Imp> chosen;
	uint32_t left = on;
	while (left)
	{
		int best = -1, best_n = -1, best_lits = 99;
		for (size_t i = 0; i < primes.size(); i++)
		{
			int n = 0;
			for (uint32_t k = 0; k < (1u << nvars); k++)
				if (((left >> k) & 1u) && covers(primes[i], k)) n++;
			const int lits = __builtin_popcount(primes[i].care);
			if (n > best_n || (n == best_n && lits < best_lits)) { best = (int)i; best_n = n; best_lits = lits; }
		}
		chosen.push_back(primes[best]);
		for (uint32_t k = 0; k < (1u << nvars); k++)
			if (covers(primes[best], k)) left &= ~(1u << k);
	}
	return chosen;
}

int cost_of(const std::vector<Imp> &c)
{
	int s = 0;
	for (const Imp &p : c) s += 1 + __builtin_popcount(p.care);
	return s;
}

} // namespace

void compile_rule_prog(uint32_t onset_mask, uint32_t max_count, RuleProg *out)
{
	const int nvars = bits_for(max_count);
	const uint32_t universe = (nvars == 0) ? 1u : ((nvars == 5) ? 0xFFFFFFFFu : ((1u << (1u << nvars)) - 1u));
	const uint32_t reachable = (max_count >= 31) ? 0xFFFFFFFFu : ((1u << (max_count + 1)) - 1u);
	const uint32_t on = onset_mask & reachable;
	const uint32_t off = reachable & ~on;
	const uint32_t dc = universe & ~reachable;
	std::vector<Imp> pos = qm_cover(on, dc, nvars);
	std::vector<Imp> neg = qm_cover(off, dc, nvars);
	// Prefer a cover the kernels can keep in registers (<= 2 cubes), then the cheaper one.
	auto key = [](const std::vector<Imp> &c) { return (c.size() > 2 ? 1000 : 0) + cost_of(c); };
	const bool use_neg = (off == 0) || (on != 0 && key(neg) < key(pos));
	const std::vector<Imp> &c = use_neg ? neg : pos;
	*out = RuleProg{};
	out->invert = use_neg ? 0xFFFFFFFFu : 0u;
	out->n = (uint32_t)std::min<size_t>(c.size(), kMaxCubes);
	for (uint32_t i = 0; i < out->n; i++) out->cubes[i] = (c[i].care & 31u) | ((c[i].val & c[i].care & 31u) << 8);
	// A 5-variable function needs at most 16 cubes (parity), so the clamp above never drops one.
}

static uint32_t class_set(MainKind k)
{
	// 27-bit set over codes c = (dx+1) + 3*(dy+1) + 9*(dz+1)
	uint32_t s = 0;
	for (int dz = -1; dz <= 1; dz++)
		for (int dy = -1; dy <= 1; dy++)
			for (int dx = -1; dx <= 1; dx++)
			{
				const int nz = (dx != 0) + (dy != 0) + (dz != 0);
				bool in = false;
				switch (k)
				{
				case MAIN_VN: in = nz == 1; break;
				case MAIN_VN2D: in = nz == 1 && dz == 0; break;
				case MAIN_MOORE: in = nz >= 1; break;
				case MAIN_MOORE2D: in = nz >= 1 && dz == 0; break;
				case MAIN_EDGES: in = nz == 2; break;
				case MAIN_CORNERS: in = nz == 3; break;
				default: break;
				}
				if (in) s |= 1u << ((dx + 1) + 3 * (dy + 1) + 9 * (dz + 1));
			}
	return s;
}

int canonicalize_rules(const int32_t *main_offs, uint32_t n_main, const int32_t *edge_offs, uint32_t n_edge,
                       const int32_t *corner_offs, uint32_t n_corner, const uint32_t *survive,
                       const uint32_t *born, CanonRules *out, std::string *err)
{
	const int32_t *lists[3] = {main_offs, edge_offs, corner_offs};
	const uint32_t ns[3] = {n_main, n_edge, n_corner};
	static const char *names[3] = {"main", "edges", "corners"};
	CanonRules r;
	if (!survive || !born) { *err = "survive/born LUT pointers must not be NULL"; return CA3D_ERR_INVALID_ARGUMENT; }
	uint32_t set_mask[3] = {0, 0, 0};
	bool simple[3] = {true, true, true}; // no duplicates, no (0,0,0)
	for (int s = 0; s < 3; s++)
	{
		if (ns[s] % 3u) { *err = std::string(names[s]) + " offsets: length must be a multiple of 3 (xyz triples)"; return CA3D_ERR_INVALID_ARGUMENT; }
		if (ns[s] && !lists[s]) { *err = std::string(names[s]) + " offsets: NULL pointer with non-zero length"; return CA3D_ERR_INVALID_ARGUMENT; }
		const uint32_t cnt = ns[s] / 3u;
		if (cnt > (uint32_t)kMaxOffsets) { *err = std::string(names[s]) + " offsets: more than 26 triples would index outside the rule-set's 27 LUT slots"; return CA3D_ERR_UNSUPPORTED; }
		r.lists.n[s] = cnt;
		for (uint32_t i = 0; i < cnt; i++)
		{
			const int32_t dx = lists[s][3 * i], dy = lists[s][3 * i + 1], dz = lists[s][3 * i + 2];
			if (dx < -1 || dx > 1 || dy < -1 || dy > 1 || dz < -1 || dz > 1)
			{
				*err = std::string(names[s]) + " offsets: components must be in {-1,0,1} (the reference's tables never leave the 3x3x3 shell)";
				return CA3D_ERR_UNSUPPORTED;
			}
			r.lists.code[s][i] = (uint8_t)((dx + 1) | ((dy + 1) << 2) | ((dz + 1) << 4));
			const uint32_t bit = 1u << ((dx + 1) + 3 * (dy + 1) + 9 * (dz + 1));
			if ((set_mask[s] & bit) || (dx == 0 && dy == 0 && dz == 0)) simple[s] = false;
			set_mask[s] |= bit;
		}
	}
	for (int i = 0; i < CA3D_LUT_LEN; i++) { r.survive_raw[i] = survive[i]; r.born_raw[i] = born[i]; }
	// packed: an entry is set iff == 1 (compute_clustered.wgsl:232)
	for (int s = 0; s < 3; s++)
	{
		uint32_t b = 0, v = 0;
		for (uint32_t k = 0; k < 27; k++)
		{
			if (born[k + 27 * s] == 1u) b |= 1u << k;
			if (survive[k + 27 * s] == 1u) v |= 1u << k;
		}
		r.onset_born[s] = b;
		r.onset_survive[s] = v;
		compile_rule_prog(b, r.lists.n[s], &r.prog.set[s].born);
		compile_rule_prog(v, r.lists.n[s], &r.prog.set[s].survive);
		r.need[s] = r.prog.set[s].born.n != 0 || r.prog.set[s].survive.n != 0;
	}
	// unpacked: `> 0` on slots 0..26 (compute.wgsl:160-166)
	for (uint32_t k = 0; k < 27; k++)
	{
		if (born[k] > 0u) r.unpacked_born |= 1u << k;
		if (survive[k] > 0u) r.unpacked_survive |= 1u << k;
	}
	compile_rule_prog(r.unpacked_born, r.lists.n[0], &r.unpacked_prog.born);
	compile_rule_prog(r.unpacked_survive, r.lists.n[0], &r.unpacked_prog.survive);
	// fast path: every list that matters is a plain set equal to a named class union
	r.main = MAIN_GENERIC;
	if (simple[0])
		for (int k = MAIN_VN; k <= MAIN_CORNERS; k++)
			if (set_mask[0] == class_set((MainKind)k)) r.main = (MainKind)k;
	const bool edges_ok = !r.need[1] || (simple[1] && set_mask[1] == class_set(MAIN_EDGES));
	const bool corners_ok = !r.need[2] || (simple[2] && set_mask[2] == class_set(MAIN_CORNERS));
	r.fast = r.main != MAIN_GENERIC && edges_ok && corners_ok;
	r.unpacked_fast = r.main != MAIN_GENERIC;
	r.valid = true;
	*out = r;
	return CA3D_OK;
}

} // namespace ca3d
