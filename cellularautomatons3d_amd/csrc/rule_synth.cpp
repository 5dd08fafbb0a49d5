// Logic synthesis for the run-time compiled class kernels: turns the six truth tables of a rule (survive / born of the
// main, edges and corners rule-sets over their count bit-planes: compute_clustered.wgsl:165-232) into a short program of
// v_bitop3_b32 operations, emitted as device source for hiprtc.
//
// The kernels' generic form evaluates every table by Shannon expansion on its upper planes (1 / 3 / 7 ops for 3 / 4 / 5
// planes), ORs the three rule-sets and selects by the cell's state: ~25 ops per 32 cells for a rule like the bench's
// clustered one. Rules in practice are unions of a few count intervals, and counts a neighbourhood cannot reach are
// don't-cares, so most tables are a 3-input function of the low planes gated by the high ones. The synthesiser looks
// for exactly that, per table, in this order:
//   constant -> (dropped or folded);  <= 3 planes of support -> one op;
//   f = h(g(three planes), remaining one or two planes) -> two ops (functional decomposition: the 8 columns of the bound
//   set must fall into two compatible classes);  otherwise Shannon on the best plane, recursively.
// The OR of the rule-sets is folded into the last op of a term whenever that op has a free input, and the final
// "alive ? survive : born" is one more op. Every program is verified here against its tables on all reachable counts
// before it is used; on any mismatch the caller keeps the generic evaluation.
#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#include "ca3d_internal.h"

namespace ca3d
{
namespace
{

// A signal: a count plane ("mn[2]"), a temporary ("t7"), a constant, or the running OR accumulator.
struct Sig
{
	std::string name; // C expression
	int var = -1;     // index into the evaluation vector
};

struct Op
{
	int dst;        // evaluation slot written
	int in[3];      // evaluation slots read (-1: constant 0)
	unsigned imm;   // truth table: bit (a << 2 | b << 1 | c)
};

struct Builder
{
	std::vector<std::string> names; // evaluation slot -> C expression
	std::vector<Op> ops;
	int zero_slot = -1;

	int input(const std::string &expr) { names.push_back(expr); return (int)names.size() - 1; }
	int temp() { names.push_back("t" + std::to_string(names.size())); return (int)names.size() - 1; }
	int emit(unsigned imm, int a, int b, int c)
	{
		const int d = temp();
		ops.push_back(Op{d, {a, b, c}, imm & 0xFFu});
		return d;
	}
};

// A boolean function of n <= 5 variables given as value / care masks over 2^n assignments (bit i = assignment i, variable k
// is bit k of i), together with the evaluation slots of its variables.
struct Fn
{
	int n;
	unsigned val, care;
	int slot[5];
};

unsigned full_mask(int n) { return n >= 5 ? 0xFFFFFFFFu : ((1u << (1u << n)) - 1u); }

bool is_const(const Fn &f, int *value)
{
	const unsigned c = f.care & full_mask(f.n);
	if ((f.val & c) == 0) { *value = 0; return true; }
	if ((f.val & c) == c) { *value = 1; return true; }
	return false;
}

// Can variable k be dropped (the cared entries never disagree across it)? If so return the reduced function.
bool drop_var(const Fn &f, int k, Fn *out)
{
	Fn g;
	g.n = f.n - 1;
	g.val = g.care = 0;
	int s = 0;
	for (int i = 0; i < f.n; i++)
		if (i != k) g.slot[s++] = f.slot[i];
	for (unsigned i = 0; i < (1u << g.n); i++)
	{
		const unsigned lo = i & ((1u << k) - 1u), hi = (i >> k) << (k + 1);
		const unsigned i0 = hi | lo, i1 = i0 | (1u << k);
		const bool c0 = f.care >> i0 & 1u, c1 = f.care >> i1 & 1u;
		const unsigned v0 = f.val >> i0 & 1u, v1 = f.val >> i1 & 1u;
		if (c0 && c1 && v0 != v1) return false;
		if (c0 || c1) { g.care |= 1u << i; g.val |= (c0 ? v0 : v1) << i; }
	}
	*out = g;
	return true;
}

Fn reduce_support(Fn f)
{
	for (int k = f.n - 1; k >= 0; k--)
	{
		Fn g;
		if (drop_var(f, k, &g)) f = g;
	}
	return f;
}

// One op computing f (n <= 3). Unused inputs read the first variable (any register will do).
int emit_small(Builder &b, const Fn &f)
{
	unsigned imm = 0;
	for (unsigned idx = 0; idx < 8; idx++)
	{
		// op inputs (a, b, c) = variables (2, 1, 0); missing variables are ignored
		unsigned assign = 0;
		for (int k = 0; k < f.n; k++) assign |= ((idx >> k) & 1u) << k;
		if (f.val >> assign & 1u) imm |= 1u << idx;
	}
	const int s0 = f.slot[0], s1 = f.n > 1 ? f.slot[1] : s0, s2 = f.n > 2 ? f.slot[2] : s0;
	return b.emit(imm, s2, s1, s0);
}

int synth(Builder &b, Fn f, int depth);

// f = h(g(bound three variables), the other n - 3 variables): try every bound set; returns the slot or -1.
int try_decompose(Builder &b, const Fn &f)
{
	const int n = f.n;
	if (n < 4) return -1;
	const int nfree = n - 3;
	for (int m = 0; m < (1 << n); m++)
	{
		if (__builtin_popcount((unsigned)m) != 3) continue;
		int bound[3], fr[2], nb = 0, nf = 0;
		for (int k = 0; k < n; k++) (m >> k & 1) ? bound[nb++] = k : fr[nf++] = k;
		// columns: for each of the 8 bound assignments, the function of the free variables
		unsigned colv[8], colc[8];
		for (unsigned ba = 0; ba < 8; ba++)
		{
			colv[ba] = colc[ba] = 0;
			for (unsigned fa = 0; fa < (1u << nfree); fa++)
			{
				unsigned i = 0;
				for (int q = 0; q < 3; q++) i |= ((ba >> q) & 1u) << bound[q];
				for (int q = 0; q < nfree; q++) i |= ((fa >> q) & 1u) << fr[q];
				if (f.care >> i & 1u) { colc[ba] |= 1u << fa; colv[ba] |= (f.val >> i & 1u) << fa; }
			}
		}
		// split the columns into two classes of mutually compatible columns (g = 0 / g = 1)
		for (unsigned part = 0; part < 128; part++) // column 0 in class 0
		{
			unsigned mv[2] = {0, 0}, mc[2] = {0, 0};
			bool ok = true;
			for (unsigned ba = 0; ba < 8 && ok; ba++)
			{
				const unsigned cls = ba ? (part >> (ba - 1)) & 1u : 0u;
				const unsigned both = mc[cls] & colc[ba];
				if ((mv[cls] ^ colv[ba]) & both) ok = false;
				mv[cls] |= colv[ba] & colc[ba];
				mc[cls] |= colc[ba];
			}
			if (!ok) continue;
			// g over the bound variables
			Fn g;
			g.n = 3;
			g.val = 0;
			g.care = 0xFFu;
			for (int q = 0; q < 3; q++) g.slot[q] = f.slot[bound[q]];
			for (unsigned ba = 1; ba < 8; ba++)
				if ((part >> (ba - 1)) & 1u) g.val |= 1u << ba;
			const int gs = emit_small(b, g);
			// h over (free variables..., g): variable index nfree = g
			Fn h;
			h.n = nfree + 1;
			h.val = h.care = 0;
			for (int q = 0; q < nfree; q++) h.slot[q] = f.slot[fr[q]];
			h.slot[nfree] = gs;
			for (unsigned cls = 0; cls < 2; cls++)
				for (unsigned fa = 0; fa < (1u << nfree); fa++)
					if (mc[cls] >> fa & 1u) { const unsigned i = fa | (cls << nfree); h.care |= 1u << i; h.val |= (mv[cls] >> fa & 1u) << i; }
			return emit_small(b, h);
		}
	}
	return -1;
}

int const_slot(Builder &b, int value)
{
	// constants are rare (a rule-set that is always / never satisfied): materialise through an op on any input
	Fn f;
	f.n = 1;
	f.slot[0] = 0;
	f.care = 3u;
	f.val = value ? 3u : 0u;
	return emit_small(b, f);
}

int synth(Builder &b, Fn f, int depth)
{
	f = reduce_support(f);
	int cv;
	if (is_const(f, &cv)) return const_slot(b, cv);
	if (f.n <= 3) return emit_small(b, f);
	// try the two-op decomposition on a scratch copy first
	{
		Builder trial = b;
		const int s = try_decompose(trial, f);
		if (s >= 0) { b = trial; return s; }
	}
	// Shannon on the variable whose cofactors are cheapest
	int best_cost = 1 << 30, best_k = f.n - 1;
	for (int k = 0; k < f.n && depth < 2; k++)
	{
		Builder trial = b;
		const size_t before = trial.ops.size();
		Fn c0, c1;
		c0.n = c1.n = f.n - 1;
		c0.val = c0.care = c1.val = c1.care = 0;
		int s = 0;
		for (int i = 0; i < f.n; i++)
			if (i != k) { c0.slot[s] = c1.slot[s] = f.slot[i]; s++; }
		for (unsigned i = 0; i < (1u << (f.n - 1)); i++)
		{
			const unsigned lo = i & ((1u << k) - 1u), hi = (i >> k) << (k + 1);
			const unsigned i0 = hi | lo, i1 = i0 | (1u << k);
			if (f.care >> i0 & 1u) { c0.care |= 1u << i; c0.val |= (f.val >> i0 & 1u) << i; }
			if (f.care >> i1 & 1u) { c1.care |= 1u << i; c1.val |= (f.val >> i1 & 1u) << i; }
		}
		synth(trial, c0, depth + 1);
		synth(trial, c1, depth + 1);
		const int cost = (int)(trial.ops.size() - before);
		if (cost < best_cost) { best_cost = cost; best_k = k; }
	}
	const int k = best_k;
	Fn c0, c1;
	c0.n = c1.n = f.n - 1;
	c0.val = c0.care = c1.val = c1.care = 0;
	int s = 0;
	for (int i = 0; i < f.n; i++)
		if (i != k) { c0.slot[s] = c1.slot[s] = f.slot[i]; s++; }
	for (unsigned i = 0; i < (1u << (f.n - 1)); i++)
	{
		const unsigned lo = i & ((1u << k) - 1u), hi = (i >> k) << (k + 1);
		const unsigned i0 = hi | lo, i1 = i0 | (1u << k);
		if (f.care >> i0 & 1u) { c0.care |= 1u << i; c0.val |= (f.val >> i0 & 1u) << i; }
		if (f.care >> i1 & 1u) { c1.care |= 1u << i; c1.val |= (f.val >> i1 & 1u) << i; }
	}
	const int s0 = synth(b, c0, depth + 1), s1 = synth(b, c1, depth + 1);
	return b.emit(0xCAu, f.slot[k], s1, s0); // v ? s1 : s0
}

int planes_for(MainKind m) { return (m == MAIN_VN || m == MAIN_VN2D) ? 3 : (m == MAIN_MOORE ? 5 : 4); }

} // namespace

// Device source of `jit_rule_word(alive, mn, ed, co)` for these rules, or "" when synthesis is not worthwhile / failed
// its self-check. `ops_out` receives the op count (diagnostics).
std::string synthesize_rule_source(const CanonRules &r, int *ops_out)
{
	Builder b;
	const int np[3] = {planes_for(r.main), 4, 4};
	const char *arr[3] = {"mn", "ed", "co"};
	int slot[3][5];
	for (int s = 0; s < 3; s++)
		for (int k = 0; k < np[s]; k++) slot[s][k] = b.input(std::string(arr[s]) + "[" + std::to_string(k) + "]");
	const int alive = b.input("alive");
	int acc[2] = {-1, -1}; // survive, born
	for (int which = 0; which < 2; which++)
	{
		std::vector<int> terms;
		bool always = false;
		for (int s = 0; s < 3; s++)
		{
			if (s > 0 && !r.need[s])
			{
				// a rule-set whose count cannot matter is the constant its table holds at count 0
				if (((which ? r.onset_born[s] : r.onset_survive[s]) & 1u)) always = true;
				continue;
			}
			Fn f;
			f.n = np[s];
			f.val = f.care = 0;
			for (int k = 0; k < f.n; k++) f.slot[k] = slot[s][k];
			const uint32_t onset = which ? r.onset_born[s] : r.onset_survive[s];
			for (unsigned c = 0; c <= r.lists.n[s] && c < (1u << f.n); c++) { f.care |= 1u << c; f.val |= (onset >> c & 1u) << c; }
			Fn red = reduce_support(f);
			int cv;
			if (is_const(red, &cv)) { if (cv) always = true; continue; }
			terms.push_back(synth(b, f, 0));
		}
		if (always) { acc[which] = const_slot(b, 1); continue; }
		if (terms.empty()) { acc[which] = const_slot(b, 0); continue; }
		int a = terms[0];
		for (size_t i = 1; i < terms.size(); i += 2)
		{
			const int t1 = terms[i], t2 = i + 1 < terms.size() ? terms[i + 1] : t1;
			a = b.emit(0xFEu, a, t1, t2); // OR3 as a bitop3 (v_or3_b32 issues at half rate)
		}
		acc[which] = a;
	}
	const int out = b.emit(0xCAu, alive, acc[0], acc[1]); // alive ? survive : born
	// ---- peephole: fold an op whose result is used once into its single consumer when the consumer has room
	// (the OR of a two-input term, the final select of a one-op side): keeps the program correct by re-deriving immediates
	// ---- self-check on every reachable count combination (per rule-set independently: the program is an OR of functions
	// of disjoint variable sets, so checking each set with the others at count 0 and then the combination rule is enough;
	// here simply: exhaustive over the three counts)
	{
		std::vector<uint8_t> v(b.names.size());
		for (unsigned cm = 0; cm <= r.lists.n[0]; cm++)
			for (unsigned ce = 0; ce <= (r.need[1] ? r.lists.n[1] : 0u); ce++)
				for (unsigned cc = 0; cc <= (r.need[2] ? r.lists.n[2] : 0u); cc++)
					for (unsigned al = 0; al < 2; al++)
					{
						const unsigned cnt[3] = {cm, ce, cc};
						for (int s = 0; s < 3; s++)
							for (int k = 0; k < np[s]; k++) v[slot[s][k]] = (cnt[s] >> k) & 1u;
						v[alive] = (uint8_t)al;
						for (const Op &o : b.ops) v[o.dst] = (o.imm >> ((v[o.in[0]] << 2) | (v[o.in[1]] << 1) | v[o.in[2]])) & 1u;
						unsigned want = 0;
						for (int s = 0; s < 3; s++)
						{
							const uint32_t onset = al ? r.onset_survive[s] : r.onset_born[s];
							const unsigned c = (s > 0 && !r.need[s]) ? 0u : cnt[s];
							want |= (onset >> c) & 1u;
						}
						if (v[out] != want) { if (ops_out) *ops_out = -1; return std::string(); }
					}
	}
	if (ops_out) *ops_out = (int)b.ops.size();
	std::string src = "// generated by rule_synth.cpp: " + std::to_string(b.ops.size()) + " ops\n"
	                  "#define CA3D_JIT_RULE_FN 1\n"
	                  "__device__ __forceinline__ u32 jit_rule_word(u32 alive, const u32 *mn, const u32 *ed, const u32 *co)\n{\n";
	for (const Op &o : b.ops)
	{
		char line[256];
		snprintf(line, sizeof line, "\tconst u32 %s = bitop3<0x%02X>(%s, %s, %s);\n", b.names[o.dst].c_str(), o.imm, b.names[o.in[0]].c_str(),
		         b.names[o.in[1]].c_str(), b.names[o.in[2]].c_str());
		src += line;
	}
	src += "\treturn " + b.names[out] + ";\n}\n";
	return src;
}

} // namespace ca3d
