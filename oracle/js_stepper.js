#!/usr/bin/env node
/*
 * JavaScript CPU stepper — TEST INFRASTRUCTURE / CPU BASELINE ONLY (never loaded by the product).
 * BASELINE.md 3 asks for a JS stepper timed on the GPU box's host cores next to the GPU numbers. This is the
 * word-parallel restatement of shaders/compute_clustered.wgsl (same formulation as oracle/ca_oracle.c's fast
 * form: per-offset shifted row words added into bit-sliced counters; padded-grid boundary of SURVEY Appendix A).
 *
 *   node oracle/js_stepper.js bench <G> <seconds> [rule]     -> one JSON line {gcells_s, steps, ...}
 *   node oracle/js_stepper.js hash  <G> <steps>              -> FNV-1a-32 of the default-rule run from the seed
 */
"use strict";
const path = require("path");
const host = require(path.join(__dirname, "..", "cellularautomatons3d_amd", "js", "ca3d.js"));

function makeStepper(G, lists, survive, born)
{
	const C = G / 32;
	const planeWords = C * G;
	const sets = lists.map((l, s) => {
		const offs = [];
		for (let i = 0; i + 2 < l.length; i += 3) { offs.push([l[i], l[i + 1], l[i + 2]]); }
		const bornK = [], surviveK = [];
		for (let k = 0; k < 27; k++) { if (born[k + 27 * s] === 1) { bornK.push(k); } if (survive[k + 27 * s] === 1) { surviveK.push(k); } }
		return { offs, bornK, surviveK };
	});
	const p = [new Uint32Array(C), new Uint32Array(C), new Uint32Array(C), new Uint32Array(C), new Uint32Array(C)];
	return function step(inp, out)
	{
		const S = new Uint32Array(C), B = new Uint32Array(C);
		for (let z = 0; z < G; z++) for (let y = 0; y < G; y++)
		{
			S.fill(0); B.fill(0);
			for (const set of sets)
			{
				if (set.bornK.length === 0 && set.surviveK.length === 0) { continue; }
				for (let q = 0; q < 5; q++) { p[q].fill(0); }
				for (const [dx, dy, dz] of set.offs)
				{
					let yy = y + dy, zz = z + dz;
					if (yy < 0 || zz < 0) { continue; }           // dropped by the >= 0 test
					if (yy === G) { yy = 0; }
					if (zz === G) { zz = 0; }                      // <= G passes and wraps
					const row = (zz * G + yy) * C;
					for (let cx = 0; cx < C; cx++)
					{
						const w = inp[row + cx];
						let v;
						if (dx === 0) { v = w; }
						else if (dx < 0) { v = (w << 1) | (cx > 0 ? inp[row + cx - 1] >>> 31 : 0); }
						else { v = (w >>> 1) | ((cx + 1 < C ? inp[row + cx + 1] : inp[row]) << 31); }
						let carry = v;
						for (let q = 0; q < 5 && carry !== 0; q++) { const t = p[q][cx] & carry; p[q][cx] ^= carry; carry = t; }
					}
				}
				for (let cx = 0; cx < C; cx++)
				{
					const eq = (k) => { let e = -1; for (let q = 0; q < 5; q++) { e &= ((k >> q) & 1) ? p[q][cx] : ~p[q][cx]; } return e; };
					for (const k of set.bornK) { B[cx] |= eq(k); }
					for (const k of set.surviveK) { S[cx] |= eq(k); }
				}
			}
			const row = (z * G + y) * C;
			for (let cx = 0; cx < C; cx++) { const a = inp[row + cx]; out[row + cx] = (a & S[cx]) | (~a & B[cx]); }
		}
	};
}

function fnv1a32(words)
{
	const b = new Uint8Array(words.buffer, words.byteOffset, words.byteLength);
	let h = 2166136261;
	for (let i = 0; i < b.length; i++) { h ^= b[i]; h = Math.imul(h, 16777619) >>> 0; }
	return h >>> 0;
}

function rulesFor(name)
{
	const r = name === "clustered" ? { neighbourhood: "moore", born: "5-7", survive: "4-7", bornEdges: "4", surviveEdges: "3-5", bornCorners: "3", surviveCorners: "2-4" } : {};
	const rr = Object.assign({}, host.DEFAULT_RULES, r);
	const lut = host.recalculateRulesValues(rr);
	return { lists: [host.NEIGHBOURHOOD_MAP[rr.neighbourhood], host.NEIGHBOURHOOD_MAP["edges"], host.NEIGHBOURHOOD_MAP["corners"]], lut };
}

if (require.main === module)
{
	const [mode, Gs, n, rule] = process.argv.slice(2);
	const G = parseInt(Gs, 10);
	const { lists, lut } = rulesFor(rule || "default");
	const step = makeStepper(G, lists, lut.survive, lut.born);
	if (mode === "hash")
	{
		let a = host.initialState(G), b = new Uint32Array(a.length);
		const hashes = [];
		for (let i = 0; i < parseInt(n, 10); i++) { step(a, b); const t = a; a = b; b = t; hashes.push(("00000000" + fnv1a32(a).toString(16)).slice(-8)); }
		console.log(JSON.stringify({ G, hashes }));
	}
	else
	{
		let a = host.randomFill(a_len(G)), b = new Uint32Array(a.length);
		step(a, b); // warm-up (JIT)
		const t0 = process.hrtime.bigint();
		let steps = 0, dt = 0;
		do { step(b, a); const t = a; a = b; b = t; steps++; dt = Number(process.hrtime.bigint() - t0) / 1e9; } while (dt < parseFloat(n));
		console.log(JSON.stringify({ gcells_s: G * G * G * steps / dt / 1e9, steps, seconds: dt, G, cores: 1, node: process.version, cpus: require("os").cpus().length, cpu_model: require("os").cpus()[0].model }));
	}
}

function a_len(G) { return (G / 32) * G * G; }

module.exports = { makeStepper, fnv1a32 };
