// JS host helpers (cellularautomatons3d_amd/js/ca3d.js) against the values captured from the reference's own host
// JavaScript (tests/golden/reference_host.json). Prints "ok" or throws.
"use strict";
const assert = require("assert");
const path = require("path");
const root = path.join(__dirname, "..", "..");
const c = require(path.join(root, "cellularautomatons3d_amd", "js", "ca3d.js"));
const g = require(path.join(root, "tests", "golden", "reference_host.json"));

for (const s of Object.keys(g.rules_components)) { assert.deepStrictEqual(c.rulesComponentsToValues(s), g.rules_components[s], s); }
for (const v of Object.keys(g.grid_size_formatter)) { assert.strictEqual(c.gridSizeUIFormatter(parseInt(v, 10)), g.grid_size_formatter[v]); }
for (const rc of g.rule_configs)
{
	assert.deepStrictEqual(Array.from(c.NEIGHBOURHOOD_MAP[rc.neighbourhood]), rc.main_offsets, rc.name);
	assert.deepStrictEqual(Array.from(c.NEIGHBOURHOOD_MAP["edges"]), rc.edges_offsets);
	assert.deepStrictEqual(Array.from(c.NEIGHBOURHOOD_MAP["corners"]), rc.corners_offsets);
	const lut = c.recalculateRulesValues(rc.strings);
	assert.deepStrictEqual(Array.from(lut.born), rc.born, rc.name);
	assert.deepStrictEqual(Array.from(lut.survive), rc.survive, rc.name);
}
for (const G of Object.keys(g.initial_state))
{
	const st = c.initialState(parseInt(G, 10));
	const nz = [];
	st.forEach((w, i) => { if (w) { nz.push([i, w]); } });
	assert.deepStrictEqual(nz, g.initial_state[G].cell_state_0.nonzero);
	assert.strictEqual(st.length, g.initial_state[G].cell_state_0.length);
}
for (const G of Object.keys(g.random_state))
{
	const draws = g.random_state[G].draws.slice();
	const st = c.initialState(parseInt(G, 10), true, () => draws.shift());
	const nz = [];
	st.forEach((w, i) => { if (w) { nz.push([i, w]); } });
	assert.deepStrictEqual(nz, g.random_state[G].cell_state_0.nonzero);
}
for (const r of g.cluster_idx) { assert.strictEqual(c.getClusterIdxFromGridCoordinates(r.G, { x: r.cell[0], y: r.cell[1], z: r.cell[2] }), r.idx); }
for (const G of Object.keys(g.dispatch))
{
	const d = g.dispatch[G].calls.filter((x) => x[0] === "dispatch")[0];
	assert.deepStrictEqual(c.dispatchShape(parseInt(G, 10)), d.slice(1));
}
// the addon loads and exports the ABI; without a GPU engine creation must throw, not fall back
const a = c.loadAddon();
assert.strictEqual(a.abiVersion(), 7);
for (const f of ["create", "destroy", "configure", "configureSlab", "setRules", "uploadState", "readState", "step", "slabStep", "synchronize", "info", "stats", "render", "renderStats", "setOption", "deviceCount"]) { assert.strictEqual(typeof a[f], "function", f); }
console.log("ok");
