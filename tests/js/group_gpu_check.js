// GPU check of the single-thread multi-GPU surface from the JavaScript host: EngineGroup (ca3d_group_*) with several slabs
// on device 0 (peer-copy transport) against one full-grid Engine — states and frames must be identical; the RCCL transport
// with a one-slab group (RCCL refuses two ranks on one device).
"use strict";
const assert = require("assert");
const fs = require("fs");
const path = require("path");
const root = path.join(__dirname, "..", "..");
const c = require(path.join(root, "cellularautomatons3d_amd", "js", "ca3d.js"));

const outDir = process.argv[2];
const G = 256, K = 4;
const rules = { neighbourhood: "moore", born: "5-7", survive: "4-7", bornEdges: "4", surviveEdges: "3-5", bornCorners: "3", surviveCorners: "2-4" };
const st = c.randomFill((G / 32) * G * G, 31, 1);
const one = new c.Engine(0);
one.configure(G);
for (const [ranks, r] of [[4, rules], [8, {}], [2, {}]])
{
	one.setRuleStrings(r);
	one.uploadState(st);
	const grp = new c.EngineGroup(new Array(ranks).fill(0));
	grp.configure(G, K);
	grp.setRuleStrings(r);
	grp.uploadState(st);
	assert.strictEqual(grp.info(ranks - 1).z0, G - G / ranks);
	// 13 steps = three full batches and a short one, then 5 more (the ghosts are valid on entry)
	for (const n of [13, 5]) { grp.step(n); one.step(n); }
	assert.deepStrictEqual(Buffer.from(grp.readState().buffer), Buffer.from(one.readState().buffer), `${ranks} slabs`);
	if (ranks === 4)
	{
		const ub = fs.readFileSync(path.join(outDir, "uniforms.f32"));
		const u = new Float32Array(ub.buffer.slice(ub.byteOffset, ub.byteOffset + 512));
		const W = 320, H = 176;
		const a = { presentation: new Uint8Array(W * H * 4), light: new Uint16Array(W * H * 4), depth: new Uint16Array(W * H * 2) };
		const b = { presentation: new Uint8Array(W * H * 4), light: new Uint16Array(W * H * 4), depth: new Uint16Array(W * H * 2) };
		grp.render(u, W, H, 4, a);
		one.render(u, W, H, 4, b);
		for (const k of ["presentation", "light", "depth"]) assert.deepStrictEqual(Buffer.from(a[k].buffer), Buffer.from(b[k].buffer), k);
		assert.ok(a.presentation.some((v, i) => i % 4 !== 3 && v !== 0), "the frame shows something");
	}
	grp.close();
}
// RCCL transport: communicators from ncclCommInitAll, the exchange one ncclGroupStart / End — a one-slab group wraps onto itself
{
	one.setRuleStrings({});
	one.uploadState(st);
	const grp = new c.EngineGroup([0]);
	assert.throws(() => new c.EngineGroup([0, 0]).setOption("transport", 1), /one device per slab/);
	grp.configure(G, 8);
	grp.setRuleStrings({});
	grp.setOption("transport", 1);
	grp.uploadState(st);
	grp.step(20);
	one.step(20);
	assert.deepStrictEqual(Buffer.from(grp.readState().buffer), Buffer.from(one.readState().buffer), "rccl transport");
	grp.close();
}
one.close();
console.log("ok");
