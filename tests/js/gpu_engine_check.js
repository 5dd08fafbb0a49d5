// GPU check of the JS host path: Engine (N-API addon -> libca3d.so) against the JS CPU stepper and the known
// default-rule hashes; writes a rendered frame for the Python side to compare with the oracle.
"use strict";
const assert = require("assert");
const fs = require("fs");
const path = require("path");
const root = path.join(__dirname, "..", "..");
const c = require(path.join(root, "cellularautomatons3d_amd", "js", "ca3d.js"));
const js = require(path.join(root, "oracle", "js_stepper.js"));

const outDir = process.argv[2];
const eng = new c.Engine(0);
// default rule from the single seed: FNV-1a anchors (SURVEY 8(c))
eng.restartSim(32, {});
const want = "8ebd1f9c c3f1930e 6b170f5a 7815260a dec86ca6 afa3b49c b79051df 23672217".split(" ");
for (let i = 0; i < 8; i++) { eng.step(1); assert.strictEqual(("00000000" + js.fnv1a32(eng.readState()).toString(16)).slice(-8), want[i]); }
assert.strictEqual(eng.info().step, 8);

// clustered rule on a random 128^3 grid vs the JS stepper
const G = 128;
const rules = { neighbourhood: "moore", born: "5-7", survive: "4-7", bornEdges: "4", surviveEdges: "3-5", bornCorners: "3", surviveCorners: "2-4" };
eng.configure(G);
eng.setRuleStrings(rules);
const st = c.randomFill((G / 32) * G * G, 77, 1);
eng.uploadState(st);
eng.step(3);
const lut = c.recalculateRulesValues(rules);
const step = js.makeStepper(G, [c.NEIGHBOURHOOD_MAP["moore"], c.NEIGHBOURHOOD_MAP["edges"], c.NEIGHBOURHOOD_MAP["corners"]], lut.survive, lut.born);
let a = st.slice(), b = new Uint32Array(a.length);
for (let i = 0; i < 3; i++) { step(a, b); const t = a; a = b; b = t; }
assert.deepStrictEqual(Buffer.from(eng.readState().buffer), Buffer.from(a.buffer));
assert.ok(/class/.test(eng.info().kernelName));
assert.ok(eng.stats().gpuMs > 0);

// queued submission: step() encodes, flush() / any state-reading call submits; the same states as call-by-call
{
	const before = eng.readState();
	eng.setOption("queue", 16);
	const l0 = eng.info().launchesTotal;
	for (let i = 0; i < 5; i++) eng.step(2);
	eng.flush();
	assert.strictEqual(eng.info().step, 13);
	const queued = eng.readState();
	eng.setOption("queue", 0);
	eng.uploadState(before);
	for (let i = 0; i < 5; i++) eng.step(2);
	assert.deepStrictEqual(Buffer.from(eng.readState().buffer), Buffer.from(queued.buffer));
	assert.ok(eng.info().launchesTotal > l0);
	// back to the state the render check below expects (3 steps from `st`)
	eng.uploadState(st);
	eng.step(3);
}

// render the current state with a uniform block supplied by the caller (written by the Python test)
const ub = fs.readFileSync(path.join(outDir, "uniforms.f32")); // small reads come from a pooled ArrayBuffer: honour byteOffset
const u = new Float32Array(ub.buffer.slice(ub.byteOffset, ub.byteOffset + 512));
const W = 160, H = 90;
const pres = new Uint8Array(W * H * 4), light = new Uint16Array(W * H * 4), depth = new Uint16Array(W * H * 2);
eng.render(u, W, H, 1, { presentation: pres, light, depth });
fs.writeFileSync(path.join(outDir, "state.u32"), Buffer.from(eng.readState().buffer));
fs.writeFileSync(path.join(outDir, "presentation.u8"), Buffer.from(pres.buffer));
fs.writeFileSync(path.join(outDir, "light.f16"), Buffer.from(light.buffer));
assert.strictEqual(eng.renderStats().primaryRays, W * H);
// errors surface as exceptions carrying ca3d_last_error()
assert.throws(() => eng.configure(48), /multiple of 32/);
assert.throws(() => eng.uploadState(new Uint32Array(3)), /expected/);

// a slab with the exchange inside the engine (RCCL, one-rank chain: the wrap message goes to the same GPU) equals the full grid
{
	const Gs = 256, K = 4;
	const se = new c.Engine(0), fe = new c.Engine(0);
	se.configureSlab(Gs, 0, Gs, K);
	se.setRuleStrings({});
	fe.configure(Gs);
	fe.setRuleStrings({});
	const s0 = c.randomFill((Gs / 32) * Gs * Gs, 5, 0);
	se.uploadState(s0);
	fe.uploadState(s0);
	se.slabCommInit(c.Engine.commUniqueId(), 0, 1);
	se.slabRun(10, false);
	se.slabRun(3, true);
	fe.step(13);
	assert.deepStrictEqual(Buffer.from(se.readState().buffer), Buffer.from(fe.readState().buffer));
	se.close();
	fe.close();
}

// asynchronous forms: the same bytes as the blocking calls, the event loop keeps turning meanwhile, errors reject
(async () => {
	const stateSync = eng.readState();
	let ticks = 0;
	const timer = setInterval(() => { ticks++; }, 0);
	const pres2 = new Uint8Array(W * H * 4), light2 = new Uint16Array(W * H * 4);
	const jobs = [eng.readStateAsync(), eng.renderAsync(u, W, H, 1, { presentation: pres2, light: light2 }), eng.synchronizeAsync(), eng.readStateAsync()];
	// while a worker thread is inside the engine nothing else may enter it, and it cannot be destroyed under the worker
	assert.throws(() => eng.step(1), /asynchronous call is pending/);
	assert.throws(() => eng.close(), /asynchronous call is pending/);
	const done = await Promise.all(jobs);
	clearInterval(timer);
	assert.deepStrictEqual(Buffer.from(done[0].buffer), Buffer.from(stateSync.buffer));
	assert.deepStrictEqual(Buffer.from(done[3].buffer), Buffer.from(stateSync.buffer));
	assert.deepStrictEqual(Buffer.from(pres2.buffer), Buffer.from(pres.buffer));
	assert.deepStrictEqual(Buffer.from(light2.buffer), Buffer.from(light.buffer));
	await assert.rejects(eng.renderAsync(u, W, H, 3, {}), /spp must be 1 or 4/);
	const last = eng.readStateAsync();
	await eng.closeAsync(); // waits for what is pending, then destroys
	assert.deepStrictEqual(Buffer.from((await last).buffer), Buffer.from(stateSync.buffer));
	console.log("ok");
})().catch((e) => { console.error(e); process.exit(1); });
