// N2 check (build container only: needs /root/reference): the reference's UNMODIFIED main_pathtraced.js + ui.js +
// MemoryManager.js run under Node with DOM shims and the navigator.gpu façade; a recording mock stands in for the
// engine (no GPU here). Asserts that the host's own init + frame loop reach the engine with the right payloads.
"use strict";
const assert = require("assert");
const fs = require("fs");
const os = require("os");
const path = require("path");
const REF = process.argv[2] || "/root/reference";
const root = path.join(__dirname, "..", "..");
const { installFacade } = require(path.join(root, "cellularautomatons3d_amd", "js", "webgpu_facade.js"));
const host = require(path.join(root, "cellularautomatons3d_amd", "js", "ca3d.js"));

const calls = [];
const mock = {
	configure(G) { calls.push(["configure", G]); },
	setRules(m, e, c, s, b) { calls.push(["setRules", Array.from(m), Array.from(e), Array.from(c), Array.from(s), Array.from(b)]); },
	setRuleStrings(r) { calls.push(["setRuleStrings", r]); },
	uploadState(w) { calls.push(["uploadState", Array.from(w).map((v, i) => [i, v >>> 0]).filter((x) => x[1] !== 0), w.length]); },
	step(n) { calls.push(["step", n]); },
	render(u, W, H, spp, t) { calls.push(["render", Array.from(u), W, H, spp, !!(t && t.presentation)]); }
};

async function main()
{
	const tmp = fs.mkdtempSync(path.join(os.tmpdir(), "ca3d-facade-"));
	const log = console.log;
	try
	{
		fs.mkdirSync(path.join(tmp, "libs"));
		for (const f of ["main_pathtraced.js", "MemoryManager.js", "ui.js", "libs/wgpu-matrix.module.js"]) { fs.copyFileSync(path.join(REF, f), path.join(tmp, f)); }
		fs.writeFileSync(path.join(tmp, "package.json"), JSON.stringify({ type: "module" }));
		const W = 640, H = 360;
		let now = 1000;
		global.performance = { now: () => now };
		let raf = null;
		global.requestAnimationFrame = (cb) => { raf = cb; };
		global.window = { innerWidth: W, innerHeight: H, devicePixelRatio: 1, addEventListener() {} };
		const stubEl = { addEventListener() {}, removeEventListener() {}, querySelectorAll() { return []; }, classList: { add() {}, remove() {} } };
		const facade = installFacade(global, { engine: mock, width: W, height: H });
		const canvas = { width: 0, height: 0, getContext: () => facade.canvasContext, addEventListener() {}, requestPointerLock() {} };
		global.document = { querySelector: (q) => (q === ".main-canvas" ? canvas : stubEl), body: { insertAdjacentHTML() {} }, pointerLockElement: null, addEventListener() {} };
		global.fetch = async (url) => ({ text: async () => fs.readFileSync(path.join(REF, url), "utf8") });
		if (!String.prototype.replaceAll) { String.prototype.replaceAll = function (a, b) { return this.split(a).join(b); }; } // Node 12
		console.log = () => {};
		await import(path.join(tmp, "main_pathtraced.js"));
		global.window.onload();
		for (let i = 0; i < 200 && !raf; i++) { await new Promise((r) => setTimeout(r, 10)); }
		console.log = log;
		assert.ok(raf, "init() did not reach the frame loop");
		const mm = global.window.mm;
		// init() already ran one _updateLoop; run 4 more frames 50 ms apart (each crosses the 48 ms step gate)
		for (let f = 0; f < 4; f++) { now += 50; const cb = raf; raf = null; cb(); }
		const kinds = calls.map((c) => c[0]);
		const conf = calls.filter((c) => c[0] === "configure");
		assert.deepStrictEqual(conf[0], ["configure", 64]);
		const renders = calls.filter((c) => c[0] === "render");
		assert.ok(renders.length >= 5, "renders: " + renders.length);
		assert.strictEqual(renders[0][1].length, 128);
		assert.strictEqual(renders[0][1][68], W); assert.strictEqual(renders[0][1][69], H);
		assert.ok(Math.abs(renders[0][1][18] - 0.75) < 1e-6); // viewMat translation z (4 + 14)
		const up = calls.filter((c) => c[0] === "uploadState");
		assert.ok(up.length >= 1);
		assert.deepStrictEqual(up[0][1], [[4030, 0x80000000]]); // the single seed at G = 64
		assert.strictEqual(up[0][2], 8192);
		const rules = calls.filter((c) => c[0] === "setRules")[0];
		const lut = host.recalculateRulesValues(host.DEFAULT_RULES);
		assert.deepStrictEqual(rules[1], Array.from(host.NEIGHBOURHOOD_MAP["von neumann"]));
		assert.deepStrictEqual(rules[2], Array.from(host.NEIGHBOURHOOD_MAP["edges"]));
		assert.deepStrictEqual(rules[3], Array.from(host.NEIGHBOURHOOD_MAP["corners"]));
		assert.deepStrictEqual(rules[4], Array.from(lut.survive));
		assert.deepStrictEqual(rules[5], Array.from(lut.born));
		const steps = calls.filter((c) => c[0] === "step").length;
		assert.strictEqual(steps, mm._simulationStep);
		assert.ok(steps >= 4, "steps: " + steps);
		// render precedes the first step in the first submission (main_pathtraced.js:1842-1844)
		assert.ok(kinds.indexOf("render") < kinds.indexOf("step"));
		console.log(JSON.stringify({ ok: true, frames: renders.length, steps }));
	}
	finally
	{
		console.log = log;
		fs.rmdirSync(tmp, { recursive: true });
	}
}
main().catch((e) => { console.error(e); process.exit(1); });
