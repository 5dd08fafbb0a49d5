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
eng.close();
console.log("ok");
