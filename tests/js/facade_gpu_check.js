// N2 on the GPU box (no reference sources there): a driver that issues the same WebGPU call sequence as
// main_pathtraced.js (buffers + labels of _setupStorageBuffers 1314-1369, bind groups 1562-1673, one render pass +
// one compute pass per submit, ping-pong by step parity) against the façade backed by the REAL engine.
"use strict";
const assert = require("assert");
const path = require("path");
const root = path.join(__dirname, "..", "..");
const c = require(path.join(root, "cellularautomatons3d_amd", "js", "ca3d.js"));
const { installFacade } = require(path.join(root, "cellularautomatons3d_amd", "js", "webgpu_facade.js"));
const js = require(path.join(root, "oracle", "js_stepper.js"));

const G = 64, W = 192, H = 108;
const g = {};
const eng = new c.Engine(0);
const f = installFacade(g, { engine: eng, width: W, height: H });

(async () => {
	const adapter = await g.navigator.gpu.requestAdapter({ powerPreference: "high-performance" });
	const dev = await adapter.requestDevice();
	const U = g.GPUBufferUsage;
	const mk = (label, data, usage) => { const b = dev.createBuffer({ label, size: data.byteLength, usage }); dev.queue.writeBuffer(b, 0, data); return b; };
	const rules = { neighbourhood: "moore", born: "5-7", survive: "4-7", bornEdges: "4", surviveEdges: "3-5", bornCorners: "3", surviveCorners: "2-4" };
	const lut = c.recalculateRulesValues(rules);
	const seed = c.randomFill((G / 32) * G * G, 5, 1);
	const grid = mk("grid uniforms", new Float32Array([G, G, G]), U.UNIFORM | U.COPY_DST);
	const uniforms = new Float32Array(128);
	// default pose of the reference: identity rotation, camera at (0, 0, 0.75); light (0.721, 1, 1, 5)
	uniforms.set([0.721, 1, 1, 5], 0);
	uniforms.set([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0.75, 1], 4);
	uniforms.set([W, H, 0.5, 35, 30, 0.85, 0, 0.1, 0.17, 0.17, 0.17, 0.29, 0, 0, 0, 2], 68);
	const common = mk("common buffer f32", uniforms, U.UNIFORM | U.COPY_DST);
	const cs = [mk("cell_state_0", seed, U.STORAGE | U.COPY_DST), mk("cell_state_1", seed, U.STORAGE | U.COPY_DST)];
	const nb = mk("neighbourhood buffer", c.NEIGHBOURHOOD_MAP[rules.neighbourhood], U.STORAGE | U.COPY_DST);
	const eb = mk("edges neighbourhood buffer", c.NEIGHBOURHOOD_MAP["edges"], U.STORAGE | U.COPY_DST);
	const cb = mk("corners neighbourhood buffer", c.NEIGHBOURHOOD_MAP["corners"], U.STORAGE | U.COPY_DST);
	const sb = mk("survive rules buffer", lut.survive, U.STORAGE | U.COPY_DST);
	const bb = mk("born rules buffer", lut.born, U.STORAGE | U.COPY_DST);
	const bg = (entries) => dev.createBindGroup({ entries: entries.map((b, i) => ({ binding: Array.isArray(b) ? b[0] : i, resource: { buffer: Array.isArray(b) ? b[1] : b } })) });
	const commonBG = bg([[0, grid], [11, common]]);
	const cellBG = [bg([cs[0], cs[1]]), bg([cs[1], cs[0]])];
	const rulesBG = bg([nb, eb, cb, sb, bb]);
	const texBG = dev.createBindGroup({ entries: [] });
	const renderPipe = dev.createRenderPipeline({ fragment: { entryPoint: "fragment_main" } });
	const computePipe = dev.createComputePipeline({ compute: { entryPoint: "compute_main" } });
	let step = 0;
	for (let frame = 0; frame < 4; frame++)
	{
		const enc = dev.createCommandEncoder();
		const rp = enc.beginRenderPass({ colorAttachments: [{ view: f.canvasContext.getCurrentTexture().createView() }, {}, {}] });
		rp.setPipeline(renderPipe); rp.setBindGroup(0, commonBG); rp.setBindGroup(1, texBG); rp.setBindGroup(2, cellBG[step % 2]);
		rp.drawIndexed(6); rp.end();
		const cp = enc.beginComputePass();
		cp.setPipeline(computePipe); cp.setBindGroup(0, commonBG); cp.setBindGroup(1, cellBG[step % 2]); cp.setBindGroup(2, rulesBG);
		cp.dispatchWorkgroups(G / 32, Math.ceil(G / 16), Math.ceil(G / 16)); cp.end();
		step++;
		dev.queue.submit([enc.finish()]);
	}
	assert.strictEqual(f.state.frames, 4); assert.strictEqual(f.state.steps, 4);
	const stepper = js.makeStepper(G, [c.NEIGHBOURHOOD_MAP["moore"], c.NEIGHBOURHOOD_MAP["edges"], c.NEIGHBOURHOOD_MAP["corners"]], lut.survive, lut.born);
	let a = seed.slice(), b = new Uint32Array(a.length);
	for (let i = 0; i < 4; i++) { stepper(a, b); const t = a; a = b; b = t; }
	assert.deepStrictEqual(Buffer.from(eng.readState().buffer), Buffer.from(a.buffer));
	const frame = f.canvasContext.frame;
	assert.strictEqual(frame.length, W * H * 4);
	let lit = 0;
	for (let i = 0; i < frame.length; i += 4) { if (frame[i] + frame[i + 1] + frame[i + 2] > 0) { lit++; } }
	assert.ok(lit > 200, "lit pixels: " + lit);
	// a restart (new buffers with new data) is picked up: the step counter restarts with the new state
	const seed2 = c.initialState(G);
	const cs2 = [mk("cell_state_0", seed2, U.STORAGE | U.COPY_DST), mk("cell_state_1", seed2, U.STORAGE | U.COPY_DST)];
	const cell2 = bg([cs2[0], cs2[1]]);
	const enc = dev.createCommandEncoder();
	const cp = enc.beginComputePass();
	cp.setPipeline(computePipe); cp.setBindGroup(0, commonBG); cp.setBindGroup(1, cell2); cp.setBindGroup(2, rulesBG);
	cp.dispatchWorkgroups(2, 4, 4); cp.end();
	dev.queue.submit([enc.finish()]);
	assert.strictEqual(eng.info().step, 1);
	eng.close();
	console.log("ok");
})().catch((e) => { console.error(e); process.exit(1); });
