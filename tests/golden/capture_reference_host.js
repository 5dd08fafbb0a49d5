#!/usr/bin/env node
/*
 * Fixture generator: runs the REFERENCE's own host JavaScript (main_pathtraced.js + MemoryManager.js +
 * ui.js + libs/wgpu-matrix.module.js) under Node with DOM/WebGPU shims and a recording fake GPUDevice, and
 * writes the values it produces (rule LUTs, neighbourhood tables, packed initial state, dispatch shape,
 * uniform block) to tests/golden/reference_host.json.
 *
 * Only DATA is written into the repo. The reference sources are copied to a throw-away directory under
 * os.tmpdir() for the duration of the run (they have no package.json, so Node needs {"type":"module"} beside
 * them) and deleted afterwards. This script only works where /root/reference is mounted (the build
 * container); the GPU box never runs it and only consumes the JSON.
 *
 * Usage: node tests/golden/capture_reference_host.js [/root/reference]
 */
"use strict";
const fs = require("fs");
const os = require("os");
const path = require("path");

const REF = process.argv[2] || "/root/reference";
const OUT = path.join(__dirname, "reference_host.json");

function copyReference(dst)
{
	fs.mkdirSync(path.join(dst, "libs"), { recursive: true });
	for (const f of ["main_pathtraced.js", "MemoryManager.js", "ui.js", "libs/wgpu-matrix.module.js"])
	{
		fs.copyFileSync(path.join(REF, f), path.join(dst, f));
	}
	fs.writeFileSync(path.join(dst, "package.json"), JSON.stringify({ type: "module" }));
}

function installShims(width, height)
{
	global.performance = require("perf_hooks").performance;
	global.window = {
		innerWidth: width, innerHeight: height, devicePixelRatio: 1,
		addEventListener() {}
	};
	global.document = { querySelector() { return null; }, body: { insertAdjacentHTML() {} } };
	global.GPUBufferUsage = { STORAGE: 1, COPY_DST: 2, COPY_SRC: 4, UNIFORM: 8, VERTEX: 16, INDEX: 32 };
	global.GPUShaderStage = { VERTEX: 1, FRAGMENT: 2, COMPUTE: 4 };
	global.GPUTextureUsage = { RENDER_ATTACHMENT: 1, COPY_SRC: 2, TEXTURE_BINDING: 4 };
	if (!String.prototype.replaceAll)
	{
		// Node 12 lacks it; the reference's rule parser uses it with a plain-string pattern.
		// eslint-disable-next-line no-extend-native
		String.prototype.replaceAll = function (a, b) { return this.split(a).join(b); };
	}
}

function makeFakeDevice(record)
{
	return {
		createBuffer(d) { return { label: d.label, size: d.size, destroy() {} }; },
		createBindGroupLayout(d) { return d; },
		createBindGroup(d) { return d; },
		queue: {
			writeBuffer(buf, off, data)
			{
				let copy;
				if (data instanceof ArrayBuffer) { copy = Array.from(new Float32Array(data.slice(0))); }
				else { copy = Array.from(data); }
				record.push({ label: buf.label, offset: off, ctor: data.constructor.name, data: copy });
			}
		}
	};
}

function sparse(arr)
{
	const nz = [];
	for (let i = 0; i < arr.length; i++) { if (arr[i] !== 0) { nz.push([i, arr[i] >>> 0]); } }
	return { length: arr.length, nonzero: nz };
}

async function main()
{
	const tmp = fs.mkdtempSync(path.join(os.tmpdir(), "ca3d-refcap-"));
	const W = 1920, H = 1080;
	const log = console.log;
	try
	{
		copyReference(tmp);
		installShims(W, H);
		console.log = function () {};
		process.on("unhandledRejection", function () {}); // init() rejects at navigator.gpu, as expected.
		const MM = await import(path.join(tmp, "MemoryManager.js"));
		await import(path.join(tmp, "main_pathtraced.js"));
		global.window.onload();
		await new Promise((r) => setTimeout(r, 50));
		const mm = global.window.mm;
		const out = { generator: "tests/golden/capture_reference_host.js", node: process.version, window: [W, H] };

		// --- rule-string parser samples (main_pathtraced.js:554-581)
		out.rules_components = {};
		for (const s of ["1,3", "0-6", "27", "1,2-5,7, 30", "4", "5-7", "4-7", "3-5", "2-4", "0", "26", "0-26", "13-14,17-19", " 2 , 6 ,9"])
		{
			out.rules_components[s] = mm._rulesComponentsToValues(s);
		}

		// --- grid-size formatter samples (675-693)
		out.grid_size_formatter = {};
		for (const v of [3, 16, 17, 32, 48, 49, 64, 100, 112, 113, 256, 500, 1000, 1024])
		{
			out.grid_size_formatter[v] = mm._gridSizeUIFormatter(v);
		}

		// --- storage buffers for every neighbourhood name + LUTs for a set of rule configurations
		const ruleConfigs = [
			{ name: "default", n: "von neumann", b: "1,3", s: "0-6", be: "27", se: "27", bc: "27", sc: "27" },
			{ name: "clustered", n: "moore", b: "5-7", s: "4-7", be: "4", se: "3-5", bc: "3", sc: "2-4" },
			{ name: "life2d", n: "moore 2D", b: "3", s: "2,3", be: "27", se: "27", bc: "27", sc: "27" },
			{ name: "vn2d", n: "von neumann 2D", b: "1", s: "", be: "27", se: "27", bc: "27", sc: "27" },
			{ name: "edges_main", n: "edges", b: "2,6,9", s: "4,6,8-9", be: "2", se: "27", bc: "27", sc: "1" },
			{ name: "corners_main", n: "corners", b: "1", s: "0-8", be: "27", se: "0", bc: "8", sc: "27" },
			{ name: "moore_b4s4", n: "moore", b: "4", s: "4", be: "27", se: "27", bc: "27", sc: "27" },
			{ name: "vn_b1", n: "von neumann", b: "1", s: "", be: "27", se: "27", bc: "27", sc: "27" }
		];
		out.rule_configs = [];
		for (const rc of ruleConfigs)
		{
			mm._neighbourhood = rc.n;
			mm._bornRulesString = rc.b; mm._surviveRulesString = rc.s;
			mm._bornRulesStringEdges = rc.be; mm._surviveRulesStringEdges = rc.se;
			mm._bornRulesStringCorners = rc.bc; mm._surviveRulesStringCorners = rc.sc;
			mm._gridSize = 32;
			mm._randomInitialState = false;
			mm._recalculateRulesValues();
			const rec = [];
			mm._device = makeFakeDevice(rec);
			mm._cellStorageBuffers = []; mm._storageBuffers = {};
			mm._setupStorageBuffers();
			const byLabel = {};
			for (const r of rec) { byLabel[r.label] = r; }
			out.rule_configs.push({
				name: rc.name, neighbourhood: rc.n,
				strings: { born: rc.b, survive: rc.s, bornEdges: rc.be, surviveEdges: rc.se, bornCorners: rc.bc, surviveCorners: rc.sc },
				main_offsets: byLabel["neighbourhood buffer"].data,
				edges_offsets: byLabel["edges neighbourhood buffer"].data,
				corners_offsets: byLabel["corners neighbourhood buffer"].data,
				survive: byLabel["survive rules buffer"].data,
				born: byLabel["born rules buffer"].data,
				offsets_ctor: byLabel["neighbourhood buffer"].ctor,
				lut_ctor: byLabel["born rules buffer"].ctor
			});
		}

		// restore defaults
		mm._neighbourhood = "von neumann"; mm._bornRulesString = "1,3"; mm._surviveRulesString = "0-6";
		mm._bornRulesStringEdges = "27"; mm._surviveRulesStringEdges = "27";
		mm._bornRulesStringCorners = "27"; mm._surviveRulesStringCorners = "27";
		mm._recalculateRulesValues();

		// --- default (single-seed) initial state per grid size; both ping-pong buffers (1228-1362)
		out.initial_state = {};
		for (const G of [32, 64, 96, 128, 256])
		{
			mm._gridSize = G; mm._randomInitialState = false;
			const rec = [];
			mm._device = makeFakeDevice(rec);
			mm._setupStorageBuffers();
			const cs0 = rec.find((r) => r.label === "cell_state_0");
			const cs1 = rec.find((r) => r.label === "cell_state_1");
			out.initial_state[G] = { cell_state_0: sparse(cs0.data), cell_state_1: sparse(cs1.data), ctor: cs0.ctor };
		}

		// --- random 5x5x5 initial state with Math.random replaced by a fixed sequence so it can be replayed
		out.random_state = {};
		const realRandom = Math.random;
		for (const G of [32, 64, 128])
		{
			let k = 0;
			const seq = [];
			Math.random = function () { k = (k * 1103515245 + 12345) & 0x7fffffff; const v = k / 0x80000000; seq.push(v); return v; };
			mm._gridSize = G; mm._randomInitialState = true;
			const rec = [];
			mm._device = makeFakeDevice(rec);
			mm._setupStorageBuffers();
			Math.random = realRandom;
			const cs0 = rec.find((r) => r.label === "cell_state_0");
			out.random_state[G] = { draws: seq, cell_state_0: sparse(cs0.data) };
		}
		mm._randomInitialState = false;

		// --- word index helper samples (1170-1178)
		out.cluster_idx = [];
		for (const G of [32, 64, 96])
		{
			mm._gridSize = G;
			for (const c of [[0, 0, 0], [31, 0, 0], [32, 1, 0], [G - 1, G - 1, G - 1], [G, 5, 7], [5, G, 7], [5, 7, G], [33, 2, 3]])
			{
				out.cluster_idx.push({ G, cell: c, idx: mm._getClusterIdxFromGridCoordinates({ x: c[0], y: c[1], z: c[2] }) });
			}
		}

		// --- dispatch shape and ping-pong order (1796-1809)
		out.dispatch = {};
		for (const G of [32, 64, 96, 256, 512, 1024])
		{
			mm._gridSize = G; mm._simulationStep = 0;
			mm._computePipeline = "pipeline"; mm._commonBindGroup = "common"; mm._automatonRulesBindGroup = "rules";
			mm._cellStatesBindGroups = ["cs0_in0_out1", "cs1_in1_out0"];
			const calls = [];
			const enc = { beginComputePass() { return {
				setPipeline() {}, setBindGroup(i, g) { calls.push(["bind", i, g]); },
				dispatchWorkgroups(x, y, z) { calls.push(["dispatch", x, y, z]); }, end() {} }; } };
			mm._computePass(enc); mm._computePass(enc); mm._computePass(enc);
			out.dispatch[G] = { calls, simulationStepAfter: mm._simulationStep };
		}
		mm._simulationStep = 0;

		// --- bind-group wiring of the state buffers (1558-1610): which buffer is binding 0 (in) / 1 (out)
		{
			mm._gridSize = 64;
			const rec = [];
			mm._device = makeFakeDevice(rec);
			mm._setupStorageBuffers();
			mm._setupCellStorageBindGroups();
			out.cell_bind_groups = mm._cellStatesBindGroups.map((g) => g.entries.map((e) => ({ binding: e.binding, buffer: e.resource.buffer.label })));
			mm._setupAutomatonRulesBindGroup();
			out.rules_bind_group = mm._automatonRulesBindGroup.entries.map((e) => ({ binding: e.binding, buffer: e.resource.buffer.label }));
		}

		// --- uniform block (MemoryManager.js + 464-493, 504-518, 1747-1773)
		mm._setupUniformsMemoryCPU();
		mm._lightSource.update();
		mm._updateMatrices();
		mm._updateUIValues();
		out.uniform_block = {
			f32: Array.from(MM.bufferf32),
			indices: {
				light: mm._lightSource._bufferIndex,
				viewMatrices: mm._viewMatricesBufferIndex, windowSize: mm._windowSizeIndex,
				elapsedTime: mm._elapsedTimeIndex, depthSamples: mm._depthRaySamplesIndex,
				shadowSamples: mm._shadowRaySamplesIndex, cellSize: mm._cellSizeIndex,
				showDepthOverlay: mm._showDepthOverlayIndex, temporalAlpha: mm._temporalAlphaIndex,
				baseReflectivity: mm._baseReflectivityIndex, roughness: mm._roughnessIndex,
				materialColor: mm._materialColorIndex, gamma: mm._gammaIndex
			},
			viewMat: Array.from(mm._viewMat), inverseViewMat: Array.from(mm._inverseViewMat),
			projectionMat: Array.from(mm._projectionMat), projViewMatInv: Array.from(mm._projViewMatInv),
			fov: mm._fov
		};
		// One frame later: previous-frame matrices are populated by _updatePrevMatrices (520-524).
		mm._updatePrevMatrices();
		mm._updateMatrices();
		out.uniform_block.f32_second_frame = Array.from(MM.bufferf32);

		fs.writeFileSync(OUT, JSON.stringify(out));
		console.log = log;
		console.log("wrote", OUT, fs.statSync(OUT).size, "bytes");
	}
	finally
	{
		console.log = log;
		fs.rmdirSync(tmp, { recursive: true });
	}
}

main().catch((e) => { console.error(e); process.exit(1); });
