/*
 * A `navigator.gpu`-shaped façade over the MI355X engine (SURVEY 8(f) N2): just enough of the WebGPU JS API, as
 * main_pathtraced.js uses it (call sites listed in SURVEY 8(b)), for the reference's UNMODIFIED host to drive
 * libca3d.so. Nothing is interpreted: the two pipelines are recognised by their entry points
 * (`compute_main`, `fragment_main`), bind groups are resolved to the typed arrays the host uploaded with
 * queue.writeBuffer, and at queue.submit time
 *     a compute pass  ->  engine.configure / setRules / uploadState (when their source buffers changed) + engine.step(1)
 *     a render pass   ->  engine.render(common uniform block, canvas size) into the canvas context's RGBA8 frame.
 *
 *   const { installFacade } = require("./webgpu_facade.js");
 *   const gpu = installFacade(globalThis, { engine: new Engine(0), width: 1920, height: 1080 });
 *   // then load index.html's module (main_pathtraced.js) in a DOM shim / Electron window as usual;
 *   // gpu.canvasContext.frame is the Uint8Array(W*H*4) presentation after every submit.
 */
"use strict";

const BUFFER_USAGE = { MAP_READ: 1, MAP_WRITE: 2, COPY_SRC: 4, COPY_DST: 8, INDEX: 16, VERTEX: 32, UNIFORM: 64, STORAGE: 128, INDIRECT: 256, QUERY_RESOLVE: 512 };
const SHADER_STAGE = { VERTEX: 1, FRAGMENT: 2, COMPUTE: 4 };
const TEXTURE_USAGE = { COPY_SRC: 1, COPY_DST: 2, TEXTURE_BINDING: 4, STORAGE_BINDING: 8, RENDER_ATTACHMENT: 16 };

let nextBufferId = 1;

class FBuffer
{
	constructor(desc) { this.id = nextBufferId++; this.label = desc.label || ""; this.size = desc.size; this.usage = desc.usage; this.bytes = new Uint8Array(desc.size); this.version = 0; this.destroyed = false; }
	destroy() { this.destroyed = true; }
}

class FTexture
{
	constructor(desc) { this.label = desc.label || ""; this.width = desc.size[0] !== undefined ? desc.size[0] : desc.size.width; this.height = desc.size[1] !== undefined ? desc.size[1] : desc.size.height; this.format = desc.format; }
	createView() { return { texture: this }; }
	destroy() {}
}

function entryPointsOf(code)
{
	const out = [];
	const re = /@(compute|vertex|fragment)[\s\S]{0,80}?fn\s+([A-Za-z_0-9]+)/g;
	let m;
	while ((m = re.exec(code))) { out.push(m[2]); }
	return out;
}

function installFacade(target, opts)
{
	const engine = opts.engine;
	const state = {
		configuredGrid: 0, rulesVersion: "", stateVersion: "", frames: 0, steps: 0,
		canvasContext: null
	};

	const queue = {
		writeBuffer(buffer, offset, data, dataOffset, size)
		{
			let src;
			if (data instanceof ArrayBuffer) { src = new Uint8Array(data); }
			else { src = new Uint8Array(data.buffer, data.byteOffset, data.byteLength); }
			if (dataOffset || size) { src = src.subarray(dataOffset || 0, size ? (dataOffset || 0) + size : undefined); }
			buffer.bytes.set(src.subarray(0, Math.min(src.length, buffer.size - offset)), offset);
			buffer.version++;
		},
		submit(commandBuffers)
		{
			for (const cb of commandBuffers) { for (const pass of cb.passes) { execute(pass); } }
		}
	};

	function words(buffer, Ctor) { return new Ctor(buffer.bytes.buffer, buffer.bytes.byteOffset, buffer.bytes.byteLength / Ctor.BYTES_PER_ELEMENT); }

	function bufferAt(group, binding)
	{
		const e = group.entries.find((x) => x.binding === binding);
		return e && e.resource && e.resource.buffer;
	}

	function execute(pass)
	{
		if (pass.kind === "compute" && pass.dispatched)
		{
			// group 0: common (binding 0 = grid vec3f); group 1: state in/out; group 2: offsets x3, survive, born
			const grid = words(bufferAt(pass.groups[0], 0), Float32Array);
			const G = grid[0] >>> 0;
			if (G !== state.configuredGrid) { engine.configure(G); state.configuredGrid = G; state.rulesVersion = ""; state.stateVersion = ""; }
			const rg = pass.groups[2];
			const rb = [0, 1, 2, 3, 4].map((b) => bufferAt(rg, b));
			const rv = rb.map((b) => b.id + ":" + b.version).join("|");
			if (rv !== state.rulesVersion)
			{
				engine.setRules(words(rb[0], Int32Array), words(rb[1], Int32Array), words(rb[2], Int32Array), words(rb[3], Uint32Array), words(rb[4], Uint32Array));
				state.rulesVersion = rv;
			}
			const inBuf = bufferAt(pass.groups[1], 0), outBuf = bufferAt(pass.groups[1], 1);
			const sv = [inBuf, outBuf].map((b) => b.id + ":" + b.version).sort().join("|");
			if (sv !== state.stateVersion)
			{
				// the host wrote the state buffers (initial seed, 1361-1362): that data is the new step-0 state
				engine.uploadState(words(inBuf, Uint32Array));
				state.stateVersion = sv;
			}
			engine.step(1);
			state.steps++;
		}
		else if (pass.kind === "render" && pass.drawn)
		{
			const common = bufferAt(pass.groups[0], 11);
			const target = pass.desc.colorAttachments[0].view.texture;
			const ctx = state.canvasContext;
			const W = target.width, H = target.height;
			if (!ctx.frame || ctx.frame.length !== W * H * 4) { ctx.frame = new Uint8Array(W * H * 4); }
			// the render pass may precede the first compute pass (main_pathtraced.js:1842-1844): make sure the state is there
			ensureStateForRender(pass);
			engine.render(words(common, Float32Array).subarray(0, 128), W, H, 1, { presentation: ctx.frame });
			state.frames++;
		}
	}

	function ensureStateForRender(pass)
	{
		const grid = words(bufferAt(pass.groups[0], 0), Float32Array);
		const G = grid[0] >>> 0;
		if (G !== state.configuredGrid) { engine.configure(G); state.configuredGrid = G; state.rulesVersion = ""; state.stateVersion = ""; }
		const cells = bufferAt(pass.groups[2], 0); // group 2 of the render pass: cell state (1788)
		if (state.stateVersion === "" && cells)
		{
			if (state.rulesVersion === "" && opts.defaultRules) { engine.setRuleStrings(opts.defaultRules); }
			engine.uploadState(words(cells, Uint32Array));
			state.stateVersion = "render:" + cells.id + ":" + cells.version;
		}
	}

	function makePass(kind, desc)
	{
		const pass = { kind, desc, groups: [], pipeline: null, dispatched: false, drawn: false };
		pass.setPipeline = (p) => { pass.pipeline = p; };
		pass.setBindGroup = (i, g) => { pass.groups[i] = g; };
		pass.setVertexBuffer = () => {};
		pass.setIndexBuffer = () => {};
		pass.dispatchWorkgroups = (x, y, z) => { pass.dispatched = true; pass.dispatch = [x, y, z]; };
		pass.drawIndexed = (n) => { pass.drawn = true; pass.indexCount = n; };
		pass.draw = (n) => { pass.drawn = true; pass.indexCount = n; };
		pass.end = () => {};
		return pass;
	}

	const device = {
		limits: { maxComputeWorkgroupsPerDimension: 65535 },
		queue,
		createBuffer: (d) => new FBuffer(d),
		createTexture: (d) => new FTexture(d),
		createSampler: (d) => ({ sampler: d || {} }),
		createShaderModule: (d) => ({ entryPoints: entryPointsOf(d.code || ""), label: d.label }),
		createBindGroupLayout: (d) => ({ layout: d }),
		createPipelineLayout: (d) => ({ pipelineLayout: d }),
		createBindGroup: (d) => ({ label: d.label, entries: d.entries }),
		createRenderPipeline: (d) => ({ kind: "render", fragment: d.fragment && d.fragment.entryPoint }),
		createComputePipeline: (d) => ({ kind: "compute", compute: d.compute && d.compute.entryPoint }),
		createCommandEncoder()
		{
			const passes = [];
			return {
				beginRenderPass(desc) { const p = makePass("render", desc); passes.push(p); return p; },
				beginComputePass(desc) { const p = makePass("compute", desc); passes.push(p); return p; },
				finish() { return { passes }; }
			};
		},
		destroy() {}
	};

	const canvasContext = {
		frame: null,
		configure() {},
		getCurrentTexture() { return new FTexture({ label: "canvas", size: [opts.width, opts.height], format: "rgba8unorm" }); }
	};
	state.canvasContext = canvasContext;

	const gpu = {
		requestAdapter: async () => ({ requestDevice: async () => device, limits: device.limits, features: new Set() }),
		getPreferredCanvasFormat: () => "rgba8unorm"
	};

	target.GPUBufferUsage = BUFFER_USAGE;
	target.GPUShaderStage = SHADER_STAGE;
	target.GPUTextureUsage = TEXTURE_USAGE;
	if (!target.navigator) { target.navigator = {}; }
	target.navigator.gpu = gpu;
	return { gpu, device, canvasContext, state };
}

module.exports = { installFacade, entryPointsOf };
