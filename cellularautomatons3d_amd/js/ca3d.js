/*
 * JavaScript host of the MI355X engine: the rule / grid / step surface of the reference's MainModule
 * (main_pathtraced.js) over the N-API addon (addon/ca3d_napi.c -> include/ca3d.h). CommonJS, Node >= 12.
 *
 * Host-side helpers mirror the reference's own (same names without the underscore, same quirks): they are the
 * JS twin of cellularautomatons3d_amd/host.py and are checked against the same captured fixtures.
 * There is no CPU stepping here: without the addon and a GPU, `new Engine()` throws.
 */
"use strict";
const path = require("path");

const NEIGHBOURS_STORAGE_LEN = 27; // main_pathtraced.js:10
const WORK_GROUP_SIZE = 16; // main_pathtraced.js:5
const LAYOUT_PACKED32 = 0;
const LAYOUT_UNPACKED = 1;

// main_pathtraced.js:13-94
const vn = [1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1];
const vn2d = vn.slice(0, 12);
const moore2d = vn2d.concat([1, 1, 0, -1, 1, 0, 1, -1, 0, -1, -1, 0]);
const layer = (z) => [1, 0, z, -1, 0, z, 0, 1, z, 0, -1, z, 1, 1, z, -1, 1, z, 1, -1, z, -1, -1, z, 0, 0, z];
const NEIGHBOURHOOD_MAP = {
	"moore": new Int32Array(moore2d.concat(layer(1), layer(-1))),
	"moore 2D": new Int32Array(moore2d),
	"von neumann": new Int32Array(vn),
	"von neumann 2D": new Int32Array(vn2d),
	"edges": new Int32Array([1, 1, 0, -1, 1, 0, 0, 1, 1, 0, 1, -1, 1, -1, 0, -1, -1, 0, 0, -1, 1, 0, -1, -1, 1, 0, 1, -1, 0, 1, 1, 0, -1, -1, 0, -1]),
	"corners": new Int32Array([1, 1, 1, -1, 1, 1, 1, 1, -1, -1, 1, -1, 1, -1, 1, -1, -1, 1, 1, -1, -1, -1, -1, -1])
};

// _rulesComponentsToValues (main_pathtraced.js:554-581); NaN components are dropped where the reference lets its
// typed-array store ignore them.
function rulesComponentsToValues(rulesComponents)
{
	const result = [];
	const components = rulesComponents.split(" ").join("").split(",");
	for (let i = 0; i < components.length; i++)
	{
		if (components[i].indexOf("-") > -1)
		{
			const range = components[i].split("-");
			const start = parseInt(range[0], 10);
			const end = parseInt(range[1], 10);
			for (let j = start; j <= end; j++) { result.push(Math.min(j, 26)); }
		}
		else
		{
			const v = Math.min(parseInt(components[i], 10), 26);
			if (!Number.isNaN(v)) { result.push(v); }
		}
	}
	return result;
}

// _recalculateRulesValues (583-622) -> { born: Uint32Array(81), survive: Uint32Array(81) }
function recalculateRulesValues(r)
{
	const rulesets = [r.born, r.survive, r.bornEdges, r.surviveEdges, r.bornCorners, r.surviveCorners].map(rulesComponentsToValues);
	const born = new Uint32Array(NEIGHBOURS_STORAGE_LEN * 3);
	const survive = new Uint32Array(NEIGHBOURS_STORAGE_LEN * 3);
	let offset = 0;
	for (let i = 0; i < rulesets.length; i += 2)
	{
		for (const v of rulesets[i]) { born[v + offset] = 1; }
		for (const v of rulesets[i + 1]) { survive[v + offset] = 1; }
		offset += NEIGHBOURS_STORAGE_LEN;
	}
	return { born, survive };
}

const DEFAULT_RULES = { neighbourhood: "von neumann", born: "1,3", survive: "0-6", bornEdges: "27", surviveEdges: "27", bornCorners: "27", surviveCorners: "27" };

// _gridSizeUIFormatter (675-693)
function gridSizeUIFormatter(v)
{
	let out = v;
	const m = v % 32;
	if (m > 0) { out = m <= 16 ? v - m : v - m + 32; }
	return out;
}

// _getClusterIdxFromGridCoordinates (1170-1178)
function getClusterIdxFromGridCoordinates(gridSize, c)
{
	const cols = gridSize / 32;
	return (Math.floor(c.x / 32) % cols) + (c.y % gridSize) * cols + (c.z % gridSize) * cols * gridSize;
}

// initial state of _setupStorageBuffers (1241-1297); `random` replaces the unseeded Math.random
function initialState(gridSize, randomInitialState, random)
{
	if (!(gridSize > 0) || gridSize % 32) { throw new RangeError("grid size must be a positive multiple of 32"); }
	const data = new Uint32Array((gridSize / 32) * gridSize * gridSize);
	const center = Math.floor(gridSize * 0.5) - 1;
	if (randomInitialState)
	{
		const rnd = random || mulberry32(0xCA3D0001);
		for (let i = -2; i < 3; i++) for (let j = -2; j < 3; j++) for (let k = -2; k < 3; k++)
		{
			const idx = getClusterIdxFromGridCoordinates(gridSize, { x: center + i, y: center + j, z: center + k });
			if (rnd() > 0.5) { data[idx] = data[idx] | (1 << center + i); }
			else { data[idx] = data[idx] & ~(1 << center + i); }
		}
	}
	else
	{
		data[getClusterIdxFromGridCoordinates(gridSize, { x: center, y: center, z: center })] = 1 << (center % 32);
	}
	return data;
}

function mulberry32(a)
{
	return function () { a |= 0; a = a + 0x6D2B79F5 | 0; let t = Math.imul(a ^ a >>> 15, 1 | a); t = t + Math.imul(t ^ t >>> 7, 61 | t) ^ t; return ((t ^ t >>> 14) >>> 0) / 4294967296; };
}

function dispatchShape(gridSize)
{
	const wg = Math.ceil(gridSize / WORK_GROUP_SIZE);
	return [gridSize / 32, wg, wg]; // main_pathtraced.js:1805-1806
}

// counter-based synthetic fill shared with host.py / the oracle (SURVEY 8(d))
function randomFill(nWords, seed, andRounds)
{
	seed = seed === undefined ? 0xCA3D0001 : seed;
	const mix = (i, r) => { let x = (Math.imul(i, 0x9E3779B9) + seed + Math.imul(r, 0x85EBCA6B)) >>> 0; x ^= x >>> 16; x = Math.imul(x, 0x7FEB352D) >>> 0; x ^= x >>> 15; x = Math.imul(x, 0x846CA68B) >>> 0; x ^= x >>> 16; return x >>> 0; };
	const out = new Uint32Array(nWords);
	for (let i = 0; i < nWords; i++)
	{
		let w = mix(i, 0);
		for (let r = 1; r <= (andRounds || 0); r++) { w &= mix(i, r); }
		out[i] = w;
	}
	return out;
}

// Checkpoint file: 'CA3D' | u32 version | u32 grid | u32 layout | u64 step | u64 words | LE u32 words (host.py twin)
function saveCheckpoint(file, words, gridSize, step, layout)
{
	const fs = require("fs");
	const head = Buffer.alloc(32);
	head.write("CA3D", 0, "latin1");
	head.writeUInt32LE(1, 4); head.writeUInt32LE(gridSize, 8); head.writeUInt32LE(layout || 0, 12);
	head.writeBigUInt64LE(BigInt(step || 0), 16); head.writeBigUInt64LE(BigInt(words.length), 24);
	fs.writeFileSync(file, Buffer.concat([head, Buffer.from(words.buffer, words.byteOffset, words.byteLength)]));
}

function loadCheckpoint(file)
{
	const fs = require("fs");
	const b = fs.readFileSync(file);
	if (b.length < 32 || b.toString("latin1", 0, 4) !== "CA3D" || b.readUInt32LE(4) !== 1) { throw new Error("not a CA3D checkpoint"); }
	const gridSize = b.readUInt32LE(8), layout = b.readUInt32LE(12), step = Number(b.readBigUInt64LE(16)), n = Number(b.readBigUInt64LE(24));
	if (b.length !== 32 + 4 * n) { throw new Error("truncated checkpoint"); }
	const words = new Uint32Array(n);
	Buffer.from(words.buffer).set(b.subarray(32));
	return { words, gridSize, layout, step };
}

let addon = null;
function loadAddon()
{
	if (!addon)
	{
		try { addon = require(path.join(__dirname, "ca3d_napi.node")); }
		catch (e) { throw new Error("ca3d_napi.node is missing or unloadable (build: make -C cellularautomatons3d_amd/js/addon): " + e.message + " — this engine has no CPU fallback"); }
	}
	return addon;
}

class Engine
{
	constructor(device)
	{
		this._a = loadAddon();
		this._h = this._a.create(device || 0);
		this.gridSize = 0;
	}

	// An asynchronous job holds the engine on a worker thread until its promise settles: nothing else may enter the engine
	// meanwhile (it is not thread-safe), and it must not be destroyed under the worker.
	_idle(what)
	{
		if (this._inflight) { throw new Error("ca3d: " + what + "() while an asynchronous call is pending on this engine — await it first"); }
	}
	close() { this._idle("close"); if (this._h) { this._a.destroy(this._h); this._h = null; } }
	/** close() once every asynchronous call issued so far has settled. */
	closeAsync() { return (this._pending || Promise.resolve()).then(() => this.close()); }

	configure(gridSize, layout) { this._idle("configure"); this._a.configure(this._h, gridSize, layout || LAYOUT_PACKED32); this.gridSize = gridSize; }

	setRules(mainOffsets, edgesOffsets, cornersOffsets, survive, born) { this._idle("setRules"); this._a.setRules(this._h, mainOffsets, edgesOffsets, cornersOffsets, survive, born); }

	setRuleStrings(rules)
	{
		const r = Object.assign({}, DEFAULT_RULES, rules || {});
		const lut = recalculateRulesValues(r);
		this.setRules(NEIGHBOURHOOD_MAP[r.neighbourhood], NEIGHBOURHOOD_MAP["edges"], NEIGHBOURHOOD_MAP["corners"], lut.survive, lut.born);
	}

	// _restartSim (624-637)
	restartSim(gridSize, rules, randomInitialState, random)
	{
		this.configure(gridSize);
		this.setRuleStrings(rules);
		this.uploadState(initialState(gridSize, randomInitialState, random));
	}

	uploadState(words) { this._idle("uploadState"); this._a.uploadState(this._h, words); }

	readState()
	{
		this._idle("readState");
		const out = new Uint32Array(this.info().stateWords);
		this._a.readState(this._h, out);
		return out;
	}

	// _computePass (1796-1809), n times
	step(n) { this._idle("step"); this._a.step(this._h, n === undefined ? 1 : n); }
	/** device.queue.submit: submits the steps encoded under setOption("queue", n). */
	flush() { this._idle("flush"); this._a.flush(this._h); }

	// Z-slab mode (multi-GPU hosts; SURVEY 8(e)): see include/ca3d.h. phase: 0 whole batch, 1 edge zones, 2 interior.
	configureSlab(gridSize, z0, nz, ghost, layout) { this._a.configureSlab(this._h, gridSize, layout === undefined ? LAYOUT_PACKED32 : layout, z0, nz, ghost); }
	slabStep(n) { this._a.slabStep(this._h, n); }
	slabStepPhase(n, phase) { this._a.slabStepPhase(this._h, n, phase); }
	// RCCL transport inside the engine: one process per GPU; rank 0 creates the id (Engine.commUniqueId()) and hands it to
	// the other ranks by the host's own IPC; slabRun(n, overlap) = n steps with the ghost planes exchanged between batches
	static commUniqueId() { return loadAddon().commUniqueId(); }
	slabCommInit(id, rank, world) { this._a.slabCommInit(this._h, id, rank, world); }
	slabRun(n, overlap) { this._a.slabRun(this._h, n, overlap ? 1 : 0); }
	slabExchange() { this._a.slabExchange(this._h); }
	slabGather(full) { this._a.slabGather(this._h, full._h); }

	synchronize() { this._idle("synchronize"); this._a.synchronize(this._h); }

	// _renderPass (1775-1794) with the reference's 128-float block (MemoryManager.bufferf32)
	render(uniforms, width, height, spp, targets)
	{
		const t = targets || {};
		this._idle("render");
		this._a.render(this._h, uniforms, width, height, spp || 1, t.presentation || null, t.light || null, t.depth || null);
	}

	// Asynchronous forms (napi_async_work): the wait for the GPU runs on a worker thread and the call returns a Promise, so
	// a UI thread never blocks in a read-back. The engine takes one call at a time: do not call anything else on it until
	// the promise has settled (asynchronous calls issued meanwhile queue up behind it by themselves).
	_queue(start)
	{
		this._inflight = (this._inflight || 0) + 1;
		const done = () => { this._inflight--; };
		const run = () => { const inflight = this._inflight; this._inflight = 0; try { return start(); } finally { this._inflight = inflight; } };
		const p = (this._pending || Promise.resolve()).then(run, run);
		this._pending = p.then(done, done);
		return p;
	}
	readStateAsync()
	{
		return this._queue(() => {
			const out = new Uint32Array(this.info().stateWords);
			return this._a.readStateAsync(this._h, out).then(() => out);
		});
	}
	renderAsync(uniforms, width, height, spp, targets)
	{
		const t = targets || {};
		return this._queue(() => this._a.renderAsync(this._h, uniforms, width, height, spp || 1, t.presentation || null, t.light || null, t.depth || null).then(() => t));
	}
	synchronizeAsync() { return this._queue(() => this._a.synchronizeAsync(this._h)); }

	info() { this._idle("info"); return this._a.info(this._h); }
	stats() { this._idle("stats"); return this._a.stats(this._h); }
	renderStats() { this._idle("renderStats"); return this._a.renderStats(this._h); }
	renderPipeline() { this._idle("renderPipeline"); return this._a.renderPipeline(this._h); } // converged frames in flight (option render_pipeline)
	setOption(name, value) { this._idle("setOption"); this._a.setOption(this._h, name, value); }
	/** resident launches that timed out and were re-run through the per-step kernels (include/ca3d.h) */
	recoveredLaunches() { this._idle("recoveredLaunches"); return this._a.recoveredLaunches(this._h); }
}

// ca3d_group_*: the Z-slab split of a grid over the GPUs of a node, driven by this one JavaScript thread (the reference's host
// is one thread that enqueues everything, main_pathtraced.js:1821-1854). `devices`: one GPU index per slab, in z order; the
// same index may repeat (several slabs on one GPU). Same rule / state / step / render surface as Engine, on the FULL grid.
class EngineGroup
{
	constructor(devices)
	{
		this._a = loadAddon();
		this.devices = Int32Array.from(devices);
		this._g = this._a.groupCreate(this.devices);
		this.gridSize = 0;
	}
	close() { if (this._g) { this._a.groupDestroy(this._g); this._g = null; } }
	/** ghost = planes kept of each neighbour = steps between two exchanges */
	configure(gridSize, ghost, layout) { this._a.groupConfigure(this._g, gridSize, layout || LAYOUT_PACKED32, ghost); this.gridSize = gridSize; }
	setRules(mainOffsets, edgesOffsets, cornersOffsets, survive, born) { this._a.groupSetRules(this._g, mainOffsets, edgesOffsets, cornersOffsets, survive, born); }
	setRuleStrings(rules)
	{
		const r = Object.assign({}, DEFAULT_RULES, rules || {});
		const lut = recalculateRulesValues(r);
		this.setRules(NEIGHBOURHOOD_MAP[r.neighbourhood], NEIGHBOURHOOD_MAP["edges"], NEIGHBOURHOOD_MAP["corners"], lut.survive, lut.born);
	}
	uploadState(words) { this._a.groupUploadState(this._g, words); this._words = words.length; }
	readState() { const out = new Uint32Array(this._words); this._a.groupReadState(this._g, out); return out; }
	step(n) { this._a.groupStep(this._g, n === undefined ? 1 : n); }
	synchronize() { this._a.groupSynchronize(this._g); }
	/** "transport": 0 peer copies (default), 1 RCCL; any other option goes to every slab engine */
	setOption(name, value) { this._a.groupSetOption(this._g, name, value); }
	info(rank) { return this._a.groupInfo(this._g, rank || 0); }
	render(uniforms, width, height, spp, targets)
	{
		const t = targets || {};
		this._a.groupRender(this._g, uniforms, width, height, spp || 1, t.presentation || null, t.light || null, t.depth || null);
	}
}

module.exports = {
	Engine, EngineGroup, NEIGHBOURHOOD_MAP, DEFAULT_RULES, LAYOUT_PACKED32, LAYOUT_UNPACKED, NEIGHBOURS_STORAGE_LEN,
	rulesComponentsToValues, recalculateRulesValues, gridSizeUIFormatter, getClusterIdxFromGridCoordinates,
	initialState, dispatchShape, randomFill, loadAddon, saveCheckpoint, loadCheckpoint
};
