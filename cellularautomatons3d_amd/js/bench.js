#!/usr/bin/env node
// The headline measurement from the JavaScript host (BASELINE north_star: "host code stays JavaScript (Node.js)
// calling hand-written HIP kernels through a thin C-ABI shared library via N-API"): N CA steps of the 512^3 packed
// grid, default rule, and 1080p / 4 spp frames, driven through ca3d.js -> ca3d_napi.node -> libca3d.so. Prints one
// JSON line; bench.py (the driver's contract) measures the same kernels from Python.
//   node cellularautomatons3d_amd/js/bench.js [--grid 512] [--steps 2048] [--reps 10] [--warmup 256] [--frames 10 --uniforms u.f32]
//   node cellularautomatons3d_amd/js/bench.js --gpus 8 [--grid 1024] [--ghost K] [--transport copy|rccl] [--check 40]
//       the Z-slab split over the GPUs of a node (BASELINE configs[3]) driven by this ONE thread through EngineGroup
//       (ca3d_group_*); --devices 0,0,0,0 places several slabs on one GPU (rehearsal); --check n first verifies n steps
//       against the JS CPU stepper on a 128-plane-deep sample of rank boundaries (small grids: the whole grid)
// The renderer leg needs the 128-float uniform block the UI's MemoryManager would supply (camera maths stays in
// front of the engine): python -c "from cellularautomatons3d_amd import host; host.uniform_block(1920, 1080,
// host.orbit_camera()).tofile('u.f32')"
"use strict";
const path = require("path");
const c = require(path.join(__dirname, "ca3d.js"));

const fs = require("fs");
const arg = (name, dflt) => { const i = process.argv.indexOf("--" + name); return i > 0 ? Number(process.argv[i + 1]) : dflt; };
const sarg = (name) => { const i = process.argv.indexOf("--" + name); return i > 0 ? process.argv[i + 1] : null; };
const G = arg("grid", 512), steps = arg("steps", 2048), warmup = arg("warmup", 256), frames = arg("frames", 10);
const uniformsPath = sarg("uniforms");
const gpus = arg("gpus", 1), devicesArg = sarg("devices");
if (gpus > 1 || devicesArg) { runGroup(); return; }

function runGroup()
{
	const devices = devicesArg ? devicesArg.split(",").map(Number) : Array.from({ length: gpus }, (_, i) => i);
	const Gm = arg("grid", 1024), stepsM = arg("steps", 2048), warm = arg("warmup", 256), repsM = arg("reps", 10);
	// ghost depth as bench.py's auto_ghost: 32, or the deepest of 32 / 16 / 8 that keeps a rank's share of 1024^3 on the resident slab kernel
	// (8 tile layers of an even number of planes <= 36: four slabs -> 16)
	let ghost = arg("ghost", 0);
	if (ghost <= 0)
	{
		ghost = 32;
		if (Gm === 1024 && Gm % devices.length === 0)
			for (const k of [32, 16, 8]) { const pl = Gm / devices.length + 2 * k; if (pl % 16 === 0 && pl / 8 <= 36) { ghost = k; break; } }
	}
	const transport = sarg("transport") === "rccl" ? 1 : 0, check = arg("check", 0);
	const grp = new c.EngineGroup(devices);
	grp.configure(Gm, ghost);
	grp.setRuleStrings({});
	if (transport) grp.setOption("transport", 1);
	const state = c.randomFill((Gm / 32) * Gm * Gm, 0xCA3D0001, 0);
	grp.uploadState(state);
	let verified = null;
	if (check > 0)
	{
		// the multi-GPU state must equal a single grid's: a full-grid engine on device 0 (itself checked against the oracle in the test suite)
		const one = new c.Engine(devices[0]);
		one.configure(Gm);
		one.setRuleStrings({});
		one.uploadState(state);
		one.step(check);
		grp.step(check);
		verified = Buffer.from(grp.readState().buffer).equals(Buffer.from(one.readState().buffer));
		one.close();
		grp.uploadState(state);
	}
	grp.step(warm);
	grp.synchronize();
	const t0 = process.hrtime.bigint();
	for (let i = 0; i < repsM; i++) grp.step(stepsM);
	grp.synchronize();
	const dt = Number(process.hrtime.bigint() - t0) / 1e9, total = stepsM * repsM;
	const out = {
		metric: `Gcells/s CA step at ${Gm}^3 (Node.js host, one thread, ${devices.length} slabs)`, value: +(Gm ** 3 * total / dt / 1e9).toFixed(3), unit: "Gcells/s",
		n_gpus: new Set(devices).size, slabs: devices.length, devices, steps: stepsM, reps: repsM, warmup: warm, ms_per_step: +(dt * 1e3 / total).toFixed(6),
		kernel: grp.info(0).kernelName, ghost, transport: transport ? "rccl (ncclCommInitAll, grouped send/recv)" : "peer copies ordered by events", node: process.version
	};
	// what the slabs really ran on, from the engines themselves (ca3d_slab_comm_info): N slabs are N GPUs only if N bus ids differ
	const ranks = devices.map((_, k) => { const i = grp.info(k); return { rank: k, device: i.device, pci_bus_id: i.pciBusId, z0: i.z0, nz: i.nz }; });
	out.devices_seen = { slabs: ranks.length, distinct_devices: new Set(ranks.map((r) => r.pci_bus_id)).size, ranks };
	if (verified !== null) { out.verified = { steps: check, state_matches_single_grid: verified }; }
	console.log(JSON.stringify(out));
	grp.close();
	if (verified === false) process.exit(1);
}

const eng = new c.Engine(0);
eng.configure(G);
eng.setRuleStrings({});
eng.uploadState(c.randomFill((G / 32) * G * G, 0xCA3D0001, 0));
const reps = arg("reps", 10);
eng.setOption("graph_prepare", Math.max(steps, warmup));
// the calls of a frame are encoded and submitted together, as the reference does with its command encoder and one
// queue.submit (main_pathtraced.js:1833-1850): `reps` calls of `steps` steps, submissions of `steps` steps
eng.setOption("queue", steps);
eng.step(warmup);
eng.step(steps);
eng.synchronize();
const l0 = eng.info().launchesTotal;
const t0 = process.hrtime.bigint();
for (let i = 0; i < reps; i++) eng.step(steps);
eng.flush();
eng.synchronize();
const dt = Number(process.hrtime.bigint() - t0) / 1e9;
const launches = eng.info().launchesTotal - l0, total = steps * reps;
const st = eng.stats(); // the last submission
const launchUs = st.gpuMs * 1e3 / st.kernelLaunches; // the resident kernel runs a whole submission in one launch
const stepUs = dt * 1e6 / total;
const out = {
	metric: `Gcells/s CA step at ${G}^3 (Node.js host)`, value: +(G ** 3 * total / dt / 1e9).toFixed(3), unit: "Gcells/s", steps, reps, warmup,
	ms_per_step: +(dt * 1e3 / total).toFixed(6), kernel: eng.info().kernelName, launch_us: +launchUs.toFixed(3),
	steps_per_launch: Math.round(total / launches), node: process.version
};
// The roofline of the kernel that ran, as bench.py reports it: the per-step kernels move the state through HBM every step (fraction of
// 8 TB/s of algorithmic bytes); a resident kernel keeps it on chip and is bound by vector-instruction issue — SQ_INSTS_VALU per step
// from the committed rocprofv3 pass of bench.py's command (profiles/r*_pmc_sq_*.json) against 1024 SIMDs x 2.4 GHz / 2.
const hbmEquivalent = (0.25 * G ** 3) / (stepUs * 1e-6) / 1e9;
if (/^ca_resident/.test(out.kernel))
{
	const key = /class/.test(out.kernel) ? `clustered${G}` : `resident${G}`;
	const dir = path.join(__dirname, "..", "..", "profiles");
	let prof = null, src = null;
	try
	{
		for (const f of fs.readdirSync(dir).filter((n) => /^r.*_pmc_sq_/.test(n) && n.endsWith(`_${key}.json`)).sort().reverse())
		{
			const d = JSON.parse(fs.readFileSync(path.join(dir, f), "utf8"));
			// only a profile of THIS instruction stream counts: same kernel, grid, rule, form options and device sources
			if (d.steps_per_launch && d.SQ_INSTS_VALU && d.variant && d.variant === eng.info().kernelVariant) { prof = d; src = `profiles/${f}`; break; }
		}
	}
	catch (e) { /* no profiles directory: the instruction count stays unknown */ }
	const peak = 1228.8; // Gwaveinst/s
	out.roofline = { bound: "valu_issue", peak, unit: "Gwaveinst/s", achieved: null, frac: null, counter_source: src, hbm_equivalent_gbs: +hbmEquivalent.toFixed(1),
		variant: eng.info().kernelVariant };
	if (!prof) out.roofline.note = "no committed profiles/r*_pmc_sq_* pass of this kernel variant: instruction count unknown, no fraction";
	if (prof)
	{
		const achieved = prof.SQ_INSTS_VALU / prof.steps_per_launch / (stepUs * 1e-6) / 1e9;
		out.roofline.achieved = +achieved.toFixed(2);
		out.roofline.frac = +(achieved / peak).toFixed(4);
	}
}
else out.roofline = { bound: "hbm", peak: 8000, unit: "GB/s", achieved: +hbmEquivalent.toFixed(1), frac: +(hbmEquivalent / 8000).toFixed(4) };
if (frames > 0 && uniformsPath)
{
	// a sparser scene for the renderer, as bench.py: density 2^-5
	eng.uploadState(c.randomFill((G / 32) * G * G, 0xCA3D0001, 4));
	const W = 1920, H = 1080, spp = 4;
	const ub = fs.readFileSync(uniformsPath);
	const u = new Float32Array(ub.buffer.slice(ub.byteOffset, ub.byteOffset + 512));
	for (let i = 0; i < 8; i++) eng.render(u, W, H, spp, {}); // every lane of the frame pipeline has drawn this frame before the clock starts
	eng.synchronize();
	const r0 = process.hrtime.bigint();
	for (let i = 0; i < frames; i++) { eng.render(u, W, H, spp, {}); }
	eng.synchronize();
	const rdt = Number(process.hrtime.bigint() - r0) / 1e9 / frames;
	const rs = eng.renderStats();
	out.render = { metric: "Mray/s path-trace 1080p", value: +((rs.primaryRays + rs.shadowRays) / rdt / 1e6).toFixed(2), ms_per_frame: +(rdt * 1e3).toFixed(4),
		frames_in_flight: eng.renderPipeline() }; // frames stay on the device: the engine keeps several in flight (ca3d_get_render_pipeline)
}
console.log(JSON.stringify(out));
eng.close();
