// C-ABI layer (include/ca3d.h): engine object, device buffers, ping-pong stepping, slab sub-steps, stats.
// Replaces the WebGPU calls of main_pathtraced.js listed per entry point in the header. No CPU fallback.
#include <cstdarg>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <map>
#include <new>
#include <stdexcept>
#include <string>

#include "ca3d_internal.h"

using namespace ca3d;

namespace
{

// the message slot of ca3d_last_error(): a fixed buffer per thread — setting it never allocates, so it can be set while reporting
// std::bad_alloc (a std::string here could throw from inside the handler that reports the failure)
thread_local char g_last_error[1024] = "";

void set_last_error(const char *msg) noexcept
{
	if (!msg) msg = "";
	size_t n = strlen(msg);
	if (n >= sizeof g_last_error) n = sizeof g_last_error - 1;
	memmove(g_last_error, msg, n); // (msg may point into the slot itself)
	g_last_error[n] = 0;
}

int fail(int code, const char *fmt, ...)
{
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	set_last_error(buf);
	return code;
}

#define HIP_TRY(expr)                                                                                          \
	do                                                                                                         \
	{                                                                                                          \
		hipError_t e_ = (expr);                                                                                \
		if (e_ != hipSuccess)                                                                                  \
			return fail(e_ == hipErrorOutOfMemory ? CA3D_ERR_OUT_OF_MEMORY : CA3D_ERR_DEVICE, "%s: %s", #expr, \
			            hipGetErrorString(e_));                                                                \
	} while (0)

} // namespace

namespace ca3d
{
// for the other translation units of the library (ca3d_group.cpp): same message slot as every entry point here
int set_error(int code, const char *fmt, ...)
{
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	set_last_error(buf);
	return code;
}

// Every entry point of include/ca3d.h is a function-try-block (CA3D_API_TRY ... CA3D_API_CATCH, ca3d_internal.h) whose handler lands
// here: the header promises status codes, and an exception leaving an extern "C" function called from Node.js / ctypes ends the
// process. What can throw inside the library is host allocation (std::string / std::vector / std::map in the rule canonicaliser,
// the run-time compiler, the group) and whatever the standard library reports as std::exception.
int exception_to_status() noexcept
{
	try
	{
		throw;
	}
	catch (const std::bad_alloc &)
	{
		set_last_error("out of host memory (std::bad_alloc inside the library)");
		return CA3D_ERR_OUT_OF_MEMORY;
	}
	catch (const std::exception &e)
	{
		char buf[1024];
		snprintf(buf, sizeof buf, "internal error: %s", e.what());
		set_last_error(buf);
		return CA3D_ERR_DEVICE;
	}
	catch (...)
	{
		set_last_error("internal error: unknown C++ exception inside the library");
		return CA3D_ERR_DEVICE;
	}
}
} // namespace ca3d

struct ca3d_engine
{
	int device = 0;
	hipStream_t own_stream = nullptr;
	hipStream_t stream = nullptr; // active (own or caller's)
	hipEvent_t ev_start = nullptr, ev_stop = nullptr;
	bool ev_valid = false;

	bool configured = false;
	uint32_t G = 0;
	int layout = CA3D_LAYOUT_PACKED32;
	bool slab = false;
	uint32_t z0 = 0, nz = 0, ghost = 0;
	uint32_t nplanes = 0;   // planes per buffer including ghosts
	size_t plane_words = 0; // u32 per z-plane
	uint32_t *buf[2] = {nullptr, nullptr};
	bool has_state = false;
	bool binary_state = false; // unpacked layout: every cell is 0 or 1 (checked on upload, true after any step)
	uint64_t step = 0;
	uint32_t cur = 0; // physical buffer holding the current state (== step % 2 whenever control returns to the caller)

	CanonRules rules;
	int variant = 0;
	int use_graph = 1;
	int render_mode = 0;
	int render_indirect = 0; // add calculateIndirectLighting (pathtraced_fragment_clustered.wgsl:307-377; commented out at the reference's call site)
	int render_sched = 1; // dynamic ray scheduling in the converged-frame renderer (render.hip); 0: one pixel per lane, in order
	uint32_t render_row0 = 0, render_row1 = 0; // rows [row0, row1) of the frame are rendered (0, 0: all): a rank's band
	int use_fused = 0; // the two-step fused kernel is bit-exact but measured slower than two single steps

	// captured batches of full-grid steps, keyed by (steps in the batch, buffer it starts from); invalidated on any
	// change of rules, kernels, stream or buffers
	struct StepGraph { hipGraphExec_t exec = nullptr; uint32_t launches = 0; };
	std::map<uint64_t, StepGraph> step_graphs;
	// Shorter batches are launched kernel by kernel: measured at 512^3 (tools/step_gap.py) back-to-back 20-step batches
	// run 6.48 us per step as graphs and 6.13 launched one by one, 64-step batches 5.96 / 5.85, 256-step batches 5.78 / 5.76
	// — a graph launch has a start-up and a completion cost of its own, worth paying once the host would fall behind.
	uint32_t graph_min = 128;
	int want_stats = 1; // record the event pair ca3d_get_stats reads (a marker packet each: costs ~1 us of GPU idle per call)
	std::map<uint64_t, hipGraphExec_t> slab_graphs; // (phase, start buffer, sub-steps) -> captured slab batch
	uint32_t pending_edges = 0;                     // sub-steps of an edge phase awaiting its interior phase
	int roll_z = 0;       // forced planes per thread of the rolling-window kernel (0: automatic)
	int roll_tile = 1;    // tile form of the rolling-window kernel (x-shifted rows shared through LDS)
	int use_roll = 1;     // rolling-window form of the class kernels where it applies (needs use_jit)
	int use_jit = 1;      // specialise kernels for the rule at run time (hiprtc) where a specialisation exists
	VnJit vn_jit;         // valid when vn_jit.cvl >= 0
	ClassJit class_jit;   // valid when class_jit.main >= 0
	RowsJit rows_jit;     // valid when rows_jit.main >= 0: the rows kernel for this grid and these rules
	int use_rows = 1;     // option "rows"
	RollJit roll_jit;     // valid when roll_jit.cvl >= 0
	std::string jit_log;  // why the last specialisation attempt failed (empty: none failed)

	// resident multi-step kernel (ca_resident.hip): face mailboxes, status word (device + pinned host copy), tag counter
	int use_resident = 1;
	bool res_ready = false;       // the current rules / grid have a resident kernel
	bool res_class = false;       // ... and it is the class form (ca_resident_class_kernel.inc)
	bool res_failed = false;      // a launch timed out: the path stays off until the next configure
	bool res_check = false;       // a resident launch has been issued since the status was last looked at
	void *res_jit_fn = nullptr;   // run-time compiled kernel for the current tables (null: the pre-built rule)
	void *res_slab_fn = nullptr;  // slab form for the current slab geometry and tables (run-time compiled), or null
	size_t res_mail_bytes = 0;
	unsigned long long *res_mail = nullptr;
	uint32_t *res_status = nullptr, *res_status_host = nullptr;
	uint32_t res_epoch = 0;
	uint32_t res_min = 8;                 // shorter batches take the per-step kernels
	uint32_t queue_max = 0;               // > 0: ca3d_step calls are encoded and submitted together (option "queue")
	uint32_t queued = 0;                  // steps encoded, not yet submitted
	uint64_t launches_total = 0;          // kernel launches the step calls issued since ca3d_create
	bool res_pair = true;                 // 512^3 von Neumann form: the row-pair kernel (option "resident_pair"; 2.48 against 2.52 us per step)
	uint32_t res_rows = 32;               // rows per tile of the von Neumann form (ca_resident_kernel.inc: 32 or 16)
	uint32_t res_zsplit = 1;              // thread groups along z of the von Neumann form (option "resident_zsplit"; 2 = twice the threads, four waves per
	                                      // SIMD: measured SLOWER with 32-row tiles — 2.61 vs 2.52 us per step at 512^3, 1.37 vs 1.26 at 256^3 — and faster
	                                      // only with 16-row tiles, 2.89 vs 3.31: profiles/r3_l_resident_zsplit.txt)
	uint32_t res_timeout_ticks = 20000000; // 200 ms of s_memrealtime per wait
	// Recovery of a resident launch that gave up (full-grid engines). A launch of n >= 2 steps never writes the buffer it reads:
	// the final state goes to a third buffer (`spare`), the state one step earlier to the other ping-pong buffer, and the three
	// pointers rotate, so that buf[step % 2] / buf[(step + 1) % 2] keep the reference's meaning. Launches whose completion the
	// host has not looked at yet are remembered; when one of them timed out (it, and every launch queued behind it, wrote
	// nothing: ca_resident_kernel.inc res_must_skip) the engine goes back to that launch's input and runs all their steps
	// through the per-step kernels.
	uint32_t *spare = nullptr;
	struct ResPending { uint32_t epoch0, n, cur_before; uint64_t step_before; uint32_t *in, *other, *spare; };
	std::vector<ResPending> res_pending;
	uint32_t res_fault_tile = 0;          // option "resident_fault_tile": applies to the next resident launch only
	uint32_t res_recovered = 0;           // launches recovered from since ca3d_create
	std::string res_note;                 // why the resident path is off although the rules / grid have a resident kernel

	// halo transport inside the engine (RCCL, loaded on first use): communicator over the ranks of the slab chain, a second
	// stream so that an exchange can run under the interior phase, the events that order the two
	void *comm = nullptr; // ncclComm_t
	int comm_rank = 0, comm_world = 0;
	hipStream_t comm_stream = nullptr;
	hipEvent_t ev_edges = nullptr, ev_comm = nullptr, ev_gather = nullptr;
	bool ghosts_valid = false; // the ghost planes hold the neighbours' planes of the current step
	int comm_graph = 0;        // capture batch + exchange into one graph (unsplit batches)
	std::map<uint64_t, hipGraphExec_t> comm_graphs;

	ca3d_stats stats{};
	const char *kernel_name = "";

	// renderer targets: presentation + two history pairs (light RGBA16F, depth RG16F), swapped per frame
	uint32_t rw = 0, rh = 0;
	uint32_t *r_present = nullptr;
	void *r_light[2] = {nullptr, nullptr};
	uint32_t *r_depth[2] = {nullptr, nullptr};
	unsigned long long *r_counters = nullptr;
	size_t r_counter_words = 0;
	unsigned long long *r_occ = nullptr; // block-occupancy bits of the current state + count, rebuilt per frame (render.hip)
	size_t r_occ_words = 0;
	int render_skip = 1; // empty-space skipping on sparse volumes
	int render_stream = 1; // dense packed volumes: the ray-stream passes (render_stream.hip) instead of the in-wave scheduled kernel
	int render_stream_check = 0; // diagnostics: count filter / slab-test contradictions (ca3d_get_render_stats is unchanged; see "render_stream_contradictions")
	int render_frame_bricks = 1; // literal frame mode: the batched march over a bricked copy of the volume (render_frame.hip); 0: ca_render_frame_packed
	uint32_t *r_bricks = nullptr;
	size_t r_bricks_bytes = 0;
	// what the renderer's derived buffers (occupancy bits, bricks) were last built from: serial (bumped by everything that writes the state
	// other than a step: uploads, buffers handed out, gathers), step count, buffer
	uint64_t state_serial = 1, r_occ_key[3] = {0, 0, 0}, r_bricks_key[3] = {0, 0, 0};
	// ca3d_device_buffer handed a pointer out: until the call that ends its validity (step / upload / configure) the caller may write the
	// state at any time without telling the engine, so no frame may reuse what an earlier frame derived from it
	bool buffers_exposed = false;
	void *r_stream = nullptr;    // scratch of the stream passes
	size_t r_stream_bytes = 0;
	int r_swap = 0;
	hipEvent_t rev_start = nullptr, rev_stop = nullptr;
	hipStream_t r_aux = nullptr;             // renderer: the plain kernel around the volume's screen rectangle runs here, beside the scheduled launch
	hipEvent_t r_fork = nullptr, r_join = nullptr;
	bool rev_valid = false;
	ca3d_render_stats rstats{};
	// Converged frames in flight (option "render_pipeline", default 1: four of them up to 24 M samples a frame, three above). A frame's two persistent walk launches each end in a tail with
	// most of the chip idle (render_stream.hip: a third to a half of a 1080p launch) and its passes depend on each other — but not on
	// the frame before: a converged frame has no history. Frames that stay on the device (no host pointers) and are drawn by the stream
	// passes alternate between LANES — a stream, scratch, counters and events each — so that the next frames' walks
	// fill the tails of this one's. A lane waits for the engine's stream at the moment of the call (steps, uploads before the frame);
	// the engine's stream waits for the lanes LAZILY: the next call that touches the state, a target or the stream joins them
	// (bind_device). The presentation surface is shared: a frame's pixel-writing kernels wait for the frame before (RenderLaunch::after).
	// Only on the engine's own stream: a caller who set a stream of their own expects every frame ordered on it.
	static constexpr int kMaxLanes = 4;
	struct FrameLane
	{
		hipStream_t s = nullptr;
		hipEvent_t start = nullptr, stop = nullptr, done = nullptr;
		bool need_state = true; // the engine's stream has had work since this lane's last frame: wait for ev_state first
		void *scratch = nullptr;
		size_t scratch_bytes = 0;
		unsigned long long *counters = nullptr;
		bool pending = false; // frames on this lane the engine's stream has not been made to wait for
		bool used = false;    // `done` has been recorded at least once
	} lanes[kMaxLanes];
	int n_lanes = 0; // lanes created (streams on pairwise different hardware queues); 0: not tried yet
	hipEvent_t ev_state = nullptr; // "everything the engine's stream held when the frame was asked for"
	bool main_touched = true;      // an entry point other than a pipelined ca3d_render has run since ev_state was recorded (bind_device)
	bool state_touched = true;     // an entry point other than ca3d_render has run since the last frame: the next frame is not pipelined (ca3d_render)
	std::vector<hipStream_t> lane_spares; // streams that turned out to share a hardware queue with lane 0 (kept: destroying one hands its queue to the next)
	int render_pipeline = 1; // 0: off; 1: the default depth (render_default_lanes: by frame size); 2 .. kMaxLanes: that many
	int lane_next = 0;
	int lanes_in_use = 0; // depth of the last pipelined frame (ca3d_get_render_pipeline)
	bool lanes_exhausted = false; // the probe found fewer side-by-side streams than asked for
	int last_lane = -1; // the lane of the last frame (-1: it went down the engine's stream) — whose events and counters ca3d_get_render_stats reads

	size_t buffer_words() const { return plane_words * nplanes; }
	size_t state_words() const { return plane_words * (slab ? nz : G); }
	double cells_per_plane() const { return (double)G * G; }
	double bytes_per_cell_step() const { return layout == CA3D_LAYOUT_PACKED32 ? 0.25 : 8.0; }
};

namespace
{

void drop_graph(ca3d_engine *h)
{
	for (auto &kv : h->step_graphs) hipGraphExecDestroy(kv.second.exec);
	h->step_graphs.clear();
	for (auto &kv : h->slab_graphs) hipGraphExecDestroy(kv.second);
	h->slab_graphs.clear();
	for (auto &kv : h->comm_graphs) hipGraphExecDestroy(kv.second);
	h->comm_graphs.clear();
}

void free_render_targets(ca3d_engine *h)
{
	if (h->r_present) hipFree(h->r_present);
	for (int i = 0; i < 2; i++)
	{
		if (h->r_light[i]) hipFree(h->r_light[i]);
		if (h->r_depth[i]) hipFree(h->r_depth[i]);
		h->r_light[i] = nullptr;
		h->r_depth[i] = nullptr;
	}
	h->r_present = nullptr;
	h->rw = h->rh = 0;
}

void free_resident(ca3d_engine *h)
{
	if (h->res_mail) hipFree(h->res_mail);
	if (h->res_status) hipFree(h->res_status);
	if (h->res_status_host) hipHostFree(h->res_status_host);
	h->res_mail = nullptr;
	h->res_status = h->res_status_host = nullptr;
	h->res_epoch = 0;
	h->res_check = false;
	h->res_failed = false;
	h->res_pending.clear();
	h->res_note.clear();
}

void free_buffers(ca3d_engine *h)
{
	drop_graph(h);
	free_resident(h);
	if (h->spare) hipFree(h->spare);
	h->spare = nullptr;
	for (int i = 0; i < 2; i++)
	{
		if (h->buf[i]) hipFree(h->buf[i]);
		h->buf[i] = nullptr;
	}
	h->configured = false;
	h->has_state = false;
	h->step = 0;
	h->cur = 0;
	h->pending_edges = 0; // an edge phase belongs to the state that has just gone
	h->ghosts_valid = false;
}

// the engine's stream waits for the frames in flight on the lanes (nothing is waited for on the host)
// Converged frames in flight when option render_pipeline is 1, and the share of the chip's wave slots each frame's persistent walk launches
// ask for while other frames are in flight beside it. Measured on the bench's dense 512^3 scene (tools/sweep_stream_wgs.sh, ms per frame;
// lanes x share): 1080p 4 spp   3 x 100 % 0.485 | 3 x 34 % 0.393 | 4 x 25 % 0.367 | 4 x 17 % 0.397      (one frame at a time: 0.627)
//                 2560 x 1440   3 x 100 % 0.746 | 3 x 34 % 0.641 | 4 x 25 % 0.612
//                 3200 x 1800   3 x 100 % 1.054 | 3 x 34 % 0.968 | 4 x 25 % 0.936
//                 3840 x 2160   2 x 100 % 1.386 | 2 x 50 % 1.35-1.40 | 3 x 100 % 1.389 | 3 x 67 % 1.37-1.38 | 3 x 34 % 1.419 | 4 x 25 % 1.466   (one at a time: 1.539)
// — frames whose walks run SIDE BY SIDE on equal shares of the chip beat frames that fill the chip one after the other and overlap only
// tail to head; how many of them depends on the frame: small frames want many narrow ones (their walks are short against their tails),
// at 3840 x 2160 (33 M samples, 0.7 GB of scratch per frame in flight) narrow walks only get in the way of the frame's other, full-width
// passes and nothing is more than 2 % from anything else.
// So: up to 24 M samples a frame four frames on a quarter of the chip each, above that three on two thirds each.
// CA3D_RENDER_LANES=2..4 / CA3D_STREAM_WGS_PCT (tuning) override both.
int render_default_lanes(size_t samples)
{
	static const int env = getenv("CA3D_RENDER_LANES") ? atoi(getenv("CA3D_RENDER_LANES")) : 0;
	if (env >= 2 && env <= ca3d_engine::kMaxLanes) return env;
	return samples <= (24u << 20) ? 4 : 3;
}
int render_walk_share(size_t samples, int lanes) { return lanes < 2 ? 100 : (samples <= (24u << 20) ? 100 / lanes : 67); }

int join_frames(ca3d_engine *h)
{
	for (auto &L : h->lanes)
		if (L.pending)
		{
			HIP_TRY(hipStreamWaitEvent(h->stream, L.done, 0));
			L.pending = false;
		}
	return CA3D_OK;
}

int bind_device(ca3d_engine *h, bool join = true)
{
	HIP_TRY(hipSetDevice(h->device));
	if (join) h->main_touched = h->state_touched = true; // (whatever the caller is about to put on the engine's stream: the next pipelined frame waits for it)
	if (join)
		for (auto &L : h->lanes)
			if (L.pending) return join_frames(h);
	return CA3D_OK;
}

int allocate(ca3d_engine *h)
{
	const size_t bytes = h->buffer_words() * sizeof(uint32_t);
	for (int i = 0; i < 2; i++)
	{
		hipError_t e = hipMalloc((void **)&h->buf[i], bytes);
		if (e != hipSuccess)
		{
			free_buffers(h);
			return fail(CA3D_ERR_OUT_OF_MEMORY, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
		}
	}
	HIP_TRY(hipMemsetAsync(h->buf[0], 0, bytes, h->stream));
	HIP_TRY(hipMemsetAsync(h->buf[1], 0, bytes, h->stream));
	h->configured = true;
	return CA3D_OK;
}

// One launch reading buffer `src` over output planes [lo, hi) (plus [lo2, hi2) when given: the packed class kernels
// take both ranges in one launch): a single step, or a fused multi-step pass.
int enqueue_step(ca3d_engine *h, int src, uint32_t lo, uint32_t hi, hipStream_t s, bool fused = false, uint32_t lo2 = 0, uint32_t hi2 = 0)
{
	PlaneRange pr;
	pr.G = h->G;
	pr.nplanes = h->nplanes;
	pr.zbase = h->slab ? (int32_t)h->z0 - (int32_t)h->ghost : 0;
	pr.lo = lo;
	pr.hi = hi;
	pr.wrap_full = h->slab ? 0u : 1u;
	hipError_t e;
	if (h->layout == CA3D_LAYOUT_PACKED32)
	{
		pr.lo2 = lo2;
		pr.hi2 = hi2;
		PackedLaunch l{h->buf[src], h->buf[src ^ 1], pr, &h->rules, h->variant, h->vn_jit.cvl >= 0 ? &h->vn_jit : nullptr, h->class_jit.main >= 0 ? &h->class_jit : nullptr, (h->roll_jit.cvl >= 0 || h->roll_jit.cv_np2 > 0) ? &h->roll_jit : nullptr, h->roll_z, h->roll_tile, h->rows_jit.main >= 0 ? &h->rows_jit : nullptr};
		e = fused ? launch_packed_fused(l, s, &h->kernel_name) : launch_packed_step(l, s, &h->kernel_name);
	}
	else
	{
		UnpackedLaunch l{h->buf[src], h->buf[src ^ 1], pr, &h->rules, h->binary_state && h->variant == 0};
		e = launch_unpacked_step(l, s, &h->kernel_name);
		if (e == hipSuccess && hi2 > lo2)
		{
			l.pr.lo = lo2;
			l.pr.hi = hi2;
			e = launch_unpacked_step(l, s, &h->kernel_name);
		}
		h->binary_state = true; // the kernel writes only 0 / 1 (compute.wgsl:160-174)
	}
	if (e != hipSuccess) return fail(CA3D_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(e));
	return CA3D_OK;
}

// (Re)select the kernels for the current rules and grid; compiles the rule's specialisation when one applies. Called
// whenever rules, grid or the relevant options change — never from the step path (the WebGPU analogue is pipeline
// creation). A failed compile leaves the ahead-of-time kernels in charge.
// the 512^3 von Neumann form runs as the row-pair kernel (32-row tiles, one z group: its own geometry)
int vn_pair(const ca3d_engine *h) { return h->res_pair && h->G == 512u && h->res_rows == 32u && h->res_zsplit == 1u ? 1 : 0; } // another geometry asked for: the general form

void select_kernels(ca3d_engine *h)
{
	h->vn_jit = VnJit{};
	h->class_jit = ClassJit{};
	h->rows_jit = RowsJit{};
	h->roll_jit = RollJit{};
	h->res_ready = false;
	h->res_class = false;
	h->res_jit_fn = nullptr;
	h->res_slab_fn = nullptr;
	h->jit_log.clear();
	if (!h->configured || !h->rules.valid) return;
	if (h->layout != CA3D_LAYOUT_PACKED32) { h->kernel_name = "ca_unpacked_literal"; return; }
	h->kernel_name = packed_kernel_name(h->rules, h->G, h->variant);
	if (h->slab && h->use_resident && h->use_jit)
	{
		// a rank's share of a 1024^3 grid: K sub-steps per launch with the slab on chip (ca_resident_kernel.inc, slab form)
		const int pz = resident_slab_planes(h->rules, h->G, h->nplanes, h->variant);
		if (pz && hipSetDevice(h->device) == hipSuccess)
		{
			uint32_t ls1 = 0, lb1 = 0;
			vn_tables(h->rules, &ls1, &lb1);
			if (jit_resident_slab_kernel(h->device, ls1, lb1, pz, &h->res_slab_fn, &h->jit_log) != CA3D_OK) h->res_slab_fn = nullptr;
		}
	}
	{
		// the resident kernel of the start-up rule is pre-built: available with or without the run-time compiler
		uint32_t ls0 = 0, lb0 = 0;
		if (h->use_resident && !h->slab && resident_kernel_applies(h->rules, h->G, h->variant))
		{
			vn_tables(h->rules, &ls0, &lb0);
			if (vn_tables_prebuilt(ls0, lb0)) h->res_ready = true;
		}
	}
	if (!h->use_jit) return;
	if (h->use_roll && roll_np2_applies(h->rules, h->G, h->variant))
	{
		// rows of 3, 5, 6 or 7 uint4 and a rule with diagonal neighbour classes: the rolling-window kernel's whole-rows-per-wave form
		if (hipSetDevice(h->device) != hipSuccess) return;
		RollJit rj;
		if (jit_roll_np2_kernels(h->device, h->rules, (int)(h->G / 128u), &rj, &h->jit_log) == CA3D_OK) { h->roll_jit = rj; h->kernel_name = "ca_packed_roll_np2(jit)"; return; }
	}
	if (h->use_rows && rows_kernel_applies(h->rules, h->G, h->variant))
	{
		// grids without a uint4 kernel that has the rule compiled in (not a power of two, or rows that are not whole uint4)
		if (hipSetDevice(h->device) != hipSuccess) return;
		RowsJit rj;
		if (jit_rows_kernels(h->device, h->rules, h->G, &rj, &h->jit_log) == CA3D_OK) { h->rows_jit = rj; h->kernel_name = "ca_packed_rows(jit)"; }
		if (h->G == 64u && !h->res_ready && h->use_resident && !h->slab && resident_kernel_applies(h->rules, h->G, h->variant))
		{
			// 64^3, a von Neumann table pair other than the start-up rule's: the one-workgroup resident kernel compiled for the tables
			uint32_t ls64 = 0, lb64 = 0;
			vn_tables(h->rules, &ls64, &lb64);
			if (jit_resident_kernel(h->device, ls64, lb64, 64u, 1u, 0, &h->res_jit_fn, &h->jit_log) == CA3D_OK) h->res_ready = true;
		}
		return;
	}
	if (!vn_kernel_applies(h->rules, h->G, h->variant))
	{
		// class kernels on power-of-two grids: the rule's truth tables become compile-time constants
		const uint32_t cv = h->G / 128u;
		if (!use_class_kernel(h->rules, h->G, h->variant) || h->G % 128u || cv > 64u) return;
		if (hipSetDevice(h->device) != hipSuccess) return;
		ClassJit cj;
		if (jit_class_kernels(h->device, h->rules, &cj, &h->jit_log) == CA3D_OK) h->class_jit = cj;
		else return;
		if (cv & (cv - 1u)) return; // not a power of two: the class kernel's np2 entry points, nothing else
		if (h->use_roll && roll_kernel_applies(h->rules, h->G, h->variant))
		{
			RollJit rj;
			if (jit_roll_kernels(h->device, h->rules, vn_grid_log2(h->G), &rj, &h->jit_log) == CA3D_OK) h->roll_jit = rj;
		}
		if (h->use_resident && !h->slab && resident_class_applies(h->rules, h->G, h->variant) &&
		    jit_resident_class_kernel(h->device, h->rules, h->G, resident_class_zsplit(h->G), &h->res_jit_fn, &h->jit_log) == CA3D_OK)
		{
			h->res_ready = true;
			h->res_class = true;
		}
		return;
	}
	uint32_t ls = 0, lb = 0;
	vn_tables(h->rules, &ls, &lb);
	const bool resident = h->use_resident && !h->slab && resident_kernel_applies(h->rules, h->G, h->variant);
	if (vn_tables_prebuilt(ls, lb)) return;
	if (hipSetDevice(h->device) != hipSuccess) return;
	VnJit j;
	if (jit_vn_kernels(h->device, vn_grid_log2(h->G), ls, lb, &j, &h->jit_log) == CA3D_OK)
	{
		h->vn_jit = j;
		h->kernel_name = "ca_packed_vn(jit)";
	}
	if (resident && jit_resident_kernel(h->device, ls, lb, h->G == 256u ? 256u : h->res_rows, h->res_zsplit, vn_pair(h), &h->res_jit_fn, &h->jit_log) == CA3D_OK) h->res_ready = true;
}

// A resident launch only completes when ALL its workgroups are on the chip at once (they wait for each other's faces). Ask the
// runtime before selecting one: occupancy of the chosen kernel per CU x the CUs the engine's stream may use (a CU mask, a
// partitioned device) against the tile count. Too few: the per-step kernels run, and ca3d_last_error says why. What the
// query cannot see (another process or stream holding CUs) is left to the kernels' bounded waits and the recovery below.
void check_residency(ca3d_engine *h)
{
	h->res_note.clear();
	if (!h->res_ready && !h->res_slab_fn) return;
	if (hipSetDevice(h->device) != hipSuccess) return;
	uint32_t tiles = 0, cap = 0;
	char buf[256];
	if (h->res_ready)
	{
		const uint32_t rows = (h->res_class || h->G == 256u || vn_pair(h)) ? 32u : h->res_rows;
		if (resident_capacity(h->G, rows, h->res_class ? resident_class_zsplit(h->G) : h->res_zsplit, h->res_class ? 0 : vn_pair(h), h->res_jit_fn, h->stream, &tiles, &cap) && cap < tiles)
		{
			h->res_ready = false;
			h->res_class = false;
			snprintf(buf, sizeof buf, "resident multi-step kernel not selected: it needs %u co-resident workgroups, this device / stream holds %u; per-step kernels in use", tiles, cap);
			h->res_note = buf;
		}
	}
	if (h->res_slab_fn && resident_slab_capacity(h->res_slab_fn, h->stream, &tiles, &cap) && cap < tiles)
	{
		h->res_slab_fn = nullptr;
		snprintf(buf, sizeof buf, "resident slab kernel not selected: it needs %u co-resident workgroups, this device / stream holds %u; per-step kernels in use", tiles, cap);
		h->res_note = buf;
	}
}

void refresh_kernels(ca3d_engine *h)
{
	select_kernels(h);
	check_residency(h);
}

// A failed specialisation is not an error of the call that triggered it (the ahead-of-time kernels take over), but it
// must not be silent: the message goes where the caller looks (ca3d_last_error, ca3d_get_jit_log). Likewise a resident
// kernel that exists for the rules but cannot be co-resident on this device / stream.
void note_jit_failure(const ca3d_engine *h)
{
	if (!h->jit_log.empty()) fail(0, "run-time kernel specialisation failed, pre-built kernels in use: %s", h->jit_log.c_str());
	else if (!h->res_note.empty()) set_last_error(h->res_note.c_str());
}

int check_ready(ca3d_engine *h)
{
	if (!h) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL engine handle");
	if (!h->configured) return fail(CA3D_ERR_NOT_CONFIGURED, "ca3d_configure has not been called");
	if (!h->rules.valid) return fail(CA3D_ERR_NOT_CONFIGURED, "ca3d_set_rules has not been called");
	if (!h->has_state) return fail(CA3D_ERR_NOT_CONFIGURED, "ca3d_upload_state has not been called");
	return CA3D_OK;
}

// Longest captured batch: consecutive graph launches leave a few microseconds of idle GPU between them, negligible
// against 1024 steps; a batch of n < 1024 steps gets a graph of exactly n steps (cached per n and start buffer).
constexpr uint32_t kMaxGraphSteps = 1024;
constexpr size_t kMaxStepGraphs = 24;

// Launch plan for n steps that keeps the reference's ping-pong invariant (main_pathtraced.js:1580-1609): the
// state after n steps sits in buffer (start + n) % 2 and the other buffer holds the state one step earlier. A
// fused pass advances T = 2 steps but flips the buffer once, so fused passes come in even numbers and the batch
// always ends with single steps.
void plan_steps(const ca3d_engine *h, uint32_t n, uint32_t *n_fused, uint32_t *n_single)
{
	uint32_t f = 0;
	if (h->use_fused && h->layout == CA3D_LAYOUT_PACKED32 && !h->slab && packed_fused_steps(h->rules, h->G, h->variant) == 2 && n >= 3)
	{
		f = (n - 1u) / 2u;
		f &= ~1u;
	}
	*n_fused = f;
	*n_single = n - 2u * f;
}

int enqueue_batch(ca3d_engine *h, uint32_t n, uint32_t start_buf, hipStream_t s, uint64_t *launches)
{
	uint32_t f, single;
	plan_steps(h, n, &f, &single);
	uint32_t cur = start_buf;
	for (uint32_t i = 0; i < f; i++, cur ^= 1u)
	{
		int rc = enqueue_step(h, (int)cur, 0, h->G, s, true);
		if (rc) return rc;
	}
	for (uint32_t i = 0; i < single; i++, cur ^= 1u)
	{
		int rc = enqueue_step(h, (int)cur, 0, h->G, s, false);
		if (rc) return rc;
	}
	if (launches) *launches += f + single;
	return CA3D_OK;
}

// Captured batch of n steps starting from buffer `start` (built on first use).
int step_graph(ca3d_engine *h, uint32_t n, uint32_t start, ca3d_engine::StepGraph **out)
{
	const uint64_t key = ((uint64_t)start << 32) | n;
	auto it = h->step_graphs.find(key);
	if (it != h->step_graphs.end()) { *out = &it->second; return CA3D_OK; }
	if (h->step_graphs.size() >= kMaxStepGraphs)
	{
		// a caller cycling through many batch lengths: start over rather than grow without bound
		HIP_TRY(hipStreamSynchronize(h->stream));
		for (auto &kv : h->step_graphs) hipGraphExecDestroy(kv.second.exec);
		h->step_graphs.clear();
	}
	// Launch boundaries stay; the host cost per launch drops from ~4 us to the graph's amortised cost.
	hipGraph_t graph = nullptr;
	HIP_TRY(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
	uint64_t launches = 0;
	int rc = enqueue_batch(h, n, start, h->stream, &launches);
	hipError_t e = hipStreamEndCapture(h->stream, &graph);
	if (rc != CA3D_OK) { if (graph) hipGraphDestroy(graph); return rc; }
	if (e != hipSuccess) return fail(CA3D_ERR_DEVICE, "hipStreamEndCapture: %s", hipGetErrorString(e));
	ca3d_engine::StepGraph g;
	e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
	hipGraphDestroy(graph);
	if (e != hipSuccess) return fail(CA3D_ERR_DEVICE, "hipGraphInstantiate: %s", hipGetErrorString(e));
	g.launches = (uint32_t)launches;
	*out = &h->step_graphs.emplace(key, g).first->second;
	return CA3D_OK;
}

constexpr size_t kResStatusBytes = (4 + 1024) * sizeof(uint32_t); // abort word + per-tile progress words
} // namespace
static int submit_steps(ca3d_engine *h, uint32_t n_steps);
namespace
{

// Looks at the resident launches issued since the last look; the stream must have been waited for. Slab engines: a launch
// that timed out leaves an invalid state behind (the neighbours' ghosts were refreshed from it) — an error, the path goes
// off. Full-grid engines recover (see ca3d_engine::res_pending): the failed launch and the ones behind it wrote nothing, so
// the engine returns to the failed launch's input and runs all their steps through the per-step kernels, then waits for
// them. Success with the resident path switched off; ca3d_last_error carries the note.
int check_resident(ca3d_engine *h)
{
	if (!h->res_status_host) return CA3D_OK;
	if (!h->res_check && h->res_pending.empty()) return CA3D_OK;
	h->res_check = false;
	if (*h->res_status_host == 0) { h->res_pending.clear(); return CA3D_OK; }
	const uint32_t who = h->res_status_host[0], ep = h->res_status_host[1];
	h->res_failed = true; // per-step kernels from here on
	size_t idx = h->res_pending.size();
	for (size_t i = 0; i < h->res_pending.size(); i++)
		if (h->res_pending[i].epoch0 == ep) { idx = i; break; }
	if (h->slab || idx == h->res_pending.size())
	{
		h->res_pending.clear();
		h->has_state = false;
		return fail(CA3D_ERR_DEVICE, "resident multi-step kernel: a wait for neighbour tile faces timed out (tile %u gave up first) — were all its "
		            "workgroups resident? The state is invalid: upload it again; the engine now uses the per-step kernels", who - 1u);
	}
	const ca3d_engine::ResPending p = h->res_pending[idx];
	uint64_t total = 0;
	for (size_t i = idx; i < h->res_pending.size(); i++) total += h->res_pending[i].n;
	h->res_pending.clear();
	drop_graph(h);
	// The three buffers only rotate: whichever of them is neither the failed launch's input nor its other buffer is the spare — not
	// `p.spare`, which is null when that launch was a one-step one issued before a later queued launch allocated the third buffer
	// (restoring null would leak it).
	uint32_t *third = p.spare;
	for (uint32_t *q : {h->buf[0], h->buf[1], h->spare})
		if (q && q != p.in && q != p.other) third = q;
	h->buf[p.cur_before] = p.in;
	h->buf[p.cur_before ^ 1u] = p.other;
	h->spare = third;
	h->cur = p.cur_before;
	h->step = p.step_before;
	h->state_serial++; // whatever the renderer derived from the buffers of the failed launches is void
	{
		// ... and so is what frames drawn in the meantime left behind: ca3d_render without host pointers does not wait for a pending
		// resident launch (the frame loop must not stall on it), so a frame may have been drawn from the unwritten output of the launch
		// that has now turned out to have failed — wrong once on screen, but in the literal mode it was also blended into the history
		// surfaces and would linger for several frames (EMA, alpha 0.1). A fresh canvas instead.
		const size_t px = (size_t)h->rw * h->rh;
		for (int i = 0; i < 2 && px && h->r_light[i] && h->r_depth[i]; i++)
		{
			HIP_TRY(hipMemsetAsync(h->r_light[i], 0, px * 8, h->stream));
			HIP_TRY(hipMemsetAsync(h->r_depth[i], 0, px * 4, h->stream));
		}
	}
	HIP_TRY(hipMemsetAsync(h->res_mail, 0, h->res_mail_bytes, h->stream));
	HIP_TRY(hipMemsetAsync(h->res_status, 0, kResStatusBytes, h->stream));
	h->res_status_host[0] = h->res_status_host[1] = 0;
	h->res_epoch = 0;
	h->res_recovered++;
	char note[256];
	snprintf(note, sizeof note, "resident multi-step kernel: a wait for neighbour tile faces timed out (tile %u gave up first); the %llu steps "
	         "it and the launches behind it covered were re-run through the per-step kernels, which stay in use", who - 1u, (unsigned long long)total);
	h->res_note = note;
	while (total)
	{
		const uint32_t n = total > 0x40000000ull ? 0x40000000u : (uint32_t)total;
		int rc = submit_steps(h, n);
		if (rc) return rc;
		total -= n;
	}
	HIP_TRY(hipStreamSynchronize(h->stream));
	set_last_error(h->res_note.c_str());
	return CA3D_OK;
}

// Before anything that reads the state, hands out its buffers or changes how steps run: make sure no unverified resident
// launch is outstanding (wait for the stream, recover if one gave up). Costs nothing when none is.
int settle_resident(ca3d_engine *h)
{
	if (h->res_pending.empty() && !h->res_check) return CA3D_OK;
	int rc = bind_device(h);
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(h->stream));
	return check_resident(h);
}

// n steps as ONE launch of the resident kernel (state in registers between steps).
int resident_buffers(ca3d_engine *h, uint32_t n)
{
	if (!h->res_mail)
	{
		const size_t bytes = h->slab ? resident_slab_mail_bytes() : resident_mail_bytes(h->G, 16u); // sized for the finer tiling
		h->res_mail_bytes = bytes;
		HIP_TRY(hipMalloc((void **)&h->res_mail, bytes));
		HIP_TRY(hipMalloc((void **)&h->res_status, kResStatusBytes));
		HIP_TRY(hipHostMalloc((void **)&h->res_status_host, 16, hipHostMallocDefault));
		*h->res_status_host = 0;
		HIP_TRY(hipMemsetAsync(h->res_mail, 0, bytes, h->stream));
		HIP_TRY(hipMemsetAsync(h->res_status, 0, kResStatusBytes, h->stream));
		h->res_epoch = 0;
	}
	if (h->res_epoch > 0xFFFFFFFFu - n - 4u)
	{
		// the 32-bit state tags would wrap: start the numbering again from clean mailboxes
		HIP_TRY(hipMemsetAsync(h->res_mail, 0, h->res_mail_bytes, h->stream));
		h->res_epoch = 0;
	}
	return CA3D_OK;
}

// n sub-steps of a slab as ONE launch (state tiles in registers, faces through the mailboxes); the whole array is updated,
// the planes outside [n, L - n) are stale afterwards like after a per-step batch.
int resident_slab_steps(ca3d_engine *h, uint32_t n)
{
	int rc = resident_buffers(h, n);
	if (rc) return rc;
	ResidentSlabLaunch l;
	l.in = h->buf[h->cur];
	l.out = h->buf[(h->cur + n) & 1u];
	l.mail = h->res_mail;
	l.status = h->res_status;
	l.host_flag = h->res_status_host;
	l.steps = n;
	l.epoch0 = h->res_epoch;
	l.timeout_ticks = h->res_timeout_ticks;
	const int64_t zbase = (int64_t)h->z0 - (int64_t)h->ghost;
	const int64_t dead = ((-zbase) % (int64_t)h->G + (int64_t)h->G) % (int64_t)h->G; // array plane with global z == 0
	l.dead_plane = dead < (int64_t)h->nplanes ? (int)dead : -1;
	l.fn = h->res_slab_fn;
	hipError_t e = launch_resident_slab(l, h->stream);
	if (e != hipSuccess) return fail(CA3D_ERR_DEVICE, "resident slab kernel launch failed: %s", hipGetErrorString(e));
	h->res_epoch += n;
	h->res_check = true;
	h->kernel_name = "ca_resident_slab_vn(jit)";
	return CA3D_OK;
}

int resident_steps(ca3d_engine *h, uint32_t n)
{
	int rc0 = resident_buffers(h, n);
	if (rc0) return rc0;
	if (n >= 2u && !h->spare)
	{
		HIP_TRY(hipMalloc((void **)&h->spare, h->buffer_words() * sizeof(uint32_t)));
	}
	uint32_t *in = h->buf[h->cur], *other = h->buf[h->cur ^ 1u];
	ResidentLaunch l;
	l.in = in;
	// n >= 2: nothing is written to the input (see ca3d_engine::spare); n == 1: the other buffer receives the new state and
	// the input IS the state one step earlier
	l.out_last = n >= 2u ? h->spare : other;
	l.out_prev = n >= 2u ? other : in;
	l.G = h->G;
	l.mail = h->res_mail;
	l.status = h->res_status;
	l.host_flag = h->res_status_host;
	l.steps = n;
	l.epoch0 = h->res_epoch;
	l.timeout_ticks = h->res_timeout_ticks;
	l.fault_tile = h->res_fault_tile;
	h->res_fault_tile = 0;
	l.lut_s = l.lut_b = 0;
	if (!h->res_class) vn_tables(h->rules, &l.lut_s, &l.lut_b);
	l.jit_fn = h->res_jit_fn;
	l.rows = (h->res_class || vn_pair(h)) ? 32u : h->res_rows;
	l.zsplit = h->res_class ? resident_class_zsplit(h->G) : h->res_zsplit;
	l.pair = h->res_class ? 0 : vn_pair(h);
	hipError_t e = launch_resident(l, h->stream);
	if (e != hipSuccess) return fail(CA3D_ERR_DEVICE, "resident kernel launch failed: %s", hipGetErrorString(e));
	h->res_pending.push_back({h->res_epoch, n, h->cur, h->step, in, other, h->spare});
	if (n >= 2u)
	{
		// rotate: buf[(cur + n) % 2] = the final state, the other one = the state one step earlier, the input becomes the spare
		if (!h->step_graphs.empty() || !h->slab_graphs.empty() || !h->comm_graphs.empty()) drop_graph(h); // they hold the old pointers
		const uint32_t f = (h->cur + n) & 1u;
		h->buf[f] = h->spare;
		h->buf[f ^ 1u] = other;
		h->spare = in;
	}
	h->res_epoch += n;
	h->kernel_name = h->res_class ? "ca_resident_class(jit)" : (h->res_jit_fn ? "ca_resident_vn(jit)" : "ca_resident_vn");
	return CA3D_OK;
}

// The unpacked layout's first step after an upload with cell values > 1 must be the literal kernel (raw u32 sums,
// compute.wgsl:160-174); every later state is 0 / 1. Graphs are only ever captured in the 0 / 1 regime, so a
// cached graph can never replay the wrong kernel after a new upload.
bool graphs_allowed(const ca3d_engine *h)
{
	return h->use_graph && h->stream != nullptr && !(h->layout == CA3D_LAYOUT_UNPACKED && !h->binary_state);
}

// ---------------------------------------------------------------------------------------------- RCCL transport
// librccl is loaded on first use (dlopen): single-GPU hosts and the Node.js addon never pay for it, and a process that
// has torch's copy of librccl.so.1 loaded gets that same copy.
typedef struct { char internal[128]; } ncclUniqueIdBytes; // ncclUniqueId (NCCL_UNIQUE_ID_BYTES)

struct Rccl
{
	void *lib = nullptr;
	int (*GetUniqueId)(void *) = nullptr;
	int (*CommInitRank)(void **, int, ncclUniqueIdBytes, int) = nullptr;
	int (*CommDestroy)(void *) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
	int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
	int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
	int (*CommCount)(void *, int *) = nullptr;
	int (*CommUserRank)(void *, int *) = nullptr;
	int (*CommCuDevice)(void *, int *) = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
	std::string error;
};

Rccl &rccl()
{
	static Rccl r;
	static bool tried = false;
	if (tried) return r;
	tried = true;
	for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
	{
		r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
		if (r.lib) break;
	}
	if (!r.lib) { r.error = std::string("librccl.so.1 could not be loaded: ") + dlerror(); return r; }
	auto sym = [&](const char *n) { void *p = dlsym(r.lib, n); if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + n; return p; };
	r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
	r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
	r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
	r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
	r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
	r.Send = (decltype(r.Send))sym("ncclSend");
	r.Recv = (decltype(r.Recv))sym("ncclRecv");
	r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
	r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
	r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
	r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
	r.CommCuDevice = (decltype(r.CommCuDevice))sym("ncclCommCuDevice");
	return r;
}

#define NCCL_TRY(expr)                                                                                                   \
	do                                                                                                                   \
	{                                                                                                                    \
		int r_ = (expr);                                                                                                 \
		if (r_ != 0) return fail(CA3D_ERR_DEVICE, "%s: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(r_) : "?"); \
	} while (0)

constexpr int kNcclUint32 = 3; // ncclUint32 (rccl.h)

// Refresh the ghost planes of the buffer ca3d_slab_region refers to, on stream `s`: the same plan as slab.halo_plan —
// packed: open at the bottom (z == -1 is dead), closed at the top (plane G wraps to plane 0); unpacked: a ring.
int comm_exchange(ca3d_engine *h, hipStream_t s)
{
	Rccl &r = rccl();
	const int P = h->comm_world, me = h->comm_rank, below = (me + P - 1) % P, above = (me + 1) % P;
	const bool ring = h->layout == CA3D_LAYOUT_UNPACKED, top = me == P - 1, bottom = me == 0;
	void *p[4];
	size_t bytes[4];
	for (int region = 0; region < 4; region++)
	{
		int rc = ca3d_slab_region(h, region, &p[region], &bytes[region]);
		if (rc) return rc;
	}
	const size_t n = bytes[0] / sizeof(uint32_t);
	NCCL_TRY(r.GroupStart());
	// sends low-then-high, receives high-then-low: the two messages a pair of ranks exchanges in one direction (world == 2)
	// then match in order
	NCCL_TRY(r.Send(p[CA3D_SLAB_SEND_LOW], n, kNcclUint32, below, h->comm, s));
	if (ring || !top) NCCL_TRY(r.Send(p[CA3D_SLAB_SEND_HIGH], n, kNcclUint32, above, h->comm, s));
	NCCL_TRY(r.Recv(p[CA3D_SLAB_RECV_HIGH], n, kNcclUint32, above, h->comm, s));
	if (ring || !bottom) NCCL_TRY(r.Recv(p[CA3D_SLAB_RECV_LOW], n, kNcclUint32, below, h->comm, s));
	NCCL_TRY(r.GroupEnd());
	return CA3D_OK;
}

} // namespace

// queue.submit of the steps encoded so far (option "queue"). Every entry point that looks at the state, the stream or the
// options goes through here first, so a caller only ever sees the order it asked for.
int flush_queued(ca3d_engine *h)
{
	if (!h || !h->queued) return CA3D_OK;
	const uint32_t n = h->queued;
	h->queued = 0;
	return submit_steps(h, n);
}

#define FLUSH_QUEUED(h)                \
	do                                 \
	{                                  \
		int rcq_ = flush_queued(h);    \
		if (rcq_) return rcq_;         \
	} while (0)


extern "C"
{

int ca3d_abi_version(void) { return CA3D_ABI_VERSION; }

const char *ca3d_last_error(void) { return g_last_error; }

// Test hook: throws inside a guarded body so that the exception boundary itself can be exercised without a GPU.
// kind 0: std::bad_alloc, 1: std::runtime_error, 2: a non-std exception, 3: a real allocation failure (a vector of SIZE_MAX / 2 bytes)
int ca3d_selftest_exception(int kind) CA3D_API_TRY
{
	if (kind == 0) throw std::bad_alloc();
	if (kind == 1) throw std::runtime_error("selftest");
	if (kind == 2) throw 42;
	if (kind == 3)
	{
		std::vector<char> v;
		v.resize(v.max_size() / 2u); // std::length_error or std::bad_alloc, whichever the allocator reports
		return fail(CA3D_ERR_DEVICE, "selftest: the allocation of %zu bytes succeeded", v.size());
	}
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_device_count(int *out_count) CA3D_API_TRY
{
	if (!out_count) return fail(CA3D_ERR_INVALID_ARGUMENT, "out_count is NULL");
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess) { *out_count = 0; return fail(CA3D_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
	*out_count = n;
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_create(int device, ca3d_t **out) CA3D_API_TRY
{
	if (!out) return fail(CA3D_ERR_INVALID_ARGUMENT, "out is NULL");
	*out = nullptr;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0)
		return fail(CA3D_ERR_DEVICE, "no HIP device available (%s); this engine has no CPU fallback",
		            e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
	if (device < 0 || device >= n) return fail(CA3D_ERR_INVALID_ARGUMENT, "device %d out of range [0,%d)", device, n);
	HIP_TRY(hipSetDevice(device));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(CA3D_ERR_UNSUPPORTED, "device %d is %s; this library carries gfx950 (MI355X) code objects only", device, prop.gcnArchName);
	ca3d_engine *h = new (std::nothrow) ca3d_engine();
	if (!h) return fail(CA3D_ERR_OUT_OF_MEMORY, "out of host memory");
	h->device = device;
	hipError_t err = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
	if (err == hipSuccess) err = hipEventCreate(&h->ev_start);
	if (err == hipSuccess) err = hipEventCreate(&h->ev_stop);
	if (err == hipSuccess) err = hipEventCreate(&h->rev_start);
	if (err == hipSuccess) err = hipEventCreate(&h->rev_stop);
	if (err != hipSuccess)
	{
		ca3d_destroy(h); // releases whatever was created
		return fail(CA3D_ERR_DEVICE, "engine set-up failed: %s", hipGetErrorString(err));
	}
	h->stream = h->own_stream;
	*out = h;
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_destroy(ca3d_t *h) CA3D_API_TRY
{
	if (!h) return CA3D_OK;
	hipSetDevice(h->device);
	h->queued = 0; // never submitted: the state goes away with the engine
	if (h->stream || h->own_stream) hipStreamSynchronize(h->stream);
	for (auto &L : h->lanes)
		if (L.s) hipStreamSynchronize(L.s); // frames in flight read the state
	resident_stream_retired(h->stream);
	free_buffers(h);
	free_render_targets(h);
	if (h->comm && rccl().CommDestroy) rccl().CommDestroy(h->comm);
	if (h->ev_edges) hipEventDestroy(h->ev_edges);
	if (h->ev_comm) hipEventDestroy(h->ev_comm);
	if (h->ev_gather) hipEventDestroy(h->ev_gather);
	if (h->comm_stream) hipStreamDestroy(h->comm_stream);
	if (h->r_counters) hipFree(h->r_counters);
	if (h->r_occ) hipFree(h->r_occ);
	if (h->r_stream) hipFree(h->r_stream);
	if (h->r_bricks) hipFree(h->r_bricks);
	for (auto &L : h->lanes)
	{
		if (L.s) { hipStreamSynchronize(L.s); hipStreamDestroy(L.s); }
		for (hipEvent_t e : {L.start, L.stop, L.done})
			if (e) hipEventDestroy(e);
		if (L.scratch) hipFree(L.scratch);
		if (L.counters) hipFree(L.counters);
	}
	for (hipStream_t sp : h->lane_spares) hipStreamDestroy(sp);
	if (h->ev_state) hipEventDestroy(h->ev_state);
	if (h->r_aux) hipStreamDestroy(h->r_aux);
	if (h->r_fork) hipEventDestroy(h->r_fork);
	if (h->r_join) hipEventDestroy(h->r_join);
	if (h->rev_start) hipEventDestroy(h->rev_start);
	if (h->rev_stop) hipEventDestroy(h->rev_stop);
	if (h->ev_start) hipEventDestroy(h->ev_start);
	if (h->ev_stop) hipEventDestroy(h->ev_stop);
	if (h->own_stream) hipStreamDestroy(h->own_stream);
	delete h;
	return CA3D_OK;
}
CA3D_API_CATCH

static int configure_common(ca3d_t *h, uint32_t g, int layout)
{
	if (!h) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL engine handle");
	if (layout != CA3D_LAYOUT_PACKED32 && layout != CA3D_LAYOUT_UNPACKED) return fail(CA3D_ERR_INVALID_ARGUMENT, "unknown layout %d", layout);
	if (g == 0) return fail(CA3D_ERR_INVALID_ARGUMENT, "grid size must be positive");
	if (layout == CA3D_LAYOUT_PACKED32 && (g % 32u)) return fail(CA3D_ERR_INVALID_ARGUMENT, "packed layout needs a grid size that is a multiple of 32 (got %u)", g);
	if (layout == CA3D_LAYOUT_UNPACKED && (g % 4u)) return fail(CA3D_ERR_INVALID_ARGUMENT, "unpacked layout needs a grid size that is a multiple of 4 (got %u)", g);
	if (g > 8192u) return fail(CA3D_ERR_UNSUPPORTED, "grid size %u exceeds the supported maximum 8192", g);
	int rc = bind_device(h);
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(h->stream));
	free_buffers(h);
	h->buffers_exposed = false;
	h->G = g;
	h->layout = layout;
	h->plane_words = layout == CA3D_LAYOUT_PACKED32 ? (size_t)(g / 32u) * g : (size_t)g * g;
	return CA3D_OK;
}

int ca3d_configure(ca3d_t *h, uint32_t gx, uint32_t gy, uint32_t gz, int layout) CA3D_API_TRY
{
	if (gx != gy || gy != gz) return fail(CA3D_ERR_UNSUPPORTED, "only cubic grids exist in the reference (got %ux%ux%u)", gx, gy, gz);
	if (h) h->queued = 0; // steps of a state that is being thrown away
	int rc = configure_common(h, gx, layout);
	if (rc) return rc;
	h->slab = false;
	h->z0 = 0;
	h->nz = gx;
	h->ghost = 0;
	h->nplanes = gx;
	rc = allocate(h);
	if (rc == CA3D_OK) { refresh_kernels(h); note_jit_failure(h); }
	return rc;
}
CA3D_API_CATCH

int ca3d_configure_slab(ca3d_t *h, uint32_t g, int layout, uint32_t z0, uint32_t nz, uint32_t ghost) CA3D_API_TRY
{
	if (h) h->queued = 0;
	int rc = configure_common(h, g, layout);
	if (rc) return rc;
	if (nz == 0 || z0 + nz > g) return fail(CA3D_ERR_INVALID_ARGUMENT, "slab [%u, %u) is outside the grid of %u planes", z0, z0 + nz, g);
	if (ghost == 0 || ghost > nz) return fail(CA3D_ERR_INVALID_ARGUMENT, "ghost depth must be in [1, nz] (got %u, nz %u)", ghost, nz);
	if (layout == CA3D_LAYOUT_UNPACKED && (g & (g - 1u))) return fail(CA3D_ERR_UNSUPPORTED, "unpacked slabs need a power-of-two grid size: the legacy kernel's -1 wrap is a torus only then");
	h->slab = true;
	h->z0 = z0;
	h->nz = nz;
	h->ghost = ghost;
	h->nplanes = nz + 2u * ghost;
	rc = allocate(h);
	if (rc == CA3D_OK) { refresh_kernels(h); note_jit_failure(h); }
	return rc;
}
CA3D_API_CATCH

int ca3d_set_rules(ca3d_t *h, const int32_t *main_offsets, uint32_t n_main, const int32_t *edges_offsets, uint32_t n_edges,
                   const int32_t *corners_offsets, uint32_t n_corners, const uint32_t survive[CA3D_LUT_LEN],
                   const uint32_t born[CA3D_LUT_LEN]) CA3D_API_TRY
{
	if (!h) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL engine handle");
	FLUSH_QUEUED(h); // the steps encoded so far run under the rules they were encoded with
	if (int rcs = settle_resident(h)) return rcs; // ... and a recovery re-runs them under those rules too
	CanonRules r;
	std::string err;
	int rc = canonicalize_rules(main_offsets, n_main, edges_offsets, n_edges, corners_offsets, n_corners, survive, born, &r, &err);
	if (rc) return fail(rc, "%s", err.c_str());
	rc = bind_device(h);
	if (rc) return rc;
	drop_graph(h);
	h->rules = r;
	refresh_kernels(h);
	note_jit_failure(h);
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_upload_state(ca3d_t *h, const uint32_t *words, size_t n_words) CA3D_API_TRY
{
	if (!h) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL engine handle");
	if (!h->configured) return fail(CA3D_ERR_NOT_CONFIGURED, "ca3d_configure has not been called");
	if (!words) return fail(CA3D_ERR_INVALID_ARGUMENT, "words is NULL");
	h->queued = 0; // the state they would have produced is overwritten
	if (n_words != h->state_words()) return fail(CA3D_ERR_INVALID_ARGUMENT, "state has %zu words, expected %zu", n_words, h->state_words());
	int rc = bind_device(h);
	if (rc) return rc;
	const size_t off = h->slab ? (size_t)h->ghost * h->plane_words : 0;
	const size_t bytes = n_words * sizeof(uint32_t);
	// Same data into both ping-pong buffers (main_pathtraced.js:1361-1362); ghosts are cleared.
	if (h->slab)
	{
		HIP_TRY(hipMemsetAsync(h->buf[0], 0, h->buffer_words() * sizeof(uint32_t), h->stream));
		HIP_TRY(hipMemsetAsync(h->buf[1], 0, h->buffer_words() * sizeof(uint32_t), h->stream));
	}
	HIP_TRY(hipMemcpyAsync(h->buf[0] + off, words, bytes, hipMemcpyHostToDevice, h->stream));
	HIP_TRY(hipMemcpyAsync(h->buf[1] + off, h->buf[0] + off, bytes, hipMemcpyDeviceToDevice, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	h->step = 0;
	h->cur = 0;
	h->state_serial++;
	h->buffers_exposed = false;
	h->pending_edges = 0; // a restart between the two phases of a batch abandons the batch
	h->ghosts_valid = false;
	h->res_pending.clear(); // their results have just been overwritten
	if (h->res_status_host && *h->res_status_host)
	{
		// a resident launch gave up earlier: clean mailboxes and status for whoever turns the path on again
		HIP_TRY(hipMemsetAsync(h->res_mail, 0, h->res_mail_bytes, h->stream));
		HIP_TRY(hipMemsetAsync(h->res_status, 0, kResStatusBytes, h->stream));
		h->res_status_host[0] = h->res_status_host[1] = 0;
		h->res_epoch = 0;
		h->res_check = false;
		h->res_failed = true;
	}
	h->has_state = true;
	h->binary_state = false;
	if (h->layout == CA3D_LAYOUT_UNPACKED)
	{
		bool bin = true;
		for (size_t i = 0; i < n_words && bin; i++) bin = words[i] <= 1u;
		h->binary_state = bin;
	}
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_read_state(ca3d_t *h, uint32_t *words, size_t n_words) CA3D_API_TRY
{
	if (!h) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL engine handle");
	if (!h->configured || !h->has_state) return fail(CA3D_ERR_NOT_CONFIGURED, "no state to read: configure and upload first");
	if (!words) return fail(CA3D_ERR_INVALID_ARGUMENT, "words is NULL");
	FLUSH_QUEUED(h);
	if (n_words != h->state_words()) return fail(CA3D_ERR_INVALID_ARGUMENT, "state has %zu words, expected %zu", n_words, h->state_words());
	int rc = bind_device(h);
	if (rc) return rc;
	rc = settle_resident(h); // a resident launch that gave up is recovered from before the state is looked at
	if (rc) return rc;
	const size_t off = h->slab ? (size_t)h->ghost * h->plane_words : 0;
	HIP_TRY(hipMemcpyAsync(words, h->buf[h->cur] + off, n_words * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	return CA3D_OK;
}
CA3D_API_CATCH

} // extern "C"

// n steps onto the stream now.
static int submit_steps(ca3d_engine *h, uint32_t n_steps)
{
	int rc = bind_device(h);
	if (rc) return rc;
	if (n_steps == 0) return CA3D_OK;
	h->buffers_exposed = false; // (ca3d_device_buffer: the pointer it handed out was valid until this call)
	if (h->want_stats) HIP_TRY(hipEventRecord(h->ev_start, h->stream));
	uint32_t left = n_steps;
	uint64_t launches = 0;
	const bool want_resident = h->res_ready && h->use_resident && !h->res_failed && n_steps >= h->res_min && h->stream != nullptr;
	if (!h->res_pending.empty() && (!want_resident || h->res_pending.size() >= 64u))
	{
		// Per-step kernels write the ping-pong buffers whatever happened before them — one of which is the input a recovery
		// would start from (a launch queued behind a failed one does nothing, a per-step kernel cannot know): verify the
		// resident launches still outstanding first. Also for a host that never looks at the state, so that the list stays short.
		rc = settle_resident(h);
		if (rc) return rc;
	}
	if (h->res_ready && h->use_resident && !h->res_failed && n_steps >= h->res_min && h->stream != nullptr)
	{
		rc = resident_steps(h, n_steps);
		if (rc) return rc;
		h->step += n_steps;
		h->cur = (h->cur + n_steps) & 1u;
		launches = 1;
		left = 0;
	}
	while (left)
	{
		uint32_t n = left > kMaxGraphSteps ? kMaxGraphSteps : left;
		if (!graphs_allowed(h) && h->use_graph && h->stream != nullptr) n = 1; // non-binary unpacked state: one literal step, then graphs
		if (graphs_allowed(h) && n >= h->graph_min)
		{
			ca3d_engine::StepGraph *g = nullptr;
			rc = step_graph(h, n, h->cur, &g);
			if (rc) return rc;
			HIP_TRY(hipGraphLaunch(g->exec, h->stream));
			launches += g->launches;
		}
		else
		{
			rc = enqueue_batch(h, n, h->cur, h->stream, &launches);
			if (rc) return rc;
		}
		h->step += n;
		h->cur = (h->cur + n) & 1u;
		left -= n;
	}
	if (h->want_stats) HIP_TRY(hipEventRecord(h->ev_stop, h->stream));
	h->ev_valid = h->want_stats != 0;
	h->stats.steps = n_steps;
	h->stats.kernel_launches = launches;
	h->launches_total += launches;
	h->stats.cell_steps = (double)n_steps * h->cells_per_plane() * h->G;
	h->stats.algorithmic_bytes = h->stats.cell_steps * h->bytes_per_cell_step();
	return CA3D_OK;
}

extern "C"
{

int ca3d_step(ca3d_t *h, uint32_t n_steps) CA3D_API_TRY
{
	int rc = check_ready(h);
	if (rc) return rc;
	if (h->slab) return fail(CA3D_ERR_INVALID_ARGUMENT, "engine is a slab: use ca3d_slab_step and refresh the ghosts between batches");
	if (h->queue_max)
	{
		// encode only (the reference's commandEncoder, main_pathtraced.js:1833-1850): the steps of consecutive calls go to
		// the GPU as one submission, which lets the resident kernel run them as one launch
		if (n_steps > 0xFFFFFFFFu - h->queued) FLUSH_QUEUED(h);
		h->queued += n_steps;
		if (h->queued >= h->queue_max) return flush_queued(h);
		return CA3D_OK;
	}
	return submit_steps(h, n_steps);
}
CA3D_API_CATCH

int ca3d_flush(ca3d_t *h) CA3D_API_TRY
{
	if (!h) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL engine handle");
	return flush_queued(h);
}
CA3D_API_CATCH

// Slab batch of n sub-steps, whole or in two phases (include/ca3d.h). Array planes: ghost [0,K), owned [K,K+nz),
// ghost [K+nz, L). Sub-step s (1..n) of the whole batch updates [s, L-s). The phased form splits that range:
//   low edge   [s, 2K+n-s)          ends at s = n as [n, 2K): covers the planes sent down, [K, 2K)
//   high edge  [L-2K-n+s, L-s)      ends as [L-2K, L-n): covers the planes sent up, [nz, nz+K)
//   interior   [2K+n-s, L-2K-n+s)   grows by one plane per side per sub-step
// Each edge chain reads only its own previous sub-step; both zones go into ONE launch per sub-step (the packed
// class kernels take two output ranges: a second stream with fork / join events inside the captured graph cost
// ~40 us of host time per graph launch). The interior reads one plane of each edge per sub-step, which the edge
// chains — finished first — never overwrite afterwards (their ranges shrink).
int slab_batch(ca3d_engine *h, uint32_t n_steps, int phase)
{
	int rc = check_ready(h);
	if (rc) return rc;
	if (!h->slab) return fail(CA3D_ERR_INVALID_ARGUMENT, "engine is not a slab: use ca3d_step");
	if (n_steps > h->ghost) return fail(CA3D_ERR_INVALID_ARGUMENT, "%u sub-steps exceed the ghost depth %u", n_steps, h->ghost);
	if (phase == CA3D_SLAB_PHASE_EDGES && h->pending_edges) return fail(CA3D_ERR_INVALID_ARGUMENT, "edge phase issued twice: the interior phase must follow");
	if (phase == CA3D_SLAB_PHASE_INTERIOR && h->pending_edges != n_steps) return fail(CA3D_ERR_INVALID_ARGUMENT, "interior phase of %u sub-steps does not follow an edge phase of the same length", n_steps);
	if (phase == CA3D_SLAB_PHASE_ALL && h->pending_edges) return fail(CA3D_ERR_INVALID_ARGUMENT, "an edge phase is pending: finish it with the interior phase");
	rc = bind_device(h);
	if (rc) return rc;
	if (n_steps == 0) return CA3D_OK;
	h->buffers_exposed = false;
	const uint32_t L = h->nplanes, K = h->ghost, n = n_steps;
	// The packed kernel's bottom face is dead (z == -1 is dropped): the slab that owns global plane 0 never needs
	// its low ghost.
	const uint32_t lo_floor = (h->layout == CA3D_LAYOUT_PACKED32 && h->z0 == 0) ? K : 0u;
	const bool splittable = h->nz + 2u > 2u * K + 2u * n; // interior non-empty in every sub-step
	if (phase != CA3D_SLAB_PHASE_INTERIOR && h->want_stats) HIP_TRY(hipEventRecord(h->ev_start, h->stream));
	int what = phase; // what this call enqueues
	if (!splittable)
	{
		// thin slab: the edge phase does the whole batch, the interior phase only commits it
		what = phase == CA3D_SLAB_PHASE_INTERIOR ? -1 : CA3D_SLAB_PHASE_ALL;
	}
	auto enqueue_all = [&](uint32_t start_buf) -> int {
		uint32_t cur = start_buf;
		for (uint32_t s = 1; s <= n; s++, cur ^= 1u)
		{
			const uint32_t lo = s > lo_floor ? s : lo_floor, hi = L - s;
			const uint32_t e_lo = 2u * K + n - s, e_hi = L - 2u * K - n + s;
			int r2 = CA3D_OK;
			if (what == CA3D_SLAB_PHASE_ALL) r2 = enqueue_step(h, (int)cur, lo, hi, h->stream);
			else if (what == CA3D_SLAB_PHASE_INTERIOR) r2 = enqueue_step(h, (int)cur, e_lo, e_hi, h->stream);
			else r2 = enqueue_step(h, (int)cur, lo, e_lo, h->stream, false, e_hi, hi); // both edge zones, one launch
			if (r2) return r2;
		}
		return CA3D_OK;
	};
	const bool graphable = h->use_graph && h->stream != nullptr && n > 1 &&
	                       !(h->layout == CA3D_LAYOUT_UNPACKED && !h->binary_state);
	const bool resident = what == CA3D_SLAB_PHASE_ALL && h->res_slab_fn && h->use_resident && !h->res_failed && n >= h->res_min && h->stream != nullptr;
	if (what < 0) { /* nothing to enqueue */ }
	else if (resident)
	{
		rc = resident_slab_steps(h, n);
		if (rc) return rc;
	}
	else if (graphable)
	{
		// one graph launch per batch: the host cost of a K-step batch must stay below its GPU time for the ranks
		// to scale (8 launches of ~7 us kernels would otherwise be host-bound)
		const uint64_t key = ((uint64_t)what << 40) | ((uint64_t)h->cur << 32) | n;
		auto it = h->slab_graphs.find(key);
		if (it == h->slab_graphs.end())
		{
			hipGraph_t graph = nullptr;
			HIP_TRY(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
			rc = enqueue_all(h->cur);
			hipError_t e = hipStreamEndCapture(h->stream, &graph);
			if (rc != CA3D_OK) { if (graph) hipGraphDestroy(graph); return rc; }
			if (e != hipSuccess) return fail(CA3D_ERR_DEVICE, "hipStreamEndCapture: %s", hipGetErrorString(e));
			hipGraphExec_t exec = nullptr;
			e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
			hipGraphDestroy(graph);
			if (e != hipSuccess) return fail(CA3D_ERR_DEVICE, "hipGraphInstantiate: %s", hipGetErrorString(e));
			it = h->slab_graphs.emplace(key, exec).first;
		}
		HIP_TRY(hipGraphLaunch(it->second, h->stream));
	}
	else
	{
		rc = enqueue_all(h->cur);
		if (rc) return rc;
	}
	if (h->layout == CA3D_LAYOUT_UNPACKED) h->binary_state = true;
	if (phase == CA3D_SLAB_PHASE_EDGES)
	{
		h->pending_edges = n; // ca3d_slab_region now refers to the buffer the batch ends in
		return CA3D_OK;
	}
	h->pending_edges = 0;
	h->step += n;
	h->cur = (h->cur + n) & 1u;
	h->ghosts_valid = false; // the caller (or ca3d_slab_run) refreshes them
	if (h->want_stats) HIP_TRY(hipEventRecord(h->ev_stop, h->stream));
	h->ev_valid = h->want_stats != 0;
	h->stats.steps = n;
	h->stats.kernel_launches = resident ? 1u : (phase == CA3D_SLAB_PHASE_ALL || !splittable ? n : 2u * n);
	h->launches_total += h->stats.kernel_launches;
	h->stats.cell_steps = (double)n * h->cells_per_plane() * h->nz; // owned cells only: ghost recompute is overhead
	h->stats.algorithmic_bytes = h->stats.cell_steps * h->bytes_per_cell_step();
	return CA3D_OK;
}

int ca3d_slab_step(ca3d_t *h, uint32_t n_steps) CA3D_API_TRY
{
	return slab_batch(h, n_steps, CA3D_SLAB_PHASE_ALL);
}
CA3D_API_CATCH

int ca3d_slab_step_phase(ca3d_t *h, uint32_t n_steps, int phase) CA3D_API_TRY
{
	if (phase != CA3D_SLAB_PHASE_ALL && phase != CA3D_SLAB_PHASE_EDGES && phase != CA3D_SLAB_PHASE_INTERIOR)
		return fail(CA3D_ERR_INVALID_ARGUMENT, "unknown slab phase %d", phase);
	return slab_batch(h, n_steps, phase);
}
CA3D_API_CATCH

int ca3d_slab_region(ca3d_t *h, int region, void **device_ptr, size_t *n_bytes) CA3D_API_TRY
{
	if (!h || !device_ptr || !n_bytes) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL argument");
	if (!h->configured || !h->slab) return fail(CA3D_ERR_NOT_CONFIGURED, "engine is not configured as a slab");
	uint32_t *base = h->buf[(h->cur + h->pending_edges) & 1u]; // after an edge phase: the buffer its results are in
	const size_t pw = h->plane_words;
	const uint32_t K = h->ghost, nz = h->nz;
	size_t first = 0, count = K;
	switch (region)
	{
	case CA3D_SLAB_SEND_LOW: first = K; break;
	case CA3D_SLAB_SEND_HIGH: first = nz; break; // K + nz - K
	case CA3D_SLAB_RECV_LOW: first = 0; break;
	case CA3D_SLAB_RECV_HIGH: first = (size_t)K + nz; break;
	case CA3D_SLAB_OWNED: first = K; count = nz; break;
	default: return fail(CA3D_ERR_INVALID_ARGUMENT, "unknown slab region %d", region);
	}
	*device_ptr = base + first * pw;
	*n_bytes = count * pw * sizeof(uint32_t);
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_comm_unique_id(void *id) CA3D_API_TRY
{
	if (!id) return fail(CA3D_ERR_INVALID_ARGUMENT, "id is NULL");
	Rccl &r = rccl();
	if (!r.error.empty()) return fail(CA3D_ERR_UNSUPPORTED, "%s", r.error.c_str());
	NCCL_TRY(r.GetUniqueId(id));
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_slab_comm_init(ca3d_t *h, const void *id, int rank, int world) CA3D_API_TRY
{
	if (!h || !id) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL argument");
	if (world < 1 || rank < 0 || rank >= world) return fail(CA3D_ERR_INVALID_ARGUMENT, "rank %d of %d", rank, world);
	Rccl &r = rccl();
	if (!r.error.empty()) return fail(CA3D_ERR_UNSUPPORTED, "%s", r.error.c_str());
	int rc = bind_device(h);
	if (rc) return rc;
	if (h->comm) { r.CommDestroy(h->comm); h->comm = nullptr; }
	ncclUniqueIdBytes uid;
	memcpy(&uid, id, sizeof uid);
	NCCL_TRY(r.CommInitRank(&h->comm, world, uid, rank));
	h->comm_rank = rank;
	h->comm_world = world;
	if (!h->comm_stream)
	{
		HIP_TRY(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
		HIP_TRY(hipEventCreateWithFlags(&h->ev_edges, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&h->ev_comm, hipEventDisableTiming));
	}
	h->ghosts_valid = false;
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_slab_comm_info(ca3d_t *h, ca3d_comm_info *out) CA3D_API_TRY
{
	if (!h || !out) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL argument");
	memset(out, 0, sizeof *out);
	out->comm_ranks = out->comm_rank = out->comm_device = -1;
	out->device = h->device;
	int rc = bind_device(h);
	if (rc) return rc;
	HIP_TRY(hipDeviceGetPCIBusId(out->pci_bus_id, (int)sizeof out->pci_bus_id, h->device));
	if (h->comm)
	{
		// what the COMMUNICATOR says, not what the caller passed to ca3d_slab_comm_init
		NCCL_TRY(rccl().CommCount(h->comm, &out->comm_ranks));
		NCCL_TRY(rccl().CommUserRank(h->comm, &out->comm_rank));
		NCCL_TRY(rccl().CommCuDevice(h->comm, &out->comm_device));
	}
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_slab_exchange(ca3d_t *h) CA3D_API_TRY
{
	int rc = check_ready(h);
	if (rc) return rc;
	if (!h->slab || !h->comm) return fail(CA3D_ERR_NOT_CONFIGURED, "engine is not a slab with a communicator (ca3d_configure_slab, ca3d_slab_comm_init)");
	rc = bind_device(h);
	if (rc) return rc;
	rc = comm_exchange(h, h->stream);
	if (rc == CA3D_OK) h->ghosts_valid = true;
	return rc;
}
CA3D_API_CATCH

int ca3d_slab_run(ca3d_t *h, uint32_t n_steps, int overlap) CA3D_API_TRY
{
	int rc = check_ready(h);
	if (rc) return rc;
	if (!h->slab || !h->comm) return fail(CA3D_ERR_NOT_CONFIGURED, "engine is not a slab with a communicator (ca3d_configure_slab, ca3d_slab_comm_init)");
	if (h->pending_edges) return fail(CA3D_ERR_INVALID_ARGUMENT, "an edge phase is pending: finish it with the interior phase");
	rc = bind_device(h);
	if (rc) return rc;
	if (!h->ghosts_valid)
	{
		rc = comm_exchange(h, h->stream);
		if (rc) return rc;
		h->ghosts_valid = true;
	}
	const bool keep_stats = h->want_stats != 0;
	hipEvent_t first = nullptr;
	uint32_t left = n_steps;
	uint64_t launches = 0;
	if (keep_stats && left) HIP_TRY(hipEventRecord(h->ev_start, h->stream));
	h->want_stats = 0; // the batches below would each re-record the pair
	auto restore = [&]() { h->want_stats = keep_stats ? 1 : 0; };
	while (left)
	{
		const uint32_t k = left < h->ghost ? left : h->ghost;
		if (overlap)
		{
			// edge zones -> their planes travel on the communication stream while the interior runs -> the next batch
			// (and anything else on the engine's stream) waits for the receives
			rc = slab_batch(h, k, CA3D_SLAB_PHASE_EDGES);
			if (rc) { restore(); return rc; }
			HIP_TRY(hipEventRecord(h->ev_edges, h->stream));
			HIP_TRY(hipStreamWaitEvent(h->comm_stream, h->ev_edges, 0));
			rc = comm_exchange(h, h->comm_stream);
			if (rc) { restore(); return rc; }
			HIP_TRY(hipEventRecord(h->ev_comm, h->comm_stream));
			rc = slab_batch(h, k, CA3D_SLAB_PHASE_INTERIOR);
			if (rc) { restore(); return rc; }
			HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_comm, 0));
		}
		else
		{
			rc = slab_batch(h, k, CA3D_SLAB_PHASE_ALL);
			if (rc) { restore(); return rc; }
			rc = comm_exchange(h, h->stream);
			if (rc) { restore(); return rc; }
		}
		h->ghosts_valid = true;
		launches += h->stats.kernel_launches;
		left -= k;
	}
	restore();
	if (keep_stats && n_steps)
	{
		HIP_TRY(hipEventRecord(h->ev_stop, h->stream));
		h->ev_valid = true;
		h->stats.steps = n_steps;
		h->stats.kernel_launches = launches;
		h->stats.cell_steps = (double)n_steps * h->cells_per_plane() * h->nz;
		h->stats.algorithmic_bytes = h->stats.cell_steps * h->bytes_per_cell_step();
	}
	(void)first;
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_slab_gather(ca3d_t *h, ca3d_t *full) CA3D_API_TRY
{
	if (!h || !full) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL argument");
	if (!h->slab || !h->comm || !h->has_state) return fail(CA3D_ERR_NOT_CONFIGURED, "engine is not a slab with a communicator and a state");
	if (!full->configured || full->slab || full->G != h->G || full->layout != h->layout || full->device != h->device)
		return fail(CA3D_ERR_INVALID_ARGUMENT, "the target must be a full-grid engine of the same grid, layout and device");
	FLUSH_QUEUED(full);
	if (int rcs = settle_resident(full)) return rcs;
	if ((size_t)h->nz * h->comm_world != h->G) return fail(CA3D_ERR_UNSUPPORTED, "the slabs must split the grid evenly");
	int rc = bind_device(h);
	if (rc) return rc;
	void *owned;
	size_t bytes;
	rc = ca3d_slab_region(h, CA3D_SLAB_OWNED, &owned, &bytes);
	if (rc) return rc;
	// ncclAllGather straight between the engines' device buffers, in rank (= z) order, on the slab engine's stream. When the
	// target engine runs on another stream the gather waits for what that stream still does with the buffer (a frame being
	// rendered from it) and that stream waits for the gather before it touches the buffer again.
	const bool cross = full->stream != h->stream;
	if (cross)
	{
		if (!h->ev_gather) HIP_TRY(hipEventCreateWithFlags(&h->ev_gather, hipEventDisableTiming));
		HIP_TRY(hipEventRecord(h->ev_gather, full->stream));
		HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_gather, 0));
	}
	NCCL_TRY(rccl().AllGather(owned, full->buf[full->cur], bytes / sizeof(uint32_t), kNcclUint32, h->comm, h->stream));
	if (cross)
	{
		HIP_TRY(hipEventRecord(h->ev_gather, h->stream));
		HIP_TRY(hipStreamWaitEvent(full->stream, h->ev_gather, 0));
	}
	full->has_state = true;
	full->state_serial++;
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_render_target(ca3d_t *h, int which, void **device_ptr, size_t *n_bytes) CA3D_API_TRY
{
	if (!h || !device_ptr || !n_bytes) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL argument");
	if (!h->r_present) return fail(CA3D_ERR_NOT_CONFIGURED, "ca3d_render has not been called yet");
	if (int rcb = bind_device(h)) return rcb; // (whoever reads the target after the engine's stream also reads it after the frames in flight)
	const size_t px = (size_t)h->rw * h->rh;
	switch (which)
	{
	case 0: *device_ptr = h->r_present; *n_bytes = px * 4; break;
	case 1: *device_ptr = h->r_light[h->r_swap ^ 1]; *n_bytes = px * 8; break; // the surfaces the LAST frame was written to
	case 2: *device_ptr = h->r_depth[h->r_swap ^ 1]; *n_bytes = px * 4; break;
	default: return fail(CA3D_ERR_INVALID_ARGUMENT, "target must be 0 (presentation), 1 (light) or 2 (depth)");
	}
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_synchronize(ca3d_t *h) CA3D_API_TRY
{
	if (!h) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL engine handle");
	FLUSH_QUEUED(h);
	int rc = bind_device(h);
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(h->stream));
	return check_resident(h);
}
CA3D_API_CATCH

int ca3d_measure_copy(ca3d_t *h, size_t n_bytes, uint32_t reps, double *gb_per_s) CA3D_API_TRY
{
	if (!h || !gb_per_s) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL argument");
	if (n_bytes < (1u << 20) || n_bytes % 16u || reps == 0 || reps > 4096u) return fail(CA3D_ERR_INVALID_ARGUMENT, "n_bytes must be a multiple of 16 of at least 1 MiB, reps in [1, 4096]");
	FLUSH_QUEUED(h);
	int rc = bind_device(h);
	if (rc) return rc;
	void *a = nullptr, *b = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	auto cleanup = [&]() { if (a) hipFree(a); if (b) hipFree(b); if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); };
#define COPY_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(e_ == hipErrorOutOfMemory ? CA3D_ERR_OUT_OF_MEMORY : CA3D_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); } } while (0)
	COPY_TRY(hipMalloc(&a, n_bytes));
	COPY_TRY(hipMalloc(&b, n_bytes));
	COPY_TRY(hipEventCreate(&e0));
	COPY_TRY(hipEventCreate(&e1));
	COPY_TRY(hipMemsetAsync(a, 0x5A, n_bytes, h->stream));
	COPY_TRY(launch_copy_f4(a, b, n_bytes, h->stream)); // warm: page tables, clocks
	COPY_TRY(hipEventRecord(e0, h->stream));
	for (uint32_t i = 0; i < reps; i++) COPY_TRY(launch_copy_f4((i & 1u) ? b : a, (i & 1u) ? a : b, n_bytes, h->stream));
	COPY_TRY(hipEventRecord(e1, h->stream));
	COPY_TRY(hipEventSynchronize(e1));
	float ms = 0.f;
	COPY_TRY(hipEventElapsedTime(&ms, e0, e1));
#undef COPY_TRY
	cleanup();
	*gb_per_s = ms > 0.f ? 2.0 * (double)n_bytes * reps / (ms * 1e-3) / 1e9 : 0.0; // bytes read + bytes written
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_recovered_launches(ca3d_t *h, uint32_t *out_count) CA3D_API_TRY
{
	if (!h || !out_count) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL argument");
	*out_count = h->res_recovered;
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_set_stream(ca3d_t *h, void *hip_stream) CA3D_API_TRY
{
	if (!h) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL engine handle");
	FLUSH_QUEUED(h);
	int rc = bind_device(h);
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(h->stream));
	rc = check_resident(h);
	if (rc) return rc;
	resident_stream_retired(h->stream);
	drop_graph(h);
	h->stream = (hipStream_t)hip_stream;
	h->ev_valid = false;
	refresh_kernels(h); // the new stream may be confined to fewer CUs (CU mask): residency is checked per stream
	note_jit_failure(h);
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_use_own_stream(ca3d_t *h) CA3D_API_TRY
{
	if (!h) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL engine handle");
	FLUSH_QUEUED(h);
	int rc = bind_device(h);
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(h->stream));
	rc = check_resident(h);
	if (rc) return rc;
	resident_stream_retired(h->stream);
	drop_graph(h);
	h->stream = h->own_stream;
	h->ev_valid = false;
	refresh_kernels(h);
	note_jit_failure(h);
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_device_buffer(ca3d_t *h, int which, void **device_ptr, size_t *n_bytes) CA3D_API_TRY
{
	if (!h || !device_ptr || !n_bytes) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL argument");
	if (!h->configured) return fail(CA3D_ERR_NOT_CONFIGURED, "ca3d_configure has not been called");
	if (which != 0 && which != 1) return fail(CA3D_ERR_INVALID_ARGUMENT, "buffer index must be 0 or 1");
	FLUSH_QUEUED(h);
	if (int rcs = settle_resident(h)) return rcs;
	*device_ptr = h->buf[which];
	*n_bytes = h->buffer_words() * sizeof(uint32_t);
	h->state_serial++; // the caller may write through the pointer: what the renderer derived from the state is stale from here on,
	h->buffers_exposed = true; // and again before every frame while the pointer is valid (ca3d_render)
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_get_info(ca3d_t *h, ca3d_info *out) CA3D_API_TRY
{
	if (!h || !out) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL argument");
	FLUSH_QUEUED(h);
	memset(out, 0, sizeof *out);
	out->grid_size = h->G;
	out->layout = h->layout;
	out->z0 = h->z0;
	out->nz = h->nz;
	out->ghost = h->ghost;
	out->step = h->step;
	out->state_words = h->configured ? h->state_words() : 0;
	out->current_buffer = (int32_t)h->cur;
	out->device = h->device;
	out->launches_total = h->launches_total;
	const char *name = "";
	if (h->configured && h->rules.valid)
	{
		if (h->layout != CA3D_LAYOUT_PACKED32) name = h->step > 0 && h->kernel_name[0] ? h->kernel_name : "ca_unpacked";
		else if (h->use_fused && !h->slab && packed_fused_steps(h->rules, h->G, h->variant) == 2) name = "ca_packed_fused+ca_packed_class";
		else if (h->res_ready && h->use_resident && !h->res_failed && h->res_class) name = "ca_resident_class(jit)";
		else if (h->res_ready && h->use_resident && !h->res_failed) name = h->res_jit_fn ? "ca_resident_vn(jit)" : "ca_resident_vn";
		else if (h->slab && h->res_slab_fn && h->use_resident && !h->res_failed) name = "ca_resident_slab_vn(jit)";
		else if (h->roll_jit.cv_np2 > 0) name = "ca_packed_roll_np2(jit)";
		else if (h->rows_jit.main >= 0) name = "ca_packed_rows(jit)";
		else name = h->vn_jit.cvl >= 0 ? "ca_packed_vn(jit)" : packed_kernel_name(h->rules, h->G, h->variant);
	}
	const bool class_jit = h->configured && h->rules.valid && h->layout == CA3D_LAYOUT_PACKED32 && h->class_jit.main >= 0 && strncmp(name, "ca_resident", 11) != 0 &&
	                       !(h->use_fused && !h->slab && packed_fused_steps(h->rules, h->G, h->variant) == 2);
	if (class_jit && h->roll_jit.cvl >= 0 && !strncmp(name, "ca_packed_class", 15))
		snprintf(out->kernel_name, sizeof out->kernel_name, "ca_packed_class_roll%s(jit)", name + 15); // rolling-window form
	else
		snprintf(out->kernel_name, sizeof out->kernel_name, "%s%s", name, class_jit ? "(jit)" : "");
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_get_jit_log(ca3d_t *h, char *buf, size_t n_bytes, size_t *needed) CA3D_API_TRY
{
	if (!h) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL engine handle");
	if (needed) *needed = h->jit_log.size() + 1;
	if (buf && n_bytes)
	{
		const size_t n = h->jit_log.size() < n_bytes - 1 ? h->jit_log.size() : n_bytes - 1;
		memcpy(buf, h->jit_log.data(), n);
		buf[n] = '\0';
	}
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_get_jit_stats(ca3d_jit_stats *out) CA3D_API_TRY
{
	if (!out) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL argument");
	jit_stats(out);
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_get_kernel_variant(ca3d_t *h, char *buf, size_t n_bytes, size_t *needed) CA3D_API_TRY
{
	if (!h) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL engine handle");
	ca3d_info info;
	int rc = ca3d_get_info(h, &info);
	if (rc) return rc;
	uint64_t rh = 1469598103934665603ull;
	auto mix = [&](const void *p, size_t n) { for (size_t i = 0; i < n; i++) { rh ^= static_cast<const unsigned char *>(p)[i]; rh *= 1099511628211ull; } };
	if (h->rules.valid)
	{
		mix(&h->rules.lists, sizeof h->rules.lists);
		mix(h->rules.survive_raw, sizeof h->rules.survive_raw);
		mix(h->rules.born_raw, sizeof h->rules.born_raw);
	}
	const char *zs = getenv("CA3D_RC256_ZS");
	char text[320];
	const bool res = !strncmp(info.kernel_name, "ca_resident", 11);
	if (res)
		snprintf(text, sizeof text, "%s;G=%u;rule=%016llx;rows=%u;zsplit=%u;pair=%d;rc256zs=%s;src=%016llx", info.kernel_name, h->G, (unsigned long long)rh,
		         (h->res_class || vn_pair(h)) ? 32u : h->res_rows, h->res_class ? resident_class_zsplit(h->G) : h->res_zsplit, h->res_class ? 0 : vn_pair(h),
		         zs ? zs : "-", (unsigned long long)jit_sources_hash());
	else
		snprintf(text, sizeof text, "%s;G=%u;rule=%016llx;variant=%d;src=%016llx", info.kernel_name, h->G, (unsigned long long)rh, h->variant, (unsigned long long)jit_sources_hash());
	const size_t len = strlen(text);
	if (needed) *needed = len + 1;
	if (buf && n_bytes)
	{
		const size_t n = len < n_bytes - 1 ? len : n_bytes - 1;
		memcpy(buf, text, n);
		buf[n] = '\0';
	}
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_get_stats(ca3d_t *h, ca3d_stats *out) CA3D_API_TRY
{
	if (!h || !out) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL argument");
	FLUSH_QUEUED(h);
	if (!h->ev_valid) return fail(CA3D_ERR_NOT_CONFIGURED, "no step batch has been timed yet (or option \"stats\" is 0)");
	int rc = bind_device(h);
	if (rc) return rc;
	HIP_TRY(hipEventSynchronize(h->ev_stop));
	float ms = 0.f;
	HIP_TRY(hipEventElapsedTime(&ms, h->ev_start, h->ev_stop));
	h->stats.gpu_ms = ms;
	*out = h->stats;
	HIP_TRY(hipStreamSynchronize(h->stream));
	return check_resident(h);
}
CA3D_API_CATCH

int ca3d_render(ca3d_t *h, const float uniforms[128], uint32_t width, uint32_t height, uint32_t spp,
                uint8_t *presentation_rgba8, uint16_t *light_rgba16f, uint16_t *depth_rg16f) CA3D_API_TRY
{
	if (!h) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL engine handle");
	if (!h->configured || !h->has_state) return fail(CA3D_ERR_NOT_CONFIGURED, "no state to render: configure and upload first");
	FLUSH_QUEUED(h);
	if (h->slab) return fail(CA3D_ERR_UNSUPPORTED, "the renderer reads a full grid, not a slab");
	if (h->layout == CA3D_LAYOUT_UNPACKED && h->render_mode != 0) return fail(CA3D_ERR_UNSUPPORTED, "the literal frame mode is implemented for the packed layout only");
	if (!uniforms) return fail(CA3D_ERR_INVALID_ARGUMENT, "uniforms is NULL");
	if (width == 0 || height == 0 || width > 16384u || height > 16384u) return fail(CA3D_ERR_INVALID_ARGUMENT, "bad target size %ux%u", width, height);
	if (spp != 1 && spp != 4) return fail(CA3D_ERR_INVALID_ARGUMENT, "spp must be 1 or 4");
	if (h->render_mode == 1 && spp != 1) return fail(CA3D_ERR_INVALID_ARGUMENT, "the literal frame mode takes one jittered sample per pixel (spp = 1)");
	// frames in flight (FrameLane): converged frames of a packed volume that stay on the device and go down the stream passes
	static const char *trace_path = getenv("CA3D_RENDER_TRACE");
	static const bool aux_off = getenv("CA3D_RENDER_AUX") && atoi(getenv("CA3D_RENDER_AUX")) == 0; // tuning: everything on one stream
	// The first frame after a step, an upload or any other call on the engine's stream is drawn ON that stream: it has to wait for that call,
	// which waited for every earlier frame — nothing can be in flight beside it, and on a lane it would only pay two cross-stream hand-offs
	// (a host that steps between frames: 0.628 against 0.564 ms per step + frame, tools/run_render_step_loop.py). The frames behind it go
	// down the lanes.
	const bool first_after_touch = h->state_touched;
	const bool pipelined = !first_after_touch && h->render_pipeline && h->render_mode == 0 && !presentation_rgba8 && !light_rgba16f && !depth_rg16f && h->stream == h->own_stream &&
	                       h->layout == CA3D_LAYOUT_PACKED32 && h->render_stream && h->render_sched && !h->render_indirect && !h->render_stream_check && !trace_path &&
	                       !aux_off && !h->render_row0 && !h->render_row1 && width == h->rw && height == h->rh;
	int rc = bind_device(h, !pipelined);
	if (rc) return rc;
	// The frame shows a state the engine has verified — when the caller gets the frame back on the host. A frame that stays on the
	// device (no host pointers: the reference's render pass, which only enqueues) does not block on the step batch in front of it: a
	// resident launch that has ALREADY given up (pinned flag set) is recovered first, one still running is left pending — the next
	// call that waits for the stream verifies it, and a frame drawn from a launch that later turns out to have timed out (a foreign
	// kernel holding CUs for 200 ms) is simply the wrong frame once.
	if (presentation_rgba8 || light_rgba16f || depth_rg16f || (h->res_status_host && *h->res_status_host)) rc = settle_resident(h);
	if (rc) return rc;
	const size_t px = (size_t)width * height;
	if (width != h->rw || height != h->rh)
	{
		// _createResolutionDependentAssests (main_pathtraced.js:729-779)
		HIP_TRY(hipStreamSynchronize(h->stream)); // (a frame of a new size is never pipelined: the lanes were joined above)
		free_render_targets(h);
		HIP_TRY(hipMalloc((void **)&h->r_present, px * 4));
		for (int i = 0; i < 2; i++)
		{
			HIP_TRY(hipMalloc(&h->r_light[i], px * 8));
			HIP_TRY(hipMalloc((void **)&h->r_depth[i], px * 4));
			HIP_TRY(hipMemsetAsync(h->r_light[i], 0, px * 8, h->stream));
			HIP_TRY(hipMemsetAsync(h->r_depth[i], 0, px * 4, h->stream));
		}
		h->rw = width;
		h->rh = height;
		h->r_swap = 0;
	}
	// diagnostics: CA3D_RENDER_TRACE=<file> makes every wave of the scheduled kernel record when and where it ran
	// (tools/render_trace.py draws the occupancy timeline from the file)
	const size_t trace_waves = trace_path ? ((size_t)(width + 31u) / 32u * 2u) * ((height + 15u) / 16u * 4u) : 0u; // wave tiles of 16 x 4 pixels
	const size_t counter_words = 8u + 4u * trace_waves;
	if (h->r_counters && h->r_counter_words < counter_words) { HIP_TRY(hipFree(h->r_counters)); h->r_counters = nullptr; }
	if (!h->r_counters)
	{
		HIP_TRY(hipMalloc((void **)&h->r_counters, counter_words * sizeof(unsigned long long)));
		h->r_counter_words = counter_words;
	}
	// where this frame runs: the engine's stream, or the next lane
	ca3d_engine::FrameLane *L = nullptr;
	int walk_share = 100;
	hipStream_t rs = h->stream;
	unsigned long long *counters = h->r_counters;
	const size_t frame_samples = (size_t)width * height * spp;
	const int want_lanes = h->render_pipeline >= 2 ? (h->render_pipeline < ca3d_engine::kMaxLanes ? h->render_pipeline : ca3d_engine::kMaxLanes) : render_default_lanes(frame_samples);
	if (pipelined && h->n_lanes < want_lanes && !h->lanes_exhausted)
	{
		// the lanes: streams that the runtime has put on pairwise DIFFERENT hardware queues (probed: ca_diag.hip) — two streams on one queue
		// run in order and a frame would only queue up behind the other. Fewer than two such streams: no pipeline.
		for (auto &fl : h->lanes)
			if (fl.s && fl.pending) HIP_TRY(hipStreamSynchronize(fl.s)); // (the probe needs idle streams)
		if (!h->lanes[0].s) { HIP_TRY(hipStreamCreateWithFlags(&h->lanes[0].s, hipStreamNonBlocking)); h->n_lanes = 1; }
		for (int tries = 0; tries < 10 && h->n_lanes < want_lanes; tries++)
		{
			hipStream_t cand = nullptr;
			HIP_TRY(hipStreamCreateWithFlags(&cand, hipStreamNonBlocking));
			bool side_by_side = true;
			for (int i = 0; i < h->n_lanes && side_by_side; i++) HIP_TRY(streams_concurrent(h->lanes[i].s, cand, &side_by_side));
			if (side_by_side) h->lanes[h->n_lanes++].s = cand;
			else h->lane_spares.push_back(cand);
		}
		if (h->n_lanes < want_lanes) h->lanes_exhausted = true; // the runtime has no more queues to give: do not probe again on every frame
		if (h->n_lanes < 2) h->render_pipeline = 0;
		else
			for (int i = 0; i < h->n_lanes; i++)
			{
				ca3d_engine::FrameLane &fl = h->lanes[i];
				if (fl.done) continue;
				HIP_TRY(hipEventCreateWithFlags(&fl.done, hipEventDisableTiming));
				HIP_TRY(hipEventCreate(&fl.start));
				HIP_TRY(hipEventCreate(&fl.stop));
				HIP_TRY(hipMalloc((void **)&fl.counters, 8u * sizeof(unsigned long long)));
			}
		if (!h->ev_state) HIP_TRY(hipEventCreateWithFlags(&h->ev_state, hipEventDisableTiming));
	}
	const int active_lanes = h->n_lanes < want_lanes ? h->n_lanes : want_lanes;
	if (pipelined && h->render_pipeline && active_lanes >= 2)
	{
		if (h->lane_next >= active_lanes) h->lane_next = 0;
		L = &h->lanes[h->lane_next];
		h->lanes_in_use = active_lanes;
		// is another frame still in flight beside this one? Then this frame's walks take their share of the chip (render_walk_share); a
		// frame that finds the lanes idle — a host that draws one frame per display refresh — takes the whole chip and is done sooner.
		// (A frame that has to wait for the engine's stream — a step or an upload since the last frame — starts after every earlier frame:
		// the engine's stream joined them before that call's work. It runs alone whatever is still in flight now.)
		bool beside = false;
		for (int i = 0; i < active_lanes && !beside && !h->main_touched; i++)
			if (&h->lanes[i] != L && h->lanes[i].used)
			{
				const hipError_t q = hipEventQuery(h->lanes[i].done);
				if (q == hipErrorNotReady) { beside = true; (void)hipGetLastError(); }
				else if (q != hipSuccess) HIP_TRY(q);
			}
		walk_share = beside ? render_walk_share(frame_samples, active_lanes) : 100;
		if (h->main_touched)
		{
			// the steps and uploads in front of this frame — recorded only when an entry point has touched the engine's stream since the last
			// record (a marker behind another lane's frames on a shared hardware queue would make this frame wait for them)
			HIP_TRY(hipEventRecord(h->ev_state, h->stream));
			h->main_touched = false;
			for (auto &fl : h->lanes) fl.need_state = true;
		}
		if (L->need_state)
		{
			HIP_TRY(hipStreamWaitEvent(L->s, h->ev_state, 0));
			L->need_state = false;
		}
		rs = L->s;
		counters = L->counters;
	}
	HIP_TRY(hipMemsetAsync(counters, 0, (trace_path ? counter_words : 8u) * sizeof(unsigned long long), rs)); // [3]: the tile queue's head
	RenderLaunch l;
	l.trace = trace_path != nullptr;
	l.walk_share_pct = walk_share;
	l.cells = h->buf[h->cur];
	l.G = h->G;
	l.W = width;
	l.H = height;
	l.spp = spp;
	l.uniforms = uniforms;
	l.presentation = h->r_present;
	l.light = h->r_light[h->r_swap];
	l.depth = h->r_depth[h->r_swap];
	l.counters = counters;
	if (h->buffers_exposed) h->state_serial++; // a caller holds a pointer to the state and may have written it since the last frame
	const uint64_t state_key[3] = {h->state_serial, h->step, (uint64_t)(uintptr_t)l.cells}; // what the occupancy bits / the bricks were built from
	bool occ_built = false, bricks_built = false;
	if (h->render_skip && h->layout == CA3D_LAYOUT_PACKED32)
	{
		const size_t fine = (size_t)(h->G / 32u) * (h->G / 8u) * (h->G / 8u); // fine bits, count word, coarse bits (render.hip)
		const size_t words = (fine + 63u) / 64u + 1u + (fine / 64u + 63u) / 64u + 3u; // (+ the six words of the live box)
		if (words != h->r_occ_words)
		{
			h->r_occ_key[0] = 0;
			for (auto &fl : h->lanes)
				if (fl.s) HIP_TRY(hipStreamSynchronize(fl.s));
			if (h->r_occ) HIP_TRY(hipFree(h->r_occ));
			h->r_occ = nullptr;
			h->r_occ_words = 0;
			HIP_TRY(hipMalloc((void **)&h->r_occ, words * sizeof(unsigned long long)));
			h->r_occ_words = words;
		}
		l.occ = h->r_occ;
		l.occ_valid = !memcmp(state_key, h->r_occ_key, sizeof state_key);
		l.occ_built = &occ_built;
	}
	l.mode = h->render_mode;
	l.sched = h->render_sched;
	l.indirect = h->render_indirect != 0;
	if (l.indirect && (h->render_mode != 0 || h->layout != CA3D_LAYOUT_PACKED32)) return fail(CA3D_ERR_UNSUPPORTED, "render_indirect is implemented for the converged-frame mode over the packed layout");
	l.row0 = h->render_row0;
	l.row1 = h->render_row1 > height ? height : h->render_row1;
	if (l.row1 && l.row0 >= l.row1) return fail(CA3D_ERR_INVALID_ARGUMENT, "render rows [%u, %u) are empty for a target of %u rows", l.row0, h->render_row1, height);
	if ((l.row0 || l.row1) && h->render_mode != 0) return fail(CA3D_ERR_UNSUPPORTED, "row bands are implemented for the converged-frame mode only");
	l.legacy = h->layout == CA3D_LAYOUT_UNPACKED; // legacy volume -> legacy shader (pathtraced_fragment.wgsl)
	l.prev_light = h->r_light[h->r_swap ^ 1]; // group 1 of the render pass: last frame's targets (1519-1555, 1787)
	l.prev_depth = h->r_depth[h->r_swap ^ 1];
	if (L) {} // (a lane is ONE stream: its side kernels run behind its stream passes, the other lane's frame fills the chip meanwhile)
	else if (!aux_off && h->render_mode == 0 && h->render_sched && !trace_path)
	{
		if (!h->r_aux)
		{
			HIP_TRY(hipStreamCreateWithFlags(&h->r_aux, hipStreamNonBlocking));
			HIP_TRY(hipEventCreateWithFlags(&h->r_fork, hipEventDisableTiming));
			HIP_TRY(hipEventCreateWithFlags(&h->r_join, hipEventDisableTiming));
		}
		l.aux = h->r_aux;
		l.ev_fork = h->r_fork;
		l.ev_join = h->r_join;
	}
	if (h->render_stream && h->render_mode == 0 && h->render_sched && !l.legacy && !l.indirect && !trace_path)
	{
		size_t o0, o1, o2;
		const size_t need = stream_scratch_bytes(width, height, spp, &o0, &o1, &o2);
		void *&scratch = L ? L->scratch : h->r_stream;
		size_t &scratch_bytes = L ? L->scratch_bytes : h->r_stream_bytes;
		if (scratch_bytes < need)
		{
			HIP_TRY(hipStreamSynchronize(rs));
			if (scratch) HIP_TRY(hipFree(scratch));
			scratch = nullptr;
			scratch_bytes = 0;
			HIP_TRY(hipMalloc(&scratch, need));
			scratch_bytes = need;
		}
		l.stream_scratch = scratch;
		l.stream_check = h->render_stream_check != 0;
		// the check below reads the passes' control words after the frame: zero them here, for a frame whose stream passes do not run
		// (volume off screen or outside the band) would otherwise report an earlier frame's counts — or, on fresh scratch, noise
		if (l.stream_check) HIP_TRY(hipMemsetAsync(scratch, 0, 4096, rs));
	}
	if (h->render_frame_bricks && frame_bricks_applies(h->G) && (h->render_mode == 1 || l.stream_scratch))
	{
		const size_t need = frame_bricks_bytes(h->G);
		if (h->r_bricks_bytes != need)
		{
			HIP_TRY(hipStreamSynchronize(h->stream));
			for (auto &fl : h->lanes)
				if (fl.s) HIP_TRY(hipStreamSynchronize(fl.s));
			if (h->r_bricks) HIP_TRY(hipFree(h->r_bricks));
			h->r_bricks = nullptr;
			h->r_bricks_bytes = 0;
			HIP_TRY(hipMalloc((void **)&h->r_bricks, need));
			h->r_bricks_bytes = need;
			h->r_bricks_key[0] = 0;
		}
		l.bricks = h->r_bricks;
		l.bricks_valid = !memcmp(state_key, h->r_bricks_key, sizeof state_key);
		l.bricks_built = &bricks_built;
	}
	if (L)
	{
		// the presentation surface is shared: this frame's pixels after those of the frame before it (which waited for the one before that)
		if (h->last_lane >= 0 && h->last_lane != h->lane_next && h->lanes[h->last_lane].used) l.after = h->lanes[h->last_lane].done;
		// the occupancy bits and the bricks are shared too: a frame that rebuilds them waits for the frames that may still be reading them
		const bool occ_rebuild = l.occ && !l.occ_valid, bricks_rebuild = l.bricks && !l.bricks_valid;
		if (occ_rebuild || bricks_rebuild)
			for (auto &fl : h->lanes)
				if (&fl != L && fl.used) HIP_TRY(hipStreamWaitEvent(rs, fl.done, 0));
	}
	HIP_TRY(hipEventRecord(L ? L->start : h->rev_start, rs));
	hipError_t e = launch_render(l, rs);
	if (e != hipSuccess)
	{
		h->r_occ_key[0] = h->r_bricks_key[0] = 0; // whatever was half built is not to be trusted
		return fail(CA3D_ERR_DEVICE, "render launch failed: %s", hipGetErrorString(e));
	}
	// the derived buffers this call rebuilt now describe this state; the ones it did not touch keep the key of the state they were built from
	if (occ_built) memcpy(h->r_occ_key, state_key, sizeof state_key);
	if (bricks_built) memcpy(h->r_bricks_key, state_key, sizeof state_key);
	HIP_TRY(hipEventRecord(L ? L->stop : h->rev_stop, rs));
	h->rev_valid = true;
	h->last_lane = L ? h->lane_next : -1;
	if (L)
	{
		HIP_TRY(hipEventRecord(L->done, rs));
		L->pending = L->used = true;
		h->lane_next = (h->lane_next + 1) % active_lanes;
	}
	if (trace_path)
	{
		std::vector<unsigned long long> t(counter_words);
		HIP_TRY(hipStreamSynchronize(h->stream));
		HIP_TRY(hipMemcpy(t.data(), h->r_counters, counter_words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
		if (FILE *f = fopen(trace_path, "wb")) // the last frame wins
		{
			fwrite(t.data(), sizeof(unsigned long long), counter_words, f);
			fclose(f);
		}
	}
	if (l.stream_scratch && l.stream_check)
	{
		// diagnostics: the stream passes counted where their interval filter and the reference's slab test disagreed (must be nowhere)
		uint32_t bad[4] = {0, 0, 0, 0};
		HIP_TRY(hipStreamSynchronize(h->stream));
		HIP_TRY(hipMemcpy(bad, static_cast<const uint32_t *>(l.stream_scratch) + 2, sizeof bad, hipMemcpyDeviceToHost));
		if (bad[0]) return fail(CA3D_ERR_DEVICE, "render_stream_check: the interval filter contradicted the slab test at %u live cells", bad[0]);
		if (bad[1]) return fail(CA3D_ERR_DEVICE, "render_stream_check: %u looked-up answers had not been given in this frame (jobs lost by the queues; jobs %u .. %u)", bad[1], ~bad[2], bad[3]);
	}
	h->rstats.primary_rays = (uint64_t)width * ((l.row1 ? l.row1 : height) - l.row0) * spp;
	if (presentation_rgba8) HIP_TRY(hipMemcpyAsync(presentation_rgba8, h->r_present, px * 4, hipMemcpyDeviceToHost, h->stream));
	if (light_rgba16f) HIP_TRY(hipMemcpyAsync(light_rgba16f, h->r_light[h->r_swap], px * 8, hipMemcpyDeviceToHost, h->stream));
	if (depth_rg16f) HIP_TRY(hipMemcpyAsync(depth_rg16f, h->r_depth[h->r_swap], px * 4, hipMemcpyDeviceToHost, h->stream));
	if (presentation_rgba8 || light_rgba16f || depth_rg16f) HIP_TRY(hipStreamSynchronize(h->stream));
	h->r_swap ^= 1;
	h->state_touched = false; // (set again by the next entry point that is not a frame: bind_device)
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_get_render_stats(ca3d_t *h, ca3d_render_stats *out) CA3D_API_TRY
{
	if (!h || !out) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL argument");
	if (!h->rev_valid) return fail(CA3D_ERR_NOT_CONFIGURED, "ca3d_render has not been called yet");
	int rc = bind_device(h);
	if (rc) return rc;
	// the last frame's events and counters: the engine's, or those of the lane it ran on (frames in flight)
	const ca3d_engine::FrameLane *L = h->last_lane >= 0 ? &h->lanes[h->last_lane] : nullptr;
	HIP_TRY(hipEventSynchronize(L ? L->stop : h->rev_stop));
	float ms = 0.f;
	HIP_TRY(hipEventElapsedTime(&ms, L ? L->start : h->rev_start, L ? L->stop : h->rev_stop));
	unsigned long long c[3] = {0, 0, 0};
	HIP_TRY(hipMemcpy(c, L ? L->counters : h->r_counters, sizeof c, hipMemcpyDeviceToHost));
	h->rstats.gpu_ms = ms;
	h->rstats.shadow_rays = c[0];
	h->rstats.primary_cell_visits = c[1];
	h->rstats.shadow_cell_visits = c[2];
	*out = h->rstats;
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_get_render_pipeline(ca3d_t *h, int32_t *frames_in_flight) CA3D_API_TRY
{
	if (!h || !frames_in_flight) return fail(CA3D_ERR_INVALID_ARGUMENT, "ca3d_get_render_pipeline: NULL argument");
	*frames_in_flight = h->render_pipeline && h->n_lanes >= 2 ? h->lanes_in_use : 0; // of the last pipelined frame (the default depth follows the frame's size)
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_set_option(ca3d_t *h, const char *name, int64_t value) CA3D_API_TRY
{
	if (!h || !name) return fail(CA3D_ERR_INVALID_ARGUMENT, "NULL argument");
	FLUSH_QUEUED(h); // options apply to the steps encoded after them
	if (strcmp(name, "queue") && strcmp(name, "stats"))
		if (int rcs = settle_resident(h)) return rcs; // ... and a recovery re-runs steps under the options they were issued with
	if (!strcmp(name, "resident_fault_tile"))
	{
		// diagnostics: tile `value - 1` of the NEXT resident launch leaves at once, as a workgroup that never became resident
		// would; its neighbours time out and the engine recovers (tests/test_gpu_ca_parity.py)
		if (value < 0 || value > 1024) return fail(CA3D_ERR_INVALID_ARGUMENT, "resident_fault_tile must be in [0, 1024]");
		h->res_fault_tile = (uint32_t)value;
		return CA3D_OK;
	}
	if (!strcmp(name, "queue"))
	{
		if (value < 0 || value > 1000000) return fail(CA3D_ERR_INVALID_ARGUMENT, "queue must be in [0, 1000000] steps");
		h->queue_max = (uint32_t)value;
		return CA3D_OK;
	}
	if (!strcmp(name, "graph")) { h->use_graph = value ? 1 : 0; return CA3D_OK; }
	if (!strcmp(name, "graph_prepare"))
	{
		// capture and instantiate, now, the graphs a later ca3d_step(value) replays (otherwise built on first use)
		int rc = check_ready(h);
		if (rc) return rc;
		rc = bind_device(h);
		if (rc) return rc;
		if (!graphs_allowed(h) || h->slab || value <= 0) return CA3D_OK;
		// the batches ca3d_step(value) will replay, from either buffer (an odd batch length alternates)
		uint64_t left = (uint64_t)value;
		uint32_t cur = h->cur;
		for (int pass = 0; pass < 2; pass++)
		{
			for (uint64_t l = left; l;)
			{
				const uint32_t n = l > kMaxGraphSteps ? kMaxGraphSteps : (uint32_t)l;
				if (n >= h->graph_min)
				{
					ca3d_engine::StepGraph *g = nullptr;
					rc = step_graph(h, n, cur, &g);
					if (rc) return rc;
				}
				cur = (cur + n) & 1u;
				l -= n;
			}
			if (cur == h->cur) break; // even total: the next call starts from the same buffer
		}
		return CA3D_OK;
	}
	if (!strcmp(name, "stats")) { h->want_stats = value ? 1 : 0; if (!value) h->ev_valid = false; return CA3D_OK; }
	if (!strcmp(name, "graph_min"))
	{
		if (value < 1 || value > kMaxGraphSteps) return fail(CA3D_ERR_INVALID_ARGUMENT, "graph_min must be in [1, %u]", kMaxGraphSteps);
		h->graph_min = (uint32_t)value;
		return CA3D_OK;
	}
	if (!strcmp(name, "render_indirect")) { h->render_indirect = value ? 1 : 0; return CA3D_OK; }
	if (!strcmp(name, "render_sched")) { h->render_sched = value ? 1 : 0; return CA3D_OK; }
	if (!strcmp(name, "render_skip")) { h->render_skip = value ? 1 : 0; return CA3D_OK; }
	if (!strcmp(name, "rows"))
	{
		h->use_rows = value ? 1 : 0;
		refresh_kernels(h);
		note_jit_failure(h);
		return CA3D_OK;
	}
	if (!strcmp(name, "render_frame_bricks")) { h->render_frame_bricks = value ? 1 : 0; return CA3D_OK; }
	if (!strcmp(name, "render_stream")) { h->render_stream = value ? 1 : 0; return CA3D_OK; }
	if (!strcmp(name, "render_pipeline")) // converged frames in flight (FrameLane): 0 off, 1 the default depth, 2 .. kMaxLanes that many
	{
		if (value < 0 || value > ca3d_engine::kMaxLanes) return fail(CA3D_ERR_INVALID_ARGUMENT, "render_pipeline must be 0 (off), 1 (default depth) or 2 .. %d frames in flight", ca3d_engine::kMaxLanes);
		h->render_pipeline = (int)value;
		return CA3D_OK;
	}
	if (!strcmp(name, "render_stream_check")) { h->render_stream_check = value ? 1 : 0; return CA3D_OK; }
	if (!strcmp(name, "render_row_begin") || !strcmp(name, "render_row_end"))
	{
		if (value < 0 || value > 16384) return fail(CA3D_ERR_INVALID_ARGUMENT, "row %lld is outside any target", (long long)value);
		if (name[11] == 'b')
		{
			if (value % 16) return fail(CA3D_ERR_INVALID_ARGUMENT, "render_row_begin must be a multiple of 16 (the renderer's tile height)");
			h->render_row0 = (uint32_t)value;
		}
		else h->render_row1 = (uint32_t)value;
		return CA3D_OK;
	}
	if (!strcmp(name, "render_mode"))
	{
		if (value != 0 && value != 1) return fail(CA3D_ERR_INVALID_ARGUMENT, "render_mode must be 0 (converged frame) or 1 (one literal reference frame)");
		h->render_mode = (int)value;
		return CA3D_OK;
	}
	if (!strcmp(name, "render_reset_history"))
	{
		// forget the temporal history (a fresh canvas): the next literal frame sees zeros, as on the reference's first frame
		int rc2 = bind_device(h);
		if (rc2) return rc2;
		const size_t px = (size_t)h->rw * h->rh;
		for (int i = 0; i < 2 && px; i++)
		{
			HIP_TRY(hipMemsetAsync(h->r_light[i], 0, px * 8, h->stream));
			HIP_TRY(hipMemsetAsync(h->r_depth[i], 0, px * 4, h->stream));
		}
		return CA3D_OK;
	}
	if (!strcmp(name, "fused")) { drop_graph(h); h->use_fused = value ? 1 : 0; return CA3D_OK; }
	if (!strcmp(name, "variant"))
	{
		if (value != 0 && value != 1) return fail(CA3D_ERR_INVALID_ARGUMENT, "variant must be 0 (auto) or 1 (generic kernel)");
		drop_graph(h);
		h->variant = (int)value;
		refresh_kernels(h);
		note_jit_failure(h);
		return CA3D_OK;
	}
	if (!strcmp(name, "resident"))
	{
		h->use_resident = value ? 1 : 0;
		if (value) h->res_failed = false;
		refresh_kernels(h);
		note_jit_failure(h);
		return CA3D_OK;
	}
	if (!strcmp(name, "resident_rows"))
	{
		if (value != 16 && value != 32) return fail(CA3D_ERR_INVALID_ARGUMENT, "resident_rows must be 16 or 32");
		if ((uint32_t)value != h->res_rows)
		{
			// the mailboxes are indexed by tile: start from clean ones
			int rc2 = bind_device(h);
			if (rc2) return rc2;
			HIP_TRY(hipStreamSynchronize(h->stream));
			free_resident(h);
			h->res_rows = (uint32_t)value;
			refresh_kernels(h);
			note_jit_failure(h);
		}
		return CA3D_OK;
	}
	if (!strcmp(name, "resident_pair"))
	{
		if (value != 0 && value != 1) return fail(CA3D_ERR_INVALID_ARGUMENT, "resident_pair must be 0 or 1");
		if ((value != 0) != h->res_pair)
		{
			int rc2 = bind_device(h);
			if (rc2) return rc2;
			HIP_TRY(hipStreamSynchronize(h->stream));
			free_resident(h); // the tiling may change with it (32-row tiles): start from clean mailboxes
			h->res_pair = value != 0;
			refresh_kernels(h);
			note_jit_failure(h);
		}
		return CA3D_OK;
	}
	if (!strcmp(name, "resident_zsplit"))
	{
		if (value != 1 && value != 2) return fail(CA3D_ERR_INVALID_ARGUMENT, "resident_zsplit must be 1 or 2");
		h->res_zsplit = (uint32_t)value;
		refresh_kernels(h);
		note_jit_failure(h);
		return CA3D_OK;
	}
	if (!strcmp(name, "resident_min"))
	{
		if (value < 1) return fail(CA3D_ERR_INVALID_ARGUMENT, "resident_min must be >= 1");
		h->res_min = (uint32_t)value;
		return CA3D_OK;
	}
	if (!strcmp(name, "resident_timeout_us"))
	{
		if (value < 1 || value > 40000000) return fail(CA3D_ERR_INVALID_ARGUMENT, "resident_timeout_us must be in [1, 40000000]");
		h->res_timeout_ticks = (uint32_t)(value * 100);
		return CA3D_OK;
	}
	if (!strcmp(name, "roll_z"))
	{
		if (value != 0 && value != 2 && value != 4 && value != 8 && value != 16 && value != 15 && value != 30) return fail(CA3D_ERR_INVALID_ARGUMENT, "roll_z must be 0 (automatic), 2, 4, 8, 16 (tile form only), or 15 / 30 (looped forms)");
		drop_graph(h);
		h->roll_z = (int)value;
		return CA3D_OK;
	}
	if (!strcmp(name, "roll_tile"))
	{
		drop_graph(h);
		if (value < 0 || value > 3) return fail(CA3D_ERR_INVALID_ARGUMENT, "roll_tile must be 0 (every thread shifts its rows), 1 (256-thread tiles), 2 (wave tiles) or 3 (two words per thread)");
		h->roll_tile = (int)value;
		return CA3D_OK;
	}
	if (!strcmp(name, "roll"))
	{
		drop_graph(h);
		h->use_roll = value ? 1 : 0;
		refresh_kernels(h);
		note_jit_failure(h);
		return CA3D_OK;
	}
	if (!strcmp(name, "jit"))
	{
		drop_graph(h);
		h->use_jit = value ? 1 : 0;
		refresh_kernels(h);
		note_jit_failure(h);
		return CA3D_OK;
	}
	return fail(CA3D_ERR_INVALID_ARGUMENT, "unknown option '%s'", name);
}
CA3D_API_CATCH

} // extern "C"

// ------------------------------------------------------------------------------------------------ internals for ca3d_group.cpp
namespace ca3d
{
int engine_device(const ca3d_engine *h) { return h->device; }
hipStream_t engine_stream(const ca3d_engine *h) { return h->stream; }

// A full-grid engine that only ever RECEIVES its state on the device (the group's frame: peer copies of the slabs): both buffers
// cleared on the engine's stream, marked as holding a state — no host copy of the grid, no synchronous upload.
int engine_mark_state(ca3d_engine *h)
{
	if (!h || !h->configured || h->slab) return fail(CA3D_ERR_NOT_CONFIGURED, "engine_mark_state: a configured full-grid engine is needed");
	int rc = bind_device(h);
	if (rc) return rc;
	HIP_TRY(hipMemsetAsync(h->buf[0], 0, h->buffer_words() * sizeof(uint32_t), h->stream));
	HIP_TRY(hipMemsetAsync(h->buf[1], 0, h->buffer_words() * sizeof(uint32_t), h->stream));
	h->step = 0;
	h->cur = 0;
	h->has_state = true;
	h->binary_state = true;
	h->state_serial++;
	h->buffers_exposed = false;
	return CA3D_OK;
}
int engine_state_buffer(ca3d_engine *h, int which, void **device_ptr, size_t *n_bytes)
{
	const bool was = h ? h->buffers_exposed : false;
	const int rc = ca3d_device_buffer(h, which, device_ptr, n_bytes);
	if (h) h->buffers_exposed = was;
	return rc;
}
void engine_set_ghosts_valid(ca3d_engine *h, bool valid) { h->ghosts_valid = valid; }
bool engine_ghosts_valid(const ca3d_engine *h) { return h->ghosts_valid; }

// One communicator per slab engine, all created by THIS process (ncclCommInitAll: one host thread, n devices) — the
// single-process form of ca3d_slab_comm_init. RCCL refuses two ranks on one device.
int engines_rccl_init_all(ca3d_engine **engines, int n)
{
	Rccl &r = rccl();
	if (!r.error.empty()) return fail(CA3D_ERR_UNSUPPORTED, "%s", r.error.c_str());
	typedef int (*InitAll)(void **, int, const int *);
	InitAll init_all = (InitAll)dlsym(r.lib, "ncclCommInitAll");
	if (!init_all) return fail(CA3D_ERR_UNSUPPORTED, "librccl lacks ncclCommInitAll");
	std::vector<void *> comms((size_t)n, nullptr);
	std::vector<int> devs((size_t)n);
	for (int k = 0; k < n; k++) devs[(size_t)k] = engines[k]->device;
	NCCL_TRY(init_all(comms.data(), n, devs.data()));
	for (int k = 0; k < n; k++)
	{
		ca3d_engine *h = engines[k];
		if (h->comm) r.CommDestroy(h->comm);
		h->comm = comms[(size_t)k];
		h->comm_rank = k;
		h->comm_world = n;
		h->ghosts_valid = false;
	}
	return CA3D_OK;
}

// The ghost refresh of every rank as ONE RCCL group (a single thread cannot post rank 0's sends and wait for them before
// rank 1's receives exist): ncclGroupStart, every rank's sends and receives on its own stream, ncclGroupEnd.
int engines_rccl_exchange_all(ca3d_engine **engines, int n)
{
	Rccl &r = rccl();
	NCCL_TRY(r.GroupStart());
	for (int k = 0; k < n; k++)
	{
		int rc = bind_device(engines[k]);
		if (rc == CA3D_OK) rc = comm_exchange(engines[k], engines[k]->stream);
		if (rc) { r.GroupEnd(); return rc; }
	}
	NCCL_TRY(r.GroupEnd());
	for (int k = 0; k < n; k++) engines[k]->ghosts_valid = true;
	return CA3D_OK;
}
} // namespace ca3d
