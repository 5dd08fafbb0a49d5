// Internal declarations shared by the C-ABI layer and the HIP kernels. Not installed; the public surface is
// include/ca3d.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "ca3d.h"
#include "ca_device_types.h"

namespace ca3d
{

constexpr int kMaxOffsets = 26; // per list; a count must stay inside its 27-slot LUT slice

struct OffsetLists
{
	uint32_t n[3];
	uint8_t code[3][32]; // (dx+1) | (dy+1) << 2 | (dz+1) << 4
};

struct CanonRules
{
	bool valid = false;
	// packed layout
	MainKind main = MAIN_GENERIC;
	bool need[3] = {false, false, false}; // whether the rule-set's count influences the result at all
	bool fast = false;                    // the three lists are the named class unions -> class kernels
	OffsetLists lists{};
	PackedRuleArgs prog{};
	uint32_t onset_survive[3]{}, onset_born[3]{}; // bit k: LUT slot k of the rule-set == 1
	// unpacked layout: main list + slots 0..26 with `> 0`
	uint32_t unpacked_survive = 0, unpacked_born = 0; // bit k = LUT[k] > 0
	RuleSetProg unpacked_prog{};                        // the same as cube programs over the main list's count
	bool unpacked_fast = false;                         // main list is a named class union (ballot kernel applies)
	// raw copies (returned for diagnostics)
	uint32_t survive_raw[CA3D_LUT_LEN]{}, born_raw[CA3D_LUT_LEN]{};
};

// rules.cpp
int canonicalize_rules(const int32_t *main_offs, uint32_t n_main, const int32_t *edge_offs, uint32_t n_edge,
                       const int32_t *corner_offs, uint32_t n_corner, const uint32_t *survive,
                       const uint32_t *born, CanonRules *out, std::string *err);
// Minimal OR-of-cubes cover of `onset` over `nvars` count planes; counts > max_count are don't-cares.
void compile_rule_prog(uint32_t onset_mask, uint32_t max_count, RuleProg *out);

// ---------------------------------------------------------------------------------------------- launch params

// Run-time compiled (hiprtc) von Neumann kernels for one (grid, survive table, born table): ca_jit.cpp
struct VnJit
{
	void *zr1 = nullptr, *zr2 = nullptr, *zr4 = nullptr, *zr8 = nullptr; // hipFunction_t of the 1-, 2-, 4- and (grids of 2048 and up) 8-planes-per-thread entry points
	int cvl = -1;
	uint32_t lut_s = 0, lut_b = 0;
};

// Run-time compiled rows kernel (ca_packed_rows_kernel.inc) for one (grid, main table, live rule-sets, rule): any multiple of 32
struct RowsJit
{
	void *deep = nullptr, *flat = nullptr; // zrun planes per thread / 1 plane per thread
	uint32_t G = 0;
	int zrun = 0, main = -1;
	bool e = false, c = false;
	uint32_t tables[6]{};
};

// Run-time compiled class kernels (rule as compile-time truth tables) for one (main table, live rule-sets, rule)
struct ClassJit
{
	void *deep = nullptr, *deep_za = nullptr, *flat = nullptr; // hipFunction_t: ZRUN planes per thread (general /
	                                                           // z-aligned ranges) / 1 plane per thread
	void *deep_np2 = nullptr, *flat_np2 = nullptr;             // the same for rows of whole uint4 that are not a power of two of them
	int main = -1;
	bool e = false, c = false;
	uint32_t tables[6] = {0, 0, 0, 0, 0, 0}; // survive / born of main, edges, corners (bit k = value at count k)
};

// Run-time compiled rolling-window class kernels (ca_packed_roll_kernel.inc) for one (grid, main table, live rule-sets,
// rule): entry points with 2, 4 and 8 output planes per thread
struct RollJit
{
	void *z[3] = {nullptr, nullptr, nullptr}; // hipFunction_t for Z = 2, 4, 8
	void *tile[4] = {nullptr, nullptr, nullptr, nullptr}; // the tile form (x-shifted rows shared through LDS) for Z = 2, 4, 8, 16
	void *tile_x = nullptr; int zx = 0; // tuning (CA3D_ROLL_ZX=<planes>): one more depth of the tile form
	void *loop[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; // looped forms [tile][Z = 15, 30]: the plane loop rolled up in groups of three
	void *w2[2] = {nullptr, nullptr};    // two words per thread (five waves per SIMD) for Z = 8, 16; null on rows of 32 uint4 and more
	void *wtile[2] = {nullptr, nullptr}; // wave tiles (one wave per workgroup, LDS exchange without a barrier) for Z = 8, 16; null on rows of 64 uint4
	void *np2[3] = {nullptr, nullptr, nullptr}; // roll_step_np2 for Z = 2, 4, 8: the module of a grid with cv_np2 uint4 per row (not a power of two)
	int cv_np2 = 0;                            // uint4 per row of the np2 module; 0: none
	int cvl = -1;                              // log2(G / 128); -1: none
	int main = -1;
	bool e = false, c = false;
	uint32_t tables[6] = {0, 0, 0, 0, 0, 0};
};

struct PackedLaunch
{
	const uint32_t *in;
	uint32_t *out;
	PlaneRange pr;
	const CanonRules *rules;
	int variant; // -1 auto
	const VnJit *vn_jit = nullptr; // specialised kernels for exactly these rules and this grid, or null
	const ClassJit *class_jit = nullptr;
	const RollJit *roll_jit = nullptr;
	int roll_z = 0; // 0: the launcher picks the planes per thread of the rolling-window kernel; 2 / 4 / 8 (tile form: 16 too): forced (tests, tuning)
	int roll_tile = 1; // 1: the tile form of the rolling-window kernel (ca_packed_roll_kernel.inc, tile_step); 0: every thread shifts its three rows itself;
	                   // 2: wave tiles (the tile form with one wave per workgroup: no barrier); 3: two words per thread, no LDS (roll_step_w2)
	const RowsJit *rows_jit = nullptr; // the rows kernel compiled for exactly this grid and these rules, or null
};

// One launch of the resident multi-step kernel (ca_resident.hip): `steps` steps from `in`, state on chip in between
struct ResidentLaunch
{
	const uint32_t *in;
	uint32_t *out_last, *out_prev; // state after `steps` steps / one step earlier (the two ping-pong buffers)
	uint32_t G;
	unsigned long long *mail;      // face mailboxes, resident_mail_bytes(G)
	uint32_t *status;              // device words: [0] != 0 when a wait timed out, [4 + tile] steps finished by that tile
	uint32_t *host_flag;           // pinned host word the kernel sets on a timeout
	uint32_t steps, epoch0, timeout_ticks;
	uint32_t lut_s, lut_b;         // von Neumann truth tables (vn_tables)
	void *jit_fn = nullptr;        // hipFunction_t of the run-time compiled kernel for these tables, or null (pre-built rule)
	uint32_t rows = 32;            // rows per tile: 32 (one 512-thread workgroup per CU) or 16 (two 256-thread workgroups per CU)
	uint32_t fault_tile = 0;       // diagnostics: tile + 1 that leaves at once (a workgroup that never became resident), 0: none
	uint32_t zsplit = 1;           // thread groups along z: 2 = twice the threads per tile, four waves per SIMD (ca_resident_kernel.inc, ZS)
	int pair = 0;                  // 512^3 von Neumann form: a thread owns two adjacent rows x 16 planes; rows = 32, zsplit = 1 (resident_pair_run)
};

struct UnpackedLaunch
{
	const uint32_t *in;
	uint32_t *out;
	PlaneRange pr;
	const CanonRules *rules;
	bool binary_state; // every cell is 0 or 1 (true after any step; checked on upload)
};

struct RenderLaunch
{
	const uint32_t *cells; // packed state, current buffer
	uint32_t G, W, H, spp;
	const float *uniforms; // host pointer, >= 84 floats
	uint32_t *presentation; // device RGBA8 or null
	void *light;            // device RGBA16F or null
	uint32_t *depth;        // device RG16F or null
	unsigned long long *counters; // device, 3 counters, or null
	bool legacy;                  // unpacked volume + legacy shading (shaders/pathtraced_fragment.wgsl)
	int mode;                     // 0 converged frame (exact walk), 1 one literal reference frame with history
	const void *prev_light;       // previous frame's RGBA16F (mode 1)
	const uint32_t *prev_depth;   // previous frame's RG16F (mode 1)
	int sched = 1;                // mode 0: dynamic ray scheduling inside each wave (same frame, bit for bit)
	unsigned long long *occ = nullptr; // mode 0, packed: scratch for the block-occupancy bits + their count,
	                                   // sized by ca3d_render (null: no empty-space skipping)
	uint32_t row0 = 0, row1 = 0;  // mode 0: render image rows [row0, row1) only (row1 == 0: all); row0 % 16 == 0
	bool indirect = false;        // mode 0, packed: add the one-bounce neighbour lighting term (wgsl :307-377)
	bool trace = false;           // diagnostics: per-wave {start, end, HW_ID, visits} after the counters (the buffer must hold them)
	// mode 0, scheduled kernel: a second stream (+ two events) for the plain kernel that renders the tiles AROUND the volume's screen
	// rectangle — it then runs beside the persistent scheduled launch (whose tail leaves CUs idle) instead of after it. Null: one stream.
	hipStream_t aux = nullptr;
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	// mode 0, packed, scheduled: scratch of the ray-stream pipeline (render_stream.hip; stream_scratch_bytes(W, H, spp) bytes) — with it the
	// dense-volume part of the frame is drawn by the stream passes instead of ca_render_packed_sched (same frame, bit for bit). Null: not used.
	uint32_t *bricks = nullptr; // room for the bricked copy of the volume (frame_bricks_bytes(G)); with it (any grid that is a multiple of 32 up to 2048) mode 1's frame is
	                            // drawn by render_frame.hip's batched march instead of ca_render_frame_packed, and mode 0's stream walks read the bricks
	                            // (same frames, bit for bit)
	// The occupancy bits / the bricks in those buffers were built from exactly this state by an earlier frame (the engine tracks what the
	// state is: step count, uploads, buffers handed out): the passes that rebuild them are skipped. A static scene — a camera moving
	// around a paused automaton — pays for them once (512^3: 30 us of a 0.87 ms frame; 2048^3: 2 ms of 3.7).
	bool occ_valid = false, bricks_valid = false;
	// ... and told back: set by the passes that (re)build them in THIS call — the engine takes a buffer for current only then (a frame in
	// the literal mode builds no occupancy bits, a frame whose volume is off screen no bricks: taking either for built would leave the
	// next frame of the same state with stale ones — a converged frame right after literal ones came out black that way)
	bool *occ_built = nullptr, *bricks_built = nullptr;
	void *stream_scratch = nullptr;
	// two frames in flight (ca3d_api.cpp, FrameLane): the frame before this one runs on another stream; everything of THIS frame that
	// writes a target both share (the presentation surface) waits for this event first. Null: nothing to wait for.
	hipEvent_t after = nullptr;
	int walk_share_pct = 100; // frames in flight: the share of the chip's wave slots each persistent walk launch of this frame asks for (100: the frame has the chip to itself)
	bool stream_check = false; // diagnostics: every live-cell decision of the interval filter is checked against the slab test and contradictions counted
};

// ca_diag.hip: float4 device-to-device copy (measurement only)
hipError_t launch_copy_f4(const void *in, void *out, size_t bytes, hipStream_t stream);
// ca_diag.hip: do two (idle) streams run side by side, i.e. sit on different hardware queues? (probe: ~2 ms)
hipError_t streams_concurrent(hipStream_t a, hipStream_t b, bool *out);
// render.hip
hipError_t launch_render(const RenderLaunch &l, hipStream_t stream);
// render_stream.hip: bytes of scratch a frame of this size needs at most (and where its three arrays start); the passes themselves
// (`params`: render.hip's launch parameters with the volume's screen rectangle filled in)
size_t stream_scratch_bytes(uint32_t W, uint32_t H, uint32_t spp, size_t *hit_off, size_t *occl_off, size_t *rays_off);
// render_frame.hip: the literal frame over a bricked copy of the volume (`frame_params`: render.hip's FrameParams)
size_t frame_bricks_bytes(uint32_t G);
bool frame_bricks_applies(uint32_t G);
hipError_t launch_render_frame_bricks(const void *frame_params, uint32_t *bricks, bool bricks_valid, hipStream_t stream, bool *bricks_built = nullptr);
hipError_t launch_brick_volume(const uint32_t *cells, uint32_t *bricks, uint32_t G, hipStream_t stream);
hipError_t launch_render_stream(const void *params, void *scratch, uint32_t W, uint32_t H, bool check, uint32_t *bricks, bool bricks_valid, hipStream_t stream, bool *bricks_built = nullptr, hipEvent_t before_resolve = nullptr, int walk_share_pct = 100,
                                bool *sparse_too = nullptr); // *sparse_too: the passes launched draw scattered sparse volumes as well (render.hip then leaves its scheduled kernel out)

// ca_packed.hip / ca_unpacked.hip
hipError_t launch_packed_step(const PackedLaunch &l, hipStream_t stream, const char **kernel_name);
const char *packed_kernel_name(const CanonRules &r, uint32_t G, int variant);
// ca_packed_vn.hip: the specialised von Neumann kernel (truth-table rules, power-of-two grids)
bool vn_rule_applies(const CanonRules &r, int variant); // main list von Neumann, edges / corners rule-sets without effect
bool vn_kernel_applies(const CanonRules &r, uint32_t G, int variant);
hipError_t launch_packed_vn(const PackedLaunch &l, hipStream_t stream);
// Canonical truth tables of the von Neumann kernel for these rules (entry 7 is a don't-care: see ca_packed_vn.hip)
void vn_tables(const CanonRules &r, uint32_t *lut_s, uint32_t *lut_b);
// Whether ca_packed_vn.hip carries an ahead-of-time specialisation for these tables (then no JIT is needed)
bool vn_tables_prebuilt(uint32_t lut_s, uint32_t lut_b);
int vn_grid_log2(uint32_t G); // log2(G / 128)
// ca_jit.cpp: compile (or fetch from the cache) the kernels specialised for (grid 128 << cvl, tables)
uint64_t jit_sources_hash(); // of every device source the run-time compiler is given
int jit_vn_kernels(int device, int cvl, uint32_t lut_s, uint32_t lut_b, VnJit *out, std::string *log);
// Truth tables of the three rule-sets over their count planes, unreachable counts filled (class kernels, JIT)
void class_tables(const CanonRules &r, uint32_t tables[6]);
int class_zrun(const CanonRules &r); // planes per thread of the deep variant for these rules
bool use_class_kernel(const CanonRules &r, uint32_t G, int variant);
int jit_class_kernels(int device, const CanonRules &r, ClassJit *out, std::string *log);
int jit_rows_kernels(int device, const CanonRules &r, uint32_t G, RowsJit *out, std::string *log);
// Whether the rows kernel is the one to use for these rules on this grid: named classes, and no uint4 kernel with the rule compiled in
bool rows_kernel_applies(const CanonRules &r, uint32_t G, int variant);
int jit_roll_kernels(int device, const CanonRules &r, int cvl, RollJit *out, std::string *log);
// ... for a grid of `cv` uint4 per row, cv not a power of two (384, 640, 768, 896): out->np2, out->cv_np2
int jit_roll_np2_kernels(int device, const CanonRules &r, int cv, RollJit *out, std::string *log);
bool roll_np2_applies(const CanonRules &r, uint32_t G, int variant);
// Whether the rolling-window kernel is the one to use for these rules on this grid (diagonal neighbour classes in
// play, power-of-two grid of 256 and up)
bool roll_kernel_applies(const CanonRules &r, uint32_t G, int variant);
// rule_synth.cpp: device source of jit_rule_word() for these rules ("" when the self-check fails), ca_jit.cpp's wrapper
std::string synthesize_rule_source(const CanonRules &r, int *ops_out);
std::string rule_function_source(const CanonRules &r);
// ca_resident.hip
bool resident_kernel_applies(const CanonRules &r, uint32_t G, int variant);
size_t resident_mail_bytes(uint32_t G, uint32_t rows);
hipError_t launch_resident(const ResidentLaunch &l, hipStream_t stream);
// Workgroups a resident launch of this shape needs (`tiles`) and how many of them the device can hold at once on `stream`
// (occupancy of the kernel per CU x the CUs the stream may use): the launch only completes when capacity >= tiles.
// Returns false when the runtime cannot tell (the caller then relies on the kernel's bounded waits alone).
bool resident_capacity(uint32_t G, uint32_t rows, uint32_t zsplit, int pair, void *jit_fn, hipStream_t stream, uint32_t *tiles, uint32_t *capacity);
bool resident_slab_capacity(void *fn, hipStream_t stream, uint32_t *tiles, uint32_t *capacity);
int jit_resident_kernel(int device, uint32_t lut_s, uint32_t lut_b, uint32_t rows, uint32_t zsplit, int pair, void **fn, std::string *log);
// the resident kernel for rules with diagonal neighbour classes (ca_resident_class_kernel.inc; 512^3 and 256^3, run-time compiled)
bool resident_class_applies(const CanonRules &r, uint32_t G, int variant);
int jit_resident_class_kernel(int device, const CanonRules &r, uint32_t G, uint32_t zsplit, void **fn, std::string *log); // G: 512 (zsplit 1) or 256 (2 | 1)
uint32_t resident_class_zsplit(uint32_t G); // z groups of the class form's tiles
// Slab form (one rank's share of a 1024^3 grid, K sub-steps per launch): run-time compiled only, per planes-per-tile count
struct ResidentSlabLaunch
{
	const uint32_t *in;
	uint32_t *out;
	unsigned long long *mail; // resident_slab_mail_bytes()
	uint32_t *status, *host_flag;
	uint32_t steps, epoch0, timeout_ticks;
	int dead_plane; // array plane whose global z is 0, or -1
	void *fn;       // hipFunction_t from jit_resident_slab_kernel
};
// planes per tile of the slab form for an array of `nplanes` planes of a G-wide grid with these rules, or 0 when it does not apply
int resident_slab_planes(const CanonRules &r, uint32_t G, uint32_t nplanes, int variant);
size_t resident_slab_mail_bytes();
hipError_t launch_resident_slab(const ResidentSlabLaunch &l, hipStream_t stream);
void resident_stream_retired(hipStream_t stream); // call after waiting for a stream the engine stops using
int jit_resident_slab_kernel(int device, uint32_t lut_s, uint32_t lut_b, int pz, void **fn, std::string *log);
// Steps one fused launch advances for these rules / grid (0 = no fused kernel applies).
int packed_fused_steps(const CanonRules &r, uint32_t G, int variant);
hipError_t launch_packed_fused(const PackedLaunch &l, hipStream_t stream, const char **kernel_name);
hipError_t launch_unpacked_step(const UnpackedLaunch &l, hipStream_t stream, const char **kernel_name);


// ca3d_api.cpp internals used by ca3d_group.cpp
int set_error(int code, const char *fmt, ...);
// No C++ exception leaves an extern "C" entry point: each one is `int ca3d_x(...) CA3D_API_TRY { ... } CA3D_API_CATCH` — a
// function-try-block whose handler maps what was thrown to a status code and a ca3d_last_error() message (ca3d_api.cpp;
// tests/test_capi_cpu.py checks that every function include/ca3d.h declares is defined this way, and runs the handler on the CPU
// through ca3d_selftest_exception)
int exception_to_status() noexcept;
void jit_stats(ca3d_jit_stats *out); // ca_jit.cpp
#define CA3D_API_TRY try
#define CA3D_API_CATCH                          \
	catch (...)                                 \
	{                                           \
		return ca3d::exception_to_status();     \
	}
int engine_device(const ca3d_engine *h);
hipStream_t engine_stream(const ca3d_engine *h);
int engine_mark_state(ca3d_engine *h); // both buffers zeroed on the engine's stream, "has a state" — for engines whose state arrives by device copies
// ca3d_device_buffer for a writer inside the library (the group's peer copies): the state counts as rewritten by THIS call only — the
// public call must assume writes at any later time and makes every frame rebuild its derived buffers while the pointer is valid
int engine_state_buffer(ca3d_engine *h, int which, void **device_ptr, size_t *n_bytes);
void engine_set_ghosts_valid(ca3d_engine *h, bool valid);
bool engine_ghosts_valid(const ca3d_engine *h);
int engines_rccl_init_all(ca3d_engine **engines, int n);
int engines_rccl_exchange_all(ca3d_engine **engines, int n);

} // namespace ca3d
