/*
 * ca3d.h — C ABI of the MI355X-native engine for the two hot paths of lightest/cellularautomatons3d:
 * the 3D cellular-automaton step and (see ca3d_render*) the per-pixel volume renderer.
 *
 * This is the drop-in boundary: the reference drives its kernels through the browser's WebGPU API from
 * main_pathtraced.js; each entry point below names the reference call sites it replaces. Plain C, opaque handle,
 * plain pointers and sizes. Every function returns 0 (CA3D_OK) or a negative ca3d_status; ca3d_last_error()
 * gives the message for the calling thread. Caller-owned host buffers are fully consumed before a call
 * returns. Calls on one handle are not thread-safe (the reference host is a single JS thread). One engine
 * owns one HIP stream on one device; work is enqueued asynchronously exactly like the reference's
 * queue.submit — only ca3d_read_state / ca3d_synchronize / the stats getters wait for the GPU.
 *
 * There is no CPU fallback: without a usable HIP device ca3d_create fails with CA3D_ERR_DEVICE.
 */
#ifndef CA3D_H
#define CA3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CA3D_ABI_VERSION 7 /* 5: + ca3d_get_kernel_variant; 6: + ca3d_selftest_exception, the kernel cache (ca3d_get_jit_log reports it); 7: + ca3d_get_render_pipeline */
#define CA3D_LUT_LEN 81 /* 3 rule-sets x 27 slots (main_pathtraced.js:10, 155-159) */

typedef struct ca3d_engine ca3d_t;

enum ca3d_status
{
	CA3D_OK = 0,
	CA3D_ERR_INVALID_ARGUMENT = -1,
	CA3D_ERR_NOT_CONFIGURED = -2, /* call order: configure -> set_rules -> upload_state -> step */
	CA3D_ERR_DEVICE = -3,         /* HIP error, or no GPU */
	CA3D_ERR_OUT_OF_MEMORY = -4,
	CA3D_ERR_UNSUPPORTED = -5
};

enum ca3d_layout
{
	CA3D_LAYOUT_PACKED32 = 0, /* 32 x-adjacent cells per u32: shaders/compute_clustered.wgsl (the live kernel) */
	CA3D_LAYOUT_UNPACKED = 1  /* one u32 (0/1) per cell, toroidal: shaders/compute.wgsl (legacy kernel) */
};

int ca3d_abi_version(void);
const char *ca3d_last_error(void);
/* No C++ exception crosses this boundary: every entry point catches what the library's internals throw (host allocation
 * failures above all) and returns CA3D_ERR_OUT_OF_MEMORY for std::bad_alloc, CA3D_ERR_DEVICE for anything else, with the
 * message in ca3d_last_error(). Test hook for exactly that path, usable without a GPU: throws inside a guarded body —
 * kind 0 std::bad_alloc, 1 std::runtime_error, 2 a non-std exception, 3 a real oversized host allocation — and returns
 * the status the boundary mapped it to (any other kind: CA3D_OK). */
int ca3d_selftest_exception(int kind);
int ca3d_device_count(int *out_count);

/* navigator.gpu.requestAdapter / requestDevice (main_pathtraced.js:222-225). */
int ca3d_create(int device, ca3d_t **out);
int ca3d_destroy(ca3d_t *h);

/*
 * Grid uniform + state buffers (main_pathtraced.js:1204-1217, 635, 1241, 1314-1326). The reference always
 * writes a cubic [G,G,G]; gx == gy == gz is required. PACKED32: G a positive multiple of 32
 * (_gridSizeUIFormatter, 675-693). UNPACKED: G a positive multiple of 4 (workgroup 4x4x4, compute.wgsl:50).
 * Drops any previous state and resets the step counter (as _restartSim, 624-637).
 */
int ca3d_configure(ca3d_t *h, uint32_t gx, uint32_t gy, uint32_t gz, int layout);

/*
 * Z-slab of a G^3 grid for multi-GPU runs (no reference counterpart): this engine owns global
 * planes [z0, z0+nz) and keeps `ghost` planes below and above them. ghost >= 1. After every ca3d_slab_step
 * batch the host refreshes the ghosts (ca3d_slab_region + its transport, e.g. RCCL send/recv).
 * UNPACKED slabs need a power-of-two G (the legacy kernel's -1 wrap is only a torus then).
 */
int ca3d_configure_slab(ca3d_t *h, uint32_t g, int layout, uint32_t z0, uint32_t nz, uint32_t ghost);

/*
 * Rule buffers exactly as the reference uploads them (main_pathtraced.js:1330-1369) and binds them
 * (1647-1673: 0 main offsets, 1 edges offsets, 2 corners offsets, 3 survive, 4 born). Offset lists are flat
 * xyz triples of i32 (n_* = number of i32, a multiple of 3, every component in {-1,0,1}; duplicates and
 * (0,0,0) are legal and count as the reference kernel would count them). survive/born: 81 u32, slots
 * 0-26 main, 27-53 edges, 54-80 corners. PACKED32 treats an entry as set iff it == 1
 * (compute_clustered.wgsl:232); UNPACKED uses the main list and slots 0-26 with `> 0` (compute.wgsl:160-166).
 */
int ca3d_set_rules(ca3d_t *h,
                   const int32_t *main_offsets, uint32_t n_main,
                   const int32_t *edges_offsets, uint32_t n_edges,
                   const int32_t *corners_offsets, uint32_t n_corners,
                   const uint32_t survive[CA3D_LUT_LEN], const uint32_t born[CA3D_LUT_LEN]);

/*
 * queue.writeBuffer(cell_state_0 / cell_state_1) (main_pathtraced.js:1361-1362): the same words go to both
 * ping-pong buffers and the step counter restarts at 0. n_words must equal the state size:
 * PACKED32 (G/32)*G*G, UNPACKED G^3 (slab: the owned planes only; ghosts are filled by the halo exchange).
 */
int ca3d_upload_state(ca3d_t *h, const uint32_t *words, size_t n_words);

/* Read-back of the current state = buffer [step % 2] (no reference counterpart: parity checks, checkpoints). */
int ca3d_read_state(ca3d_t *h, uint32_t *words, size_t n_words);

/*
 * _computePass (main_pathtraced.js:1796-1809) n_steps times: step k reads buffer k % 2, writes buffer
 * (k+1) % 2; afterwards the current state is buffer [total_steps % 2] and the other buffer holds the state one
 * step earlier, as in the reference. Asynchronous.
 *
 * The reference encodes its passes into a command encoder and hands them to the GPU with ONE queue.submit per frame
 * (main_pathtraced.js:1833-1850). ca3d_set_option("queue", n > 0) gives ca3d_step that meaning: a call only ENCODES its
 * steps; they are submitted together by ca3d_flush, by any other call on the engine that looks at the state, the stream or
 * the options (read-back, render, synchronize, set_rules, get_info ...), or as soon as n steps are waiting. What a caller can
 * observe through the engine is the same either way; consecutive short batches then cost one launch of the resident
 * multi-step kernel instead of one each. A caller that records its own events on the stream calls ca3d_flush first.
 * Off by default ("queue" 0: every ca3d_step submits its own steps).
 */
int ca3d_step(ca3d_t *h, uint32_t n_steps);
int ca3d_flush(ca3d_t *h);

/* Slab mode: n_steps <= ghost sub-steps on a shrinking plane range; then the ghosts must be refreshed. */
int ca3d_slab_step(ca3d_t *h, uint32_t n_steps);

/*
 * The same batch in two phases, so that the halo exchange overlaps the bulk of the compute (BASELINE configs[4]:
 * "halo overlapped with compute"):
 *   CA3D_SLAB_PHASE_EDGES     all n sub-steps on the two edge zones only — afterwards the planes the neighbours
 *                             need (SEND_LOW / SEND_HIGH of ca3d_slab_region) hold their final values of this batch;
 *   (the caller starts the exchange of those planes into the neighbours' ghosts here)
 *   CA3D_SLAB_PHASE_INTERIOR  all n sub-steps on the planes in between, commits the batch (step counter, current
 *                             buffer). It reads no ghost plane, so the exchange may run concurrently with it.
 * EDGES then INTERIOR with the same n equals one ca3d_slab_step(n). Slabs too thin to split (nz + 2 <= 2*ghost + 2*n)
 * run the whole batch in the edge phase. Between the two phases ca3d_slab_region refers to the buffer the batch
 * ends in (where the edge results are, and where the incoming ghosts belong).
 */
enum ca3d_slab_phase
{
	CA3D_SLAB_PHASE_ALL = 0,
	CA3D_SLAB_PHASE_EDGES = 1,
	CA3D_SLAB_PHASE_INTERIOR = 2
};
int ca3d_slab_step_phase(ca3d_t *h, uint32_t n_steps, int phase);

enum ca3d_slab_region_id
{
	CA3D_SLAB_SEND_LOW = 0,  /* first `ghost` owned planes  -> lower neighbour's high ghost */
	CA3D_SLAB_SEND_HIGH = 1, /* last `ghost` owned planes   -> upper neighbour's low ghost  */
	CA3D_SLAB_RECV_LOW = 2,  /* this engine's low ghost planes  */
	CA3D_SLAB_RECV_HIGH = 3, /* this engine's high ghost planes */
	CA3D_SLAB_OWNED = 4      /* all owned planes */
};
/* Device pointer + byte size of a region of the CURRENT buffer (changes with step parity; after an edge phase: of
 * the buffer that batch ends in). */
int ca3d_slab_region(ca3d_t *h, int region, void **device_ptr, size_t *n_bytes);

/*
 * Halo transport inside the engine: RCCL send / receive over xGMI between the ranks of a Z-slab chain, one process per
 * GPU (no reference counterpart). librccl is loaded on first use. Bootstrap: rank 0 calls
 * ca3d_comm_unique_id and hands the 128 bytes to the other ranks by any means (torch.distributed, MPI, a socket, a
 * file); every rank then calls ca3d_slab_comm_init on its slab engine (collective: all ranks must call it).
 *   ca3d_slab_run(n, overlap)  n CA steps in batches of <= ghost sub-steps with the ghost planes refreshed between
 *                              batches; overlap != 0: the edge zones of a batch first, their planes travel on a second
 *                              stream while the interior runs (BASELINE configs[4]). Asynchronous; everything is
 *                              enqueued by this one call (no per-batch host work beyond the launches).
 *   ca3d_slab_exchange         one refresh of the ghost planes of the current state (ca3d_slab_run does it by itself)
 *   ca3d_slab_gather           ncclAllGather of every rank's owned planes into the current buffer of `full`, a full-grid
 *                              engine on the same device: the volume a renderer needs (shadow rays cross slabs)
 * The chain follows the kernel's boundary: packed — open at the bottom, closed at the top; unpacked — a ring.
 */
#define CA3D_COMM_ID_BYTES 128
int ca3d_comm_unique_id(void *id_bytes);
int ca3d_slab_comm_init(ca3d_t *h, const void *id_bytes, int rank, int world);
int ca3d_slab_run(ca3d_t *h, uint32_t n_steps, int overlap);
int ca3d_slab_exchange(ca3d_t *h);
/* Evidence of what a multi-GPU run really ran on (bench.py prints it per rank): the engine's HIP device and its PCI bus id
 * (hipDeviceGetPCIBusId), and — once ca3d_slab_comm_init has built the engine's RCCL communicator — what the COMMUNICATOR
 * reports: ncclCommCount, ncclCommUserRank, ncclCommCuDevice (-1 each without one). Eight ranks are eight GPUs only if
 * the eight bus ids differ. */
typedef struct ca3d_comm_info
{
	int32_t device;       /* the engine's HIP device ordinal */
	int32_t comm_ranks;   /* ncclCommCount */
	int32_t comm_rank;    /* ncclCommUserRank */
	int32_t comm_device;  /* ncclCommCuDevice */
	char pci_bus_id[32];  /* "0000:05:00.0" */
} ca3d_comm_info;
int ca3d_slab_comm_info(ca3d_t *h, ca3d_comm_info *out);
int ca3d_slab_gather(ca3d_t *h, ca3d_t *full);

/*
 * The same split driven by ONE host thread (SURVEY 8(b) sketched `ca3d_create(const int* device_ids, int n_devices, ...)`):
 * the reference's host is a single JavaScript thread that enqueues everything (main_pathtraced.js:1821-1854), and BASELINE's
 * north star keeps the host in JavaScript while the grid is Z-slabbed over the GPUs of a node. A group = one slab engine per
 * entry of the device list, rank k owning planes [k G/n, (k+1) G/n) (a device may be listed more than once: several slabs
 * on one GPU, which is how the path is tested on one GPU), plus the ghost exchange between them.
 *   ca3d_group_configure     grid, layout, ghost depth K = steps between exchanges (ca3d_configure_slab on every engine)
 *   ca3d_group_set_rules     ca3d_set_rules on every engine (payload as there)
 *   ca3d_group_upload_state  the FULL grid in the reference's layout; each slab takes its planes
 *   ca3d_group_read_state    the full grid back
 *   ca3d_group_step          n steps: batches of <= K sub-steps on every device, ghosts refreshed between batches; everything
 *                            is enqueued by this one call, nothing waits for a GPU
 *   ca3d_group_render        ca3d_render for the whole grid: every rank gets the full packed volume (peer copies), renders
 *                            its band of image rows and the bands land in the caller's buffers
 *   ca3d_group_engine        the slab engine of a rank (for ca3d_get_info, ca3d_get_stats ...); owned by the group
 *   ca3d_group_set_option    "transport" 0 (default): the ghost planes travel as peer-to-peer device copies over xGMI on the
 *                            receiving engine's stream, ordered by events; 1: ncclSend / ncclRecv on communicators from
 *                            ncclCommInitAll, one ncclGroupStart / End per exchange (one device per slab). Every other name
 *                            goes to ca3d_set_option of every engine.
 * Packed grids: the chain is open at the bottom and closed at the top; unpacked: a ring (as ca3d_slab_comm_init).
 */
typedef struct ca3d_group ca3d_group_t;
int ca3d_group_create(const int *device_ids, int n_devices, ca3d_group_t **out);
int ca3d_group_destroy(ca3d_group_t *g);
int ca3d_group_size(ca3d_group_t *g, int *out_n);
int ca3d_group_engine(ca3d_group_t *g, int rank, ca3d_t **out);
int ca3d_group_configure(ca3d_group_t *g, uint32_t grid_size, int layout, uint32_t ghost);
int ca3d_group_set_rules(ca3d_group_t *g,
                         const int32_t *main_offsets, uint32_t n_main,
                         const int32_t *edges_offsets, uint32_t n_edges,
                         const int32_t *corners_offsets, uint32_t n_corners,
                         const uint32_t survive[CA3D_LUT_LEN], const uint32_t born[CA3D_LUT_LEN]);
int ca3d_group_upload_state(ca3d_group_t *g, const uint32_t *words, size_t n_words);
int ca3d_group_read_state(ca3d_group_t *g, uint32_t *words, size_t n_words);
int ca3d_group_step(ca3d_group_t *g, uint32_t n_steps);
int ca3d_group_synchronize(ca3d_group_t *g);
int ca3d_group_set_option(ca3d_group_t *g, const char *name, int64_t value);
int ca3d_group_render(ca3d_group_t *g, const float uniforms[128], uint32_t width, uint32_t height, uint32_t spp,
                      uint8_t *presentation_rgba8, uint16_t *light_rgba16f, uint16_t *depth_rg16f);

/* Device pointer + byte size of the targets of the last ca3d_render call: 0 presentation RGBA8, 1 light RGBA16F,
 * 2 depth RG16F (row-major, top row first) — e.g. to gather the bands of a frame shared between GPUs without a
 * round trip through the host. */
int ca3d_render_target(ca3d_t *h, int which, void **device_ptr, size_t *n_bytes);

int ca3d_synchronize(ca3d_t *h);

/* Resident launches that gave up (a wait for neighbour tile faces timed out: not all workgroups were on the chip) and
 * whose steps the engine re-ran through the per-step kernels, since ca3d_create. The calls that noticed returned CA3D_OK
 * with the final state intact; ca3d_last_error() carried a note and the resident path stays off until
 * ca3d_set_option("resident", 1). Slab engines do not recover (the neighbours' ghosts came from the failed launch):
 * they return CA3D_ERR_DEVICE and want a new upload. */
int ca3d_recovered_launches(ca3d_t *h, uint32_t *out_count);

/* Interop: run on a caller-owned hipStream_t (e.g. torch's current stream). NULL is HIP's legacy default stream
 * (what torch uses unless told otherwise), not "none". ca3d_use_own_stream goes back to the engine's stream. */
int ca3d_set_stream(ca3d_t *h, void *hip_stream);
int ca3d_use_own_stream(ca3d_t *h);
/* Device pointer of ping-pong buffer `which` (0/1) — whole allocation including ghosts. Valid until the next ca3d_step /
 * ca3d_upload_state / ca3d_configure on the engine: a resident multi-step launch writes its result to a third buffer and
 * rotates the three (buffer [step % 2] is always the current state, the other one the state one step earlier).
 * The caller may WRITE the state through the pointer (on the engine's stream, or ordered against it) without telling the
 * engine: while the pointer is valid every ca3d_render rebuilds what it derives from the state (occupancy bits, the bricked
 * copy) instead of reusing an earlier frame's — one extra pass over the state per frame, the price of not having to
 * announce writes. */
int ca3d_device_buffer(ca3d_t *h, int which, void **device_ptr, size_t *n_bytes);

typedef struct ca3d_info
{
	uint32_t grid_size;
	int32_t layout;
	uint32_t z0, nz, ghost; /* slab (full grid: 0, G, 0) */
	uint64_t step;          /* steps since the last upload */
	uint64_t state_words;   /* words ca3d_upload_state / ca3d_read_state expect */
	int32_t current_buffer; /* step % 2 */
	int32_t device;
	char kernel_name[64]; /* kernel variant the current rules select */
	uint64_t launches_total; /* kernel launches issued by the step calls since ca3d_create */
} ca3d_info;
int ca3d_get_info(ca3d_t *h, ca3d_info *out);

/*
 * createShaderModule / createComputePipeline (main_pathtraced.js:1421-1433) report compile problems through the
 * browser console; here the step kernel is specialised for the rule at run time inside ca3d_set_rules /
 * ca3d_configure (option "jit"), and a failed compile does NOT fail those calls — the pre-built kernels take over,
 * ca3d_get_info().kernel_name lacks its "(jit)" suffix, ca3d_last_error() holds the message right after the call and
 * this getter returns the compiler log of the most recent attempt (empty string: no failure). `needed` (nullable)
 * receives the size including the terminator; the text is truncated to n_bytes.
 */
int ca3d_get_jit_log(ca3d_t *h, char *buf, size_t n_bytes, size_t *needed);

/*
 * The run-time compiler's bookkeeping for THIS PROCESS (all engines): how many programs were compiled, how many came
 * from the on-disk cache of code objects ($CA3D_CACHE_DIR, else $XDG_CACHE_HOME/ca3d, else $HOME/.cache/ca3d;
 * CA3D_JIT_CACHE=0 turns it off; files are keyed on architecture, compiler version, options and the full program
 * text, so they never go stale) and how many were already loaded, with the milliseconds each kind cost. A process
 * that finds its rules in the cache compiles nothing: programs_compiled == 0 — `_restartSim` (main_pathtraced.js:
 * 624-637) stays cheap from the second start on. Needs no engine.
 */
typedef struct ca3d_jit_stats
{
	uint64_t programs_compiled;    /* hiprtc compiles */
	uint64_t programs_from_disk;   /* code objects read from the cache directory */
	uint64_t programs_from_memory; /* requests answered by a module this process had loaded already */
	double compile_ms;             /* inside hiprtc */
	double disk_read_ms;           /* reading + verifying cached objects */
	double load_ms;                /* hipModuleLoadData (+ writing new objects to the cache) */
	char cache_dir[256];           /* "" when the disk cache is off or its directory cannot be created */
} ca3d_jit_stats;
int ca3d_get_jit_stats(ca3d_jit_stats *out);

/*
 * Diagnostics (no reference counterpart): everything that decides WHICH instruction stream the next step batch runs, as one
 * string — kernel name, grid, a hash of the rule payload, the resident-kernel form options, a hash of the device sources the
 * run-time compiler is given. A profile (instruction counts per step, tools/pmc_sq_reduce.py) is only comparable with a run
 * whose string is the same: bench.py / js/bench.js price a resident kernel's time against a committed profile only then.
 * `needed` (nullable) receives the size including the terminator; the text is truncated to n_bytes.
 */
int ca3d_get_kernel_variant(ca3d_t *h, char *buf, size_t n_bytes, size_t *needed);

typedef struct ca3d_stats
{
	uint64_t steps;          /* steps in the last ca3d_step / ca3d_slab_step batch */
	uint64_t kernel_launches; /* kernel launches that batch issued */
	double gpu_ms;           /* hipEvent time around the batch on the engine's stream (waits for it) */
	double cell_steps;       /* cells updated x steps */
	double algorithmic_bytes; /* 0.25 B (PACKED32) or 8 B (UNPACKED) per cell-step: SURVEY 8(d) */
} ca3d_stats;
int ca3d_get_stats(ca3d_t *h, ca3d_stats *out);

/*
 * _updateUniforms + _renderPass (main_pathtraced.js:1747-1750, 1775-1794) and the history textures (729-779):
 * renders the CURRENT state (buffer step % 2, as bind group 2 of the render pass, 1788) through the volume
 * renderer. `uniforms` is the reference's 128-float common block verbatim (MemoryManager.bufferf32; layout
 * pathtraced_fragment_clustered.wgsl:17-34); width/height give the pixel grid (the shader's windowSize
 * uniform supplies the aspect ratio, as in the reference). spp is 1 (pixel centre) or 4 (2x2 stratified
 * sub-samples averaged in linear light before gamma). Outputs, each nullable, row-major, top row first:
 *   presentation_rgba8  width*height*4 bytes   pow(rgb, 1/gamma), alpha  -> the canvas attachment
 *   light_rgba16f       width*height*4 halfs   linear rgb, 1              -> light history attachment
 *   depth_rg16f         width*height*2 halfs   distance from camera, 1    -> depth history attachment
 * The engine keeps the device-side targets and swaps its two history surfaces per call (1793). Full grid only.
 * Synchronous only when an output pointer is given.
 *
 * What a frame is: by default the frame the reference's jittered, temporally accumulated process converges to
 * under a static camera (exact cell walk). ca3d_set_option("render_mode", 1) switches to ONE literal
 * reference frame per call (jittered marches + history look-ups + temporal blend; spp must be 1; packed layout);
 * "render_reset_history" clears the history surfaces. An engine configured with CA3D_LAYOUT_UNPACKED renders
 * through the legacy shader model (shaders/pathtraced_fragment.wgsl).
 *
 * A call without output pointers only enqueues and does NOT wait for a pending resident multi-step launch to be
 * verified (ca3d_recovered_launches): should that launch later turn out to have timed out, the frame was drawn from
 * its unwritten output — wrong once; the engine then re-runs the steps and clears the history surfaces, so the
 * literal mode does not keep blending that frame in. A call WITH an output pointer settles the launch first.
 */
int ca3d_render(ca3d_t *h, const float uniforms[128], uint32_t width, uint32_t height, uint32_t spp,
                uint8_t *presentation_rgba8, uint16_t *light_rgba16f, uint16_t *depth_rg16f);

/* Measurement helper (no reference counterpart; SURVEY 8(d) "also report against a measured device-to-device copy
 * ceiling"): `reps` float4-per-lane copies of n_bytes between two scratch buffers on the engine's stream, HIP events
 * around them; *gb_per_s = bytes read + bytes written per second / 1e9. Waits for the GPU. */
int ca3d_measure_copy(ca3d_t *h, size_t n_bytes, uint32_t reps, double *gb_per_s);

typedef struct ca3d_render_stats
{
	double gpu_ms;          /* hipEvent time of the last ca3d_render kernel */
	uint64_t primary_rays;  /* width * height * spp */
	uint64_t shadow_rays;   /* samples that reached the shading gate and traced a shadow ray */
	uint64_t primary_cell_visits, shadow_cell_visits; /* cells the two walks stepped through */
} ca3d_render_stats;
int ca3d_get_render_stats(ca3d_t *h, ca3d_render_stats *out);

/* How many converged frames the engine keeps in flight (option "render_pipeline" below): the number of internal streams — on pairwise
 * different hardware queues, probed when the first pipelined frame is drawn — that such frames alternate between; 0 before that frame,
 * with the option off, or when the runtime gave the engine no two streams that run side by side. Does not wait for the GPU. */
int ca3d_get_render_pipeline(ca3d_t *h, int32_t *frames_in_flight);

/* Options (not part of the reference surface): "queue" n: queued submission (see ca3d_step; 0 = off); "graph" 0/1 hipGraph batching; "stats" 0/1: record the event pair
 * ca3d_get_stats reads around every ca3d_step batch (on by default; a host that steps in small batches and never asks
 * for stats saves two marker packets per call); "graph_prepare" n builds now the
 * graphs a later ca3d_step(n) replays — a batch of any length up to 1024 steps is one graph of exactly that many
 * steps (otherwise built on first use); "fused" 0/1 two-step fused kernel
 * (bit-exact, off by default); "variant" 1 forces the generic / literal kernels; "jit" 0/1 run-time (hiprtc) specialisation of the step
 * kernel for the current rule, compiled inside ca3d_set_rules / ca3d_configure (on by default; a failed compile
 * keeps the pre-built kernels and is reported through ca3d_get_jit_log); "resident" 0/1: batches of "resident_min" (default 8) steps and more run as ONE launch of
 * the resident multi-step kernel where one exists (512^3 and 256^3 — von Neumann rule tables, and rule-sets with diagonal classes
 * compiled at run time: the state stays in registers and only tile faces cross the chip; 64^3, von Neumann rule tables: the
 * whole grid in one workgroup; a resident kernel is only selected when the runtime says all its workgroups fit on the CUs the
 * engine's stream may use; every in-kernel wait is bounded by "resident_timeout_us", default 200 000 — after a timeout the
 * engine re-runs the affected steps through the per-step kernels at the next call that looks at the state and turns the
 * path off: ca3d_recovered_launches; "resident_fault_tile" t: diagnostics, tile t - 1 of the next resident launch leaves at
 * once, which makes that launch time out);
 * "roll" 0/1 the rolling-window form of the run-time compiled class kernels (on by
 * default where it applies), "roll_tile" 0-3 which of its forms (0 every thread shifts its three rows, 1 workgroup tiles sharing the
 * shifted rows through LDS, 2 wave tiles, 3 two words per thread), "roll_z" 0/2/4/8/16/15/30 its planes per thread
 * (0: chosen per launch); "resident_zsplit" 1/2, "resident_rows" 32/16: tiling of the resident von Neumann kernels, "resident_pair" 1/0: at 512^3 with the
 * default tiling a thread owns two adjacent rows x 16 planes instead of one row x 32 (less LDS traffic; default 1); "graph_min" n: batches shorter than n
 * steps are launched kernel by kernel instead of as a captured graph; "render_mode" 0/1; "render_row_begin" / "render_row_end":
 * ca3d_render then fills image rows [begin, end) only (begin a multiple of 16; 0 / 0 = the whole frame) — a rank's
 * band when the GPUs of a node share one frame; "render_skip" 0/1 empty-space skipping by
 * 32x8x8-cell occupancy blocks, active on sparse volumes only (on by default; within the renderer's tolerance of
 * the cell-by-cell walk, not bit-identical to it); "render_sched" 0/1 dynamic ray
 * scheduling inside each wave of the renderer (on by default; the frame is the same bit for bit);
 * "render_reset_history"; "render_indirect" 0/1 adds the one-bounce neighbour lighting of calculateIndirectLighting
 * (pathtraced_fragment_clustered.wgsl:307-377 — present in the reference, its call commented out at :424; off by
 * default like there; converged-frame mode, packed layout); "render_stream" 0/1 the converged frame of a dense packed volume as
 * ray-stream passes (render_stream.hip; on by default; 0: the in-wave scheduled kernel; the same frame bit for bit);
 * "render_stream_check" 0/1 diagnostics: the stream passes count where their interval filter and the slab test disagree and which
 * answers were looked up unset — ca3d_render fails with CA3D_ERR_DEVICE if any (off by default; takes the frame off the pipeline);
 * "render_pipeline" 0 / 1 / 2-4 converged frames in flight (1, the default: four up to 24 M samples a frame, three above; 0: none;
 * 2-4: that many): frames that stay on the device (no host pointers), are drawn by the stream passes and go down the engine's OWN
 * stream alternate between that many internal streams; a frame that finds another one still in flight sizes its persistent walk
 * launches for its share of the chip, so that the frames' walks run side by side (a frame that finds the engine idle takes the whole
 * chip: a host that draws one frame per display refresh loses nothing; the first frame after a step, an upload or any other call
 * on the engine's stream is drawn on that stream); the engine's stream waits for them at the next call that
 * touches the state, a render target (ca3d_render_target, ca3d_get_render_stats) or the stream. Each frame is the frame of
 * one-at-a-time rendering, bit for bit; a caller on a stream of its own (ca3d_set_stream), a frame with host pointers, a band or a
 * literal frame is never pipelined; ca3d_get_render_pipeline reports the depth in use; the first pipelined frame of an engine probes
 * the runtime's streams for separate hardware queues (a few milliseconds per pair, once);
 * "render_frame_bricks" 0/1 the literal frame as a batched march over the bricked volume (on by default; 0: the statement-by-
 * statement form, the same frame bit for bit); "rows" 0/1 the run-time compiled rows kernel on grids that are not a power of two (on by
 * default; 0: the kernels that served them before — tests, tuning). */
int ca3d_set_option(ca3d_t *h, const char *name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* CA3D_H */
