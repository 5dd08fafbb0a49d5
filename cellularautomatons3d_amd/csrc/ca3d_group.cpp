// ca3d_group_*: ONE host thread drives the Z-slab split of a grid over the GPUs of a node (include/ca3d.h). The reference's
// host is a single JavaScript thread that owns everything (main_pathtraced.js:1821-1854); BASELINE's north star keeps the host
// in JavaScript and spreads the grid over 8 GPUs — so the split has to be drivable from one thread, without a process per
// GPU. A group is n slab engines (ca3d_configure_slab), one per entry of the device list (the same device may appear more
// than once: that is how the path is tested on one GPU), plus the ghost-plane exchange between them:
//   * transport "copy" (default): peer-to-peer device copies over xGMI, enqueued on the receiving engine's stream and
//     ordered by events — no collective library, any mix of devices;
//   * transport "rccl": one communicator per engine from ncclCommInitAll, every exchange one ncclGroupStart / End
//     around all ranks' sends and receives (distinct devices only: RCCL refuses two ranks on one device).
// The chain follows the kernel's boundary: packed — open at the bottom, closed at the top; unpacked — a ring.
#include <cstring>
#include <new>
#include <vector>

#include "ca3d_internal.h"

using namespace ca3d;

struct ca3d_group
{
	std::vector<int> devices;
	std::vector<ca3d_t *> eng;   // slab engines, rank order = z order
	std::vector<ca3d_t *> full;  // full-grid engines for rendering (one per rank, created at the first ca3d_group_render)
	std::vector<hipEvent_t> ev_done, ev_copied;
	uint32_t G = 0, ghost = 0;
	int layout = CA3D_LAYOUT_PACKED32;
	bool configured = false, has_rules = false, has_state = false;
	bool ghosts_valid = false;
	int transport = 0; // 0 copy, 1 rccl
	bool rccl_ready = false;
	uint64_t step = 0;
	// the rules as handed in, for the full-grid render engines
	std::vector<int32_t> r_main, r_edges, r_corners;
	uint32_t r_survive[CA3D_LUT_LEN] = {}, r_born[CA3D_LUT_LEN] = {};
	uint32_t rw = 0, rh = 0;
};

namespace
{

#define G_HIP_TRY(expr)                                                                                                              \
	do                                                                                                                               \
	{                                                                                                                                \
		hipError_t e_ = (expr);                                                                                                      \
		if (e_ != hipSuccess) return set_error(e_ == hipErrorOutOfMemory ? CA3D_ERR_OUT_OF_MEMORY : CA3D_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
	} while (0)
#define G_TRY(expr)           \
	do                        \
	{                         \
		int rc_ = (expr);     \
		if (rc_) return rc_;  \
	} while (0)

int P(const ca3d_group *g) { return (int)g->eng.size(); }

// device-to-device copy onto `stream` of the destination's device; a peer copy when the devices differ
int copy_planes(void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t stream)
{
	if (dst_dev == src_dev) G_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
	else G_HIP_TRY(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, stream));
	return CA3D_OK;
}

// Refresh every rank's ghost planes from its neighbours' owned planes of the current state. Ordering, per exchange:
// rank s has finished its batch (ev_done[s]) before a neighbour copies out of it; a rank's next batch starts only after the
// neighbours that read from it have done so (ev_copied[d]) — two batches later it overwrites those planes.
int exchange_copy(ca3d_group *g)
{
	const int n = P(g);
	const bool ring = g->layout == CA3D_LAYOUT_UNPACKED;
	for (int k = 0; k < n; k++)
	{
		G_HIP_TRY(hipSetDevice(g->devices[(size_t)k]));
		G_HIP_TRY(hipEventRecord(g->ev_done[(size_t)k], engine_stream(g->eng[(size_t)k])));
	}
	for (int d = 0; d < n; d++)
	{
		const int above = (d + 1) % n, below = (d + n - 1) % n;
		ca3d_t *e = g->eng[(size_t)d];
		hipStream_t s = engine_stream(e);
		G_HIP_TRY(hipSetDevice(g->devices[(size_t)d]));
		void *dst, *src;
		size_t bytes, sb;
		// high ghost <- the first planes of the rank above (the top rank: rank 0's, plane G wraps to plane 0)
		G_TRY(ca3d_slab_region(e, CA3D_SLAB_RECV_HIGH, &dst, &bytes));
		G_TRY(ca3d_slab_region(g->eng[(size_t)above], CA3D_SLAB_SEND_LOW, &src, &sb));
		if (above != d) G_HIP_TRY(hipStreamWaitEvent(s, g->ev_done[(size_t)above], 0));
		G_TRY(copy_planes(dst, g->devices[(size_t)d], src, g->devices[(size_t)above], bytes, s));
		// low ghost <- the last planes of the rank below; packed: rank 0 has none (z == -1 is dead)
		if (ring || d != 0)
		{
			G_TRY(ca3d_slab_region(e, CA3D_SLAB_RECV_LOW, &dst, &bytes));
			G_TRY(ca3d_slab_region(g->eng[(size_t)below], CA3D_SLAB_SEND_HIGH, &src, &sb));
			if (below != d) G_HIP_TRY(hipStreamWaitEvent(s, g->ev_done[(size_t)below], 0));
			G_TRY(copy_planes(dst, g->devices[(size_t)d], src, g->devices[(size_t)below], bytes, s));
		}
		G_HIP_TRY(hipEventRecord(g->ev_copied[(size_t)d], s));
	}
	for (int k = 0; k < n; k++)
	{
		// who read from rank k: the rank below (k's first planes -> its high ghost) and the rank above (k's last planes -> its low ghost)
		const int above = (k + 1) % n, below = (k + n - 1) % n;
		G_HIP_TRY(hipSetDevice(g->devices[(size_t)k]));
		hipStream_t s = engine_stream(g->eng[(size_t)k]);
		if (below != k) G_HIP_TRY(hipStreamWaitEvent(s, g->ev_copied[(size_t)below], 0));
		if (above != k && above != below && (ring || above != 0)) G_HIP_TRY(hipStreamWaitEvent(s, g->ev_copied[(size_t)above], 0));
		engine_set_ghosts_valid(g->eng[(size_t)k], true);
	}
	return CA3D_OK;
}

int exchange(ca3d_group *g)
{
	if (g->transport == 1)
	{
		if (!g->rccl_ready)
		{
			G_TRY(engines_rccl_init_all(g->eng.data(), P(g)));
			g->rccl_ready = true;
		}
		G_TRY(engines_rccl_exchange_all(g->eng.data(), P(g)));
	}
	else G_TRY(exchange_copy(g));
	g->ghosts_valid = true;
	return CA3D_OK;
}

int check_group(const ca3d_group *g, bool need_state)
{
	if (!g) return set_error(CA3D_ERR_INVALID_ARGUMENT, "NULL group handle");
	if (!g->configured) return set_error(CA3D_ERR_NOT_CONFIGURED, "ca3d_group_configure has not been called");
	if (need_state && !g->has_rules) return set_error(CA3D_ERR_NOT_CONFIGURED, "ca3d_group_set_rules has not been called");
	if (need_state && !g->has_state) return set_error(CA3D_ERR_NOT_CONFIGURED, "ca3d_group_upload_state has not been called");
	return CA3D_OK;
}

void destroy_full(ca3d_group *g)
{
	for (ca3d_t *f : g->full) ca3d_destroy(f);
	g->full.clear();
	g->rw = g->rh = 0;
}

// image rows [begin, end) rank renders when the ranks share a frame: bands of whole 16-row tiles (slab.band_rows)
void band_rows(uint32_t height, int world, int rank, uint32_t *y0, uint32_t *y1)
{
	const uint32_t tiles = (height + 15u) / 16u;
	const uint32_t lo = (uint32_t)((uint64_t)tiles * (uint32_t)rank / (uint32_t)world), hi = (uint32_t)((uint64_t)tiles * (uint32_t)(rank + 1) / (uint32_t)world);
	*y0 = lo * 16u < height ? lo * 16u : height;
	*y1 = hi * 16u < height ? hi * 16u : height;
}

} // namespace

extern "C"
{

int ca3d_group_create(const int *device_ids, int n_devices, ca3d_group_t **out) CA3D_API_TRY
{
	if (!out) return set_error(CA3D_ERR_INVALID_ARGUMENT, "out is NULL");
	*out = nullptr;
	if (!device_ids || n_devices < 1 || n_devices > 64) return set_error(CA3D_ERR_INVALID_ARGUMENT, "a group takes 1 to 64 devices");
	ca3d_group *g = new (std::nothrow) ca3d_group();
	if (!g) return set_error(CA3D_ERR_OUT_OF_MEMORY, "out of host memory");
	for (int k = 0; k < n_devices; k++)
	{
		ca3d_t *e = nullptr;
		int rc = ca3d_create(device_ids[k], &e);
		hipEvent_t a = nullptr, b = nullptr;
		if (rc == CA3D_OK && (hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess))
			rc = set_error(CA3D_ERR_DEVICE, "hipEventCreate failed on device %d", device_ids[k]);
		if (rc)
		{
			if (e) ca3d_destroy(e);
			if (a) hipEventDestroy(a);
			ca3d_group_destroy(g);
			return rc;
		}
		g->devices.push_back(device_ids[k]);
		g->eng.push_back(e);
		g->ev_done.push_back(a);
		g->ev_copied.push_back(b);
	}
	// peer access between chain neighbours (and to device 0, where frames are assembled): the copies then go over xGMI
	for (int k = 0; k < n_devices; k++)
		for (int j : {(k + 1) % n_devices, (k + n_devices - 1) % n_devices, 0})
		{
			const int a = g->devices[(size_t)k], b = g->devices[(size_t)j];
			int can = 0;
			if (a == b || hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) continue;
			if (hipSetDevice(a) == hipSuccess && hipDeviceEnablePeerAccess(b, 0) != hipSuccess) (void)hipGetLastError(); // already enabled
		}
	*out = g;
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_group_destroy(ca3d_group_t *g) CA3D_API_TRY
{
	if (!g) return CA3D_OK;
	destroy_full(g);
	for (size_t k = 0; k < g->eng.size(); k++)
	{
		hipSetDevice(g->devices[k]);
		ca3d_destroy(g->eng[k]);
		if (g->ev_done[k]) hipEventDestroy(g->ev_done[k]);
		if (g->ev_copied[k]) hipEventDestroy(g->ev_copied[k]);
	}
	delete g;
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_group_size(ca3d_group_t *g, int *out_n) CA3D_API_TRY
{
	if (!g || !out_n) return set_error(CA3D_ERR_INVALID_ARGUMENT, "NULL argument");
	*out_n = P(g);
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_group_engine(ca3d_group_t *g, int rank, ca3d_t **out) CA3D_API_TRY
{
	if (!g || !out) return set_error(CA3D_ERR_INVALID_ARGUMENT, "NULL argument");
	if (rank < 0 || rank >= P(g)) return set_error(CA3D_ERR_INVALID_ARGUMENT, "rank %d of %d", rank, P(g));
	*out = g->eng[(size_t)rank];
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_group_configure(ca3d_group_t *g, uint32_t grid_size, int layout, uint32_t ghost) CA3D_API_TRY
{
	if (!g) return set_error(CA3D_ERR_INVALID_ARGUMENT, "NULL group handle");
	const uint32_t n = (uint32_t)P(g);
	if (grid_size == 0 || grid_size % n) return set_error(CA3D_ERR_INVALID_ARGUMENT, "a grid of %u planes does not split evenly over %u slabs", grid_size, n);
	const uint32_t nz = grid_size / n;
	if (ghost == 0 || ghost > nz) return set_error(CA3D_ERR_INVALID_ARGUMENT, "ghost depth must be in [1, %u] (planes per slab)", nz);
	destroy_full(g);
	g->configured = g->has_state = false;
	for (uint32_t k = 0; k < n; k++) G_TRY(ca3d_configure_slab(g->eng[k], grid_size, layout, k * nz, nz, ghost));
	g->G = grid_size;
	g->layout = layout;
	g->ghost = ghost;
	g->configured = true;
	g->ghosts_valid = false;
	g->step = 0;
	if (g->has_rules) // rules outlive a re-configuration, as on a single engine
		for (uint32_t k = 0; k < n; k++)
			G_TRY(ca3d_set_rules(g->eng[k], g->r_main.data(), (uint32_t)g->r_main.size(), g->r_edges.data(), (uint32_t)g->r_edges.size(), g->r_corners.data(),
			                     (uint32_t)g->r_corners.size(), g->r_survive, g->r_born));
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_group_set_rules(ca3d_group_t *g, const int32_t *main_offsets, uint32_t n_main, const int32_t *edges_offsets, uint32_t n_edges,
                         const int32_t *corners_offsets, uint32_t n_corners, const uint32_t survive[CA3D_LUT_LEN], const uint32_t born[CA3D_LUT_LEN]) CA3D_API_TRY
{
	if (!g) return set_error(CA3D_ERR_INVALID_ARGUMENT, "NULL group handle");
	for (ca3d_t *e : g->eng) G_TRY(ca3d_set_rules(e, main_offsets, n_main, edges_offsets, n_edges, corners_offsets, n_corners, survive, born));
	for (ca3d_t *f : g->full) G_TRY(ca3d_set_rules(f, main_offsets, n_main, edges_offsets, n_edges, corners_offsets, n_corners, survive, born));
	g->r_main.assign(main_offsets, main_offsets + n_main);
	g->r_edges.assign(edges_offsets, edges_offsets + n_edges);
	g->r_corners.assign(corners_offsets, corners_offsets + n_corners);
	memcpy(g->r_survive, survive, sizeof g->r_survive);
	memcpy(g->r_born, born, sizeof g->r_born);
	g->has_rules = true;
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_group_upload_state(ca3d_group_t *g, const uint32_t *words, size_t n_words) CA3D_API_TRY
{
	G_TRY(check_group(g, false));
	if (!words) return set_error(CA3D_ERR_INVALID_ARGUMENT, "words is NULL");
	const size_t n = (size_t)P(g);
	ca3d_info info;
	G_TRY(ca3d_get_info(g->eng[0], &info));
	const size_t per = (size_t)info.state_words;
	if (n_words != per * n) return set_error(CA3D_ERR_INVALID_ARGUMENT, "state has %zu words, expected %zu", n_words, per * n);
	for (size_t k = 0; k < n; k++) G_TRY(ca3d_upload_state(g->eng[k], words + k * per, per)); // z is the slowest index: a slab is contiguous
	g->has_state = true;
	g->ghosts_valid = false;
	g->step = 0;
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_group_read_state(ca3d_group_t *g, uint32_t *words, size_t n_words) CA3D_API_TRY
{
	G_TRY(check_group(g, false));
	if (!g->has_state) return set_error(CA3D_ERR_NOT_CONFIGURED, "no state to read: upload first");
	if (!words) return set_error(CA3D_ERR_INVALID_ARGUMENT, "words is NULL");
	const size_t n = (size_t)P(g);
	ca3d_info info;
	G_TRY(ca3d_get_info(g->eng[0], &info));
	const size_t per = (size_t)info.state_words;
	if (n_words != per * n) return set_error(CA3D_ERR_INVALID_ARGUMENT, "state has %zu words, expected %zu", n_words, per * n);
	for (size_t k = 0; k < n; k++) G_TRY(ca3d_read_state(g->eng[k], words + k * per, per));
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_group_step(ca3d_group_t *g, uint32_t n_steps) CA3D_API_TRY
{
	G_TRY(check_group(g, true));
	if (n_steps == 0) return CA3D_OK;
	if (!g->ghosts_valid) G_TRY(exchange(g));
	uint32_t left = n_steps;
	while (left)
	{
		const uint32_t k = left < g->ghost ? left : g->ghost;
		for (ca3d_t *e : g->eng) G_TRY(ca3d_slab_step(e, k)); // asynchronous: every device gets its batch before any exchange is posted
		G_TRY(exchange(g));
		left -= k;
	}
	g->step += n_steps;
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_group_synchronize(ca3d_group_t *g) CA3D_API_TRY
{
	if (!g) return set_error(CA3D_ERR_INVALID_ARGUMENT, "NULL group handle");
	for (ca3d_t *e : g->eng) G_TRY(ca3d_synchronize(e));
	for (ca3d_t *f : g->full) G_TRY(ca3d_synchronize(f));
	return CA3D_OK;
}
CA3D_API_CATCH

int ca3d_group_set_option(ca3d_group_t *g, const char *name, int64_t value) CA3D_API_TRY
{
	if (!g || !name) return set_error(CA3D_ERR_INVALID_ARGUMENT, "NULL argument");
	if (!strcmp(name, "transport"))
	{
		if (value != 0 && value != 1) return set_error(CA3D_ERR_INVALID_ARGUMENT, "transport must be 0 (peer copies) or 1 (RCCL send / receive)");
		if (value == 1)
			for (size_t i = 0; i < g->devices.size(); i++)
				for (size_t j = i + 1; j < g->devices.size(); j++)
					if (g->devices[i] == g->devices[j]) return set_error(CA3D_ERR_UNSUPPORTED, "the RCCL transport needs one device per slab (device %d appears twice)", g->devices[i]);
		g->transport = (int)value;
		g->ghosts_valid = false;
		return CA3D_OK;
	}
	for (ca3d_t *e : g->eng) G_TRY(ca3d_set_option(e, name, value));
	if (!strncmp(name, "render_", 7))
		for (ca3d_t *f : g->full) G_TRY(ca3d_set_option(f, name, value));
	return CA3D_OK;
}
CA3D_API_CATCH

// The frame of the whole grid, shared between the GPUs: shadow rays cross slabs, so every rank gets the full
// packed volume (peer copies of every slab's owned planes into a full-grid engine per rank), renders its band of image rows
// ("replicas over pixels") and the bands land in the caller's buffers. A band is bit-identical to the same rows of a
// single-GPU frame.
int ca3d_group_render(ca3d_group_t *g, const float uniforms[128], uint32_t width, uint32_t height, uint32_t spp, uint8_t *presentation_rgba8,
                      uint16_t *light_rgba16f, uint16_t *depth_rg16f) CA3D_API_TRY
{
	G_TRY(check_group(g, true));
	if (g->layout != CA3D_LAYOUT_PACKED32) return set_error(CA3D_ERR_UNSUPPORTED, "the shared frame takes the packed layout");
	if (!uniforms) return set_error(CA3D_ERR_INVALID_ARGUMENT, "uniforms is NULL");
	const int n = P(g);
	if (g->full.empty())
	{
		for (int k = 0; k < n; k++)
		{
			ca3d_t *f = nullptr;
			int rc = ca3d_create(g->devices[(size_t)k], &f);
			// this engine never steps: no run-time compilation, no resident kernel (three kernel selections — configure, set_rules,
			// set_stream — would otherwise each compile for nothing)
			if (rc == CA3D_OK) rc = ca3d_set_option(f, "jit", 0);
			if (rc == CA3D_OK) rc = ca3d_set_option(f, "resident", 0);
			if (rc == CA3D_OK) rc = ca3d_configure(f, g->G, g->G, g->G, g->layout);
			if (rc == CA3D_OK) rc = ca3d_set_rules(f, g->r_main.data(), (uint32_t)g->r_main.size(), g->r_edges.data(), (uint32_t)g->r_edges.size(), g->r_corners.data(),
			                                        (uint32_t)g->r_corners.size(), g->r_survive, g->r_born);
			if (rc == CA3D_OK) rc = ca3d_set_stream(f, engine_stream(g->eng[(size_t)k])); // one stream per device: the gather, the band and the next batch stay ordered
			if (rc) { if (f) ca3d_destroy(f); destroy_full(g); return rc; }
			g->full.push_back(f);
		}
		// a full-grid engine wants a state before it renders; the gather below overwrites it (cleared on the device: no host copy of
		// the grid — 1 GiB at 2048^3 — and no synchronous upload)
		for (ca3d_t *f : g->full) G_TRY(engine_mark_state(f));
	}
	// every slab's owned planes -> every rank's full volume (the ranks' own batches are done: ev_done)
	for (int k = 0; k < n; k++)
	{
		G_HIP_TRY(hipSetDevice(g->devices[(size_t)k]));
		G_HIP_TRY(hipEventRecord(g->ev_done[(size_t)k], engine_stream(g->eng[(size_t)k])));
	}
	for (int d = 0; d < n; d++)
	{
		ca3d_info fi;
		G_TRY(ca3d_get_info(g->full[(size_t)d], &fi));
		void *vol;
		size_t vol_bytes;
		G_TRY(engine_state_buffer(g->full[(size_t)d], fi.current_buffer, &vol, &vol_bytes));
		G_HIP_TRY(hipSetDevice(g->devices[(size_t)d]));
		hipStream_t s = engine_stream(g->eng[(size_t)d]);
		for (int k = 0; k < n; k++)
		{
			void *owned;
			size_t bytes;
			G_TRY(ca3d_slab_region(g->eng[(size_t)k], CA3D_SLAB_OWNED, &owned, &bytes));
			if (k != d) G_HIP_TRY(hipStreamWaitEvent(s, g->ev_done[(size_t)k], 0));
			G_TRY(copy_planes((char *)vol + (size_t)k * bytes, g->devices[(size_t)d], owned, g->devices[(size_t)k], bytes, s));
		}
		G_HIP_TRY(hipEventRecord(g->ev_copied[(size_t)d], s));
	}
	for (int k = 0; k < n; k++) // a slab is not stepped on while somebody still copies out of it
	{
		G_HIP_TRY(hipSetDevice(g->devices[(size_t)k]));
		for (int d = 0; d < n; d++)
			if (d != k) G_HIP_TRY(hipStreamWaitEvent(engine_stream(g->eng[(size_t)k]), g->ev_copied[(size_t)d], 0));
	}
	// bands
	const size_t px_row = (size_t)width;
	for (int k = 0; k < n; k++)
	{
		uint32_t y0, y1;
		band_rows(height, n, k, &y0, &y1);
		if (y1 <= y0) continue;
		ca3d_t *f = g->full[(size_t)k];
		G_TRY(ca3d_set_option(f, "render_row_begin", y0));
		G_TRY(ca3d_set_option(f, "render_row_end", y1));
		G_TRY(ca3d_render(f, uniforms, width, height, spp, nullptr, nullptr, nullptr));
	}
	for (int k = 0; k < n; k++)
	{
		uint32_t y0, y1;
		band_rows(height, n, k, &y0, &y1);
		if (y1 <= y0) continue;
		ca3d_t *f = g->full[(size_t)k];
		G_HIP_TRY(hipSetDevice(g->devices[(size_t)k]));
		hipStream_t s = engine_stream(g->eng[(size_t)k]);
		void *t;
		size_t tb;
		if (presentation_rgba8)
		{
			G_TRY(ca3d_render_target(f, 0, &t, &tb));
			G_HIP_TRY(hipMemcpyAsync(presentation_rgba8 + (size_t)y0 * px_row * 4u, (char *)t + (size_t)y0 * px_row * 4u, (size_t)(y1 - y0) * px_row * 4u, hipMemcpyDeviceToHost, s));
		}
		if (light_rgba16f)
		{
			G_TRY(ca3d_render_target(f, 1, &t, &tb));
			G_HIP_TRY(hipMemcpyAsync((char *)light_rgba16f + (size_t)y0 * px_row * 8u, (char *)t + (size_t)y0 * px_row * 8u, (size_t)(y1 - y0) * px_row * 8u, hipMemcpyDeviceToHost, s));
		}
		if (depth_rg16f)
		{
			G_TRY(ca3d_render_target(f, 2, &t, &tb));
			G_HIP_TRY(hipMemcpyAsync((char *)depth_rg16f + (size_t)y0 * px_row * 4u, (char *)t + (size_t)y0 * px_row * 4u, (size_t)(y1 - y0) * px_row * 4u, hipMemcpyDeviceToHost, s));
		}
	}
	if (presentation_rgba8 || light_rgba16f || depth_rg16f)
		for (int k = 0; k < n; k++)
		{
			G_HIP_TRY(hipSetDevice(g->devices[(size_t)k]));
			G_HIP_TRY(hipStreamSynchronize(engine_stream(g->eng[(size_t)k])));
		}
	g->rw = width;
	g->rh = height;
	return CA3D_OK;
}
CA3D_API_CATCH

} // extern "C"
