// Resident multi-step kernel (ca_resident_kernel.inc): ahead-of-time instantiation for the reference's start-up rule
// and the launcher. K steps of the packed von Neumann CA at 512^3 in one launch, state in registers, tile faces
// exchanged through tagged granules; see the device source for the design.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "ca3d_internal.h"

namespace ca3d
{
namespace
{
#include "ca_bitops.inc"

#include "ca_resident_kernel.inc"

constexpr int kDefaultS = 0xFF, kDefaultB = 0x0A; // von Neumann B1,3 / S0-6, canonical tables (ca_packed_vn.hip)

// A resident launch needs every one of its workgroups on a CU at the same time, and one workgroup fills most of a CU's
// LDS: two such launches running side by side (two engines of one process on different streams) would each get part of
// the chip and wait for tiles that cannot start. Resident launches on one device are therefore chained: a launch on another
// stream than the previous one waits for an event recorded on that stream (launches on the same stream are ordered by the
// stream itself and pay for no event).
struct DeviceChain
{
	std::mutex m;
	hipEvent_t ev[64] = {};
	hipStream_t last[64] = {};
	bool any[64] = {};
};
DeviceChain &chain()
{
	static DeviceChain c;
	return c;
}

template <typename F>
hipError_t chained_launch(hipStream_t stream, F launch)
{
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess) return e;
	if (dev < 0 || dev >= 64) return launch();
	DeviceChain &c = chain();
	std::lock_guard<std::mutex> lock(c.m);
	if (c.any[dev] && c.last[dev] != stream)
	{
		if (!c.ev[dev])
		{
			e = hipEventCreateWithFlags(&c.ev[dev], hipEventDisableTiming);
			if (e != hipSuccess) return e;
		}
		// after everything the other stream holds now, which includes its last resident launch
		e = hipEventRecord(c.ev[dev], c.last[dev]);
		if (e == hipSuccess) e = hipStreamWaitEvent(stream, c.ev[dev], 0);
		if (e != hipSuccess) return e;
	}
	e = launch();
	if (e != hipSuccess) return e;
	c.last[dev] = stream;
	c.any[dev] = true;
	return hipSuccess;
}

} // namespace

// A stream that is about to be destroyed (its work has been waited for) leaves the chain.
void resident_stream_retired(hipStream_t stream)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
	DeviceChain &c = chain();
	std::lock_guard<std::mutex> lock(c.m);
	if (c.any[dev] && c.last[dev] == stream) c.any[dev] = false;
}

// The rule is a pair of truth tables over the von Neumann count (as ca_packed_vn) and the grid is the one the tile
// geometry is built for.
bool resident_kernel_applies(const CanonRules &r, uint32_t G, int variant)
{
	if (G == 64u) return vn_rule_applies(r, variant); // the one-workgroup form (ca_resident_kernel.inc, resident64_run): the per-step kernel there is the rows kernel
	return (G == 512u || G == 256u) && vn_kernel_applies(r, G, variant);
}

bool resident_class_applies(const CanonRules &r, uint32_t G, int variant)
{
	return (G == 512u || G == 256u) && use_class_kernel(r, G, variant) && !vn_kernel_applies(r, G, variant) && roll_kernel_applies(r, G, variant);
}

// 256^3: two z groups per tile (512 threads, two waves per SIMD; CA3D_RC256_ZS=1 / 4 for the one- / four-group forms: 2.92 / 2.60 us per
// step against 2.26); 512^3: one
uint32_t resident_class_zsplit(uint32_t G)
{
	if (G != 256u) return 1u;
	const char *e = getenv("CA3D_RC256_ZS");
	const int v = e ? atoi(e) : 0;
	return v == 1 ? 1u : v == 4 ? 4u : 2u;
}

size_t resident_mail_bytes(uint32_t G, uint32_t rows)
{
	if (G == 256u) return 2u * (size_t)256u * 4u * 512u * sizeof(unsigned long long); // 256 tiles, faces of up to 512 words (sized for the widest face any 256^3 form publishes)
	const size_t tiles = (size_t)(G / rows) * (G / kResTileRows);
	return 2u * tiles * 4u * kResFaceWords * sizeof(unsigned long long);
}


int resident_slab_planes(const CanonRules &r, uint32_t G, uint32_t nplanes, int variant)
{
	if (G != 1024u || !vn_kernel_applies(r, G, variant) || nplanes % (uint32_t)kSlabTZ) return 0;
	const uint32_t pz = nplanes / (uint32_t)kSlabTZ;
	// even (the face pass runs whole waves); up to 24 planes a thread holds the rows either side of its own for every plane (128 VGPRs
	// at 1024 threads per workgroup), up to 36 — a share of a quarter of the grid with ghosts up to 16 planes deep; the one LDS image
	// of 34 rows x 36 planes is 153 of the CU's 160 KB — it reads them four planes at a time inside the pass (kLate)
	return pz >= 4u && pz <= (uint32_t)kSlabMaxPZ && pz % 2u == 0 ? (int)pz : 0;
}

namespace
{
// CUs a kernel on `stream` may be placed on: the device's, or fewer under a CU mask (hipExtStreamCreateWithCUMask)
uint32_t usable_cus(hipStream_t stream)
{
	int dev = 0, cus = 0;
	if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
	if (stream)
	{
		uint32_t mask[16] = {};
		if (hipExtStreamGetCUMask(stream, 16, mask) == hipSuccess)
		{
			uint32_t n = 0;
			for (uint32_t w : mask) n += (uint32_t)__builtin_popcount(w);
			if (n && n < (uint32_t)cus) cus = (int)n;
		}
		else (void)hipGetLastError();
	}
	return (uint32_t)cus;
}
} // namespace

bool resident_capacity(uint32_t G, uint32_t rows, uint32_t zsplit, int pair, void *jit_fn, hipStream_t stream, uint32_t *tiles, uint32_t *capacity)
{

	if (G == 64u) { *tiles = 1u; *capacity = usable_cus(stream) ? 1u : 0u; return *capacity != 0u; } // one workgroup: it waits for nobody
	if (pair) { rows = 32u; zsplit = 1u; } // the row-pair form: the tiles and the 512 threads of the 32-row form
	const uint32_t threads = (G == 256u ? 256u : 16u * rows) * zsplit;
	const bool z2 = zsplit == 2u;
	*tiles = G == 256u ? 256u : (G / rows) * (G / kResTileRows);
	*capacity = 0;
	int per_cu = 0;
	hipError_t e;
	if (jit_fn) e = hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (hipFunction_t)jit_fn, (int)threads, 0);
	else if (pair) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)ca_resident_vn_pair<kDefaultS, kDefaultB>, (int)threads, 0);
	else if (G == 256u) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, z2 ? (const void *)ca_resident_vn256<kDefaultS, kDefaultB, 2> : (const void *)ca_resident_vn256<kDefaultS, kDefaultB, 1>, (int)threads, 0);
	else if (rows == 16u) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, z2 ? (const void *)ca_resident_vn<kDefaultS, kDefaultB, 16, 2> : (const void *)ca_resident_vn<kDefaultS, kDefaultB, 16, 1>, (int)threads, 0);
	else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, z2 ? (const void *)ca_resident_vn<kDefaultS, kDefaultB, 32, 2> : (const void *)ca_resident_vn<kDefaultS, kDefaultB, 32, 1>, (int)threads, 0);
	const uint32_t cus = usable_cus(stream);
	if (e != hipSuccess || per_cu <= 0 || cus == 0) { (void)hipGetLastError(); return false; }
	*capacity = (uint32_t)per_cu * cus;
	return true;
}

bool resident_slab_capacity(void *fn, hipStream_t stream, uint32_t *tiles, uint32_t *capacity)
{
	*tiles = (uint32_t)(kSlabTY * kSlabTZ);
	*capacity = 0;
	int per_cu = 0;
	const hipError_t e = hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (hipFunction_t)fn, kSlabThreads, 0);
	const uint32_t cus = usable_cus(stream);
	if (e != hipSuccess || per_cu <= 0 || cus == 0) { (void)hipGetLastError(); return false; }
	*capacity = (uint32_t)per_cu * cus;
	return true;
}

size_t resident_slab_mail_bytes() { return 2u * (size_t)(kSlabTY * kSlabTZ) * 4u * kSlabFace * sizeof(unsigned long long); }

hipError_t launch_resident_slab(const ResidentSlabLaunch &l, hipStream_t stream)
{
	ResidentSlabArgs a;
	a.in = l.in;
	a.out = l.out;
	a.mail = l.mail;
	a.status = l.status;
	a.host_flag = l.host_flag;
	a.steps = l.steps;
	a.epoch0 = l.epoch0;
	a.timeout_ticks = l.timeout_ticks;
	a.dead_plane = l.dead_plane;
	void *args[] = {(void *)&a};
	return chained_launch(stream, [&]() { return hipModuleLaunchKernel((hipFunction_t)l.fn, kSlabTY * kSlabTZ, 1, 1, kSlabThreads, 1, 1, 0, stream, args, nullptr); });
}

hipError_t launch_resident(const ResidentLaunch &l, hipStream_t stream)
{
	ResidentArgs a;
	a.in = l.in;
	a.out_last = l.out_last;
	a.out_prev = l.out_prev;
	a.mail = l.mail;
	a.status = l.status;
	a.host_flag = l.host_flag;
	a.steps = l.steps;
	a.epoch0 = l.epoch0;
	a.timeout_ticks = l.timeout_ticks;
	a.fault_tile = l.fault_tile;
	const bool z2 = l.zsplit == 2u;
	if (l.zsplit != 1u && l.zsplit != 2u && !(l.zsplit == 4u && l.G == 256u && l.jit_fn)) return hipErrorInvalidValue;
	if (l.G == 64u)
	{
		// the whole grid in one workgroup of 1024 threads
		if (l.jit_fn)
		{
			void *args[] = {(void *)&a};
			return chained_launch(stream, [&]() { return hipModuleLaunchKernel((hipFunction_t)l.jit_fn, 1, 1, 1, 1024, 1, 1, 0, stream, args, nullptr); });
		}
		if (l.lut_s != (u32)kDefaultS || l.lut_b != (u32)kDefaultB) return hipErrorInvalidValue;
		return chained_launch(stream, [&]() {
			hipLaunchKernelGGL((ca_resident_vn64<kDefaultS, kDefaultB, 4>), dim3(1), dim3(1024), 0, stream, a);
			return hipGetLastError();
		});
	}
	if (l.G == 256u)
	{
		// 8 x 32 tiles of 32 rows x 8 planes, 256 (x 2 with the z split) threads each (ca_resident_kernel.inc: CW = 8, PZ = 8)
		const u32 threads = 256u * l.zsplit;
		if (l.jit_fn)
		{
			void *args[] = {(void *)&a};
			return chained_launch(stream, [&]() { return hipModuleLaunchKernel((hipFunction_t)l.jit_fn, 256, 1, 1, threads, 1, 1, 0, stream, args, nullptr); });
		}
		if (l.lut_s != (u32)kDefaultS || l.lut_b != (u32)kDefaultB) return hipErrorInvalidValue;
		return chained_launch(stream, [&]() {
			if (z2) hipLaunchKernelGGL((ca_resident_vn256<kDefaultS, kDefaultB, 2>), dim3(256), dim3(threads), 0, stream, a);
			else hipLaunchKernelGGL((ca_resident_vn256<kDefaultS, kDefaultB, 1>), dim3(256), dim3(threads), 0, stream, a);
			return hipGetLastError();
		});
	}
	if (l.pair)
	{
		if (l.G != 512u) return hipErrorInvalidValue;
		if (l.jit_fn)
		{
			void *args[] = {(void *)&a};
			return chained_launch(stream, [&]() { return hipModuleLaunchKernel((hipFunction_t)l.jit_fn, 256, 1, 1, 512, 1, 1, 0, stream, args, nullptr); });
		}
		if (l.lut_s != (u32)kDefaultS || l.lut_b != (u32)kDefaultB) return hipErrorInvalidValue;
		return chained_launch(stream, [&]() {
			hipLaunchKernelGGL((ca_resident_vn_pair<kDefaultS, kDefaultB>), dim3(256), dim3(512), 0, stream, a);
			return hipGetLastError();
		});
	}
	if (l.G != 512u || (l.rows != 16u && l.rows != 32u)) return hipErrorInvalidValue;
	const u32 tiles = (l.G / l.rows) * (l.G / kResTileRows), threads = 16u * l.rows * l.zsplit;
	if (l.jit_fn)
	{
		void *args[] = {(void *)&a};
		return chained_launch(stream, [&]() { return hipModuleLaunchKernel((hipFunction_t)l.jit_fn, tiles, 1, 1, threads, 1, 1, 0, stream, args, nullptr); });
	}
	if (l.lut_s != (u32)kDefaultS || l.lut_b != (u32)kDefaultB) return hipErrorInvalidValue;
	return chained_launch(stream, [&]() {
		if (l.rows == 16u && z2) hipLaunchKernelGGL((ca_resident_vn<kDefaultS, kDefaultB, 16, 2>), dim3(tiles), dim3(threads), 0, stream, a);
		else if (l.rows == 16u) hipLaunchKernelGGL((ca_resident_vn<kDefaultS, kDefaultB, 16, 1>), dim3(tiles), dim3(threads), 0, stream, a);
		else if (z2) hipLaunchKernelGGL((ca_resident_vn<kDefaultS, kDefaultB, 32, 2>), dim3(tiles), dim3(threads), 0, stream, a);
		else hipLaunchKernelGGL((ca_resident_vn<kDefaultS, kDefaultB, 32, 1>), dim3(tiles), dim3(threads), 0, stream, a);
		return hipGetLastError();
	});
}

} // namespace ca3d
