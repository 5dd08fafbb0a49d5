// Stand-alone driver of the resident class kernel (ca_resident_class_kernel.inc) with the clustered rule-set, for phase timing:
// built by tools/run_class_probe.py (which supplies ca_jit_rule.inc and the table macros). argv: steps launches grid(512|256) zgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ca_device_types.h"
namespace ca3d
{
namespace jit
{
#include "ca_bitops.inc"
#include "ca_jit_rule.inc"
#include "ca_bitslice.inc"
#include "ca_packed_roll_kernel.inc"
#include "ca_resident_kernel.inc"
#include "ca_resident_class_kernel.inc"
}
}
using namespace ca3d::jit;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv)
{
	const u32 steps = argc > 1 ? atoi(argv[1]) : 64;
	const int launches = argc > 2 ? atoi(argv[2]) : 3;
	const u32 G = argc > 3 ? atoi(argv[3]) : 256;
	const int zs = argc > 4 ? atoi(argv[4]) : 2;
	const size_t words = (size_t)G / 32 * G * G;
	std::vector<u32> h(words);
	u32 x = 12345;
	for (auto &w : h) { x = x * 1664525u + 1013904223u; w = x; }
	u32 *b0, *b1, *status;
	unsigned long long *mail;
	const size_t mail_bytes = 2u * 512u * 4u * 512u * 8u;
	CK(hipMalloc(&b0, words * 4)); CK(hipMalloc(&b1, words * 4)); CK(hipMalloc(&mail, mail_bytes)); CK(hipMalloc(&status, 1024 * 4));
	CK(hipMemcpy(b0, h.data(), words * 4, hipMemcpyHostToDevice));
	CK(hipMemset(mail, 0, mail_bytes)); CK(hipMemset(status, 0, 1024 * 4));
	u32 *hflag;
	CK(hipHostMalloc((void **)&hflag, 16, hipHostMallocDefault));
	*hflag = 0;
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	u32 epoch = 0, cur = 0;
	u32 *buf[2] = {b0, b1};
	for (int l = 0; l < launches; l++)
	{
		ResidentArgs a{buf[cur], buf[(cur + steps) & 1], buf[(cur + steps + 1) & 1], mail, status, hflag, steps, epoch, 5000000u};
		a.fault_tile = 0xFFFFFFFFu;
		CK(hipEventRecord(e0));
		if (G == 512) hipLaunchKernelGGL(ca3d_jit_resident_class, dim3(256), dim3(512), 0, 0, a);
		else if (zs == 2) hipLaunchKernelGGL(ca3d_jit_resident_class256, dim3(256), dim3(512), 0, 0, a);
		else hipLaunchKernelGGL(ca3d_jit_resident_class256_z1, dim3(256), dim3(256), 0, 0, a);
		CK(hipGetLastError());
		CK(hipEventRecord(e1));
		CK(hipDeviceSynchronize());
		epoch += steps;
		cur = (cur + steps) & 1;
		float ms;
		CK(hipEventElapsedTime(&ms, e0, e1));
		std::vector<u32> tt(700);
		CK(hipMemcpy(tt.data(), status, 700 * 4, hipMemcpyDeviceToHost));
		printf("launch %d steps %u: %.3f ms (%.3f us/step), status %u\n", l, steps, ms, ms * 1e3 / steps, tt[0]);
#ifdef CA3D_RES_STAMPS
		static const char *nm[6] = {"poll", "halo+barrier", "face pass", "row reads", "sweep", "to_image"};
		const int waves = (G == 512 ? 512 : 256 * zs) / 64;
		if (l == launches - 1) for (int w = 0; w < waves; w++) { printf("  wave %d cycles/step:", w); u32 tot = 0; for (int i = 0; i < 6; i++) { printf(" %s %u", nm[i], tt[600 + w * 8 + i]); tot += tt[600 + w * 8 + i]; } printf("  total %u\n", tot); }
#endif
		if (tt[0]) break;
	}
	return 0;
}
