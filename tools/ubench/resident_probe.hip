// Stand-alone driver of the resident kernel (ca_resident_kernel.inc) for bring-up: runs K steps on a random 512^3 state,
// prints the status word and, after a timeout, how far every tile got.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ca_bitops.inc"
#include "ca_resident_kernel.inc"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv)
{
	const int rows = argc > 3 ? atoi(argv[3]) : 32; // 32 / 16: rows per tile at 512^3; 256: the 256^3 form
	const bool deep = rows == 257; // 257: the two-steps-per-hand-off form at 256^3
	const u32 steps = argc > 1 ? atoi(argv[1]) : 4, G = rows == 256 || deep ? 256 : 512;
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
	const int launches = argc > 2 ? atoi(argv[2]) : 3;
	const int zs = argc > 4 ? atoi(argv[4]) : 1; // thread groups along z: 1 or 2
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	u32 epoch = 0, cur = 0;
	u32 *buf[2] = {b0, b1};
	for (int l = 0; l < launches; l++)
	{
		ResidentArgs a{buf[cur], buf[(cur + steps) & 1], buf[(cur + steps + 1) & 1], mail, status, hflag, steps, epoch, 5000000u};
		CK(hipEventRecord(e0));
		if (deep || rows == 34) { fprintf(stderr, "the deep (256^3) and staggered (512^3) forms were removed in round 5\n"); return 2; }
		else if (rows == 33) hipLaunchKernelGGL((ca_resident_vn_pair<0xFF, 0x0A>), dim3(256), dim3(512), 0, 0, a); // 33: the row-pair form
		else if (rows == 256 && zs == 2) hipLaunchKernelGGL((ca_resident_vn256<0xFF, 0x0A, 2>), dim3(256), dim3(512), 0, 0, a);
		else if (rows == 256) hipLaunchKernelGGL((ca_resident_vn256<0xFF, 0x0A, 1>), dim3(256), dim3(256), 0, 0, a);
		else if (rows == 16 && zs == 2) hipLaunchKernelGGL((ca_resident_vn<0xFF, 0x0A, 16, 2>), dim3(512), dim3(512), 0, 0, a);
		else if (rows == 16) hipLaunchKernelGGL((ca_resident_vn<0xFF, 0x0A, 16, 1>), dim3(512), dim3(256), 0, 0, a);
		else if (zs == 2) hipLaunchKernelGGL((ca_resident_vn<0xFF, 0x0A, 32, 2>), dim3(256), dim3(1024), 0, 0, a);
		else hipLaunchKernelGGL((ca_resident_vn<0xFF, 0x0A, 32, 1>), dim3(256), dim3(512), 0, 0, a);
		CK(hipGetLastError());
		CK(hipEventRecord(e1));
		CK(hipDeviceSynchronize());
		epoch += steps;
		cur = (cur + steps) & 1;
		float ms;
		CK(hipEventElapsedTime(&ms, e0, e1));
		std::vector<u32> st(4 + 256);
		CK(hipMemcpy(st.data(), status, st.size() * 4, hipMemcpyDeviceToHost));
		printf("launch %d steps %u: %.3f ms (%.3f us/step), status %u\n", l, steps, ms, ms * 1e3 / steps, st[0]);
#ifdef CA3D_RES_STAMPS
		if (deep)
		{
			std::vector<u32> tt(700);
			CK(hipMemcpy(tt.data(), status, 700 * 4, hipMemcpyDeviceToHost));
			static const char *nm[7] = {"poll", "halo+barrier", "ring rows", "column step A", "image1+barrier", "step B", "publish+image0"};
			if (l == launches - 1) for (int w = 0; w < 4 * zs; w++) { printf("  wave %d cycles/round:", w); u32 tot = 0; for (int i = 0; i < 7; i++) { printf(" %s %u", nm[i], tt[600 + w * 8 + i]); tot += tt[600 + w * 8 + i]; } printf("  total %u\n", tot); }
		}
		else
		{

			std::vector<u32> tt(700);
			CK(hipMemcpy(tt.data(), status, 616 * 4, hipMemcpyDeviceToHost));
			static const char *nm[7] = {"poll", "halo+barrier", "face pass", "ym/yp reads", "z faces+prefetch", "main pass", "to_image"};
			CK(hipMemcpy(tt.data(), status, 700 * 4, hipMemcpyDeviceToHost));
			static const char *nm34[7] = {"faces+reads+z", "main 1", "barrier", "main 2", "to_image", "poll+halo", "barrier"};
			if (rows == 34) { for (int w = 0; w < 8; w++) { printf("  wave %d cycles/step:", w); u32 tot = 0; for (int i = 0; i < 7; i++) { printf(" %s %u", nm34[i], tt[600 + w * 8 + i]); tot += tt[600 + w * 8 + i]; } printf("  total %u  stale polls per 1000 steps %u\n", tot, tt[680 + w]); } }
			else if (rows == 33) { for (int w = 0; w < 8; w++) { printf("  wave %d cycles/step:", w); u32 tot = 0; for (int i = 0; i < 7; i++) { printf(" %s %u", nm[i], tt[600 + w * 8 + i]); tot += tt[600 + w * 8 + i]; } printf("  total %u  (request answered %u) stale polls per 1000 steps %u\n", tot, tt[600 + w * 8 + 7], tt[680 + w]); } }
			else for (int w = 0; w < 2; w++) { printf("  wave %d cycles/step:", w * (rows >= 32 ? 4 : 2) * zs); u32 tot = 0; for (int i = 0; i < 7; i++) { printf(" %s %u", nm[i], tt[600 + w * 8 + i]); tot += tt[600 + w * 8 + i]; } printf("  total %u\n", tot); }
		}
#endif
		if (st[0])
		{
			for (int tz = 0; tz < 16; tz++) { for (int ty = 0; ty < 16; ty++) printf("%4u", st[4 + tz * 16 + ty]); printf("\n"); }
			break;
		}
	}
	return 0;
}
