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
	const u32 steps = argc > 1 ? atoi(argv[1]) : 4, G = 512;
	const size_t words = (size_t)G / 32 * G * G;
	std::vector<u32> h(words);
	u32 x = 12345;
	for (auto &w : h) { x = x * 1664525u + 1013904223u; w = x; }
	u32 *b0, *b1, *status;
	unsigned long long *mail;
	const size_t mail_bytes = 2u * 256u * 4u * 512u * 8u;
	CK(hipMalloc(&b0, words * 4)); CK(hipMalloc(&b1, words * 4)); CK(hipMalloc(&mail, mail_bytes)); CK(hipMalloc(&status, (4 + 256) * 4));
	CK(hipMemcpy(b0, h.data(), words * 4, hipMemcpyHostToDevice));
	CK(hipMemset(mail, 0, mail_bytes)); CK(hipMemset(status, 0, (4 + 256) * 4));
	u32 *hflag;
	CK(hipHostMalloc((void **)&hflag, 16, hipHostMallocDefault));
	*hflag = 0;
	const int launches = argc > 2 ? atoi(argv[2]) : 3;
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	u32 epoch = 0, cur = 0;
	u32 *buf[2] = {b0, b1};
	for (int l = 0; l < launches; l++)
	{
		ResidentArgs a{buf[cur], buf[(cur + steps) & 1], buf[(cur + steps + 1) & 1], mail, status, hflag, steps, epoch, 5000000u};
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL((ca_resident_vn<0xFF, 0x0A>), dim3(256), dim3(512), 0, 0, a);
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
		if (st[0])
		{
			for (int tz = 0; tz < 16; tz++) { for (int ty = 0; ty < 16; ty++) printf("%4u", st[4 + tz * 16 + ty]); printf("\n"); }
			break;
		}
	}
	return 0;
}
