// how long do hipMalloc / hipFree / first touch take for large buffers?  (the CLI's first-call cliff)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
	double t0 = now(); hipFree(0); printf("runtime init %.1f ms\n", now() - t0);
	for (size_t gb : {1, 4, 16, 64}) {
		void *p = nullptr; t0 = now(); hipError_t e = hipMalloc(&p, gb << 30); double a = now() - t0;
		t0 = now(); hipMemset(p, 0, gb << 30); hipDeviceSynchronize(); double m = now() - t0;
		t0 = now(); hipMemset(p, 0, gb << 30); hipDeviceSynchronize(); double m2 = now() - t0;
		t0 = now(); hipFree(p); double f = now() - t0;
		printf("%3zu GiB: hipMalloc %.1f ms (%d), first memset %.1f ms, second %.1f ms, hipFree %.1f ms\n", gb, a, (int)e, m, m2, f);
	}
	std::vector<void *> v; t0 = now();
	for (int i = 0; i < 100; i++) { void *p; hipMalloc(&p, (size_t)256 << 20); v.push_back(p); }
	printf("100 x 256 MiB hipMalloc: %.1f ms\n", now() - t0);
	t0 = now(); for (void *p : v) hipFree(p); printf("100 x hipFree: %.1f ms\n", now() - t0);
	void *h; t0 = now(); hipHostMalloc(&h, (size_t)256 << 20, 0); printf("hipHostMalloc 256 MiB: %.1f ms\n", now() - t0);
	void *d; hipMalloc(&d, (size_t)256 << 20);
	t0 = now(); hipMemcpy(d, h, (size_t)256 << 20, hipMemcpyHostToDevice); printf("H2D 256 MiB pinned: %.1f ms\n", now() - t0);
	void *pg = malloc((size_t)256 << 20); memset(pg, 1, (size_t)256 << 20);
	t0 = now(); hipMemcpy(d, pg, (size_t)256 << 20, hipMemcpyHostToDevice); printf("H2D 256 MiB pageable: %.1f ms\n", now() - t0);
	t0 = now(); hipHostRegister(pg, (size_t)256 << 20, 0); printf("hipHostRegister 256 MiB: %.1f ms\n", now() - t0);
	t0 = now(); hipMemcpy(d, pg, (size_t)256 << 20, hipMemcpyHostToDevice); printf("H2D 256 MiB registered: %.1f ms\n", now() - t0);
	return 0;
}
