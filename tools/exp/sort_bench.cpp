// experiment: rocPRIM onesweep radix_sort_pairs (u32 key, u32 value) on gfx950 with the library's default configuration (gfx950 has no tuned
// entry: "unknown" = 256 threads x 16 items) against explicit ones.  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/sort_bench.cpp -o tools/exp/sort_bench
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>

template <class Config>
static double run(const char *name, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, unsigned bits)
{
	size_t need = 0;
	if (rocprim::radix_sort_pairs<Config>(nullptr, need, kin, kout, vin, vout, n, 0u, bits, 0) != hipSuccess) { printf("%s: size query failed\n", name); return 0; }
	void *tmp = nullptr;
	hipMalloc(&tmp, need);
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	float best = 1e9f;
	for (int it = 0; it < 4; it++) {
		hipEventRecord(a, 0);
		if (rocprim::radix_sort_pairs<Config>(tmp, need, kin, kout, vin, vout, n, 0u, bits, 0) != hipSuccess) { printf("%s: sort failed\n", name); return 0; }
		hipEventRecord(b, 0);
		hipEventSynchronize(b);
		float ms = 0; hipEventElapsedTime(&ms, a, b);
		if (ms < best) best = ms;
	}
	std::vector<uint32_t> h(1024);
	hipMemcpy(h.data(), kout + n / 2, 4096, hipMemcpyDeviceToHost);
	bool ok = true;
	for (int i = 1; i < 1024; i++) if ((h[i - 1] & ((1u << bits) - 1)) > (h[i] & ((1u << bits) - 1))) ok = false;
	printf("%-40s %2u bits: %7.3f ms  (%s)\n", name, bits, best, ok ? "sorted" : "NOT SORTED");
	hipFree(tmp);
	return best;
}

using namespace rocprim;
template <unsigned BS, unsigned IPT, unsigned RB> using cfg = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<BS, IPT>, kernel_config<BS, IPT>, RB, block_radix_rank_algorithm::match>>;

int main(int argc, char **argv)
{
	const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 130000000;
	uint32_t *kin, *kout, *vin, *vout;
	hipMalloc(&kin, n * 4); hipMalloc(&kout, n * 4); hipMalloc(&vin, n * 4); hipMalloc(&vout, n * 4);
	std::vector<uint32_t> h(n);
	std::mt19937 rng(1);
	for (size_t i = 0; i < n; i++) h[i] = rng();
	hipMemcpy(kin, h.data(), n * 4, hipMemcpyHostToDevice);
	hipMemcpy(vin, h.data(), n * 4, hipMemcpyHostToDevice);
	for (unsigned bits : {19u, 25u}) {
		run<default_config>("default (unknown arch: 256 x 16, 8 bits)", kin, kout, vin, vout, n, bits);
		run<cfg<1024, 16, 8>>("1024 x 16, 8 bits (the gfx942 entry)", kin, kout, vin, vout, n, bits);
		run<cfg<1024, 12, 8>>("1024 x 12, 8 bits", kin, kout, vin, vout, n, bits);
		run<cfg<512, 16, 8>>("512 x 16, 8 bits", kin, kout, vin, vout, n, bits);
		run<cfg<512, 24, 8>>("512 x 24, 8 bits", kin, kout, vin, vout, n, bits);
		run<cfg<1024, 8, 8>>("1024 x 8, 8 bits", kin, kout, vin, vout, n, bits);
		run<cfg<1024, 16, 7>>("1024 x 16, 7 bits", kin, kout, vin, vout, n, bits);
	}
	return 0;
}
