/* tests/hip/div_check.hip -- the decoder's range / total (dec_div, decode.hip: double-precision reciprocal aimed 2^-43 low + a one-sided fix-up) against the
 * integer division of the device itself, on the operand ranges the decoder has (n <= 2^31, 0 < t < 2^28) and on the cases a reciprocal gets wrong first:
 * exact multiples, one below and one above them, t - 1 above them, for small, large, power-of-two and random divisors.  Built and run by
 * tests/test_gpu_parity.py::test_decoder_division_is_exact (hipcc, on the GPU box).  dec_div returns a wave-uniform value: one pair per wavefront at a time. */
#include "../../x3_compressor_amd/csrc/decode.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void div_check_kernel(const uint2 *pairs, uint32_t npairs, unsigned long long *bad, uint2 *first_bad)
{
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / X3_WAVE, nwaves = gridDim.x * blockDim.x / X3_WAVE;
	for (uint32_t i = wave; i < npairs; i += nwaves) {
		const uint32_t n = x3_uniform(pairs[i].x), t = x3_uniform(pairs[i].y);
		const uint32_t q = dec_div(n, t);
		if (q != n / t && x3_lane() == 0) { if (atomicAdd(bad, 1ull) == 0) *first_bad = pairs[i]; }
	}
}

int main()
{
	std::vector<uint2> h;
	uint64_t seed = 88172645463325252ull;
	auto rnd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
	auto add = [&](uint64_t n, uint64_t t) { if (t >= 1 && t < (1ull << 28) && n <= (1ull << 31)) { uint2 p; p.x = (uint32_t)n; p.y = (uint32_t)t; h.push_back(p); } };
	std::vector<uint64_t> ts;
	for (uint64_t t = 1; t <= 4096; t++) ts.push_back(t);
	for (int b = 1; b < 28; b++) for (int d = -2; d <= 2; d++) ts.push_back((1ull << b) + d);
	for (int i = 0; i < 20000; i++) ts.push_back(1 + rnd() % ((1ull << (1 + rnd() % 28)) - 1));
	for (uint64_t t : ts) {
		if (t < 1 || t >= (1ull << 28)) continue;
		const uint64_t kmax = (1ull << 31) / t;
		for (int j = 0; j < 12; j++) {
			const uint64_t k = j == 0 ? kmax : j == 1 ? 1 : j == 2 ? 2 : 1 + rnd() % (kmax ? kmax : 1);
			add(k * t, t); add(k * t - 1, t); add(k * t + 1, t); add(k * t + t - 1, t); add(k * t + t / 2, t);
		}
		add(1ull << 31, t); add((1ull << 31) - 1, t); add((1ull << 29) + 1, t); add(rnd() % ((1ull << 31) + 1), t);
	}
	uint2 *d_pairs, *d_first; unsigned long long *d_bad, bad = 0; uint2 first; first.x = first.y = 0;
	if (hipMalloc(&d_pairs, h.size() * sizeof(uint2)) != hipSuccess || hipMalloc(&d_bad, 8) != hipSuccess || hipMalloc(&d_first, sizeof(uint2)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 2; }
	hipMemcpy(d_pairs, h.data(), h.size() * sizeof(uint2), hipMemcpyHostToDevice);
	hipMemset(d_bad, 0, 8);
	hipLaunchKernelGGL(div_check_kernel, dim3(1024), dim3(256), 0, 0, d_pairs, (uint32_t)h.size(), d_bad, d_first);
	if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 2; }
	hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&first, d_first, sizeof(uint2), hipMemcpyDeviceToHost);
	printf("pairs %zu wrong %llu", h.size(), bad);
	if (bad) printf(" first: %u / %u", first.x, first.y);
	printf("\n");
	return bad ? 1 : 0;
}
