// experiment: does a long-running kernel of a few lone wavefronts on one HIP stream slow down a chain of short kernels on another stream?
// (the sliced schedule's mid-size batches: the feature kernels of slice k + 1 beside the coder segment of slice k take 2-6x their time)
//   hipcc -O3 --offload-arch=gfx950 -o overlap_probe tools/exp/overlap_probe.cpp && ./overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

// A: `blocks` lone wavefronts busy for `iters` iterations of a dependent scalar chain (no waits: they saturate their issue slots like the coder recurrence);
// mode 0: scalar ALU only, 1: + a scalar store per 8 iterations (s_store_dword, like the coder's state records), 2: + a scalar load per 8 iterations, 3: + a vector store
__global__ void __launch_bounds__(64) busy_kernel(uint32_t *buf, unsigned long long iters, int mode)
{
	uint32_t acc = __builtin_amdgcn_readfirstlane(blockIdx.x * 2654435761u + 1u);
	uint32_t *mine = buf + (size_t)blockIdx.x * 4096;
	for (unsigned long long i = 0; i < iters; i++) {
#pragma unroll
		for (int k = 0; k < 8; k++) { acc = acc * 1664525u + 1013904223u; acc ^= acc >> 13; }
		const uint32_t slot = (uint32_t)(i & 1023u) * 4u;
		if (mode == 1) asm volatile("s_store_dword %0, %1, %2" :: "s"(acc), "s"(mine), "s"(slot) : "memory");
		if (mode == 2) { uint32_t v; asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(mine), "s"(slot) : "memory"); acc += v; }
		if (mode == 3 && threadIdx.x == 0) mine[slot] = acc;
	}
	if (acc == 12345u) buf[0] = acc;
}
// B: a short latency-bound kernel, one workgroup per "stream": a pointer chase of `steps` dependent loads per wavefront + a store
__global__ void __launch_bounds__(256) chase_kernel(const uint32_t *next, uint32_t *out, uint32_t steps)
{
	uint32_t p = (blockIdx.x * 9973u + (threadIdx.x >> 6) * 31u) & 0xFFFFu;
	for (uint32_t i = 0; i < steps; i++) p = next[p];
	out[blockIdx.x * 256 + threadIdx.x] = p;
}

int main()
{
	const int nb = 40;
	uint32_t *busy_buf, *next, *out;
	hipMalloc(&busy_buf, (size_t)nb * 4096 * 4 + 64); hipMalloc(&next, 65536 * 4); hipMalloc(&out, nb * 256 * 4);
	std::vector<uint32_t> h(65536);
	for (uint32_t i = 0; i < 65536; i++) h[i] = (i * 40503u + 12345u) & 0xFFFFu;
	hipMemcpy(next, h.data(), 65536 * 4, hipMemcpyHostToDevice);
	hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	// calibrate: how many iterations make ~30 ms
	unsigned long long busy_iters = 100000;
	{ hipEventRecord(e0, sa); hipLaunchKernelGGL(busy_kernel, dim3(nb), dim3(64), 0, sa, busy_buf, busy_iters, 0); hipEventRecord(e1, sa); hipEventSynchronize(e1); float ms = 0; hipEventElapsedTime(&ms, e0, e1); busy_iters = (unsigned long long)(busy_iters * 30.0 / ms); printf("busy kernel: %llu iterations of 8 dependent multiply-adds = 30 ms\n", busy_iters); }
	for (int mode = -1; mode < 4; mode++) { // -1: nothing beside the chain
		for (int nk : { 1, 20 }) {
			float best = 1e9f;
			for (int rep = 0; rep < 5; rep++) {
				hipDeviceSynchronize();
				if (mode >= 0) hipLaunchKernelGGL(busy_kernel, dim3(nb), dim3(64), 0, sa, busy_buf, busy_iters, mode);
				hipEventRecord(e0, sb);
				for (int k = 0; k < nk; k++) hipLaunchKernelGGL(chase_kernel, dim3(nb), dim3(256), 0, sb, next, out, 400u * 20 / nk);
				hipEventRecord(e1, sb);
				hipEventSynchronize(e1);
				float ms = 0; hipEventElapsedTime(&ms, e0, e1);
				if (ms < best) best = ms;
				hipDeviceSynchronize();
			}
			printf("beside: %-28s  the same dependent loads as %2d kernel(s): %7.3f ms\n",
			       mode < 0 ? "nothing" : mode == 0 ? "40 busy waves (scalar ALU)" : mode == 1 ? "40 busy waves + s_store" : mode == 2 ? "40 busy waves + s_load" : "40 busy waves + vector store", nk, best);
		}
	}
	return 0;
}
