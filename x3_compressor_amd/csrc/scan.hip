/*
 * scan.hip -- K1: the forward-window best-match scan (reference: backend.c:56-78).
 *
 * For every input position p the reference counts, over the candidates s in [p+1, p+W-33]
 * (backend.c:60,66), how many share a prefix of length i+1 with p, i < 32 (backend.c:67-73), and then
 * picks the longest prefix that repeats more than tc times for the largest workable tc <= T
 * (backend.c:76-78).  Without the dictionary filters that selection has the closed form
 *      m[p] = max{ i : count[i] > min(T, count[0]-1) }      (0 when T <= 0 or count[0] < 2),
 * and find_best_match(p) == 1 + max{ i <= m[p] : filters pass } (proven against the
 * reference by tests/test_oracle_golden.py::test_closed_form_m_equals_faithful_selection).  The scan
 * depends on the input bytes only, so it is evaluated for ALL positions at once; K2 applies the filters.
 *
 * v1 mapping (north star: one thread per candidate, coalesced window reads, LDS-staged look-ahead tile):
 *   workgroup = tile of X3_SCAN_TP consecutive positions; the tile's 32-byte look-aheads are staged in LDS
 *   as pre-aligned dwords; the candidate range of the tile is swept 256 candidates at a time, each thread
 *   holding its candidate's 32 bytes in registers (aligned dword loads + funnel shift, coalesced) and
 *   comparing them with every position of the tile; per-wave  count[i] += popc(ballot(lcp > i)).
 */
#include "x3_kernels.h"

__device__ static __forceinline__ uint32_t scan_load_u32_unaligned(const uint8_t *b, uint64_t q)
{
	/* two aligned dwords + funnel shift: no reliance on unaligned vector-memory support */
	const uint32_t *w = (const uint32_t *)(b + (q & ~(uint64_t)3));
	uint32_t sh = (uint32_t)(q & 3) * 8;
	uint32_t lo = w[0], hi = w[1];
	return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
}

__device__ static void x3_scan_body(const X3ScanArgs &a)
{
	X3_LDS uint32_t look[X3_SCAN_TP][8];
	X3_LDS uint32_t cnt[X3_SCAN_TP][32];

	const X3Chunk ck = a.chunks[blockIdx.y];
	const uint32_t n = ck.len;
	const uint32_t p0 = blockIdx.x * X3_SCAN_TP;
	if (p0 >= n) return; /* whole workgroup leaves together */
	const uint8_t *b = a.bytes + ck.byte_off; /* byte_off is a multiple of 256 */
	const uint32_t tid = threadIdx.x, lane = x3_lane();
	const uint32_t npos = (n - p0 < X3_SCAN_TP) ? n - p0 : X3_SCAN_TP;
	const uint32_t nc = a.window > X3_MAXLEN + 1 ? a.window - X3_MAXLEN - 1 : 0; /* candidates per position: s - p in [1, nc] */

	for (uint32_t i = tid; i < X3_SCAN_TP * 8; i += X3_SCAN_THREADS)
		look[i >> 3][i & 7] = scan_load_u32_unaligned(b, (uint64_t)p0 + (i >> 3) + 4 * (i & 7));
	for (uint32_t i = tid; i < X3_SCAN_TP * 32; i += X3_SCAN_THREADS) (&cnt[0][0])[i] = 0;
	__syncthreads();

	if (nc > 0) {
		const uint64_t s_first = (uint64_t)p0 + 1, s_last = (uint64_t)p0 + npos - 1 + nc;
		for (uint64_t base = s_first; base <= s_last; base += X3_SCAN_THREADS) {
			const uint64_t s = base + tid;
			uint32_t cw[8];
			{
				const uint32_t *w = (const uint32_t *)(b + (s & ~(uint64_t)3));
				const uint32_t sh = (uint32_t)(s & 3) * 8;
				uint32_t prev = w[0];
#pragma unroll
				for (int k = 0; k < 8; k++) {
					uint32_t nxt = w[k + 1];
					cw[k] = sh ? (prev >> sh) | (nxt << (32 - sh)) : prev;
					prev = nxt;
				}
			}
			for (uint32_t j = 0; j < npos; j++) {
				const uint64_t p = (uint64_t)p0 + j;
				uint32_t lcp = 0;
				if (s > p && s <= p + nc) {
					lcp = 32;
#pragma unroll
					for (int k = 7; k >= 0; k--) {
						uint32_t x = cw[k] ^ look[j][k];
						if (x) lcp = 4 * k + ((uint32_t)x3_ctz32(x) >> 3);
					}
				}
				for (uint32_t i = 0; i < 32; i++) { /* wave-uniform trip count */
					uint64_t mask = x3_ballot(lcp > i);
					if (!mask) break;
					if (lane == 0) atomicAdd(&cnt[j][i], (uint32_t)x3_popc64(mask));
				}
			}
		}
	}
	__syncthreads();

	if (tid < npos) {
		const uint32_t *c = cnt[tid];
		uint32_t m = 0;
		if (a.max_match_count > 0 && c[0] >= 2) {
			uint32_t thr = c[0] - 1 < (uint32_t)a.max_match_count ? c[0] - 1 : (uint32_t)a.max_match_count;
			for (uint32_t i = 1; i < 32; i++)
				if (c[i] > thr) m = i;
		}
		a.m[ck.byte_off + p0 + tid] = (uint8_t)m;
	}
	if (a.counts && blockIdx.y == 0) {
		for (uint32_t i = tid; i < npos * 32; i += X3_SCAN_THREADS) a.counts[(uint64_t)p0 * 32 + i] = (&cnt[0][0])[i];
	}
}

#ifndef X3_EMU
__global__ void __launch_bounds__(X3_SCAN_THREADS) x3_scan_kernel(X3ScanArgs a) { x3_scan_body(a); }

extern "C" void x3k_launch_scan(const X3ScanArgs *a, uint32_t max_len, uint32_t nchunks, hipStream_t st)
{
	dim3 grid((max_len + X3_SCAN_TP - 1) / X3_SCAN_TP, nchunks);
	if (grid.x == 0) return;
	hipLaunchKernelGGL(x3_scan_kernel, grid, dim3(X3_SCAN_THREADS), 0, st, *a);
}
#else
static void scan_tramp(void *p) { x3_scan_body(*(const X3ScanArgs *)p); }
extern "C" void x3k_launch_scan(const X3ScanArgs *a, uint32_t max_len, uint32_t nchunks, void *)
{
	dim3 grid((max_len + X3_SCAN_TP - 1) / X3_SCAN_TP, nchunks);
	if (grid.x == 0) return;
	x3emu_launch(scan_tramp, (void *)a, grid, dim3(X3_SCAN_THREADS));
}
#endif
