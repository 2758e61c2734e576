/*
 * scan2.hip -- K1 v2: the forward-window best-match scan without the O(N*W) sweep (reference: backend.c:56-78).
 *
 * What K2 needs from the scan is one number per position (scan.hip header):
 *      m[p] = max{ i : count_i(p) >= K(p) },   K(p) = min(T+1, count_0(p)),   (0 when T <= 0 or count_0(p) < 2)
 * where count_i(p) = #{ s in [p+1, p+W-33] : bytes p..p+i equal bytes s..s+i } (backend.c:62-74).
 * count_i(p) >= K  <=>  the K-th NEXT OCCURRENCE of the (i+1)-gram at p lies inside the window.  So:
 *   1. sort all positions of the (zero padded) batch by their 1-, 2-, 3- and 4-gram with a stable radix sort: inside a class
 *      the positions are ascending, and R_l[p] is p's index in list S_l;
 *   2. levels i = 0..3 are O(1): look at S_{i+1}[R_{i+1}[p] + K] -- same gram and still <= p+W-33 ?  (count_0 itself is only
 *      needed when it is <= T: a binary search over at most T+1 list entries);
 *   3. only positions whose 4-gram already repeats K times in their window ("active") go deeper: one wavefront per active
 *      position sweeps the candidates of its 4-gram class that lie in the window -- ONE LANE PER CANDIDATE, coalesced reads of
 *      the class list, a 32-byte look-ahead of p broadcast from LDS -- and builds count_4..31 with ballot/popcount.
 * The result is exactly the reference's (verified against the brute-force kernel of scan.hip and the oracle); the cost no
 * longer depends on W.  The padding zeros of every chunk are ordinary positions of the sort, and a chunk's window never reaches
 * the next chunk's bytes (slots are W + X3_PAD_EXTRA apart), so one global sort serves a whole batch.
 */
#include "x3_host.h"

#include <vector>

#define NONE32 0xFFFFFFFFu

__device__ static __forceinline__ uint32_t load_gram(const uint8_t *b, uint64_t q, uint32_t l)
{
	/* the l (1..4) bytes at q as an integer: aligned dwords + funnel shift, masked */
	const uint32_t *w = (const uint32_t *)(b + (q & ~(uint64_t)3));
	const uint32_t sh = (uint32_t)(q & 3) * 8;
	const uint32_t lo = w[0], hi = w[1];
	const uint32_t v = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
	return l >= 4 ? v : v & ((1u << (8 * l)) - 1);
}

struct X3WalkArgs {
	const uint8_t *bytes;
	const uint32_t *S4, *R4;     /* positions sorted by 4-gram; inverse */
	const uint32_t *active;      /* positions to walk */
	const uint32_t *active_k;    /* their K */
	const uint32_t *nactive;
	uint8_t *m;
	uint32_t total;              /* sorted entries */
	uint32_t ncand;              /* W - 33 */
};

#define X3_WALK_WAVES 4

__device__ static void x3_walk_body(const X3WalkArgs &a)
{
	X3_LDS uint32_t look[X3_WALK_WAVES][8];
	const uint32_t lane = x3_lane(), wv = threadIdx.x / X3_WAVE;
	const uint32_t idx = blockIdx.x * X3_WALK_WAVES + wv;
	const bool live = idx < *a.nactive; /* wave-uniform; no workgroup barrier below */
	if (!live) return;
	const uint32_t p = a.active[idx], K = a.active_k[idx];
	const uint8_t *b = a.bytes;
	if (lane < 8) look[wv][lane] = load_gram(b, (uint64_t)p + 4 * lane, 4); /* the look-ahead tile of this position */
	x3_wave_sync();
	const uint32_t g = look[wv][0];
	const uint64_t wend = (uint64_t)p + a.ncand;
	uint32_t cnt[28]; /* count_4 .. count_31 (wave-uniform) */
#pragma unroll
	for (int i = 0; i < 28; i++) cnt[i] = 0;
	uint32_t done = 0;
	for (uint32_t base = a.R4[p] + 1; base < a.total && !done; base += X3_WAVE) {
		const uint32_t j = base + lane;
		uint32_t s = j < a.total ? a.S4[j] : NONE32;
		uint32_t lcp = 0;
		bool inwin = false;
		if (s != NONE32 && s <= wend) {
			if (load_gram(b, s, 4) == g) {
				inwin = true;
				lcp = 32;
#pragma unroll
				for (int k = 7; k >= 1; k--) {
					const uint32_t x = load_gram(b, (uint64_t)s + 4 * k, 4) ^ look[wv][k];
					if (x) lcp = 4 * k + ((uint32_t)x3_ctz32(x) >> 3);
				}
			}
		}
		const uint64_t inm = x3_ballot(inwin);
		/* the class list is ascending inside the class: the first lane that is out of the class/window ends the walk */
		if (inm != ~(uint64_t)0) {
			const uint32_t first_out = inm == 0 ? 0 : (uint32_t)x3_ctz64(~inm);
			if (lane >= first_out) lcp = 0;
			done = 1;
		}
#pragma unroll
		for (int i = 0; i < 28; i++) { /* wave-uniform early exit */
			const uint64_t mk = x3_ballot(lcp > (uint32_t)(i + 4));
			if (!mk) break;
			cnt[i] += (uint32_t)x3_popc64(mk);
		}
		if (cnt[27] >= K) done = 1; /* count_31 >= K: nothing can be longer */
	}
	uint32_t m = 3; /* count_3 >= K is what made the position active */
#pragma unroll
	for (int i = 0; i < 28; i++) if (cnt[i] >= K) m = (uint32_t)i + 4;
	if (lane == 0) a.m[p] = (uint8_t)m;
}

#ifndef X3_EMU
__global__ void __launch_bounds__(X3_WAVE *X3_WALK_WAVES) x3_walk_kernel(X3WalkArgs a) { x3_walk_body(a); }
static void launch_walk(const X3WalkArgs &a, uint32_t nact_upper, hipStream_t st)
{
	if (!nact_upper) return;
	hipLaunchKernelGGL(x3_walk_kernel, dim3((nact_upper + X3_WALK_WAVES - 1) / X3_WALK_WAVES), dim3(X3_WAVE * X3_WALK_WAVES), 0, st, a);
}
#else
static void walk_tramp(void *p) { x3_walk_body(*(const X3WalkArgs *)p); }
static void launch_walk(const X3WalkArgs &a, uint32_t nact_upper, hipStream_t)
{
	if (!nact_upper) return;
	x3emu_launch(walk_tramp, (void *)&a, dim3((nact_upper + X3_WALK_WAVES - 1) / X3_WALK_WAVES), dim3(X3_WAVE * X3_WALK_WAVES));
}
#endif

/* m[] for every position of every chunk.  d_bytes/d_m use the padded layout (X3Chunk::byte_off), `total` = padded bytes. */
int x3_scan_v2_run(X3Scan2Bufs &B, DevBuf &tmp, hipStream_t st, int nchunks, const X3Chunk *h_chunks, const X3Chunk *d_chunks,
                   const uint8_t *d_bytes, uint8_t *d_m, uint64_t total, uint32_t window, int32_t T)
{
	const uint32_t nc = (uint32_t)nchunks;
	if (total >= 0xFFFFFF00ull) return X3H_E_ARG;
	const size_t P = (size_t)total;
	const uint32_t ncand = window > X3_MAXLEN + 1 ? window - X3_MAXLEN - 1 : 0;
	uint64_t nsum = 0;
	std::vector<uint32_t> po(nc + 1); /* real (unpadded) positions, for the per-position kernels */
	for (uint32_t c = 0; c < nc; c++) { po[c] = (uint32_t)nsum; nsum += h_chunks[c].len; }
	po[nc] = (uint32_t)nsum;
	const size_t N = (size_t)nsum;
	if (T <= 0 || ncand == 0 || N == 0) { /* backend.c:76,99: no threshold can be met -> length 1 everywhere */
		HIPCHK(hipMemsetAsync(d_m, 0, P, st));
		return X3H_OK;
	}
	for (int i = 0; i < 12; i++) CHK(B.a[i].reserve((P + 8) * 4));
	CHK(B.misc.reserve((size_t)(nc + 1) * 4 + 64));
	uint32_t *S[5], *R[5];
	for (int l = 1; l <= 4; l++) { S[l] = B.a[2 * (l - 1)].as<uint32_t>(); R[l] = B.a[2 * (l - 1) + 1].as<uint32_t>(); }
	uint32_t *keys = B.a[8].as<uint32_t>(), *iota = B.a[9].as<uint32_t>(), *ks = B.a[10].as<uint32_t>(), *act = B.a[11].as<uint32_t>();
	uint32_t *d_po = B.misc.as<uint32_t>(), *d_nact = d_po + (nc + 1);
	HIPCHK(hipMemcpyAsync(d_po, po.data(), (nc + 1) * 4, hipMemcpyHostToDevice, st));
	HIPCHK(hipMemsetAsync(d_nact, 0, 4, st));

	/* ---- 1. positions sorted by l-gram, l = 1..4 (stable: ascending positions inside a class), and the inverse ---- */
	x3_foreach(P, st, X3_LAMBDA(size_t q) { iota[q] = (uint32_t)q; });
	for (uint32_t l = 1; l <= 4; l++) {
		uint32_t *Sl = S[l], *Rl = R[l];
		x3_foreach(P, st, X3_LAMBDA(size_t q) { keys[q] = load_gram(d_bytes, q, l); });
		CHK(x3p_sort_pairs(tmp, keys, ks, iota, Sl, P, 8 * (int)l, st));
		x3_foreach(P, st, X3_LAMBDA(size_t j) { Rl[Sl[j]] = (uint32_t)j; });
	}

	/* ---- 2. per position: K, then levels 0..3 by K-th-next-occurrence lookups; deeper candidates are queued ---- */
	const uint32_t *S1 = S[1], *R1 = R[1], *S2 = S[2], *R2 = R[2], *S3 = S[3], *R3 = R[3], *S4 = S[4], *R4 = R[4];
	uint32_t *act_k = ks; /* the sorted-key scratch is dead now */
	const uint32_t Tu = (uint32_t)T, Pn = (uint32_t)P;
	x3_foreach(N, st, X3_LAMBDA(size_t gi) {
		uint32_t lo = 0, hi = nc; /* chunk of this position */
		while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (d_po[mid] <= (uint32_t)gi) lo = mid; else hi = mid; }
		const uint64_t p64 = d_chunks[lo].byte_off + ((uint32_t)gi - d_po[lo]);
		const uint32_t p = (uint32_t)p64;
		const uint64_t wend = p64 + ncand; /* last candidate position (backend.c:66: s < p + W - 32) */
		const uint32_t b0 = d_bytes[p];
		/* K = min(T+1, count_0) */
		const uint32_t j1 = R1[p];
		uint32_t K;
		{
			const uint64_t u = (uint64_t)j1 + Tu + 1;
			if (u < Pn && S1[u] <= wend && d_bytes[S1[u]] == b0) K = Tu + 1;
			else { /* count_0 <= T: largest u in [j1, j1+T] still in class and window */
				uint32_t a = 0, bnd = Tu; /* predicate true at a */
				if ((uint64_t)j1 + bnd >= Pn) bnd = Pn - 1 - j1;
				while (a < bnd) {
					const uint32_t mid = (a + bnd + 1) >> 1;
					const uint32_t s = S1[j1 + mid];
					if (s <= wend && d_bytes[s] == b0) a = mid; else bnd = mid - 1;
				}
				K = a; /* == count_0 */
			}
		}
		uint32_t m = 0;
		bool deeper = false;
		if (K >= 2) {
			const uint32_t g4 = load_gram(d_bytes, p, 4);
			uint64_t u = (uint64_t)R2[p] + K;
			if (u < Pn && S2[u] <= wend && load_gram(d_bytes, S2[u], 2) == (g4 & 0xFFFFu)) {
				m = 1;
				u = (uint64_t)R3[p] + K;
				if (u < Pn && S3[u] <= wend && load_gram(d_bytes, S3[u], 3) == (g4 & 0xFFFFFFu)) {
					m = 2;
					u = (uint64_t)R4[p] + K;
					if (u < Pn && S4[u] <= wend && load_gram(d_bytes, S4[u], 4) == g4) { m = 3; deeper = true; }
				}
			}
		}
		d_m[p] = (uint8_t)m;
		if (deeper) {
			const uint32_t slot = atomicAdd(d_nact, 1u);
			act[slot] = p;
			act_k[slot] = K;
		}
	});

	/* ---- 3. active positions: one wavefront each over the in-window candidates of its 4-gram class ---- */
	uint32_t nact = 0;
	HIPCHK(hipMemcpyAsync(&nact, d_nact, 4, hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	X3WalkArgs wa;
	wa.bytes = d_bytes; wa.S4 = S4; wa.R4 = R4; wa.active = act; wa.active_k = act_k; wa.nactive = d_nact; wa.m = d_m;
	wa.total = Pn; wa.ncand = ncand;
	launch_walk(wa, nact, st);
	HIPCHK(hipGetLastError());
	return X3H_OK;
}
