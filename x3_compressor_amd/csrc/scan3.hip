/*
 * scan3.hip -- K1 for batches of MANY chunks: the sorted n-gram lists of scan2.hip, built by ONE WORKGROUP PER CHUNK
 * (reference: backend.c:56-78, the same closed form m[p] = max{ i : count_i(p) >= K(p) } as scan2.hip).
 *
 * scan2.hip sorts every position of the whole batch with one chip-wide LSD radix sort and then, level by level, stores one byte
 * per passing position at a random address: 0.7 KB of HBM traffic per input byte on a 1024-chunk batch.  Chunks are independent
 * streams, so here the sort is SEGMENTED: a 1024-thread workgroup owns one chunk from the first key to its m[] bytes --
 *   phase 0  keys: element q (an END position) = the four bytes ending at q, written as (key, position) pairs; one byte histogram in
 *            LDS serves all four passes (the digit of pass l is byte q-(l-1): the same multiset of bytes, the few bytes that fall
 *            off either end are padding zeros on both sides);
 *   pass l   one stable counting-sort pass on key byte l-1 in tiles of 4096 elements: per wavefront a ballot match gives every lane its
 *            rank among the lanes with the same digit, a [digit][wave] counter table in LDS and one workgroup scan turn that into the
 *            tile-sorted order, the tile is staged in LDS and leaves as coalesced runs, one run per digit;
 *   level l  on the list just written (ordered by the l-gram starting at p = q-(l-1), positions ascending inside a class):
 *            count_{l-1}(p) >= K  <=>  entry j+K has the same l-gram and lies inside p's window -- scan2.hip's O(1) test.  A level that
 *            passes adds one to a 2-BIT COUNTER of the position IN LDS (levels pass in order: the counter IS m for m <= 3), so m[] leaves
 *            the chip once, as coalesced bytes, instead of as one scattered byte store per level.  The W padding zeros behind a chunk are
 *            COUNTED where a window reaches them (seg_cpad), never sorted;
 *   list 4   (4-gram classes) and the positions whose 4-gram still repeats K times go to the walk kernel / the dense-class refinement of
 *            scan2.hip unchanged (same list layout: chunk c's list occupies the entries of its slot of the padded layout).
 * HBM traffic: 16 B per element and pass + the level reads, all of it sequential or in digit runs.
 */
#include "x3_host.h"

#include <stdlib.h>

#define X3_SEG_WAVES (X3_SEG_THREADS / X3_WAVE)
#define X3_SEG_E     4u                            /* elements per thread and tile */
#define X3_SEG_TILE  (X3_SEG_THREADS * X3_SEG_E)

__device__ static __forceinline__ uint32_t seg_bswap(uint32_t v) { return __builtin_bswap32(v); }

/* mask of the lanes (among `valid` ones) that hold the same 8-bit digit as this lane: one ballot per digit bit */
__device__ static __forceinline__ uint64_t seg_match8(uint32_t d, bool valid)
{
	uint64_t mask = x3_ballot(valid);
#pragma unroll
	for (uint32_t b = 0; b < 8; b++) {
		const bool bit = (d >> b) & 1u;
		const uint64_t bal = x3_ballot(bit);
		mask &= bit ? bal : ~bal;
	}
	return mask;
}

/* Padding.  The W zero bytes behind a chunk take part in its windows (x3.c:579,590), but they are not sorted: the lists hold the END positions
 * q < n + 3 only (every gram that starts inside the data).  An end position e >= n + 3 ends an all-zero gram of any length <= 4, so a query
 * whose gram is all zero meets   cpad = #{ e in [n + 3, q + ncand] }   further occurrences behind everything the list shows it (they are the
 * class's largest positions): "the K-th next occurrence lies inside the window"  <=>  cpad >= K, or entry j + (K - cpad) is of the class and
 * inside the window. */
__device__ static __forceinline__ uint32_t seg_cpad(uint32_t qrel, uint32_t ncand, uint32_t n) { const uint32_t we = qrel + ncand; return we >= n + 3u ? we - n - 2u : 0u; }

/* the test of level lv (gram length lv) for entry j = `it` of list lv; `la` = entry j + T + 1 of the same list (position 0xFFFFFFFF when
 * there is none); list = list lv as (key, position) pairs.  Level 1 fixes K = min(T+1, count_0) (positions with a smaller K are marked in
 * `rbits`, their K goes to kexact); levels 2, 3 add one to the position's 2-bit counter when count_{lv-1} >= K. */
__device__ static __forceinline__ void seg_level(uint32_t lv, const uint2 it, const uint2 la, uint32_t j, const uint2 *list, uint32_t L, uint32_t n,
                                                 uint32_t base, uint32_t ncand, uint32_t Tu, uint32_t *mfield, uint32_t *rbits, uint32_t *kexact)
{
	if (lv == 1u) {
		const uint32_t prel = it.y - base;
		if (prel >= n) return; /* a padding start: an occurrence, never a query */
		const uint32_t kj = it.x & 0xFFu, wend = prel + ncand;
		const uint32_t cpad = kj == 0u ? seg_cpad(prel, ncand, n) : 0u;
		/* is the (T+1)-th next occurrence of the byte inside the window?  then K = T+1 */
		if (cpad == 0u) { if ((la.x & 0xFFu) == kj && la.y - base <= wend) return; }
		else {
			if (cpad > Tu) return;
			const uint32_t u = j + (Tu + 1u - cpad);
			if (u < L) { const uint2 eu = list[u]; if ((eu.x & 0xFFu) == kj && eu.y - base <= wend) return; }
		}
		uint32_t lo = 0, bnd = Tu - cpad; /* count the listed ones: binary search, predicate true at lo */
		if (j + bnd >= L) bnd = L - 1u - j;
		while (lo < bnd) {
			const uint32_t mid = (lo + bnd + 1u) >> 1;
			const uint2 em = list[j + mid];
			if ((em.x & 0xFFu) == kj && em.y - base <= wend) lo = mid; else bnd = mid - 1u;
		}
		kexact[it.y] = lo + cpad; /* == count_0 */
		atomicOr(&rbits[prel >> 5], 1u << (prel & 31u));
		return;
	}
	const uint32_t back = lv - 1u, msk = (1u << (8u * lv)) - 1u; /* lv = 2, 3 */
	const uint32_t qrel = it.y - base;
	if (qrel < back || qrel - back >= n) return; /* the gram starts before the chunk or in its padding */
	const uint32_t prel = qrel - back, wend = qrel + ncand; /* both sides of the window test carry the same +back */
	const uint32_t cpad = (it.x & msk) == 0u ? seg_cpad(qrel, ncand, n) : 0u;
	/* first with K = T+1, from the list alone (K <= T+1: if even the (T+1)-th next occurrence is inside the window the level passes);
	 * only if that fails can a smaller K matter, and only marked positions have one */
	bool pass;
	if (cpad == 0u) pass = ((la.x ^ it.x) & msk) == 0u && la.y - base <= wend;
	else if (cpad > Tu) pass = true;
	else {
		const uint32_t u = j + (Tu + 1u - cpad);
		pass = false;
		if (u < L) { const uint2 eu = list[u]; pass = ((eu.x ^ it.x) & msk) == 0u && eu.y - base <= wend; }
	}
	if (!pass && ((rbits[prel >> 5] >> (prel & 31u)) & 1u)) {
		const uint32_t K = kexact[base + prel];
		if (K >= 2u) { /* K < 2: count_0 < 2, nothing repeats */
			if (cpad >= K) pass = true;
			else if (j + (K - cpad) < L) {
				const uint2 eu = list[j + (K - cpad)];
				pass = ((eu.x ^ it.x) & msk) == 0u && eu.y - base <= wend;
			}
		}
	}
	if (pass) atomicAdd(&mfield[prel >> 4], 1u << (2u * (prel & 15u)));
}

/* BIG = false: chunks of at most X3_SEG_MAXLEN bytes, their level counters and small-K marks live in LDS; BIG = true: longer chunks, both
 * live in global memory (a.gmf / a.rare, zeroed by the caller; the atomics execute in L2) */
template <bool BIG>
__device__ static void x3_segscan_body(const X3SegArgs &a)
{
	X3_LDS uint32_t mfield_l[BIG ? 1 : X3_SEG_MAXLEN / 16]; /* 2 bits per position: levels 1..3 passed so far */
	X3_LDS uint32_t rbits_l[BIG ? 1 : X3_SEG_MAXLEN / 32];  /* positions whose K is below T+1 */
	X3_LDS uint32_t cnt[256 * X3_SEG_WAVES];           /* [digit][wave]: counts, then exclusive prefix in tile-sorted order */
	X3_LDS uint2 stage[X3_SEG_TILE];                   /* the tile in sorted order (phase 0: eight copies of the byte histogram) */
	X3_LDS uint32_t bbase[256], bcur[256], wtot[X3_SEG_WAVES];

	const uint32_t tid = threadIdx.x, lane = x3_lane(), wv = tid / X3_WAVE;
	const X3Chunk ck = a.chunks[blockIdx.x];
	const uint32_t n = ck.len, base = (uint32_t)ck.byte_off;
	/* elements q = 0 .. L-1: the END positions of every l-gram (l <= 4) that starts inside the data; the padding behind is counted, not sorted (seg_cpad) */
	const uint32_t L = n ? n + 3u : 0u;
	const uint32_t ncand = a.ncand, Tu = a.Tu;
	uint2 *A = a.la + base, *Bq = a.lb + base;
	uint32_t *S4 = a.S4 + base, *K4 = a.K4 + base;

	/* entries of the slot behind the list are never read: the walk kernel stops at the list's end, and scan2.hip turns everything from entry
	 * len + 3 on into fillers before it refines dense classes -- which leaves the first three entries of an EMPTY chunk's slot to this kernel */
	if (!n) { if (tid < 3u) { S4[tid] = 2u; K4[tid] = 0xFFFFFFFFu; } return; }

	uint64_t tclk = a.prof ? x3_clock() : 0;
#define SEG_MARK(k) do { if (a.prof && tid == 0) { const uint64_t now_ = x3_clock(); atomicAdd((unsigned long long *)&a.prof[k], (unsigned long long)(now_ - tclk)); tclk = now_; } } while (0)
	uint32_t *hist = (uint32_t *)stage;
	uint32_t *const mfield = BIG ? a.gmf + (base >> 4) : mfield_l; /* (slots are 256-byte aligned) */
	uint32_t *const rbits = BIG ? a.rare + (base >> 5) : rbits_l;
	if (!BIG) {
		for (uint32_t i = tid; i < (n + 15u) / 16u; i += X3_SEG_THREADS) mfield[i] = 0u;
		for (uint32_t i = tid; i < (n + 31u) / 32u; i += X3_SEG_THREADS) rbits[i] = 0u;
	}
	for (uint32_t i = tid; i < 8u * 256u; i += X3_SEG_THREADS) hist[i] = 0u;
	__syncthreads();

	/* ---- phase 0: keys + byte histogram.  Four elements per thread from two aligned dwords. ---- */
	{
		const uint32_t *w32 = (const uint32_t *)(a.bytes + base); /* slots are 256-byte aligned; bytes before slot c > 0 are slot c-1's zero tail */
		uint32_t *hmine = hist + (lane & 7u) * 256u;
		for (uint32_t q4 = tid * 4u; q4 < L; q4 += X3_SEG_THREADS * 4u) {
			const uint32_t w1 = w32[q4 >> 2];
			const uint32_t w0 = (base + q4) ? w32[(int32_t)(q4 >> 2) - 1] : 0u;
			const uint64_t v = (uint64_t)w0 | ((uint64_t)w1 << 32);
			uint2 e[4];
#pragma unroll
			for (uint32_t k = 0; k < 4; k++) {
				e[k].x = seg_bswap((uint32_t)(v >> (8u * (k + 1u)))); /* byte q in bits 0-7, q-1 in 8-15, q-2, q-3 */
				e[k].y = base + q4 + k;
				if (q4 + k < L) atomicAdd(&hmine[e[k].x & 0xFFu], 1u);
			}
			if (q4 + 4u <= L) {
				uint4 *d4 = (uint4 *)(A + q4);
				d4[0] = make_uint4(e[0].x, e[0].y, e[1].x, e[1].y);
				d4[1] = make_uint4(e[2].x, e[2].y, e[3].x, e[3].y);
			} else {
				for (uint32_t k = 0; k < 4 && q4 + k < L; k++) A[q4 + k] = e[k];
			}
		}
	}
	__syncthreads();
	{
		uint32_t h = 0, incl = 0;
		if (tid < 256u) {
#pragma unroll
			for (uint32_t k = 0; k < 8; k++) h += hist[k * 256u + tid];
			incl = x3_wave_incl_scan_u32(h);
			if (lane == X3_WAVE - 1u) wtot[wv] = incl;
		}
		__syncthreads();
		if (tid < 256u) {
			uint32_t off = 0;
			for (uint32_t w = 0; w < wv; w++) off += wtot[w];
			bbase[tid] = off + incl - h;
		}
	}
	__syncthreads();
	SEG_MARK(0);

	const uint2 none = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
	for (uint32_t l = 1; l <= 4; l++) {
		/* ---- pass l: stable counting sort on key byte l-1 of list l-1 -> list l; the entries read are tested for level l-1 on the way ---- */
		const uint2 *in = (l & 1u) ? A : Bq;
		uint2 *out = (l & 1u) ? Bq : A;
		const uint32_t sh = 8u * (l - 1u), lv = l - 1u;
		if (tid < 256u) bcur[tid] = bbase[tid];
		uint2 nx[X3_SEG_E];
		{
			const uint32_t i0 = wv * (X3_SEG_E * X3_WAVE) + lane;
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) nx[e] = i0 + e * X3_WAVE < L ? in[i0 + e * X3_WAVE] : none;
		}
		for (uint32_t t0 = 0; t0 < L; t0 += X3_SEG_TILE) {
			*(uint4 *)&cnt[tid * 4u] = make_uint4(0u, 0u, 0u, 0u);
			uint2 it[X3_SEG_E], la[X3_SEG_E];
			uint32_t rk[X3_SEG_E];
			const uint32_t i0 = t0 + wv * (X3_SEG_E * X3_WAVE) + lane;
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) it[e] = nx[e];
			if (t0 + X3_SEG_TILE < L) { /* the next tile's entries are on their way while this one is ranked */
#pragma unroll
				for (uint32_t e = 0; e < X3_SEG_E; e++) { const uint32_t idx = i0 + X3_SEG_TILE + e * X3_WAVE; nx[e] = idx < L ? in[idx] : none; }
			}
			if (lv) {
#pragma unroll
				for (uint32_t e = 0; e < X3_SEG_E; e++) { const uint32_t u = i0 + e * X3_WAVE + Tu + 1u; la[e] = u < L ? in[u] : none; }
			}
			__syncthreads();
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) {
				const bool valid = i0 + e * X3_WAVE < L;
				const uint32_t d = (it[e].x >> sh) & 0xFFu;
				const uint64_t mask = seg_match8(d, valid);
				const uint32_t lower = (uint32_t)x3_popc64(mask & (((uint64_t)1 << lane) - 1u));
				const uint32_t prev = valid ? cnt[d * X3_SEG_WAVES + wv] : 0u;
				x3_wave_order();
				if (valid && lower == 0u) cnt[d * X3_SEG_WAVES + wv] = prev + (uint32_t)x3_popc64(mask);
				x3_wave_order();
				rk[e] = prev + lower;
			}
			if (lv) {
#pragma unroll
				for (uint32_t e = 0; e < X3_SEG_E; e++)
					if (i0 + e * X3_WAVE < L) seg_level(lv, it[e], la[e], i0 + e * X3_WAVE, in, L, n, base, ncand, Tu, mfield, rbits, a.kexact);
			}
			__syncthreads();
			/* exclusive scan of the counter table in (digit, wave) order = the tile-sorted order */
			const uint4 c = *(const uint4 *)&cnt[tid * 4u];
			const uint32_t s = c.x + c.y + c.z + c.w;
			const uint32_t incl = x3_wave_incl_scan_u32(s);
			if (lane == X3_WAVE - 1u) wtot[wv] = incl;
			__syncthreads();
			uint32_t ex = incl - s;
			for (uint32_t w = 0; w < wv; w++) ex += wtot[w];
			*(uint4 *)&cnt[tid * 4u] = make_uint4(ex, ex + c.x, ex + c.x + c.y, ex + c.x + c.y + c.z);
			__syncthreads();
			const uint32_t tile_n = L - t0 < X3_SEG_TILE ? L - t0 : X3_SEG_TILE;
			uint32_t delta = 0; /* entries of digit `tid` in this tile */
			if (tid < 256u) delta = (tid < 255u ? cnt[(tid + 1u) * X3_SEG_WAVES] : tile_n) - cnt[tid * X3_SEG_WAVES];
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) {
				if (i0 + e * X3_WAVE < L) stage[cnt[((it[e].x >> sh) & 0xFFu) * X3_SEG_WAVES + wv] + rk[e]] = it[e];
			}
			__syncthreads();
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) {
				const uint32_t i = e * X3_SEG_THREADS + tid;
				if (i < tile_n) {
					const uint2 item = stage[i];
					const uint32_t d = (item.x >> sh) & 0xFFu;
					const uint32_t dest = bcur[d] + (i - cnt[d * X3_SEG_WAVES]);
					if (l < 4u) out[dest] = item;
					else { S4[dest] = item.y; K4[dest] = item.x; }
				}
			}
			__syncthreads();
			if (tid < 256u) bcur[tid] += delta; /* (read again only behind the next tile's barriers) */
		}
		__syncthreads();
		SEG_MARK(l);
	}

	/* ---- level 4 on list 4: count_3 >= K -> m >= 3, and deeper levels need the candidates themselves (walk kernel) -- unless the class has
	 * more than dense_at members inside the window: then it is refined instead (scan2.hip) ---- */
	for (uint32_t j0 = 0; j0 < L; j0 += X3_SEG_TILE) { /* uniform trip count: the queue push below is a wave operation */
		uint32_t kk[X3_SEG_E], ss[X3_SEG_E], ku[X3_SEG_E], su[X3_SEG_E];
#pragma unroll
		for (uint32_t e = 0; e < X3_SEG_E; e++) {
			const uint32_t j = j0 + e * X3_SEG_THREADS + tid, u = j + Tu + 1u;
			kk[e] = j < L ? K4[j] : 0u; ss[e] = j < L ? S4[j] : 0xFFFFFFFFu;
			ku[e] = u < L ? K4[u] : 0u; su[e] = u < L ? S4[u] : 0xFFFFFFFFu;
		}
#pragma unroll
		for (uint32_t e = 0; e < X3_SEG_E; e++) {
			const uint32_t j = j0 + e * X3_SEG_THREADS + tid;
			bool push = false;
			uint32_t K = Tu + 1u, gp = 0;
			const uint32_t qrel = ss[e] - base;
			if (j < L && qrel >= 3u && qrel - 3u < n) {
				const uint32_t prel = qrel - 3u, wend = qrel + ncand;
				const uint32_t cpad = kk[e] == 0u ? seg_cpad(qrel, ncand, n) : 0u;
				gp = base + prel;
				bool pass;
				if (cpad == 0u) pass = ku[e] == kk[e] && su[e] - base <= wend;
				else if (cpad > Tu) pass = true;
				else { const uint32_t u = j + (Tu + 1u - cpad); pass = u < L && K4[u] == kk[e] && S4[u] - base <= wend; }
				if (!pass && ((rbits[prel >> 5] >> (prel & 31u)) & 1u)) {
					K = a.kexact[gp];
					if (K >= 2u) {
						if (cpad >= K) pass = true;
						else if (j + (K - cpad) < L) pass = K4[j + (K - cpad)] == kk[e] && S4[j + (K - cpad)] - base <= wend;
					}
				}
				if (pass) {
					atomicAdd(&mfield[prel >> 4], 1u << (2u * (prel & 15u)));
					const uint32_t ud = j + a.dense_at;
					if (ud < L && K4[ud] == kk[e] && S4[ud] - base <= wend && S4[ud] - base - 3u < n) a.nact[1] = 1u;
					else push = true;
				}
			}
			const uint64_t mk = x3_ballot(push);
			if (mk) {
				uint32_t s0 = 0;
				if (lane == (uint32_t)x3_ctz64(mk)) s0 = atomicAdd(&a.nact[0], (uint32_t)x3_popc64(mk));
				s0 = x3_readlane_u32(s0, (uint32_t)x3_ctz64(mk));
				if (push) {
					const uint32_t sl = s0 + (uint32_t)x3_popc64(mk & (((uint64_t)1 << lane) - 1u));
					a.act[sl] = gp; a.act_k[sl] = K; a.act_j[sl] = base + j;
				}
			}
		}
	}
	__syncthreads();
	SEG_MARK(5);

	/* ---- m[] of the chunk: the 2-bit level counters as bytes, 16 positions (one counter word) per thread; the marks of the positions with a small K ---- */
	for (uint32_t w = tid; w < (n + 15u) / 16u; w += X3_SEG_THREADS) {
		const uint32_t f = mfield[w];
		uint32_t o[4];
#pragma unroll
		for (uint32_t k = 0; k < 4; k++) {
			const uint32_t g = f >> (8u * k);
			o[k] = (g & 3u) | ((g >> 2) & 3u) << 8 | ((g >> 4) & 3u) << 16 | ((g >> 6) & 3u) << 24;
		}
		*(uint4 *)(a.m + base + 16u * w) = make_uint4(o[0], o[1], o[2], o[3]); /* (the last word may spill <= 15 bytes into the slot's padding) */
	}
	if (!BIG) for (uint32_t w = tid; w < (n + 31u) / 32u; w += X3_SEG_THREADS) a.rare[(base >> 5) + w] = rbits[w];
}

#ifndef X3_EMU
__global__ void __launch_bounds__(X3_SEG_THREADS) x3_segscan_kernel(X3SegArgs a) { x3_segscan_body<false>(a); }
__global__ void __launch_bounds__(X3_SEG_THREADS) x3_segscan_big_kernel(X3SegArgs a) { x3_segscan_body<true>(a); }
int x3_scan_seg_launch(const X3SegArgs &a, uint32_t nchunks, hipStream_t st)
{
	if (a.gmf) hipLaunchKernelGGL(x3_segscan_big_kernel, dim3(nchunks), dim3(X3_SEG_THREADS), 0, st, a);
	else hipLaunchKernelGGL(x3_segscan_kernel, dim3(nchunks), dim3(X3_SEG_THREADS), 0, st, a);
	HIPCHK(hipGetLastError());
	return X3H_OK;
}
#else
static void segscan_tramp(void *p) { x3_segscan_body<false>(*(const X3SegArgs *)p); }
static void segscan_big_tramp(void *p) { x3_segscan_body<true>(*(const X3SegArgs *)p); }
int x3_scan_seg_launch(const X3SegArgs &a, uint32_t nchunks, hipStream_t)
{
	x3emu_launch(a.gmf ? segscan_big_tramp : segscan_tramp, (void *)&a, dim3(nchunks), dim3(X3_SEG_THREADS));
	return X3H_OK;
}
#endif

/* One workgroup per chunk pays off once there are enough chunks to fill the chip.  -> 0: no (the chip-wide sort of scan2.hip), 1: yes, the
 * chunk's level counters fit LDS, 2: yes, with the counters in global memory (chunks up to X3_SEG_MAXLEN_BIG; needs enough of them that a
 * quarter of the CUs are busy with one chunk each for the whole scan).  X3H_SEG_MIN=n moves the first threshold (tests), 0 turns it off. */
int x3_scan_seg_applies(uint32_t nchunks, uint64_t max_len)
{
	uint32_t min_streams = X3_SEG_MIN_STREAMS, min_big = X3_SEG_MIN_STREAMS_BIG;
	if (const char *e = getenv("X3H_SEG_MIN")) { const int v = atoi(e); if (v >= 1) min_streams = min_big = (uint32_t)v; else if (v == 0 && *e == '0') return 0; }
	uint64_t small_max = X3_SEG_MAXLEN;
	if (const char *e = getenv("X3H_SEG_SMALL_MAX")) { const long long v = atoll(e); if (v >= 0) small_max = (uint64_t)v < X3_SEG_MAXLEN ? (uint64_t)v : X3_SEG_MAXLEN; } /* (tests: force the global-memory form) */
	if (nchunks >= min_streams && max_len <= small_max) return 1;
	if (nchunks >= min_big && max_len <= X3_SEG_MAXLEN_BIG) return 2;
	return 0;
}
