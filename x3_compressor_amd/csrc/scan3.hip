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
 *   level l  tested on the entries of list l WHILE pass l+1 reads them (level 4: one more sweep over list 4): count_{l-1}(p) >= K  <=>
 *            entry j+K has the same l-gram and lies inside p's window -- scan2.hip's O(1) test.  A level that passes adds one to a 2-BIT
 *            COUNTER of the position IN LDS (levels pass in order: the counter IS m for m <= 3), so m[] leaves the chip once, as coalesced
 *            bytes, instead of as one scattered byte store per level.  The W padding zeros behind a chunk are COUNTED where a window reaches
 *            them (seg_cpad), never sorted;
 *   deeper   the positions whose 4-gram still repeats K times: x3_segrefine_kernel below extends the classes byte by byte (levels 5..32) with
 *            the same workgroup-per-chunk machinery -- every class with a passing member for chunks up to 256 KiB; longer chunks (level
 *            counters in global memory) hand list 4 to the walk kernel / the chip-wide refinement of scan2.hip (same list layout: chunk
 *            c's list occupies the entries of its slot of the padded layout).
 * HBM traffic: 16 B per element and pass + 8 B of keys + the level-4 sweep, all of it sequential or in digit runs (80 B per element
 * algorithmic; measured 1.3x that since the level tests read their list entries out of LDS -- round 3: 2.1x).  What limits the kernel is
 * instruction issue and LDS, not HBM: sixteen wavefronts, six barriers per tile, one workgroup per CU.
 */
#include "x3_host.h"

#include <stdlib.h>

#include "seg_rank.h"

__device__ static __forceinline__ uint32_t seg_bswap(uint32_t v) { return __builtin_bswap32(v); }

/* Padding.  The W zero bytes behind a chunk take part in its windows (x3.c:579,590), but they are not sorted: the lists hold the END positions
 * q < n + 3 only (every gram that starts inside the data).  An end position e >= n + 3 ends an all-zero gram of any length <= 4, so a query
 * whose gram is all zero meets   cpad = #{ e in [n + 3, q + ncand] }   further occurrences behind everything the list shows it (they are the
 * class's largest positions): "the K-th next occurrence lies inside the window"  <=>  cpad >= K, or entry j + (K - cpad) is of the class and
 * inside the window. */
__device__ static __forceinline__ uint32_t seg_cpad(uint32_t qrel, uint32_t ncand, uint32_t n) { const uint32_t we = qrel + ncand; return we >= n + 3u ? we - n - 2u : 0u; }

/* The list a level test reads.  A test of entry j looks at entries j + 1 .. j + T + 1 only, so while a tile of the list is ranked, the tile and the
 * X3_SEG_HALO entries behind it stand in LDS in list order (`lds`, entry t0 first) and the tests read them there: asked of memory, every one of
 * these reads is a round trip of a microsecond that a whole workgroup waits for (level 1 of a byte that is rare in its window searches ~8 entries
 * one after the other; a third of all positions of Zipf-distributed bytes).  T + 1 > X3_SEG_HALO, or X3H_SEG_LA_LDS=0: `lds` = nullptr, memory. */
#define X3_SEG_HALO 512u
struct SegList {
	const uint2 *g;   /* the list in memory */
	const uint2 *lds; /* entries t0 .. t0 + X3_SEG_TILE + X3_SEG_HALO - 1 of it, or nullptr */
	uint32_t t0;
	__device__ __forceinline__ uint2 at(uint32_t idx) const { return (lds && idx - t0 < X3_SEG_TILE + X3_SEG_HALO) ? lds[idx - t0] : g[idx]; }
};

/* the test of level lv (gram length lv) for entry j = `it` of list lv; `la` = entry j + T + 1 of the same list (position 0xFFFFFFFF when
 * there is none); list = list lv as (key, position) pairs.  Level 1 fixes K = min(T+1, count_0) (positions with a smaller K are marked in
 * `rbits`, their K goes to kexact); levels 2, 3 add one to the position's 2-bit counter when count_{lv-1} >= K. */
__device__ static __forceinline__ void seg_level(uint32_t lv, const uint2 it, const uint2 la, uint32_t j, const SegList &list, uint32_t L, uint32_t n,
                                                 uint32_t base, uint32_t ncand, uint32_t Tu, uint32_t *mfield, uint32_t *rbits, uint32_t *kexact, bool mf_lds)
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
			if (u < L) { const uint2 eu = list.at(u); if ((eu.x & 0xFFu) == kj && eu.y - base <= wend) return; }
		}
		uint32_t lo = 0, bnd = Tu - cpad; /* count the listed ones: binary search, predicate true at lo */
		if (j + bnd >= L) bnd = L - 1u - j;
		while (lo < bnd) {
			const uint32_t mid = (lo + bnd + 1u) >> 1;
			const uint2 em = list.at(j + mid);
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
		if (u < L) { const uint2 eu = list.at(u); pass = ((eu.x ^ it.x) & msk) == 0u && eu.y - base <= wend; }
	}
	/* (levels pass in order -- count_l <= count_{l-1} -- so a marked position whose counter says that it failed the level before fails this one, and
	 * its K stays where it is: the load is a scattered one, and most marked positions fail level 2.  Timing switches (round 4, profiles/r04_segscan_phase_counters.txt): the small-K path was
	 * three quarters of what the level tests cost.) */
	if (!pass && ((rbits[prel >> 5] >> (prel & 31u)) & 1u) && (lv < 3u || !mf_lds || ((mfield[prel >> 4] >> (2u * (prel & 15u))) & 3u) >= lv - 2u)) {
		const uint32_t K = kexact[base + prel];
		if (K >= 2u) { /* K < 2: count_0 < 2, nothing repeats */
			if (cpad >= K) pass = true;
			else if (j + (K - cpad) < L) {
				const uint2 eu = list.at(j + (K - cpad));
				pass = ((eu.x ^ it.x) & msk) == 0u && eu.y - base <= wend;
			}
		}
	}
	if (pass) atomicAdd(&mfield[prel >> 4], 1u << (2u * (prel & 15u)));
}

/* BIG = false: chunks of at most X3_SEG_MAXLEN bytes, their level counters and small-K marks live in LDS; BIG = true: longer chunks, both
 * live in global memory (a.gmf / a.rare, zeroed by the caller; the atomics execute in L2) */
/* element q of a chunk's list 0: key = the four bytes ENDING at q (byte q in bits 0-7, q-1 in 8-15, ...; bytes before the buffer read as 0 --
 * before a later slot they are the previous slot's zero tail), value = the global position */
__device__ static __forceinline__ uint2 seg_gen(const uint8_t *bytes, uint32_t base, uint32_t q)
{
	uint32_t w;
	if (base + q >= 3u) __builtin_memcpy(&w, bytes + (base + q - 3u), 4); /* (an unaligned dword load) */
	else { __builtin_memcpy(&w, bytes, 4); w <<= 8u * (3u - q); }
	return make_uint2(seg_bswap(w), base + q);
}

template <bool BIG>
__device__ static void x3_segscan_body(const X3SegArgs &a)
{
	X3_LDS uint32_t mfield_l[BIG ? 1 : X3_SEG_MAXLEN / 16]; /* 2 bits per position: levels 1..3 passed so far */
	X3_LDS uint32_t rbits_l[BIG ? 1 : X3_SEG_MAXLEN / 32];  /* positions whose K is below T+1 */
	X3_LDS uint32_t cnt[256 * X3_SEG_CS];              /* [digit][wave]: counts, then exclusive prefix in tile-sorted order; a digit's counters X3_SEG_CS = 17 words apart: the lanes
	                                                    * of a wavefront (one wave number, many digits) meet in every LDS bank instead of two (bank conflicts were 63 % of the LDS cycles) */
	X3_LDS uint2 stage[X3_SEG_TILE + X3_SEG_HALO];     /* the tile in list order + the entries behind it (level tests), then the tile in sorted order (phase 0: eight copies of the byte histogram) */
	X3_LDS uint32_t bbase[256], bcur[256];
	X3_LDS __attribute__((aligned(16))) uint32_t wtot[X3_SEG_WAVES];

	const uint32_t tid = threadIdx.x, lane = x3_lane(), wv = tid / X3_WAVE;
	const X3Chunk ck = a.chunks[blockIdx.x];
	const uint32_t n = ck.len, base = (uint32_t)ck.byte_off;
	/* elements q = 0 .. L-1: the END positions of every l-gram (l <= 4) that starts inside the data; the padding behind is counted, not sorted (seg_cpad) */
	const uint32_t L = n ? n + 3u : 0u;
	const uint32_t ncand = a.ncand, Tu = a.Tu;
	const bool la_lds = a.la_lds != 0u && a.Tu + 1u <= X3_SEG_HALO;
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

	/* ---- phase 0: the byte histogram (the digit of element q in pass 1 is byte q; bytes behind the data read as the zeros they are).  The
	 * (key, position) pairs themselves are never written out: pass 1 makes them from the input bytes as it reads (seg_gen). ---- */
	{
		const uint32_t *w32 = (const uint32_t *)(a.bytes + base); /* slots are 256-byte aligned */
		uint32_t *hmine = hist + (lane & 7u) * 256u;
		for (uint32_t q4 = tid * 4u; q4 < L; q4 += X3_SEG_THREADS * 4u) {
			const uint32_t w1 = w32[q4 >> 2];
#pragma unroll
			for (uint32_t k = 0; k < 4; k++)
				if (q4 + k < L) atomicAdd(&hmine[(w1 >> (8u * k)) & 0xFFu], 1u);
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
			bbase[tid] = seg_waves_before(wtot, wv) + incl - h;
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
			for (uint32_t e = 0; e < X3_SEG_E; e++) nx[e] = i0 + e * X3_WAVE < L ? (l == 1u ? seg_gen(a.bytes, base, i0 + e * X3_WAVE) : in[i0 + e * X3_WAVE]) : none;
		}
		const bool stg = lv && la_lds;
		uint2 hx = none; /* one of the X3_SEG_HALO entries behind the tile, asked for a tile ahead */
		if (stg && tid < X3_SEG_HALO && X3_SEG_TILE + tid < L) hx = in[X3_SEG_TILE + tid];
		for (uint32_t t0 = 0; t0 < L; t0 += X3_SEG_TILE) {
			for (uint32_t i = tid; i < 256u * X3_SEG_CS; i += X3_SEG_THREADS) cnt[i] = 0u;
			uint2 it[X3_SEG_E], la[X3_SEG_E];
			uint32_t rk[X3_SEG_E];
			const uint32_t i0 = t0 + wv * (X3_SEG_E * X3_WAVE) + lane, loc0 = wv * (X3_SEG_E * X3_WAVE) + lane;
			const SegList list = { in, stg ? stage : nullptr, t0 };
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) it[e] = nx[e];
			if (stg) {
#pragma unroll
				for (uint32_t e = 0; e < X3_SEG_E; e++) stage[loc0 + e * X3_WAVE] = it[e];
				if (tid < X3_SEG_HALO) stage[X3_SEG_TILE + tid] = hx;
			} else if (lv) {
#pragma unroll
				for (uint32_t e = 0; e < X3_SEG_E; e++) { const uint32_t u = i0 + e * X3_WAVE + Tu + 1u; la[e] = u < L ? in[u] : none; }
			}
			if (t0 + X3_SEG_TILE < L) { /* the next tile's entries are on their way while this one is ranked */
#pragma unroll
				for (uint32_t e = 0; e < X3_SEG_E; e++) { const uint32_t idx = i0 + X3_SEG_TILE + e * X3_WAVE; nx[e] = idx < L ? (l == 1u ? seg_gen(a.bytes, base, idx) : in[idx]) : none; }
				if (stg && tid < X3_SEG_HALO) { const uint32_t idx = t0 + 2u * X3_SEG_TILE + tid; hx = idx < L ? in[idx] : none; }
			}
			__syncthreads();
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) {
				const bool valid = i0 + e * X3_WAVE < L;
				const uint32_t d = (it[e].x >> sh) & 0xFFu;
				uint32_t mlo, mhi;
				seg_match<8>(d, valid, mlo, mhi);
				const uint32_t lower = seg_lower(mlo, mhi);
				const uint32_t prev = valid ? cnt[d * X3_SEG_CS + wv] : 0u;
				x3_wave_order();
				if (valid && lower == 0u) cnt[d * X3_SEG_CS + wv] = prev + seg_size(mlo, mhi);
				x3_wave_order();
				rk[e] = prev + lower;
			}
			if (lv) {
#pragma unroll
				for (uint32_t e = 0; e < X3_SEG_E; e++)
					if (stg) la[e] = stage[loc0 + e * X3_WAVE + Tu + 1u]; /* (entries behind the list's end read as `none`) */
#pragma unroll
				for (uint32_t e = 0; e < X3_SEG_E; e++)
					if (i0 + e * X3_WAVE < L) seg_level(lv, it[e], la[e], i0 + e * X3_WAVE, list, L, n, base, ncand, Tu, mfield, rbits, a.kexact, !BIG);
			}
			__syncthreads();
			/* exclusive scan of the counter table in (digit, wave) order = the tile-sorted order */
			uint32_t *const c4 = &cnt[(tid >> 2) * X3_SEG_CS + (tid & 3u) * 4u]; /* four waves' counters of digit tid / 4 */
			const uint4 c = make_uint4(c4[0], c4[1], c4[2], c4[3]);
			const uint32_t s = c.x + c.y + c.z + c.w;
			const uint32_t incl = x3_wave_incl_scan_u32(s);
			if (lane == X3_WAVE - 1u) wtot[wv] = incl;
			__syncthreads();
			const uint32_t ex = incl - s + seg_waves_before(wtot, wv);
			c4[0] = ex; c4[1] = ex + c.x; c4[2] = ex + c.x + c.y; c4[3] = ex + c.x + c.y + c.z;
			__syncthreads();
			const uint32_t tile_n = L - t0 < X3_SEG_TILE ? L - t0 : X3_SEG_TILE;
			uint32_t delta = 0; /* entries of digit `tid` in this tile */
			if (tid < 256u) delta = (tid < 255u ? cnt[(tid + 1u) * X3_SEG_CS] : tile_n) - cnt[tid * X3_SEG_CS];
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) {
				if (i0 + e * X3_WAVE < L) stage[cnt[((it[e].x >> sh) & 0xFFu) * X3_SEG_CS + wv] + rk[e]] = it[e];
			}
			__syncthreads();
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) {
				const uint32_t i = e * X3_SEG_THREADS + tid;
				if (i < tile_n) {
					const uint2 item = stage[i];
					const uint32_t d = (item.x >> sh) & 0xFFu;
					const uint32_t dest = bcur[d] + (i - cnt[d * X3_SEG_CS]);
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
	uint32_t nk[X3_SEG_E], ns[X3_SEG_E];
#pragma unroll
	for (uint32_t e = 0; e < X3_SEG_E; e++) { const uint32_t j = e * X3_SEG_THREADS + tid; nk[e] = j < L ? K4[j] : 0u; ns[e] = j < L ? S4[j] : 0xFFFFFFFFu; }
	uint2 hx4 = make_uint2(0u, 0xFFFFFFFFu); /* (key, position) of one of the entries behind the tile */
	if (la_lds && tid < X3_SEG_HALO && X3_SEG_TILE + tid < L) hx4 = make_uint2(K4[X3_SEG_TILE + tid], S4[X3_SEG_TILE + tid]);
	for (uint32_t j0 = 0; j0 < L; j0 += X3_SEG_TILE) { /* uniform trip count: the queue push below is a wave operation */
		uint32_t kk[X3_SEG_E], ss[X3_SEG_E], ku[X3_SEG_E], su[X3_SEG_E];
#pragma unroll
		for (uint32_t e = 0; e < X3_SEG_E; e++) {
			const uint32_t loc = e * X3_SEG_THREADS + tid, u = j0 + loc + Tu + 1u;
			kk[e] = nk[e]; ss[e] = ns[e];
			if (la_lds) stage[loc] = make_uint2(kk[e], ss[e]);
			else { ku[e] = u < L ? K4[u] : 0u; su[e] = u < L ? S4[u] : 0xFFFFFFFFu; }
		}
		if (la_lds && tid < X3_SEG_HALO) stage[X3_SEG_TILE + tid] = hx4;
		if (j0 + X3_SEG_TILE < L) { /* the next tile's entries, asked for now */
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) { const uint32_t j = j0 + X3_SEG_TILE + e * X3_SEG_THREADS + tid; nk[e] = j < L ? K4[j] : 0u; ns[e] = j < L ? S4[j] : 0xFFFFFFFFu; }
			if (la_lds && tid < X3_SEG_HALO) { const uint32_t idx = j0 + 2u * X3_SEG_TILE + tid; hx4 = idx < L ? make_uint2(K4[idx], S4[idx]) : make_uint2(0u, 0xFFFFFFFFu); }
		}
		if (la_lds) {
			__syncthreads();
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) { const uint2 t = stage[e * X3_SEG_THREADS + tid + Tu + 1u]; ku[e] = t.x; su[e] = t.y; }
		}
		/* entry idx of list 4 as (key, position): out of the LDS copy where it has it */
#define SEG_L4(idx) ((la_lds && (idx) - j0 < X3_SEG_TILE + X3_SEG_HALO) ? stage[(idx) - j0] : make_uint2(K4[idx], S4[idx]))
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
				else { const uint32_t u = j + (Tu + 1u - cpad); pass = false; if (u < L) { const uint2 eu = SEG_L4(u); pass = eu.x == kk[e] && eu.y - base <= wend; } }
				if (!pass && ((rbits[prel >> 5] >> (prel & 31u)) & 1u) && (BIG || ((mfield[prel >> 4] >> (2u * (prel & 15u))) & 3u) >= 2u)) { /* (failed level 3: fails this one) */
					K = a.kexact[gp];
					if (K >= 2u) {
						if (cpad >= K) pass = true;
						else if (j + (K - cpad) < L) { const uint2 eu = SEG_L4(j + (K - cpad)); pass = eu.x == kk[e] && eu.y - base <= wend; }
					}
				}
				if (pass) {
					atomicAdd(&mfield[prel >> 4], 1u << (2u * (prel & 15u)));
					const uint32_t ud = j + a.dense_at;
					uint2 ed = make_uint2(0u, 0u);
					if (ud < L) ed = SEG_L4(ud);
					if (ud < L && ed.x == kk[e] && ed.y - base <= wend && ed.y - base - 3u < n) { a.nact[1] = 1u; a.dense_chunk[blockIdx.x] = 1u; }
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
		if (la_lds) __syncthreads(); /* (the next tile overwrites the LDS copy) */
	}
#undef SEG_L4
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


/* ============================================================================================================
 * Dense classes, refined by the chunk's own workgroup.  A 4-gram class with more than dense_at members inside some member's window (zero
 * runs, sparse 16-bit samples, periodic data) is not swept by the walk kernel but extended one byte at a time (scan2.hip header, step 4):
 * level test "the K-th next member of my class lies inside my window", classes without a passing member dropped, the rest split by the
 * next byte with positions staying ascending inside a class.  scan2.hip does that chip-wide -- ten launches, two host round trips and a
 * 26-bit pair sort per level, 28 levels.  The classes of ONE chunk number about a thousand at most (a class needs K members inside a
 * window), so here one workgroup keeps a chunk's list in its two halves of the K1 list buffers and per level makes five sweeps over it:
 *   test      entry j passes  <=>  entry j + K - cpad is of its class and inside its window (cpad: padding zeros counted, seg_cpad's rule in
 *             start coordinates); a pass is recorded IN the entry (the level), its class marked in an LDS bitmap;
 *   compact   entries of marked classes move on (new class number = rank of the class among the marked ones), the byte behind the gram
 *             is attached; entries that leave write their m[] byte -- ONE store per position for the whole refinement;
 *   sort      stable counting-sort passes on 9-bit digits of (class number << 8 | byte): two passes up to 1024 classes;
 *   heads     neighbours with different (class, byte) start a class; a scan numbers the classes of the new list.
 * Elements are START positions relative to the chunk from here on.  Entry = { position (19 bits) | byte << 19 | last level passed << 27,
 * class number | all-zero gram << 31 }.
 * ============================================================================================================ */
#define X3_REF_DB 9u
#define X3_REF_DIGITS (1u << X3_REF_DB)
#define X3_REF_POS(x) ((x) & 0x7FFFFu)
#define X3_REF_NOPOS 0x7FFFFu /* a gram that starts before the chunk: never a query, never inside a window (chunks are at most 2^18 bytes here) */
#define X3_REF_BYTE(x) (((x) >> 19) & 0xFFu)
#define X3_REF_LVL(x) ((x) >> 27)
#define X3_REF_CLS(y) ((y) & 0x7FFFFFFFu)
#define X3_REF_ZERO(y) ((y) >> 31)
#define X3_REF_MAXPASS 3u /* 9-bit digits of (class number << 8 | byte): 27 bits, more than a chunk's entries can need */

/* LDS atomics of a wavefront whose lanes mostly hit the SAME word (the zero class, the zero byte): if every participating lane names the same
 * word, one lane does the update for all of them -- 64 same-address atomics serialise otherwise */
__device__ static __forceinline__ void ref_wave_add(uint32_t *table, uint32_t idx, bool on)
{
	const uint64_t m = x3_ballot(on);
	if (!m) return;
	const uint32_t first = (uint32_t)x3_ctz64(m), i0 = x3_readlane_u32(idx, first);
	if (x3_ballot(on && idx != i0) == 0) { if (x3_lane() == first) atomicAdd(&table[i0], (uint32_t)x3_popc64(m)); }
	else if (on) atomicAdd(&table[idx], 1u);
}
__device__ static __forceinline__ void ref_wave_or(uint32_t *table, uint32_t idx, uint32_t bits, bool on)
{
	const uint64_t m = x3_ballot(on);
	if (!m) return;
	const uint32_t first = (uint32_t)x3_ctz64(m), i0 = x3_readlane_u32(idx, first), b0 = x3_readlane_u32(bits, first);
	if (x3_ballot(on && (idx != i0 || bits != b0)) == 0) { if (x3_lane() == first) atomicOr(&table[i0], b0); }
	else if (on) atomicOr(&table[idx], bits);
}

/* exclusive prefix of `v` over the workgroup's threads in thread order, plus the total; `wt` = X3_SEG_WAVES words of LDS; two barriers */
__device__ static __forceinline__ uint32_t ref_block_excl(uint32_t v, uint32_t *wt, uint32_t lane, uint32_t wv, uint32_t &total)
{
	const uint32_t incl = x3_wave_incl_scan_u32(v);
	__syncthreads(); /* (the previous use of wt is over) */
	if (lane == X3_WAVE - 1u) wt[wv] = incl;
	__syncthreads();
	total = seg_waves_before(wt, X3_SEG_WAVES);
	return incl - v + seg_waves_before(wt, wv);
}

/* one stable counting-sort pass over list `in` (cnt entries) on digit (key >> sh) & 511 with key = class << 8 | byte; hist = this digit's
 * histogram (512 counters, consumed); LDS: cnt_t (512 digits x 16 waves, rows X3_SEG_CS words apart), stage (4096 entries), bcur (512), wt */
__device__ static void ref_sort_pass(const uint2 *in, uint2 *out, uint32_t n, uint32_t sh, uint32_t *hist, uint32_t *cnt_t, uint2 *stage,
                                     uint32_t *bcur, uint32_t *wt, uint32_t tid, uint32_t lane, uint32_t wv)
{
	/* bucket bases: exclusive scan of the histogram */
	{
		uint32_t tot;
		const uint32_t h = tid < X3_REF_DIGITS ? hist[tid] : 0u;
		const uint32_t ex = ref_block_excl(h, wt, lane, wv, tot);
		if (tid < X3_REF_DIGITS) bcur[tid] = ex;
	}
	__syncthreads();
	uint2 nx[X3_SEG_E];
	{
		const uint32_t i0 = wv * (X3_SEG_E * X3_WAVE) + lane;
#pragma unroll
		for (uint32_t e = 0; e < X3_SEG_E; e++) { if (i0 + e * X3_WAVE < n) nx[e] = in[i0 + e * X3_WAVE]; else { nx[e].x = 0u; nx[e].y = 0u; } }
	}
	for (uint32_t t0 = 0; t0 < n; t0 += X3_SEG_TILE) {
		for (uint32_t i = tid; i < X3_REF_DIGITS * X3_SEG_CS; i += X3_SEG_THREADS) cnt_t[i] = 0u;
		uint2 it[X3_SEG_E];
		uint32_t rk[X3_SEG_E], dg[X3_SEG_E];
		const uint32_t i0 = t0 + wv * (X3_SEG_E * X3_WAVE) + lane;
#pragma unroll
		for (uint32_t e = 0; e < X3_SEG_E; e++) {
			it[e] = nx[e];
			dg[e] = (uint32_t)(((((uint64_t)X3_REF_CLS(it[e].y)) << 8) | X3_REF_BYTE(it[e].x)) >> sh) & (X3_REF_DIGITS - 1u);
		}
		if (t0 + X3_SEG_TILE < n) { /* the next tile's entries are on their way while this one is ranked */
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) { const uint32_t idx = i0 + X3_SEG_TILE + e * X3_WAVE; if (idx < n) nx[e] = in[idx]; else { nx[e].x = 0u; nx[e].y = 0u; } }
		}
		__syncthreads();
#pragma unroll
		for (uint32_t e = 0; e < X3_SEG_E; e++) {
			const bool valid = i0 + e * X3_WAVE < n;
			uint32_t mlo, mhi;
			seg_match<X3_REF_DB>(dg[e], valid, mlo, mhi);
			const uint32_t lower = seg_lower(mlo, mhi);
			const uint32_t prev = valid ? cnt_t[dg[e] * X3_SEG_CS + wv] : 0u;
			x3_wave_order();
			if (valid && lower == 0u) cnt_t[dg[e] * X3_SEG_CS + wv] = prev + seg_size(mlo, mhi);
			x3_wave_order();
			rk[e] = prev + lower;
		}
		__syncthreads();
		{ /* exclusive scan of the counter table in (digit, wave) order: 8 counters per thread */
			const uint32_t per = (X3_REF_DIGITS * X3_SEG_WAVES) / X3_SEG_THREADS;
			uint32_t c[8], sum = 0, tot;
			uint32_t *const cp = &cnt_t[((tid * per) / X3_SEG_WAVES) * X3_SEG_CS + (tid * per) % X3_SEG_WAVES]; /* (a digit's counters are X3_SEG_CS words apart) */
#pragma unroll
			for (uint32_t k = 0; k < 8; k++) { c[k] = k < per ? cp[k] : 0u; sum += c[k]; }
			uint32_t ex = ref_block_excl(sum, wt, lane, wv, tot);
#pragma unroll
			for (uint32_t k = 0; k < 8; k++) if (k < per) { cp[k] = ex; ex += c[k]; }
		}
		__syncthreads();
		const uint32_t tile_n = n - t0 < X3_SEG_TILE ? n - t0 : X3_SEG_TILE;
		uint32_t delta = 0;
		if (tid < X3_REF_DIGITS) delta = (tid + 1u < X3_REF_DIGITS ? cnt_t[(tid + 1u) * X3_SEG_CS] : tile_n) - cnt_t[tid * X3_SEG_CS];
#pragma unroll
		for (uint32_t e = 0; e < X3_SEG_E; e++) if (i0 + e * X3_WAVE < n) stage[cnt_t[dg[e] * X3_SEG_CS + wv] + rk[e]] = it[e];
		__syncthreads();
#pragma unroll
		for (uint32_t e = 0; e < X3_SEG_E; e++) {
			const uint32_t i = e * X3_SEG_THREADS + tid;
			if (i < tile_n) {
				const uint2 item = stage[i];
				const uint32_t d = (uint32_t)(((((uint64_t)X3_REF_CLS(item.y)) << 8) | X3_REF_BYTE(item.x)) >> sh) & (X3_REF_DIGITS - 1u);
				out[bcur[d] + (i - cnt_t[d * X3_SEG_CS])] = item;
			}
		}
		__syncthreads();
		if (tid < X3_REF_DIGITS) bcur[tid] += delta;
	}
	__syncthreads();
}

__device__ static void x3_segrefine_body(const X3SegArgs &a)
{
	X3_LDS uint32_t cnt_t[X3_REF_DIGITS * X3_SEG_CS];      /* 34 KiB */
	X3_LDS uint2 stage[X3_SEG_TILE];                       /* 32 KiB */
	X3_LDS uint32_t keepbits[X3_SEG_MAXLEN / 32 + 2];      /* one bit per class of the current list: a member passed this level */
	X3_LDS uint32_t kpre[X3_SEG_MAXLEN / 32 + 2];          /* kept classes before each bitmap word */
	X3_LDS uint32_t hist[X3_REF_MAXPASS][X3_REF_DIGITS], bcur[X3_REF_DIGITS];
	X3_LDS __attribute__((aligned(16))) uint32_t wt[X3_SEG_WAVES];
	const uint32_t tid = threadIdx.x, lane = x3_lane(), wv = tid / X3_WAVE;
	const uint32_t c = blockIdx.x;
	if (!a.dense_chunk[c]) return; /* (uniform) no class of this chunk was dense */
	const X3Chunk ck = a.chunks[c];
	const uint32_t n = ck.len, base = (uint32_t)ck.byte_off, L = n + 3u;
	const uint32_t ncand = a.ncand, Tu = a.Tu;
	const uint8_t *bytes = a.bytes + base;
	uint2 *A = a.la + base, *Bq = a.lb + base;
	const uint32_t *S4 = a.S4 + base, *K4 = a.K4 + base;
	const uint32_t *rare = a.rare;
	uint8_t *mout = a.m + base;

	/* ---- list 4 as { start position | zero flag, class number }: the class number of an entry = heads before it (a scan over the list) ---- */
	{
		uint32_t carry = 0;
		for (uint32_t t0 = 0; t0 < L; t0 += X3_SEG_TILE) {
			const uint32_t j0 = t0 + tid * X3_SEG_E;
			uint32_t k[X3_SEG_E + 1], s[X3_SEG_E], hd = 0;
			k[0] = j0 > 0 && j0 <= L ? K4[j0 - 1] : 0xFFFFFFFFu;
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) { k[e + 1] = j0 + e < L ? K4[j0 + e] : 0u; s[e] = j0 + e < L ? S4[j0 + e] - base : 0u; }
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) if (j0 + e < L && (j0 + e == 0 || k[e + 1] != k[e])) hd += 1u;
			uint32_t tot;
			uint32_t ord = carry + ref_block_excl(hd, wt, lane, wv, tot);
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) {
				if (j0 + e < L) {
					if (j0 + e == 0 || k[e + 1] != k[e]) ord++;
					uint2 en;
					en.x = s[e] >= 3u ? s[e] - 3u : X3_REF_NOPOS; /* (an END position < 3: the gram starts before the chunk) */
					en.y = (ord - 1u) | (k[e + 1] == 0u ? 0x80000000u : 0u);
					A[j0 + e] = en;
				}
			}
			carry += tot;
		}
	}
	uint2 *cur = A, *oth = Bq;
	uint32_t nl = L;
	__syncthreads();

	for (uint32_t len = 4; ; len++) {
		/* ---- test: entry j passes level `len` ---- */
		for (uint32_t i = tid; i < nl / 32u + 1u; i += X3_SEG_THREADS) keepbits[i] = 0u;
		__syncthreads();
		for (uint32_t j0 = 0; j0 < nl; j0 += X3_SEG_TILE) { /* uniform trip count: the class marks are wave operations */
			uint2 ev[X3_SEG_E];
			uint32_t Kv[X3_SEG_E];
#pragma unroll
			for (uint32_t q = 0; q < X3_SEG_E; q++) { /* the loads of four entries are in flight together */
				const uint32_t j = j0 + q * X3_SEG_THREADS + tid;
				if (j < nl) ev[q] = cur[j]; else { ev[q].x = X3_REF_NOPOS; ev[q].y = 0; }
			}
#pragma unroll
			for (uint32_t q = 0; q < X3_SEG_E; q++) {
				const uint32_t p = X3_REF_POS(ev[q].x), gp = base + p;
				Kv[q] = Tu + 1u;
				if (p < n && ((rare[gp >> 5] >> (gp & 31u)) & 1u)) Kv[q] = a.kexact[gp];
			}
#pragma unroll
			for (uint32_t q = 0; q < X3_SEG_E; q++) {
				const uint32_t j = j0 + q * X3_SEG_THREADS + tid;
				const uint2 e = ev[q];
				const uint32_t p = X3_REF_POS(e.x), K = Kv[q];
				bool pass = p < n && K >= 2u; /* (p >= n: starts before the chunk; K < 2: nothing repeats) */
				const uint32_t wend = p + ncand;
				if (pass) {
					const uint32_t cpad = (X3_REF_ZERO(e.y) && wend >= n) ? wend - n + 1u : 0u; /* padding START positions inside the window */
					if (cpad < K) {
						const uint32_t u = j + (K - cpad);
						pass = false;
						if (u < nl) { const uint2 eu = cur[u]; pass = eu.y == e.y && X3_REF_POS(eu.x) <= wend; }
					}
				}
				bool dense = pass && len > 4u; /* a class that was dense at length 4 is refined to the end (scan2.hip) */
				if (pass) {
					if (len > 4u) cur[j].x = (e.x & 0x07FFFFFFu) | ((len - 1u) << 27); /* m >= len - 1 (level 3 is in m[] already) */
					else {
						const uint32_t ud = j + a.dense_at;
						if (ud < nl) { const uint2 ed = cur[ud]; dense = ed.y == e.y && X3_REF_POS(ed.x) <= wend; }
					}
				}
				ref_wave_or(keepbits, X3_REF_CLS(e.y) >> 5, 1u << (e.y & 31u), dense && len < X3_MAXLEN);
			}
		}
		__syncthreads();
		/* ---- kept classes before each bitmap word; none kept (or the last level): every entry leaves ---- */
		uint32_t nkept_cls = 0;
		{
			const uint32_t nw = nl / 32u + 1u;
			uint32_t carry = 0;
			for (uint32_t w0 = 0; w0 < nw; w0 += X3_SEG_THREADS) {
				const uint32_t w = w0 + tid;
				const uint32_t pc = w < nw ? (uint32_t)__builtin_popcount(keepbits[w]) : 0u;
				uint32_t tot;
				const uint32_t ex = ref_block_excl(pc, wt, lane, wv, tot);
				if (w < nw) kpre[w] = carry + ex;
				carry += tot;
			}
			nkept_cls = carry;
		}
		__syncthreads();
		/* ---- compact: entries of kept classes move on with the byte behind their gram; the others write their m[] ---- */
		for (uint32_t i = tid; i < X3_REF_MAXPASS * X3_REF_DIGITS; i += X3_SEG_THREADS) (&hist[0][0])[i] = 0u;
		__syncthreads();
		uint32_t nnew = 0;
		{
			uint32_t carry = 0;
			for (uint32_t t0 = 0; t0 < nl; t0 += X3_SEG_TILE) {
				const uint32_t j0 = t0 + tid * X3_SEG_E;
				uint2 e[X3_SEG_E];
				uint32_t keep = 0, cntk = 0, okey[X3_SEG_E] = { 0, 0, 0, 0 };
#pragma unroll
				for (uint32_t q = 0; q < X3_SEG_E; q++) {
					if (j0 + q < nl) {
						e[q] = cur[j0 + q];
						const bool kp = nkept_cls && X3_REF_POS(e[q].x) < n && ((keepbits[X3_REF_CLS(e[q].y) >> 5] >> (e[q].y & 31u)) & 1u);
						if (kp) { keep |= 1u << q; cntk++; }
						else if (X3_REF_POS(e[q].x) < n && X3_REF_LVL(e[q].x) >= 4u) mout[X3_REF_POS(e[q].x)] = (uint8_t)X3_REF_LVL(e[q].x);
					}
				}
				uint32_t tot;
				uint32_t dst = carry + ref_block_excl(cntk, wt, lane, wv, tot);
#pragma unroll
				for (uint32_t q = 0; q < X3_SEG_E; q++) {
					if (keep & (1u << q)) {
						const uint32_t p = X3_REF_POS(e[q].x), bt = bytes[p + len];
						const uint32_t cl = X3_REF_CLS(e[q].y);
						const uint32_t rank = kpre[cl >> 5] + (uint32_t)__builtin_popcount(keepbits[cl >> 5] & ((1u << (cl & 31u)) - 1u));
						uint2 en;
						en.x = (e[q].x & ~(0xFFu << 19)) | (bt << 19);
						en.y = rank | ((X3_REF_ZERO(e[q].y) && !bt) ? 0x80000000u : 0u); /* all zero only while every byte is */
						oth[dst++] = en;
						const uint32_t key = (rank << 8) | bt;
						okey[q] = key;
					}
				}
#pragma unroll
				for (uint32_t q = 0; q < X3_SEG_E; q++)
#pragma unroll
					for (uint32_t ps = 0; ps < X3_REF_MAXPASS; ps++) ref_wave_add(hist[ps], (okey[q] >> (ps * X3_REF_DB)) & (X3_REF_DIGITS - 1u), (keep >> q) & 1u);
				carry += tot;
			}
			nnew = carry;
		}
		__syncthreads();
		if (!nnew) break;
		/* ---- sort by (class number, byte): positions stay ascending inside a (class, byte) group; as many 9-bit passes as the class numbers need ---- */
		uint2 *src = oth, *dst2 = cur;
		{
			uint32_t keybits = 8u;
			while (keybits < 27u && ((nkept_cls - 1u) >> (keybits - 8u))) keybits++;
			const uint32_t npass = (keybits + X3_REF_DB - 1u) / X3_REF_DB;
			for (uint32_t ps = 0; ps < npass; ps++) {
				ref_sort_pass(src, dst2, nnew, ps * X3_REF_DB, hist[ps], cnt_t, stage, bcur, wt, tid, lane, wv);
				uint2 *t = src; src = dst2; dst2 = t;
			}
		}
		/* ---- heads: neighbours with a different (class, byte) start a class; number the classes of the new list (src -> dst2) ---- */
		{
			uint32_t carry = 0;
			for (uint32_t t0 = 0; t0 < nnew; t0 += X3_SEG_TILE) {
				const uint32_t j0 = t0 + tid * X3_SEG_E;
				uint2 e[X3_SEG_E + 1];
				uint32_t hd = 0;
				if (j0 > 0 && j0 <= nnew) e[0] = src[j0 - 1]; else { e[0].x = 0; e[0].y = 0xFFFFFFFFu; }
#pragma unroll
				for (uint32_t q = 0; q < X3_SEG_E; q++) { if (j0 + q < nnew) e[q + 1] = src[j0 + q]; else { e[q + 1].x = 0; e[q + 1].y = 0; } }
#pragma unroll
				for (uint32_t q = 0; q < X3_SEG_E; q++)
					if (j0 + q < nnew && (j0 + q == 0 || e[q + 1].y != e[q].y || X3_REF_BYTE(e[q + 1].x) != X3_REF_BYTE(e[q].x))) hd += 1u;
				uint32_t tot;
				uint32_t ord = carry + ref_block_excl(hd, wt, lane, wv, tot);
#pragma unroll
				for (uint32_t q = 0; q < X3_SEG_E; q++) {
					if (j0 + q < nnew) {
						if (j0 + q == 0 || e[q + 1].y != e[q].y || X3_REF_BYTE(e[q + 1].x) != X3_REF_BYTE(e[q].x)) ord++;
						uint2 en = e[q + 1];
						en.y = (ord - 1u) | (en.y & 0x80000000u);
						dst2[j0 + q] = en;
					}
				}
				carry += tot;
			}
		}
		cur = dst2; oth = src;
		nl = nnew;
		__syncthreads();
	}
}

#ifndef X3_EMU
__global__ void __launch_bounds__(X3_SEG_THREADS) x3_segscan_kernel(X3SegArgs a) { x3_segscan_body<false>(a); }
__global__ void __launch_bounds__(X3_SEG_THREADS) x3_segscan_big_kernel(X3SegArgs a) { x3_segscan_body<true>(a); }
__global__ void __launch_bounds__(X3_SEG_THREADS) x3_segrefine_kernel(X3SegArgs a) { x3_segrefine_body(a); }
int x3_scan_seg_refine_launch(const X3SegArgs &a, uint32_t nchunks, hipStream_t st)
{
	hipLaunchKernelGGL(x3_segrefine_kernel, dim3(nchunks), dim3(X3_SEG_THREADS), 0, st, a);
	HIPCHK(hipGetLastError());
	return X3H_OK;
}
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
static void segrefine_tramp(void *p) { x3_segrefine_body(*(const X3SegArgs *)p); }
int x3_scan_seg_refine_launch(const X3SegArgs &a, uint32_t nchunks, hipStream_t)
{
	x3emu_launch(segrefine_tramp, (void *)&a, dim3(nchunks), dim3(X3_SEG_THREADS));
	return X3H_OK;
}
int x3_scan_seg_launch(const X3SegArgs &a, uint32_t nchunks, hipStream_t)
{
	x3emu_launch(a.gmf ? segscan_big_tramp : segscan_tramp, (void *)&a, dim3(nchunks), dim3(X3_SEG_THREADS));
	return X3H_OK;
}
#endif

/* One workgroup per chunk pays off once there are enough chunks to fill the chip.  -> 0: no (the chip-wide sort of scan2.hip), 1: yes, the
 * chunk's level counters fit LDS, 2: yes, with the counters in global memory (chunks up to X3_SEG_MAXLEN_BIG; needs enough of them that a
 * quarter of the CUs are busy with one chunk each for the whole scan).  X3H_SEG_MIN=n moves the first threshold (tests), 0 turns it off. */
int x3_scan_seg_applies(uint32_t nchunks, uint64_t max_len, uint64_t padded_total)
{
	uint32_t min_streams = X3_SEG_MIN_STREAMS, min_big = X3_SEG_MIN_STREAMS_BIG;
	bool forced = false;
	if (const char *e = getenv("X3H_SEG_MIN")) { const int v = atoi(e); if (v >= 1) { min_streams = min_big = (uint32_t)v; forced = true; } else if (v == 0 && *e == '0') return 0; }
	/* between 48 and ~70 chunks the chip-wide sort can be the faster one: a workgroup per chunk is a chain whose length is the CHUNK (0.2 ms + 9.2 ns per byte of
	 * text, in rounds of 256 chunks), the chip-wide form streams the padded layout of the BATCH (0.9 ms + 45 ns per KB): 64 chunks of 159 KB take 1.67 against 1.50 ms,
	 * 80 chunks of 127 KB 1.39 against 1.50 (round 4: the dickens-sized bytes as 48 .. 256 chunks; profiles/r04_midsize_marks_and_timeline.txt) */
	if (!forced && padded_total && nchunks >= min_streams && max_len <= X3_SEG_MAXLEN) {
		const double seg_ms = (0.2 + 9.2e-6 * (double)max_len) * (double)((nchunks + 255) / 256);
		const double chip_ms = 0.9 + 4.5e-8 * (double)padded_total;
		if (chip_ms < seg_ms) return 0;
	}
	uint64_t small_max = X3_SEG_MAXLEN;
	if (const char *e = getenv("X3H_SEG_SMALL_MAX")) { const long long v = atoll(e); if (v >= 0) small_max = (uint64_t)v < X3_SEG_MAXLEN ? (uint64_t)v : X3_SEG_MAXLEN; } /* (tests: force the global-memory form) */
	if (nchunks >= min_streams && max_len <= small_max) return 1;
	if (nchunks >= min_big && max_len <= X3_SEG_MAXLEN_BIG) return 2;
	return 0;
}
