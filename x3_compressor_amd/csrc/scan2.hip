/*
 * scan2.hip -- K1 v2: the forward-window best-match scan without the O(N*W) sweep (reference: backend.c:56-78).
 *
 * What K2 needs from the scan is one number per position (scan.hip header):
 *      m[p] = max{ i : count_i(p) >= K(p) },   K(p) = min(T+1, count_0(p)),   (0 when T <= 0 or count_0(p) < 2)
 * where count_i(p) = #{ s in [p+1, p+W-33] : bytes p..p+i equal bytes s..s+i } (backend.c:62-74).
 * count_i(p) >= K  <=>  the K-th NEXT OCCURRENCE of the (i+1)-gram at p lies inside the window.  So:
 *   1. sort all positions of the (zero padded) batch by their 1-, 2-, 3- and 4-gram with a stable radix sort: inside a class
 *      the positions are ascending.  The four orders are the four successive 8-bit passes of ONE LSD radix sort: element q
 *      carries the key byte[q] | byte[q-1] << 8 | byte[q-2] << 16 | byte[q-3] << 24; after pass l the list is ordered by the low
 *      l key bytes, i.e. by the l-gram that STARTS at p = q-(l-1), with q (hence p) ascending inside a class;
 *   2. levels i = 0..3 are O(1): entry j of list l is position p; look at entry j+K -- same gram and still <= p+W-33 ?
 *      (count_0 itself is only needed when it is <= T: a binary search over at most T+1 list entries);
 *   3. only positions whose 4-gram already repeats K times in their window ("active") go deeper: one wavefront per active
 *      position sweeps the candidates of its 4-gram class that lie in the window -- ONE LANE PER CANDIDATE, coalesced reads of
 *      the class list, a 32-byte look-ahead of p broadcast from LDS -- and builds count_4..31 with ballot/popcount;
 *   4. ... unless the class is DENSE around the position (more than X3_WALK_DENSE members inside its window: zero runs, sparse
 *      16-bit samples, periodic data), where that sweep would cost O(W) per position.  Dense classes are refined one more byte at a
 *      time instead (a stable sort of the surviving elements by (class, next byte): classes split, positions stay ascending inside
 *      them), the level test is the same O(1) "K-th next occurrence of my class" lookup, and as soon as a position's class is no
 *      longer dense around it, the sweep of step 3 finishes it inside that (smaller) class.  At most 28 such levels, each over a
 *      list that only holds the classes still dense somewhere.
 * The result is exactly the reference's (verified against the brute-force kernel of scan.hip and the oracle); the cost no
 * longer depends on W.  The padding zeros of every chunk are ordinary positions of the sort, and a chunk's window never reaches
 * the next chunk's bytes (slots are W + X3_PAD_EXTRA apart), so one global sort serves a whole batch.
 */
#include "x3_host.h"

#include <vector>
#include <stdio.h>
#include <stdlib.h>

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
	const uint32_t *S4;          /* the class-sorted list: element L stands for position L - back (list 4 holds END positions: back = 3) */
	const uint32_t *cls;         /* ... and the class of every entry (list 4: its 4-gram; refined lists: a class id) */
	uint32_t back;
	uint32_t kwords;             /* whole 4-byte words the members of a class are known to share (class gram length / 4) */
	const uint32_t *padbits;     /* one bit per position: inside a chunk's zero padding */
	uint32_t pad_dropped;        /* the list holds no padding positions: those inside the window are counted, not met */
	const uint32_t *dataend;     /* per 256-byte block of the layout: first padding position of the chunk it lies in */
	uint32_t npos;
	const uint32_t *active_j;    /* index of each active position in the list */
	const uint32_t *active;      /* positions to walk */
	const uint32_t *active_k;    /* their K */
	const uint32_t *nactive;
	uint8_t *m;
	uint32_t total;              /* sorted entries */
	uint32_t seg;                /* the list is a concatenation of per-chunk lists (scan3.hip): chunk c's entries are [byte_off, byte_off + len + 3), no padding positions */
	uint32_t ncand;              /* W - 33 */
};

#define X3_WALK_WAVES 4
#ifndef X3_WALK_DENSE
#define X3_WALK_DENSE 2048u /* a position with more members of its class than this inside its window is not swept: its class is refined */
#endif

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
	const uint32_t myj = a.active_j[idx], g = a.cls[myj];
	const uint64_t wend = (uint64_t)p + a.ncand;
	uint32_t cnt[28]; /* count_4 .. count_31 (wave-uniform) */
#pragma unroll
	for (int i = 0; i < 28; i++) cnt[i] = 0;
	uint32_t done = 0, npad = 0;
	uint32_t lend = a.total; /* one past the last list entry a walk from here may read */
	if (a.seg) { const uint32_t e = a.dataend[p >> 8] + 3u; lend = e < lend ? e : lend; }
	for (uint32_t base = myj + 1; base < lend && !done; base += X3_WAVE) {
		const uint32_t j = base + lane;
		uint32_t s = j < lend ? a.S4[j] - a.back : NONE32; /* an end position < back wraps to a huge value: fails the window test like any out-of-class entry */
		uint32_t lcp = 0;
		bool inwin = false;
		if (s != NONE32 && s <= wend && a.cls[j] == g) { inwin = true; lcp = 4 * a.kwords; } /* same class: the class's gram in common */
		/* A candidate inside the chunk's zero padding: every later in-window member of the class is padding too (positions ascend, the padding
		 * is the slot's tail), and each of them shares exactly p's leading zero run with p.  They are not swept but counted (below). */
		const bool padc = inwin && ((a.padbits[s >> 5] >> (s & 31)) & 1u);
		const uint64_t padm = x3_ballot(padc);
		if (padm) {
			const uint32_t fp = x3_readlane_u32(s, (uint32_t)x3_ctz64(padm)); /* the first padding position of the class inside the window */
			npad = (uint32_t)(wend - fp + 1);
			if (padc) inwin = false; /* ends the walk like any out-of-window entry */
		}
		/* extend word by word while some lane still matches everything so far (typical common prefixes are short: one or two rounds
		 * instead of seven unconditional gram loads per candidate) */
		for (uint32_t k = a.kwords; k < 8; k++) {
			const bool alive = inwin && lcp == 4 * k;
			if (!x3_ballot(alive)) break;
			if (alive) {
				const uint32_t x = load_gram(b, (uint64_t)s + 4 * k, 4) ^ look[wv][k];
				lcp = x ? 4 * k + ((uint32_t)x3_ctz32(x) >> 3) : 4 * k + 4;
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
	if (a.pad_dropped && wend < a.npos && ((a.padbits[wend >> 5] >> (wend & 31)) & 1u)) { /* the window reaches the padding */
		npad = (uint32_t)(wend - a.dataend[p >> 8] + 1);
	}
	if (npad) { /* the padding members: common prefix with p = p's run of zero bytes (<= 32) */
		uint32_t zr = 32;
		for (int k = 7; k >= 0; k--) { const uint32_t wd = look[wv][k]; if (wd) zr = 4u * (uint32_t)k + ((uint32_t)x3_ctz32(wd) >> 3); }
#pragma unroll
		for (int i = 0; i < 28; i++) if ((uint32_t)i + 4 < zr) cnt[i] += npad;
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

/* m[] for every position of every chunk.  d_bytes/d_m use the padded layout (X3Chunk::byte_off), `total` = padded bytes.
 * All passes run over the SORTED lists: entry j of list l is position S_l[j]; its class neighbours are the adjacent entries
 * (the sorted key array says where the class ends), so "K-th next occurrence inside the window" is  S_l[j+K] <= p + W-33
 * with equal keys -- sequential reads of the list plus one random read/write of a per-position state word {K, m}.
 * Padding positions are processed like any other (their results are never read). */
int x3_scan_v2_run(X3Scan2Bufs &B, DevBuf &tmp, hipStream_t st, int nchunks, const X3Chunk *h_chunks, const X3Chunk *d_chunks,
                   const uint8_t *d_bytes, uint8_t *d_m, uint64_t total, uint32_t window, int32_t T)
{
	const uint32_t nc = (uint32_t)nchunks;
	if (total >= 0xFFFFFF00ull) return X3H_E_ARG;
	const size_t P = (size_t)total;
	const uint32_t ncand = window > X3_MAXLEN + 1 ? window - X3_MAXLEN - 1 : 0;
	uint64_t nsum = 0;
	for (uint32_t c = 0; c < nc; c++) nsum += h_chunks[c].len;
	if (T <= 0 || ncand == 0 || nsum == 0) { /* backend.c:76,99: no threshold can be met -> length 1 everywhere */
		HIPCHK(hipMemsetAsync(d_m, 0, P, st));
		return X3H_OK;
	}
	uint64_t max_len = 0;
	for (uint32_t c = 0; c < nc; c++) if (h_chunks[c].len > max_len) max_len = h_chunks[c].len;
	const int seg_form = x3_scan_seg_applies(nc, max_len, total); /* many chunks: one workgroup per chunk sorts and tests its own positions (scan3.hip) */
	const bool seg = seg_form != 0;
	for (int i = 0; i < 8; i++) CHK(B.a[i].reserve((P + 8) * (seg && i < 2 ? 8 : 4)));
	CHK(B.a[9].reserve(((size_t)P / 32 + 2) * 4 + 64)); /* padding bitmap */
	CHK(B.misc.reserve(64));
	uint32_t *keys = B.a[0].as<uint32_t>(), *iota = B.a[1].as<uint32_t>();
	/* K = T+1 for almost every position (its first byte occurs more than T times in its window); the others are marked in a bitmap
	 * (small enough to stay cached) and only they have an exact K in `kexact`: the level passes below touch no per-position word
	 * unless they have to, and m[] is written in place (one byte, only when a level passes) */
	uint32_t *kexact = B.a[4].as<uint32_t>();
	CHK(B.a[8].reserve((P / 32 + 2) * 4));
	uint32_t *rare = B.a[8].as<uint32_t>();
	HIPCHK(hipMemsetAsync(rare, 0, (P / 32 + 2) * 4, st));
	if (!seg) HIPCHK(hipMemsetAsync(d_m, 0, P, st));
	uint32_t *act = B.a[5].as<uint32_t>(), *act_k = B.a[6].as<uint32_t>(), *act_j = B.a[7].as<uint32_t>();
	uint32_t *d_nact = B.misc.as<uint32_t>(), *d_dense = d_nact + 1;
	HIPCHK(hipMemsetAsync(d_nact, 0, 8, st));
	/* positions in a chunk's padding are ordinary occurrences (the zeros take part in the window, x3.c:579,590) but never QUERIES (their m
	 * is not read): one bit per position */
	uint32_t *padbits = B.a[9].as<uint32_t>();
	x3_foreach(P / 32 + 1, st, X3_LAMBDA(size_t w) {
		const uint64_t p0 = (uint64_t)w * 32;
		uint32_t lo = 0, hi = nc;
		while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (d_chunks[mid].byte_off <= p0) lo = mid; else hi = mid; }
		uint32_t bits = 0;
		for (uint32_t b = 0; b < 32; b++) {
			const uint64_t p = p0 + b;
			if (lo + 1 < nc && d_chunks[lo + 1].byte_off <= p) lo++;
			if (p - d_chunks[lo].byte_off >= d_chunks[lo].len) bits |= 1u << b;
		}
		padbits[w] = bits;
	});
	/* ... and, per 256-byte block of the layout (a block lies inside ONE chunk's slot: slots are 256-byte aligned), where that chunk's
	 * data ends = its first padding position */
	CHK(B.a[21].reserve(((size_t)P / 256 + 2) * 4));
	uint32_t *dataend = B.a[21].as<uint32_t>();
	x3_foreach(P / 256 + 1, st, X3_LAMBDA(size_t b) {
		const uint64_t p0 = (uint64_t)b * 256;
		uint32_t lo = 0, hi = nc;
		while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (d_chunks[mid].byte_off <= p0) lo = mid; else hi = mid; }
		dataend[b] = (uint32_t)(d_chunks[lo].byte_off + d_chunks[lo].len);
	});
	uint32_t dense_at = X3_WALK_DENSE;
	bool dense_env = false;
	if (const char *e = getenv("X3H_WALK_DENSE")) { const int v = atoi(e); if (v >= 1) { dense_at = (uint32_t)v; dense_env = true; } } /* tuning / tests */
	if (window > (1u << 24)) return X3H_E_ARG; /* K is kept in 24 bits */
	/* K = min(T+1, count_0) and count_0 <= ncand: any T >= ncand behaves like T = ncand */
	const uint32_t Tu = (uint32_t)T > ncand ? ncand : (uint32_t)T, Pn = (uint32_t)P;
	{
		/* per-chunk lists of chunks up to 256 KiB: the chunk's own workgroup refines its dense classes (x3_segrefine_kernel), which is cheaper than
		 * sweeping a class's members position by position as soon as a class has the K members a level needs at all -- every class with a passing
		 * member is refined (1024 x 256 KiB of text + Zipf bytes at -t 256: scan 21 -> 15 ms; mr-like samples: 57 -> 30 ms) */
		const char *e = getenv("X3H_SEG_REFINE");
		if (seg_form == 1 && !dense_env && !(e && e[0] == '0') && dense_at > Tu + 1u) dense_at = Tu + 1u;
	}

	uint32_t *S = nullptr;
	const uint32_t *ks4 = nullptr;
	X3SegArgs seg_args = {};
	if (seg) {
		X3SegArgs &ga = seg_args;
		ga.bytes = d_bytes; ga.chunks = d_chunks; ga.la = B.a[0].as<uint2>(); ga.lb = B.a[1].as<uint2>();
		ga.S4 = B.a[2].as<uint32_t>(); ga.K4 = B.a[3].as<uint32_t>(); ga.m = d_m; ga.rare = rare; ga.kexact = kexact;
		ga.act = act; ga.act_k = act_k; ga.act_j = act_j; ga.nact = d_nact;
		ga.window = window; ga.ncand = ncand; ga.Tu = Tu; ga.dense_at = dense_at;
		ga.gmf = nullptr;
		CHK(B.a[13].reserve((size_t)nc * 4 + 64)); /* (a[10..20] are the chip-wide refinement's; it does not run behind the per-chunk one) */
		HIPCHK(hipMemsetAsync(B.a[13].p, 0, (size_t)nc * 4, st));
		ga.dense_chunk = B.a[13].as<uint32_t>();
		if (seg_form == 2) { CHK(B.a[23].reserve(P / 4 + 64)); HIPCHK(hipMemsetAsync(B.a[23].p, 0, P / 4 + 64, st)); ga.gmf = B.a[23].as<uint32_t>(); }
		if (getenv("X3H_DEBUG")) fprintf(stderr, "[x3h] scan: one workgroup per chunk (%u chunks, longest %llu)\n", nc, (unsigned long long)max_len);
		ga.prof = nullptr;
		ga.la_lds = 1u; ga._pad = 0u;
		if (const char *e = getenv("X3H_SEG_LA_LDS")) ga.la_lds = atoi(e) != 0;
		const bool prof = getenv("X3H_SEG_PROF") != nullptr;
		if (prof) { CHK(B.a[22].reserve(128)); HIPCHK(hipMemsetAsync(B.a[22].p, 0, 128, st)); ga.prof = B.a[22].as<uint64_t>(); }
		CHK(x3_scan_seg_launch(ga, nc, st));
		if (prof) {
			uint64_t h[16];
			HIPCHK(hipMemcpyAsync(h, B.a[22].p, 128, hipMemcpyDeviceToHost, st));
			HIPCHK(hipStreamSynchronize(st));
			fprintf(stderr, "[x3h] scan3 kcycles per chunk: keys %llu | passes (+ level of the list read) %llu %llu %llu %llu | level 4 %llu\n", (unsigned long long)(h[0] / nc / 1000),
			        (unsigned long long)(h[1] / nc / 1000), (unsigned long long)(h[2] / nc / 1000), (unsigned long long)(h[3] / nc / 1000), (unsigned long long)(h[4] / nc / 1000),
			        (unsigned long long)(h[5] / nc / 1000));
		}
		S = ga.S4; ks4 = ga.K4;
	} else {
	/* element q: key = the four bytes ENDING at q (bytes before the buffer read as 0), value = q */
	x3_foreach(P, st, X3_LAMBDA(size_t q) {
		iota[q] = (uint32_t)q;
		keys[q] = q >= 3 ? __builtin_bswap32(load_gram(d_bytes, q - 3, 4)) : __builtin_bswap32(load_gram(d_bytes, 0, 4) << (8 * (3 - (uint32_t)q)));
	});
	uint32_t *kin = keys, *vin = iota, *ks = B.a[2].as<uint32_t>();
	S = B.a[3].as<uint32_t>();
	for (uint32_t l = 1; l <= 4; l++) {
		/* pass l of the LSD sort: the list is now ordered by the l-gram starting at p = q-(l-1) (stable: ascending positions inside a class) */
		CHK(x3p_sort_pairs_bits(tmp, kin, ks, vin, S, P, 8 * ((int)l - 1), 8 * (int)l, st)); /* prims.hip: histogram, scan, ranked scatter */
		const uint32_t msk = l >= 4 ? 0xFFFFFFFFu : ((1u << (8 * l)) - 1), back = l - 1;
		const uint32_t *ksc = ks, *Sc = S;
		if (l == 1) {
			/* K = min(T+1, count_0): is the (T+1)-th next occurrence of the byte inside the window?  else count them (binary search) */
			x3_foreach(P, st, X3_LAMBDA(size_t j) {
				const uint32_t p = Sc[j], kj = ksc[j] & msk;
				const uint64_t wend = (uint64_t)p + ncand;
				const uint64_t u = (uint64_t)j + Tu + 1;
				uint32_t K;
				if (u < Pn && (ksc[u] & msk) == kj && Sc[u] <= wend) K = Tu + 1;
				else {
					uint32_t a = 0, bnd = Tu; /* predicate true at a */
					if ((uint64_t)j + bnd >= Pn) bnd = Pn - 1 - (uint32_t)j;
					while (a < bnd) {
						const uint32_t mid = (a + bnd + 1) >> 1;
						if ((ksc[j + mid] & msk) == kj && Sc[j + mid] <= wend) a = mid; else bnd = mid - 1;
					}
					K = a; /* == count_0 */
				}
				if (K != Tu + 1) { kexact[p] = K; atomicOr(&rare[p >> 5], 1u << (p & 31)); }
			});
		} else {
			x3_foreach(P, st, X3_LAMBDA(size_t j) {
				const uint32_t q = Sc[j];
				if (q < back) return; /* the l-gram would start before the buffer */
				const uint32_t p = q - back;
				/* count_{l-1}(p) >= K?  First with K = T+1, from the list alone: if even the (T+1)-th next occurrence is inside the window,
				 * the level passes whatever K is (K <= T+1).  Only if that fails can a smaller K matter, and only marked positions have one.
				 * (A level can only pass if the previous one did -- occurrences of the longer gram are occurrences of the shorter -- so the
				 * levels need not look at each other's result.) */
				uint32_t K = Tu + 1;
				uint64_t u = (uint64_t)j + K;
				bool pass = u < Pn && ((ksc[u] ^ ksc[j]) & msk) == 0 && (uint64_t)Sc[u] <= (uint64_t)q + ncand; /* both sides carry the same +back */
				if (!pass) {
					if (!((rare[p >> 5] >> (p & 31)) & 1u)) return;
					K = kexact[p];
					if (K < 2) return; /* count_0 < 2 */
					u = (uint64_t)j + K;
					pass = u < Pn && ((ksc[u] ^ ksc[j]) & msk) == 0 && (uint64_t)Sc[u] <= (uint64_t)q + ncand;
				}
				if (pass) {
					d_m[p] = (uint8_t)(l - 1);
					if (l == 4) { /* count_3 >= K: deeper levels need the candidates themselves -- unless p is padding (never read) */
						if ((padbits[p >> 5] >> (p & 31)) & 1u) return;
						/* more than X3_WALK_DENSE members of the class inside the window: no sweep, the class is refined instead (below) */
						const uint64_t ud = (uint64_t)j + dense_at;
						if (ud < Pn && ksc[ud] == ksc[j] && (uint64_t)Sc[ud] <= (uint64_t)q + ncand && !((padbits[(Sc[ud] - 3u) >> 5] >> ((Sc[ud] - 3u) & 31)) & 1u)) { *d_dense = 1u; return; }
						const uint32_t slot = atomicAdd(d_nact, 1u);
						act[slot] = p; act_k[slot] = K; act_j[slot] = (uint32_t)j;
					}
				}
			});
		}
		uint32_t *t;
		t = kin; kin = ks; ks = t;
		t = vin; vin = S; S = t;
	}
	S = vin; /* list 4 (the last pass's output); kin holds its sorted keys */
	ks4 = kin;
	}

	/* active positions: one wavefront each over the in-window candidates of its 4-gram class */
	uint32_t hcnt[2] = { 0, 0 };
	HIPCHK(hipMemcpyAsync(hcnt, d_nact, 8, hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	X3WalkArgs wa;
	wa.bytes = d_bytes; wa.S4 = S; wa.cls = ks4; wa.back = 3; wa.kwords = 1; wa.active_j = act_j; wa.active = act; wa.active_k = act_k; wa.nactive = d_nact; wa.m = d_m;
	wa.total = Pn; wa.ncand = ncand; wa.padbits = padbits; wa.pad_dropped = seg ? 1u : 0u; wa.dataend = dataend; wa.npos = Pn; wa.seg = seg ? 1u : 0u;
	launch_walk(wa, hcnt[0], st);
	HIPCHK(hipGetLastError());
	if (!hcnt[1]) return X3H_OK;
	if (seg_form == 1) { /* dense classes of per-chunk lists: refined by the chunk's own workgroup (scan3.hip); X3H_SEG_REFINE=0: the chip-wide refinement below */
		const char *e = getenv("X3H_SEG_REFINE");
		if (!(e && e[0] == '0')) {
			if (getenv("X3H_DEBUG")) fprintf(stderr, "[x3h] scan: dense classes refined per chunk\n");
			return x3_scan_seg_refine_launch(seg_args, nc, st);
		}
	}

	/* ---- dense classes: refine them byte by byte (header, step 4).  Element = START position from here on. ---- */
	wa.seg = 0; /* the refined lists below are chip-wide again */
	if (seg) { /* per-chunk lists end at len + 3 entries: what lies behind them in a slot becomes entries that are never a query, never inside a window */
		uint32_t *S4w = S, *K4w = (uint32_t *)ks4;
		x3_foreach(P, st, X3_LAMBDA(size_t j) { if ((uint32_t)j >= dataend[j >> 8] + 3u) { S4w[j] = 2u; K4w[j] = 0xFFFFFFFFu; } });
	}
	for (int i = 10; i < 21; i++) CHK(B.a[i].reserve((P + 8) * 4));
	uint32_t *Lp = B.a[10].as<uint32_t>(), *Lc = B.a[11].as<uint32_t>(), *Np = B.a[12].as<uint32_t>(), *Nc = B.a[13].as<uint32_t>();
	uint32_t *keep = B.a[14].as<uint32_t>(), *flg = B.a[15].as<uint32_t>(), *scn = B.a[16].as<uint32_t>();
	uint32_t *key = B.a[17].as<uint32_t>(), *keys_ = B.a[18].as<uint32_t>(), *idx = B.a[19].as<uint32_t>(), *perm = B.a[20].as<uint32_t>();
	HIPCHK(hipMemsetAsync(keep, 0, (P + 8) * 4, st));
	/* list 4 with start identity and class ids (index of the class's first entry) */
	{
		const uint32_t *k4 = ks4, *S4 = S;
		x3_foreach(P, st, X3_LAMBDA(size_t j) { flg[j] = (j > 0 && k4[j] != k4[j - 1]) ? (uint32_t)j : 0u; Lp[j] = S4[j] - 3u; /* (an end position < 3 wraps: never a query, never inside a window) */ });
		CHK(x3p_incl_max_scan(tmp, flg, Lc, P, st));
	}
	size_t n = P;
	for (uint32_t len = 4; len <= X3_MAXLEN && n > 0; len++) {
		const uint32_t nn = (uint32_t)n, stamp = len;
		const bool pad_dropped = seg || len > 4; /* the chip-wide list 4 still holds the padding positions; the first compaction drops them and they are counted analytically from then on (per-chunk lists never held them) */
		HIPCHK(hipMemsetAsync(d_nact, 0, 8, st));
		/* the level test (gram length len): K-th next member of my class inside my window?  Then: finished by a sweep, or still dense */
		x3_foreach(n, st, X3_LAMBDA(size_t j) {
			const uint32_t p = Lp[j];
			if (p >= Pn || ((padbits[p >> 5] >> (p & 31)) & 1u)) return; /* padding (or wrapped): an occurrence, not a query */
			uint32_t K = Tu + 1;
			if ((rare[p >> 5] >> (p & 31)) & 1u) { K = kexact[p]; if (K < 2) return; }
			const uint32_t cj = Lc[j];
			/* padding members of the class are not in these lists: if p's window reaches its chunk's padding and p's gram is all zeros, every
			 * padding position inside the window is an occurrence */
			const uint64_t wend = (uint64_t)p + ncand;
			uint32_t cpad = 0;
			if (pad_dropped && wend < Pn && ((padbits[wend >> 5] >> (wend & 31)) & 1u)) {
				uint32_t nz = 0;
				for (uint32_t k = 0; k < len; k += 4) nz |= load_gram(d_bytes, (uint64_t)p + k, len - k >= 4 ? 4 : len - k);
				if (!nz) cpad = (uint32_t)(wend - dataend[p >> 8] + 1);
			}
			if (cpad < K) {
				const uint64_t u = (uint64_t)j + (K - cpad);
				if (!(u < nn && Lc[u] == cj && (uint64_t)Lp[u] <= wend)) return;
			}
			if (len > 4) d_m[p] = (uint8_t)(len - 1); /* (level 3 was stored by the pass above) */
			/* a class that was dense at length 4 is refined to the end: sweeping its members later (deep common prefixes, hundreds of
			 * candidates each) measured several times slower than carrying them through the remaining levels */
			const uint64_t ud = (uint64_t)j + dense_at;
			const bool dense = len > 4 || (ud < nn && Lc[ud] == cj && (uint64_t)Lp[ud] <= wend && !((padbits[Lp[ud] >> 5] >> (Lp[ud] & 31)) & 1u));
			if (dense && len < X3_MAXLEN) { keep[cj] = stamp; *d_dense = 1u; } /* (every writer stores the same value) */
			else if (len > 4 && len < X3_MAXLEN) { /* (at length 4 the pass above queued the others already) */
				const uint32_t slot = atomicAdd(d_nact, 1u);
				act[slot] = p; act_k[slot] = K; act_j[slot] = (uint32_t)j;
			}
		});
		HIPCHK(hipMemcpyAsync(hcnt, d_nact, 8, hipMemcpyDeviceToHost, st));
		HIPCHK(hipStreamSynchronize(st));
		if (len > 4 && hcnt[0]) {
			wa.S4 = Lp; wa.cls = Lc; wa.back = 0; wa.kwords = len / 4; wa.total = nn; wa.pad_dropped = pad_dropped ? 1u : 0u;
			launch_walk(wa, hcnt[0], st);
			HIPCHK(hipGetLastError());
		}
		if (!hcnt[1] || len == X3_MAXLEN) break;
		/* keep the classes that are still dense somewhere; number them 0, 1, 2, ... in list order */
		x3_foreach(n, st, X3_LAMBDA(size_t j) { const uint32_t p = Lp[j]; flg[j] = (keep[Lc[j]] == stamp && p < Pn && !((padbits[p >> 5] >> (p & 31)) & 1u)) ? 1u : 0u; });
		CHK(x3p_excl_scan(tmp, flg, scn, n, st));
		x3_foreach(n, st, X3_LAMBDA(size_t j) { if (flg[j]) { const uint32_t d = scn[j]; Np[d] = Lp[j]; Nc[d] = Lc[j]; } });
		uint32_t nlive = 0;
		HIPCHK(hipMemcpyAsync(&nlive, scn + n, 4, hipMemcpyDeviceToHost, st));
		HIPCHK(hipStreamSynchronize(st));
		if (getenv("X3H_DEBUG")) fprintf(stderr, "[x3h] scan: gram length %u: %zu list entries, %u swept, %u stay in dense classes\n", len, n, hcnt[0], nlive);
		n = nlive;
		if (!n) break;
		x3_foreach(n, st, X3_LAMBDA(size_t i) { flg[i] = (i > 0 && Nc[i] != Nc[i - 1]) ? 1u : 0u; });
		CHK(x3p_excl_scan(tmp, flg, scn, n, st)); /* scn[i] + flg[i] = ordinal of entry i's class */
		uint32_t ncls = 0;
		HIPCHK(hipMemcpyAsync(&ncls, scn + n, 4, hipMemcpyDeviceToHost, st));
		HIPCHK(hipStreamSynchronize(st));
		/* next level: stable sort by (class, byte behind the gram): classes split, positions stay ascending inside them */
		const uint32_t ext = len; /* the gram at p grows by bytes[p + len] */
		x3_foreach(n, st, X3_LAMBDA(size_t i) { key[i] = ((scn[i] + flg[i]) << 8) | (uint32_t)d_bytes[(uint64_t)Np[i] + ext]; idx[i] = (uint32_t)i; });
		CHK(x3p_sort_pairs(tmp, key, keys_, idx, perm, n, 8 + (ncls ? 32 - __builtin_clz(ncls) : 1), st));
		x3_foreach(n, st, X3_LAMBDA(size_t j) { Lp[j] = Np[perm[j]]; flg[j] = (j > 0 && keys_[j] != keys_[j - 1]) ? (uint32_t)j : 0u; });
		CHK(x3p_incl_max_scan(tmp, flg, Lc, n, st));
	}
	return X3H_OK;
}
