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
	const uint32_t *S4;          /* list 4: element q stands for position q - 3 */
	const uint32_t *active_j;    /* index of each active position in S4 */
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
	for (uint32_t base = a.active_j[idx] + 1; base < a.total && !done; base += X3_WAVE) {
		const uint32_t j = base + lane;
		uint32_t s = j < a.total ? a.S4[j] - 3u : NONE32; /* q < 3 wraps to a huge value: fails the window test like any out-of-class entry */
		uint32_t lcp = 0;
		bool inwin = false;
		if (s != NONE32 && s <= wend && load_gram(b, s, 4) == g) { inwin = true; lcp = 4; }
		/* extend word by word while some lane still matches everything so far (typical common prefixes are short: one or two rounds
		 * instead of seven unconditional gram loads per candidate) */
		for (uint32_t k = 1; k < 8; k++) {
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
	for (int i = 0; i < 8; i++) CHK(B.a[i].reserve((P + 8) * 4));
	CHK(B.misc.reserve(64));
	uint32_t *keys = B.a[0].as<uint32_t>(), *iota = B.a[1].as<uint32_t>();
	/* K = T+1 for almost every position (its first byte occurs more than T times in its window); the others are marked in a bitmap
	 * (small enough to stay cached) and only they have an exact K in `kexact`: the level passes below touch no per-position word
	 * unless they have to, and m[] is written in place (one byte, only when a level passes) */
	uint32_t *kexact = B.a[4].as<uint32_t>();
	CHK(B.a[8].reserve((P / 32 + 2) * 4));
	uint32_t *rare = B.a[8].as<uint32_t>();
	HIPCHK(hipMemsetAsync(rare, 0, (P / 32 + 2) * 4, st));
	HIPCHK(hipMemsetAsync(d_m, 0, P, st));
	uint32_t *act = B.a[5].as<uint32_t>(), *act_k = B.a[6].as<uint32_t>(), *act_j = B.a[7].as<uint32_t>();
	uint32_t *d_nact = B.misc.as<uint32_t>();
	HIPCHK(hipMemsetAsync(d_nact, 0, 4, st));
	if (window > (1u << 24)) return X3H_E_ARG; /* K is kept in 24 bits */
	/* K = min(T+1, count_0) and count_0 <= ncand: any T >= ncand behaves like T = ncand */
	const uint32_t Tu = (uint32_t)T > ncand ? ncand : (uint32_t)T, Pn = (uint32_t)P;

	/* element q: key = the four bytes ENDING at q (bytes before the buffer read as 0), value = q */
	x3_foreach(P, st, X3_LAMBDA(size_t q) {
		iota[q] = (uint32_t)q;
		keys[q] = q >= 3 ? __builtin_bswap32(load_gram(d_bytes, q - 3, 4)) : __builtin_bswap32(load_gram(d_bytes, 0, 4) << (8 * (3 - (uint32_t)q)));
	});
	uint32_t *kin = keys, *vin = iota, *ks = B.a[2].as<uint32_t>(), *S = B.a[3].as<uint32_t>();
	for (uint32_t l = 1; l <= 4; l++) {
		/* pass l of the LSD sort: the list is now ordered by the l-gram starting at p = q-(l-1) (stable: ascending positions inside a class) */
		/* (rocPRIM 4.2's small-input merge path builds its digit mask with 1 << end_bit, which is wrong for end_bit == 32: below its
		 * switch-over the last pass is a full-key stable sort of the pass-3 list instead -- the same order, and the cost does not matter there) */
		if (l == 4 && P <= ((size_t)8 << 20)) CHK(x3p_sort_pairs(tmp, kin, ks, vin, S, P, 32, st));
		else CHK(x3p_sort_pairs_bits(tmp, kin, ks, vin, S, P, 8 * ((int)l - 1), 8 * (int)l, st));
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
						uint32_t lo = 0, hi = nc;
						while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (d_chunks[mid].byte_off <= p) lo = mid; else hi = mid; }
						if (p - (uint32_t)d_chunks[lo].byte_off >= d_chunks[lo].len) return;
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
	S = vin; /* list 4 (the last pass's output) */

	/* active positions: one wavefront each over the in-window candidates of its 4-gram class (S still holds list 4) */
	uint32_t nact = 0;
	HIPCHK(hipMemcpyAsync(&nact, d_nact, 4, hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	X3WalkArgs wa;
	wa.bytes = d_bytes; wa.S4 = S; wa.active_j = act_j; wa.active = act; wa.active_k = act_k; wa.nactive = d_nact; wa.m = d_m;
	wa.total = Pn; wa.ncand = ncand;
	launch_walk(wa, nact, st);
	HIPCHK(hipGetLastError());
	return X3H_OK;
}
