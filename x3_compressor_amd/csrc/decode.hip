/*
 * decode.hip -- the x3 decoder (reference: decompress() x3.c:285-353, decode_tag() x3.c:58-129, decode_match() x3.c:272-283,
 * arithmetic decoder ac.c:128-198, bit reader bio.c:30-42,74-103).
 *
 * Unlike the encoder, the decoder cannot run ahead of its models: which table is consulted next depends on the symbol
 * just decoded, so the path is one dependent chain per stream.  Mapping: one wavefront per stream; the chain's state
 * (interval, bit reader, contexts) is wave-uniform, and every linear table walk of the reference is a 64-lane sweep:
 *   index_of_value (ac.c:167-179)        -> inclusive wave scan of 64 frequencies + ballot for the first cum > value
 *   dict_get_index_by_tag (dict.c:174-183) -> ballot search of the move-to-front list
 *   dict_query_elem (dict.c:148-157)     -> exact hash lookup; an element is a (pos,len) reference into the OUTPUT
 *   dict_update_costs + qsort            -> move-to-front (x3_tables.h)
 * The chain is latency-bound (a dozen dependent global loads per step in the naive order), so the loop PREFETCHES the next step's
 * context state as soon as it is known: the pair looked up / inserted at the end of a hit step (x3.c:213-222) IS the (prev, context1)
 * pair the next step would look up, so its ordinal is carried over instead of looked up again -- and it needs no hash map at all: a
 * pair (context1, tag) exists exactly when `tag` is in the item list of context1 (x3.c:197-222 add both in the same step), so every
 * item of a context1 list carries the ordinal of its pair.  Both context headers plus the first 64 items of either list are loaded at
 * the end of a step: they are in flight while the next event symbol is decoded.
 * Streams of a batch decode concurrently (grid = streams).  The reference's unchecked 64x output buffer (x3.c:621) is
 * replaced by a capacity check (X3_ST_OUT_FULL), malformed input ends in X3_ST_CORRUPT instead of abort() (ac.c:178).
 */
#include "x3_tables.h"

#ifdef X3_DEC_PROFILE /* experiment builds: cycles per section of the hit path, reported in the unused event slots */
#define DPROF_T(var) const uint64_t var = x3_clock();
#define DPROF_ADD(acc, a, b) acc += (b) - (a);
#else
#define DPROF_T(var)
#define DPROF_ADD(acc, a, b)
#endif

/* The chain's state is wave-uniform, but every value that comes out of a (vector) load looks divergent to the compiler: pinning the
 * loaded words with readfirstlane keeps the interval arithmetic, the bit reader and the control flow on the scalar unit. */
__device__ static __forceinline__ uint64_t uni64(uint64_t v) { return ((uint64_t)x3_uniform((uint32_t)(v >> 32)) << 32) | x3_uniform((uint32_t)v); }
__device__ static __forceinline__ X3CtxHdr uni_hdr(const X3CtxHdr h) { X3CtxHdr r; r.off = x3_uniform(h.off); r.items = x3_uniform(h.items); r.cap = x3_uniform(h.cap); r.total = x3_uniform(h.total); return r; }

#ifndef X3_EMU
__device__ static __forceinline__ uint32_t dec_brev32(uint32_t v) { return __brev(v); }
#else
static inline uint32_t dec_brev32(uint32_t v) { uint32_t r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }
#endif

struct BitReader { /* bio.c:5-42: bit k of the stream = bit k mod 32 (LSB first) of the little-endian word k / 32 */
	const uint8_t *p, *end;
	uint64_t w;   /* unread bits, next one at bit 0 */
	uint32_t nb;  /* how many */
};

/* the next n (0..31) bits of the stream, first one most significant -- what n calls of get_bit shifted into mBuffer would leave */
__device__ static __forceinline__ uint32_t br_take(BitReader &r, uint32_t n)
{
	if (r.nb < n) { /* one more word (bio.c:10,35-39: past the last whole word the reader feeds 0x80000000) */
		uint32_t nx = 0x80000000u;
		if (r.end - r.p >= 4) { nx = x3_uniform((uint32_t)r.p[0] | (uint32_t)r.p[1] << 8 | (uint32_t)r.p[2] << 16 | (uint32_t)r.p[3] << 24); r.p += 4; }
		r.w |= (uint64_t)nx << r.nb;
		r.nb += 32;
	}
	if (!n) return 0;
	const uint32_t field = (uint32_t)r.w & (0xFFFFFFFFu >> (32 - n));
	r.w >>= n;
	r.nb -= n;
	return dec_brev32(field) >> (32 - n);
}

struct Dec { uint32_t lo, hi, buf; };

/* ac_decode_symbol's interval update + ac_decode_scale (ac.c:192-195,142-165) in closed form, like the encoder's chain (code2.hip):
 * all E1/E2/E3 shifts together are  s = clz(D) - 1 - carry  with D = hi - lo after narrowing; lo and the range scale by 2^s, and since
 * every kind of shift removes the same offset from mBuffer as from mLow,  buffer - low  scales too and takes the s new bits.
 * false: not an interval a valid stream can produce (the reference would spin in its E1/E2 loop or read garbage). */
__device__ static __forceinline__ bool dec_narrow(Dec &d, BitReader &r, uint32_t step, uint32_t cum_lo, uint32_t cum_hi)
{
	const uint32_t nlo = d.lo + step * cum_lo, nhi = d.lo + step * cum_hi - 1, D = nhi - nlo;
	if (D == 0 || ((nlo | nhi) >> 31) || nhi < nlo || d.buf < nlo || d.buf > nhi) return false;
	const uint32_t cz = (uint32_t)x3_clz32(D), t = 31u - cz;
	const uint32_t sh = cz - 1 - ((((nlo ^ nhi) >> t) & 1u) ^ 1u);
	const uint32_t lo = (nlo << sh) & 0x3FFFFFFFu;
	d.buf = lo + ((d.buf - nlo) << sh) + br_take(r, sh);
	d.hi = lo + ((D + 1) << sh) - 1;
	d.lo = lo;
	return true;
}

__device__ static __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) { (void)lane; return x3_wave_incl_scan_u32(v); }
/* value < incl  with  value = off / step (ac.c:128-131), without dividing:  off < incl * step  (step == 0: never, like a value out of range) */
__device__ static __forceinline__ bool dec_below(uint32_t off, uint32_t step, uint32_t incl) { return (uint64_t)off < (uint64_t)incl * step; }

/* find the symbol of a frequency array (global memory, `count` entries) that holds `value`; returns 0xFFFFFFFF if none */
__device__ static __forceinline__ uint32_t find_in_array(const uint32_t *freq, uint32_t count, uint32_t off, uint32_t step, uint32_t lane, uint32_t &cum_out, uint32_t &fq_out)
{
	uint32_t carry = 0;
	for (uint32_t base = 0; base < count; base += X3_WAVE) {
		const uint32_t i = base + lane;
		const uint32_t fq = i < count ? freq[i] : 0;
		const uint32_t incl = wave_incl_scan(fq, lane) + carry;
		const uint64_t mask = x3_ballot(i < count && dec_below(off, step, incl));
		if (mask) {
			const uint32_t l = (uint32_t)x3_ctz64(mask);
			fq_out = x3_readlane_u32(fq, l);
			cum_out = x3_readlane_u32(incl, l) - fq_out;
			return base + l;
		}
		carry = x3_readlane_u32(incl, X3_WAVE - 1);
	}
	return 0xFFFFFFFFu;
}

/* same over a context's item list; returns the list position */
__device__ static uint32_t find_in_ctx(const X3CtxHdr h, const uint64_t *pool, uint32_t off, uint32_t step, uint32_t lane, uint64_t first, uint32_t &cum_out, uint32_t &fq_out, uint32_t &tag_out)
{
	uint32_t carry = 0;
	if (h.items <= X3_WAVE) { /* the usual case as straight-line code: no load, so nothing here makes the wave wait for loads that are in flight */
		const uint32_t fq = (uint32_t)first; /* lanes beyond the list hold 0 */
		const uint32_t incl = wave_incl_scan(fq, lane);
		const uint64_t mask = x3_ballot(lane < h.items && dec_below(off, step, incl));
		if (!mask) return 0xFFFFFFFFu;
		const uint32_t l = (uint32_t)x3_ctz64(mask);
		fq_out = x3_readlane_u32(fq, l);
		cum_out = x3_readlane_u32(incl, l) - fq_out;
		tag_out = x3_readlane_u32((uint32_t)(first >> 32), l);
		return l;
	}
	for (uint32_t base = 0; base < h.items; base += X3_WAVE) {
		const uint32_t i = base + lane;
		const uint64_t it = base == 0 ? first : (i < h.items ? pool[(uint64_t)h.off + i] : 0); /* items [0, 64) were prefetched with the header */
		const uint32_t fq = (uint32_t)it;
		const uint32_t incl = wave_incl_scan(fq, lane) + carry;
		const uint64_t mask = x3_ballot(i < h.items && dec_below(off, step, incl));
		if (mask) {
			const uint32_t l = (uint32_t)x3_ctz64(mask);
			fq_out = x3_readlane_u32(fq, l);
			cum_out = x3_readlane_u32(incl, l) - fq_out;
			tag_out = x3_readlane_u32((uint32_t)(it >> 32), l);
			return base + l;
		}
		carry = x3_readlane_u32(incl, X3_WAVE - 1);
	}
	return 0xFFFFFFFFu;
}

/* position of `tag` in a context's item list (ctx_query_tag_item, context.c:20-40): all the model update needs -- no frequency sums */
__device__ static CtxQ ctx_find_tag(const X3CtxHdr h, const uint64_t *pool, uint32_t tag, uint32_t lane, uint64_t first)
{
	CtxQ q;
	q.found = 0; q.pos = 0; q.freq = 0; q.cum = 0;
	if (h.items <= X3_WAVE) { /* straight-line, no load (see find_in_ctx) */
		const uint64_t mask = x3_ballot(lane < h.items && (uint32_t)(first >> 32) == tag);
		if (mask) { q.found = 1; q.pos = (uint32_t)x3_ctz64(mask); }
		return q;
	}
	for (uint32_t base = 0; base < h.items; base += X3_WAVE) {
		const uint32_t i = base + lane;
		const uint64_t it = base == 0 ? first : (i < h.items ? pool[(uint64_t)h.off + i] : 0);
		const uint64_t mask = x3_ballot(i < h.items && (uint32_t)(it >> 32) == tag);
		if (mask) { q.found = 1; q.pos = base + (uint32_t)x3_ctz64(mask); break; }
	}
	return q;
}

/* ctx_touch (x3_tables.h) for the decoder: the lanes already hold items [0, 64) of the list (`first`), so bumping a frequency is a
 * plain store from the lane that holds the item -- no read-modify-write round trip on the chain.  pord != nullptr: a context1 list,
 * whose new item also records the ordinal of the pair (context1, tag) it stands for. */
__device__ static void dec_ctx_touch(X3CtxHdr *hp, X3CtxHdr &h, const CtxQ q, uint32_t tag, uint64_t first, uint32_t ord, uint64_t *pool, uint32_t *pord,
                                     uint64_t &pool_top, uint64_t pool_cap, uint32_t &status, uint32_t lane)
{
	if (q.found) {
		if (q.pos < X3_WAVE) { if (lane == q.pos) pool[(uint64_t)h.off + q.pos] = first + 1; }
		else if (lane == 0) pool[(uint64_t)h.off + q.pos] += 1;
	} else {
		if (h.items == h.cap) {
			const uint32_t ncap = h.cap ? 2 * h.cap : 2;
			if (pool_top + ncap > pool_cap) { status = X3_ST_POOL_FULL; return; }
			const uint32_t noff = (uint32_t)pool_top;
			pool_top += ncap;
			for (uint32_t i = lane; i < h.items; i += X3_WAVE) {
				pool[(uint64_t)noff + i] = pool[(uint64_t)h.off + i];
				if (pord) pord[(uint64_t)noff + i] = pord[(uint64_t)h.off + i];
			}
			h.off = noff;
			h.cap = ncap;
		}
		if (lane == 0) { pool[(uint64_t)h.off + h.items] = ((uint64_t)tag << 32) | 1u; if (pord) pord[(uint64_t)h.off + h.items] = ord; }
		h.items++;
	}
	h.total++;
	if (lane == 0) *hp = h;
}

#ifndef X3_DEC_LDS
#define X3_DEC_LDS 8192u /* dictionary elements whose recency list, index-model frequencies and (position, length) live in LDS: 10 bytes each */
#endif
#ifndef X3_DEC_LDS_SMALL
#define X3_DEC_LDS_SMALL 1024u /* ... in batches of many streams: 10 KiB of LDS per stream, so sixteen streams share a CU (the batch rate is streams in flight x the per-stream rate) */
#endif

/* The move-to-front list (dict.c:132-146), typed by where it lives: uint16_t in LDS, uint32_t in global memory.  Separate instantiations
 * on purpose: through one generic pointer every access is a FLAT instruction, and each of those waits for all outstanding global stores. */
template <typename T>
__device__ static __forceinline__ void dec_mtf_to_front(T *mtf, uint32_t r, uint32_t tag, uint32_t lane)
{
	if (r < X3_WAVE) { /* recent elements are the usual ones: one read and one write, straight-line */
		const bool act = lane >= 1 && lane <= r;
		const T v = act ? mtf[lane - 1] : (T)tag;
		x3_wave_order(); /* every lane has read before any lane overwrites its neighbour's source */
		if (lane <= r) mtf[lane] = v;
		return;
	}
	for (int base = (int)(r & ~(uint32_t)(X3_WAVE - 1)); base >= 0; base -= X3_WAVE) {
		const uint32_t j = (uint32_t)base + lane;
		const bool act = j >= 1 && j <= r;
		const T v = act ? mtf[j - 1] : (T)0;
		x3_wave_order();
		if (act) mtf[j] = v;
	}
	if (lane == 0) mtf[0] = (T)tag;
}
/* dict_get_index_by_tag (dict.c:174-183): rank of `tag`, 0xFFFFFFFF if it is not in the list */
template <typename T>
__device__ static __forceinline__ uint32_t dec_mtf_rank(const T *mtf, uint32_t D, uint32_t tag, uint32_t lane)
{
	for (uint32_t base = 0; base < D; base += X3_WAVE) {
		const uint32_t i = base + lane;
		const uint64_t mask = x3_ballot(i < D && (uint32_t)mtf[i] == tag);
		if (mask) return base + (uint32_t)x3_ctz64(mask);
	}
	return 0xFFFFFFFFu;
}

#define DFNV_OFF 2166136261u
#define DFNV_MUL 16777619u

__device__ static __forceinline__ uint32_t dht_slot(uint32_t h, uint32_t len, uint32_t hlog)
{
	uint32_t x = (h ^ (len * 0x9E3779B1u)) * 0x85EBCA6Bu;
	x ^= x >> 15;
	x *= 0xC2B2AE35u;
	return x >> (32 - hlog);
}

template <uint32_t NLDS>
__device__ static void x3_decode_body(const X3DecArgs &a)
{
	/* the two tables every step sweeps (move-to-front list, model_index1 frequencies) start in LDS and migrate to their global
	 * arrays only if the stream's dictionary outgrows NLDS elements */
	X3_LDS uint16_t s_mtf[NLDS];
	X3_LDS uint32_t s_idx[NLDS];
	X3_LDS uint32_t s_el[NLDS]; /* element: position in the output << 5 | length - 1  (positions < 2^27 = X3H_MAX_CHUNK) */
	static_assert(NLDS <= 65536, "tags in the LDS list are 16 bits wide");
	const X3DecChunk ck = a.chunks[blockIdx.x];
	const uint32_t lane = x3_lane();
	uint8_t *out = a.out + ck.out_off;
	uint32_t *dpos = a.dict_pos + ck.tag_off;
	uint8_t *dlen = a.dict_len + ck.tag_off;
	uint32_t *ht = a.ht + ck.ht_off;
	const uint32_t hlog = ck.ht_log2, hmask = (1u << hlog) - 1;
	uint32_t *gmtf = a.mtf + ck.tag_off, *gidx = a.idxfreq + ck.tag_off;
	bool lds = true; /* wave-uniform */
	X3CtxHdr *ctx1 = a.ctx1 + ck.tag_off, *ctx0 = a.ctx0 + ck.ctx0_off;
	uint64_t *pool = a.items + ck.item_off;
	uint32_t *pord = a.item_ord + ck.item_off;
	const uint32_t cap = ck.out_cap;

	BitReader br;
	br.p = a.in + ck.in_off; br.end = br.p + ck.in_len; br.w = 0; br.nb = 0; /* bio_open(READ), bio.c:14-15 */
	Dec d;
	d.lo = 0; d.hi = 0x7FFFFFFFu; d.buf = 0; /* ac_init */
	d.buf = br_take(br, 31); /* ac_decode_init, ac.c:133-140 */

	uint32_t evf = lane < 2 ? 1024u : lane < 5 ? 1u : 0u, evtotal = 2051; /* create(), x3.c:236-244: the frequency of event `lane` lives in that lane (an array indexed by the decision would live in scratch memory: a memory round trip per read on the chain) */
	uint32_t lf = 1, lftotal = 32;
	uint32_t cf0 = 1, cf1 = 1, cf2 = 1, cf3 = 1, cftotal = 256;
	uint32_t D = 0, idxtotal = 0, npairs = 0, status = X3_ST_OK;
	uint64_t pool_top = 0;
	uint64_t pc_ev = 0, pc_sym = 0, pc_rank = 0, pc_ctx = 0, pc_tail = 0; (void)pc_ev; (void)pc_sym; (void)pc_rank; (void)pc_ctx; (void)pc_tail;
	uint32_t ctx1tag = 0; /* context1; the (prev, context1) pair of x3.c:139 is tracked as its ordinal n_c0id */
	uint32_t p = 0;
	/* the context state of the NEXT hit step, loaded ahead: ctx0 ordinal (0 when the pair is unknown, x3.c:142-145), both headers, items [0,64) of both lists */
	uint32_t n_c0id = 0;
	X3CtxHdr n_h0 = ctx0[0], n_h1 = ctx1[0]; /* both empty at the start (zeroed workspace) */
	uint64_t n_it0 = 0, n_it1 = 0;
	uint32_t n_po1 = 0;               /* ... and the pair ordinals of the context1 items */
	uint32_t ord00 = 0, have00 = 0;   /* the pair (0, 0): what both contexts are after a new fragment (x3.c:321-322) */

	for (;;) {
		DPROF_T(t_a)
		/* ---- the event (x3.c:293-295) ---- */
		/* ac_decode_target + index_of_value (ac.c:128-131,167-179) without the second division:  (buf-lo)/step < c  <=>  buf-lo < c*step */
		uint32_t step = (d.hi - d.lo + 1) / evtotal;
		uint32_t decision;
		{
			const uint32_t incl = x3_row8_incl_scan_u32(evf);
			const uint64_t mask = x3_ballot(lane < 5 && dec_below(d.buf - d.lo, step, incl));
			if (!mask) { status = X3_ST_CORRUPT; break; } /* the reference abort()s, ac.c:178 */
			decision = (uint32_t)x3_ctz64(mask);
			const uint32_t fq = x3_readlane_u32(evf, decision), cum = x3_readlane_u32(incl, decision) - fq;
			if (!dec_narrow(d, br, step, cum, cum + fq)) { status = X3_ST_CORRUPT; break; }
			if (lane == decision) evf++;
			evtotal++;
		}
		if (decision == X3_E_EOF) break;
#ifdef X3_DEC_TRACE
		if (lane == 0) fprintf(stderr, "T %u %u D %u\n", p, decision, D);
#endif

		if (decision == X3_E_NEW) {
			/* ---- decode_match, x3.c:272-283 ---- */
			uint32_t len;
			{
				step = (d.hi - d.lo + 1) / lftotal;
				const uint32_t incl = wave_incl_scan(lane < 32 ? lf : 0u, lane);
				const uint64_t mask = x3_ballot(lane < 32 && dec_below(d.buf - d.lo, step, incl));
				if (!mask) { status = X3_ST_CORRUPT; break; }
				const uint32_t l = (uint32_t)x3_ctz64(mask);
				const uint32_t fq = x3_readlane_u32(lf, l), cl = x3_readlane_u32(incl, l) - fq;
				if (!dec_narrow(d, br, step, cl, cl + fq)) { status = X3_ST_CORRUPT; break; }
				if (lane == l) lf++;
				lftotal++;
				len = l + 1;
			}
			if ((uint64_t)p + len > cap) { status = X3_ST_OUT_FULL; break; }
			uint32_t h = DFNV_OFF;
			int bad = 0;
			for (uint32_t j = 0; j < len; j++) {
				step = (d.hi - d.lo + 1) / cftotal;
				const uint32_t offb = d.buf - d.lo;
				const uint32_t s4 = cf0 + cf1 + cf2 + cf3;
				const uint32_t incl = wave_incl_scan(s4, lane);
				const uint64_t mask = x3_ballot(dec_below(offb, step, incl));
				if (!mask) { bad = 1; break; }
				const uint32_t l = (uint32_t)x3_ctz64(mask);
				const uint32_t b0 = x3_readlane_u32(cf0, l), b1 = x3_readlane_u32(cf1, l), b2 = x3_readlane_u32(cf2, l), b3 = x3_readlane_u32(cf3, l);
				uint32_t cl = x3_readlane_u32(incl, l) - (b0 + b1 + b2 + b3), sub, fq;
				if (dec_below(offb, step, cl + b0)) { sub = 0; fq = b0; }
				else if (dec_below(offb, step, cl + b0 + b1)) { sub = 1; fq = b1; cl += b0; }
				else if (dec_below(offb, step, cl + b0 + b1 + b2)) { sub = 2; fq = b2; cl += b0 + b1; }
				else { sub = 3; fq = b3; cl += b0 + b1 + b2; }
				if (!dec_narrow(d, br, step, cl, cl + fq)) { bad = 1; break; }
				if (lane == l) { if (sub == 0) cf0++; else if (sub == 1) cf1++; else if (sub == 2) cf2++; else cf3++; }
				cftotal++;
				const uint32_t ch = 4 * l + sub;
				if (lane == 0) out[p + j] = (uint8_t)ch;
				h = (h ^ ch) * DFNV_MUL;
			}
			if (bad) { status = X3_ST_CORRUPT; break; }
			x3_wave_order(); /* the fragment is in memory for every lane */
			/* dict_query_elem (x3.c:309): exact lookup of (len, bytes) */
			int dup = 0;
			uint32_t slot = dht_slot(h, len, hlog);
			for (uint32_t e = ht[slot]; e != 0; slot = (slot + 1) & hmask, e = ht[slot]) {
				const uint32_t tg = e - 1;
				if (dlen[tg] != len) continue;
				const uint8_t *ds = out + dpos[tg];
				uint32_t k = 0;
				while (k < len && ds[k] == out[p + k]) k++;
				if (k == len) { dup = 1; break; }
			}
			x3_wave_order();
			if (!dup) { /* x3.c:310-317 */
				if (D == NLDS && lds) { /* outgrew LDS: continue in global memory (dpos/dlen are written there from the start) */
					for (uint32_t i = lane; i < D; i += X3_WAVE) { gmtf[i] = s_mtf[i]; gidx[i] = s_idx[i]; }
					x3_wave_order();
					lds = false;
				}
				if (lane == 0) { dpos[D] = p; dlen[D] = (uint8_t)len; ht[slot] = D + 1; }
				if (lds) {
					dec_mtf_to_front(s_mtf, D, D, lane);
					if (lane == 0) { s_idx[D] = 1; s_el[D] = (p << 5) | (len - 1); }
				} else {
					dec_mtf_to_front(gmtf, D, D, lane);
					if (lane == 0) gidx[D] = 1;
				}
				D++;
				idxtotal++;
			}
			p += len;
			ctx1tag = 0; /* x3.c:321-322: both contexts reset */
			x3_wave_order();
			/* the next hit step's contexts: pair (0, 0) if it is known, else context 0 (x3.c:142-145) */
			n_c0id = have00 ? ord00 : 0u;
			n_h0 = ctx0[n_c0id]; n_h1 = ctx1[0];
			n_it0 = lane < n_h0.items ? pool[(uint64_t)n_h0.off + lane] : 0;
			n_it1 = lane < n_h1.items ? pool[(uint64_t)n_h1.off + lane] : 0;
			n_po1 = lane < n_h1.items ? pord[(uint64_t)n_h1.off + lane] : 0;
			continue;
		}

		/* ---- decode_tag, x3.c:58-129 ---- */
		DPROF_T(t_b)
		DPROF_ADD(pc_ev, t_a, t_b)
		if (D == 0) { status = X3_ST_CORRUPT; break; }
		const uint32_t c0id = n_c0id;
		X3CtxHdr *h0p = ctx0 + c0id, *h1p = ctx1 + ctx1tag;
		const X3CtxHdr h0 = uni_hdr(n_h0), h1 = uni_hdr(n_h1); /* pinned here, not where the loads were issued: they stay in flight until now */
		const uint64_t it0 = n_it0, it1 = n_it1;
		const uint32_t po1 = n_po1;
		uint32_t tag = 0, rank = 0, cpos = 0;
		if (decision == X3_E_IDX1) {
			step = (d.hi - d.lo + 1) / idxtotal;
			uint32_t cl = 0, fq = 0;
			rank = lds ? find_in_array(s_idx, D, d.buf - d.lo, step, lane, cl, fq) : find_in_array(gidx, D, d.buf - d.lo, step, lane, cl, fq);
			if (rank == 0xFFFFFFFFu) { status = X3_ST_CORRUPT; break; }
			if (!dec_narrow(d, br, step, cl, cl + fq)) { status = X3_ST_CORRUPT; break; }
			if (lds) { tag = x3_uniform(s_mtf[rank]); x3_wave_order(); if (lane == 0) s_idx[rank] = fq + 1; } /* inc_model(&model_index1, index), x3.c:89 */
			else { tag = x3_uniform(gmtf[rank]); x3_wave_order(); if (lane == 0) gidx[rank] = fq + 1; }
			idxtotal++;
		} else {
			const X3CtxHdr hc = decision == X3_E_CTX0 ? h0 : h1;
			if (hc.items == 0 || hc.total == 0) { status = X3_ST_CORRUPT; break; }
			step = (d.hi - d.lo + 1) / hc.total;
			uint32_t cl = 0, fq = 0;
			const uint32_t pos = find_in_ctx(hc, pool, d.buf - d.lo, step, lane, decision == X3_E_CTX0 ? it0 : it1, cl, fq, tag);
			if (pos == 0xFFFFFFFFu) { status = X3_ST_CORRUPT; break; }
			cpos = pos;
			if (!dec_narrow(d, br, step, cl, cl + fq)) { status = X3_ST_CORRUPT; break; }
			/* dict_get_index_by_tag (x3.c:79,84) */
			rank = lds ? dec_mtf_rank(s_mtf, D, tag, lane) : dec_mtf_rank(gmtf, D, tag, lane);
			if (rank == 0xFFFFFFFFu) { status = X3_ST_CORRUPT; break; }
		}
		DPROF_T(t_c)
		DPROF_ADD(pc_sym, t_b, t_c)
		/* the tag is known: everything the rest of the step and the next step will wait for is requested NOW, in the order it will be needed --
		 * the element's bytes, then the header of the next context1 list (the tag just decoded; its list is only touched by this step when the
		 * tag follows itself, and then the updated header below is the one to use) */
		uint32_t len, src;
		if (lds) { const uint32_t e = x3_uniform(s_el[tag]); len = (e & 31u) + 1; src = e >> 5; }
		else { len = x3_uniform(dlen[tag]); src = x3_uniform(dpos[tag]); }
		if ((uint64_t)p + len > cap) { status = X3_ST_OUT_FULL; break; }
		const uint8_t piece = lane < len ? out[src + lane] : (uint8_t)0; /* len <= 32; src + len <= p */
		const bool self1 = tag == ctx1tag;
		const X3CtxHdr e_h1 = ctx1[tag]; /* unconditional, into its own registers: nothing below needs it before the updates are done, so the load stays in flight */
		/* x3.c:99-126: both contexts learn the tag, (context1, tag) becomes a known pair */
		CtxQ q0, q1; /* the context the tag was decoded from already told its list position */
		if (decision == X3_E_CTX0) { q0.found = 1; q0.pos = cpos; q0.freq = q0.cum = 0; } else q0 = ctx_find_tag(h0, pool, tag, lane, it0);
		if (decision == X3_E_CTX1) { q1.found = 1; q1.pos = cpos; q1.freq = q1.cum = 0; } else q1 = ctx_find_tag(h1, pool, tag, lane, it1);
		x3_wave_order();
		/* the pair (context1, tag) is this step's item in the context1 list -- and the (prev, context1) pair of the NEXT step */
		uint32_t ord;
		if (q1.found) ord = q1.pos < X3_WAVE ? x3_readlane_u32(po1, q1.pos) : x3_uniform(pord[(uint64_t)h1.off + q1.pos]);
		else {
			ord = npairs;
			if (ctx1tag == 0 && tag == 0) { ord00 = npairs; have00 = 1; }
			npairs++;
		}
		n_c0id = ord;
		/* the next context0 list: a new pair's header is still all zero (the workspace is cleared per batch), and when the pair repeats itself
		 * the list is this step's own, updated below */
		const bool self0 = ord == c0id;
		const X3CtxHdr e_h0 = ctx0[ord]; /* ord <= pairs so far <= output capacity: inside the table; all zero for a new pair */
		X3CtxHdr u0 = h0, u1 = h1;
		dec_ctx_touch(h0p, u0, q0, tag, it0, 0, pool, nullptr, pool_top, ck.item_cap, status, lane);
		dec_ctx_touch(h1p, u1, q1, tag, it1, ord, pool, pord, pool_top, ck.item_cap, status, lane);
		if (status != X3_ST_OK) break;
		/* x3.c:332-348: the element moves to the front (LDS work, while the header loads above are in flight) */
		if (lds) dec_mtf_to_front(s_mtf, rank, tag, lane); else dec_mtf_to_front(gmtf, rank, tag, lane);
		x3_wave_order(); /* lane 0's new item / the bumped frequency is stored before the lane that holds that list position loads it below */
		n_h1 = self1 ? u1 : e_h1;
		n_h0 = self0 ? u0 : e_h0;
		DPROF_T(t_d)
		DPROF_ADD(pc_ctx, t_c, t_d)
		/* items [0, 64) of both lists: in flight while the element is copied and the next event symbol is decoded (this step's updates are already stored) */
		n_it1 = lane < n_h1.items ? pool[(uint64_t)n_h1.off + lane] : 0;
		n_po1 = lane < n_h1.items ? pord[(uint64_t)n_h1.off + lane] : 0;
		n_it0 = lane < n_h0.items ? pool[(uint64_t)n_h0.off + lane] : 0;
		if (lane < len) out[p + lane] = piece; /* x3.c:332-340: the element's bytes */
		ctx1tag = tag; /* x3.c:346-347 */
		p += len;
		x3_wave_order();
		DPROF_T(t_e)
		DPROF_ADD(pc_tail, t_d, t_e)
	}

	if (lane == 0) {
		X3CodeResult r;
		r.out_len = p; r.status = status; r.pairs = npairs; r._r = D;
		for (int i = 0; i < 8; i++) r.events[i] = 0;
#ifdef X3_DEC_PROFILE
		r.events[4] = (uint32_t)(pc_ev >> 10); r.events[5] = (uint32_t)(pc_sym >> 10); r.events[6] = (uint32_t)(pc_ctx >> 10); r.events[7] = (uint32_t)(pc_tail >> 10);
#endif
		a.result[blockIdx.x] = r;
	}
	if (lane < 4) a.result[blockIdx.x].events[lane] = evf - (lane < 2 ? 1024u : 1u); /* events decoded = what the model counted */
}

#ifndef X3_EMU
__global__ void __launch_bounds__(X3_WAVE) x3_decode_kernel(X3DecArgs a) { x3_decode_body<X3_DEC_LDS>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_decode_many_kernel(X3DecArgs a) { x3_decode_body<X3_DEC_LDS_SMALL>(a); }
extern "C" void x3k_launch_decode(const X3DecArgs *a, uint32_t nchunks, hipStream_t st)
{
	/* one wavefront per stream; the LDS tables decide how many streams share a CU: a few streams get the big tables (no spill to
	 * global memory up to 16384 elements), a batch that oversubscribes the chip gets small ones (ten streams per CU) */
	if (nchunks > 256) hipLaunchKernelGGL(x3_decode_many_kernel, dim3(nchunks), dim3(X3_WAVE), 0, st, *a);
	else hipLaunchKernelGGL(x3_decode_kernel, dim3(nchunks), dim3(X3_WAVE), 0, st, *a);
}
#else
static void decode_tramp(void *p) { x3_decode_body<X3_DEC_LDS>(*(const X3DecArgs *)p); }
extern "C" void x3k_launch_decode(const X3DecArgs *a, uint32_t nchunks, void *)
{
	x3emu_launch(decode_tramp, (void *)a, dim3(nchunks), dim3(X3_WAVE));
}
#endif
