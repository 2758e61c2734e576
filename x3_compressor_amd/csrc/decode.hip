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
 *   dict_update_costs + qsort            -> move-to-front (dec_mtf_to_front)
 * The chain is latency-bound (a dozen dependent global loads per step in the naive order), so the loop PREFETCHES the next step's
 * context state as soon as it is known: the pair looked up / inserted at the end of a hit step (x3.c:213-222) IS the (prev, context1)
 * pair the next step would look up, so its ordinal is carried over instead of looked up again -- and it needs no hash map at all: a
 * pair (context1, tag) exists exactly when `tag` is in the item list of context1 (x3.c:197-222 add both in the same step), so every
 * item of a context1 list carries the ordinal of its pair.  Both context headers plus the first 64 items of either list are loaded at
 * the end of a step: they are in flight while the next event symbol is decoded.
 * Streams of a batch decode concurrently (grid = streams).  The reference's unchecked 64x output buffer (x3.c:621) is
 * replaced by a capacity check (X3_ST_OUT_FULL), malformed input ends in X3_ST_CORRUPT instead of abort() (ac.c:178).
 */
#include "x3_kernels.h"

struct CtxQ { uint32_t found, pos, freq, cum; }; /* where a tag sits in a context's item list */

/* experiment builds (tools/exp/dec_prof.py): -DX3_DEC_PROFILE=1 cycles per section of a hit step, =2 the context section in four parts,
 * =3 cycles spent in s_waitcnt vmcnt(0) at three points of the step; reported in the unused event slots of the result */
#if defined(X3_DEC_PROFILE)
#define DPROF_T(var) const uint64_t var = x3_clock();
#else
#define DPROF_T(var)
#endif
#if defined(X3_DEC_PROFILE) && X3_DEC_PROFILE == 1
#define DPROF1_ADD(acc, a, b) acc += (b) - (a);
#else
#define DPROF1_ADD(acc, a, b)
#endif
#if defined(X3_DEC_PROFILE) && X3_DEC_PROFILE == 2
#define DPROF2_T(var) const uint64_t var = x3_clock();
#define DPROF2_ADD(acc, a, b) acc += (b) - (a);
#else
#define DPROF2_T(var)
#define DPROF2_ADD(acc, a, b)
#endif
#if defined(X3_DEC_PROFILE) && X3_DEC_PROFILE == 3
#define DPROF3_WAIT(acc) { const uint64_t w0_ = x3_clock(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc += x3_clock() - w0_; }
#else
#define DPROF3_WAIT(acc)
#endif

/* A lone wave pays for every taken branch with an instruction-fetch bubble, and this loop is ~80 branches per step: the likely side of each
 * is marked so that the hot path is laid out as fall-through code. */
#define X3_LIKELY(x)   __builtin_expect(!!(x), 1)
#define X3_UNLIKELY(x) __builtin_expect(!!(x), 0)

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
	const uint32_t *base; /* the stream's words; whole words only, like the reference's reader (bio.c:35-39) */
	uint32_t nwords, wi;  /* ... how many, and the index of the next one to take */
	uint32_t cur, nxt;    /* lane l: word (block * 64 + l) of the block being consumed and of the one after it -- the stream is read 256 bytes at a
	                       * time, one block ahead, so taking the next word is a v_readlane and no load ever sits on the chain */
	uint64_t w;           /* unread bits, next one at bit 0 */
	uint32_t nb;          /* how many */
};

__device__ static __forceinline__ uint32_t br_block(const BitReader &r, uint32_t blk)
{
	const uint32_t i = blk * X3_WAVE + x3_lane();
	return i < r.nwords ? r.base[i] : 0x80000000u; /* bio.c:10,35-39: past the last whole word the reader feeds 0x80000000 */
}
__device__ static __forceinline__ void br_open(BitReader &r, const uint8_t *in, uint32_t len)
{
	r.base = (const uint32_t *)in; r.nwords = len >> 2; r.wi = 0; r.w = 0; r.nb = 0; /* bio_open(READ), bio.c:14-15 */
	r.cur = br_block(r, 0); r.nxt = br_block(r, 1);
}

/* the next n (0..31) bits of the stream, first one most significant -- what n calls of get_bit shifted into mBuffer would leave */
__device__ static __forceinline__ uint32_t br_take(BitReader &r, uint32_t n)
{
	if (X3_UNLIKELY(r.nb < n)) { /* one more word */
		const uint32_t nx = x3_readlane_u32(r.cur, r.wi & (X3_WAVE - 1));
		r.wi++;
		if (X3_UNLIKELY((r.wi & (X3_WAVE - 1)) == 0)) { r.cur = r.nxt; r.nxt = br_block(r, (r.wi >> 6) + 1); }
		r.w |= (uint64_t)nx << r.nb;
		r.nb += 32;
	}
	if (X3_UNLIKELY(!n)) return 0;
	const uint32_t field = (uint32_t)r.w & (0xFFFFFFFFu >> (32 - n));
	r.w >>= n;
	r.nb -= n;
	return dec_brev32(field) >> (32 - n);
}

struct Dec { uint32_t lo, hi, buf; };

/* range / total (ac.c:128-131) on the chain: a double-precision reciprocal and one fix-up step instead of the ~28-instruction integer
 * expansion of a 32-bit division (n <= 2^31, 0 < t < 2^28: the quotient estimate is off by at most one, and q * t cannot overflow) */
__device__ static __forceinline__ uint32_t dec_div(uint32_t n, uint32_t t)
{
#ifndef X3_EMU
	const double td = (double)t;
	double r = __builtin_amdgcn_rcp(td);       /* v_rcp_f64: an approximation ... */
	r = __builtin_fma(r, __builtin_fma(-td, r, 1.0), r); /* ... one Newton step makes it good to ~2^-50 whatever its accuracy class */
	uint32_t q = (uint32_t)((double)n * r);
	const int32_t rem = (int32_t)(n - q * t);
	q += rem < 0 ? 0xFFFFFFFFu : (uint32_t)rem >= t ? 1u : 0u;
	return x3_uniform(q);
#else
	return n / t;
#endif
}

/* ac_decode_symbol's interval update + ac_decode_scale (ac.c:192-195,142-165) in closed form, like the encoder's chain (code2.hip):
 * all E1/E2/E3 shifts together are  s = clz(D) - 1 - carry  with D = hi - lo after narrowing; lo and the range scale by 2^s, and since
 * every kind of shift removes the same offset from mBuffer as from mLow,  buffer - low  scales too and takes the s new bits.
 * false: not an interval a valid stream can produce (the reference would spin in its E1/E2 loop or read garbage). */
__device__ static __forceinline__ bool dec_narrow(Dec &d, BitReader &r, uint32_t step, uint32_t cum_lo, uint32_t cum_hi)
{
	const uint32_t nlo = d.lo + step * cum_lo, nhi = d.lo + step * cum_hi - 1, D = nhi - nlo;
	if (X3_UNLIKELY(D == 0 || ((nlo | nhi) >> 31) || nhi < nlo || d.buf < nlo || d.buf > nhi)) return false;
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

/* ... with entries [0, 64) already in registers (`first`: lane l holds freq[l]) */
__device__ static __forceinline__ uint32_t find_in_array_pre(const uint32_t *freq, uint32_t first, uint32_t count, uint32_t off, uint32_t step, uint32_t lane, uint32_t &cum_out, uint32_t &fq_out)
{
	uint32_t carry;
	{
		const uint32_t fq = lane < count ? first : 0;
		const uint32_t incl = wave_incl_scan(fq, lane);
		const uint64_t mask = x3_ballot(lane < count && dec_below(off, step, incl));
		if (X3_LIKELY(mask != 0)) {
			const uint32_t l = (uint32_t)x3_ctz64(mask);
			fq_out = x3_readlane_u32(fq, l);
			cum_out = x3_readlane_u32(incl, l) - fq_out;
			return l;
		}
		carry = x3_readlane_u32(incl, X3_WAVE - 1);
	}
	for (uint32_t base = X3_WAVE; base < count; base += X3_WAVE) {
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
	if (X3_LIKELY(h.items <= X3_WAVE)) { /* the usual case as straight-line code: no load, so nothing here makes the wave wait for loads that are in flight */
		const uint32_t fq = (uint32_t)first; /* lanes beyond the list hold 0 */
		const uint32_t incl = wave_incl_scan(fq, lane);
		const uint64_t mask = x3_ballot(lane < h.items && dec_below(off, step, incl));
		if (X3_UNLIKELY(!mask)) return 0xFFFFFFFFu;
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
	if (X3_LIKELY(h.items <= X3_WAVE)) { /* straight-line, no load (see find_in_ctx) */
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

/* x3.c:197-209 (add the tag with frequency 1 or bump its frequency) for the decoder: the lanes already hold items [0, 64) of the list (`first`), so bumping a frequency is a
 * plain store from the lane that holds the item -- no read-modify-write round trip on the chain.  `first` (and `po`, the pair ordinals
 * of a context1 list: with_ord) come back as the list reads AFTER the update, so a list that is also the next step's context needs no
 * reload; the caller stores the updated header `h` wherever headers of that kind live. */
__device__ static __forceinline__ void dec_ctx_touch(X3CtxHdr &h, const CtxQ q, uint32_t tag, uint64_t &first, uint32_t &po, bool with_ord, uint32_t ord,
                                                     uint64_t *pool, uint32_t *pord, uint32_t &pool_top, uint32_t pool_cap, uint32_t &status, uint32_t lane)
{
	if (q.found) {
		if (X3_LIKELY(q.pos < X3_WAVE)) { if (lane == q.pos) { first += 1; pool[(uint64_t)h.off + q.pos] = first; } }
		else if (lane == 0) pool[(uint64_t)h.off + q.pos] += 1;
	} else {
		if (X3_UNLIKELY(h.items == h.cap)) {
			const uint32_t ncap = h.cap ? 2 * h.cap : 2;
			if ((uint64_t)pool_top + ncap > pool_cap) { status = X3_ST_POOL_FULL; return; }
			const uint32_t noff = pool_top;
			pool_top += ncap;
			for (uint32_t i = lane; i < h.items; i += X3_WAVE) {
				pool[(uint64_t)noff + i] = pool[(uint64_t)h.off + i];
				if (with_ord) pord[(uint64_t)noff + i] = pord[(uint64_t)h.off + i];
			}
			h.off = noff;
			h.cap = ncap;
		}
		const uint64_t it = ((uint64_t)tag << 32) | 1u;
		if (lane == 0) { pool[(uint64_t)h.off + h.items] = it; if (with_ord) pord[(uint64_t)h.off + h.items] = ord; }
		if (lane == h.items) { first = it; po = ord; } /* items < 64: the lane of the new list position */
		h.items++;
	}
	h.total++;
}
/* list capacities only ever double from 2 (above), so the LDS copy of a header does not store one */
__device__ static __forceinline__ uint32_t dec_cap_of(uint32_t n) { return n == 0 ? 0u : n <= 2 ? 2u : 1u << (32 - x3_clz32(n - 1)); }

#ifndef X3_DEC_LDS
#define X3_DEC_LDS 4096u /* dictionary elements whose tables live in LDS, 20 bytes each: recency list, index-model frequency, (position, length), header of the element's context1 list */
#endif
#ifndef X3_DEC_LDS_MID
#define X3_DEC_LDS_MID 2048u /* ... in batches of up to 1024 streams: 40 KiB per stream, four streams (one per SIMD) share a CU */
#endif
#ifndef X3_DEC_LDS_SMALL
#define X3_DEC_LDS_SMALL 512u /* ... in batches of many streams: 10 KiB of LDS per stream, so sixteen streams share a CU (the batch rate is streams in flight x the per-stream rate) */
#endif

/* The move-to-front list (dict.c:132-146), typed by where it lives: uint16_t in LDS, uint32_t in global memory.  Separate instantiations
 * on purpose: through one generic pointer every access is a FLAT instruction, and each of those waits for all outstanding global stores. */
template <typename T>
__device__ static __forceinline__ void dec_mtf_to_front(T *mtf, uint32_t r, uint32_t tag, uint32_t lane)
{
	if (X3_LIKELY(r < X3_WAVE)) { /* recent elements are the usual ones: one read and one write, straight-line */
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
__device__ static __forceinline__ uint32_t dec_mtf_rank(const T *mtf, uint32_t D, uint32_t tag, uint32_t lane, uint32_t from = 0)
{
	for (uint32_t base = from; base < D; base += X3_WAVE) {
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

/* per-stream tables in global memory + constants (wave-uniform) */
struct DecT {
	uint8_t *out;
	uint32_t *dpos; uint8_t *dlen; uint32_t *ht; uint32_t hlog, hmask;
	uint32_t *gmtf, *gidx;
	X3CtxHdr *ctx1, *ctx0;
	uint64_t *pool; uint32_t *pord; uint32_t pool_cap;
	uint32_t cap;
};
/* the chain's state, carried from the LDS-resident loop into the spilled one */
struct DecS {
	BitReader br; Dec d;
	uint32_t evf, evtotal;            /* event model: the frequency of event `lane` lives in that lane (an array indexed by the decision would sit in scratch memory: a round trip per read on the chain) */
	uint32_t lf;                      /* length model, one symbol per lane (its total is 32 + the fragments so far: the event model counts those) */
	uint32_t cf0, cf1, cf2, cf3;      /* byte model, four symbols per lane (total = their sum, taken when a fragment starts) */
	uint32_t D, npairs, status;       /* (the index model's total is D + the index events so far) */
	uint32_t pool_top;
	uint32_t ctx1tag, p;
	/* the context state of the NEXT hit step, loaded ahead: ctx0 ordinal (0 when the pair is unknown, x3.c:142-145), both headers, items [0,64) of both lists, pair ordinals of the context1 items */
	uint32_t n_c0id; X3CtxHdr n_h0, n_h1; uint64_t n_it0, n_it1; uint32_t n_po1;
	uint32_t ord00;                   /* ordinal of the pair (0, 0), what both contexts are after a new fragment (x3.c:321-322); 0xFFFFFFFF while unknown */
	uint64_t pc_ev, pc_sym, pc_ctx, pc_tail;
};

/* One instantiation per residence of the per-element tables: LDS = true while the dictionary has fewer than NLDS elements (recency
 * list, index-model frequencies, element (position, length) and the headers of the context1 lists are LDS arrays: the only global
 * round trip between decoding a tag and having the next step's context1 items is the item load itself), LDS = false after they
 * migrated to global memory.  Returns true when the LDS tables are full (the caller migrates them and continues with the other loop). */
template <uint32_t NLDS, bool LDS>
__device__ static __forceinline__ bool dec_loop(const DecT &t, DecS &s, uint16_t *s_mtf, uint32_t *s_idx, uint32_t *s_el, uint32_t *s_c1off, uint32_t *s_c1tot, uint16_t *s_c1n, const uint32_t lane)
{
	uint8_t *const out = t.out;
	uint64_t *const pool = t.pool;
	uint32_t *const pord = t.pord;
	BitReader &br = s.br;
	Dec &d = s.d;
	for (;;) {
		if (LDS && X3_UNLIKELY(s.D == NLDS)) return true;
		/* ranks [0, 64) of the recency list and of the index model, read ahead of the event symbol: recent elements are the usual ones, and then
		 * neither the rank search nor the move-to-front waits for an LDS read (lanes >= D hold nothing meaningful; the new-fragment path below
		 * changes both tables and comes back here) */
		uint32_t m0 = 0, i0 = 0;
		if (LDS) { m0 = s_mtf[lane]; i0 = s_idx[lane]; }
		DPROF_T(t_a)
		/* ---- the event (x3.c:293-295) ---- */
		/* ac_decode_target + index_of_value (ac.c:128-131,167-179) without the second division:  (buf-lo)/step < c  <=>  buf-lo < c*step */
		uint32_t step = dec_div(d.hi - d.lo + 1, s.evtotal);
		uint32_t decision;
		{
			const uint32_t incl = x3_row8_incl_scan_u32(s.evf);
			const uint64_t mask = x3_ballot(lane < 5 && dec_below(d.buf - d.lo, step, incl));
			if (X3_UNLIKELY(!mask)) { s.status = X3_ST_CORRUPT; break; } /* the reference abort()s, ac.c:178 */
			decision = (uint32_t)x3_ctz64(mask);
			const uint32_t fq = x3_readlane_u32(s.evf, decision), cum = x3_readlane_u32(incl, decision) - fq;
			if (X3_UNLIKELY(!dec_narrow(d, br, step, cum, cum + fq))) { s.status = X3_ST_CORRUPT; break; }
			if (lane == decision) s.evf++;
			s.evtotal++;
		}
		if (X3_UNLIKELY(decision == X3_E_EOF)) break;

		if (X3_UNLIKELY(decision == X3_E_NEW)) {
			/* ---- decode_match, x3.c:272-283 ---- */
			uint32_t len;
			{
				const uint32_t lftotal = 30u + x3_readlane_u32(s.evf, X3_E_NEW); /* 32 + the fragments before this one: the event model started at 1 and has counted this one already */
				step = dec_div(d.hi - d.lo + 1, lftotal);
				const uint32_t incl = wave_incl_scan(lane < 32 ? s.lf : 0u, lane);
				const uint64_t mask = x3_ballot(lane < 32 && dec_below(d.buf - d.lo, step, incl));
				if (X3_UNLIKELY(!mask)) { s.status = X3_ST_CORRUPT; break; }
				const uint32_t l = (uint32_t)x3_ctz64(mask);
				const uint32_t fq = x3_readlane_u32(s.lf, l), cl = x3_readlane_u32(incl, l) - fq;
				if (X3_UNLIKELY(!dec_narrow(d, br, step, cl, cl + fq))) { s.status = X3_ST_CORRUPT; break; }
				if (lane == l) s.lf++;
				len = l + 1;
			}
			const uint32_t p = s.p;
			if (X3_UNLIKELY((uint64_t)p + len > t.cap)) { s.status = X3_ST_OUT_FULL; break; }
			uint32_t h = DFNV_OFF, cftotal = x3_wave_sum_u32(s.cf0 + s.cf1 + s.cf2 + s.cf3);
			int bad = 0;
			for (uint32_t j = 0; j < len; j++) {
				step = dec_div(d.hi - d.lo + 1, cftotal);
				const uint32_t offb = d.buf - d.lo;
				const uint32_t s4 = s.cf0 + s.cf1 + s.cf2 + s.cf3;
				const uint32_t incl = wave_incl_scan(s4, lane);
				const uint64_t mask = x3_ballot(dec_below(offb, step, incl));
				if (X3_UNLIKELY(!mask)) { bad = 1; break; }
				const uint32_t l = (uint32_t)x3_ctz64(mask);
				const uint32_t b0 = x3_readlane_u32(s.cf0, l), b1 = x3_readlane_u32(s.cf1, l), b2 = x3_readlane_u32(s.cf2, l), b3 = x3_readlane_u32(s.cf3, l);
				uint32_t cl = x3_readlane_u32(incl, l) - (b0 + b1 + b2 + b3), sub, fq;
				if (dec_below(offb, step, cl + b0)) { sub = 0; fq = b0; }
				else if (dec_below(offb, step, cl + b0 + b1)) { sub = 1; fq = b1; cl += b0; }
				else if (dec_below(offb, step, cl + b0 + b1 + b2)) { sub = 2; fq = b2; cl += b0 + b1; }
				else { sub = 3; fq = b3; cl += b0 + b1 + b2; }
				if (X3_UNLIKELY(!dec_narrow(d, br, step, cl, cl + fq))) { bad = 1; break; }
				if (lane == l) { if (sub == 0) s.cf0++; else if (sub == 1) s.cf1++; else if (sub == 2) s.cf2++; else s.cf3++; }
				cftotal++;
				const uint32_t ch = 4 * l + sub;
				if (lane == 0) out[p + j] = (uint8_t)ch;
				h = (h ^ ch) * DFNV_MUL;
			}
			if (X3_UNLIKELY(bad)) { s.status = X3_ST_CORRUPT; break; }
			x3_wave_order(); /* the fragment is in memory for every lane */
			/* dict_query_elem (x3.c:309): exact lookup of (len, bytes) */
			int dup = 0;
			uint32_t slot = dht_slot(h, len, t.hlog);
			for (uint32_t e = t.ht[slot]; e != 0; slot = (slot + 1) & t.hmask, e = t.ht[slot]) {
				const uint32_t tg = e - 1;
				if (t.dlen[tg] != len) continue;
				const uint8_t *ds = out + t.dpos[tg];
				uint32_t k = 0;
				while (k < len && ds[k] == out[p + k]) k++;
				if (k == len) { dup = 1; break; }
			}
			x3_wave_order();
			if (!dup) { /* x3.c:310-317 */
				const uint32_t D = s.D;
				if (lane == 0) { t.dpos[D] = p; t.dlen[D] = (uint8_t)len; t.ht[slot] = D + 1; } /* the hash table reads these two, wherever the other tables live */
				if (LDS) {
					dec_mtf_to_front(s_mtf, D, D, lane);
					if (lane == 0) { s_idx[D] = 1; s_el[D] = (p << 5) | (len - 1); s_c1off[D] = 0; s_c1tot[D] = 0; s_c1n[D] = 0; }
				} else {
					dec_mtf_to_front(t.gmtf, D, D, lane);
					if (lane == 0) t.gidx[D] = 1;
				}
				s.D++;
			}
			s.p = p + len;
			s.ctx1tag = 0; /* x3.c:321-322: both contexts reset */
			x3_wave_order();
			/* the next hit step's contexts: pair (0, 0) if it is known, else context 0 (x3.c:142-145) */
			s.n_c0id = s.ord00 != 0xFFFFFFFFu ? s.ord00 : 0u;
			s.n_h0 = t.ctx0[s.n_c0id];
			if (LDS) { s.n_h1.off = x3_uniform(s_c1off[0]); s.n_h1.items = x3_uniform((uint32_t)s_c1n[0]); s.n_h1.total = x3_uniform(s_c1tot[0]); s.n_h1.cap = dec_cap_of(s.n_h1.items); }
			else s.n_h1 = t.ctx1[0];
			s.n_it0 = lane < s.n_h0.items ? pool[(uint64_t)s.n_h0.off + lane] : 0;
			s.n_it1 = lane < s.n_h1.items ? pool[(uint64_t)s.n_h1.off + lane] : 0;
			s.n_po1 = lane < s.n_h1.items ? pord[(uint64_t)s.n_h1.off + lane] : 0;
			continue;
		}

		/* ---- decode_tag, x3.c:58-129 ---- */
		DPROF_T(t_b)
		DPROF1_ADD(s.pc_ev, t_a, t_b)
		const uint32_t D = s.D;
		if (X3_UNLIKELY(D == 0)) { s.status = X3_ST_CORRUPT; break; }
		DPROF3_WAIT(s.pc_ev)
		const uint32_t c0id = s.n_c0id, ctx1tag = s.ctx1tag;
		const X3CtxHdr h0 = uni_hdr(s.n_h0), h1 = uni_hdr(s.n_h1); /* pinned here, not where the loads were issued: they stay in flight until now */
		const uint64_t it0 = s.n_it0, it1 = s.n_it1;
		const uint32_t po1 = s.n_po1;
		uint32_t tag = 0, rank = 0, cpos = 0;
		if (decision == X3_E_IDX1) {
			const uint32_t idxtotal = D + x3_readlane_u32(s.evf, X3_E_IDX1) - 2u; /* one per element + one per index event before this one */
			step = dec_div(d.hi - d.lo + 1, idxtotal);
			uint32_t cl = 0, fq = 0;
			rank = LDS ? find_in_array_pre(s_idx, i0, D, d.buf - d.lo, step, lane, cl, fq) : find_in_array(t.gidx, D, d.buf - d.lo, step, lane, cl, fq);
			if (X3_UNLIKELY(rank == 0xFFFFFFFFu)) { s.status = X3_ST_CORRUPT; break; }
			if (X3_UNLIKELY(!dec_narrow(d, br, step, cl, cl + fq))) { s.status = X3_ST_CORRUPT; break; }
			if (LDS) { tag = X3_LIKELY(rank < X3_WAVE) ? x3_readlane_u32(m0, rank) : x3_uniform((uint32_t)s_mtf[rank]); x3_wave_order(); if (lane == 0) s_idx[rank] = fq + 1; } /* inc_model(&model_index1, index), x3.c:89 */
			else { tag = x3_uniform(t.gmtf[rank]); x3_wave_order(); if (lane == 0) t.gidx[rank] = fq + 1; }
		} else {
			const X3CtxHdr hc = decision == X3_E_CTX0 ? h0 : h1;
			if (X3_UNLIKELY(hc.items == 0 || hc.total == 0)) { s.status = X3_ST_CORRUPT; break; }
			step = dec_div(d.hi - d.lo + 1, hc.total);
			uint32_t cl = 0, fq = 0;
			const uint32_t pos = find_in_ctx(hc, pool, d.buf - d.lo, step, lane, decision == X3_E_CTX0 ? it0 : it1, cl, fq, tag);
			if (X3_UNLIKELY(pos == 0xFFFFFFFFu)) { s.status = X3_ST_CORRUPT; break; }
			cpos = pos;
			if (X3_UNLIKELY(!dec_narrow(d, br, step, cl, cl + fq))) { s.status = X3_ST_CORRUPT; break; }
			/* dict_get_index_by_tag (x3.c:79,84) */
			if (LDS) {
				const uint64_t mask = x3_ballot(lane < D && m0 == tag);
				rank = X3_LIKELY(mask != 0) ? (uint32_t)x3_ctz64(mask) : dec_mtf_rank(s_mtf, D, tag, lane, X3_WAVE);
			} else rank = dec_mtf_rank(t.gmtf, D, tag, lane);
			if (X3_UNLIKELY(rank == 0xFFFFFFFFu)) { s.status = X3_ST_CORRUPT; break; }
		}
		DPROF_T(t_c)
		DPROF1_ADD(s.pc_sym, t_b, t_c)
		/* the tag is known: everything the rest of the step and the next step will wait for is requested NOW -- the element's bytes, and the
		 * next context1 list (the one of the tag just decoded; this step only touches it when the tag follows itself, and then the registers
		 * updated below are used instead of what is loaded here) */
		uint32_t len, src;
		X3CtxHdr e_h1;
		if (LDS) { /* four LDS reads in flight together */
			const uint32_t e = s_el[tag], ho = s_c1off[tag], hn = s_c1n[tag], ht = s_c1tot[tag];
			len = (x3_uniform(e) & 31u) + 1; src = x3_uniform(e) >> 5;
			e_h1.off = x3_uniform(ho); e_h1.items = x3_uniform(hn); e_h1.total = x3_uniform(ht); e_h1.cap = dec_cap_of(e_h1.items);
		} else { len = x3_uniform(t.dlen[tag]); src = x3_uniform(t.dpos[tag]); }
		const uint32_t p = s.p;
		if (X3_UNLIKELY((uint64_t)p + len > t.cap)) { s.status = X3_ST_OUT_FULL; break; }
		const uint8_t piece = lane < len ? out[src + lane] : (uint8_t)0; /* len <= 32; src + len <= p */
		const bool self1 = tag == ctx1tag;
		uint64_t e_it1 = 0;
		uint32_t e_po1 = 0;
		if (LDS) {
			e_it1 = lane < e_h1.items ? pool[(uint64_t)e_h1.off + lane] : 0;
			e_po1 = lane < e_h1.items ? pord[(uint64_t)e_h1.off + lane] : 0;
		} else e_h1 = t.ctx1[tag]; /* into its own registers: nothing below needs it before the updates are done, so the load stays in flight */
		DPROF2_T(u_a)
		DPROF2_ADD(s.pc_ev, t_c, u_a)
		/* x3.c:99-126: both contexts learn the tag, (context1, tag) becomes a known pair */
		CtxQ q0, q1; /* the context the tag was decoded from already told its list position */
		if (decision == X3_E_CTX0) { q0.found = 1; q0.pos = cpos; q0.freq = q0.cum = 0; } else q0 = ctx_find_tag(h0, pool, tag, lane, it0);
		if (decision == X3_E_CTX1) { q1.found = 1; q1.pos = cpos; q1.freq = q1.cum = 0; } else q1 = ctx_find_tag(h1, pool, tag, lane, it1);
		/* the pair (context1, tag) is this step's item in the context1 list -- and the (prev, context1) pair of the NEXT step */
		uint32_t ord;
		if (X3_LIKELY(q1.found)) ord = X3_LIKELY(q1.pos < X3_WAVE) ? x3_readlane_u32(po1, q1.pos) : x3_uniform(pord[(uint64_t)h1.off + q1.pos]);
		else {
			ord = s.npairs;
			if (ctx1tag == 0 && tag == 0) s.ord00 = s.npairs;
			s.npairs++;
		}
		/* the next context0 list: a new pair's header is still all zero (the workspace is cleared per batch), and when the pair repeats itself
		 * the list is this step's own, updated below */
		const bool self0 = ord == c0id;
		const X3CtxHdr e_h0 = t.ctx0[ord]; /* ord <= pairs so far <= output capacity: inside the table */
		DPROF2_T(u_b)
		DPROF2_ADD(s.pc_sym, u_a, u_b)
		X3CtxHdr u0 = h0, u1 = h1;
		uint64_t p_it0 = it0, p_it1 = it1;
		uint32_t p_po1 = po1, nopo = 0;
		dec_ctx_touch(u0, q0, tag, p_it0, nopo, false, 0, pool, pord, s.pool_top, t.pool_cap, s.status, lane);
		dec_ctx_touch(u1, q1, tag, p_it1, p_po1, true, ord, pool, pord, s.pool_top, t.pool_cap, s.status, lane);
		if (X3_UNLIKELY(s.status != X3_ST_OK)) break;
		if (lane == 0) {
			t.ctx0[c0id] = u0;
			if (LDS) { s_c1off[ctx1tag] = u1.off; s_c1n[ctx1tag] = (uint16_t)u1.items; s_c1tot[ctx1tag] = u1.total; }
			else t.ctx1[ctx1tag] = u1;
		}
		DPROF2_T(u_c)
		DPROF2_ADD(s.pc_ctx, u_b, u_c)
		/* x3.c:332-348: the element moves to the front (LDS work, while the loads above are in flight) */
		if (LDS) {
			if (X3_LIKELY(rank < X3_WAVE)) { const uint32_t below = x3_wave_shr1_u32(m0); if (lane <= rank) s_mtf[lane] = (uint16_t)(lane ? below : tag); } /* no read: ranks [0, 64) are in registers */
			else dec_mtf_to_front(s_mtf, rank, tag, lane);
		} else dec_mtf_to_front(t.gmtf, rank, tag, lane);
		x3_wave_order(); /* this step's stores come before the loads below in program order, also for the lanes that did not store */
		DPROF3_WAIT(s.pc_sym)
		DPROF_T(t_d)
		DPROF2_ADD(s.pc_tail, u_c, t_d)
		DPROF1_ADD(s.pc_ctx, t_c, t_d)
		if (!LDS) { /* spilled: the header came from global memory, its items are requested only now */
			e_it1 = lane < e_h1.items ? pool[(uint64_t)e_h1.off + lane] : 0;
			e_po1 = lane < e_h1.items ? pord[(uint64_t)e_h1.off + lane] : 0;
		}
		s.n_c0id = ord;
		s.n_h1 = self1 ? u1 : e_h1; s.n_it1 = self1 ? p_it1 : e_it1; s.n_po1 = self1 ? p_po1 : e_po1;
		const uint64_t e_it0 = lane < e_h0.items ? pool[(uint64_t)e_h0.off + lane] : 0; /* in flight while the next event symbol is decoded */
		s.n_h0 = self0 ? u0 : e_h0; s.n_it0 = self0 ? p_it0 : e_it0;
		DPROF3_WAIT(s.pc_ctx)
		if (lane < len) out[p + lane] = piece; /* x3.c:332-340: the element's bytes */
		s.ctx1tag = tag; /* x3.c:346-347 */
		s.p = p + len;
		x3_wave_order();
		DPROF_T(t_e)
		DPROF1_ADD(s.pc_tail, t_d, t_e)
	}
	return false;
}

template <uint32_t NLDS>
__device__ static void x3_decode_body(const X3DecArgs &a)
{
	/* per dictionary element while there are fewer than NLDS of them: rank -> tag of the move-to-front list, model_index1 frequency by rank,
	 * (position << 5 | length - 1) of the element (positions < 2^27 = X3H_MAX_CHUNK), offset / total / item count of its context1 list */
	X3_LDS uint16_t s_mtf[NLDS];
	X3_LDS uint32_t s_idx[NLDS];
	X3_LDS uint32_t s_el[NLDS];
	X3_LDS uint32_t s_c1off[NLDS];
	X3_LDS uint32_t s_c1tot[NLDS];
	X3_LDS uint16_t s_c1n[NLDS];
	static_assert(NLDS >= X3_WAVE && NLDS <= 65536, "tags and list lengths in the LDS tables are 16 bits wide; ranks [0, 64) are read without a bounds check");
	const X3DecChunk ck = a.chunks[blockIdx.x];
	const uint32_t lane = x3_lane();
	DecT t;
	t.out = a.out + ck.out_off;
	t.dpos = a.dict_pos + ck.tag_off; t.dlen = a.dict_len + ck.tag_off;
	t.ht = a.ht + ck.ht_off; t.hlog = ck.ht_log2; t.hmask = (1u << ck.ht_log2) - 1;
	t.gmtf = a.mtf + ck.tag_off; t.gidx = a.idxfreq + ck.tag_off;
	t.ctx1 = a.ctx1 + ck.tag_off; t.ctx0 = a.ctx0 + ck.ctx0_off;
	t.pool = a.items + ck.item_off; t.pord = a.item_ord + ck.item_off; t.pool_cap = (uint32_t)ck.item_cap; /* 8 * capacity + 64 <= 2^30 + 64 */
	t.cap = ck.out_cap;

	DecS s;
	br_open(s.br, a.in + ck.in_off, ck.in_len);
	s.d.lo = 0; s.d.hi = 0x7FFFFFFFu; s.d.buf = 0; /* ac_init */
	s.d.buf = br_take(s.br, 31); /* ac_decode_init, ac.c:133-140 */
	s.evf = lane < 2 ? 1024u : lane < 5 ? 1u : 0u; s.evtotal = 2051; /* create(), x3.c:236-244 */
	s.lf = 1;
	s.cf0 = s.cf1 = s.cf2 = s.cf3 = 1;
	s.D = 0; s.npairs = 0; s.status = X3_ST_OK;
	s.pool_top = 0;
	s.ctx1tag = 0; s.p = 0;
	s.n_c0id = 0;
	s.n_h0.off = s.n_h0.items = s.n_h0.cap = s.n_h0.total = 0; /* both contexts are empty at the start */
	s.n_h1 = s.n_h0;
	s.n_it0 = s.n_it1 = 0; s.n_po1 = 0;
	s.ord00 = 0xFFFFFFFFu;
	s.pc_ev = s.pc_sym = s.pc_ctx = s.pc_tail = 0;
	if (lane == 0) { s_c1off[0] = 0; s_c1tot[0] = 0; s_c1n[0] = 0; } /* the list of context 0 is looked at before element 0 exists */
	x3_wave_order();

	if (dec_loop<NLDS, true>(t, s, s_mtf, s_idx, s_el, s_c1off, s_c1tot, s_c1n, lane)) {
		/* NLDS elements: the tables continue in global memory (dpos/dlen were written there from the start) */
		for (uint32_t i = lane; i < NLDS; i += X3_WAVE) {
			t.gmtf[i] = s_mtf[i]; t.gidx[i] = s_idx[i];
			X3CtxHdr h; h.off = s_c1off[i]; h.items = s_c1n[i]; h.cap = dec_cap_of(h.items); h.total = s_c1tot[i];
			t.ctx1[i] = h;
		}
		x3_wave_order();
		(void)dec_loop<NLDS, false>(t, s, s_mtf, s_idx, s_el, s_c1off, s_c1tot, s_c1n, lane);
	}

	if (lane == 0) {
		X3CodeResult r;
		r.out_len = s.p; r.status = s.status; r.pairs = s.npairs; r._r = s.D;
		for (int i = 0; i < 8; i++) r.events[i] = 0;
#ifdef X3_DEC_PROFILE
		r.events[4] = (uint32_t)(s.pc_ev >> 10); r.events[5] = (uint32_t)(s.pc_sym >> 10); r.events[6] = (uint32_t)(s.pc_ctx >> 10); r.events[7] = (uint32_t)(s.pc_tail >> 10);
#endif
		a.result[blockIdx.x] = r;
	}
	x3_wave_order();
	if (lane < 4) a.result[blockIdx.x].events[lane] = s.evf - (lane < 2 ? 1024u : 1u); /* events decoded = what the model counted */
}

#ifndef X3_EMU
__global__ void __launch_bounds__(X3_WAVE) x3_decode_kernel(X3DecArgs a) { x3_decode_body<X3_DEC_LDS>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_decode_mid_kernel(X3DecArgs a) { x3_decode_body<X3_DEC_LDS_MID>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_decode_many_kernel(X3DecArgs a) { x3_decode_body<X3_DEC_LDS_SMALL>(a); }
extern "C" void x3k_launch_decode(const X3DecArgs *a, uint32_t nchunks, hipStream_t st)
{
	/* one wavefront per stream; the LDS tables decide how many streams share a CU: up to one stream per CU gets the big tables (4096
	 * elements before they migrate to global memory), up to one per SIMD the middle ones, a batch beyond that the small ones (16 per CU) */
	if (nchunks > 1024) hipLaunchKernelGGL(x3_decode_many_kernel, dim3(nchunks), dim3(X3_WAVE), 0, st, *a);
	else if (nchunks > 256) hipLaunchKernelGGL(x3_decode_mid_kernel, dim3(nchunks), dim3(X3_WAVE), 0, st, *a);
	else hipLaunchKernelGGL(x3_decode_kernel, dim3(nchunks), dim3(X3_WAVE), 0, st, *a);
}
#else
static void decode_tramp(void *p) { x3_decode_body<X3_DEC_LDS>(*(const X3DecArgs *)p); }
extern "C" void x3k_launch_decode(const X3DecArgs *a, uint32_t nchunks, void *)
{
	x3emu_launch(decode_tramp, (void *)a, dim3(nchunks), dim3(X3_WAVE));
}
#endif
