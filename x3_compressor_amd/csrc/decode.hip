/*
 * decode.hip -- the x3 decoder (reference: decompress() x3.c:285-353, decode_tag() x3.c:58-129, decode_match() x3.c:272-283,
 * arithmetic decoder ac.c:128-198, bit reader bio.c:30-42,74-103).
 *
 * Unlike the encoder, the decoder cannot run ahead of its models: which table is consulted next depends on the symbol just decoded, so
 * a stream is one dependent chain, and a lone wavefront pays ~5 cycles for every instruction it issues.  The decoder is therefore split
 * the way the encoder is (K2 / K3):
 *
 *   stage 1, the CHAIN (x3_decode_kernel, one wavefront per stream): events, tags and the models -- nothing else.  It writes ONE TAG PER
 *     PARSE STEP (a new fragment is the tag of the dictionary element it becomes, or repeats: x3.c:306-326) and the bytes of every new
 *     element (`lit`, in order of creation).  It never reads or writes the output, keeps no output position and no element lengths.
 *   stage 2 (x3_dec_lens / x3_dec_scan / x3_dec_copy, the whole chip): element lengths of the tags -> output positions by prefix sum ->
 *     the bytes, copied from `lit`.  x3.c:332-340.
 *
 * The chain's state is wave-uniform and lives in scalar registers; every linear table walk of the reference is a 64-lane sweep:
 *   index_of_value (ac.c:167-179)          -> inclusive wave scan of 64 frequencies, one multiply-compare per lane, first lane below
 *   dict_get_index_by_tag (dict.c:174-183) -> compare of the 64 most recent tags, held in a register
 *   dict_update_costs + qsort              -> move-to-front: one DPP shift of that register
 *
 * Context lists (context.c) are BLOCKS of a pool, header first: entry 0 = {total, items}, entry 1 + i = item i = {freq, tag}.  One load of 64
 * entries brings the header and the first 63 items into the lanes, where they stay while the list is a current context; an update is a
 * store of the changed entries from the lanes that hold them (header and item in ONE store instruction), never a read-modify-write.
 * Items are in insertion order (ctx_sort is a no-op, context.c:75-86), so the frequent tags of a long list are among the first 63.
 * A context1 list (x3.c:100-107; one per tag) has 16-byte entries: its item for `tag` also carries the pool offset of the context0 list of
 * the pair (list's tag, tag) -- a pair exists exactly when its tag is in that list (x3.c:197-222 add both in one step), so there is no
 * pair map, no table of context0 headers and no pair ordinal: the context0 list of the next step is named by the context1 item this
 * step touches anyway.  Lists grow by doubling; a context0 block that moved leaves a forwarding entry behind, and whoever follows it
 * patches the item that sent it there.
 *
 * Streams of a batch decode concurrently (grid = streams).  The reference's unchecked 64x output buffer (x3.c:621) is replaced by a
 * capacity check (X3_ST_OUT_FULL), malformed input ends in X3_ST_CORRUPT instead of abort() (ac.c:178).
 */
#include "x3_kernels.h"

/* A lone wave pays for every taken branch with an instruction-fetch bubble: the likely side of each is marked so that the hot path is
 * laid out as fall-through code. */
#define X3_LIKELY(x)   __builtin_expect(!!(x), 1)
#define X3_UNLIKELY(x) __builtin_expect(!!(x), 0)

#ifdef X3_EMU
/* emulator builds: how often the rare paths ran (tests/test_emu_kernels.py makes sure its inputs reach them): [0] a forwarding entry followed, [1] a symbol decoded
 * beyond entry 63, [2] a tag looked up beyond entry 63, [3] a block moved, [4] a rank beyond 63, [5] the item that names a moved block patched in the lanes */
extern "C" { unsigned x3emu_dec_cover[8]; }
#define DEC_COVER(i) { if (x3_lane() == 0) x3emu_dec_cover[i]++; }
#else
#define DEC_COVER(i)
#endif

#define X3D_NONE 0xFFFFFFFFu /* "no position" / "no block"; in the `items` word of a context0 header: the block moved, `total` says where to */

/* return codes of the chain's loop */
#define X3D_EOF     0u
#define X3D_MIGRATE 1u /* the LDS tables are full: continue on the tables in global memory */
#define X3D_FAIL    2u /* + status */

#ifndef X3_EMU
/* first set bit of a lane mask, X3D_NONE for an empty one (s_ff1_i32_b64 returns -1 for zero: no test needed) */
__device__ static __forceinline__ uint32_t dec_first(uint64_t m) { return (uint32_t)(__ffsll((long long)m) - 1); }
__device__ static __forceinline__ uint32_t dec_brev32(uint32_t v) { return __brev(v); }
/* v of the lane below; lane 0 gets `fill` (wave_shr:1 without bound_ctrl keeps the old value where there is no source lane) */
__device__ static __forceinline__ uint32_t dec_shr1_fill(uint32_t v, uint32_t fill) { return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false); }
#else
static inline uint32_t dec_first(uint64_t m) { return m ? (uint32_t)__builtin_ctzll(m) : 0xFFFFFFFFu; }
static inline uint32_t dec_brev32(uint32_t v) { uint32_t r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }
static inline uint32_t dec_shr1_fill(uint32_t v, uint32_t fill) { const int l = (int)x3_lane(); const uint32_t u = x3emu_shfl(v, l ? l - 1 : 0); return l ? u : fill; }
#endif

/* Scalars of the chain that only its rare paths touch live in LDS (s_cold[]), not in registers: the step's state is ~100 scalars, and what the compiler
 * spills when they do not fit is not ours to choose (it took the token pointer).  Read with dec_cold(), written by lane 0. */
/* The first 64 * X3_DEC_MTFR ranks of the recency list stay in registers (one register per 64 ranks).  Round 4 kept 64: on the 1 MB mr-like stream 52 % of the steps found
 * their tag BEHIND rank 63 (31 % behind 127, 2.4 % behind 255; 256 KiB of text: 38 % / 1 % / 0), and each of those swept the list in LDS twice -- once for the rank, once to
 * move everything in front of it down: 750 of the step's ~2 400 cycles (section clocks of a profile build, tools/dec_prof.py).  (The emulator build of the tests keeps two,
 * so that its inputs still reach the sweeps.) */
#ifndef X3_DEC_MTFR
#define X3_DEC_MTFR 4
#endif
/* A context0 list (one per tag pair) starts with room for X3_DEC_CAP0 items (a power of two) and doubles from there.  Round 4 started at 2: 38 % of the steps append an
 * item to their context0 list, the lists average 11 items, so a list moved at 2, 4 and 8 items -- 7-8 % of ALL steps copied a block through memory (load, wait, store,
 * forwarding entry, patch of the item that names it) in the middle of the chain. */
#ifndef X3_DEC_CAP0
#define X3_DEC_CAP0 8u
#endif
enum { DC_IN_LO, DC_IN_HI, DC_NWORDS, DC_LITPOS, DC_O00, DC_OFIRST, DC_E3, DC_E4, DC_COUNT, DC_MODELS = DC_COUNT + 2 * X3_WAVE /* behind the two stream blocks */ };
#ifndef X3_EMU
__device__ static __forceinline__ uint32_t dec_cold(const uint32_t *s_cold, int i) { return x3_uniform(s_cold[i]); }
#else
static inline uint32_t dec_cold(const uint32_t *s_cold, int i) { return x3_bcast_u32(s_cold[i], 0); } /* (a rendezvous: the emulator's lanes run one after the other between those, and lane 0 may be about to overwrite the word) */
#endif
__device__ static __forceinline__ void dec_cold_set(uint32_t *s_cold, int i, uint32_t v) { if (x3_lane() == 0) s_cold[i] = v; x3_wave_order(); }
/* a real s_waitcnt vmcnt(0) (the compiler accounts for it, unlike one in an asm statement): behind a load on a rare path whose result is used much later -- left
 * pending, it would make the compiler wait for ALL loads at the next use or reuse of that register on the hot path, the prefetched context blocks included */
__device__ static __forceinline__ void dec_wait_loads()
{
#ifndef X3_EMU
	__builtin_amdgcn_s_waitcnt(0x0F70); /* gfx9 encoding: vmcnt = 0, expcnt = 7, lgkmcnt = 15 (no wait) */
#endif
}

struct BitReader { /* bio.c:5-42: bit k of the stream = bit k mod 32 (LSB first) of the little-endian word k / 32 */
	uint32_t wi;          /* index of the next word to take (the stream's address and its number of whole words: s_cold[DC_IN_LO, DC_IN_HI, DC_NWORDS]).  The words
	                       * come through LDS (s_in[128]: the 64-word block being consumed and the one after it, fetched one block ahead), so taking a word is an LDS read
	                       * and no global load ever sits on the chain */
	uint64_t w;           /* unread bits, first one at bit 63 */
	uint32_t nb;          /* how many */
};

/* words [64 blk, 64 blk + 64) of the stream -> their half of s_in */
__device__ static __forceinline__ void br_block(uint32_t *s_cold, uint32_t blk)
{
	const uint32_t *base = (const uint32_t *)(((uint64_t)dec_cold(s_cold, DC_IN_HI) << 32) | dec_cold(s_cold, DC_IN_LO)); /* whole words only, like the reference's reader (bio.c:35-39) */
	const uint32_t i = blk * X3_WAVE + x3_lane();
	const uint32_t v = i < dec_cold(s_cold, DC_NWORDS) ? base[i] : 0x80000000u; /* bio.c:10,35-39: past the last whole word the reader feeds 0x80000000 */
	(s_cold + DC_COUNT)[(blk & 1u) * X3_WAVE + x3_lane()] = v;
	dec_wait_loads();
	x3_wave_order();
}
__device__ static __forceinline__ void br_open(BitReader &r, uint32_t *s_cold, const uint8_t *in, uint32_t len)
{
	dec_cold_set(s_cold, DC_IN_LO, (uint32_t)(uintptr_t)in); dec_cold_set(s_cold, DC_IN_HI, (uint32_t)((uint64_t)(uintptr_t)in >> 32)); dec_cold_set(s_cold, DC_NWORDS, len >> 2);
	r.wi = 0; r.w = 0; r.nb = 0; /* bio_open(READ), bio.c:14-15 */
	br_block(s_cold, 0); br_block(s_cold, 1);
}

/* makes sure the window holds n (0..31) unread bits */
__device__ static __forceinline__ void br_need(BitReader &r, uint32_t *s_cold, uint32_t n)
{
	if (X3_UNLIKELY(r.nb < n)) { /* one more word, turned round once so that taking bits is a shift */
		const uint32_t nx = dec_brev32(dec_cold(s_cold, DC_COUNT + (r.wi & (2 * X3_WAVE - 1))));
		r.wi++;
		if (X3_UNLIKELY((r.wi & (X3_WAVE - 1)) == 0)) br_block(s_cold, (r.wi >> 6) + 1); /* the block just finished is replaced by the one after the next */
		r.w |= (uint64_t)nx << (32 - r.nb); /* nb < n <= 31 */
		r.nb += 32;
	}
}
/* the next n (0..31) bits of the stream, first one most significant -- what n calls of get_bit shifted into mBuffer would leave */
__device__ static __forceinline__ uint32_t br_take(BitReader &r, uint32_t *s_cold, uint32_t n)
{
	br_need(r, s_cold, n);
	const uint32_t field = (uint32_t)((r.w >> 1) >> (63 - n)); /* n == 0: nothing */
	r.w <<= n;
	r.nb -= n;
	return field;
}

/* The interval as (low, range, buffer - low): the decoder never needs mHigh or mBuffer themselves. */
struct Dec { uint32_t lo, rng, off; };

/* range / total (ac.c:128-131) on the chain: a double-precision reciprocal and one fix-up step instead of the ~28-instruction integer
 * expansion of a 32-bit division (n <= 2^31, 0 < t < 2^28: the quotient estimate is off by at most one, and q * t cannot overflow) */
__device__ static __forceinline__ uint32_t dec_div(uint32_t n, uint32_t t)
{
#ifndef X3_EMU
	const double td = (double)t;
	double r = __builtin_amdgcn_rcp(td);       /* v_rcp_f64: an approximation ... */
	/* ... one Newton step makes it good to ~2^-50 whatever its accuracy class -- aimed 2^-43 BELOW 1 / t (the constant is 1 - 2^-43 instead of 1): the
	 * estimate n * r is then never above n / t and less than 2^-11 below it (n <= 2^31), so the quotient is exact or one short, and the fix-up is one-sided */
	r = __builtin_fma(r, __builtin_fma(-td, r, 1.0 - 0x1p-43), r);
	uint32_t q = (uint32_t)((double)n * r);
	q += n - q * t >= t ? 1u : 0u;
	return x3_uniform(q);
#else
	return t ? n / t : 0xFFFFFFFFu;
#endif
}

/* ac_decode_symbol's interval update + ac_decode_scale (ac.c:192-195,142-165) in closed form, like the encoder's chain (code2.hip):
 * the symbol's slice is [lo + cs, lo + cs + rs) with cs = step * cum, rs = step * freq.  All E1/E2/E3 shifts together are
 * s = clz(D) - 1 - carry with D = rs - 1; low and range scale by 2^s, and since every kind of shift removes the same offset from mBuffer as
 * from mLow, buffer - low scales too and takes the s new bits: (buffer - low) and the unread bits behind it are shifted as ONE 64-bit value.
 * No validity test: after a shift the range is >= 2^29 and every total is < 2^28 (one count per parse step or byte on top of <= 2051 initial counts, X3H_MAX_CHUNK = 2^28 - 4096), so
 * step >= 2, D >= 1; the symbol was chosen as the first with off < step * cum_incl, so cs <= off < cs + rs -- whatever the stream holds. */
__device__ static __forceinline__ void ac_narrow(Dec &d, BitReader &r, uint32_t *s_cold, uint32_t cs, uint32_t rs)
{
	const uint32_t nlo = d.lo + cs, D = rs - 1, nhi = nlo + D;
#ifndef X3_EMU
	const uint32_t cz = (uint32_t)__builtin_clz(D);
#else
	const uint32_t cz = (uint32_t)x3_clz32(D);
#endif
	const uint32_t t = 31u - cz;
	const uint32_t sh = cz - 1 - ((((nlo ^ nhi) >> t) & 1u) ^ 1u);
	br_need(r, s_cold, sh);
	d.lo = (nlo << sh) & 0x3FFFFFFFu;
	d.rng = rs << sh;
	d.off = (uint32_t)(((((uint64_t)(d.off - cs)) << 32 | (r.w >> 32)) << sh) >> 32); /* (off - cs) << sh fits: it is below the new range */
	r.w <<= sh;
	r.nb -= sh;
}

/* ---- context blocks ------------------------------------------------------------------------------------------------------------
 * Pool offsets count 8-byte units.  STR = units per entry: 1 for a context0 block, 2 for a context1 block (16-byte entries, the third
 * word of an item = pool offset of the context0 block of the pair).  Entry 0 is the header {total, items}, entry e >= 1 is item e - 1
 * {freq, tag}: "where a tag is" is its entry number (lane e holds entry e for e < 64), X3D_NONE if it is not in the list.
 * Capacities double from 2, so a list of n items is full when n is a power of two >= 2 -- no capacity field. */
template <int STR> __device__ static __forceinline__ uint32_t *blk_entry(uint64_t *pool, uint32_t o, uint32_t e) { return (uint32_t *)(pool + (uint64_t)o + (uint64_t)e * STR); }
__device__ static __forceinline__ void blk_load0(uint64_t *pool, uint32_t o, uint32_t lane, uint32_t &f, uint32_t &t)
{
	const uint64_t e = (pool + o)[lane];
	f = (uint32_t)e; t = (uint32_t)(e >> 32);
}
__device__ static __forceinline__ void blk_load1(uint64_t *pool, uint32_t o, uint32_t lane, uint32_t &f, uint32_t &t, uint32_t &c)
{
	const uint32_t *e = (const uint32_t *)(pool + o) + 4 * lane; /* three of the entry's four words (a fourth register would only be a destination to wait for) */
	const uint64_t ft = *(const uint64_t *)e;
	f = (uint32_t)ft; t = (uint32_t)(ft >> 32); c = e[2];
}
/* lanes 1 .. min(n, 63): the lanes that hold items */
__device__ static __forceinline__ uint64_t blk_lanes(uint32_t n, uint32_t lane) { return x3_ballot(lane <= n) & ~(uint64_t)1; }

/* The blocks of the NEXT step's contexts are requested at the end of a step and used after the next event has been decoded: two loads that must stay in
 * flight across the loop's back edge.  Left to the compiler, the registers they write become loop-carried values with several definitions, and it copies them
 * around right behind the loads -- every such copy is a full wait for the data (measured: 40 % of the step).  So the two loads and the point where their
 * data is taken over are written out: fixed registers above anything the kernel allocates (v120..v124: the kernel then needs 125 registers, which still lets four wavefronts share a SIMD -- the
 * sixteen streams per CU of the many-stream variant; with v230.. it was two) (the clobber lists keep them out of the compiler's hands
 * across each statement, tests/test_build.py checks that nothing else in the kernel names them), one s_waitcnt where the data is first needed.
 * dec_request: entries [0, 64) of the context1 block at o1 and of the context0 block at o0;  dec_take: wait for them and hand them to the compiler.
 * The emulator has no latency to hide: it loads at the request, like the hardware, into a struct. */
struct DecPend { uint32_t f0, t0, f1, t1, c1; };
#ifndef X3_EMU
__device__ static __forceinline__ void dec_request(uint64_t *pool, uint32_t o1, uint32_t o0, uint32_t lane, DecPend &)
{
	const uint64_t *p0 = pool + o0, *p1 = pool + o1;
	asm volatile("global_load_dwordx3 v[122:124], %0, %1\n\tglobal_load_dwordx2 v[120:121], %2, %3"
	             :: "v"(lane * 16u), "s"(p1), "v"(lane * 8u), "s"(p0) : "v120", "v121", "v122", "v123", "v124", "memory");
}
__device__ static __forceinline__ void dec_take(const DecPend &, uint32_t &b0f, uint32_t &b0t, uint32_t &b1f, uint32_t &b1t, uint32_t &b1c)
{
	asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, v120\n\tv_mov_b32 %1, v121\n\tv_mov_b32 %2, v122\n\tv_mov_b32 %3, v123\n\tv_mov_b32 %4, v124"
	             : "=v"(b0f), "=v"(b0t), "=v"(b1f), "=v"(b1t), "=v"(b1c) :: "memory");
}
#else
static inline void dec_request(uint64_t *pool, uint32_t o1, uint32_t o0, uint32_t lane, DecPend &p) { blk_load1(pool, o1, lane, p.f1, p.t1, p.c1); blk_load0(pool, o0, lane, p.f0, p.t0); }
static inline void dec_take(const DecPend &p, uint32_t &b0f, uint32_t &b0t, uint32_t &b1f, uint32_t &b1t, uint32_t &b1c) { b0f = p.f0; b0t = p.t0; b1f = p.f1; b1t = p.t1; b1c = p.c1; }
#endif

/* entry of `tag` among entries [64, n] (ctx_query_tag_item, context.c:20-40, for the part of a long list that is not in the lanes) */
template <int STR>
__device__ static uint32_t blk_find_far(uint64_t *pool, uint32_t o, uint32_t n, uint32_t tag, uint32_t lane)
{
	for (uint32_t base = X3_WAVE; base <= n; base += X3_WAVE) {
		const uint32_t e = base + lane;
		const uint32_t tg = e <= n ? blk_entry<STR>(pool, o, e)[1] : 0;
		const uint64_t m = x3_ballot(e <= n && tg == tag);
		if (m) return base + (uint32_t)x3_ctz64(m);
	}
	return X3D_NONE;
}
/* index_of_value over entries [64, n]; carry = sum of the frequencies before them */
template <int STR>
__device__ static uint32_t blk_decode_far(uint64_t *pool, uint32_t o, uint32_t n, uint32_t off, uint32_t step, uint32_t carry, uint32_t lane, uint32_t &cum, uint32_t &fq, uint32_t &tag)
{
	for (uint32_t base = X3_WAVE; base <= n; base += X3_WAVE) {
		const uint32_t e = base + lane;
		const uint32_t *p = blk_entry<STR>(pool, o, e <= n ? e : 1);
		const uint32_t f = e <= n ? p[0] : 0, tg = p[1];
		const uint32_t incl = x3_wave_incl_scan_u32(f) + carry;
		const uint64_t m = x3_ballot(e <= n && off < incl * step);
		if (m) {
			const uint32_t l = (uint32_t)x3_ctz64(m);
			fq = x3_readlane_u32(f, l); cum = x3_readlane_u32(incl, l) - fq; tag = x3_readlane_u32(tg, l);
			return base + l;
		}
		carry = x3_readlane_u32(incl, X3_WAVE - 1);
	}
	return X3D_NONE;
}

#ifndef X3_DEC_LDS
#define X3_DEC_LDS 8192u /* dictionary elements whose tables live in LDS, 10 bytes each: recency list, index-model frequency, offset of the element's context1 block */
#endif
#ifndef X3_DEC_LDS_MID
#define X3_DEC_LDS_MID 3584u /* ... in batches of up to 1024 streams: 35 KiB + the 1.8 KiB of s_cold per stream, so four streams (one per SIMD) share a CU's 160 KiB */
#endif
#ifndef X3_DEC_LDS_EIGHT
#define X3_DEC_LDS_EIGHT 1792u /* ... in batches of up to 2048 streams: 17.5 + 1.8 KiB per stream, eight streams per CU -- with the small tables below such a batch would sit sixteen to a CU on half of the chip */
#endif
#ifndef X3_DEC_LDS_SMALL
#define X3_DEC_LDS_SMALL 768u /* ... in batches of many streams: 9.3 KiB of LDS per stream, so sixteen streams share a CU (the batch rate is streams in flight x the per-stream rate) */
#endif

/* The move-to-front list (dict.c:132-146) beyond its first 64 ranks, typed by where it lives: uint16_t in LDS, uint32_t in global memory.
 * Separate instantiations on purpose: through one generic pointer every access is a FLAT instruction, and each of those waits for all
 * outstanding global stores. */
template <typename T>
__device__ static __forceinline__ void dec_mtf_to_front(T *mtf, uint32_t r, uint32_t tag, uint32_t lane)
{
	for (int base = (int)(r & ~(uint32_t)(X3_WAVE - 1)); base >= 0; base -= X3_WAVE) {
		const uint32_t j = (uint32_t)base + lane;
		const bool act = j >= 1 && j <= r;
		const T v = act ? mtf[j - 1] : (T)0;
		x3_wave_order(); /* every lane has read before any lane overwrites its neighbour's source */
		if (act) mtf[j] = v;
	}
	if (lane == 0) mtf[0] = (T)tag;
	x3_wave_order();
}
/* dict_get_index_by_tag (dict.c:174-183) from rank `from` on (behind the ranks held in registers): rank of `tag`, X3D_NONE if it is not in the list */
template <typename T>
__device__ static __forceinline__ uint32_t dec_mtf_rank_far(const T *mtf, uint32_t D, uint32_t tag, uint32_t lane, uint32_t from = X3_WAVE)
{
	for (uint32_t base = from; base < D; base += X3_WAVE) {
		const uint32_t i = base + lane;
		const uint64_t mask = x3_ballot(i < D && (uint32_t)mtf[i] == tag);
		if (mask) return base + (uint32_t)x3_ctz64(mask);
	}
	return X3D_NONE;
}
/* index_of_value over the index model from rank 64 on */
__device__ static __forceinline__ uint32_t dec_idx_far(const uint32_t *freq, uint32_t count, uint32_t off, uint32_t step, uint32_t carry, uint32_t lane, uint32_t &cum, uint32_t &fq)
{
	for (uint32_t base = X3_WAVE; base < count; base += X3_WAVE) {
		const uint32_t i = base + lane;
		const uint32_t f = i < count ? freq[i] : 0;
		const uint32_t incl = x3_wave_incl_scan_u32(f) + carry;
		const uint64_t m = x3_ballot(i < count && off < incl * step);
		if (m) {
			const uint32_t l = (uint32_t)x3_ctz64(m);
			fq = x3_readlane_u32(f, l); cum = x3_readlane_u32(incl, l) - fq;
			return base + l;
		}
		carry = x3_readlane_u32(incl, X3_WAVE - 1);
	}
	return X3D_NONE;
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
	uint32_t *dpos; uint8_t *dlen; uint32_t *ht; uint32_t hlog, hmask;
	uint32_t *gmtf, *gidx, *gc1;
	uint64_t *pool; uint32_t pool_cap;
	uint32_t *tok; uint8_t *lit;
	uint32_t cap; /* output capacity = token capacity */
};
/* the chain's state, carried from the LDS-resident loop into the spilled one */
struct DecS {
	BitReader br; Dec d;
	uint32_t e0, e1, e2, evtotal;     /* event model (x3.c:236-244), one scalar per hit event (new fragments, end of stream: s_cold[DC_E3, DC_E4]) */
	/* the models of new fragments live in LDS, one word per lane each (s_cold + DC_MODELS): the length model, one symbol per lane (its total is 32 + the fragments
	 * so far: the event model counts those), and the byte model, four symbols per lane (total = their sum, taken when a fragment starts) */
	uint32_t D, npairs, status;       /* (the index model's total is D + the index events so far) */
	uint32_t pool_top, ntok;          /* (bytes of the elements so far: s_cold[DC_LITPOS]) */
	uint32_t ctx1tag;
	uint32_t o0, o1;                  /* the current contexts: pool offsets of their blocks */
	uint64_t pc_wait, pc_flight, pc_chain; /* (profile builds) */
	uint32_t ref0;                    /* the context1 item (pool offset of its entry) that names the current context0 block; X3D_NONE: reached through o00 / ofirst */
	/* s_cold[DC_O00]: context0 block of the pair (0, 0), what both contexts are after a new fragment (x3.c:321-322; X3D_NONE while that pair is unknown);
	 * s_cold[DC_OFIRST]: the block of pair number 0, the default (x3.c:142-145) -- it exists, and learns tags, before the first pair does */
};

/* a context0 block moved to `noff`: the item that names it (and the lane that holds that item, if the list is the current context1) learns the new place */
__device__ static __forceinline__ void dec_patch_ref(uint64_t *pool, uint32_t ref0, uint32_t noff, uint32_t o1, uint32_t &b1c, uint32_t lane)
{
	if (ref0 == X3D_NONE) return;
	if (lane == 0) ((uint32_t *)(pool + ref0))[2] = noff;
	const uint32_t rel = ref0 - o1; /* entry e of the current context1 block sits at o1 + 2 e */
	if (ref0 >= o1 && rel < 2 * X3_WAVE && !(rel & 1u)) DEC_COVER(5)
	b1c = (ref0 >= o1 && rel < 2 * X3_WAVE && !(rel & 1u) && lane == (rel >> 1)) ? noff : b1c;
}

/* x3.c:197-209 on a block whose first 64 entries are in the lanes (f, t, c): bump the frequency of entry e, or append `tag` with frequency 1 (e == X3D_NONE).
 * Returns false when the pool is exhausted.  STR == 1: a context0 block (o, and what refers to it: ref0 / s_cold[DC_O00, DC_OFIRST]); STR == 2: the context1 block of ctx1tag. */
template <int STR, bool LDS>
__device__ static __forceinline__ bool dec_touch(uint64_t *pool, uint32_t pool_cap, uint32_t &pool_top, uint32_t &o, uint32_t n, uint32_t e, uint32_t tag, uint32_t c0,
                                                 uint32_t &f, uint32_t &t, uint32_t &c, uint32_t lane,
                                                 uint32_t ref0, uint32_t *s_cold, uint32_t o1, uint32_t &b1c,                /* STR == 1 */
                                                 uint32_t ctx1tag, uint32_t *s_c1, uint32_t *gc1)                               /* STR == 2 */
{
	if (X3_LIKELY(e != X3D_NONE)) {
		if (X3_LIKELY(e < X3_WAVE)) { /* header (total) and item (freq): both in lanes, one store */
			if (lane == 0 || lane == e) { f += 1; blk_entry<STR>(pool, o, lane)[0] = f; }
		} else if (lane == 0) { f += 1; blk_entry<STR>(pool, o, 0)[0] = f; blk_entry<STR>(pool, o, e)[0] += 1; }
		return true;
	}
	if (X3_UNLIKELY(n >= (STR == 1 ? X3_DEC_CAP0 : 2u) && (n & (n - 1)) == 0)) { /* full: twice the capacity somewhere else */
		DEC_COVER(3)
		const uint32_t top = STR == 2 ? (pool_top + 1) & ~1u : pool_top, units = STR * (1 + 2 * n);
		if (X3_UNLIKELY((uint64_t)top + units + 2 * X3_WAVE > pool_cap)) return false;
		for (uint32_t i = lane; i < STR * (1 + n); i += X3_WAVE) pool[(uint64_t)top + i] = pool[(uint64_t)o + i];
		if (STR == 1) {
			if (lane == 0) pool[o] = ((uint64_t)X3D_NONE << 32) | top; /* forwarding entry */
			if (dec_cold(s_cold, DC_O00) == o) dec_cold_set(s_cold, DC_O00, top);
			if (dec_cold(s_cold, DC_OFIRST) == o) dec_cold_set(s_cold, DC_OFIRST, top);
			dec_patch_ref(pool, ref0, top, o1, b1c, lane);
		} else {
			if (LDS) { if (lane == 0) s_c1[ctx1tag] = top; } else if (lane == 0) gc1[ctx1tag] = top;
		}
		o = top; pool_top = top + units;
		x3_wave_order();
	}
	if (X3_LIKELY(n + 1 < X3_WAVE)) {
		if (lane == 0) { f += 1; t += 1; }
		if (lane == n + 1) { f = 1; t = tag; c = c0; }
		if (lane == 0 || lane == n + 1) {
			uint32_t *p = blk_entry<STR>(pool, o, lane);
			if (STR == 1) *(uint64_t *)p = ((uint64_t)t << 32) | f;
			else { *(uint64_t *)p = ((uint64_t)t << 32) | f; p[2] = c; }
		}
	} else if (lane == 0) {
		f += 1; t += 1;
		uint32_t *h = blk_entry<STR>(pool, o, 0), *p = blk_entry<STR>(pool, o, n + 1);
		h[0] = f; h[1] = t;
		p[0] = 1; p[1] = tag; if (STR == 2) p[2] = c0;
	}
	return true;
}

/* One instantiation per residence of the per-element tables: LDS = true while the dictionary has fewer than NLDS elements (recency list,
 * index-model frequencies and the offsets of the context1 blocks are LDS arrays), LDS = false after they migrated to global memory.
 * Works on copies of the state and writes them back when it returns X3D_EOF or X3D_MIGRATE: a failing stream (X3D_FAIL + S.status) leaves
 * through returns that keep nothing alive, so the error exits cost the hot path no registers.
 * NOTE for whoever edits this loop: every branch on a LANE-dependent condition next to one of the failing exits can make the compiler's
 * uniformity analysis give up on the loop ("cycle with divergent exit"); the chain's scalars then all become vector registers and the
 * step takes twice as long.  tests/test_build.py checks the register counts of the compiled kernel for exactly that. */
template <uint32_t NLDS, bool LDS>
__device__ static __forceinline__ uint32_t dec_loop(const DecT &T, DecS &S, uint16_t *s_mtf, uint32_t *s_idx, uint32_t *s_c1, uint32_t *s_cold, const uint32_t lane)
{
	uint64_t *const pool = T.pool;
	BitReader br = S.br;
	Dec d = S.d;
	uint32_t e0 = S.e0, e1 = S.e1, e2 = S.e2, evtotal = S.evtotal;
	uint32_t D = S.D, npairs = S.npairs, pool_top = S.pool_top, ctx1tag = S.ctx1tag;
	uint32_t *tokp = T.tok + S.ntok; /* where the next token goes, and how many more fit */
	uint32_t tokleft = T.cap - S.ntok;
	uint32_t o0 = S.o0, o1 = S.o1, ref0 = S.ref0;
	DecPend pend = { 0, 0, 0, 0, 0 };
	uint32_t nouse = 0;
	/* ranks [0, 64) of the recency list and of the index model stay in registers: recent elements are the usual ones, and then neither the rank
	 * search nor the move-to-front nor the index model reads a table (lanes >= D: no tag / frequency 0) */
	uint32_t m0, i0;
	uint32_t mx[X3_DEC_MTFR > 1 ? X3_DEC_MTFR - 1 : 1]; /* ranks [64, 64 * X3_DEC_MTFR): X3D_NONE behind the last element */
	if (LDS) { m0 = s_mtf[lane]; i0 = s_idx[lane]; } else { m0 = T.gmtf[lane]; i0 = T.gidx[lane]; }
#define DEC_MTF_RELOAD                                                                                                              \
	_Pragma("unroll") for (uint32_t k_ = 0; k_ + 1 < X3_DEC_MTFR; k_++) {                                                           \
		const uint32_t j_ = (k_ + 1) * X3_WAVE + lane;                                                                              \
		mx[k_] = j_ < D ? (LDS ? (uint32_t)s_mtf[j_ < NLDS ? j_ : 0] : T.gmtf[j_]) : X3D_NONE;                                       \
	}
	DEC_MTF_RELOAD
	uint32_t code;
	/* The blocks of the next step's contexts are requested (dec_request) as soon as their addresses are known -- before this step's interval update, rank search,
	 * move-to-front and token -- and taken over (dec_take) after the NEXT event has been decoded. */
	dec_request(pool, o1, o0, lane, pend);
#ifdef X3_DEC_PROFILE /* experiment builds (-DX3_DEC_PROFILE): shader cycles waiting for the requested blocks / between request and use / from use to the next request */
	uint64_t pc_wait = 0, pc_flight = 0, pc_chain = 0, pc_req = x3_clock();
#endif
#define DEC_FAIL(st) { S.status = (st); return X3D_FAIL; }
#ifdef X3_DEC_PROFILE
#define DEC_PROF_REQ pc_req = x3_clock();
#else
#define DEC_PROF_REQ
#endif
	for (;;) {

		/* ---- the event (x3.c:293-295): ac_decode_target + index_of_value (ac.c:128-131,167-179) without the second division --
		 * (buf - lo) / step < c  <=>  buf - lo < c * step -- as a scalar compare chain: the likely events come first */
		uint32_t step = dec_div(d.rng, evtotal);
		evtotal++;
		/* (E_CTX1 is tested first -- t0 <= off < t1 as ONE unsigned compare -- because it is the usual event: 76 % of the steps of text, 99 % of 16-bit samples) */
		const uint32_t t0 = e0 * step, w1 = e1 * step;
		if (X3_LIKELY(d.off - t0 < w1)) {
			e1++; ac_narrow(d, br, s_cold, t0, w1);
#define DEC_EV X3_E_CTX1
#include "decode_hit.inc"
#undef DEC_EV
		} else {
			if (X3_LIKELY(d.off < t0)) {
				e0++; ac_narrow(d, br, s_cold, 0, t0);
#define DEC_EV X3_E_CTX0
#include "decode_hit.inc"
#undef DEC_EV
			} else {
				const uint32_t t1 = t0 + w1;
				const uint32_t t2 = t1 + e2 * step;
				if (X3_LIKELY(d.off < t2)) {
					e2++; ac_narrow(d, br, s_cold, t1, t2 - t1);
#define DEC_EV X3_E_IDX1
#include "decode_hit.inc"
#undef DEC_EV
				} else {
					const uint32_t t3 = t2 + dec_cold(s_cold, DC_E3) * step;
					if (X3_UNLIKELY(d.off >= t3)) {
						if (X3_LIKELY(d.off < t3 + dec_cold(s_cold, DC_E4) * step)) { dec_cold_set(s_cold, DC_E4, dec_cold(s_cold, DC_E4) + 1); code = X3D_EOF; break; } /* E_EOF, x3.c:295 */
						DEC_FAIL(X3_ST_CORRUPT) /* the reference abort()s, ac.c:178 */
					}
					dec_cold_set(s_cold, DC_E3, dec_cold(s_cold, DC_E3) + 1); ac_narrow(d, br, s_cold, t2, t3 - t2);
					/* ---- decode_match, x3.c:272-283 ---- */
					uint32_t *const models = s_cold + DC_MODELS + lane;
					uint32_t lf = models[0], cf0 = models[X3_WAVE], cf1 = models[2 * X3_WAVE], cf2 = models[3 * X3_WAVE], cf3 = models[4 * X3_WAVE];
					uint32_t len;
					{
						const uint32_t lftotal = 30u + dec_cold(s_cold, DC_E3); /* 32 + the fragments before this one: the event model started at 1 and has counted this one already */
						step = dec_div(d.rng, lftotal);
						const uint32_t incl = x3_wave_incl_scan_u32(lane < 32 ? lf : 0u);
						const uint64_t m = x3_ballot(lane < 32 && d.off < incl * step);
						if (X3_UNLIKELY(!m)) DEC_FAIL(X3_ST_CORRUPT)
						const uint32_t l = (uint32_t)x3_ctz64(m);
						const uint32_t fq = x3_readlane_u32(lf, l), cl = x3_readlane_u32(incl, l) - fq;
						ac_narrow(d, br, s_cold, cl * step, fq * step);
						lf += lane == l ? 1u : 0u;
						len = l + 1;
					}
					const uint32_t litpos = dec_cold(s_cold, DC_LITPOS);
					if (X3_UNLIKELY(tokleft == 0 || (uint64_t)litpos + len > (uint64_t)T.cap + 32)) DEC_FAIL(X3_ST_OUT_FULL)
					uint8_t *const frag = T.lit + litpos;
					uint32_t h = DFNV_OFF, cftotal = x3_wave_sum_u32(cf0 + cf1 + cf2 + cf3);
					/* every new fragment -- one that repeats an element too -- puts its bytes into the output (x3.c:306-326), so a valid stream has coded at most `cap` of them:
					 * an adversarial run of duplicate fragments would otherwise take the byte model's total (256 + bytes coded so far) past the 2^28 the divisions assume */
					if (X3_UNLIKELY(cftotal - 256u + len > T.cap)) DEC_FAIL(X3_ST_OUT_FULL)
					for (uint32_t j = 0; j < len; j++) {
						step = dec_div(d.rng, cftotal);
						const uint32_t s4 = cf0 + cf1 + cf2 + cf3;
						const uint32_t incl = x3_wave_incl_scan_u32(s4);
						const uint64_t m = x3_ballot(d.off < incl * step);
						if (X3_UNLIKELY(!m)) DEC_FAIL(X3_ST_CORRUPT)
						const uint32_t l = (uint32_t)x3_ctz64(m);
						const uint32_t a0 = x3_readlane_u32(cf0, l), a1 = x3_readlane_u32(cf1, l), a2 = x3_readlane_u32(cf2, l), a3 = x3_readlane_u32(cf3, l);
						uint32_t cl = x3_readlane_u32(incl, l) - (a0 + a1 + a2 + a3), sub, fq;
						if (d.off < (cl + a0) * step) { sub = 0; fq = a0; }
						else if (d.off < (cl + a0 + a1) * step) { sub = 1; fq = a1; cl += a0; }
						else if (d.off < (cl + a0 + a1 + a2) * step) { sub = 2; fq = a2; cl += a0 + a1; }
						else { sub = 3; fq = a3; cl += a0 + a1 + a2; }
						ac_narrow(d, br, s_cold, cl * step, fq * step);
						cf0 += (lane == l && sub == 0) ? 1u : 0u; cf1 += (lane == l && sub == 1) ? 1u : 0u;
						cf2 += (lane == l && sub == 2) ? 1u : 0u; cf3 += (lane == l && sub == 3) ? 1u : 0u;
						cftotal++;
						const uint32_t ch = 4 * l + sub;
						if (lane == 0) frag[j] = (uint8_t)ch;
						h = (h ^ ch) * DFNV_MUL;
					}
					models[0] = lf; models[X3_WAVE] = cf0; models[2 * X3_WAVE] = cf1; models[3 * X3_WAVE] = cf2; models[4 * X3_WAVE] = cf3;
					x3_wave_order(); /* the fragment is in memory for every lane */
					/* dict_query_elem (x3.c:309): exact lookup of (len, bytes) -- among the elements' own bytes, the output is not involved */
					uint32_t dup = X3D_NONE;
					uint32_t slot = dht_slot(h, len, T.hlog);
					for (uint32_t e = T.ht[slot]; e != 0; slot = (slot + 1) & T.hmask, e = T.ht[slot]) {
						const uint32_t tg = e - 1;
						if (T.dlen[tg] != len) continue;
						const uint8_t *ds = T.lit + T.dpos[tg];
						uint32_t k = 0;
						while (k < len && ds[k] == frag[k]) k++;
						if (k == len) { dup = tg; break; }
					}
					x3_wave_order();
					dup = x3_uniform(dup);
					uint32_t newtag = dup;
					if (dup == X3D_NONE) { /* x3.c:310-317: a new element; its bytes stay where they are */
						newtag = D;
						uint32_t c1 = 0; /* element 0's context1 block exists from the start (it is the context before any element does) */
						if (D) {
							c1 = (pool_top + 1) & ~1u;
							if (X3_UNLIKELY((uint64_t)c1 + 6 + 2 * X3_WAVE > T.pool_cap)) DEC_FAIL(X3_ST_POOL_FULL)
							pool_top = c1 + 6;
							if (lane == 0) { pool[c1] = 0; pool[c1 + 1] = 0; }
						}
						if (lane == 0) { T.dpos[D] = litpos; T.dlen[D] = (uint8_t)len; T.ht[slot] = D + 1; } /* the hash table and the second stage read these two, wherever the other tables live */
						if (LDS) {
							dec_mtf_to_front(s_mtf, D, D, lane);
							if (lane == 0) { s_idx[D] = 1; if (D) s_c1[D] = c1; }
						} else {
							dec_mtf_to_front(T.gmtf, D, D, lane);
							if (lane == 0) { T.gidx[D] = 1; T.gc1[D] = c1; }
						}
						x3_wave_order();
						if (LDS) { m0 = s_mtf[lane]; i0 = s_idx[lane]; } else { m0 = T.gmtf[lane]; i0 = T.gidx[lane]; }
						dec_cold_set(s_cold, DC_LITPOS, litpos + len);
						D++;
						DEC_MTF_RELOAD
					}
					if (lane == 0) *tokp = newtag;
					tokp++; tokleft--;
					/* x3.c:321-322: both contexts reset -- context1 = tag 0, context0 = the pair (0, 0) if it is known, else pair number 0 (x3.c:142-145) */
					ctx1tag = 0;
					o1 = LDS ? x3_uniform(s_c1[0]) : x3_uniform(T.gc1[0]);
					{ const uint32_t o00 = dec_cold(s_cold, DC_O00); o0 = o00 != X3D_NONE ? o00 : dec_cold(s_cold, DC_OFIRST); }
					ref0 = X3D_NONE;
					x3_wave_order();
					dec_request(pool, o1, o0, lane, pend);
					DEC_PROF_REQ
					if (LDS && X3_UNLIKELY(D == NLDS)) { code = X3D_MIGRATE; break; } /* (only a new fragment adds an element: the hit steps need no such test) */
				}
			}
		}
	}
#undef DEC_FAIL
#undef DEC_PROF_REQ
	S.br = br; S.d = d;
	S.e0 = e0; S.e1 = e1; S.e2 = e2; S.evtotal = evtotal;
	S.D = D; S.npairs = npairs; S.pool_top = pool_top; S.ntok = T.cap - tokleft; S.ctx1tag = ctx1tag;
	S.o0 = o0; S.o1 = o1; S.ref0 = ref0;
#ifdef X3_DEC_PROFILE
	S.pc_wait += pc_wait; S.pc_flight += pc_flight; S.pc_chain += pc_chain;
#endif
	return code;
}

template <uint32_t NLDS>
__device__ static void x3_decode_body(const X3DecArgs &a)
{
	/* per dictionary element while there are fewer than NLDS of them: rank -> tag of the move-to-front list, model_index1 frequency by rank,
	 * pool offset of the element's context1 block */
	X3_LDS uint16_t s_mtf[NLDS];
	X3_LDS uint32_t s_idx[NLDS];
	X3_LDS uint32_t s_c1[NLDS];
	X3_LDS uint32_t s_cold[DC_COUNT + 2 * X3_WAVE + 5 * X3_WAVE]; /* the cold scalars, two blocks of the stream, the models of new fragments */
	static_assert(NLDS >= X3_WAVE && NLDS < 65535, "tags in the LDS recency list are 16 bits wide (0xFFFF: no element); ranks [0, 64) are read without a bounds check");
	const X3DecChunk ck = a.chunks[blockIdx.x];
	const uint32_t lane = x3_lane();
	DecT t;
	t.dpos = a.dict_pos + ck.tag_off; t.dlen = a.dict_len + ck.tag_off;
	t.ht = a.ht + ck.ht_off; t.hlog = ck.ht_log2; t.hmask = (1u << ck.ht_log2) - 1;
	t.gmtf = a.mtf + ck.tag_off; t.gidx = a.idxfreq + ck.tag_off; t.gc1 = a.c1off + ck.tag_off;
	t.pool = a.pool + ck.item_off; t.pool_cap = (uint32_t)ck.item_cap;
	t.tok = a.tokens + ck.tok_off; t.lit = a.lit + ck.lit_off;
	t.cap = ck.out_cap;

	DecS s;
	br_open(s.br, s_cold, a.in + ck.in_off, ck.in_len);
	s.d.lo = 0; s.d.rng = 0x80000000u; /* ac_init */
	s.d.off = br_take(s.br, s_cold, 31); /* ac_decode_init, ac.c:133-140 */
	s.e0 = s.e1 = 1024u; s.e2 = 1u; s.evtotal = 2051; /* create(), x3.c:236-244 */
	dec_cold_set(s_cold, DC_E3, 1u); dec_cold_set(s_cold, DC_E4, 1u);
	for (uint32_t k = 0; k < 5; k++) s_cold[DC_MODELS + k * X3_WAVE + lane] = 1;
	s.D = 0; s.npairs = 0; s.status = X3_ST_OK;
	s.ntok = 0;
	dec_cold_set(s_cold, DC_LITPOS, 0);
	s.ctx1tag = 0;
	s.pc_wait = s.pc_flight = s.pc_chain = 0;
	/* the pool starts with the context1 block of tag 0 (six units: header + two items) and the context0 block of pair number 0 (1 + X3_DEC_CAP0 units), both empty:
	 * these are the contexts of the first step, before any element or pair exists */
	s.o1 = 0; s.o0 = 6; s.ref0 = X3D_NONE;
	dec_cold_set(s_cold, DC_OFIRST, 6); dec_cold_set(s_cold, DC_O00, X3D_NONE);
	s.pool_top = 6 + 1 + X3_DEC_CAP0;
	if (lane == 0) { t.pool[0] = 0; t.pool[1] = 0; t.pool[6] = 0; }
	for (uint32_t i = lane; i < X3_WAVE; i += X3_WAVE) { s_mtf[i] = 0xFFFFu; s_idx[i] = 0; }
	if (lane == 0) s_c1[0] = 0;
	x3_wave_order();

	uint32_t code = dec_loop<NLDS, true>(t, s, s_mtf, s_idx, s_c1, s_cold, lane);
	if (code == X3D_MIGRATE) {
		/* NLDS elements: the tables continue in global memory (dpos/dlen were written there from the start) */
		for (uint32_t i = lane; i < NLDS; i += X3_WAVE) { t.gmtf[i] = s_mtf[i]; t.gidx[i] = s_idx[i]; t.gc1[i] = s_c1[i]; }
		x3_wave_order();
		code = dec_loop<NLDS, false>(t, s, s_mtf, s_idx, s_c1, s_cold, lane);
	}

	if (lane == 0) {
		X3CodeResult r;
		const bool ok = code == X3D_EOF;
		r.out_len = ok ? s.ntok : 0; r.status = ok ? X3_ST_OK : s.status; r.pairs = ok ? s.npairs : 0; r._r = ok ? s.D : 0;
		for (int i = 0; i < 8; i++) r.events[i] = 0;
		r.events[7] = ok ? s.ntok : 0; /* for the second stage */
#ifdef X3_DEC_PROFILE
		r.events[4] = (uint32_t)(s.pc_wait >> 10); r.events[5] = (uint32_t)(s.pc_flight >> 10); r.events[6] = (uint32_t)(s.pc_chain >> 10);
#endif
		if (ok) { r.events[0] = s.e0 - 1024u; r.events[1] = s.e1 - 1024u; r.events[2] = s.e2 - 1u; r.events[3] = s_cold[DC_E3] - 1u; } /* events decoded = what the model counted */
		a.result[blockIdx.x] = r;
	}
}

/* ---- stage 2: tags -> bytes ------------------------------------------------------------------------------------------------------------
 * One workgroup per tile of X3_DEC_TILE tokens; tile_first[] (from the chain's token counts) says which tiles belong to which stream.
 * The chain left its token count in result.events[7]; result.out_len becomes the number of decoded bytes. */
#define X3_DEC2_THREADS 256u
#define X3_DEC2_PER (X3_DEC_TILE / X3_DEC2_THREADS)

__device__ static __forceinline__ uint32_t dec2_stream_of(const uint32_t *tile_first, uint32_t nchunks, uint32_t tile)
{
	uint32_t lo = 0, hi = nchunks; /* last stream with tile_first <= tile */
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (tile_first[mid] <= tile) lo = mid; else hi = mid; }
	return lo;
}

/* bytes of every tile */
__device__ static void x3_dec_lens_body(const X3DecArgs &a)
{
	X3_LDS uint32_t s_sum[X3_DEC2_THREADS / X3_WAVE];
	const uint32_t tile = blockIdx.x, s = dec2_stream_of(a.tile_first, a.nchunks, tile);
	const X3DecChunk ck = a.chunks[s];
	const uint32_t ntok = a.result[s].events[7], t0 = (tile - a.tile_first[s]) * X3_DEC_TILE + threadIdx.x * X3_DEC2_PER;
	const uint32_t *tok = a.tokens + ck.tok_off;
	const uint8_t *dlen = a.dict_len + ck.tag_off;
	uint32_t sum = 0;
	for (uint32_t k = 0; k < X3_DEC2_PER; k++) if (t0 + k < ntok) sum += dlen[tok[t0 + k]];
	sum = x3_wave_sum_u32(sum);
	if (x3_lane() == 0) s_sum[threadIdx.x / X3_WAVE] = sum;
	__syncthreads();
	if (threadIdx.x == 0) { uint32_t tot = 0; for (uint32_t w = 0; w < X3_DEC2_THREADS / X3_WAVE; w++) tot += s_sum[w]; a.tile_sum[tile] = tot; }
}

/* per stream: exclusive prefix sum of its tiles' bytes (in place), the total checked against the capacity */
__device__ static void x3_dec_scan_body(const X3DecArgs &a)
{
	X3_LDS uint32_t s_part[X3_DEC2_THREADS / X3_WAVE];
	X3_LDS uint32_t s_carry, s_full;
	const uint32_t s = blockIdx.x, first = a.tile_first[s], ntile = a.tile_first[s + 1] - first, cap = a.chunks[s].out_cap;
	if (a.result[s].status != X3_ST_OK) return;
	if (threadIdx.x == 0) { s_carry = 0; s_full = 0; }
	__syncthreads();
	for (uint32_t base = 0; base < ntile; base += X3_DEC2_THREADS) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t v = i < ntile ? a.tile_sum[first + i] : 0; /* <= 2048 * 32 per tile: 256 of them fit 32 bits */
		const uint32_t incl = x3_wave_incl_scan_u32(v);
		if (x3_lane() == X3_WAVE - 1) s_part[threadIdx.x / X3_WAVE] = incl;
		__syncthreads();
		uint32_t before = 0;
		for (uint32_t w = 0; w < threadIdx.x / X3_WAVE; w++) before += s_part[w];
		const uint64_t upto = (uint64_t)s_carry + before + incl;
		if (i < ntile) a.tile_sum[first + i] = (uint32_t)(upto - v);
		__syncthreads();
		if (threadIdx.x == X3_DEC2_THREADS - 1) { if (upto > cap) s_full = 1; else s_carry = (uint32_t)upto; }
		__syncthreads();
		if (s_full) break; /* (the carry stays <= the capacity < 2^28: nothing wraps) */
	}
	if (threadIdx.x == 0) {
		if (s_full) a.result[s].status = X3_ST_OUT_FULL;
		else a.result[s].out_len = s_carry;
	}
}

/* the bytes (x3.c:332-340): token i of a tile goes to the tile's base + the lengths of the tokens before it; every thread takes X3_DEC2_PER consecutive tokens */
__device__ static void x3_dec_copy_body(const X3DecArgs &a)
{
	X3_LDS uint32_t s_part[X3_DEC2_THREADS / X3_WAVE];
	const uint32_t tile = blockIdx.x, s = dec2_stream_of(a.tile_first, a.nchunks, tile);
	if (a.result[s].status != X3_ST_OK) return; /* (uniform: the whole workgroup leaves) */
	const X3DecChunk ck = a.chunks[s];
	const uint32_t ntok = a.result[s].events[7], t0 = (tile - a.tile_first[s]) * X3_DEC_TILE + threadIdx.x * X3_DEC2_PER;
	const uint32_t *tok = a.tokens + ck.tok_off;
	const uint8_t *dlen = a.dict_len + ck.tag_off, *lit = a.lit + ck.lit_off;
	const uint32_t *dpos = a.dict_pos + ck.tag_off;
	uint8_t *out = a.out + ck.out_off;
	uint32_t tg[X3_DEC2_PER], ln[X3_DEC2_PER], mine = 0;
	for (uint32_t k = 0; k < X3_DEC2_PER; k++) {
		const bool in = t0 + k < ntok;
		tg[k] = in ? tok[t0 + k] : 0;
		ln[k] = in ? dlen[tg[k]] : 0;
		mine += ln[k];
	}
	const uint32_t incl = x3_wave_incl_scan_u32(mine);
	if (x3_lane() == X3_WAVE - 1) s_part[threadIdx.x / X3_WAVE] = incl;
	__syncthreads();
	uint32_t pos = a.tile_sum[tile] + incl - mine;
	for (uint32_t w = 0; w < threadIdx.x / X3_WAVE; w++) pos += s_part[w];
	for (uint32_t k = 0; k < X3_DEC2_PER; k++) {
		const uint8_t *src = lit + dpos[tg[k]];
		for (uint32_t j = 0; j < ln[k]; j++) out[pos + j] = src[j];
		pos += ln[k];
	}
}

#ifndef X3_EMU
__global__ void __launch_bounds__(X3_WAVE) x3_decode_kernel(X3DecArgs a) { x3_decode_body<X3_DEC_LDS>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_decode_mid_kernel(X3DecArgs a) { x3_decode_body<X3_DEC_LDS_MID>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_decode_eight_kernel(X3DecArgs a) { x3_decode_body<X3_DEC_LDS_EIGHT>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_decode_many_kernel(X3DecArgs a) { x3_decode_body<X3_DEC_LDS_SMALL>(a); }
__global__ void __launch_bounds__(X3_DEC2_THREADS) x3_dec_lens_kernel(X3DecArgs a) { x3_dec_lens_body(a); }
__global__ void __launch_bounds__(X3_DEC2_THREADS) x3_dec_scan_kernel(X3DecArgs a) { x3_dec_scan_body(a); }
__global__ void __launch_bounds__(X3_DEC2_THREADS) x3_dec_copy_kernel(X3DecArgs a) { x3_dec_copy_body(a); }
extern "C" void x3k_launch_decode(const X3DecArgs *a, uint32_t nchunks, hipStream_t st)
{
	/* one wavefront per stream; the LDS tables decide how many streams share a CU: up to one stream per CU gets the big tables (8192
	 * elements before they migrate to global memory), up to one per SIMD the middle ones, up to two per SIMD the next, a batch beyond that the small ones (16 per CU) */
	if (nchunks > 2048) hipLaunchKernelGGL(x3_decode_many_kernel, dim3(nchunks), dim3(X3_WAVE), 0, st, *a);
	else if (nchunks > 1024) hipLaunchKernelGGL(x3_decode_eight_kernel, dim3(nchunks), dim3(X3_WAVE), 0, st, *a);
	else if (nchunks > 256) hipLaunchKernelGGL(x3_decode_mid_kernel, dim3(nchunks), dim3(X3_WAVE), 0, st, *a);
	else hipLaunchKernelGGL(x3_decode_kernel, dim3(nchunks), dim3(X3_WAVE), 0, st, *a);
}
extern "C" void x3k_launch_decode_bytes(const X3DecArgs *a, uint32_t nchunks, uint32_t ntiles, hipStream_t st)
{
	if (!ntiles) return;
	hipLaunchKernelGGL(x3_dec_lens_kernel, dim3(ntiles), dim3(X3_DEC2_THREADS), 0, st, *a);
	hipLaunchKernelGGL(x3_dec_scan_kernel, dim3(nchunks), dim3(X3_DEC2_THREADS), 0, st, *a);
	hipLaunchKernelGGL(x3_dec_copy_kernel, dim3(ntiles), dim3(X3_DEC2_THREADS), 0, st, *a);
}
#else
static void decode_tramp(void *p) { x3_decode_body<X3_DEC_LDS>(*(const X3DecArgs *)p); }
static void dec_lens_tramp(void *p) { x3_dec_lens_body(*(const X3DecArgs *)p); }
static void dec_scan_tramp(void *p) { x3_dec_scan_body(*(const X3DecArgs *)p); }
static void dec_copy_tramp(void *p) { x3_dec_copy_body(*(const X3DecArgs *)p); }
extern "C" void x3k_launch_decode(const X3DecArgs *a, uint32_t nchunks, void *)
{
	x3emu_launch(decode_tramp, (void *)a, dim3(nchunks), dim3(X3_WAVE));
}
extern "C" void x3k_launch_decode_bytes(const X3DecArgs *a, uint32_t nchunks, uint32_t ntiles, void *)
{
	if (!ntiles) return;
	x3emu_launch(dec_lens_tramp, (void *)a, dim3(ntiles), dim3(X3_DEC2_THREADS));
	x3emu_launch(dec_scan_tramp, (void *)a, dim3(nchunks), dim3(X3_DEC2_THREADS));
	x3emu_launch(dec_copy_tramp, (void *)a, dim3(ntiles), dim3(X3_DEC2_THREADS));
}
#endif
