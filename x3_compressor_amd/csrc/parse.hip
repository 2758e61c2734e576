/*
 * parse.hip -- K2: the parse loop (reference: x3.c:372-429) with the dictionary search
 * (dict.c:105-130,148-157) and the dictionary-aware half of find_best_match (backend.c:76-99).
 *
 * What the reference does per step at p: L0 = longest dictionary string that prefixes p; F = find_best_match(p);
 * hit iff L0 exists, nl(L0) >= F and p+L0 <= end (x3.c:383), else a new fragment of min(F, end-p) bytes is coded
 * and, unless already present (x3.c:412), appended to the dictionary (tag = insertion ordinal, dict.c:100).
 * The parse never reads model / context / coder state, so it runs ahead of K3 and hands it a token list.
 *
 * MI355X mapping: one 1024-thread workgroup per stream.
 *   - dictionary = (pos,len) references into the input (an element IS input bytes) + an exact-match hash table
 *     keyed by (len, bytes) in global memory (L2 resident);
 *   - for a block of X3_PARSE_PB positions the workgroup caches in LDS the longest dictionary match L[q]/tag E[q] of
 *     every position (all threads probe all lengths present in the dictionary), plus m[q] from K1;
 *   - wave 0 then walks the parse serially out of LDS: lanes 0..31 evaluate the filters of backend.c:79-90 for the
 *     32 candidate lengths at once, one ballot picks F (closed form, scan.hip header);
 *   - a new element only changes L/E where its bytes occur: all threads patch the cached block (and the hash
 *     table is doubled/rebuilt cooperatively when half full); then wave 0 continues.
 */
#include "x3_kernels.h"

/* PB = positions per cached block (template parameter): PBL = PB + 32 positions have a cached L/E (the filters look 31 ahead),
 * PBB = PB + 64 bytes are staged (a 32-byte compare at the last cached position).  X3_PARSE_PB (2048) for a few streams; a batch
 * that oversubscribes the chip uses X3_PARSE_PB_SMALL (1024): half the LDS, so two workgroups share a CU. */
#ifndef X3_PARSE_PB_SMALL
#define X3_PARSE_PB_SMALL 1024
#endif

#define X3_LDS_HT_LOG2 11u
#define X3_LDS_HT (1u << X3_LDS_HT_LOG2) /* the mirror is used while the table has at most this many slots ... */
#define X3_LDS_DICT (X3_LDS_HT / 2)      /* ... i.e. at most this many elements */
#define X3_BLOOM_WORDS 8u
#ifndef X3_PARSE_HMASK
#define X3_PARSE_HMASK 0xFFFFu /* of the 16 hash bits in a table entry, those a candidate must match (the emulator build of the tests keeps 3 bits: collisions, and with them the exact second search, happen all the time) */
#endif
static_assert(X3_LDS_DICT < 2048, "a mirror entry holds tag + 1 in 11 bits beside the element's length - 1 in 5");

/* The hash of an element / of l bytes at a position: a polynomial (Horner) hash  H(s[0..l)) = sum (s[k] + 1) * B^(l-1-k)  mod 2^32, so that with the
 * prefix hashes P[j] = H(block[0..j)) of the staged block the hash of ANY (position, length) is  P[i+l] - P[i] * B^l  -- two LDS reads and a multiply-subtract,
 * no walk over the bytes.  That is what lets the block fill probe the lengths LONGEST FIRST and stop at the first match: on zero-heavy data (config 5: 24-32
 * element lengths, every one of them matching at every position of a zero run) a position costs one probe instead of one per length -- the incremental
 * hash of rounds 1-4 had to walk up through every shorter length first (fill 393 of 571 Mcycles of config 5's parse).  x3_ph_final mixes the sum before its bits
 * are used (table slot, the 16 tag bits of a mirrored entry, the per-length filter): the top bits of a short string's polynomial hash are nearly constant. */
#define X3_PH_B 0x9E3779B1u
/* inclusive scan of affine maps x -> x * m + a over the wavefront (composition, earlier lane first): the DPP pattern of x3_wave_incl_scan_u32 -- row shifts by 1, 2, 4, 8, then
 * the last lane of a row broadcast to the rows behind it -- with the identity map (1, 0) where a lane has no source; twelve register moves instead of twelve LDS round trips */
__device__ static __forceinline__ void x3_affine_wave_scan(uint32_t &m, uint32_t &a, uint32_t lane)
{
#ifndef X3_EMU
#define X3_AFF_STEP(CTRL, ROWMASK)                                                                                              \
	{                                                                                                                           \
		const uint32_t um = (uint32_t)__builtin_amdgcn_update_dpp(1, (int)m, CTRL, ROWMASK, 0xf, false);                         \
		const uint32_t ua = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, CTRL, ROWMASK, 0xf, false);                         \
		a = ua * m + a; m = um * m;                                                                                             \
	}
	X3_AFF_STEP(0x111, 0xf) X3_AFF_STEP(0x112, 0xf) X3_AFF_STEP(0x114, 0xf) X3_AFF_STEP(0x118, 0xf) /* row_shr:1,2,4,8 */
	X3_AFF_STEP(0x142, 0xa) /* row_bcast:15 -> rows 1, 3 */
	X3_AFF_STEP(0x143, 0xc) /* row_bcast:31 -> rows 2, 3 */
#undef X3_AFF_STEP
	(void)lane;
#else
	for (uint32_t dd = 1; dd < X3_WAVE; dd <<= 1) {
		const uint32_t um = x3_shfl_up_u32(m, dd), ua = x3_shfl_up_u32(a, dd);
		if (lane >= dd) { a = ua * m + a; m = um * m; }
	}
#endif
}
#define X3_PRE_BUCKETS 512u
__device__ static __forceinline__ uint32_t x3_pre_bucket(uint32_t first4) { return (first4 * 0x85EBCA6Bu) >> 23; }
__device__ static __forceinline__ uint32_t x3_ph_step(uint32_t h, uint32_t byte) { return h * X3_PH_B + byte + 1u; }
__device__ static __forceinline__ uint32_t x3_ph_final(uint32_t h) { h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; return h; }

__device__ static __forceinline__ uint32_t ht_slot(uint32_t h, uint32_t len, uint32_t hlog)
{
	uint32_t x = (h ^ (len * 0x9E3779B1u)) * 0x85EBCA6Bu;
	x ^= x >> 15;
	x *= 0xC2B2AE35u;
	return x >> (32 - hlog);
}

/* bytes[gpos .. gpos+l) (global, padded input) == sp[0 .. l) (LDS)?  Four bytes per step from two independent aligned loads:
 * one memory latency per step instead of one per byte (the element bytes were the dependent chain of the block fill). */
__device__ static __forceinline__ bool x3_eq_bytes(const uint8_t *b, uint32_t gpos, const uint8_t *sp, uint32_t l)
{
	for (uint32_t k = 0; k < l; k += 4) {
		const uint32_t *w = (const uint32_t *)(b + ((uint64_t)(gpos + k) & ~(uint64_t)3));
		const uint32_t sh = ((gpos + k) & 3) * 8;
		const uint32_t lo = w[0], hi = w[1];
		const uint32_t g = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
		const uint32_t v = (uint32_t)sp[k] | (uint32_t)sp[k + 1] << 8 | (uint32_t)sp[k + 2] << 16 | (uint32_t)sp[k + 3] << 24;
		const uint32_t rem = l - k;
		const uint32_t mask = rem >= 4 ? 0xFFFFFFFFu : ((1u << (8 * rem)) - 1);
		if ((g ^ v) & mask) return false;
	}
	return true;
}

/* eight bytes of an LDS byte array at any offset, little-endian: three aligned dword reads + funnel shifts */
__device__ static __forceinline__ uint64_t x3_lds_load8(const uint8_t *sp, uint32_t i)
{
	const uint32_t *w = (const uint32_t *)(sp + (i & ~3u));
	const uint32_t sh = (i & 3u) * 8u;
	const uint32_t a = w[0], b = w[1], c = w[2];
	const uint32_t lo = sh ? (a >> sh) | (b << (32u - sh)) : a, hi = sh ? (b >> sh) | (c << (32u - sh)) : b;
	return (uint64_t)lo | ((uint64_t)hi << 32);
}

/* state shared by the workgroup */
struct ParseShared {
	uint32_t p, blk, D, lenmask, hlog, flag, ntok, hits, mbytes;
	uint32_t new_pos, new_len, new_tag, rebuild, nanchor, nck;
};
enum { FLAG_REFILL = 1, FLAG_PATCH = 2, FLAG_DONE = 3 };

template <uint32_t PB>
__device__ static void x3_parse_body(const X3ParseArgs &a)
{
	constexpr uint32_t PBL = PB + 32, PBB = PB + 64;
	alignas(16) X3_LDS uint8_t sb[PBB + 8];
	X3_LDS uint8_t sL[PBL];
	X3_LDS uint32_t sE[PBL];
	X3_LDS uint8_t sM[PB];
	X3_LDS uint2 sN[PB]; /* the parse step at a cached position: .x = tag of the hit element, .y = 0x80 | L0 for a hit, else the new fragment's length (one 64-bit LDS read per step) */
	X3_LDS uint32_t sJ[6][PB]; /* jump tables by pointer doubling: sJ[r][i] = (cached index after up to 2^r consecutive hits from i) | (hits taken << 12) */
	X3_LDS uint16_t sAi[PB];   /* anchors of the current walk: cached index where a stride of hits starts ... */
	X3_LDS uint32_t sAt[PB];   /* ... and the token index of its first hit */
	/* LDS mirror of the dictionary while it is small (the usual case at large -t): hash table, element position / length */
	X3_LDS uint32_t sHT[X3_LDS_HT];   /* the table's mirror: 0 = empty, else tag + 1 (11 bits: at most X3_LDS_DICT elements) | (length - 1) << 11 | (top 16 bits of the element's hash) << 16 */
	X3_LDS uint32_t sDpos[X3_LDS_DICT];
	X3_LDS uint8_t sDlen[X3_LDS_DICT];
	X3_LDS uint2 sD8[X3_LDS_DICT];   /* the first eight bytes of each mirrored element (zero beyond its length): most candidates are settled without touching global memory */
	X3_LDS uint32_t sDh[X3_LDS_DICT];  /* ... and its 32-bit hash (the one the table is addressed with) */
	X3_LDS uint32_t sBloom[32][X3_BLOOM_WORDS]; /* per element length: a 256-bit filter on the top bits of the element's (mixed) hash -- a probe whose bit is clear has no element to find
	                                              * (the filter never forgets: the dictionary only grows) */
	X3_LDS uint32_t sSm[(PBL + 2 * X3_WAVE) / 32 + 2]; /* one bit per cached position: an element matching there is short enough to let a candidate length through (step table) */
	X3_LDS uint32_t sPre[X3_PRE_BUCKETS]; /* by the first four bytes (hashed): the lengths >= 4 of the elements that begin with them, one bit per length -- ONE read tells a position
	                                       * which of the 24-32 lengths present can match at all (a filter: it never forgets, the dictionary only grows) */
	X3_LDS uint32_t sP[PBB + 1];  /* prefix hashes of the staged bytes: sP[j] = H(sb[0..j)) */
	X3_LDS uint32_t sPow[33];     /* B^l */
	X3_LDS uint2 sPw[X3_PARSE_THREADS / X3_WAVE]; /* the prefix scan's per-wave totals */
	X3_LDS ParseShared S;

	const X3Chunk ck = a.chunks[blockIdx.x];
	const uint32_t n = ck.len;
	const uint8_t *b = a.bytes + ck.byte_off;
	const uint8_t *mm = a.m + ck.byte_off;
	uint32_t *dpos = a.dict_pos + ck.elem_off;
	uint8_t *dlen = a.dict_len + ck.elem_off;
	uint32_t *ht = a.ht + ck.ht_off;
	uint32_t *tinf = a.tok_info + ck.elem_off;
	const uint32_t tid = threadIdx.x, lane = x3_lane(), wave = tid / X3_WAVE;
	const uint32_t f1 = a.factor1, f2 = a.factor2;

	if (tid == 0) {
		S.p = 0; S.blk = 0; S.D = 0; S.lenmask = 0; S.hlog = X3_HT_LOG2_MIN < ck.ht_log2_max ? X3_HT_LOG2_MIN : ck.ht_log2_max;
		S.flag = n ? FLAG_REFILL : FLAG_DONE; S.ntok = 0; S.hits = 0; S.rebuild = 0; S.mbytes = 0; S.nanchor = 0; S.nck = 0;
	}
	for (uint32_t i = tid; i < (1u << (X3_HT_LOG2_MIN < ck.ht_log2_max ? X3_HT_LOG2_MIN : ck.ht_log2_max)); i += X3_PARSE_THREADS) ht[i] = 0;
	for (uint32_t i = tid; i < X3_LDS_HT; i += X3_PARSE_THREADS) sHT[i] = 0;
	for (uint32_t i = tid; i < 32u * X3_BLOOM_WORDS; i += X3_PARSE_THREADS) (&sBloom[0][0])[i] = 0;
	for (uint32_t i = tid; i < X3_PRE_BUCKETS; i += X3_PARSE_THREADS) sPre[i] = 0;
	if (tid == 0) { uint32_t pw = 1u; for (uint32_t l = 0; l <= 32u; l++) { sPow[l] = pw; pw *= X3_PH_B; } }
	__syncthreads();

	uint64_t cyc_fill = 0, cyc_patch = 0, cyc_table = 0, cyc_walk = 0, t_prev = x3_clock();
	while (S.flag != FLAG_DONE) { /* S.flag is only written between the two barriers below: uniform */
		const uint32_t flag = S.flag;
		if (flag == FLAG_REFILL) {
			/* ---- stage bytes, m, and the longest dictionary match of every position of the block ---- */
			const uint32_t blk = S.p;
			const uint32_t lenmask = S.lenmask, hlog = S.hlog, hmask = (1u << hlog) - 1;
			for (uint32_t i = tid; i < PBB; i += X3_PARSE_THREADS) sb[i] = b[(uint64_t)blk + i];
			for (uint32_t i = tid; i < PB; i += X3_PARSE_THREADS) sM[i] = mm[(uint64_t)blk + i];
			__syncthreads();
			/* prefix hashes of the block: every thread folds its PER consecutive bytes into an affine map x -> x * mul + add, the maps are composed by a scan over the
			 * workgroup (wavefront: six shuffle steps; the sixteen wave totals: a short serial fold), and every thread writes the PER prefixes behind its start value */
			{
				constexpr uint32_t PER = (PBB + X3_PARSE_THREADS - 1) / X3_PARSE_THREADS;
				const uint32_t j0 = tid * PER;
				uint32_t mul = 1u, add = 0u;
				for (uint32_t k = 0; k < PER; k++) if (j0 + k < PBB) { add = x3_ph_step(add, sb[j0 + k]); mul *= X3_PH_B; }
				uint32_t im = mul, ia = add; /* inclusive composition over the lanes up to this one */
				x3_affine_wave_scan(im, ia, lane);
				if (lane == X3_WAVE - 1) sPw[wave] = make_uint2(im, ia);
				__syncthreads();
				uint32_t x = 0u; /* value before this wave's first byte */
				for (uint32_t w = 0; w < wave; w++) { const uint2 t = sPw[w]; x = x * t.x + t.y; }
				/* value before this thread's first byte: the wave's start through the lanes before this one */
				const uint32_t pm = x3_wave_shr1_u32(im), pa = x3_wave_shr1_u32(ia);
				if (lane) x = x * pm + pa;
				if (tid == 0) sP[0] = 0u;
				for (uint32_t k = 0; k < PER; k++) if (j0 + k < PBB) { x = x3_ph_step(x, sb[j0 + k]); sP[j0 + k + 1] = x; }
			}
			__syncthreads();
			for (uint32_t i0 = 0; i0 < PBL; i0 += X3_PARSE_THREADS) { /* (every thread takes every trip: the ballot at the end is the wave's) */
				const uint32_t i = i0 + tid;
				uint32_t best = 0, btag = 0;
				if (i < PBL) {
				const uint64_t my8 = x3_lds_load8(sb, i); /* the position's first eight bytes, compared with the mirrored elements' */
				const uint32_t Pi = sP[i];
				/* Lengths are probed LONGEST FIRST and the search stops at the first match (header of the hash above).  Mirrored dictionary: a candidate is accepted on
				 * the 16 hash bits its table entry carries (one LDS access per probe) and compared in full -- first 8 bytes in LDS, the rest with the element in global
				 * memory -- once the search is over.  Should that comparison fail (a hash collision, ~2^-16 per probe), the position is searched again with every
				 * candidate compared in full: the result is exact either way. */
				/* the lengths worth a probe: 1-3 if present, longer ones only if some element of that length begins with this position's first four bytes */
				const uint32_t cand0 = lenmask & (7u | sPre[x3_pre_bucket((uint32_t)my8)]);
				/* ... of those, the lengths whose hash passes the per-length filter: four candidates per trip, no branch on what the LDS returns, so that the reads of a trip
				 * (two per candidate) are in flight together -- one candidate after the other, every probe was three dependent LDS round trips */
				uint32_t surv = 0;
				for (uint32_t cand = cand0; cand; ) {
					constexpr uint32_t U = 4u;
					uint32_t l4[U], h4[U];
#pragma unroll
					for (uint32_t u = 0; u < U; u++) { l4[u] = 32u - (uint32_t)x3_clz32(cand); cand &= ~(l4[u] ? 1u << (l4[u] - 1u) : 0u); } /* (cand == 0: l = 0, nothing cleared; l = 32 is a length like any other: no shift by 32) */
#pragma unroll
					for (uint32_t u = 0; u < U; u++) h4[u] = x3_ph_final(sP[i + l4[u]] - Pi * sPow[l4[u]]);
#pragma unroll
					for (uint32_t u = 0; u < U; u++) {
						const uint32_t bit = (sBloom[(l4[u] - 1u) & 31u][h4[u] >> 29] >> ((h4[u] >> 24) & 31u)) & 1u;
						surv |= l4[u] ? bit << (l4[u] - 1u) : 0u;
					}
				}
				for (uint32_t pass = 0; pass < 2; pass++) {
					best = 0; btag = 0;
					for (uint32_t cand = surv; cand && !best; ) {
						const uint32_t l = 32u - (uint32_t)x3_clz32(cand);
						cand ^= 1u << (l - 1);
						const uint32_t h = x3_ph_final(sP[i + l] - Pi * sPow[l]);
						uint32_t slot = ht_slot(h, l, hlog);
						if (hlog <= X3_LDS_HT_LOG2) { /* wave-uniform: the whole dictionary is mirrored in LDS */
							const uint64_t m8 = l >= 8 ? ~(uint64_t)0 : (((uint64_t)1 << (8 * l)) - 1);
							for (uint32_t e = sHT[slot]; e != 0; slot = (slot + 1) & hmask, e = sHT[slot]) {
								if (((e >> 11) & 31u) != l - 1) continue; /* the entry carries its element's length and 16 bits of its hash */
								const uint32_t tag = (e & 0x7FFu) - 1;
								if (pass == 0) { if (((e ^ h) >> 16) & X3_PARSE_HMASK) continue; } /* first search: ONE LDS access per probe, candidates accepted on the hash bits */
								else {
									const uint2 d8 = sD8[tag];
									if ((((uint64_t)d8.x | ((uint64_t)d8.y << 32)) ^ my8) & m8) continue; /* (not this element: an exact table, so the probe goes on) */
									if (l > 8 && !x3_eq_bytes(b, sDpos[tag] + 8, sb + i + 8, l - 8)) continue;
								}
								best = l; btag = tag; break;
							}
						} else {
							for (uint32_t e = ht[slot]; e != 0; slot = (slot + 1) & hmask, e = ht[slot]) {
								const uint32_t tag = e - 1;
								if (dlen[tag] != l) continue;
								if (x3_eq_bytes(b, dpos[tag], sb + i, l)) { best = l; btag = tag; break; }
							}
						}
					}
					if (pass == 0 && hlog <= X3_LDS_HT_LOG2 && best) { /* the one candidate that counts, compared in full */
						const uint2 d8 = sD8[btag];
						const uint64_t m8 = best >= 8 ? ~(uint64_t)0 : (((uint64_t)1 << (8 * best)) - 1);
						if (((((uint64_t)d8.x | ((uint64_t)d8.y << 32)) ^ my8) & m8) || (best > 8 && !x3_eq_bytes(b, sDpos[btag] + 8, sb + i + 8, best - 8))) continue; /* (a collision: once more, exactly) */
					}
					break;
				}
				sL[i] = (uint8_t)best;
				sE[i] = btag;
				}
				/* the step table's filter (below): one bit per cached position, "an element matching here is short enough to let a candidate length through" */
				const uint64_t bal = x3_ballot(i < PBL && (best == 0 || (uint64_t)best * f1 <= 33u));
				if (lane == 0 && i < PBL + X3_WAVE) { sSm[i >> 5] = (uint32_t)bal; sSm[(i >> 5) + 1] = (uint32_t)(bal >> 32); } /* (one wavefront behind the last position too: zeros a window may read) */
			}
			if (tid == 0) S.blk = blk;
		} else if (flag == FLAG_PATCH) {
			/* ---- a new element: grow the table if half full, then patch the cached block ---- */
			const uint32_t blk = S.blk, np = S.new_pos, nl = S.new_len, nt = S.new_tag;
			if (S.rebuild) {
				const uint32_t hlog = S.hlog, hmask = (1u << hlog) - 1, D = S.D;
				for (uint32_t i = tid; i <= hmask; i += X3_PARSE_THREADS) ht[i] = 0;
				__syncthreads();
				for (uint32_t t = tid; t < D; t += X3_PARSE_THREADS) {
					const uint8_t *ds = b + dpos[t];
					const uint32_t l = dlen[t];
					uint32_t h = 0u;
					for (uint32_t k = 0; k < l; k++) h = x3_ph_step(h, ds[k]);
					h = x3_ph_final(h);
					uint32_t slot = ht_slot(h, l, hlog);
					while (atomicCAS(&ht[slot], 0u, t + 1) != 0u) slot = (slot + 1) & hmask;
				}
				if (hlog <= X3_LDS_HT_LOG2) { /* same content as the global table (identical probe order is not needed: exact lookups) */
					__syncthreads();
					for (uint32_t i = tid; i <= hmask; i += X3_PARSE_THREADS) { const uint32_t e = ht[i]; sHT[i] = e ? (e | ((uint32_t)(sDlen[e - 1] - 1u) << 11) | (sDh[e - 1] & 0xFFFF0000u)) : 0u; }
				}
			}
			const uint32_t first = S.p - blk; /* positions before the parse pointer are never read again */
			const uint32_t src = np - blk;    /* the element's bytes are inside the staged block */
			for (uint32_t i0 = first & ~(X3_WAVE - 1u); i0 < PBL; i0 += X3_PARSE_THREADS) { /* (whole wavefronts, every thread every trip: the filter bits of a wavefront's 64 positions are one ballot) */
				const uint32_t i = i0 + tid;
				uint32_t Li = i < PBL ? sL[i] : 1u;
				if (i >= first && i < PBL && Li < nl) {
					uint32_t k = 0;
					while (k < nl && sb[i + k] == sb[src + k]) k++;
					if (k == nl) { sL[i] = (uint8_t)nl; sE[i] = nt; Li = nl; }
				}
				const uint64_t bal = x3_ballot(i < PBL && (Li == 0 || (uint64_t)Li * f1 <= 33u));
				if (lane == 0 && i < PBL + X3_WAVE) { sSm[i >> 5] = (uint32_t)bal; sSm[(i >> 5) + 1] = (uint32_t)(bal >> 32); }
			}
		}
		__syncthreads();
		{ const uint64_t t = x3_clock(); if (flag == FLAG_REFILL) cyc_fill += t - t_prev; else cyc_patch += t - t_prev; t_prev = t; }
		{
			/* ---- the step every cached position WOULD take under the current dictionary (backend.c:76-99 closed form +
			 * x3.c:383,402-404).  It only changes when a new element patches L[], so all threads (re)build the table and the
			 * serial walker below just follows it. ---- */
			const uint32_t blk = S.blk, first = S.p >= blk ? S.p - blk : 0;
			/* backend.c:79-83 with factor2 == 0: candidate length k + 1 survives iff k == 1, or no element matches at p + k, or its length L obeys L * factor1 <= k + 1.
			 * k <= 32, so only positions with L == 0 or L * factor1 <= 33 can let anything through: one bit per cached position (a ballot per 64), and a position looks at
			 * the bits of ITS window from the top -- zero-heavy data (every L is 9 or more: no bit) is settled without a loop, text by its first candidate, where the
			 * plain loop walked all m[p] <= 31 lengths upwards (config 5: 130 of 385 Mcycles of the parse) */
			for (uint32_t i = first + tid; i < PB; i += X3_PARSE_THREADS) {
				const uint32_t q = blk + i;
				if (q >= n) break;
				const uint32_t mp = sM[i];
				uint32_t best = 0;
				int vmax = -(1 << 20);
				if (f2 == 0 && f1 > 0) {
					best = mp ? 1u : 0u;
					if (mp >= 2) {
						const uint32_t wi = (i + 2) >> 5, sh = (i + 2) & 31;
						const uint64_t two = (uint64_t)sSm[wi] | ((uint64_t)sSm[wi + 1] << 32);
						uint32_t w = (uint32_t)(two >> sh) & (mp >= 33 ? 0xFFFFFFFFu : ((1u << (mp - 1)) - 1u)); /* bit t: k = t + 2 */
						while (w) {
							const uint32_t t = 31u - (uint32_t)x3_clz32(w), k = t + 2, Lk = sL[i + k];
							if (Lk == 0 || (uint64_t)Lk * f1 <= (uint64_t)(k + 1)) { best = k; break; }
							w ^= 1u << t;
						}
					}
				} else
				for (uint32_t k = 1; k <= mp; k++) {
					const uint32_t Lk = sL[i + k];
					bool ok = !(k >= 2 && f1 > 0 && Lk != 0 && (uint64_t)Lk * f1 > (uint64_t)(k + 1));
					if (f2 > 0) {
						if (Lk != 0 && (int)Lk - (int)k > vmax) vmax = (int)Lk - (int)k;
						if ((int64_t)vmax * (int64_t)(int)f2 > (int64_t)(k + 1)) ok = false;
					}
					if (ok) best = k;
				}
				const uint32_t F = best + 1, L0 = sL[i];
				uint32_t nlL0 = L0;
				if (a.nl_mode) nlL0 = L0 == 1 ? 1 : L0 == 2 ? 4 : L0 == 3 ? 6 : L0 == 4 ? 8 : 9999; /* x3.c:357-370 */
				uint2 e;
				e.x = sE[i];
				e.y = (L0 != 0 && nlL0 >= F && q + L0 <= n) ? (0x80u | L0) : (q + F > n ? n - q : F);
				sN[i] = e;
			}
			__syncthreads();
			/* jump tables: a miss, the block end and the input end are absorbing (0 hits taken) */
			for (uint32_t i = first + tid; i < PB; i += X3_PARSE_THREADS) {
				const uint32_t st = sN[i].y;
				sJ[0][i] = (blk + i < n && (st & 0x80u)) ? ((i + (st & 0x7Fu)) | (1u << 12)) : i;
			}
			for (uint32_t r = 1; r < 6; r++) {
				__syncthreads();
				for (uint32_t i = first + tid; i < PB; i += X3_PARSE_THREADS) {
					const uint32_t e1 = sJ[r - 1][i];
					uint32_t pos = e1 & 0xFFFu, cnt = e1 >> 12;
					if (pos < PB && pos >= first) { const uint32_t e2 = sJ[r - 1][pos]; pos = e2 & 0xFFFu; cnt += e2 >> 12; }
					sJ[r][i] = pos | (cnt << 12);
				}
			}
		}
		__syncthreads();
		{ const uint64_t t = x3_clock(); cyc_table += t - t_prev; t_prev = t; }

		if (wave == 0) {
			/* ---- serial parse out of LDS.  One wavefront issues ~1 instruction / 5 cycles, so the loop is written for
			 * instruction count: one table lookup per step, a token is one lane insert (stored 64 at a time), and nothing
			 * but p / ntok / hits is tracked -- positions and the running
			 * counts K3 needs are prefix sums over the token list, computed in parallel afterwards (api.hip). ---- */
			uint32_t p = x3_uniform(S.p), ntok = x3_uniform(S.ntok), hits = x3_uniform(S.hits), D = x3_uniform(S.D);
			uint32_t lenmask = x3_uniform(S.lenmask), hlog = x3_uniform(S.hlog), mbytes = x3_uniform(S.mbytes);
			const uint32_t blk = x3_uniform(S.blk);
			uint32_t out_flag = 0, na = 0;
			for (;;) {
				if (p >= n) { out_flag = FLAG_DONE; break; }
				const uint32_t idx = p - blk;
				if (idx >= PB) { out_flag = FLAG_REFILL; break; }
				const uint32_t jmp = x3_uniform(sJ[5][idx]);
				const uint32_t cnt = jmp >> 12;
				if (cnt) { /* up to 32 dictionary hits in one lookup; their tokens are written by the expansion phase */
					sAi[na] = (uint16_t)idx;
					sAt[na] = ntok;
					na++;
					ntok += cnt; hits += cnt;
					p = blk + (jmp & 0xFFFu);
					continue;
				}
				const uint32_t len = x3_uniform(sN[idx].y), L0 = x3_uniform(sL[idx]);
				/* x3.c:412 : is this exact fragment already an element? */
				int dup = 0;
				if (L0 == len) dup = 1;
				else if (L0 > len && ((lenmask >> (len - 1)) & 1)) {
					const uint32_t h = x3_ph_final(sP[idx + len] - sP[idx] * sPow[len]);
					const uint32_t hmask = (1u << hlog) - 1;
					uint32_t slot = ht_slot(h, len, hlog);
					for (uint32_t e = ht[slot]; e != 0; slot = (slot + 1) & hmask, e = ht[slot]) {
						const uint32_t tg = e - 1;
						if (dlen[tg] != len) continue;
						const uint8_t *ds = b + dpos[tg];
						uint32_t k = 0;
						while (k < len && ds[k] == sb[idx + k]) k++;
						if (k == len) { dup = 1; break; }
					}
				}
				if (lane == 0) tinf[ntok] = X3_TOK_MISS | (dup ? X3_TOK_DUP : 0u) | len;
				ntok++;
				mbytes += len;
				x3_wave_sync(); /* every lane has finished probing before lane 0 inserts */
				if (!dup) {
					const uint32_t ntag = D;
					uint32_t rebuild = 0;
					if (2 * (D + 1) > (1u << hlog) && hlog < ck.ht_log2_max) { hlog++; rebuild = 1; }
					if (lane == 0) {
						dpos[ntag] = p;
						dlen[ntag] = (uint8_t)len;
						if (ntag < X3_LDS_DICT) {
							sDpos[ntag] = p; sDlen[ntag] = (uint8_t)len;
							uint64_t v8 = 0;
							for (uint32_t k = 0; k < len && k < 8; k++) v8 |= (uint64_t)sb[idx + k] << (8 * k);
							sD8[ntag] = make_uint2((uint32_t)v8, (uint32_t)(v8 >> 32));
						}
						const uint32_t h = x3_ph_final(sP[idx + len] - sP[idx] * sPow[len]);
						sBloom[len - 1][h >> 29] |= 1u << ((h >> 24) & 31u);
						if (len >= 4) sPre[x3_pre_bucket((uint32_t)sb[idx] | (uint32_t)sb[idx + 1] << 8 | (uint32_t)sb[idx + 2] << 16 | (uint32_t)sb[idx + 3] << 24)] |= 1u << (len - 1);
						if (ntag < X3_LDS_DICT) sDh[ntag] = h;
						if (!rebuild) {
							const uint32_t hmask = (1u << hlog) - 1;
							uint32_t slot = ht_slot(h, len, hlog);
							while (ht[slot] != 0) slot = (slot + 1) & hmask;
							ht[slot] = ntag + 1;
							if (hlog <= X3_LDS_HT_LOG2) sHT[slot] = (ntag + 1) | ((len - 1) << 11) | (h & 0xFFFF0000u);
						}
						S.new_pos = p; S.new_len = len; S.new_tag = ntag; S.rebuild = rebuild;
					}
					D++;
					lenmask |= 1u << (len - 1);
					p += len;
					out_flag = p >= n ? FLAG_DONE : FLAG_PATCH;
					break;
				}
				p += len;
			}
			x3_wave_sync(); /* every lane has read the shared walker state before lane 0 replaces it */
			if (lane == 0) {
				S.nanchor = na;
				S.p = p; S.ntok = ntok; S.hits = hits; S.D = D; S.lenmask = lenmask; S.hlog = hlog; S.mbytes = mbytes;
				S.flag = out_flag;
			}
		}
		__syncthreads();
		{
			/* ---- expansion: anchor a covers cnt <= 32 consecutive hits; thread (a, t) finds the t-th one through the 1/2/4/8/16 tables ---- */
			const uint32_t na = S.nanchor;
			for (uint32_t w = tid; w < na * 32; w += X3_PARSE_THREADS) {
				const uint32_t an = w >> 5, t = w & 31;
				uint32_t i = sAi[an];
				if (t < (sJ[5][i] >> 12)) {
					if (t & 16) i = sJ[4][i] & 0xFFFu;
					if (t & 8) i = sJ[3][i] & 0xFFFu;
					if (t & 4) i = sJ[2][i] & 0xFFFu;
					if (t & 2) i = sJ[1][i] & 0xFFFu;
					if (t & 1) i = sJ[0][i] & 0xFFFu;
					tinf[sAt[an] + t] = sN[i].x;
				}
			}
		}
		__syncthreads();
		{ const uint64_t t = x3_clock(); cyc_walk += t - t_prev; t_prev = t; }
		if (a.ckpt && S.nck < a.nckpt && S.p >= a.ckpt_pos[blockIdx.x * X3_MAX_CKPT + S.nck] && S.flag != FLAG_DONE) { /* uniform: S is stable between barriers */
			/* every token below S.ntok (and dict_len of every element) has been stored by SOME thread of this workgroup: each
			 * thread releases its own stores to device scope, then one thread publishes the counters to the host */
			__threadfence();
			__syncthreads();
			if (tid == 0) {
				uint32_t k = S.nck;
				while (k + 1 < a.nckpt && S.p >= a.ckpt_pos[blockIdx.x * X3_MAX_CKPT + k + 1]) k++; /* a long step may cross several marks: publish the last one only */
				X3ParseCkpt *ck = a.ckpt + (size_t)blockIdx.x * X3_CKPT_SLOTS + k;
				ck->p = S.p; ck->ntok = S.ntok; ck->hits = S.hits; ck->dict_elems = S.D; ck->miss_bytes = S.mbytes;
				__threadfence_system();
				ck->seq = k + 1;
				__threadfence_system();
				S.nck = k + 1;
			}
			__syncthreads();
		}
	}
	if (a.ckpt) { /* the chunk is parsed: its "done" record (the other chunks of the batch may still be running) */
		__threadfence();
		__syncthreads();
		if (tid == 0) {
			X3ParseCkpt *ck = a.ckpt + (size_t)blockIdx.x * X3_CKPT_SLOTS + X3_MAX_CKPT;
			ck->p = S.p; ck->ntok = S.ntok; ck->hits = S.hits; ck->dict_elems = S.D; ck->miss_bytes = S.mbytes;
			__threadfence_system();
			ck->seq = X3_MAX_CKPT + 1;
			__threadfence_system();
		}
	}

	if (tid == 0) {
		X3ParseResult r;
		r.ntok = S.ntok; r.dict_elems = S.D; r.hits = S.hits; r.status = X3_ST_OK; r.miss_bytes = S.mbytes;
		r.kcyc_fill = (uint32_t)(cyc_fill >> 10); r.kcyc_patch = (uint32_t)(cyc_patch >> 10); r.kcyc_walk = (uint32_t)(cyc_walk >> 10);
		(void)cyc_table; /* folded into patch/fill by the caller's view: table time is reported with the walk */
		r.kcyc_walk += (uint32_t)(cyc_table >> 10) * 0; r._r0 = (uint32_t)(cyc_table >> 10);
		a.result[blockIdx.x] = r;
	}
}

#ifndef X3_EMU
__global__ void __launch_bounds__(X3_PARSE_THREADS) x3_parse_kernel(X3ParseArgs a) { x3_parse_body<X3_PARSE_PB>(a); }
__global__ void __launch_bounds__(X3_PARSE_THREADS, 8) x3_parse_many_kernel(X3ParseArgs a) { x3_parse_body<X3_PARSE_PB_SMALL>(a); }

extern "C" void x3k_launch_parse(const X3ParseArgs *a, uint32_t nchunks, hipStream_t st)
{
	if (nchunks > 256) hipLaunchKernelGGL(x3_parse_many_kernel, dim3(nchunks), dim3(X3_PARSE_THREADS), 0, st, *a); /* two workgroups per CU */
	else hipLaunchKernelGGL(x3_parse_kernel, dim3(nchunks), dim3(X3_PARSE_THREADS), 0, st, *a);
}
#else
static void parse_tramp(void *p) { x3_parse_body<X3_PARSE_PB>(*(const X3ParseArgs *)p); }
extern "C" void x3k_launch_parse(const X3ParseArgs *a, uint32_t nchunks, void *)
{
	x3emu_launch(parse_tramp, (void *)a, dim3(nchunks), dim3(X3_PARSE_THREADS));
}
#endif
