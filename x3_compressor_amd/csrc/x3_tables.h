/*
 * x3_tables.h -- wave-parallel primitives over the growable tables of the coding stage, shared by the v1 encoder kernel
 * (code.hip) and the decoder kernel (decode.hip): butterfly sum, context item lists (context.c), the tag-pair map
 * (tag_pair.c) and the move-to-front list (dict.c:132-146).  All of them must be called in wave-uniform control flow.
 */
#ifndef X3_TABLES_H
#define X3_TABLES_H
#include "x3_kernels.h"

__device__ static __forceinline__ uint32_t wave_sum(uint32_t v) { return x3_wave_sum_u32(v); }

struct CtxQ { uint32_t found, pos, freq, cum; };

/* one sweep of a context's items: position of `tag`, its freq and the cumulative freq before it
 * (ctx_query_tag_item / ctx_query_tag_index / count_cum_freqs, context.c:20-40,95-133) */
__device__ static CtxQ ctx_query(const X3CtxHdr h, const uint64_t *pool, uint32_t tag, uint32_t lane, bool have_first = false, uint64_t first = 0)
{
	/* have_first: the caller already holds items [0, 64) of the list (one per lane, 0 beyond the end) -- the decoder prefetches them */
	CtxQ q;
	q.found = 0; q.pos = 0; q.freq = 0; q.cum = 0;
	for (uint32_t base = 0; base < h.items; base += X3_WAVE) {
		const uint32_t i = base + lane;
		const uint64_t it = (have_first && base == 0) ? first : (i < h.items ? pool[(uint64_t)h.off + i] : 0);
		const uint32_t fq = (uint32_t)it;
		const uint64_t mask = x3_ballot(i < h.items && (uint32_t)(it >> 32) == tag);
		if (mask) {
			const uint32_t l = (uint32_t)x3_ctz64(mask);
			q.found = 1;
			q.pos = base + l;
			q.freq = x3_bcast_u32(fq, (int)l);
			q.cum += wave_sum(lane < l ? fq : 0u);
			break;
		}
		q.cum += wave_sum(fq);
	}
	return q;
}

/* x3.c:197-209 : add the tag with freq 1 or bump its freq; total tracks calc_total_freq */
__device__ static void ctx_touch(X3CtxHdr *hp, X3CtxHdr h, const CtxQ q, uint32_t tag, uint64_t *pool,
                                 uint64_t &pool_top, uint64_t pool_cap, uint32_t &status, uint32_t lane)
{
	if (q.found) {
		if (lane == 0) pool[(uint64_t)h.off + q.pos] += 1;
	} else {
		if (h.items == h.cap) {
			const uint32_t ncap = h.cap ? 2 * h.cap : 2;
			if (pool_top + ncap > pool_cap) { status = X3_ST_POOL_FULL; return; }
			const uint32_t noff = (uint32_t)pool_top;
			pool_top += ncap;
			for (uint32_t i = lane; i < h.items; i += X3_WAVE) pool[(uint64_t)noff + i] = pool[(uint64_t)h.off + i];
			h.off = noff;
			h.cap = ncap;
		}
		if (lane == 0) pool[(uint64_t)h.off + h.items] = ((uint64_t)tag << 32) | 1u;
		h.items++;
	}
	h.total++;
	if (lane == 0) *hp = h;
}

__device__ static __forceinline__ uint32_t pair_slot(uint64_t key, uint32_t plog)
{
	return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - plog));
}

/* tags [0,r) move one rank down, `tag` goes to rank 0 (dict.c:132-146 after dict_set_last_pos / dict_insert_elem) */
__device__ static void mtf_to_front(uint32_t *mtf, uint32_t r, uint32_t tag, uint32_t lane)
{
	for (int base = (int)(r & ~(uint32_t)(X3_WAVE - 1)); base >= 0; base -= X3_WAVE) {
		const uint32_t j = (uint32_t)base + lane;
		const bool act = j >= 1 && j <= r;
		const uint32_t v = act ? mtf[j - 1] : 0;
		x3_wave_sync(); /* every lane has read before any lane overwrites its neighbour's source */
		if (act) mtf[j] = v;
	}
	if (lane == 0) mtf[0] = tag;
}

#endif /* X3_TABLES_H */
