/*
 * code.hip -- K3: event coding of the token list (reference: x3.c:132-270,431-433 with dict.c:132-146,
 * context.c, tag_pair.c, ac.c:46-126, bio.c:49-112).
 *
 * One wavefront per stream walks the tokens K2 produced.  Every reference loop over a growable table becomes a
 * 64-lane sweep with ballot / butterfly reductions:
 *   - dict_update_costs + qsort (dict.c:132-146)  ==  move-to-front of the tag list; the coded `index` is the MTF
 *     rank;  rank search = ballot over 64-tag chunks, the shift = lane-parallel copy;
 *   - count_cum_freqs / index_of_symbol (ac.c:6-18,87-96) == butterfly prefix sums over the frequency chunks;
 *   - ctx_query_tag_item + the throw-away model of context.c:95-133 == one sweep of the context's item list that
 *     yields position, freq and cum_freq together;
 *   - tag_pair BST (tag_pair.c:67-130) == exact open-addressing hash map (the tree shape never reaches the stream);
 *   - ac_encode_scale (ac.c:46-75) runs redundantly in all lanes (wave-uniform), lane 0 stores finished words.
 * The mode decision reproduces the reference's float arithmetic: (float)freq / (float)total, one multiply, strict >
 * in the order IDX1, CTX0, CTX1 (x3.c:152-172).  Build with -ffp-contract=off and IEEE division.
 */
#include "x3_tables.h"

struct Coder {
	uint32_t lo, hi, pending; /* ac.h:8-15; values < 2^31 */
	uint32_t acc, cnt;        /* bio.h:15-20: bit buffer */
	uint32_t w, capw, full;
	uint32_t *out32;
};

__device__ static __forceinline__ void put_bit(Coder &c, uint32_t bit, uint32_t lane) /* bio.c:49-72 with n = 1 */
{
	c.acc |= bit << c.cnt;
	if (++c.cnt == 32) {
		if (c.w < c.capw) { if (lane == 0) c.out32[c.w] = c.acc; } else c.full = 1;
		c.w++;
		c.acc = 0;
		c.cnt = 0;
	}
}

__device__ static void ac_encode(Coder &c, uint32_t cum_lo, uint32_t cum_hi, uint32_t total, uint32_t lane) /* ac.c:77-85,46-75 */
{
	const uint32_t step = (c.hi - c.lo + 1) / total;
	c.hi = c.lo + step * cum_hi - 1;
	c.lo = c.lo + step * cum_lo;
	for (;;) {
		if (c.hi < 0x40000000u) {
			put_bit(c, 0, lane);
			c.lo = 2 * c.lo;
			c.hi = 2 * c.hi + 1;
			for (; c.pending > 0; c.pending--) put_bit(c, 1, lane);
		} else if (c.lo >= 0x40000000u) {
			put_bit(c, 1, lane);
			c.lo = 2 * (c.lo - 0x40000000u);
			c.hi = 2 * (c.hi - 0x40000000u) + 1;
			for (; c.pending > 0; c.pending--) put_bit(c, 0, lane);
		} else break;
	}
	while (c.lo >= 0x20000000u && c.hi < 0x60000000u) {
		c.pending++;
		c.lo = 2 * (c.lo - 0x20000000u);
		c.hi = 2 * (c.hi - 0x20000000u) + 1;
	}
}

__device__ static void x3_code_body(const X3CodeArgs &a)
{
	const X3Chunk ck = a.chunks[blockIdx.x];
	const X3ParseResult pr = a.parsed[blockIdx.x];
	const uint32_t lane = x3_lane();
	const uint8_t *b = a.bytes + ck.byte_off;
	const uint32_t *tpos = a.tok_pos + ck.elem_off, *tinf = a.tok_info + ck.elem_off;
	uint32_t *mtf = a.mtf + ck.tag_off, *idxfreq = a.idxfreq + ck.tag_off;
	X3CtxHdr *ctx1 = a.ctx1 + ck.tag_off, *ctx0 = a.ctx0 + ck.ctx0_off;
	uint64_t *pool = a.items + ck.item_off;
	uint64_t *pkey = a.pair_key + ck.pair_off;
	uint32_t *pval = a.pair_val + ck.pair_off;
	const uint32_t plog = ck.pair_log2, pmask = (1u << plog) - 1;

	Coder c;
	c.lo = 0; c.hi = 0x7FFFFFFFu; c.pending = 0; /* ac_init, ac.c:35-41 */
	c.acc = 0; c.cnt = 0; c.w = 0; c.capw = (uint32_t)(ck.out_cap / 4); c.full = 0;
	c.out32 = (uint32_t *)(a.out + ck.out_off);

	/* create(), x3.c:225-249 */
	uint32_t ev[5] = { 1024, 1024, 1, 1, 1 }, evtotal = 2051;
	uint32_t nev[5] = { 0, 0, 0, 0, 0 };
	uint32_t lf = 1, lftotal = 32;                     /* model_match_size: lane i < 32 holds freq of symbol i       */
	uint32_t cf0 = 1, cf1 = 1, cf2 = 1, cf3 = 1, cftotal = 256; /* model_chars: lane l holds symbols 4l..4l+3        */
	uint32_t D = 0, idxtotal = 0;                      /* model_index1 has D symbols                                  */
	uint32_t npairs = 0, status = X3_ST_OK;
	uint64_t pool_top = 0;
	uint32_t prev1 = 0, ctx1tag = 0;                   /* prev_context1, context1 (x3.c:376-377)                      */

	for (uint32_t k = 0; k < pr.ntok && status == X3_ST_OK; k++) {
		const uint32_t pos = tpos[k], info = tinf[k];
		if (!(info & X3_TOK_MISS)) {
			/* ================= dictionary hit: encode_tag, x3.c:132-223 ================= */
			const uint32_t tag = info;
			/* rank of the tag in recency order + model_index1 freq / cum_freq of that rank */
			uint32_t r = 0, rfreq = 0, rcum = 0;
			for (uint32_t base = 0; base < D; base += X3_WAVE) {
				const uint32_t i = base + lane;
				const uint32_t v = i < D ? mtf[i] : 0xFFFFFFFFu;
				const uint32_t fq = i < D ? idxfreq[i] : 0;
				const uint64_t mask = x3_ballot(v == tag);
				if (mask) {
					const uint32_t l = (uint32_t)x3_ctz64(mask);
					r = base + l;
					rfreq = x3_bcast_u32(fq, (int)l);
					rcum += wave_sum(lane < l ? fq : 0u);
					break;
				}
				rcum += wave_sum(fq);
			}
			/* contexts: ctx0 by pair ordinal (0 when the pair is unknown, x3.c:141-145), ctx1 by tag */
			uint32_t c0id = 0;
			{
				const uint64_t key = (((uint64_t)prev1 << 32) | ctx1tag) + 1;
				uint32_t s = pair_slot(key, plog);
				for (uint64_t kk = pkey[s]; kk != 0; s = (s + 1) & pmask, kk = pkey[s])
					if (kk == key) { c0id = pval[s]; break; }
			}
			X3CtxHdr *h0p = ctx0 + c0id, *h1p = ctx1 + ctx1tag;
			const X3CtxHdr h0 = *h0p, h1 = *h1p;
			const CtxQ q0 = ctx_query(h0, pool, tag, lane);
			const CtxQ q1 = ctx_query(h1, pool, tag, lane);

			/* x3.c:152-172 */
			const float fet = (float)evtotal;
			float p0 = 0.f, p1 = 0.f;
			if (q0.found) p0 = ((float)ev[X3_E_CTX0] / fet) * ((float)q0.freq / (float)h0.total);
			if (q1.found) p1 = ((float)ev[X3_E_CTX1] / fet) * ((float)q1.freq / (float)h1.total);
			const float pi = ((float)ev[X3_E_IDX1] / fet) * ((float)rfreq / (float)idxtotal);
			int mode = X3_E_IDX1;
			float best = pi;
			if (p0 > best) { mode = X3_E_CTX0; best = p0; }
			if (p1 > best) { mode = X3_E_CTX1; best = p1; }

			x3_wave_sync(); /* all lanes have read the headers / pair map before lane 0 starts updating them */

			/* x3.c:176-190 */
			if (mode == X3_E_CTX0) {
				ac_encode(c, 0, ev[0], evtotal, lane);
				ev[0]++; evtotal++; nev[0]++;
				ac_encode(c, q0.cum, q0.cum + q0.freq, h0.total, lane);
			} else if (mode == X3_E_CTX1) {
				ac_encode(c, ev[0], ev[0] + ev[1], evtotal, lane);
				ev[1]++; evtotal++; nev[1]++;
				ac_encode(c, q1.cum, q1.cum + q1.freq, h1.total, lane);
			} else {
				ac_encode(c, ev[0] + ev[1], ev[0] + ev[1] + ev[2], evtotal, lane);
				ev[2]++; evtotal++; nev[2]++;
				ac_encode(c, rcum, rcum + rfreq, idxtotal, lane);
				if (lane == 0) idxfreq[r] = rfreq + 1;
				idxtotal++;
			}

			/* x3.c:195-222 : both contexts learn the tag; (context1, tag) becomes a known pair */
			ctx_touch(h0p, h0, q0, tag, pool, pool_top, ck.item_cap, status, lane);
			ctx_touch(h1p, h1, q1, tag, pool, pool_top, ck.item_cap, status, lane);
			{
				const uint64_t key = (((uint64_t)ctx1tag << 32) | tag) + 1;
				uint32_t s = pair_slot(key, plog);
				uint64_t kk = pkey[s];
				while (kk != 0 && kk != key) { s = (s + 1) & pmask; kk = pkey[s]; }
				x3_wave_sync();
				if (kk == 0) {
					if (lane == 0) { pkey[s] = key; pval[s] = npairs; }
					npairs++;
				}
			}
			/* x3.c:389-397 */
			mtf_to_front(mtf, r, tag, lane);
			prev1 = ctx1tag;
			ctx1tag = tag;
			x3_wave_sync();
		} else {
			/* ================= new fragment: encode_match, x3.c:251-270 ================= */
			const uint32_t len = info & 0x3Fu;
			ac_encode(c, ev[0] + ev[1] + ev[2], ev[0] + ev[1] + ev[2] + ev[3], evtotal, lane);
			ev[3]++; evtotal++; nev[3]++;
			{
				const uint32_t sym = len - 1;
				const uint32_t cum = wave_sum(lane < sym ? lf : 0u); /* lanes >= 32 hold symbols that do not exist; sym < 32 */
				const uint32_t fq = x3_bcast_u32(lf, (int)sym);
				ac_encode(c, cum, cum + fq, lftotal, lane);
				if (lane == sym) lf++;
				lftotal++;
			}
			for (uint32_t j = 0; j < len; j++) {
				const uint32_t ch = b[(uint64_t)pos + j];
				const uint32_t owner = ch >> 2, sub = ch & 3;
				uint32_t part = 0;
				if (lane < owner) part = cf0 + cf1 + cf2 + cf3;
				else if (lane == owner) part = (sub > 0 ? cf0 : 0) + (sub > 1 ? cf1 : 0) + (sub > 2 ? cf2 : 0);
				const uint32_t cum = wave_sum(part);
				const uint32_t mine = sub == 0 ? cf0 : sub == 1 ? cf1 : sub == 2 ? cf2 : cf3;
				const uint32_t fq = x3_bcast_u32(mine, (int)owner);
				ac_encode(c, cum, cum + fq, cftotal, lane);
				if (lane == owner) {
					if (sub == 0) cf0++; else if (sub == 1) cf1++; else if (sub == 2) cf2++; else cf3++;
				}
				cftotal++;
			}
			if (!(info & X3_TOK_DUP)) { /* x3.c:412-420 : new element -> rank 0, model_index1 grows by one symbol */
				mtf_to_front(mtf, D, D, lane);
				if (lane == 0) idxfreq[D] = 1;
				D++;
				idxtotal++;
			}
			prev1 = 0; /* x3.c:424-425 */
			ctx1tag = 0;
			x3_wave_sync();
		}
		if (c.full) status = X3_ST_OUT_FULL;
	}

	/* x3.c:431-433, ac_encode_flush ac.c:115-126, bio_close bio.c:105-112 */
	ac_encode(c, evtotal - ev[4], evtotal, evtotal, lane);
	if (c.lo < 0x20000000u) {
		put_bit(c, 0, lane);
		for (uint32_t i = 0; i < c.pending + 1; i++) put_bit(c, 1, lane);
	} else {
		put_bit(c, 1, lane);
	}
	if (c.cnt > 0) {
		if (c.w < c.capw) { if (lane == 0) c.out32[c.w] = c.acc; } else c.full = 1;
		c.w++;
	}
	if (c.full && status == X3_ST_OK) status = X3_ST_OUT_FULL;

	if (lane == 0) {
		X3CodeResult r;
		r.out_len = c.w * 4; r.status = status; r.pairs = npairs; r._r = 0;
		for (int i = 0; i < 5; i++) r.events[i] = nev[i];
		r.events[5] = r.events[6] = r.events[7] = 0;
		a.result[blockIdx.x] = r;
	}
}

#ifndef X3_EMU
__global__ void __launch_bounds__(X3_CODE_THREADS) x3_code_kernel(X3CodeArgs a) { x3_code_body(a); }

extern "C" void x3k_launch_code(const X3CodeArgs *a, uint32_t nchunks, hipStream_t st)
{
	hipLaunchKernelGGL(x3_code_kernel, dim3(nchunks), dim3(X3_CODE_THREADS), 0, st, *a);
}
#else
static void code_tramp(void *p) { x3_code_body(*(const X3CodeArgs *)p); }
extern "C" void x3k_launch_code(const X3CodeArgs *a, uint32_t nchunks, void *)
{
	x3emu_launch(code_tramp, (void *)a, dim3(nchunks), dim3(X3_CODE_THREADS));
}
#endif
