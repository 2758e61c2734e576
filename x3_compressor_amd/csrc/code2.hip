/*
 * code2.hip -- K3 v2: event coding of the token list, restructured for the GPU (reference: x3.c:132-270,431-433 with
 * dict.c:132-146, context.c, tag_pair.c, ac.c, bio.c).
 *
 * After K2 the whole token sequence of a stream is known, and everything the reference recomputes step by step from
 * its growable tables is a COUNTING question about that sequence (SURVEY.md 7.1 tier T2):
 *   - MTF rank of a hit (the `index` coded by model_index1; dict.c:132-146 == move-to-front): number of distinct tags
 *     touched since the tag's previous touch  =  #{earlier touches j : prev(j) < prev(e)} - (prev(e)+1);
 *   - tag-pair ordinal (tag_pair.c:100-130): rank of the pair's first occurrence among first occurrences;
 *   - context item (context.c:20-56,95-133): freq = earlier hits in the same context with the same tag, total = earlier
 *     hits in the context, list position = rank of the tag's first occurrence, cum_freq = earlier hits in the context
 *     whose tag sits at a smaller list position.
 * All of them reduce to stable radix sorts, prefix sums and ONE custom primitive, "count smaller before"
 * (CSB: for every element, how many earlier elements of its bucket have a smaller key), done as an MSB-first stable
 * binary partition (wavelet-tree construction).  These run across the whole chip for all streams of a batch at once.
 * What remains serial per stream is thin:
 *   pass 1 (x3_modes_kernel): the mode choice of x3.c:152-172 -- it feeds back through model_events / model_index1;
 *   [parallel: cum_freq of the IDX1-coded ranks is again a CSB over the IDX1 subset]
 *   pass 2 (x3_ac_kernel)   : the arithmetic coder recurrence + bit output (ac.c:46-85, bio.c:49-72), one symbol triple
 *                             (cum, freq, total) per coded symbol.
 */
#include "x3_host.h"

#include <vector>

#define NONE32 0xFFFFFFFFu

#ifdef X3_EMU
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
#endif

static inline int bits_for(uint64_t maxval) { int b = 1; while (b < 32 && (maxval >> b)) b++; return b; }

/* largest c with off[c] <= idx (off has n+1 non-decreasing entries, off[n] > idx) */
__device__ static __forceinline__ uint32_t find_chunk(const uint32_t *off, uint32_t n, uint32_t idx)
{
	uint32_t lo = 0, hi = n; /* answer in [lo, hi) */
	while (hi - lo > 1) {
		uint32_t mid = (lo + hi) >> 1;
		if (off[mid] <= idx) lo = mid; else hi = mid;
	}
	return lo;
}

/* ============================================================================================================
 * CSB: cnt_out[i] = #{ j < i : bucket(j) == bucket(i) and key[j] < key[i] }, buckets = contiguous ranges [bs[i], be[i]).
 * key/bs/be are clobbered.  scratch: 9 arrays of n+1 u32.
 * ============================================================================================================ */
static int csb_run(X3Code2Bufs &B, hipStream_t st, size_t n, int bits, uint32_t *key, uint32_t *bs, uint32_t *be,
                   uint32_t *cnt_out, uint32_t *const *scratch)
{
	if (!n) return X3H_OK;
	uint32_t *key2 = scratch[0], *org = scratch[1], *org2 = scratch[2], *bs2 = scratch[3], *be2 = scratch[4];
	uint32_t *cnt = scratch[5], *cnt2 = scratch[6], *z = scratch[7], *Z = scratch[8];
	x3_foreach(n, st, X3_LAMBDA(size_t i) { org[i] = (uint32_t)i; cnt[i] = 0; });
	for (int b = bits - 1; b >= 0; b--) {
		{
			const uint32_t *k = key;
			x3_foreach(n, st, X3_LAMBDA(size_t i) { z[i] = ((k[i] >> b) & 1u) ^ 1u; });
		}
		CHK(x3p_excl_scan(B.tmp, z, Z, n, st));
		{
			const uint32_t *k = key, *o = org, *s_ = bs, *e_ = be, *c = cnt, *Zc = Z;
			x3_foreach(n, st, X3_LAMBDA(size_t i) {
				const uint32_t s = s_[i], e = e_[i];
				const uint32_t zb = Zc[i] - Zc[s], zc = Zc[e] - Zc[s];
				const uint32_t kv = k[i];
				uint32_t d, nbs, nbe, cv = c[i];
				if (!((kv >> b) & 1u)) { d = s + zb; nbs = s; nbe = s + zc; }
				else { d = s + zc + ((uint32_t)i - s - zb); nbs = s + zc; nbe = e; cv += zb; }
				key2[d] = kv; org2[d] = o[i]; bs2[d] = nbs; be2[d] = nbe; cnt2[d] = cv;
			});
		}
		uint32_t *t;
		t = key; key = key2; key2 = t;
		t = org; org = org2; org2 = t;
		t = bs; bs = bs2; bs2 = t;
		t = be; be = be2; be2 = t;
		t = cnt; cnt = cnt2; cnt2 = t;
	}
	{
		const uint32_t *o = org, *c = cnt;
		x3_foreach(n, st, X3_LAMBDA(size_t i) { cnt_out[o[i]] = c[i]; });
	}
	return X3H_OK;
}

/* ============================================================================================================
 * serial pass 1: mode choice (x3.c:152-172) + model_index1 / model_events feedback (x3.c:177,188)
 * ============================================================================================================ */
struct X3ModesArgs {
	const X3ParseResult *parsed;
	const uint32_t *ho, *dof;          /* per chunk: first hit, first tag */
	const uint32_t *f0, *t0, *f1, *t1; /* per hit: freq/total in ctx0 and ctx1 (freq 0 == tag absent) */
	const uint32_t *rank, *dk, *step;  /* per hit: MTF rank, dictionary size at that step, step index */
	uint32_t *idxfreq;                 /* per tag slot (by rank), pre-set to 1 */
	uint32_t *mode, *rfreq, *itot;     /* out per hit */
};

__device__ static void x3_modes_body(const X3ModesArgs &a)
{
	const uint32_t c = blockIdx.x, lane = x3_lane();
	const uint32_t H = a.parsed[c].hits, h0 = a.ho[c];
	uint32_t *idxf = a.idxfreq + a.dof[c];
	uint32_t ev0 = 1024, ev1 = 1024, ev2 = 1, nidx = 0;
	/* every lane walks the same sequence (wave-uniform); lane 0 owns the stores.  64 hits are fetched per round so the
	 * feature loads are coalesced and off the dependent chain. */
	for (uint32_t base = 0; base < H; base += X3_WAVE) {
		const uint32_t g = h0 + base + lane;
		const bool in = base + lane < H;
		const uint32_t vf0 = in ? a.f0[g] : 0, vt0 = in ? a.t0[g] : 1, vf1 = in ? a.f1[g] : 0, vt1 = in ? a.t1[g] : 1;
		const uint32_t vr = in ? a.rank[g] : 0, vd = in ? a.dk[g] : 1, vs = in ? a.step[g] : 0;
		/* the parts of the products that do not depend on the serial state: (float)freq / (float)total (context.c:114-133) */
		const float q0 = (float)vf0 / (float)vt0, q1 = (float)vf1 / (float)vt1;
		uint32_t mymode = 0, myrf = 0, myit = 0;
		const uint32_t cnt = H - base < X3_WAVE ? H - base : X3_WAVE;
		for (uint32_t l = 0; l < cnt; l++) {
			const uint32_t f0 = x3_bcast_u32(vf0, (int)l), f1 = x3_bcast_u32(vf1, (int)l);
			const float p0q = __uint_as_float(x3_bcast_u32(__float_as_uint(q0), (int)l));
			const float p1q = __uint_as_float(x3_bcast_u32(__float_as_uint(q1), (int)l));
			const uint32_t r = x3_bcast_u32(vr, (int)l), dk = x3_bcast_u32(vd, (int)l), sk = x3_bcast_u32(vs, (int)l);
			const uint32_t rf = idxf[r];
			const uint32_t itot = dk + nidx;
			const float fet = (float)(2051u + sk); /* model_events.total: 2051 + one per earlier step */
			float p0 = 0.f, p1 = 0.f;
			if (f0) p0 = ((float)ev0 / fet) * p0q;
			if (f1) p1 = ((float)ev1 / fet) * p1q;
			const float pi = ((float)ev2 / fet) * ((float)rf / (float)itot);
			uint32_t mode = X3_E_IDX1;
			float best = pi;
			if (p0 > best) { mode = X3_E_CTX0; best = p0; }
			if (p1 > best) { mode = X3_E_CTX1; best = p1; }
			if (mode == X3_E_CTX0) ev0++;
			else if (mode == X3_E_CTX1) ev1++;
			else {
				ev2++; nidx++;
				x3_wave_sync(); /* every lane has read idxf[r] */
				if (lane == 0) idxf[r] = rf + 1;
				x3_wave_sync(); /* the next hit may read this rank */
			}
			if (lane == l) { mymode = mode; myrf = rf; myit = itot; }
		}
		if (in) { a.mode[g] = mymode; a.rfreq[g] = myrf; a.itot[g] = myit; }
	}
}

/* ============================================================================================================
 * serial pass 2: arithmetic coder + bit output over the prepared symbols
 * ============================================================================================================ */
struct X3AcArgs {
	const uint8_t *bytes;
	const X3Chunk *chunks;
	const X3ParseResult *parsed;
	const uint32_t *tok_pos, *tok_info, *tok_hb;
	const uint32_t *ho;
	const uint32_t *scum, *sfreq, *stot, *mode; /* per hit: the tag/index symbol and the chosen mode */
	const uint32_t *npairs;                       /* per chunk (stats) */
	uint8_t *out;
	X3CodeResult *result;
};

struct Coder2 {
	uint32_t lo, hi, pending, acc, cnt, w, capw, full;
	uint32_t *out32;
};

__device__ static __forceinline__ void c2_put(Coder2 &c, uint32_t bit, uint32_t lane)
{
	c.acc |= bit << c.cnt;
	if (++c.cnt == 32) {
		if (c.w < c.capw) { if (lane == 0) c.out32[c.w] = c.acc; } else c.full = 1;
		c.w++; c.acc = 0; c.cnt = 0;
	}
}

__device__ static __forceinline__ void c2_encode(Coder2 &c, uint32_t cum_lo, uint32_t cum_hi, uint32_t total, uint32_t lane)
{
	const uint32_t step = (c.hi - c.lo + 1) / total; /* ac.c:77-85 */
	c.hi = c.lo + step * cum_hi - 1;
	c.lo = c.lo + step * cum_lo;
	for (;;) { /* ac.c:46-67 */
		if (c.hi < 0x40000000u) {
			c2_put(c, 0, lane);
			c.lo = 2 * c.lo; c.hi = 2 * c.hi + 1;
			for (; c.pending > 0; c.pending--) c2_put(c, 1, lane);
		} else if (c.lo >= 0x40000000u) {
			c2_put(c, 1, lane);
			c.lo = 2 * (c.lo - 0x40000000u); c.hi = 2 * (c.hi - 0x40000000u) + 1;
			for (; c.pending > 0; c.pending--) c2_put(c, 0, lane);
		} else break;
	}
	while (c.lo >= 0x20000000u && c.hi < 0x60000000u) { /* ac.c:69-74 */
		c.pending++;
		c.lo = 2 * (c.lo - 0x20000000u); c.hi = 2 * (c.hi - 0x20000000u) + 1;
	}
}

__device__ static __forceinline__ uint32_t wsum(uint32_t v)
{
	for (int m = 32; m >= 1; m >>= 1) v += x3_shfl_xor_u32(v, m);
	return v;
}

__device__ static void x3_ac_body(const X3AcArgs &a)
{
	const uint32_t c = blockIdx.x, lane = x3_lane();
	const X3Chunk ck = a.chunks[c];
	const X3ParseResult pr = a.parsed[c];
	const uint8_t *b = a.bytes + ck.byte_off;
	const uint32_t *tpos = a.tok_pos + ck.elem_off, *tinf = a.tok_info + ck.elem_off, *thb = a.tok_hb + ck.elem_off;
	const uint32_t h0 = a.ho[c];

	Coder2 cd;
	cd.lo = 0; cd.hi = 0x7FFFFFFFu; cd.pending = 0; cd.acc = 0; cd.cnt = 0; cd.w = 0;
	cd.capw = (uint32_t)(ck.out_cap / 4); cd.full = 0; cd.out32 = (uint32_t *)(a.out + ck.out_off);

	uint32_t ev0 = 1024, ev1 = 1024, ev2 = 1, ev3 = 1, evtotal = 2051; /* x3.c:236-244 */
	uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0;
	uint32_t lf = 1, lftotal = 32;
	uint32_t cf0 = 1, cf1 = 1, cf2 = 1, cf3 = 1, cftotal = 256;

	for (uint32_t base = 0; base < pr.ntok; base += X3_WAVE) {
		/* coalesced fetch of 64 steps and of the hit symbols they refer to */
		const bool in = base + lane < pr.ntok;
		const uint32_t vinfo = in ? tinf[base + lane] : 0, vpos = in ? tpos[base + lane] : 0;
		uint32_t vcum = 0, vfreq = 0, vtot = 1, vmode = 0;
		if (in && !(vinfo & X3_TOK_MISS)) {
			const uint32_t g = h0 + thb[base + lane];
			vcum = a.scum[g]; vfreq = a.sfreq[g]; vtot = a.stot[g]; vmode = a.mode[g];
		}
		const uint32_t cnt = pr.ntok - base < X3_WAVE ? pr.ntok - base : X3_WAVE;
		for (uint32_t l = 0; l < cnt; l++) {
			const uint32_t info = x3_bcast_u32(vinfo, (int)l);
			if (!(info & X3_TOK_MISS)) {
				const uint32_t mode = x3_bcast_u32(vmode, (int)l), cum = x3_bcast_u32(vcum, (int)l);
				const uint32_t fq = x3_bcast_u32(vfreq, (int)l), tot = x3_bcast_u32(vtot, (int)l);
				if (mode == X3_E_CTX0) { c2_encode(cd, 0, ev0, evtotal, lane); ev0++; n0++; }
				else if (mode == X3_E_CTX1) { c2_encode(cd, ev0, ev0 + ev1, evtotal, lane); ev1++; n1++; }
				else { c2_encode(cd, ev0 + ev1, ev0 + ev1 + ev2, evtotal, lane); ev2++; n2++; }
				evtotal++;
				c2_encode(cd, cum, cum + fq, tot, lane);
			} else {
				const uint32_t len = info & 0x3Fu, pos = x3_bcast_u32(vpos, (int)l);
				c2_encode(cd, ev0 + ev1 + ev2, ev0 + ev1 + ev2 + ev3, evtotal, lane); /* x3.c:253-255 */
				ev3++; n3++; evtotal++;
				{
					const uint32_t sym = len - 1;
					const uint32_t cum = wsum(lane < sym ? lf : 0u);
					const uint32_t fq = x3_bcast_u32(lf, (int)sym);
					c2_encode(cd, cum, cum + fq, lftotal, lane);
					if (lane == sym) lf++;
					lftotal++;
				}
				for (uint32_t j = 0; j < len; j++) {
					const uint32_t ch = b[(uint64_t)pos + j];
					const uint32_t owner = ch >> 2, sub = ch & 3;
					uint32_t part = 0;
					if (lane < owner) part = cf0 + cf1 + cf2 + cf3;
					else if (lane == owner) part = (sub > 0 ? cf0 : 0) + (sub > 1 ? cf1 : 0) + (sub > 2 ? cf2 : 0);
					const uint32_t cum = wsum(part);
					const uint32_t mine = sub == 0 ? cf0 : sub == 1 ? cf1 : sub == 2 ? cf2 : cf3;
					const uint32_t fq = x3_bcast_u32(mine, (int)owner);
					c2_encode(cd, cum, cum + fq, cftotal, lane);
					if (lane == owner) { if (sub == 0) cf0++; else if (sub == 1) cf1++; else if (sub == 2) cf2++; else cf3++; }
					cftotal++;
				}
			}
		}
		if (cd.full) break;
	}

	/* E_EOF (x3.c:432-433), ac_encode_flush (ac.c:115-126), bio_close (bio.c:105-112) */
	c2_encode(cd, evtotal - 1, evtotal, evtotal, lane);
	if (cd.lo < 0x20000000u) {
		c2_put(cd, 0, lane);
		for (uint32_t i = 0; i < cd.pending + 1; i++) c2_put(cd, 1, lane);
	} else c2_put(cd, 1, lane);
	if (cd.cnt > 0) {
		if (cd.w < cd.capw) { if (lane == 0) cd.out32[cd.w] = cd.acc; } else cd.full = 1;
		cd.w++;
	}
	if (lane == 0) {
		X3CodeResult r;
		r.out_len = cd.w * 4; r.status = cd.full ? X3_ST_OUT_FULL : X3_ST_OK; r.pairs = a.npairs[c]; r._r = 0;
		r.events[0] = n0; r.events[1] = n1; r.events[2] = n2; r.events[3] = n3;
		r.events[4] = r.events[5] = r.events[6] = r.events[7] = 0;
		a.result[c] = r;
	}
}

#ifndef X3_EMU
__global__ void __launch_bounds__(X3_WAVE) x3_modes_kernel(X3ModesArgs a) { x3_modes_body(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_ac_kernel(X3AcArgs a) { x3_ac_body(a); }
static void launch_modes(const X3ModesArgs &a, uint32_t nchunks, hipStream_t st) { hipLaunchKernelGGL(x3_modes_kernel, dim3(nchunks), dim3(X3_WAVE), 0, st, a); }
static void launch_ac(const X3AcArgs &a, uint32_t nchunks, hipStream_t st) { hipLaunchKernelGGL(x3_ac_kernel, dim3(nchunks), dim3(X3_WAVE), 0, st, a); }
#else
static void modes_tramp(void *p) { x3_modes_body(*(const X3ModesArgs *)p); }
static void ac_tramp(void *p) { x3_ac_body(*(const X3AcArgs *)p); }
static void launch_modes(const X3ModesArgs &a, uint32_t nchunks, hipStream_t) { x3emu_launch(modes_tramp, (void *)&a, dim3(nchunks), dim3(X3_WAVE)); }
static void launch_ac(const X3AcArgs &a, uint32_t nchunks, hipStream_t) { x3emu_launch(ac_tramp, (void *)&a, dim3(nchunks), dim3(X3_WAVE)); }
#endif

/* ============================================================================================================
 * context statistics of every hit for one context family (group id G per hit):
 *   freq  = earlier hits of the group with the same tag      (item freq; 0 == tag not in the context yet)
 *   total = earlier hits of the group                        (== sum of item freqs, calc_total_freq)
 *   cum   = earlier hits of the group whose tag was first seen in the group before this tag (== cum_freq of the item)
 * ============================================================================================================ */
static int ctx_stats(X3Code2Bufs &B, hipStream_t st, size_t nH, int gbits, int tbits, const uint32_t *G, const uint32_t *h_tag,
                     uint32_t *freq, uint32_t *total, uint32_t *cum, uint32_t *const *T /* 22 temp arrays */)
{
	uint32_t *iota = T[0], *kA = T[1], *vA = T[2], *gsf = T[3], *gsA = T[4], *gend = T[5], *kt = T[6], *kB = T[7], *vB = T[8];
	uint32_t *rsf = T[9], *rsB = T[10], *first = T[11], *ff = T[12], *PF = T[13], *posh = T[14], *key = T[15], *be = T[16];
	uint32_t *cntA = T[17], *bs = T[18];
	uint32_t *dmax = B.maxred.as<uint32_t>();

	x3_foreach(nH, st, X3_LAMBDA(size_t i) { iota[i] = (uint32_t)i; });
	CHK(x3p_sort_pairs(B.tmp, G, kA, iota, vA, nH, gbits, st)); /* arrangement A: by group, time order inside */
	x3_foreach(nH, st, X3_LAMBDA(size_t i) { gsf[i] = (i > 0 && kA[i - 1] != kA[i]) ? (uint32_t)i : 0u; });
	CHK(x3p_incl_max_scan(B.tmp, gsf, gsA, nH, st));
	x3_foreach(nH, st, X3_LAMBDA(size_t i) {
		total[vA[i]] = (uint32_t)i - gsA[i];
		if (i + 1 == nH || kA[i + 1] != kA[i]) gend[gsA[i]] = (uint32_t)i + 1;
		kt[i] = h_tag[vA[i]];
	});
	CHK(x3p_sort_pairs(B.tmp, kt, kB, vA, vB, nH, tbits, st)); /* arrangement B: by tag, then group, then time */
	x3_foreach(nH, st, X3_LAMBDA(size_t i) { rsf[i] = (i > 0 && (kB[i - 1] != kB[i] || G[vB[i - 1]] != G[vB[i]])) ? (uint32_t)i : 0u; });
	CHK(x3p_incl_max_scan(B.tmp, rsf, rsB, nH, st));
	x3_foreach(nH, st, X3_LAMBDA(size_t i) {
		freq[vB[i]] = (uint32_t)i - rsB[i];
		first[vB[i]] = vB[rsB[i]];
	});
	x3_foreach(nH, st, X3_LAMBDA(size_t i) { ff[i] = freq[vA[i]] == 0 ? 1u : 0u; });
	CHK(x3p_excl_scan(B.tmp, ff, PF, nH, st));
	x3_foreach(nH, st, X3_LAMBDA(size_t i) { posh[vA[i]] = PF[i] - PF[gsA[i]]; }); /* list position, valid for first occurrences */
	HIPCHK(hipMemsetAsync(dmax, 0, 4, st));
	x3_foreach(nH, st, X3_LAMBDA(size_t i) {
		const uint32_t p = posh[first[vA[i]]];
		key[i] = p;
		bs[i] = gsA[i];
		be[i] = gend[gsA[i]];
		if (p) atomicMax(dmax, p);
	});
	uint32_t hmax = 0;
	HIPCHK(hipMemcpyAsync(&hmax, dmax, 4, hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	CHK(csb_run(B, st, nH, bits_for(hmax), key, bs, be, cntA, T + 19 - 0 /* scratch follows */));
	x3_foreach(nH, st, X3_LAMBDA(size_t i) { cum[vA[i]] = cntA[i]; });
	return X3H_OK;
}

/* ============================================================================================================ */
int x3_code_v2_run(X3Code2Bufs &B, hipStream_t st, int nchunks, const X3Chunk *h_chunks, const X3Chunk *d_chunks,
                   const X3ParseResult *h_parsed, const X3ParseResult *d_parsed,
                   const uint8_t *d_bytes, const uint32_t *tok_pos, const uint32_t *tok_info, const uint32_t *tok_hb,
                   const uint32_t *tok_nb, uint8_t *d_out, X3CodeResult *d_result)
{
	const uint32_t nc = (uint32_t)nchunks;
	/* ---- index spaces: steps, hits, MTF events (hits + inserted elements), tags ---- */
	std::vector<uint32_t> so(nc + 1), ho(nc + 1), eo(nc + 1), dof(nc + 1);
	uint64_t s = 0, h = 0, e = 0, d = 0;
	for (uint32_t c = 0; c < nc; c++) {
		so[c] = (uint32_t)s; ho[c] = (uint32_t)h; eo[c] = (uint32_t)e; dof[c] = (uint32_t)d;
		s += h_parsed[c].ntok; h += h_parsed[c].hits; e += (uint64_t)h_parsed[c].hits + h_parsed[c].dict_elems; d += h_parsed[c].dict_elems;
	}
	if (s >= (1ull << 31) || e >= (1ull << 31)) return X3H_E_ARG; /* one batch: < 2^31 steps */
	so[nc] = (uint32_t)s; ho[nc] = (uint32_t)h; eo[nc] = (uint32_t)e; dof[nc] = (uint32_t)d;
	const size_t nS = s, nH = h, nE = e, nD = d;
	const size_t nA = (nE > nH ? nE : nH) + 4;

	CHK(B.offs.reserve((size_t)(nc + 1) * 4 * 4));
	uint32_t *d_so = B.offs.as<uint32_t>(), *d_ho = d_so + (nc + 1), *d_eo = d_ho + (nc + 1), *d_dof = d_eo + (nc + 1);
	HIPCHK(hipMemcpyAsync(d_so, so.data(), (nc + 1) * 4, hipMemcpyHostToDevice, st));
	HIPCHK(hipMemcpyAsync(d_ho, ho.data(), (nc + 1) * 4, hipMemcpyHostToDevice, st));
	HIPCHK(hipMemcpyAsync(d_eo, eo.data(), (nc + 1) * 4, hipMemcpyHostToDevice, st));
	HIPCHK(hipMemcpyAsync(d_dof, dof.data(), (nc + 1) * 4, hipMemcpyHostToDevice, st));
	CHK(B.chunkmeta.reserve((size_t)nc * 4 * 4 + 64));
	uint32_t *m_pairbase = B.chunkmeta.as<uint32_t>(), *m_npairs = m_pairbase + nc, *m_first00 = m_npairs + nc, *m_ord00 = m_first00 + nc;
	CHK(B.maxred.reserve(64));
	CHK(B.idxfreq.reserve((nD + 4) * 4));
	const int NARR = 48;
	static_assert(sizeof(B.a) / sizeof(B.a[0]) >= 48, "work arrays");
	uint32_t *A[NARR];
	for (int i = 0; i < NARR; i++) { CHK(B.a[i].reserve(nA * 4)); A[i] = B.a[i].as<uint32_t>(); }

	/* persistent per-hit arrays */
	uint32_t *h_tag = A[0], *h_c1 = A[1], *h_pv = A[2], *h_dk = A[3], *h_step = A[4], *h_rank = A[5];
	uint32_t *e_tag = A[6], *e_hit = A[7], *h_pair = A[8], *G0 = A[9];
	uint32_t *f0 = A[10], *t0 = A[11], *c0 = A[12], *f1 = A[13], *t1 = A[14], *c1 = A[15];
	uint32_t *mode = A[16], *rfreq = A[17], *itot = A[18], *rcum = A[19];
	uint32_t **T = A + 20; /* 28 temporaries */

	if (nH > 0) {
		/* ---- F1: per step -> per hit / per event records ---- */
		x3_foreach(nS, st, X3_LAMBDA(size_t gs) {
			const uint32_t c = find_chunk(d_so, nc, (uint32_t)gs);
			const uint32_t k = (uint32_t)gs - d_so[c];
			const uint64_t base = d_chunks[c].elem_off;
			const uint32_t info = tok_info[base + k], hb = tok_hb[base + k], nb = tok_nb[base + k];
			if (!(info & X3_TOK_MISS)) {
				const uint32_t gh = d_ho[c] + hb, ev = d_eo[c] + hb + nb;
				const bool pv = k > 0 && !(tok_info[base + k - 1] & X3_TOK_MISS);
				h_tag[gh] = d_dof[c] + info;
				h_c1[gh] = d_dof[c] + (pv ? tok_info[base + k - 1] : 0u); /* context1 (x3.c:390,425) */
				h_pv[gh] = pv ? 1u : 0u;
				h_dk[gh] = nb;
				h_step[gh] = k;
				e_tag[ev] = d_dof[c] + info;
				e_hit[ev] = gh;
			} else if (!(info & X3_TOK_DUP)) {
				const uint32_t ev = d_eo[c] + hb + nb;
				e_tag[ev] = d_dof[c] + nb; /* the new element's tag (dict.c:100) */
				e_hit[ev] = NONE32;
			}
		});

		/* ---- MTF rank: touches sorted by tag give prev(e); rank = CSB(prev+1) - (prev_local+1) ---- */
		{
			uint32_t *iota = T[0], *ks = T[1], *vs = T[2], *key = T[3], *keyc = T[4], *bs = T[5], *be = T[6], *cnt = T[7];
			x3_foreach(nE, st, X3_LAMBDA(size_t i) { iota[i] = (uint32_t)i; });
			CHK(x3p_sort_pairs(B.tmp, e_tag, ks, iota, vs, nE, bits_for(nD), st));
			x3_foreach(nE, st, X3_LAMBDA(size_t i) {
				const uint32_t ev = vs[i];
				const uint32_t kv = (i > 0 && ks[i - 1] == ks[i]) ? vs[i - 1] + 1 : 0u;
				key[ev] = kv;
			});
			x3_foreach(nE, st, X3_LAMBDA(size_t i) {
				const uint32_t c = find_chunk(d_eo, nc, (uint32_t)i);
				keyc[i] = key[i]; bs[i] = d_eo[c]; be[i] = d_eo[c + 1];
			});
			CHK(csb_run(B, st, nE, bits_for(nE + 1), keyc, bs, be, cnt, T + 8));
			x3_foreach(nE, st, X3_LAMBDA(size_t i) {
				const uint32_t gh = e_hit[i];
				if (gh != NONE32) {
					const uint32_t c = find_chunk(d_eo, nc, (uint32_t)i);
					h_rank[gh] = cnt[i] - (key[i] - d_eo[c]);
				}
			});
		}

		/* ---- tag-pair ordinals (tag_pair.c) and the ctx0 group of every hit (x3.c:139-147) ---- */
		{
			uint32_t *iota = T[0], *k1 = T[1], *v1 = T[2], *k2in = T[3], *k2 = T[4], *v2 = T[5], *rsf = T[6], *rs = T[7], *pf = T[8], *P = T[9], *rsflag = T[10];
			const int tb = bits_for(nD);
			x3_foreach(nH, st, X3_LAMBDA(size_t i) { iota[i] = (uint32_t)i; });
			CHK(x3p_sort_pairs(B.tmp, h_tag, k1, iota, v1, nH, tb, st));
			x3_foreach(nH, st, X3_LAMBDA(size_t i) { k2in[i] = h_c1[v1[i]]; });
			CHK(x3p_sort_pairs(B.tmp, k2in, k2, v1, v2, nH, tb, st)); /* sorted by (context1, tag), time order inside a pair */
			x3_foreach(nH, st, X3_LAMBDA(size_t i) {
				const bool start = i == 0 || k2[i - 1] != k2[i] || h_tag[v2[i - 1]] != h_tag[v2[i]];
				rsflag[i] = start ? 1u : 0u;
				rsf[i] = start ? (uint32_t)i : 0u;
			});
			CHK(x3p_incl_max_scan(B.tmp, rsf, rs, nH, st));
			x3_foreach(nH, st, X3_LAMBDA(size_t i) { pf[v2[i]] = rsflag[i]; }); /* first occurrence of its pair, in time order */
			CHK(x3p_excl_scan(B.tmp, pf, P, nH, st));
			x3_foreach(nH, st, X3_LAMBDA(size_t i) { h_pair[v2[i]] = P[v2[rs[i]]]; });
			x3_foreach(nc, st, X3_LAMBDA(size_t c) {
				const uint32_t lo = d_ho[c], hi = d_ho[c + 1];
				m_pairbase[c] = P[lo];
				m_npairs[c] = P[hi] - P[lo];
				/* the pair (0,0) of this chunk (contexts after a new fragment, x3.c:424-425): lower_bound in the sorted pairs */
				const uint32_t z = d_dof[c];
				uint32_t a = 0, b = (uint32_t)nH;
				while (a < b) {
					const uint32_t mid = (a + b) >> 1;
					const uint32_t kc = k2[mid], kt = h_tag[v2[mid]];
					if (kc < z || (kc == z && kt < z)) a = mid + 1; else b = mid;
				}
				if (hi > lo && a < nH && k2[a] == z && h_tag[v2[a]] == z) { m_first00[c] = v2[a]; m_ord00[c] = P[v2[a]]; }
				else { m_first00[c] = NONE32; m_ord00[c] = 0; }
			});
			x3_foreach(nH, st, X3_LAMBDA(size_t gh) {
				uint32_t g;
				if (h_pv[gh]) g = h_pair[gh - 1]; /* (prev_context1, context1) is the pair the previous hit registered */
				else {
					const uint32_t c = find_chunk(d_ho, nc, (uint32_t)gh);
					g = (m_first00[c] != NONE32 && m_first00[c] < gh) ? m_ord00[c] : m_pairbase[c]; /* unknown pair -> context 0 */
				}
				G0[gh] = g;
			});
		}

		/* ---- context statistics ---- */
		CHK(ctx_stats(B, st, nH, bits_for(nH), bits_for(nD), G0, h_tag, f0, t0, c0, T));
		CHK(ctx_stats(B, st, nH, bits_for(nD), bits_for(nD), h_c1, h_tag, f1, t1, c1, T));
	}

	/* ---- serial pass 1: modes ---- */
	uint32_t *idxf = B.idxfreq.as<uint32_t>();
	x3_foreach(nD + 1, st, X3_LAMBDA(size_t i) { idxf[i] = 1; });
	if (nH > 0) {
		X3ModesArgs ma;
		ma.parsed = d_parsed; ma.ho = d_ho; ma.dof = d_dof;
		ma.f0 = f0; ma.t0 = t0; ma.f1 = f1; ma.t1 = t1; ma.rank = h_rank; ma.dk = h_dk; ma.step = h_step;
		ma.idxfreq = idxf; ma.mode = mode; ma.rfreq = rfreq; ma.itot = itot;
		launch_modes(ma, nc, st);
		HIPCHK(hipGetLastError());

		/* ---- cum_freq of the IDX1-coded ranks: rank + #{earlier IDX1 hits of the stream with a smaller rank} ---- */
		uint32_t *zi = T[0], *ci = T[1], *key = T[2], *org = T[3], *bs = T[4], *be = T[5], *cnt = T[6];
		x3_foreach(nH, st, X3_LAMBDA(size_t i) { zi[i] = mode[i] == X3_E_IDX1 ? 1u : 0u; });
		CHK(x3p_excl_scan(B.tmp, zi, ci, nH, st));
		uint32_t nI = 0;
		HIPCHK(hipMemcpyAsync(&nI, ci + nH, 4, hipMemcpyDeviceToHost, st));
		HIPCHK(hipStreamSynchronize(st));
		x3_foreach(nH, st, X3_LAMBDA(size_t i) {
			if (mode[i] == X3_E_IDX1) {
				const uint32_t j = ci[i];
				const uint32_t c = find_chunk(d_ho, nc, (uint32_t)i);
				key[j] = h_rank[i]; org[j] = (uint32_t)i; bs[j] = ci[d_ho[c]]; be[j] = ci[d_ho[c + 1]];
			}
		});
		uint64_t maxD = 1;
		for (uint32_t c = 0; c < nc; c++) if (h_parsed[c].dict_elems > maxD) maxD = h_parsed[c].dict_elems;
		CHK(csb_run(B, st, nI, bits_for(maxD), key, bs, be, cnt, T + 8));
		x3_foreach(nI, st, X3_LAMBDA(size_t j) { rcum[org[j]] = h_rank[org[j]] + cnt[j]; });

		/* ---- the tag / index symbol of every hit ---- */
		uint32_t *scum = T[0], *sfreq = T[1], *stot = T[2]; /* zi/ci/key are dead now */
		x3_foreach(nH, st, X3_LAMBDA(size_t i) {
			const uint32_t m = mode[i];
			uint32_t cu, fq, to;
			if (m == X3_E_CTX0) { cu = c0[i]; fq = f0[i]; to = t0[i]; }
			else if (m == X3_E_CTX1) { cu = c1[i]; fq = f1[i]; to = t1[i]; }
			else { cu = rcum[i]; fq = rfreq[i]; to = itot[i]; }
			scum[i] = cu; sfreq[i] = fq; stot[i] = to;
		});
	} else {
		x3_foreach(nc, st, X3_LAMBDA(size_t c) { m_npairs[c] = 0; });
	}

	/* ---- serial pass 2: arithmetic coder ---- */
	X3AcArgs aa;
	aa.bytes = d_bytes; aa.chunks = d_chunks; aa.parsed = d_parsed;
	aa.tok_pos = tok_pos; aa.tok_info = tok_info; aa.tok_hb = tok_hb; aa.ho = d_ho;
	aa.scum = T[0]; aa.sfreq = T[1]; aa.stot = T[2]; aa.mode = mode; aa.npairs = m_npairs;
	aa.out = d_out; aa.result = d_result;
	launch_ac(aa, nc, st);
	HIPCHK(hipGetLastError());
	(void)h_chunks; (void)tok_nb;
	return X3H_OK;
}
