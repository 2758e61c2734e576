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
 * radix-4 partition (wavelet-tree construction).  These run across the whole chip for all streams of a batch at once.
 * What remains per stream is thin:
 *   the mode choice of x3.c:152-172, which feeds back through model_events / model_index1: either the fixed-point iteration
 *      modes_fixed_point (chip-wide, for a few long streams) or x3_modes_kernel (one wavefront per stream, for many);
 *   [parallel: cum_freq of the IDX1-coded ranks is again a CSB over the IDX1 subset; symbol operands; no-op symbols dropped]
 *   x3_ac2_kernel: the arithmetic-coder interval recurrence (ac.c:46-85) on the scalar unit, one {cum, freq, magic, shift} operand
 *      per symbol in, one chain state per 8 symbols out;
 *   [parallel: per-symbol intervals re-derived from the states, bit emission (bio.c:49-72) by prefix sums + an OR-writer].
 * x3_code_v2_run also serves the pipelined schedule of api.hip: called on growing prefixes (X3CodeSeg), it queues the recurrence of
 * the new symbols on a separate HIP stream.
 */
#include "k3_sym.h"

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
 *
 * MSB-first stable partition, TWO key bits per level (wavelet-tree construction, radix 4).  An element of digit d has every
 * earlier element of its bucket with a smaller digit as a "smaller before".  The per-digit prefix counts are not full-size scans:
 * a counting kernel writes, per element, its exclusive digit counts INSIDE its block of 1024 positions (3 x 10 bits in one word)
 * plus three totals per block; the block totals are scanned (a tiny array) and  Z_d(x) = blockbase_d[x >> 10] + field_d(packed[x]).
 * Per two key bits an element costs one 4-byte read + one 4-byte write (count) and 24 B in / 20 B out (partition), against
 * ~128 B with one scanned flag array per bit: the levels run at HBM speed, so fewer bytes is the only way to make them faster.
 * ============================================================================================================ */
#define X3_CSB_BLK 1024u
struct X3CsbCountArgs { const uint32_t *key; uint32_t *packed, *tot; uint32_t n, shift, dmask, nblk; };

__device__ static void x3_csb_count_body(const X3CsbCountArgs &a)
{
	X3_LDS uint32_t wtot[X3_CSB_BLK / X3_WAVE][3];
	const uint32_t tid = threadIdx.x, lane = x3_lane(), wave = tid / X3_WAVE;
	const uint32_t i = blockIdx.x * X3_CSB_BLK + tid;
	const uint32_t d = i < a.n ? (a.key[i] >> a.shift) & a.dmask : 3u; /* positions >= n count for no digit */
	const uint64_t below = ((uint64_t)1 << lane) - 1;
	uint32_t e[3];
#pragma unroll
	for (uint32_t dd = 0; dd < 3; dd++) {
		const uint64_t m = x3_ballot(d == dd);
		e[dd] = (uint32_t)x3_popc64(m & below);
		if (lane == 0) wtot[wave][dd] = (uint32_t)x3_popc64(m);
	}
	__syncthreads();
#pragma unroll
	for (uint32_t dd = 0; dd < 3; dd++) {
		uint32_t acc = 0;
		for (uint32_t w = 0; w < wave; w++) acc += wtot[w][dd];
		e[dd] += acc;
	}
	if (i <= a.n) a.packed[i] = e[0] | e[1] << 10 | e[2] << 20; /* exclusive counts inside the block: <= 1023 each */
	if (tid == X3_CSB_BLK - 1) {
#pragma unroll
		for (uint32_t dd = 0; dd < 3; dd++) a.tot[dd * a.nblk + blockIdx.x] = e[dd] + (d == dd ? 1u : 0u);
	}
}

#ifndef X3_EMU
__global__ void __launch_bounds__(X3_CSB_BLK) x3_csb_count_kernel(X3CsbCountArgs a) { x3_csb_count_body(a); }
static void launch_csb_count(const X3CsbCountArgs &a, hipStream_t st) { hipLaunchKernelGGL(x3_csb_count_kernel, dim3(a.nblk), dim3(X3_CSB_BLK), 0, st, a); }
#else
/* test build: the same counts by a plain loop (1024-fiber workgroups make the CPU suite crawl; X3_EMU_CSB_KERNEL=1 runs the kernel body on the emulator) */
static void csb_count_tramp(void *p) { x3_csb_count_body(*(const X3CsbCountArgs *)p); }
static void launch_csb_count(const X3CsbCountArgs &a, hipStream_t)
{
	if (getenv("X3_EMU_CSB_KERNEL")) { x3emu_launch(csb_count_tramp, (void *)&a, dim3(a.nblk), dim3(X3_CSB_BLK)); return; }
	for (uint32_t b = 0; b < a.nblk; b++) {
		uint32_t e[3] = { 0, 0, 0 };
		for (uint32_t t = 0; t < X3_CSB_BLK; t++) {
			const uint32_t i = b * X3_CSB_BLK + t;
			if (i <= a.n) a.packed[i] = e[0] | e[1] << 10 | e[2] << 20;
			const uint32_t d = i < a.n ? (a.key[i] >> a.shift) & a.dmask : 3u;
			if (d < 3) e[d]++;
		}
		for (uint32_t dd = 0; dd < 3; dd++) a.tot[dd * a.nblk + b] = e[dd];
	}
}
#endif

static int csb_run(X3Code2Bufs &B, hipStream_t st, size_t n, int bits, uint32_t *key, uint32_t *bs, uint32_t *be,
                   uint32_t *cnt_out, uint32_t *const *scratch)
{
	if (!n) return X3H_OK;
	uint32_t *key2 = scratch[0], *org = scratch[1], *org2 = scratch[2], *bs2 = scratch[3], *be2 = scratch[4];
	uint32_t *cnt = scratch[5], *cnt2 = scratch[6], *packed = scratch[7];
	if (bits < 1) bits = 1;
	const uint32_t nblk = (uint32_t)((n + 1 + X3_CSB_BLK - 1) / X3_CSB_BLK);
	CHK(B.csbsmall.reserve(((size_t)3 * nblk + 8) * 2 * 4));
	uint32_t *tot = B.csbsmall.as<uint32_t>(), *S = tot + (size_t)3 * nblk + 4;
	x3_foreach(n, st, X3_LAMBDA(size_t i) { org[i] = (uint32_t)i; cnt[i] = 0; });
	for (int hi = bits; hi > 0;) {
		const int w = (hi & 1) ? 1 : 2; /* an odd bit count starts with a one-bit level */
		const uint32_t shift = (uint32_t)(hi - w), dmask = (1u << w) - 1;
		hi -= w;
		X3CsbCountArgs ca;
		ca.key = key; ca.packed = packed; ca.tot = tot; ca.n = (uint32_t)n; ca.shift = shift; ca.dmask = dmask; ca.nblk = nblk;
		launch_csb_count(ca, st);
		HIPCHK(hipGetLastError());
		CHK(x3p_excl_scan(B.tmp, tot, S, (size_t)3 * nblk, st)); /* one scan over [digit 0 blocks | digit 1 blocks | digit 2 blocks] */
		{
			const uint32_t *k = key, *o = org, *s_ = bs, *e_ = be, *c = cnt, *pk = packed, *Sc = S;
			x3_foreach(n, st, X3_LAMBDA(size_t i) {
				const uint32_t s = s_[i], e = e_[i], kv = k[i], d = (kv >> shift) & dmask;
				const uint32_t pi = pk[i], ps = pk[s], pe = pk[e];
				const uint32_t bi = (uint32_t)i / X3_CSB_BLK, bsb = s / X3_CSB_BLK, beb = e / X3_CSB_BLK;
				uint32_t zb[3], zc[3];
#pragma unroll
				for (uint32_t dd = 0; dd < 3; dd++) {
					const uint32_t zs = Sc[dd * nblk + bsb] + ((ps >> (10 * dd)) & 1023u);
					zb[dd] = Sc[dd * nblk + bi] + ((pi >> (10 * dd)) & 1023u) - zs;  /* digit dd before i in the bucket */
					zc[dd] = Sc[dd * nblk + beb] + ((pe >> (10 * dd)) & 1023u) - zs; /* digit dd in the bucket */
				}
				uint32_t dst, nbs, nbe, cv = c[i];
				if (d == 0) { nbs = s; nbe = s + zc[0]; dst = s + zb[0]; }
				else if (d == 1) { nbs = s + zc[0]; nbe = nbs + zc[1]; dst = nbs + zb[1]; cv += zb[0]; }
				else if (d == 2) { nbs = s + zc[0] + zc[1]; nbe = nbs + zc[2]; dst = nbs + zb[2]; cv += zb[0] + zb[1]; }
				else { nbs = s + zc[0] + zc[1] + zc[2]; nbe = e; dst = nbs + ((uint32_t)i - s - zb[0] - zb[1] - zb[2]); cv += zb[0] + zb[1] + zb[2]; }
				key2[dst] = kv; org2[dst] = o[i]; bs2[dst] = nbs; be2[dst] = nbe; cnt2[dst] = cv;
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
/* lanes (among `valid`) that hold the same `v` as this lane / #{ j in cand : v_j < x }: one ballot per value bit (see code3.hip) */
__device__ static __forceinline__ uint64_t wave_same_mask_u32(uint32_t v, int bits, uint64_t valid, bool me_valid)
{
	uint64_t m = valid;
	for (int b = 0; b < bits; b++) {
		const uint64_t Bm = x3_ballot(me_valid && ((v >> b) & 1u));
		m &= ((v >> b) & 1u) ? Bm : ~Bm;
	}
	return me_valid ? m : 0;
}
__device__ static __forceinline__ uint32_t wave_count_less_u32(uint32_t v, uint32_t x, int bits, uint64_t cand)
{
	uint32_t cnt = 0;
	uint64_t A = cand;
	for (int b = bits - 1; b >= 0; b--) {
		const uint64_t Bm = x3_ballot((v >> b) & 1u);
		if ((x >> b) & 1u) { cnt += (uint32_t)x3_popc64(A & ~Bm); A &= Bm; }
		else A &= ~Bm;
	}
	return cnt;
}



/* The mode choice of x3.c:152-172 feeds back through model_events (three counters) and model_index1 (one frequency per
 * rank), so in the reference it is a strictly serial chain.  A lone wavefront executes such a chain at ~5 cycles per
 * instruction; instead the wave decides 64 hits AT ONCE, one per lane, and falls back to serial evaluation only where
 * that is provably necessary:
 *   - at hit j of a block every counter lies in a known interval: ev_x in [E_x, E_x + j], model_index1.total in
 *     [T, T + j], freq(rank) in [f, f + (same-rank hits of the block)];
 *   - correctly rounded division and multiplication of non-negative operands are monotone, so evaluating the reference's
 *     exact float expression at the interval ends brackets the true p_ctx0 / p_ctx1 / p_idx1 EXACTLY (no error terms);
 *   - if the brackets already order the three probabilities (with the reference's tie rules), the decision is certain;
 *   - the remaining lanes are resolved in order with exact counters (ballot/popcount over the modes decided so far),
 *     using one vector division for the four quotients.
 * Typically only a few percent of the hits need the serial path.  ALL_LDS: every rank fits the LDS table. */
template <bool ALL_LDS, uint32_t NIDX, bool EMIT>
__device__ static __forceinline__ void x3_modes_loop(const X3ModesArgs &a, uint32_t *sidx, uint32_t *spre, uint32_t *scr, uint32_t *idxf, uint32_t H, uint32_t h0, uint32_t lane, uint32_t Dc,
                                                     uint32_t first = 0, const uint32_t *st0 = nullptr, uint32_t *st1 = nullptr, uint32_t nnoop0 = 0, bool sliced = false)
{
	uint32_t E0 = 1024, E1 = 1024, E2 = 1, nidx = 0; /* model_events freqs (x3.c:239-241), IDX1 uses so far */
	uint32_t nnoop = nnoop0;                         /* hits so far whose symbol is coded under a model total of 1 */
	if (st0) { E0 = st0[0]; E1 = st0[1]; E2 = st0[2]; nidx = st0[3]; } /* resumed behind `first` hits (or: behind the earlier slices) */
	const uint32_t nidx_start = sliced ? nidx : 0u;  /* a slice's list of IDX1-coded hits starts at its own first entry */
	const uint64_t below = ((uint64_t)1 << lane) - 1;
	/* features of the next block are fetched while this block is decided (a lone wave cannot hide the load latency otherwise) */
	/* features of the blocks ahead are fetched while this block is decided (a lone wave cannot hide the load latency otherwise): TWO blocks ahead -- a block is
	 * decided in ~1 100 cycles, a load comes back in 2 000-5 000 -- and by UNCONDITIONAL loads from clamped indices: around a load that sits behind a branch the
	 * compiler can only wait with vmcnt(0) (it must assume the branch was not taken: no younger load to count on), which waits for the loads just issued */
	uint32_t nf0, nt0, nf1, nt1, nr, nd, ns, mf0, mt0, mf1, mt1, mr, md, ms_;
	const uint32_t hlast = H ? H - 1 : 0;
	{
		const uint32_t i0 = first + lane < H ? first + lane : hlast, i1 = first + X3_WAVE + lane < H ? first + X3_WAVE + lane : hlast;
		const uint32_t g0 = h0 + i0, g1 = h0 + i1;
		nf0 = a.f0[(size_t)g0 * a.fs]; nt0 = a.t0[(size_t)g0 * a.fs]; nf1 = a.f1[(size_t)g0 * a.fs]; nt1 = a.t1[(size_t)g0 * a.fs]; nr = a.rank[g0]; nd = a.dk[g0]; ns = a.step[g0];
		mf0 = a.f0[(size_t)g1 * a.fs]; mt0 = a.t0[(size_t)g1 * a.fs]; mf1 = a.f1[(size_t)g1 * a.fs]; mt1 = a.t1[(size_t)g1 * a.fs]; mr = a.rank[g1]; md = a.dk[g1]; ms_ = a.step[g1];
	}
	for (uint32_t base = first; base < H; base += X3_WAVE) {
		const uint32_t g = h0 + base + lane;
		const bool in = base + lane < H;
		const uint32_t vf0 = in ? nf0 : 0u, vt0 = in ? nt0 : 1u, vf1 = in ? nf1 : 0u, vt1 = in ? nt1 : 1u, vr = in ? nr : 0u, vd = in ? nd : 1u, vs = in ? ns : 0u;
		nf0 = mf0; nt0 = mt0; nf1 = mf1; nt1 = mt1; nr = mr; nd = md; ns = ms_;
		{
			const uint32_t i2 = base + 2 * X3_WAVE + lane < H ? base + 2 * X3_WAVE + lane : hlast;
			const uint32_t gn = h0 + i2;
			mf0 = a.f0[(size_t)gn * a.fs]; mt0 = a.t0[(size_t)gn * a.fs]; mf1 = a.f1[(size_t)gn * a.fs]; mt1 = a.t1[(size_t)gn * a.fs]; mr = a.rank[gn]; md = a.dk[gn]; ms_ = a.step[gn];
		}
		const float q0 = vf0 ? (float)vf0 / (float)vt0 : 0.f; /* (float)freq / (float)total of the context item; 0 == absent */
		const float q1 = vf1 ? (float)vf1 / (float)vt1 : 0.f;
		const float ftot = (float)(2051u + vs); /* model_events.total: 2051 + one per earlier step */
		const bool lds_r = ALL_LDS || vr < NIDX;
		const uint32_t rfmin = in ? (lds_r ? sidx[lds_r ? vr : 0] : idxf[vr]) : 1;
		/* same-rank hits in this block (an upper bound is enough: hashed counters, collisions only widen the bracket) */
		const uint32_t hs = vr & 255u;
		/* (the work-group is ONE wavefront and scr / sidx are LDS: a compiler barrier orders them -- the fence of x3_wave_sync would wait for the feature loads that
		 * were just issued for the blocks ahead) */
		scr[lane] = 0; scr[lane + 64] = 0; scr[lane + 128] = 0; scr[lane + 192] = 0;
		x3_wave_order();
		if (in) atomicAdd(&scr[hs], 1u);
		x3_wave_order();
		const uint32_t srmax = in ? scr[hs] - 1 : 0;
		const uint32_t j = lane; /* at most j earlier hits of the block */
		const float a0lo = (float)E0 / ftot, a0hi = (float)(E0 + j) / ftot;
		const float a1lo = (float)E1 / ftot, a1hi = (float)(E1 + j) / ftot;
		const float a2lo = (float)E2 / ftot, a2hi = (float)(E2 + j) / ftot;
		const uint32_t itmin = vd + nidx;
		const float blo = (float)rfmin / (float)(itmin + j), bhi = (float)(rfmin + srmax) / (float)itmin;
		const uint32_t P0lo = __float_as_uint(a0lo * q0), P0hi = __float_as_uint(a0hi * q0);
		const uint32_t P1lo = __float_as_uint(a1lo * q1), P1hi = __float_as_uint(a1hi * q1);
		const uint32_t Pilo = __float_as_uint(a2lo * blo), Pihi = __float_as_uint(a2hi * bhi);
		/* x3.c:162-172 as a predicate: CTX1 iff p1 > max(p0, pi); else CTX0 iff p0 > pi; else IDX1.  (floats >= +0 compare like
		 * their bit patterns) */
		const uint32_t mxlo = P0lo > Pilo ? P0lo : Pilo, mxhi = P0hi > Pihi ? P0hi : Pihi;
		uint32_t mode = 3; /* 3 = undecided */
		if (P1lo > mxhi) mode = X3_E_CTX1;
		else if (P1hi <= mxlo) {
			if (P0lo > Pihi) mode = X3_E_CTX0;
			else if (P0hi <= Pilo) mode = X3_E_IDX1;
		}
#ifdef X3_EMU
		if (getenv("X3_FORCE_AMB")) mode = 3; /* test hook: push every hit through the exact serial path */
#endif
		if (!in) mode = 4; /* padding lanes take no part */
		uint64_t m0 = x3_ballot(mode == X3_E_CTX0), m1 = x3_ballot(mode == X3_E_CTX1), m2 = x3_ballot(mode == X3_E_IDX1);
		uint64_t amb = x3_ballot(mode == 3);
		while (amb) { /* resolve in order, exact counters */
			const uint32_t l = (uint32_t)x3_ctz64(amb);
			amb &= amb - 1;
			const uint64_t bl = ((uint64_t)1 << l) - 1;
			const uint32_t e0 = E0 + (uint32_t)x3_popc64(m0 & bl), e1 = E1 + (uint32_t)x3_popc64(m1 & bl);
			const uint32_t e2 = E2 + (uint32_t)x3_popc64(m2 & bl), ni = nidx + (uint32_t)x3_popc64(m2 & bl);
			const uint32_t rl = x3_readlane_u32(vr, l);
			const uint64_t same = x3_ballot(vr == rl);
			const uint32_t rf = x3_readlane_u32(rfmin, l) + (uint32_t)x3_popc64(m2 & same & bl);
			const uint32_t tot = 2051u + x3_readlane_u32(vs, l), itot = x3_readlane_u32(vd, l) + ni;
			/* lanes 0..3: ev0/tot, ev1/tot, ev2/tot, rf/itot in ONE vector division */
			const uint32_t num = lane == 0 ? e0 : lane == 1 ? e1 : lane == 2 ? e2 : rf;
			const uint32_t den = lane == 3 ? itot : tot;
			const float quot = (float)num / (float)den;
			const uint32_t ql0 = x3_readlane_u32(__float_as_uint(q0), l), ql1 = x3_readlane_u32(__float_as_uint(q1), l);
			const uint32_t qb = x3_readlane_u32(__float_as_uint(quot), 3); /* (not inside the select: every lane must take part) */
			const uint32_t mult = lane == 0 ? ql0 : lane == 1 ? ql1 : qb;
			const uint32_t p = __float_as_uint(quot * __uint_as_float(mult));
			const uint32_t p0 = x3_readlane_u32(p, 0), p1 = x3_readlane_u32(p, 1), pi = x3_readlane_u32(p, 2);
			uint32_t md = X3_E_IDX1, best = pi;
			if (p0 > best) { md = X3_E_CTX0; best = p0; }
			if (p1 > best) md = X3_E_CTX1;
			const uint64_t bit = (uint64_t)1 << l;
			if (md == X3_E_CTX0) m0 |= bit; else if (md == X3_E_CTX1) m1 |= bit; else m2 |= bit;
		}
		const uint32_t fin = ((m0 >> lane) & 1) ? X3_E_CTX0 : ((m1 >> lane) & 1) ? X3_E_CTX1 : X3_E_IDX1;
		if (EMIT) {
			/* what x3_code_v2_run would otherwise recover from the modes with scans: the model_events freqs before the hit (x3.c:176-177), and
			 * the stream's IDX1-coded hits as a list (rank, hit) in time order for x3_idxstat_kernel */
			if (in) { a.pe0[g] = E0 + (uint32_t)x3_popc64(m0 & below); a.pe1[g] = E1 + (uint32_t)x3_popc64(m1 & below); }
			if (in && fin == X3_E_IDX1) { const uint32_t j = h0 + (nidx - nidx_start) + (uint32_t)x3_popc64(m2 & below); a.ilist_rank[j] = vr; a.ilist_hit[j] = g; }
			if (a.nzl) { /* total of the model the hit's symbol is coded under: the context's, or model_index1's = elements + earlier IDX1 uses */
				const uint32_t tsel = fin == X3_E_CTX0 ? vt0 : fin == X3_E_CTX1 ? vt1 : vd + nidx + (uint32_t)x3_popc64(m2 & below);
				const uint64_t NZ = x3_ballot(in && tsel <= 1u);
				if (in) a.nzl[g] = nnoop + (uint32_t)x3_popc64(NZ & below);
				nnoop += (uint32_t)x3_popc64(NZ);
			}
		}
		if (in) {
			a.mode[g] = fin;
			if (fin == X3_E_IDX1) { if (lds_r) atomicAdd(&sidx[vr], 1u); else atomicAdd(&idxf[vr], 1u); } /* inc_model(&model_index1, index), x3.c:188 */
		}
		E0 += (uint32_t)x3_popc64(m0); E1 += (uint32_t)x3_popc64(m1); /* inc_model(&model_events, mode), x3.c:177 */
		const uint32_t c2 = (uint32_t)x3_popc64(m2);
		E2 += c2; nidx += c2;
		if (ALL_LDS) x3_wave_order(); else x3_wave_sync(); /* (ranks beyond the LDS table live in global memory: their atomics are ordered by the fence) */
	}
	if (a.evfinal && lane == 0) { uint32_t *ef = a.evfinal + 4 * blockIdx.x; ef[0] = E0; ef[1] = E1; ef[2] = E2; ef[3] = nidx; }
	if (EMIT && a.nnoop && lane == 0) a.nnoop[blockIdx.x] = nnoop;
	if (st1 && lane == 0) { st1[0] = E0; st1[1] = E1; st1[2] = E2; st1[3] = nidx; st1[4] = H; st1[5] = Dc; }
}

template <uint32_t NIDX, bool EMIT>
__device__ static void x3_modes_body(const X3ModesArgs &a)
{
	X3_LDS uint32_t sidx[NIDX];
	uint32_t *spre = nullptr;
	X3_LDS uint32_t scr[256];
	const uint32_t c = blockIdx.x, lane = x3_lane();
	const uint32_t H = a.parsed[c].hits, h0 = a.ho[c], Dc = a.parsed[c].dict_elems;
	uint32_t *idxf = a.idxfreq + a.dof[c];
	const uint32_t nl = Dc < NIDX ? Dc : NIDX;
	/* saved state of an earlier, shorter prefix of this stream (X3ModesArgs::state): only with the whole table in LDS */
	uint32_t *st = (!EMIT && NIDX == X3_IDXF_LDS && a.state && Dc <= NIDX) ? a.state + (size_t)c * X3_MODES_STATE_STRIDE : nullptr;
	const bool resumed = st && a.resume && st[4] <= H && st[5] <= Dc && st[4] > 0;
	const uint32_t dsaved = resumed ? st[5] : 0u;
	for (uint32_t i = lane; i < nl; i += X3_WAVE) sidx[i] = i < dsaved ? st[8 + i] : 1u;
	x3_wave_sync();
	if (EMIT) x3_modes_loop<true, NIDX, true>(a, sidx, spre, scr, idxf, H, h0, lane, Dc); /* (the caller picked a table that holds every rank) */
	else if (Dc <= NIDX) x3_modes_loop<true, NIDX, false>(a, sidx, spre, scr, idxf, H, h0, lane, Dc, resumed ? st[4] : 0u, resumed ? st : nullptr, st);
	else x3_modes_loop<false, NIDX, false>(a, sidx, spre, scr, idxf, H, h0, lane, Dc);
	if (st) { x3_wave_sync(); for (uint32_t i = lane; i < nl; i += X3_WAVE) st[8 + i] = sidx[i]; }
	else if (a.state && lane == 0) a.state[(size_t)c * X3_MODES_STATE_STRIDE + 4] = 0; /* nothing to resume from */
}

/* The mode chain of one SLICE of every stream (K3 in slices, code4.hip): the same loop on the slice's hits, continued from the state the earlier
 * slices left in evfinal[4c ..] = {E0, E1, E2, IDX1 uses}, nnoop[c] and the per-rank frequencies idxfreq[elem_off + rank] (model_index1, x3.c:188,419:
 * a new element's symbol starts at 1), and leaving it there for the next slice. */
template <uint32_t NIDX>
__device__ static void x3s_modes_body(const X3ModesArgs &a)
{
	X3_LDS uint32_t sidx[NIDX];
	X3_LDS uint32_t scr[256];
	const uint32_t c = blockIdx.x, lane = x3_lane();
	const X3Slice sl = a.slice[c];
	const uint32_t H = sl.h1 - sl.h0, Dc = sl.d1;
	uint32_t *gidx = a.idxfreq + a.chunks[c].elem_off;
	for (uint32_t i = lane; i < Dc && i < NIDX; i += X3_WAVE) sidx[i] = i < sl.d0 ? gidx[i] : 1u;
	x3_wave_sync();
	uint32_t *ef = a.evfinal + 4 * c;
	const uint32_t st0[4] = { ef[0], ef[1], ef[2], ef[3] };
	x3_modes_loop<true, NIDX, true>(a, sidx, nullptr, scr, gidx, H, sl.sh, lane, Dc, 0, st0, nullptr, a.nnoop[c], true);
	x3_wave_sync();
	for (uint32_t i = lane; i < Dc && i < NIDX; i += X3_WAVE) gidx[i] = sidx[i];
}

/* ============================================================================================================
 * serial pass 2: the arithmetic-coder interval recurrence ALONE (ac.c:77-85 + the renormalisation of ac.c:46-75).
 *
 * Per coded symbol (cum, freq, total):   step = range / total;  hi = lo + step*(cum+freq) - 1;  lo = lo + step*cum.
 *   - the division is a multiply by M = floor(2^62/total)+1 computed by the 64 lanes in parallel for 64 symbols at a
 *     time (exact for range <= 2^31, total < 2^31: the error n*eps/2^62 < 2^-31 < 1/total never crosses an integer);
 *   - E1/E2 (ac.c:49-67) shifts out the n leading bits on which lo and hi agree:  n = clz(lo ^ hi) - 1;
 *   - E3 (ac.c:70-74) then drops the k leading positions (below the top bit) where lo has 1 and hi has 0.
 * Nothing is written to the bit stream here: the kernel records (n, the n emitted bits, k) per symbol, and the
 * pending-bit bookkeeping (mScale) + bit placement become prefix sums over those records (emit stage below).
 * ============================================================================================================ */
struct X3Ac2Args {
	const uint32_t *yo;                   /* per chunk: first symbol (nc+1) */
	const uint4 *sym;                     /* per symbol: {cum, freq, magic multiplier, shift} */
	uint32_t *rec_nk;                     /* out per symbol: {lo, hi} after narrowing, before the renormalisation shift, as uint2 */
	uint32_t *final_lo;                   /* out per chunk */
	const uint32_t *seg_off, *seg_len;    /* nullptr: whole streams (yo).  Else per stream: first symbol (ring coordinate) and count of this segment */
	uint32_t *seg_state;                  /* ... and {lo, R} per stream, in / out */
	uint32_t nstreams;                    /* the wide launch form (several wavefronts = streams per workgroup) guards its stream index with this */
	uint32_t compact;                     /* 1: the state of group g of stream c goes to slot (first symbol >> 3) + c + g (consecutive 8-byte stores that fill whole
	                                         lines) instead of the slot of the group's first symbol (one 8-byte store per 64 bytes) -- whole-stream form only */
};

/* The serial chain, state (lo, R = range):  step = R / total;  nlo = lo + step*cum;  sf = step*freq;  nhi = nlo + sf - 1;  then the
 * renormalisation of ac.c:46-75.  E1/E2 zoom into the lower/upper half and E3 into the middle half of the current window, so after s
 * shifts in total the window is SOME [j*h, (j+2)*h) with h = 2^(30-s), and the loops stop exactly when no window of the next level
 * holds [nlo, nhi].  That happens at t = 30 - s with (nhi >> t) - (nlo >> t) <= 1 no longer satisfiable one level down, i.e. with
 * D = nhi - nlo = sf - 1 and b = its top bit:  s = clz(D) - 1 - carry, where carry = the carry into bit b of nlo + D == NOT bit b of
 * (nlo ^ nhi)  (bit b of D is 1).  Both kinds of shift are left shifts that scale the range: lo' = (nlo << s) mod 2^30, R' = sf << s.
 * One clz, one bit test and a subtract-with-borrow replace the E1/E2/E3 loops: 13 scalar instructions per symbol in all
 * (x3_ac2_sym below; disassembly excerpt: profiles/r03_ac2_chain_disassembly.txt).
 * Nothing is written to the bit stream here, not even a record per symbol: the chain stores its STATE (lo, R) once per group of
 * X3_AC2_G = 8 symbols (at the slot of the group's first symbol); the emit stage re-runs the eight steps of every group in parallel
 * (x3_expand_records) to get the per-symbol intervals (nlo, nhi), and derives n (E1/E2 count), k (E3 count), the mScale bookkeeping
 * and the bit placement from those with prefix sums.  (A record store per symbol pair cost the chain 7 %.)
 * D != 0 always: sf >= step >= 2^29 / total, and a model total stays < 2^28 (a stream is at most 2^28 - 4096 bytes, X3H_MAX_CHUNK: x3hip.h has the bound).
 */
/* ONE definition of a chain step for every user: the scalar-unit chain of x3_ac2_kernel, the emulator build of that kernel (tests/emu runs
 * exactly this arithmetic, not a restatement) and the parallel re-run of a group in the emit stage (x3_chain_step).  Advances (lo, R) and
 * returns the narrowed interval (nlo, nhi) with lo as the chain holds it (unreduced, see below).  The only build-specific piece is the
 * borrow of the bit test: s_bitcmp0_b32 + s_subb_u32 on the scalar chain (two instructions, SCC in between), the same expression in C for
 * per-thread values and under the emulator. */
template <bool SALU> /* SALU: the values are wave-uniform and live in scalar registers (the chain of x3_ac2_kernel); else per-thread values */
__device__ static __forceinline__ uint32_t x3_ac2_shift(uint32_t x, uint32_t cz)
{
	/* cz - 1 - NOT(bit (31 - cz) of x) */
#ifndef X3_EMU
	if (SALU) {
		uint32_t sh;
		const uint32_t t = 31u - cz;
		asm("s_bitcmp0_b32 %1, %2\n\ts_subb_u32 %0, %3, 1" : "=s"(sh) : "s"(x), "s"(t), "s"(cz) : "scc");
		return sh;
	}
#endif
	return cz - 1 - (((x >> (31u - cz)) & 1u) ^ 1u);
}
template <bool SALU>
__device__ static __forceinline__ uint2 x3_ac2_sym(uint32_t &lo, uint32_t &R, uint32_t cum, uint32_t fq, uint32_t m, uint32_t msh)
{
	const uint32_t step = (uint32_t)(((uint64_t)R * m) >> 32) >> msh;
	const uint32_t sf = step * fq, D = sf - 1;
	uint2 iv;
	iv.x = lo + step * cum;
	iv.y = iv.x + D;
#ifndef X3_EMU
	const uint32_t cz = (uint32_t)__builtin_clz(D); /* D != 0 always (see above): no guard instruction on the chain */
#else
	const uint32_t cz = (uint32_t)x3_clz32(D);
#endif
	const uint32_t sh = x3_ac2_shift<SALU>(iv.x ^ iv.y, cz);
	/* both shifts in ONE s_lshl_b64 of the pair {sf (low word), nlo (high word)}: sf << sh stays below 2^32 (it is the renormalised
	 * range), so nothing crosses into the high word, whose own top bits fall off as in a 32-bit shift */
	const uint64_t pr = (((uint64_t)iv.x << 32) | sf) << sh;
	lo = (uint32_t)(pr >> 32);
	R = (uint32_t)pr;
	return iv;
}

#ifndef X3_EMU
/* Everything on the chain lives in SGPRs.  Operands arrive by s_load_dwordx16 (4 symbols per load, 8 symbols = one "group" per
 * ping-pong register set, fetched one group ahead of their use), the state leaves by one s_store_dwordx2 per group through the scalar
 * data cache (written back once at the end): no VALU, no LDS and no v_readlane on the chain.  A vector load per 512 symbols touches
 * the operand lines far ahead of the scalar loads, so those hit in L2.  Scalar memory returns out of order, so the only usable wait
 * is lgkmcnt(0); the load of the NEXT group is therefore issued right after the wait for the current one.  The asm blocks carry the
 * in-flight registers as "+s" operands so the compiler keeps them pinned and orders their uses behind the wait.
 * The operand array has X3_SYM_PAD readable entries behind the last symbol, so the fetch one group ahead needs no clamping.
 * lo is NOT reduced mod 2^30 after the shift: the two stray bits (30, 31) never reach a bit the chain looks at (they cancel in
 * nlo ^ nhi at bit 30, the only place they could matter) and are shifted out or stay put; x3_expand_records re-runs the chain with
 * the same unreduced lo and takes the stray bits off the intervals it writes: 13 instructions per symbol + 1/8 store. */
typedef uint32_t x3_u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t x3_u32x4 __attribute__((ext_vector_type(4)));

#define X3_AC2_SYM(Q, J, NLO, NHI)                                                                                              \
	{                                                                                                                           \
		const uint2 iv_ = x3_ac2_sym<true>(lo, R, (Q)[4 * (J)], (Q)[4 * (J) + 1], (Q)[4 * (J) + 2], (Q)[4 * (J) + 3]);          \
		NLO = iv_.x; NHI = iv_.y;                                                                                               \
	}
#define X3_AC2_STATE(OFF)                                                                                                       \
	{                                                                                                                           \
		const uint64_t stt = ((uint64_t)R << 32) | lo;                                                                          \
		asm volatile("s_store_dwordx2 %0, %1, %2" : : "s"(stt), "s"(recp), "n"(OFF) : "memory");                               \
	}
#define X3_AC2_GROUP(Q0, Q1, OFF)                                                                                               \
	{                                                                                                                           \
		uint32_t nl_, nh_;                                                                                                      \
		X3_AC2_STATE((OFF) * SST)                                                                                                     \
		X3_AC2_SYM(Q0, 0, nl_, nh_) X3_AC2_SYM(Q0, 1, nl_, nh_) X3_AC2_SYM(Q0, 2, nl_, nh_) X3_AC2_SYM(Q0, 3, nl_, nh_)          \
		X3_AC2_SYM(Q1, 0, nl_, nh_) X3_AC2_SYM(Q1, 1, nl_, nh_) X3_AC2_SYM(Q1, 2, nl_, nh_) X3_AC2_SYM(Q1, 3, nl_, nh_)          \
		(void)nl_; (void)nh_;                                                                                                   \
	}
/* wait for everything in flight (the group about to be used included), then start the loads of the group after it */
#define X3_AC2_FETCH(N0, N1, C0, C1, OFF)                                                                                       \
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 %0, %4, %5\n\ts_load_dwordx16 %1, %4, %5+0x40"                         \
	             : "=&s"(N0), "=&s"(N1), "+s"(C0), "+s"(C1) : "s"(symp), "n"(OFF) : "memory")

template <bool CP> /* CP: compact state slots (X3Ac2Args::compact) */
__device__ static void x3_ac2_body(const X3Ac2Args &a, const uint32_t c)
{
	constexpr uint32_t SST = CP ? 8u : 64u; /* bytes from one group's state to the next */
	const uint32_t lane = x3_lane();
	uint32_t y0, Y, lo = 0, R = 0x80000000u; /* ac_init, ac.c:35-41: [0, 0x7FFFFFFF] */
	if (a.seg_off) { y0 = x3_uniform(a.seg_off[c]); Y = x3_uniform(a.seg_len[c]); lo = x3_uniform(a.seg_state[2 * c]); R = x3_uniform(a.seg_state[2 * c + 1]); }
	else { y0 = x3_uniform(a.yo[c]); Y = x3_uniform(a.yo[c + 1]) - y0; }
	uint64_t symp = (uint64_t)(a.sym + y0);
	uint64_t recp = (uint64_t)((uint2 *)a.rec_nk + (CP ? (y0 >> 3) + c : y0));
	const uint32_t G = Y >> 3; /* whole groups of 8 symbols */
	if (G) {
		x3_u32x16 A0, A1, B0, B1;
		uint32_t dummy = 0; /* destination of the L2-warming loads: kept live to the end so its register is never reused under a load in flight */
		const uint64_t pfmax = symp + (uint64_t)Y * 16u;
		asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(A0), "=&s"(A1) : "s"(symp) : "memory");
		uint32_t g = 0;
		for (; g + 4 <= G; g += 4) { /* 32 symbols per trip */
			if ((g & 63u) == 0) { /* every 512 symbols: pull the operand lines of symbols [+512, +1024) towards L2 (result unused) */
				uint64_t pf = symp + 64u * 128u + lane * 128u;
				pf = pf < pfmax ? pf : pfmax;
				asm volatile("s_waitcnt vmcnt(0)\n\tglobal_load_dword %0, %1, off" : "+v"(dummy) : "v"(pf) : "memory");
			}
			X3_AC2_FETCH(B0, B1, A0, A1, 128);
			X3_AC2_GROUP(A0, A1, 0)
			X3_AC2_FETCH(A0, A1, B0, B1, 256);
			X3_AC2_GROUP(B0, B1, 1)
			X3_AC2_FETCH(B0, B1, A0, A1, 384);
			X3_AC2_GROUP(A0, A1, 2)
			X3_AC2_FETCH(A0, A1, B0, B1, 512);
			X3_AC2_GROUP(B0, B1, 3)
			symp += 512;
			recp += 4 * SST;
		}
		for (; g < G; g++) { /* < 4 leftover groups; A holds the current one */
			X3_AC2_FETCH(B0, B1, A0, A1, 128);
			X3_AC2_GROUP(A0, A1, 0)
			asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(B0), "+s"(B1) : : "memory"); /* landed: only now may the registers be copied */
			A0 = B0; A1 = B1;
			symp += 128;
			recp += SST;
		}
		asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : : "v"(dummy) : "memory"); /* the last prefetches */
	}
	if ((G << 3) < Y) X3_AC2_STATE(0) /* the (shorter) last group */
	for (uint32_t y = G << 3; y < Y; y++) { /* < 8 leftover symbols */
		x3_u32x4 Q;
		uint32_t nlo, nhi;
		asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(Q) : "s"(symp) : "memory");
		X3_AC2_SYM(Q, 0, nlo, nhi)
		(void)nlo; (void)nhi;
		symp += 16;
	}
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" : : : "memory"); /* records: scalar cache -> L2 */
	if (lane == 0) {
		a.final_lo[c] = lo & 0x3FFFFFFFu;
		if (a.seg_off) { a.seg_state[2 * c] = lo; a.seg_state[2 * c + 1] = R; } /* lo as the chain holds it (stray bits included): the next segment continues it */
	}
}
#else
/* the CPU emulator build (tests only): the same step function, one symbol at a time, no scalar-cache staging */
template <bool CP>
__device__ static void x3_ac2_body(const X3Ac2Args &a, const uint32_t c)
{
	const uint32_t lane = x3_lane();
	uint32_t y0, Y, lo = 0, R = 0x80000000u;
	if (a.seg_off) { y0 = a.seg_off[c]; Y = a.seg_len[c]; lo = a.seg_state[2 * c]; R = a.seg_state[2 * c + 1]; }
	else { y0 = a.yo[c]; Y = a.yo[c + 1] - y0; }
	x3_wave_sync();
	for (uint32_t y = 0; y < Y; y++) {
		const uint4 q = a.sym[y0 + y];
		if (lane == 0 && (y % X3_AC2_G) == 0) { /* the state before the group */
			const size_t slot = CP ? (size_t)(y0 >> 3) + c + y / X3_AC2_G : (size_t)(y0 + y);
			a.rec_nk[2 * slot] = lo; a.rec_nk[2 * slot + 1] = R;
		}
		(void)x3_ac2_sym<false>(lo, R, q.x, q.y, q.z, q.w); /* the device chain's own step function (lo not reduced mod 2^30) */
	}
	x3_wave_sync();
	if (lane == 0) {
		a.final_lo[c] = lo & 0x3FFFFFFFu;
		if (a.seg_off) { a.seg_state[2 * c] = lo; a.seg_state[2 * c + 1] = R; }
	}
}
#endif

#ifndef X3_EMU
__global__ void __launch_bounds__(X3_WAVE) x3_modes_kernel(X3ModesArgs a) { x3_modes_body<X3_IDXF_LDS, false>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_modes_many_kernel(X3ModesArgs a) { x3_modes_body<X3_IDXF_LDS_SMALL, false>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_modes_stream_kernel_s(X3ModesArgs a) { x3_modes_body<2048, true>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_modes_stream_kernel_l(X3ModesArgs a) { x3_modes_body<X3_STREAM_DMAX, true>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_ac2_kernel(X3Ac2Args a) { x3_ac2_body<false>(a, blockIdx.x); }
__global__ void __launch_bounds__(X3_WAVE) x3_ac2_compact_kernel(X3Ac2Args a) { x3_ac2_body<true>(a, blockIdx.x); }
/* four streams per workgroup, one per wavefront: the wavefronts of a workgroup are spread over the CU's four SIMDs */
__global__ void __launch_bounds__(4 * X3_WAVE) x3_ac2_wide_kernel(X3Ac2Args a)
{
	const uint32_t c = x3_uniform(blockIdx.x * 4u + (threadIdx.x >> 6));
	if (c < a.nstreams) x3_ac2_body<false>(a, c);
}
__global__ void __launch_bounds__(4 * X3_WAVE) x3_ac2_wide_compact_kernel(X3Ac2Args a)
{
	const uint32_t c = x3_uniform(blockIdx.x * 4u + (threadIdx.x >> 6));
	if (c < a.nstreams) x3_ac2_body<true>(a, c);
}
__global__ void __launch_bounds__(X3_WAVE) x3s_modes_kernel_s(X3ModesArgs a) { x3s_modes_body<2048>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3s_modes_kernel_l(X3ModesArgs a) { x3s_modes_body<X3_STREAM_DMAX>(a); }
int x3s_modes_launch(const X3ModesArgs &a, uint32_t nc, uint64_t max_dict, hipStream_t st)
{
	if (max_dict <= 2048) hipLaunchKernelGGL(x3s_modes_kernel_s, dim3(nc), dim3(X3_WAVE), 0, st, a);
	else hipLaunchKernelGGL(x3s_modes_kernel_l, dim3(nc), dim3(X3_WAVE), 0, st, a);
	HIPCHK(hipGetLastError());
	return X3H_OK;
}
static void launch_modes(const X3ModesArgs &a, uint32_t nchunks, hipStream_t st, uint64_t max_dict)
{
	if (a.pe0) { /* many streams, model state handed out by the kernel: a table (and its running sums) that holds every rank */
		if (max_dict <= 2048) hipLaunchKernelGGL(x3_modes_stream_kernel_s, dim3(nchunks), dim3(X3_WAVE), 0, st, a);
		else hipLaunchKernelGGL(x3_modes_stream_kernel_l, dim3(nchunks), dim3(X3_WAVE), 0, st, a);
		return;
	}
	/* the LDS table decides how many streams share a CU: a batch that oversubscribes the chip gets the small one */
	if (nchunks > 256) hipLaunchKernelGGL(x3_modes_many_kernel, dim3(nchunks), dim3(X3_WAVE), 0, st, a);
	else hipLaunchKernelGGL(x3_modes_kernel, dim3(nchunks), dim3(X3_WAVE), 0, st, a);
}
/* One-wavefront workgroups are not spread evenly over a CU's four SIMDs: with four chains per CU (1024 streams) two of them share a SIMD's
 * issue slot and a symbol costs 49 ns instead of 25; as four wavefronts of ONE workgroup they get a SIMD each (31.5 ns; 8 per CU: 52 against
 * 73; profiles/r03_coder_chains_per_workgroup.txt).  Up to two streams per CU the narrow form keeps every chain on its own CU.  X3H_AC2_WIDE = smallest stream
 * count that takes the wide form (0: never). */
static void launch_ac2(X3Ac2Args a, uint32_t nchunks, hipStream_t st)
{
	const char *we = getenv("X3H_AC2_WIDE"); /* (read per launch: the tests switch it inside one process) */
	const uint32_t wide_min = we ? (uint32_t)atoi(we) : 513u;
	a.nstreams = nchunks;
	if (a.seg_off) a.compact = 0;
	if (wide_min && nchunks >= wide_min) {
		if (a.compact) hipLaunchKernelGGL(x3_ac2_wide_compact_kernel, dim3((nchunks + 3) / 4), dim3(4 * X3_WAVE), 0, st, a);
		else hipLaunchKernelGGL(x3_ac2_wide_kernel, dim3((nchunks + 3) / 4), dim3(4 * X3_WAVE), 0, st, a);
	} else if (a.compact) hipLaunchKernelGGL(x3_ac2_compact_kernel, dim3(nchunks), dim3(X3_WAVE), 0, st, a);
	else hipLaunchKernelGGL(x3_ac2_kernel, dim3(nchunks), dim3(X3_WAVE), 0, st, a);
}
__device__ static __forceinline__ uint32_t x3_brev32(uint32_t v) { return __brev(v); }
#else
static void smodes_tramp(void *p) { x3s_modes_body<X3_STREAM_DMAX>(*(const X3ModesArgs *)p); }
int x3s_modes_launch(const X3ModesArgs &a, uint32_t nc, uint64_t, hipStream_t) { x3emu_launch(smodes_tramp, (void *)&a, dim3(nc), dim3(X3_WAVE)); return X3H_OK; }
static void modes_tramp(void *p) { x3_modes_body<X3_IDXF_LDS, false>(*(const X3ModesArgs *)p); }
static void modes_stream_tramp(void *p) { x3_modes_body<X3_STREAM_DMAX, true>(*(const X3ModesArgs *)p); }
static void ac2_tramp(void *p) { const X3Ac2Args &a = *(const X3Ac2Args *)p; if (a.compact && !a.seg_off) x3_ac2_body<true>(a, blockIdx.x); else x3_ac2_body<false>(a, blockIdx.x); }
static void launch_modes(const X3ModesArgs &a, uint32_t nchunks, hipStream_t, uint64_t) { x3emu_launch(a.pe0 ? modes_stream_tramp : modes_tramp, (void *)&a, dim3(nchunks), dim3(X3_WAVE)); }
static void launch_ac2(const X3Ac2Args &a, uint32_t nchunks, hipStream_t) { x3emu_launch(ac2_tramp, (void *)&a, dim3(nchunks), dim3(X3_WAVE)); }
static inline uint32_t x3_brev32(uint32_t v) { uint32_t r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }
#endif

/* E1/E2 count n and E3 count k of a narrowed interval (ac.c:49-74 in closed form; see x3_ac2_body) */
__device__ static __forceinline__ uint32_t x3_rec_n(uint32_t nlo, uint32_t nhi) { return (uint32_t)x3_clz32(nlo ^ nhi) - 1; }
__device__ static __forceinline__ uint32_t x3_rec_k(uint32_t nlo, uint32_t nhi)
{
	const uint32_t n = x3_rec_n(nlo, nhi);
	const uint32_t y = ((((nhi | ~nlo) << n) | ~(0xFFFFFFFFu << n)) & 0x3FFFFFFFu);
	return (uint32_t)x3_clz32(y) - 2;
}

/* One step of the coder chain exactly as x3_ac2_kernel takes it (lo unreduced), for the parallel re-run of a group: returns the
 * narrowed interval with the stray top bits of lo removed, advances (lo, R). */
__device__ static __forceinline__ uint2 x3_chain_step(uint32_t &lo, uint32_t &R, const uint4 q)
{
	const uint32_t stray = lo & 0xC0000000u;
	uint2 r = x3_ac2_sym<false>(lo, R, q.x, q.y, q.z, q.w);
	r.x -= stray; r.y -= stray;
	return r;
}

/* OR `nbits` (<= 32) bits of `val` into the little-endian 32-bit word stream at bit position `bitpos` (bio.c:49-72 layout) */
__device__ static __forceinline__ void x3_or_bits(uint32_t *out32, uint32_t capw, uint64_t bitpos, uint32_t val, uint32_t nbits)
{
	if (!nbits) return;
	if (nbits < 32) val &= (1u << nbits) - 1;
	const uint64_t w = bitpos >> 5;
	const uint32_t sh = (uint32_t)(bitpos & 31);
	const uint32_t v0 = val << sh;
	if (v0 && w < capw) atomicOr(&out32[w], v0);
	if (sh && sh + nbits > 32) {
		const uint32_t v1 = val >> (32 - sh);
		if (v1 && w + 1 < capw) atomicOr(&out32[w + 1], v1);
	}
}

/* the same into a buffer of `nw` words whose bit 0 is bit `bitpos` = 0 (LDS: the tile buffer of x3_emit_body) */
__device__ static __forceinline__ void x3_or_bits_buf(uint32_t *buf, uint32_t nw, uint32_t bitpos, uint32_t val, uint32_t nbits)
{
	if (!nbits) return;
	if (nbits < 32) val &= (1u << nbits) - 1;
	const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
	const uint32_t v0 = val << sh;
	if (v0 && w < nw) atomicOr(&buf[w], v0);
	if (sh && sh + nbits > 32) {
		const uint32_t v1 = val >> (32 - sh);
		if (v1 && w + 1 < nw) atomicOr(&buf[w + 1], v1);
	}
}
__device__ static __forceinline__ void x3_or_run_buf(uint32_t *buf, uint32_t nw, uint32_t bitpos, uint32_t bit, uint32_t count)
{
	if (!bit) return;
	while (count) {
		const uint32_t c = count < 32 ? count : 32;
		x3_or_bits_buf(buf, nw, bitpos, 0xFFFFFFFFu, c);
		bitpos += c;
		count -= c;
	}
}

/* a run of `count` equal bits */
__device__ static __forceinline__ void x3_or_run(uint32_t *out32, uint32_t capw, uint64_t bitpos, uint32_t bit, uint32_t count)
{
	if (!bit) return; /* the stream is pre-zeroed */
	while (count) {
		const uint32_t c = count < 32 ? count : 32;
		x3_or_bits(out32, capw, bitpos, 0xFFFFFFFFu, c);
		bitpos += c;
		count -= c;
	}
}

/* ============================================================================================================
 * Bit emission for batches of many streams: ONE workgroup per stream walks the stream's symbols in tiles of 1024 groups (8192 symbols)
 * and carries the pending-bit count (mScale, ac.c:49-74) and the bit position from tile to tile -- instead of chip-wide passes
 * (re-run groups, two scans, pending, lengths, scan, OR-writer) over arrays keyed by stream.  Per tile a thread re-runs the eight
 * chain steps of its group from the state x3_ac2_kernel stored (x3_chain_step), keeps (n, emitted bits, k) of its symbols in registers,
 * and three workgroup scans place its bits:
 *   pending before a thread  = k summed since the last symbol that emitted (E1/E2 count n >= 1) -- a sum scan, a max scan of "last
 *                              thread with an emitting symbol" and that thread's tail;
 *   bit offset of a thread   = sum scan of the bits the threads write (n + the pending bits in front of each emitting symbol).
 * Then ac_encode_flush (ac.c:115-126), bio_close's word padding (bio.c:105-112) and the stream's result record.
 * ============================================================================================================ */
#define X3_EMIT_THREADS 1024u /* (512: 3.2 ms, 256: 3.4 ms on the 1024-stream mix) */
#ifndef X3_EMIT_LDSW
#define X3_EMIT_LDSW 4096u
#endif                      /* words of a tile's output assembled in LDS: 16 bits per symbol of a tile of 8192 (text emits ~1.4); a tile with more ORs straight into memory */

__device__ static __forceinline__ uint32_t x3_wave_incl_maxscan_u32(uint32_t v)
{
#ifndef X3_EMU
	int x = (int)v;
#define X3_MAX_STEP(ctrl, rows) { const int y = __builtin_amdgcn_update_dpp(x, x, ctrl, rows, 0xf, false); x = (uint32_t)y > (uint32_t)x ? y : x; }
	X3_MAX_STEP(0x111, 0xf) X3_MAX_STEP(0x112, 0xf) X3_MAX_STEP(0x114, 0xf) X3_MAX_STEP(0x118, 0xf) X3_MAX_STEP(0x142, 0xa) X3_MAX_STEP(0x143, 0xc)
#undef X3_MAX_STEP
	return (uint32_t)x;
#else
	const int l = (int)x3_lane();
	for (int d = 1; d < X3_WAVE; d <<= 1) { const uint32_t u = x3emu_shfl(v, l >= d ? l - d : l); if (l >= d && u > v) v = u; }
	return v;
#endif
}

template <uint32_t NT> /* threads of the workgroup: 256 when a thousand streams share the chip, 1024 when a few long ones do */
__device__ static void x3_emit_body(const X3EmitArgs &a)
{
	const uint32_t NW = NT / X3_WAVE;
	X3_LDS uint32_t s_ksum[NT], s_tail[NT], s_lr[NT];
	X3_LDS uint32_t s_w[3][NT / X3_WAVE];
	X3_LDS uint32_t s_bits[X3_EMIT_LDSW]; /* the tile's bits, assembled here and written out as whole words (zero between tiles) */
	const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = x3_lane(), wave = tid / X3_WAVE;
	uint32_t first, end;
	if (a.seg_off) { first = a.seg_off[c]; end = first + a.seg_len[c]; } else { first = a.yoc[c]; end = a.yoc[c + 1]; }
	uint32_t *out32 = (uint32_t *)(a.out + a.chunks[c].out_off);
	const uint32_t capw = (uint32_t)(a.chunks[c].out_cap / 4);
	uint32_t carry_pend = 0;
	uint64_t carry_pos = 0;
	if (a.carry) { carry_pend = a.carry[4 * c]; carry_pos = (uint64_t)a.carry[4 * c + 1] | ((uint64_t)a.carry[4 * c + 2] << 32); }
	for (uint32_t i = tid; i < X3_EMIT_LDSW; i += NT) s_bits[i] = 0;
	for (uint32_t tb = first; tb < end; tb += NT * X3_AC2_G) {
		const uint32_t gi = tb + tid * X3_AC2_G;
		const uint32_t cnt = gi >= end ? 0u : (end - gi < X3_AC2_G ? end - gi : X3_AC2_G);
		uint32_t eb[X3_AC2_G], kq[X3_AC2_G];
		uint32_t ktot = 0, tail = 0, has_reset = 0, nsum = 0, pinner = 0;
		if (cnt) {
			const size_t slot = a.compact ? (size_t)(first >> 3) + c + ((gi - first) >> 3) : (size_t)gi;
			uint32_t lo = a.state[2 * slot], R = a.state[2 * slot + 1];
#pragma unroll
			for (uint32_t j = 0; j < X3_AC2_G; j++) {
				eb[j] = 1; kq[j] = 0;
				if (j < cnt) {
					const uint2 r = x3_chain_step(lo, R, a.sym[gi + j]);
					const uint32_t n = x3_rec_n(r.x, r.y); /* <= 30: the interval has at least two values */
					eb[j] = (1u << n) | (x3_brev32(r.x << 1) & ((1u << n) - 1)); /* bit i = i-th emitted bit = bit 30-i of lo */
					kq[j] = x3_rec_k(r.x, r.y);
					if (n >= 1) {
						nsum += n;
						if (has_reset) pinner += tail; /* the pending bits in front of this symbol come from this thread alone (those in
						                                * front of the thread's FIRST emitting symbol come from earlier threads too) */
						has_reset = 1;
						tail = 0;
					}
					if (has_reset) tail += kq[j];
					else ktot += kq[j]; /* k before the first emitting symbol */
				}
			}
		}
		/* ktot: k of the symbols before the thread's first emitting symbol (all of them if none emits); tail: k from its last emitting symbol on */
		const uint32_t kall = has_reset ? 0u : ktot; /* what the thread adds to a pending run that passes through it */
		/* ---- scan 1: sum of kall, scan 2: last thread (+1) with an emitting symbol ---- */
		const uint32_t ks_w = x3_wave_incl_scan_u32(kall);
		const uint32_t lr_w = x3_wave_incl_maxscan_u32(has_reset ? tid + 1 : 0u);
		if (lane == X3_WAVE - 1) { s_w[0][wave] = ks_w; s_w[1][wave] = lr_w; }
		__syncthreads();
		uint32_t kbase = 0, lrbase = 0;
		for (uint32_t w = 0; w < wave; w++) { kbase += s_w[0][w]; lrbase = s_w[1][w] > lrbase ? s_w[1][w] : lrbase; }
		const uint32_t ksum = ks_w + kbase;                       /* inclusive */
		const uint32_t lr = lr_w > lrbase ? lr_w : lrbase;        /* inclusive: last emitting thread <= tid, +1; 0: none */
		s_ksum[tid] = ksum; s_tail[tid] = tail; s_lr[tid] = lr;
		__syncthreads();
		/* pending count in front of this thread = after thread tid-1 */
		uint32_t incoming = carry_pend;
		if (tid > 0) {
			const uint32_t plr = s_lr[tid - 1], pks = s_ksum[tid - 1];
			incoming = plr ? s_tail[plr - 1] + (pks - s_ksum[plr - 1]) : carry_pend + pks;
		}
		const uint32_t bits = nsum + pinner + (has_reset ? incoming + ktot : 0u);
		const uint32_t bs_w = x3_wave_incl_scan_u32(bits);
		if (lane == X3_WAVE - 1) s_w[2][wave] = bs_w;
		__syncthreads();
		uint32_t bbase = 0, btot = 0;
		for (uint32_t w = 0; w < NW; w++) { if (w < wave) bbase += s_w[2][w]; btot += s_w[2][w]; }
		uint64_t bp = carry_pos + bbase + bs_w - bits;
		/* ---- write ----  A tile's bits are one contiguous range of the stream: they are ORed together in LDS and leave as whole words (plain
		 * stores; the first and the last word are shared with the neighbouring tiles: OR).  Straight into memory a tile's ~2 500 atomic ORs
		 * meet on ~100 words of one L2 channel and the kernel waits for them. */
		const uint64_t w0 = carry_pos >> 5;
		const uint32_t nw = (uint32_t)(((carry_pos + btot + 31) >> 5) - w0);
		const bool inlds = nw <= X3_EMIT_LDSW; /* (uniform) */
		const uint32_t rb = (uint32_t)(bp - (w0 << 5)); /* my first bit inside the buffer (used when inlds: < 32 * X3_EMIT_LDSW) */
		uint32_t pd = incoming, rp = rb;
#pragma unroll
		for (uint32_t j = 0; j < X3_AC2_G; j++) {
			if (j < cnt) {
				const uint32_t n = 31u - (uint32_t)x3_clz32(eb[j]);
				if (n >= 1) {
					const uint32_t rev = eb[j] ^ (1u << n);
					if (inlds) {
						if (!pd) x3_or_bits_buf(s_bits, nw, rp, rev, n);
						else {
							x3_or_bits_buf(s_bits, nw, rp, rev & 1u, 1);
							x3_or_run_buf(s_bits, nw, rp + 1, (rev & 1u) ^ 1u, pd);
							x3_or_bits_buf(s_bits, nw, rp + 1 + pd, rev >> 1, n - 1);
						}
						rp += n + pd;
					} else {
						if (!pd) x3_or_bits(out32, capw, bp, rev, n);
						else {
							x3_or_bits(out32, capw, bp, rev & 1u, 1);
							x3_or_run(out32, capw, bp + 1, (rev & 1u) ^ 1u, pd);
							x3_or_bits(out32, capw, bp + 1 + pd, rev >> 1, n - 1);
						}
						bp += n + pd;
					}
					pd = 0;
				}
				pd += kq[j];
			}
		}
		if (inlds) {
			__syncthreads();
			for (uint32_t i = tid; i < nw; i += NT) {
				const uint32_t v = s_bits[i];
				s_bits[i] = 0; /* ready for the next tile (whose ORs come behind its own barriers) */
				if (v && w0 + i < capw) { if (i == 0 || i + 1 == nw) atomicOr(&out32[w0 + i], v); else out32[w0 + i] = v; }
			}
		}
		/* carries: pending after the tile's last thread, bits so far */
		const uint32_t llr = s_lr[NT - 1], lks = s_ksum[NT - 1];
		carry_pend = llr ? s_tail[llr - 1] + (lks - s_ksum[llr - 1]) : carry_pend + lks;
		carry_pos += btot;
		__syncthreads();
	}
	if (a.carry && tid == 0) { a.carry[4 * c] = carry_pend; a.carry[4 * c + 1] = (uint32_t)carry_pos; a.carry[4 * c + 2] = (uint32_t)(carry_pos >> 32); }
	if (tid == 0 && (!a.carry || a.last)) { /* ac_encode_flush (ac.c:115-126) + bio_close (bio.c:105-112) + result */
		uint64_t nbits = carry_pos;
		if (a.final_lo[c] < 0x20000000u) {
			x3_or_run(out32, capw, nbits + 1, 1u, carry_pend + 1); /* '0' then mScale+1 ones */
			nbits += 2 + (uint64_t)carry_pend;
		} else {
			x3_or_bits(out32, capw, nbits, 1u, 1);
			nbits += 1;
		}
		const uint64_t words = (nbits + 31) / 32;
		X3CodeResult r;
		r.out_len = (uint32_t)(words * 4); r.status = words > capw ? X3_ST_OUT_FULL : X3_ST_OK; r.pairs = a.npairs[c]; r._r = 0;
		r.events[0] = a.evfinal[4 * c] - 1024; r.events[1] = a.evfinal[4 * c + 1] - 1024; r.events[2] = a.evfinal[4 * c + 2] - 1;
		r.events[3] = a.ntok ? a.ntok[c] - a.nhits[c] : a.parsed[c].ntok - a.parsed[c].hits;
		r.events[4] = r.events[5] = r.events[6] = r.events[7] = 0;
		a.result[c] = r;
	}
}

#ifndef X3_EMU
__global__ void __launch_bounds__(X3_EMIT_THREADS) x3_emit_kernel(X3EmitArgs a) { x3_emit_body<X3_EMIT_THREADS>(a); }
__global__ void __launch_bounds__(1024) x3_emit_wide_kernel(X3EmitArgs a) { x3_emit_body<1024>(a); }
static void launch_emit(const X3EmitArgs &a, uint32_t nchunks, hipStream_t st, bool wide = false)
{
	if (wide) hipLaunchKernelGGL(x3_emit_wide_kernel, dim3(nchunks), dim3(1024), 0, st, a);
	else hipLaunchKernelGGL(x3_emit_kernel, dim3(nchunks), dim3(X3_EMIT_THREADS), 0, st, a);
}
#else
static void emit_tramp(void *p) { x3_emit_body<X3_EMIT_THREADS>(*(const X3EmitArgs *)p); }
static void emit_wide_tramp(void *p) { x3_emit_body<1024>(*(const X3EmitArgs *)p); }
static void launch_emit(const X3EmitArgs &a, uint32_t nchunks, hipStream_t, bool wide = false)
{
	if (wide) x3emu_launch(emit_wide_tramp, (void *)&a, dim3(nchunks), dim3(1024));
	else x3emu_launch(emit_tramp, (void *)&a, dim3(nchunks), dim3(X3_EMIT_THREADS));
}
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
		if (p > __atomic_load_n(dmax, __ATOMIC_RELAXED)) atomicMax(dmax, p); /* the running maximum rises O(log) times: almost no hit reaches the atomic */
	});
	uint32_t hmax = 0;
	HIPCHK(hipMemcpyAsync(&hmax, dmax, 4, hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	CHK(csb_run(B, st, nH, bits_for(hmax), key, bs, be, cntA, T + 19 - 0 /* scratch follows */));
	x3_foreach(nH, st, X3_LAMBDA(size_t i) { cum[vA[i]] = cntA[i]; });
	return X3H_OK;
}

/* ============================================================================================================
 * the mode choice as a FIXED-POINT ITERATION (alternative to serial pass 1 when a batch has few, long streams).
 * mode_i = F_i(mode_0 .. mode_{i-1}): the decision of x3.c:152-172 at hit i reads model_events (three counters = counts of the
 * earlier modes) and model_index1 (freq of the hit's rank = 1 + earlier IDX1-coded hits with that rank; total = elements + earlier
 * IDX1 hits).  Given ANY candidate sequence M, all those counters are prefix sums over M, and F(M) is evaluated for every hit at
 * once with the reference's exact float expression.  F is causal, so if M is right on [0, k) then F(M) is right on [0, k]; the
 * reference's sequence is the only fixed point, and M == F(M) proves M is it.  Decisions are insensitive to small counter errors,
 * so the correct prefix grows by large jumps: a handful of chip-wide iterations replace a 5.7 M-step serial chain.
 * Not converged within max_iter (adversarial input): the caller runs the serial kernel instead.
 * T: 17 temporaries of nH+1 words. */
static int modes_fixed_point(X3Code2Bufs &B, hipStream_t st, size_t nH, uint32_t nc, const uint32_t *d_ho, int rbits,
                             const uint32_t *f0, const uint32_t *t0, const uint32_t *f1, const uint32_t *t1, const uint32_t *h_rank,
                             const uint32_t *h_dk, const uint32_t *h_step, uint32_t *mode, uint32_t *const *T, int max_iter,
                             int *iters, bool *converged)
{
	uint32_t *iota = T[0], *ks = T[1], *Rv = T[2], *hf = T[3], *segstart = T[4], *inv = T[5], *segI = T[6], *q0b = T[7], *q1b = T[8];
	uint32_t *lo_ = T[9], *z0 = T[10], *z1 = T[11], *ziS = T[12], *c0s = T[13], *c1s = T[14], *ciS = T[15], *Mn = T[16];
	uint32_t *d_flag = B.maxred.as<uint32_t>() + 4;
	*converged = false;
	*iters = 0;
	/* hits ordered by (rank, stream, time): a run = the hits that share one model_index1 symbol */
	x3_foreach(nH, st, X3_LAMBDA(size_t i) { iota[i] = (uint32_t)i; });
	CHK(x3p_sort_pairs(B.tmp, h_rank, ks, iota, Rv, nH, rbits, st));
	x3_foreach(nH, st, X3_LAMBDA(size_t j) {
		const bool head = j == 0 || ks[j - 1] != ks[j] || find_chunk(d_ho, nc, Rv[j - 1]) != find_chunk(d_ho, nc, Rv[j]);
		hf[j] = head ? (uint32_t)j : 0u;
	});
	CHK(x3p_incl_max_scan(B.tmp, hf, segstart, nH, st));
	x3_foreach(nH, st, X3_LAMBDA(size_t j) { inv[Rv[j]] = (uint32_t)j; segI[Rv[j]] = segstart[j]; });
	/* per-hit constants + the starting guess (the better context, if the tag is in one) */
	uint32_t *M = mode;
	x3_foreach(nH, st, X3_LAMBDA(size_t i) {
		const uint32_t vf0 = f0[i], vf1 = f1[i];
		const float q0 = vf0 ? (float)vf0 / (float)t0[i] : 0.f, q1 = vf1 ? (float)vf1 / (float)t1[i] : 0.f;
		q0b[i] = __float_as_uint(q0); q1b[i] = __float_as_uint(q1);
		lo_[i] = d_ho[find_chunk(d_ho, nc, (uint32_t)i)];
		const uint32_t m = (vf1 && q1 >= q0) ? X3_E_CTX1 : vf0 ? X3_E_CTX0 : X3_E_IDX1;
		M[i] = m;
		z0[i] = m == X3_E_CTX0; z1[i] = m == X3_E_CTX1; ziS[inv[i]] = m == X3_E_IDX1;
	});
	for (int it = 0; it < max_iter; it++) {
		CHK(x3p_excl_scan(B.tmp, z0, c0s, nH, st));
		CHK(x3p_excl_scan(B.tmp, z1, c1s, nH, st));
		CHK(x3p_excl_scan(B.tmp, ziS, ciS, nH, st));
		HIPCHK(hipMemsetAsync(d_flag, 0, 4, st));
		{
			const uint32_t *Mc = M;
			uint32_t *Mo = Mn;
			x3_foreach(nH, st, X3_LAMBDA(size_t i) {
				const uint32_t lo = lo_[i], hb = (uint32_t)i - lo;
				const uint32_t n0 = c0s[i] - c0s[lo], n1 = c1s[i] - c1s[lo], n2 = hb - n0 - n1; /* earlier CTX0 / CTX1 / IDX1 hits of the stream */
				const uint32_t e0 = 1024u + n0, e1 = 1024u + n1, e2 = 1u + n2;              /* model_events freqs, x3.c:239-241 + inc_model per hit */
				const uint32_t tot = 2051u + h_step[i], itot = h_dk[i] + n2;
				const uint32_t rf = 1u + ciS[inv[i]] - ciS[segI[i]];
				const float ftot = (float)tot;
				const uint32_t p0 = __float_as_uint(((float)e0 / ftot) * __uint_as_float(q0b[i]));
				const uint32_t p1 = __float_as_uint(((float)e1 / ftot) * __uint_as_float(q1b[i]));
				const uint32_t pi = __float_as_uint(((float)e2 / ftot) * ((float)rf / (float)itot));
				uint32_t md = X3_E_IDX1, best = pi; /* x3.c:162-172 (floats >= +0 compare like their bit patterns) */
				if (p0 > best) { md = X3_E_CTX0; best = p0; }
				if (p1 > best) md = X3_E_CTX1;
				if (md != Mc[i]) *d_flag = 1u;
				Mo[i] = md;
				z0[i] = md == X3_E_CTX0; z1[i] = md == X3_E_CTX1; ziS[inv[i]] = md == X3_E_IDX1;
			});
		}
		{ uint32_t *t = M; M = Mn; Mn = t; }
		uint32_t changed = 1;
		HIPCHK(hipMemcpyAsync(&changed, d_flag, 4, hipMemcpyDeviceToHost, st));
		HIPCHK(hipStreamSynchronize(st));
		*iters = it + 1;
		if (!changed) { *converged = true; break; }
	}
	if (*converged && M != mode) {
		const uint32_t *Mc = M;
		x3_foreach(nH, st, X3_LAMBDA(size_t i) { mode[i] = Mc[i]; });
	}
	return X3H_OK;
}

/* ============================================================================================================
 * Size estimates (the reference's `sizes[]`, x3.c:43): sizes[mode] += -log2f(prob) per hit (x3.c:52-55,192-193) and per coded symbol of a
 * new fragment (x3.c:253-266), accumulated in IEEE single IN CODING ORDER -- float addition is not associative, and past 2^24 bits the
 * reference's accumulator drops every term below half an ulp, which the printed estimates inherit.  The terms are made by the symbol
 * assembly pass (x3_est_term: the same float expression as the mode choice, -log2 in double rounded to single); here one wavefront per
 * stream replays the four accumulators: 64 terms are loaded at once, then taken lane by lane in order.
 * ============================================================================================================ */


__device__ static void x3_est_body(const X3EstArgs &a)
{
	const uint32_t c = blockIdx.x, lane = x3_lane();
	uint32_t y0, y1;
	float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
	if (a.seg_first) { /* a slice: the sums continue where the earlier slices left them */
		y0 = a.seg_first[c]; y1 = y0 + a.seg_count[c];
		const float *o = a.out + 4 * (size_t)c;
		acc0 = o[0]; acc1 = o[1]; acc2 = o[2]; acc3 = o[3];
	} else { y0 = a.range[c]; y1 = a.range[c + 1]; }
	for (uint32_t base = y0; base < y1; base += X3_WAVE) {
		const uint32_t y = base + lane;
		const uint32_t k = y < y1 ? a.cls[y] : X3_EST_NONE;
		const float v = k != X3_EST_NONE ? a.val[y] : 0.f;
		uint64_t m = x3_ballot(k != X3_EST_NONE);
		while (m) {
			const uint32_t l = (uint32_t)x3_ctz64(m);
			m &= m - 1;
			const uint32_t kl = x3_readlane_u32(k, l);
			const float vl = __uint_as_float(x3_readlane_u32(__float_as_uint(v), l));
			if (kl == 0) acc0 += vl; else if (kl == 1) acc1 += vl; else if (kl == 2) acc2 += vl; else acc3 += vl;
		}
	}
	if (lane == 0) { float *o = a.out + 4 * (size_t)c; o[0] = acc0; o[1] = acc1; o[2] = acc2; o[3] = acc3; }
}
#ifndef X3_EMU
__global__ void __launch_bounds__(X3_WAVE) x3_est_kernel(X3EstArgs a) { x3_est_body(a); }
static void launch_est(const X3EstArgs &a, uint32_t nchunks, hipStream_t st) { hipLaunchKernelGGL(x3_est_kernel, dim3(nchunks), dim3(X3_WAVE), 0, st, a); }
#else
static void est_tramp(void *p) { x3_est_body(*(const X3EstArgs *)p); }
static void launch_est(const X3EstArgs &a, uint32_t nchunks, hipStream_t) { x3emu_launch(est_tramp, (void *)&a, dim3(nchunks), dim3(X3_WAVE)); }
#endif

int x3s_est_launch(const X3EstArgs &a, uint32_t nc, hipStream_t st) { launch_est(a, nc, st); HIPCHK(hipGetLastError()); return X3H_OK; }
int x3s_emit_launch(const X3EmitArgs &a, uint32_t nc, hipStream_t st) { launch_emit(a, nc, st, true); HIPCHK(hipGetLastError()); return X3H_OK; }
int x3s_ac2_launch(const uint4 *sym, uint32_t *states, uint32_t *final_lo, const uint32_t *seg_off, const uint32_t *seg_len, uint32_t *seg_state, uint32_t nc, hipStream_t st)
{
	X3Ac2Args aa;
	aa.yo = nullptr; aa.sym = sym; aa.rec_nk = states; aa.final_lo = final_lo; aa.seg_off = seg_off; aa.seg_len = seg_len; aa.seg_state = seg_state; aa.compact = 0;
	launch_ac2(aa, nc, st);
	HIPCHK(hipGetLastError());
	return X3H_OK;
}

/* ============================================================================================================
 * K2 post-pass: the parse walker only emits one word per step (tag or fragment length); positions and the running
 * counts the coding stage indexes with are prefix sums over that list.
 *   tok_pos = sum of earlier step lengths (dict_len of the tag / fragment length)      (x3.c:394,422: p += len)
 *   tok_hb  = earlier hits, tok_nb = earlier inserted elements, tok_mb = earlier new-fragment bytes
 * ============================================================================================================ */
int x3_token_postpass(X3Code2Bufs &B, hipStream_t st, int nchunks, const X3Chunk *h_chunks, const X3Chunk *d_chunks,
                      const X3ParseResult *d_parsed, const uint32_t *tok_info, const uint8_t *dict_len,
                      uint32_t *tok_pos, uint32_t *tok_hb, uint32_t *tok_nb, uint32_t *tok_mb, size_t prefix_tokens, bool rebase)
{
	const uint32_t nc = (uint32_t)nchunks;
	std::vector<uint32_t> eo(nc + 1);
	uint64_t tot = 0;
	for (uint32_t c = 0; c < nc; c++) { eo[c] = (uint32_t)h_chunks[c].elem_off; tot = h_chunks[c].elem_off + h_chunks[c].len + 16; }
	if (tot >= (1ull << 31)) return X3H_E_ARG;
	eo[nc] = (uint32_t)tot;
	const size_t n = (prefix_tokens && nc == 1 && prefix_tokens + 4 < tot) ? prefix_tokens + 4 : (size_t)tot; /* one stream, tokens parsed so far */
	CHK(B.offs.reserve((size_t)(nc + 1) * 9 * 4));
	uint32_t *d_eo = B.offs.as<uint32_t>() + (size_t)(nc + 1) * 7;
	HIPCHK(hipMemcpyAsync(d_eo, eo.data(), (nc + 1) * 4, hipMemcpyHostToDevice, st));
	for (int i = 0; i < 4; i++) CHK(B.pp[i].reserve(((size_t)tot + 4) * 4)); /* full size even for a prefix: no reallocation between prefix calls */
	uint32_t *vh = B.pp[0].as<uint32_t>(), *vn = B.pp[1].as<uint32_t>(), *vm = B.pp[2].as<uint32_t>(), *vl = B.pp[3].as<uint32_t>();
	x3_foreach(n, st, X3_LAMBDA(size_t i) {
		const uint32_t c = find_chunk(d_eo, nc, (uint32_t)i);
		const uint32_t k = (uint32_t)i - d_eo[c];
		uint32_t h = 0, nw = 0, mb = 0, ln = 0;
		if (k < d_parsed[c].ntok) {
			const uint32_t info = tok_info[i];
			if (!(info & X3_TOK_MISS)) { h = 1; ln = dict_len[d_chunks[c].elem_off + info]; }
			else { ln = mb = info & 0x3Fu; nw = (info & X3_TOK_DUP) ? 0u : 1u; }
		}
		vh[i] = h; vn[i] = nw; vm[i] = mb; vl[i] = ln;
	});
	CHK(x3p_excl_scan(B.tmp, vh, tok_hb, n, st));
	CHK(x3p_excl_scan(B.tmp, vn, tok_nb, n, st));
	CHK(x3p_excl_scan(B.tmp, vm, tok_mb, n, st));
	CHK(x3p_excl_scan(B.tmp, vl, tok_pos, n, st));
	/* make them stream-relative (every stream restarts at 0); in place: element i only needs the value at its chunk start,
	 * which is read before any element of that chunk is rewritten only if the start itself goes last -> use a snapshot */
	uint32_t *base = vh; /* vh..vl are dead: 4 words per chunk */
	x3_foreach(nc, st, X3_LAMBDA(size_t c) {
		const uint32_t s = d_eo[c];
		base[4 * c] = tok_hb[s]; base[4 * c + 1] = tok_nb[s]; base[4 * c + 2] = tok_mb[s]; base[4 * c + 3] = tok_pos[s];
	});
	if (rebase) x3_foreach(n, st, X3_LAMBDA(size_t i) {
		const uint32_t c = find_chunk(d_eo, nc, (uint32_t)i);
		tok_hb[i] -= base[4 * c]; tok_nb[i] -= base[4 * c + 1]; tok_mb[i] -= base[4 * c + 2]; tok_pos[i] -= base[4 * c + 3];
	});
	return X3H_OK;
}

/* the streams are assembled with ORs: zero every chunk's slot (one launch for the batch; out_off and out_cap are multiples of 4) */
static int zero_output_slots(hipStream_t st, uint32_t nc, const X3Chunk *h_chunks, const X3Chunk *d_chunks, uint8_t *d_out)
{
	uint64_t maxcap = 0;
	for (uint32_t c = 0; c < nc; c++) if (h_chunks[c].out_cap > maxcap) maxcap = h_chunks[c].out_cap;
	const uint64_t per = (maxcap + 15) / 16;
	x3_foreach((size_t)(per * nc), st, X3_LAMBDA(size_t i) {
		const uint32_t c = (uint32_t)(i / per);
		const uint64_t o = (uint64_t)(i % per) * 16, cap = d_chunks[c].out_cap;
		if (o >= cap) return;
		uint32_t *w = (uint32_t *)(d_out + d_chunks[c].out_off + o);
		const uint32_t nw = cap - o >= 16 ? 4u : (uint32_t)((cap - o) / 4);
		for (uint32_t k = 0; k < nw; k++) w[k] = 0;
	});
	HIPCHK(hipGetLastError());
	return X3H_OK;
}

int x3_zero_output_slots(hipStream_t st, uint32_t nc, const X3Chunk *h_chunks, const X3Chunk *d_chunks, uint8_t *d_out) { return zero_output_slots(st, nc, h_chunks, d_chunks, d_out); }

/* ============================================================================================================ */
int x3_code_v2_run(X3Code2Bufs &B, hipStream_t st, int nchunks, const X3Chunk *h_chunks, const X3Chunk *d_chunks,
                   const X3ParseResult *h_parsed, const X3ParseResult *d_parsed,
                   const uint8_t *d_bytes, const uint32_t *tok_pos, const uint32_t *tok_info, const uint32_t *tok_hb,
                   const uint32_t *tok_nb, const uint32_t *tok_mb, uint8_t *d_out, X3CodeResult *d_result, X3CodeSeg *seg, const uint8_t *dict_len)
{
	const uint32_t nc = (uint32_t)nchunks;
	const bool final = !seg || seg->final;
	const uint32_t *tokb = nullptr; /* per stream: the token prefix sums {hits, elements, new-fragment bytes, position} at its first token (x3_token_postpass) */
	for (int i = 0; i < 5; i++) if (!B.ev[i]) HIPCHK(hipEventCreate(&B.ev[i]));
	HIPCHK(hipEventRecord(B.ev[0], st));
	/* ---- index spaces: steps, hits, MTF events (hits + inserted elements), tags ---- */
	std::vector<uint32_t> so(nc + 1), ho(nc + 1), eo(nc + 1), dof(nc + 1), mo(nc + 1), bo(nc + 1), yo(nc + 1);
	uint64_t s = 0, h = 0, e = 0, d = 0, mi = 0, by = 0, y = 0;
	for (uint32_t c = 0; c < nc; c++) {
		so[c] = (uint32_t)s; ho[c] = (uint32_t)h; eo[c] = (uint32_t)e; dof[c] = (uint32_t)d; mo[c] = (uint32_t)mi; bo[c] = (uint32_t)by; yo[c] = (uint32_t)y;
		s += h_parsed[c].ntok; h += h_parsed[c].hits; e += (uint64_t)h_parsed[c].hits + h_parsed[c].dict_elems; d += h_parsed[c].dict_elems;
		mi += h_parsed[c].ntok - h_parsed[c].hits; by += h_parsed[c].miss_bytes;
		y += 2ull * h_parsed[c].ntok + h_parsed[c].miss_bytes + (final ? 1 : 0); /* two symbols per step, one per new-fragment byte, E_EOF */
	}
	if (s >= (1ull << 31) || e >= (1ull << 31) || y >= (1ull << 31)) return X3H_E_ARG; /* one batch: < 2^31 coded symbols */
	so[nc] = (uint32_t)s; ho[nc] = (uint32_t)h; eo[nc] = (uint32_t)e; dof[nc] = (uint32_t)d; mo[nc] = (uint32_t)mi; bo[nc] = (uint32_t)by; yo[nc] = (uint32_t)y;
	const size_t nS = s, nH = h, nE = e, nD = d, nM = mi, nB = by, nY = y;
	size_t nA = nE > nH ? nE : nH;
	if (nM > nA) nA = nM;
	if (nB > nA) nA = nB;
	size_t nDres = nD, nYres = nY, nMSres = (nM > nB ? nM : nB);
	if (seg) { /* a growing prefix: size everything for the estimated whole stream (a reallocation would wait for the running parse / coder) */
		/* (hits + elements <= steps: every element was a miss step) */
		if (seg->res_mbytes > nA) nA = seg->res_mbytes;
		if (seg->res_steps > nA) nA = seg->res_steps;
		if (seg->res_elems > nDres) nDres = seg->res_elems;
		nYres = 3 * seg->res_bytes + 8 + (size_t)nc * 8 * (X3_MAX_CKPT + 3); /* the operand / state rings persist from call to call: hard upper bound (+ the < 8 symbols per stream and call that are put twice) */
		if (seg->res_mbytes > nMSres) nMSres = seg->res_mbytes;
		if (seg->res_steps > nMSres) nMSres = seg->res_steps;
		CHK(B.tmp.reserve(24 * nA + ((size_t)8 << 20)));
	}
	nA += 4;

	CHK(B.offs.reserve((size_t)(nc + 1) * 9 * 4));
	uint32_t *d_so = B.offs.as<uint32_t>(), *d_ho = d_so + (nc + 1), *d_eo = d_ho + (nc + 1), *d_dof = d_eo + (nc + 1);
	uint32_t *d_mo = d_dof + (nc + 1), *d_bo = d_mo + (nc + 1), *d_yo = d_bo + (nc + 1);
	uint32_t *d_yoc = d_so + (size_t)(nc + 1) * 8; /* symbol offsets after the no-op symbols are dropped (slot 7 belongs to the token post-pass) */
	HIPCHK(hipMemcpyAsync(d_mo, mo.data(), (nc + 1) * 4, hipMemcpyHostToDevice, st));
	HIPCHK(hipMemcpyAsync(d_bo, bo.data(), (nc + 1) * 4, hipMemcpyHostToDevice, st));
	HIPCHK(hipMemcpyAsync(d_yo, yo.data(), (nc + 1) * 4, hipMemcpyHostToDevice, st));
	HIPCHK(hipMemcpyAsync(d_so, so.data(), (nc + 1) * 4, hipMemcpyHostToDevice, st));
	HIPCHK(hipMemcpyAsync(d_ho, ho.data(), (nc + 1) * 4, hipMemcpyHostToDevice, st));
	HIPCHK(hipMemcpyAsync(d_eo, eo.data(), (nc + 1) * 4, hipMemcpyHostToDevice, st));
	HIPCHK(hipMemcpyAsync(d_dof, dof.data(), (nc + 1) * 4, hipMemcpyHostToDevice, st));
	CHK(B.chunkmeta.reserve((size_t)nc * 10 * 4 + 64));
	uint32_t *m_pairbase = B.chunkmeta.as<uint32_t>(), *m_npairs = m_pairbase + nc, *m_first00 = m_npairs + nc, *m_ord00 = m_first00 + nc;
	CHK(B.maxred.reserve(64));
	CHK(B.idxfreq.reserve((nDres + 4) * 4));
	CHK(B.csbsmall.reserve(((size_t)3 * ((nA + 1) / X3_CSB_BLK + 2) + 8) * 2 * 4)); /* block totals of csb_run, sized once for every call */
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
	uint32_t *m_evfinal = m_ord00 + nc, *m_finallo = m_evfinal + 4 * nc, *m_nnoop = m_finallo + nc; /* per chunk: 4 + 1 + 1 words */

	/* A batch of many streams whose dictionaries fit the LDS tables gets the per-stream kernels of code3.hip (one wavefront per stream,
	 * tables in LDS) instead of chip-wide sorts and partitions; X3H_STREAM_KERNELS=0/1 forces either form (both give the same bytes). */
	uint64_t maxDict = 1;
	for (uint32_t c = 0; c < nc; c++) if (h_parsed[c].dict_elems > maxDict) maxDict = h_parsed[c].dict_elems;
	bool streamk = !seg && nc >= X3_STREAM_MIN_STREAMS;
	if (const char *e = getenv("X3H_STREAM_KERNELS")) streamk = !seg && e[0] == '1';
	streamk = streamk && x3_stream_kernels_fit(maxDict);
	if (dict_len && (!streamk || nH == 0)) /* the caller left the token prefix sums to this call and the per-stream walk does not apply */
		CHK(x3_token_postpass(B, st, nchunks, h_chunks, d_chunks, d_parsed, tok_info, dict_len, (uint32_t *)tok_pos, (uint32_t *)tok_hb, (uint32_t *)tok_nb, (uint32_t *)tok_mb));
	CHK(B.pp[0].reserve((size_t)nc * 16 + 64));
	tokb = B.pp[0].as<uint32_t>();

	/* new fragments: model_match_size (32 symbols) and model_chars (256 symbols) are adaptive order-0 models (x3.c:259-267); their inputs, one length symbol per
	 * fragment and the fragments' bytes in order, are written by the per-stream token walk where that runs, else by an element-wise pass over the steps (below) */
	const size_t nMS = nMSres + 4;
	uint32_t *Q[16];
	for (int i = 0; i < 16; i++) { CHK(B.ms[i].reserve(nMS * 4)); Q[i] = B.ms[i].as<uint32_t>(); }
	uint32_t *lval = Q[0], *lsm = Q[1], *leq = Q[2], *bval = Q[3], *bsm = Q[4], *beq = Q[5];
	bool frag_inputs_done = false;
	if (nH > 0) {
		if (streamk && dict_len) {
			/* many streams: one workgroup per stream walks its tokens: running counts (stream-relative) + per-hit / per-touch records */
			HIPCHK(hipMemsetAsync((void *)tokb, 0, (size_t)nc * 16, st));
			CHK(x3_tokens_run(st, nc, d_chunks, d_parsed, tok_info, dict_len, (uint32_t *)tok_pos, (uint32_t *)tok_hb, (uint32_t *)tok_nb, (uint32_t *)tok_mb,
			                  d_ho, d_eo, d_dof, h_tag, h_c1, h_pv, h_dk, h_step, e_tag, e_hit, d_bytes, d_mo, d_bo, lval, bval));
			frag_inputs_done = true;
		} else {
			/* ---- F1: per step -> per hit / per event records ---- */
			x3_foreach(nS, st, X3_LAMBDA(size_t gs) {
				const uint32_t c = find_chunk(d_so, nc, (uint32_t)gs);
				const uint32_t k = (uint32_t)gs - d_so[c];
				const uint64_t base = d_chunks[c].elem_off;
				const uint32_t info = tok_info[base + k], hb = tok_hb[base + k] - tokb[4 * c], nb = tok_nb[base + k] - tokb[4 * c + 1];
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

		}

		/* ---- MTF rank ---- */
		if (streamk) {
			/* many streams: one wavefront per stream replays the list in LDS, 64 events per trip (code3.hip) */
			if (!B.side) { HIPCHK(hipStreamCreate(&B.side)); HIPCHK(hipEventCreate(&B.ev_fork)); HIPCHK(hipEventCreate(&B.ev_join)); }
			HIPCHK(hipEventRecord(B.ev_fork, st));
			HIPCHK(hipStreamWaitEvent(B.side, B.ev_fork, 0));
			CHK(x3_mtf_ranks_run(B.side, nc, maxDict, d_eo, d_dof, e_tag, e_hit, h_rank)); /* beside the context statistics: both are latency-bound */
			HIPCHK(hipEventRecord(B.ev_join, B.side));
		} else
		/* touches sorted by tag give prev(e); rank = CSB(prev+1) - (prev_local+1) */
		{
			uint32_t *iota = T[0], *ks = T[1], *vs = T[2], *key = T[3], *keyc = T[4], *bs = T[5], *be = T[6], *cnt = T[7];
			x3_foreach(nE, st, X3_LAMBDA(size_t i) { iota[i] = (uint32_t)i; });
			CHK(x3p_sort_pairs(B.tmp, e_tag, ks, iota, vs, nE, bits_for(nD), st));
			x3_foreach(nE, st, X3_LAMBDA(size_t i) {
				const uint32_t ev = vs[i];
				const uint32_t kv = (i > 0 && ks[i - 1] == ks[i]) ? vs[i - 1] + 1 : 0u;
				key[ev] = kv;
			});
			/* keys relative to the stream (0 = first touch, else previous touch + 1): the bit depth follows the largest stream, not the batch */
			x3_foreach(nE, st, X3_LAMBDA(size_t i) {
				const uint32_t c = find_chunk(d_eo, nc, (uint32_t)i);
				const uint32_t kl = key[i] ? key[i] - d_eo[c] : 0u;
				key[i] = kl; keyc[i] = kl; bs[i] = d_eo[c]; be[i] = d_eo[c + 1];
			});
			uint64_t maxE = 1;
			for (uint32_t c = 0; c < nc; c++) { const uint64_t ec = (uint64_t)h_parsed[c].hits + h_parsed[c].dict_elems; if (ec > maxE) maxE = ec; }
			CHK(csb_run(B, st, nE, bits_for(maxE + 1), keyc, bs, be, cnt, T + 8));
			x3_foreach(nE, st, X3_LAMBDA(size_t i) {
				const uint32_t gh = e_hit[i];
				if (gh != NONE32) h_rank[gh] = cnt[i] - key[i]; /* #{earlier touches with an older previous touch} - (previous touch + 1) */
			});
		}

		if (streamk) {
			/* ---- many streams: context statistics by one wavefront per stream (code3.hip); the context1 pass also finds the first use
			 *      of every (context1, tag), which IS the tag-pair map (tag_pair.c:100-130: ordinal = rank of the first occurrence) ---- */
			uint32_t *iota = T[0], *kA = T[1], *vA = T[2], *tA = T[3], *P = T[6];
			CHK(B.stat.reserve((nA + 4) * 16));
			CHK(B.stat0.reserve((nA + 4) * 16));
			uint4 *stat = B.stat.as<uint4>(), *stat0 = B.stat0.as<uint4>(); /* per hit {freq, total, cum, first}: context1 / context0.  Read in place by the mode kernel and the symbol
			                                                                  * selection (no unpacking into six arrays: 60 bytes per hit less traffic) */
			uint32_t npairs_total = 0;
			/* by context1, time order inside: chip-wide stable radix sort (prims.hip; rounds 1-4: rocPRIM onesweep, 3.6 TB/s).  X3H_ARRANGE=1: one workgroup per stream, counting
			 * sort on the stream-local key (x3_arrange_kernel, code3.hip) -- same order, measured SLOWER on the 1024-chunk batch (features 48 against
			 * 40 ms: four wavefronts per stream walk their quarter as a chain of LDS round trips), so it stays an option, not the default */
			bool seg_arrange = false;
			if (const char *e = getenv("X3H_ARRANGE")) seg_arrange = e[0] == '1';
			bool ctx_gather = true; /* the context kernel fetches the tags of its arrangement itself (X3H_CTX_GATHER=0: an element-wise pass writes them out first) */
			if (const char *e = getenv("X3H_CTX_GATHER")) ctx_gather = e[0] != '0';
			/* default from X3_SEGSORT_MIN_STREAMS streams on: one workgroup per stream with the tile machinery of scan3.hip (x3_segsort_kernel, code3.hip), as many
			 * 8-bit passes as the stream-local keys need -- the same order; X3H_SEGSORT=0: the chip-wide sort */
			bool seg_sort = !seg_arrange && nc >= X3_SEGSORT_MIN_STREAMS;
			if (const char *e = getenv("X3H_SEGSORT")) seg_sort = !seg_arrange && e[0] != '0';
			if (seg_arrange) CHK(x3_arrange_run(st, nc, d_ho, d_dof, maxDict, h_c1, h_tag, kA, vA, tA, T[7], T[8]));
			else if (seg_sort && maxDict <= X3_SEGSORT_MAX_LOCAL) {
				CHK(x3_segsort_run(st, nc, d_ho, d_dof, maxDict, h_c1, kA, vA, T[7], T[8]));
				if (!ctx_gather) x3_foreach(nH, st, X3_LAMBDA(size_t i) { tA[i] = h_tag[vA[i]]; });
			} else {
				x3_foreach(nH, st, X3_LAMBDA(size_t i) { iota[i] = (uint32_t)i; });
				CHK(x3p_sort_pairs(B.tmp, h_c1, kA, iota, vA, nH, bits_for(nD), st));
				if (!ctx_gather) x3_foreach(nH, st, X3_LAMBDA(size_t i) { tA[i] = h_tag[vA[i]]; });
			}
			HIPCHK(hipMemsetAsync(m_first00, 0xFF, (size_t)nc * 4, st)); /* NONE32: the stream never uses the pair (0, 0) */
			CHK(x3_ctx_stats_run(st, nc, maxDict, nH, d_ho, d_dof, kA, vA, ctx_gather && !seg_arrange ? nullptr : tA, h_tag, stat, m_first00));
			/* P[i] = hits before i that put their (context1, tag) pair into the map -- the scan reads the flag out of the records (no flag array) */
			CHK(x3p_excl_scan_top_bit_w(B.tmp, stat, P, nH, st));
			HIPCHK(hipMemcpyAsync(&npairs_total, P + nH, 4, hipMemcpyDeviceToHost, st));
			x3_foreach(nc, st, X3_LAMBDA(size_t c) {
				m_pairbase[c] = P[d_ho[c]];
				m_npairs[c] = P[d_ho[c + 1]] - P[d_ho[c]];
				m_ord00[c] = m_first00[c] != NONE32 ? P[m_first00[c]] : 0u; /* the pair (0,0) of the stream: both contexts after a new fragment (x3.c:424-425); found by the context kernel */
			});
			/* the context0 group of every hit: by this element-wise pass.  X3H_SEGSORT_GEN=1: made by the per-stream sort itself (X3SegSortArgs::gen) -- its scattered reads
			 * of P then stay in one XCD's L2, but a lone workgroup per stream waits for each of them: features 21.8 -> 23.6 ms on the 1024-stream batch, so not the default */
			bool gen_in_sort = false;
			if (const char *e = getenv("X3H_SEGSORT_GEN")) gen_in_sort = seg_sort && e[0] == '1';
			auto groups_elementwise = [&]() {
				x3_foreach(nH, st, X3_LAMBDA(size_t gh) {
					uint32_t g;
					if (h_pv[gh]) g = P[stat[gh - 1].w & 0x7FFFFFFFu]; /* (prev_context1, context1) is the pair the previous hit registered: its ordinal = rank of the hit that first used it */
					else {
						const uint32_t c = find_chunk(d_ho, nc, (uint32_t)gh);
						g = (m_first00[c] != NONE32 && m_first00[c] < gh) ? m_ord00[c] : m_pairbase[c]; /* unknown pair -> context 0 */
					}
					G0[gh] = g;
				});
			};
			if (!gen_in_sort) groups_elementwise();
			std::vector<uint32_t> hnp(nc);
			HIPCHK(hipMemcpyAsync(hnp.data(), m_npairs, (size_t)nc * 4, hipMemcpyDeviceToHost, st));
			HIPCHK(hipStreamSynchronize(st)); /* npairs_total, pairs per stream */
			uint64_t maxPairs = 1;
			for (uint32_t c = 0; c < nc; c++) if (hnp[c] > maxPairs) maxPairs = hnp[c];
			/* by ctx0 (pair ordinal; a stream's ordinals start at m_pairbase): the same segmented counting sort, two passes beyond 2048 pairs */
			if (seg_arrange && maxPairs <= X3_ARRANGE_MAX_LOCAL) CHK(x3_arrange_run(st, nc, d_ho, m_pairbase, maxPairs, G0, h_tag, kA, vA, tA, T[7], T[8]));
			else if (seg_sort && maxPairs <= X3_SEGSORT_MAX_LOCAL) {
				const X3SegSortGen gen = { G0, h_pv, P, m_first00, m_ord00, stat };
				CHK(x3_segsort_run(st, nc, d_ho, m_pairbase, maxPairs, G0, kA, vA, T[7], T[8], gen_in_sort ? &gen : nullptr));
				if (!ctx_gather) x3_foreach(nH, st, X3_LAMBDA(size_t i) { tA[i] = h_tag[vA[i]]; });
			} else {
				if (gen_in_sort) groups_elementwise(); /* (a stream with more pairs than the per-stream sort takes) */
				if (seg_arrange || seg_sort) x3_foreach(nH, st, X3_LAMBDA(size_t i) { iota[i] = (uint32_t)i; });
				CHK(x3p_sort_pairs(B.tmp, G0, kA, iota, vA, nH, bits_for(npairs_total ? npairs_total : 1), st));
				if (!ctx_gather) x3_foreach(nH, st, X3_LAMBDA(size_t i) { tA[i] = h_tag[vA[i]]; });
			}
			CHK(x3_ctx_stats_run(st, nc, maxDict, nH, d_ho, d_dof, kA, vA, ctx_gather && !(seg_arrange && maxPairs <= X3_ARRANGE_MAX_LOCAL) ? nullptr : tA, h_tag, stat0, nullptr));
		} else {
		/* ---- tag-pair ordinals (tag_pair.c) and the ctx0 group of every hit (x3.c:139-147) ---- */
			uint32_t npairs_total = 0;
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
				HIPCHK(hipMemcpyAsync(&npairs_total, P + nH, 4, hipMemcpyDeviceToHost, st)); /* read after the next synchronisation point below */
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
				HIPCHK(hipStreamSynchronize(st)); /* npairs_total */
			}
	
			/* ---- context statistics ---- */
			CHK(ctx_stats(B, st, nH, bits_for(npairs_total ? npairs_total : 1), bits_for(nD), G0, h_tag, f0, t0, c0, T)); /* ctx0 groups are pair ordinals: sort only as many bits as there are pairs */
			CHK(ctx_stats(B, st, nH, bits_for(nD), bits_for(nD), h_c1, h_tag, f1, t1, c1, T));
		}
	}

	if (streamk && nH > 0) HIPCHK(hipStreamWaitEvent(st, B.ev_join, 0)); /* the move-to-front ranks */

	/* ---- serial pass 1: modes ---- */
	uint32_t *idxf = B.idxfreq.as<uint32_t>();
	x3_foreach(nD + 1, st, X3_LAMBDA(size_t i) { idxf[i] = 1; });
	HIPCHK(hipEventRecord(B.ev[1], st));
	if (nH > 0) {
		uint64_t maxH = 0, maxDk = 1;
		for (uint32_t c = 0; c < nc; c++) { if (h_parsed[c].hits > maxH) maxH = h_parsed[c].hits; if (h_parsed[c].dict_elems > maxDk) maxDk = h_parsed[c].dict_elems; }
		/* the serial kernel costs ~11.5 ns per hit of the longest stream (streams run side by side); an iteration of the fixed-point
		 * form streams ~60 B per hit of the whole batch and a dozen of them are typical: pick the cheaper (X3H_MODES=serial|fixed overrides) */
		bool fixed = (double)maxH * 11.5e-9 > 2.0 * (double)nH * 0.45e-9 + 1e-3, done = false;
		if (const char *e = getenv("X3H_MODES")) fixed = e[0] == 'f';
		if (streamk) fixed = false; /* many streams: the serial kernel, which then also hands out the model state of every coded symbol */
		B.last.mode_iters = 0;
		if (seg && (fixed || streamk)) seg->prev_serial = false; /* (the saved chain state, if any, is stale after a call that did not use the serial kernel) */
		if (fixed) {
			int iters = 0;
			CHK(modes_fixed_point(B, st, nH, nc, d_ho, bits_for(maxDk), f0, t0, f1, t1, h_rank, h_dk, h_step, mode, T, 256, &iters, &done));
			B.last.mode_iters = done ? iters : -iters;
		}
		if (!done) {
			X3ModesArgs ma;
			ma.parsed = d_parsed; ma.ho = d_ho; ma.dof = d_dof;
			ma.f0 = f0; ma.t0 = t0; ma.f1 = f1; ma.t1 = t1; ma.fs = 1; ma.rank = h_rank; ma.dk = h_dk; ma.step = h_step;
			if (streamk) { /* the context kernel's records, read in place: {freq, total, cum, first} per hit */
				const uint32_t *r0 = (const uint32_t *)B.stat0.p, *r1 = (const uint32_t *)B.stat.p;
				ma.f0 = r0; ma.t0 = r0 + 1; ma.f1 = r1; ma.t1 = r1 + 1; ma.fs = 4;
			}
			ma.idxfreq = idxf; ma.mode = mode;
			ma.pe0 = ma.pe1 = ma.ilist_rank = ma.ilist_hit = ma.evfinal = nullptr; ma.nzl = ma.nnoop = nullptr;
			ma.state = nullptr; ma.resume = 0; ma.slice = nullptr; ma.chunks = nullptr;
			if (seg) {
				/* growing prefixes: the serial chain continues where the previous call stopped.  The hit arrays are laid out by the CURRENT
				 * per-stream counts, so the modes decided earlier are moved to their new places first. */
				CHK(seg->modes_state.reserve((size_t)nc * X3_MODES_STATE_STRIDE * 4));
				ma.state = seg->modes_state.as<uint32_t>();
				if (seg->prev_ho.size() == nc + 1 && seg->prev_serial) {
					const size_t nprev = seg->prev_ho[nc];
					if (nprev) {
						CHK(seg->mode_prev.reserve((nprev + 4) * 4 + (size_t)(nc + 1) * 4));
						uint32_t *old = seg->mode_prev.as<uint32_t>(), *d_pho = old + nprev + 2;
						HIPCHK(hipMemcpyAsync(old, mode, nprev * 4, hipMemcpyDeviceToDevice, st));
						HIPCHK(hipMemcpyAsync(d_pho, seg->prev_ho.data(), (nc + 1) * 4, hipMemcpyHostToDevice, st));
						x3_foreach(nprev, st, X3_LAMBDA(size_t i) { const uint32_t c = find_chunk(d_pho, nc, (uint32_t)i); mode[d_ho[c] + ((uint32_t)i - d_pho[c])] = old[i]; });
					}
					ma.resume = 1;
				} else {
					HIPCHK(hipMemsetAsync(ma.state, 0, (size_t)nc * X3_MODES_STATE_STRIDE * 4, st));
				}
				seg->prev_ho = ho; seg->prev_serial = true;
			}
			if (streamk) { ma.pe0 = T[26]; ma.pe1 = T[27]; ma.ilist_rank = T[2]; ma.ilist_hit = T[3]; ma.evfinal = m_evfinal; ma.nzl = T[4]; ma.nnoop = m_nnoop; }
			launch_modes(ma, nc, st, maxDict);
			/* model_index1 as the IDX1-coded hits saw it (x3.c:187-188): one wavefront per stream, the table and its running sums in LDS */
			if (streamk) CHK(x3_idxstat_run(st, nc, maxDict, d_ho, m_evfinal, T[2], T[3], h_dk, rfreq, rcum, itot));
			HIPCHK(hipGetLastError());
		}
		HIPCHK(hipEventRecord(B.ev[2], st));

		if (!streamk) {
			/* ---- model_events / model_index1 state at every hit, recovered from the modes by prefix sums ---- */
			uint32_t *zi = T[0], *ci = T[1], *key = T[2], *org = T[3], *bs = T[4], *be = T[5], *cnt = T[6], *kin2 = T[7];
			uint32_t *z0 = T[17], *c0s = T[18], *z1 = T[19], *c1s = T[20], *cid = T[21], *s1k = T[22], *s1v = T[23], *s2k = T[24], *s2v = T[25];
			uint32_t *pe0w = T[26], *pe1w = T[27];
			x3_foreach(nH, st, X3_LAMBDA(size_t i) {
				const uint32_t m = mode[i];
				z0[i] = m == X3_E_CTX0 ? 1u : 0u; z1[i] = m == X3_E_CTX1 ? 1u : 0u; zi[i] = m == X3_E_IDX1 ? 1u : 0u;
			});
			CHK(x3p_excl_scan(B.tmp, z0, c0s, nH, st));
			CHK(x3p_excl_scan(B.tmp, z1, c1s, nH, st));
			CHK(x3p_excl_scan(B.tmp, zi, ci, nH, st));
			uint32_t nI = 0;
			HIPCHK(hipMemcpyAsync(&nI, ci + nH, 4, hipMemcpyDeviceToHost, st));
			x3_foreach(nH, st, X3_LAMBDA(size_t i) {
				const uint32_t c = find_chunk(d_ho, nc, (uint32_t)i), lo = d_ho[c];
				pe0w[i] = 1024u + c0s[i] - c0s[lo];          /* model_events freq of E_CTX0 before this hit */
				pe1w[i] = 1024u + c1s[i] - c1s[lo];
				itot[i] = h_dk[i] + ci[i] - ci[lo];          /* model_index1.total: one per element + one per earlier IDX1 use */
				if (mode[i] == X3_E_IDX1) {
					const uint32_t j = ci[i];
					key[j] = h_rank[i]; org[j] = (uint32_t)i; bs[j] = ci[lo]; be[j] = ci[d_ho[c + 1]]; cid[j] = c;
				}
			});
			x3_foreach(nc, st, X3_LAMBDA(size_t c) {
				const uint32_t lo = d_ho[c], hi = d_ho[c + 1];
				m_evfinal[4 * c + 0] = 1024u + c0s[hi] - c0s[lo];
				m_evfinal[4 * c + 1] = 1024u + c1s[hi] - c1s[lo];
				m_evfinal[4 * c + 2] = 1u + ci[hi] - ci[lo];
				m_evfinal[4 * c + 3] = 0;
			});
			HIPCHK(hipStreamSynchronize(st));
			uint64_t maxD = 1;
			for (uint32_t c = 0; c < nc; c++) if (h_parsed[c].dict_elems > maxD) maxD = h_parsed[c].dict_elems;
			if (nI) {
				/* freq of the coded rank = 1 + earlier IDX1 hits of the stream with the same rank: runs of (stream, rank) */
				CHK(x3p_sort_pairs(B.tmp, key, s1k, org, s1v, nI, bits_for(maxD), st));
				x3_foreach(nI, st, X3_LAMBDA(size_t j) { kin2[j] = find_chunk(d_ho, nc, s1v[j]); });
				CHK(x3p_sort_pairs(B.tmp, kin2, s2k, s1v, s2v, nI, bits_for(nc), st));
				x3_foreach(nI, st, X3_LAMBDA(size_t j) {
					const bool start = j == 0 || s2k[j - 1] != s2k[j] || h_rank[s2v[j - 1]] != h_rank[s2v[j]];
					z0[j] = start ? (uint32_t)j : 0u;
				});
				CHK(x3p_incl_max_scan(B.tmp, z0, z1, nI, st));
				x3_foreach(nI, st, X3_LAMBDA(size_t j) { rfreq[s2v[j]] = 1u + (uint32_t)j - z1[j]; });
				/* cum_freq of the coded rank = rank + earlier IDX1 hits of the stream with a smaller rank */
				CHK(csb_run(B, st, nI, bits_for(maxD), key, bs, be, cnt, T + 8));
				x3_foreach(nI, st, X3_LAMBDA(size_t j) { rcum[org[j]] = h_rank[org[j]] + cnt[j]; });
			}
			(void)cid;

		}

		/* ---- the tag / index symbol of every hit (many streams: picked by the assembly pass below straight from the context kernel's records) ---- */
		uint32_t *scum = T[0], *sfreq = T[1], *stot = T[2]; /* zi/ci/key are dead now */
		if (!streamk) x3_foreach(nH, st, X3_LAMBDA(size_t i) {
			const uint32_t m = mode[i];
			uint32_t cu, fq, to;
			if (m == X3_E_CTX0) { cu = c0[i]; fq = f0[i]; to = t0[i]; }
			else if (m == X3_E_CTX1) { cu = c1[i]; fq = f1[i]; to = t1[i]; }
			else { cu = rcum[i]; fq = rfreq[i]; to = itot[i]; }
			scum[i] = cu; sfreq[i] = fq; stot[i] = to;
		});
	} else {
		x3_foreach(nc, st, X3_LAMBDA(size_t c) { m_npairs[c] = 0; m_evfinal[4 * c] = 1024; m_evfinal[4 * c + 1] = 1024; m_evfinal[4 * c + 2] = 1; m_evfinal[4 * c + 3] = 0; });
	}
	const uint32_t *hs_cum = T[0], *hs_freq = T[1], *hs_tot = T[2], *pe0 = T[26], *pe1 = T[27];
	const bool inplace = streamk && nH > 0; /* many streams: symbols picked from the context kernel's records, no-op counts from the mode kernel */
	const uint4 *recs0 = inplace ? B.stat0.as<uint4>() : nullptr, *recs1 = inplace ? B.stat.as<uint4>() : nullptr;

	/* ---- new fragments: model_match_size (32 symbols) and model_chars (256 symbols) are adaptive order-0 models
	 *      (x3.c:259-267): cum_freq = symbol + #{earlier smaller}, freq = 1 + #{earlier equal}, total = alphabet + index ---- */
	if (!frag_inputs_done) x3_foreach(nS, st, X3_LAMBDA(size_t gs) {
		const uint32_t c = find_chunk(d_so, nc, (uint32_t)gs);
		const uint32_t k = (uint32_t)gs - d_so[c];
		const uint64_t base = d_chunks[c].elem_off;
		const uint32_t info = tok_info[base + k];
		if (info & X3_TOK_MISS) {
			const uint32_t len = info & 0x3Fu, hb = tok_hb[base + k] - tokb[4 * c], mb = tok_mb[base + k] - tokb[4 * c + 2];
			lval[d_mo[c] + (k - hb)] = len - 1;
			const uint8_t *p = d_bytes + d_chunks[c].byte_off + (tok_pos[base + k] - tokb[4 * c + 3]);
			for (uint32_t j = 0; j < len; j++) bval[d_bo[c] + mb + j] = p[j];
		}
	});
	if (streamk) CHK(x3_order0_run(st, nc, d_mo, lval, lsm, leq, d_bo, bval, bsm, beq)); /* one wavefront per (stream, model), counters in LDS */
	else for (int which = 0; which < 2; which++) {
		const size_t n = which ? nB : nM;
		if (!n) continue;
		const int abits = which ? 8 : 5;
		const uint32_t *val = which ? bval : lval, *off = which ? d_bo : d_mo;
		uint32_t *osm = which ? bsm : lsm, *oeq = which ? beq : leq;
		uint32_t *iota = Q[6], *key = Q[7], *ks = Q[8], *vs = Q[9], *rsf = Q[10], *rs = Q[11], *bs = Q[12], *be = Q[13], *kc = Q[14];
		x3_foreach(n, st, X3_LAMBDA(size_t i) {
			const uint32_t c = find_chunk(off, nc, (uint32_t)i);
			iota[i] = (uint32_t)i;
			key[i] = val[i] | (c << abits);
			kc[i] = val[i]; bs[i] = off[c]; be[i] = off[c + 1];
		});
		CHK(x3p_sort_pairs(B.tmp, key, ks, iota, vs, n, abits + bits_for(nc), st));
		x3_foreach(n, st, X3_LAMBDA(size_t i) { rsf[i] = (i > 0 && ks[i - 1] != ks[i]) ? (uint32_t)i : 0u; });
		CHK(x3p_incl_max_scan(B.tmp, rsf, rs, n, st));
		x3_foreach(n, st, X3_LAMBDA(size_t i) { oeq[vs[i]] = (uint32_t)i - rs[i]; });
		CHK(csb_run(B, st, n, abits, kc, bs, be, osm, T + 8)); /* the hit/event work arrays also cover nM and nB (nA above) */
	}

	/* ---- the coded symbols, in coding order: symbol index of step k = 2k + (new-fragment bytes before k) ---- */
	uint32_t *Yv[12];
	for (int i = 0; i < 12; i++) {
		const size_t ny = (i == 0 || i == 3) ? nYres : ((final || i == 5 || i == 6) ? nY : 0); /* the emit arrays are only needed by the final call; 5, 6 also serve the compaction */
		CHK(B.y[i].reserve((ny + (i == 0 ? X3_SYM_PAD : 4)) * (i == 0 ? 16 : i == 3 ? 8 : 4)));
		Yv[i] = B.y[i].as<uint32_t>();
	}
	CHK(B.yraw.reserve((nYres + 4) * 16));
	uint4 *syr = B.yraw.as<uint4>(); /* every symbol at its closed-form index (no-ops included) */
	uint4 *sy = (uint4 *)Yv[0]; /* {cum, freq, magic, shift} per symbol */
	uint32_t *rec_nk = Yv[3]; /* 2 words per symbol */
	/* No-op symbols (total == 1, only the tag / index symbol of a hit can be one) are dropped from the chain's input; everything
	 * downstream (chain states, emission) lives in compacted symbol indices.  Stage-after-stage calls assemble straight into the compacted
	 * list: nzb[h] = no-op symbols among the hits before h (a scan over the hits).  Prefix calls of the pipelined schedule assemble the
	 * raw list first: they need the compacted index of every raw symbol to find the NEW ones (a prefix of the raw list compacts to a
	 * prefix of the compacted list). */
	uint32_t *nzf = T[3], *nzb = T[4];
	if (!seg && inplace) {
		/* nzb[hit] = no-ops among the STREAM's earlier hits (mode kernel); the streams' totals give the compacted offsets: one small serial pass */
		x3_foreach(1, st, X3_LAMBDA(size_t) { uint32_t acc = 0; for (uint32_t c = 0; c <= nc; c++) { d_yoc[c] = d_yo[c] - acc; if (c < nc) acc += m_nnoop[c]; } });
	} else if (!seg) {
		HIPCHK(hipMemsetAsync(nzb, 0, 4, st));
		if (nH > 0) {
			x3_foreach(nH, st, X3_LAMBDA(size_t gh) { nzf[gh] = hs_tot[gh] <= 1u ? 1u : 0u; });
			CHK(x3p_excl_scan(B.tmp, nzf, nzb, nH, st));
		}
		x3_foreach(nc + 1, st, X3_LAMBDA(size_t c) { d_yoc[c] = d_yo[c] - nzb[d_ho[c]]; });
	}
	uint4 *sdst = seg ? syr : sy;
	const bool direct = !seg;
	/* size estimates (optional): one term per coded symbol slot of this layout, written by the assembly pass below */
	const bool est = B.want_est && final;
	float *est_val = nullptr;
	uint8_t *est_cls = nullptr;
	if (B.want_est) { /* (sized on every call of a growing prefix, filled by the final one: nothing is reallocated while the coder runs) */
		CHK(B.est_val.reserve((nYres + 4) * 4)); CHK(B.est_cls.reserve(nYres + 4)); CHK(B.est_out.reserve((size_t)nc * 16 + 16));
		if (!B.est_stream) { HIPCHK(hipStreamCreate(&B.est_stream)); HIPCHK(hipEventCreate(&B.ev_est_fork)); HIPCHK(hipEventCreate(&B.ev_est_done)); }
		if (est) { est_val = B.est_val.as<float>(); est_cls = B.est_cls.as<uint8_t>(); HIPCHK(hipMemsetAsync(est_cls, X3_EST_NONE, nY + 4, st)); }
	}
	x3_foreach(nS, st, X3_LAMBDA(size_t gs) {
		const uint32_t c = find_chunk(d_so, nc, (uint32_t)gs);
		const uint32_t k = (uint32_t)gs - d_so[c];
		const uint64_t base = d_chunks[c].elem_off;
		const uint32_t info = tok_info[base + k], hb = tok_hb[base + k] - tokb[4 * c], mb = tok_mb[base + k] - tokb[4 * c + 2];
		uint32_t yi = d_yo[c] + 2 * k + mb;
		if (inplace) yi = d_yoc[c] + 2 * k + mb - (hb < d_ho[c + 1] - d_ho[c] ? nzb[d_ho[c] + hb] : m_nnoop[c]); /* nzb: stream-local here (mode kernel); behind the last hit: the stream's total */
		else if (direct) yi -= nzb[d_ho[c] + hb]; /* == d_yoc[c] + 2k + mb - (no-ops among the stream's earlier hits) */
		const uint32_t evtotal = 2051u + k; /* model_events: 1024+1024+1+1+1 (x3.c:236-244), +1 per step */
		if (!(info & X3_TOK_MISS)) {
			const uint32_t gh = d_ho[c] + hb, m = mode[gh], e0 = pe0[gh], e1 = pe1[gh];
			const uint32_t e2 = 2049u + hb - e0 - e1; /* every hit bumps exactly one of the three */
			sdst[yi] = x3_make_symbol(m == X3_E_CTX0 ? 0u : m == X3_E_CTX1 ? e0 : e0 + e1, m == X3_E_CTX0 ? e0 : m == X3_E_CTX1 ? e1 : e2, evtotal);
			uint32_t tfq, tto;
			if (inplace) {
				uint32_t cu, fq, to;
				if (m == X3_E_CTX0) { const uint4 r = recs0[gh]; cu = r.z; fq = r.x; to = r.y; }
				else if (m == X3_E_CTX1) { const uint4 r = recs1[gh]; cu = r.z; fq = r.x; to = r.y; }
				else { cu = rcum[gh]; fq = rfreq[gh]; to = itot[gh]; }
				if (to > 1u) sdst[yi + 1] = x3_make_symbol(cu, fq, to);
				tfq = fq; tto = to;
			} else {
				if (!direct || hs_tot[gh] > 1u) sdst[yi + 1] = x3_make_symbol(hs_cum[gh], hs_freq[gh], hs_tot[gh]);
				tfq = hs_freq[gh]; tto = hs_tot[gh];
			}
			if (est_cls) { /* x3.c:152-172,192-193: the chosen product, evaluated as the reference evaluates it */
				const float pe = (float)(m == X3_E_CTX0 ? e0 : m == X3_E_CTX1 ? e1 : e2) / (float)evtotal;
				est_val[yi] = x3_est_term(pe * ((float)tfq / (float)tto));
				est_cls[yi] = (uint8_t)m;
			}
		} else {
			const uint32_t len = info & 0x3Fu, mk = k - hb; /* mk = new fragments before this one */
			sdst[yi] = x3_make_symbol(2049u + hb, 1u + mk, evtotal); /* E_NEW */
			const uint32_t gm = d_mo[c] + mk;
			sdst[yi + 1] = x3_make_symbol((len - 1) + lsm[gm], 1u + leq[gm], 32u + mk);
			if (est_cls) { /* x3.c:253,259: event and length symbol */
				est_val[yi] = x3_est_term((float)(1u + mk) / (float)evtotal); est_cls[yi] = X3_E_NEW;
				est_val[yi + 1] = x3_est_term((float)(1u + leq[gm]) / (float)(32u + mk)); est_cls[yi + 1] = X3_E_NEW;
			}
			for (uint32_t j = 0; j < len; j++) {
				const uint32_t gb = d_bo[c] + mb + j;
				sdst[yi + 2 + j] = x3_make_symbol(bval[gb] + bsm[gb], 1u + beq[gb], 256u + mb + j);
				if (est_cls) { est_val[yi + 2 + j] = x3_est_term((float)(1u + beq[gb]) / (float)(256u + mb + j)); est_cls[yi + 2 + j] = X3_E_NEW; } /* x3.c:264 */
			}
		}
	});
	if (final) x3_foreach(nc, st, X3_LAMBDA(size_t c) { /* E_EOF, x3.c:432-433 */
		const uint32_t yi = (direct ? d_yoc[c + 1] : d_yo[c + 1]) - 1, evtotal = 2051u + d_parsed[c].ntok;
		sdst[yi] = x3_make_symbol(evtotal - 1, 1, evtotal);
	});
	B.est_pending = false;
	if (est) { /* the four float accumulators of every stream, on their own HIP stream beside the coder; api.hip joins it (est_pending) */
		X3EstArgs ea;
		ea.range = direct ? d_yoc : d_yo; ea.val = est_val; ea.cls = est_cls; ea.out = B.est_out.as<float>(); ea.seg_first = ea.seg_count = nullptr;
		HIPCHK(hipEventRecord(B.ev_est_fork, st));
		HIPCHK(hipStreamWaitEvent(B.est_stream, B.ev_est_fork, 0));
		launch_est(ea, nc, B.est_stream);
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventRecord(B.ev_est_done, B.est_stream));
		B.est_pending = true;
	}

	uint32_t *kf = Yv[5], *Pk = Yv[6];
	if (seg) {
		x3_foreach(nY, st, X3_LAMBDA(size_t y) { kf[y] = syr[y].w != X3_SYM_NOOP ? 1u : 0u; });
		CHK(x3p_excl_scan(B.tmp, kf, Pk, nY, st));
		x3_foreach(nc + 1, st, X3_LAMBDA(size_t c) { d_yoc[c] = Pk[d_yo[c]]; });
	}
	std::vector<uint32_t> yoc(nc + 1);
	HIPCHK(hipMemcpyAsync(yoc.data(), d_yoc, (nc + 1) * 4, hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	const size_t nYraw = nY;
	const size_t nYc = yoc[nc];

	/* ---- serial pass 2: interval recurrence ---- */
	X3Ac2Args aa;
	aa.yo = d_yoc; aa.sym = sy; aa.rec_nk = rec_nk; aa.final_lo = m_finallo; aa.compact = (!seg && streamk) ? 1u : 0u;
	aa.seg_off = aa.seg_len = nullptr; aa.seg_state = nullptr;
	if (nH == 0) HIPCHK(hipEventRecord(B.ev[2], st));
	if (!seg) {
		HIPCHK(hipEventRecord(B.ev[3], st));
		launch_ac2(aa, nc, st);
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventRecord(B.ev[4], st));
	} else {
		/* the NEW chain symbols of every stream go into the ring and to the coder stream; this (feature) stream carries on with the
		 * next prefix.  A non-final segment ends on a group boundary of its stream: the few symbols left over are put again next time. */
		if (seg->y_done.size() != nc) seg->y_done.assign(nc, 0u);
		const size_t kcall = seg->calls.size();
		if (kcall >= X3_MAX_CKPT + 2) return X3H_E_INTERNAL;
		seg->calls.emplace_back();
		X3CodeSegCall &cl = seg->calls.back();
		cl.base = seg->ring_top; cl.off.resize(nc + 1); cl.len.resize(nc); cl.before.resize(nc);
		uint64_t acc = 0;
		for (uint32_t c = 0; c < nc; c++) {
			const uint32_t cntc = yoc[c + 1] - yoc[c], newc = cntc - seg->y_done[c];
			cl.off[c] = (uint32_t)(cl.base + acc); cl.before[c] = seg->y_done[c];
			cl.len[c] = final ? newc : newc - newc % X3_AC2_G;
			acc += newc;
		}
		cl.off[nc] = (uint32_t)(cl.base + acc); cl.total = acc;
		if (cl.base + acc > nYres) return X3H_E_INTERNAL; /* the ring bound is a hard bound: cannot happen */
		CHK(seg->meta.reserve((size_t)(X3_MAX_CKPT + 2) * (3 * (size_t)nc + 1) * 4));
		uint32_t *d_off = seg->meta.as<uint32_t>() + kcall * (3 * (size_t)nc + 1), *d_len = d_off + (nc + 1), *d_before = d_len + nc;
		HIPCHK(hipMemcpyAsync(d_off, cl.off.data(), (nc + 1) * 4, hipMemcpyHostToDevice, st));
		HIPCHK(hipMemcpyAsync(d_len, cl.len.data(), nc * 4, hipMemcpyHostToDevice, st));
		HIPCHK(hipMemcpyAsync(d_before, cl.before.data(), nc * 4, hipMemcpyHostToDevice, st));
		{
			const uint32_t *doff = d_off, *dbef = d_before;
			x3_foreach(nY, st, X3_LAMBDA(size_t y) {
				if (!kf[y]) return;
				const uint32_t c = find_chunk(d_yo, nc, (uint32_t)y);
				const uint32_t r = Pk[y] - d_yoc[c];
				if (r >= dbef[c]) sy[doff[c] + (r - dbef[c])] = syr[y];
			});
		}
		aa.seg_off = d_off; aa.seg_len = d_len; aa.seg_state = seg->coder_state;
		HIPCHK(hipEventRecord(seg->ev_ready, st));
		HIPCHK(hipStreamWaitEvent(seg->coder_stream, seg->ev_ready, 0));
		HIPCHK(hipEventRecord(seg->ev_coder_begin, seg->coder_stream));
		launch_ac2(aa, nc, seg->coder_stream);
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventRecord(seg->ev_coder_end, seg->coder_stream));
		for (uint32_t c = 0; c < nc; c++) seg->y_done[c] += cl.len[c];
		seg->ring_top += acc;
		B.last.symbols = nYraw; B.last.chain_symbols = nYc;
		if (seg->emit_stream) {
			/* the bits of THIS segment, straight from the rings, behind its recurrence and beside the recurrence of the next one: the pending-bit
			 * count and the bit position of every stream travel from launch to launch; the last launch flushes and writes the results */
			if (kcall == 0) {
				CHK(seg->emit_state.reserve((size_t)nc * 16));
				HIPCHK(hipStreamWaitEvent(seg->emit_stream, seg->ev_ready, 0)); /* the stream table (out_off / out_cap) is in place */
				HIPCHK(hipMemsetAsync(seg->emit_state.p, 0, (size_t)nc * 16, seg->emit_stream));
				CHK(zero_output_slots(seg->emit_stream, nc, h_chunks, d_chunks, d_out));
			}
			HIPCHK(hipStreamWaitEvent(seg->emit_stream, seg->ev_coder_end, 0));
			X3EmitArgs ea;
			ea.yoc = nullptr; ea.sym = sy; ea.state = rec_nk; ea.final_lo = m_finallo; ea.chunks = d_chunks; ea.parsed = d_parsed;
			ea.npairs = m_npairs; ea.evfinal = m_evfinal; ea.out = d_out; ea.result = d_result;
			ea.seg_off = d_off; ea.seg_len = d_len; ea.carry = seg->emit_state.as<uint32_t>(); ea.last = final ? 1u : 0u; ea.compact = 0; ea.ntok = ea.nhits = nullptr;
			launch_emit(ea, nc, seg->emit_stream, true);
			HIPCHK(hipGetLastError());
			if (!final) return X3H_OK;
			HIPCHK(hipEventRecord(seg->ev_emit_done, seg->emit_stream));
			HIPCHK(hipStreamWaitEvent(st, seg->ev_emit_done, 0));
			return X3H_OK;
		}
		if (!final) return X3H_OK;
		HIPCHK(hipStreamWaitEvent(st, seg->ev_coder_end, 0)); /* emission needs every chain state */
		/* final symbol layout: the operands once more (one scatter) and the chain states of every segment gathered from the ring */
		CHK(B.yfin.reserve((nYc + X3_SYM_PAD) * 16));
		CHK(B.yfinrec.reserve((nYc + 4) * 8));
		uint4 *syf = B.yfin.as<uint4>();
		uint32_t *recf = B.yfinrec.as<uint32_t>();
		x3_foreach(nY, st, X3_LAMBDA(size_t y) { if (kf[y]) syf[Pk[y]] = syr[y]; });
		for (size_t k = 0; k < seg->calls.size(); k++) {
			const X3CodeSegCall &ck = seg->calls[k];
			if (!ck.total) continue;
			const uint32_t *koff = seg->meta.as<uint32_t>() + k * (3 * (size_t)nc + 1), *klen = koff + (nc + 1), *kbef = klen + nc;
			const uint32_t *ring = rec_nk;
			const uint32_t kbase = (uint32_t)ck.base;
			x3_foreach((size_t)ck.total, st, X3_LAMBDA(size_t t) {
				const uint32_t pos = kbase + (uint32_t)t, c = find_chunk(koff, nc, pos), rel = pos - koff[c];
				if (rel >= klen[c] || rel % X3_AC2_G) return;
				const size_t dst = (size_t)d_yoc[c] + kbef[c] + rel;
				recf[2 * dst] = ring[2 * (size_t)pos]; recf[2 * dst + 1] = ring[2 * (size_t)pos + 1];
			});
		}
		sy = syf; rec_nk = recf;
	}

	CHK(zero_output_slots(st, nc, h_chunks, d_chunks, d_out));
	if (streamk) {
		/* many streams: one workgroup per stream carries the pending-bit count and the bit position through its symbols */
		X3EmitArgs ea;
		ea.yoc = d_yoc; ea.sym = sy; ea.state = rec_nk; ea.final_lo = m_finallo; ea.chunks = d_chunks; ea.parsed = d_parsed;
		ea.npairs = m_npairs; ea.evfinal = m_evfinal; ea.out = d_out; ea.result = d_result;
		ea.seg_off = ea.seg_len = nullptr; ea.carry = nullptr; ea.last = 1; ea.compact = seg ? 0u : 1u; ea.ntok = ea.nhits = nullptr;
		launch_emit(ea, nc, st);
		HIPCHK(hipGetLastError());
	} else {
		/* ---- bit emission (ac.c:49-67 put_bit + mScale, bio.c:49-72) as prefix sums over the records ----
		 * pending after symbol i  = sum of k over (last symbol that shifted out bits (n>=1) or stream start .. i]
		 * bits written by symbol i = n + pending before it (when n >= 1): first bit, the pending bits inverted, the other n-1 bits */
		uint32_t *kk = Yv[5], *rv = Yv[6], *Kex = Yv[7], *LE = Yv[8], *len = Yv[9], *pos = Yv[10], *pend = Yv[11];
		uint32_t *ebits = (uint32_t *)syr; /* the uncompacted operand array is dead now: one word per symbol goes there: 1 << n | the n bits the symbol shifts out */
		{
			/* x3_expand_records: the chain left its state at the first symbol of every group of X3_AC2_G; one thread re-runs each group and
			 * derives, per symbol, n (E1/E2 shifts) with the n emitted bits, k (E3 shifts) and the "resets the pending count" marker */
			const uint32_t *stt = rec_nk;
			const uint4 *syc = sy;
			x3_foreach(nYc, st, X3_LAMBDA(size_t i) {
				const uint32_t c = find_chunk(d_yoc, nc, (uint32_t)i);
				if (((uint32_t)i - d_yoc[c]) % X3_AC2_G) return;
				uint32_t lo = stt[2 * i], R = stt[2 * i + 1];
				const uint32_t first = d_yoc[c], end = d_yoc[c + 1], cnt = end - (uint32_t)i < X3_AC2_G ? end - (uint32_t)i : X3_AC2_G;
				for (uint32_t j = 0; j < cnt; j++) {
					const uint2 r = x3_chain_step(lo, R, syc[i + j]);
					const uint32_t n = x3_rec_n(r.x, r.y); /* <= 30: the interval has at least two values */
					const uint32_t rev = x3_brev32(r.x << 1) & ((1u << n) - 1); /* bit j = j-th emitted bit = bit 30-j of lo */
					ebits[i + j] = (1u << n) | rev;
					kk[i + j] = x3_rec_k(r.x, r.y);
					rv[i + j] = (n >= 1 || i + j == first) ? (uint32_t)(i + j) + 1 : 0u;
				}
			});
		}
		CHK(x3p_excl_scan(B.tmp, kk, Kex, nYc, st));
		CHK(x3p_incl_max_scan(B.tmp, rv, LE, nYc, st));
		x3_foreach(nYc, st, X3_LAMBDA(size_t i) { pend[i] = Kex[i + 1] - Kex[LE[i] - 1]; });
		x3_foreach(nYc, st, X3_LAMBDA(size_t i) {
			const uint32_t n = 31u - (uint32_t)x3_clz32(ebits[i]);
			const uint32_t c = find_chunk(d_yoc, nc, (uint32_t)i);
			len[i] = n >= 1 ? n + (i == d_yoc[c] ? 0u : pend[i - 1]) : 0u;
		});
		CHK(x3p_excl_scan(B.tmp, len, pos, nYc, st));
		x3_foreach(nYc, st, X3_LAMBDA(size_t i) {
			const uint32_t ln = len[i];
			if (!ln) return;
			const uint32_t c = find_chunk(d_yoc, nc, (uint32_t)i);
			uint32_t *out32 = (uint32_t *)(d_out + d_chunks[c].out_off);
			const uint32_t capw = (uint32_t)(d_chunks[c].out_cap / 4);
			const uint32_t eb = ebits[i], n = 31u - (uint32_t)x3_clz32(eb), pd = ln - n;
			const uint64_t bp = pos[i] - pos[d_yoc[c]];
			const uint32_t rev = eb ^ (1u << n);
			if (!pd) x3_or_bits(out32, capw, bp, rev, n);
			else {
				x3_or_bits(out32, capw, bp, rev & 1u, 1);
				x3_or_run(out32, capw, bp + 1, (rev & 1u) ^ 1u, pd);
				x3_or_bits(out32, capw, bp + 1 + pd, rev >> 1, n - 1);
			}
		});
		x3_foreach(nc, st, X3_LAMBDA(size_t c) { /* ac_encode_flush (ac.c:115-126) + bio_close (bio.c:105-112) + result */
			uint32_t *out32 = (uint32_t *)(d_out + d_chunks[c].out_off);
			const uint32_t capw = (uint32_t)(d_chunks[c].out_cap / 4);
			const uint32_t last = d_yoc[c + 1] - 1;
			uint64_t nbits = pos[d_yoc[c + 1]] - pos[d_yoc[c]];
			if (m_finallo[c] < 0x20000000u) {
				x3_or_run(out32, capw, nbits + 1, 1u, pend[last] + 1); /* '0' then mScale+1 ones */
				nbits += 2 + (uint64_t)pend[last];
			} else {
				x3_or_bits(out32, capw, nbits, 1u, 1);
				nbits += 1;
			}
			const uint64_t words = (nbits + 31) / 32;
			X3CodeResult r;
			r.out_len = (uint32_t)(words * 4); r.status = words > capw ? X3_ST_OUT_FULL : X3_ST_OK; r.pairs = m_npairs[c]; r._r = 0;
			r.events[0] = m_evfinal[4 * c] - 1024; r.events[1] = m_evfinal[4 * c + 1] - 1024; r.events[2] = m_evfinal[4 * c + 2] - 1;
			r.events[3] = d_parsed[c].ntok - d_parsed[c].hits;
			r.events[4] = r.events[5] = r.events[6] = r.events[7] = 0;
			d_result[c] = r;
		});
	}
	(void)tok_nb;
	B.last.symbols = nYraw; B.last.chain_symbols = nYc;
	return X3H_OK;
}

/* ============================================================================================================
 * Stage seam (x3h_coder_chain): the interval recurrence of x3_ac2_kernel on a symbol sequence given by the caller -- the same kernel,
 * the same operand format (x3_make_symbol), one stream.  Returns the chain states the kernel stores (one per group of X3_AC2_G symbols,
 * lo reduced to the 30 bits that are the reference's mLow) and the final mLow.
 * ============================================================================================================ */
int x3_coder_chain_run(X3Code2Bufs &B, hipStream_t st, const uint32_t *h_cum, const uint32_t *h_freq, const uint32_t *h_total, size_t n,
                       uint32_t *h_states, uint32_t *h_final_lo)
{
	if (n >= (1u << 30)) return X3H_E_ARG;
	for (size_t i = 0; i < n; i++) if (h_total[i] < 2 || h_total[i] >= (1u << 28) || !h_freq[i] || (uint64_t)h_cum[i] + h_freq[i] > h_total[i]) return X3H_E_ARG;
	for (int i = 0; i < 3; i++) CHK(B.a[i].reserve((n + 4) * 4));
	CHK(B.y[0].reserve((n + X3_SYM_PAD) * 16));
	CHK(B.y[3].reserve((n + 8) * 8));
	CHK(B.offs.reserve(64));
	uint32_t *d_cum = B.a[0].as<uint32_t>(), *d_freq = B.a[1].as<uint32_t>(), *d_tot = B.a[2].as<uint32_t>();
	uint4 *sy = B.y[0].as<uint4>();
	uint32_t *rec = B.y[3].as<uint32_t>(), *d_yo = B.offs.as<uint32_t>();
	HIPCHK(hipMemcpyAsync(d_cum, h_cum, n * 4, hipMemcpyHostToDevice, st));
	HIPCHK(hipMemcpyAsync(d_freq, h_freq, n * 4, hipMemcpyHostToDevice, st));
	HIPCHK(hipMemcpyAsync(d_tot, h_total, n * 4, hipMemcpyHostToDevice, st));
	const uint32_t yo[2] = { 0, (uint32_t)n };
	HIPCHK(hipMemcpyAsync(d_yo, yo, 8, hipMemcpyHostToDevice, st));
	x3_foreach(n, st, X3_LAMBDA(size_t i) { sy[i] = x3_make_symbol(d_cum[i], d_freq[i], d_tot[i]); });
	X3Ac2Args aa;
	aa.yo = d_yo; aa.sym = sy; aa.rec_nk = rec; aa.final_lo = d_yo + 4; aa.seg_off = aa.seg_len = nullptr; aa.seg_state = nullptr; aa.compact = 0;
	launch_ac2(aa, 1, st);
	HIPCHK(hipGetLastError());
	std::vector<uint32_t> all(2 * (n + 1));
	HIPCHK(hipMemcpyAsync(all.data(), rec, 2 * n * 4, hipMemcpyDeviceToHost, st));
	HIPCHK(hipMemcpyAsync(h_final_lo, d_yo + 4, 4, hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	for (size_t g = 0; g * X3_AC2_G < n; g++) { h_states[2 * g] = all[2 * g * X3_AC2_G] & 0x3FFFFFFFu; h_states[2 * g + 1] = all[2 * g * X3_AC2_G + 1]; }
	return X3H_OK;
}
