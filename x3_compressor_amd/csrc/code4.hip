/*
 * code4.hip -- K3 in SLICES: the coding stage of x3 (reference: x3.c:132-270,431-433 with dict.c:132-146, context.c, tag_pair.c, ac.c, bio.c) as a
 * pipeline of per-stream kernels that each take ONE TIME SLICE of every stream and carry the adaptive state from slice to slice.
 *
 * Why: every stage of a stream is a dependent chain (parse: one workgroup, mode choice and coder: one wavefront each), but they are DIFFERENT chains,
 * and the slowest one -- the arithmetic coder, ~2 symbols x 25.6 ns per parse step -- needs nothing from a step but its operands.  api.hip's first
 * overlapped schedule (run_pipelined) recomputes the features of the whole PREFIX at every checkpoint of the running parse with chip-wide sorts and
 * partitions (code2.hip): fine for one long stream, but its cost per checkpoint is the prefix, not the slice, so it can neither follow a batch of a few
 * dozen 256 KiB streams (the >= 1 GB/s point of the dickens-sized workload) nor keep sixteen 8 MiB coder chains fed (BASELINE config 4).
 * Here the state every feature is a function of is CARRIED:
 *     move-to-front order       last touch of every element (an array per stream)                     dict.c:132-146
 *     context item lists        per context {offset, items, capacity, total} + items {tag, freq} in first-seen order, in a per-stream pool that
 *                               grows by doubling like the reference's realloc                          context.c:7-56
 *     tag-pair map              the ordinal of a pair lives with the context1 item that names it      tag_pair.c:100-130, x3.c:213-222
 *     model_events / index1     three counters + one frequency per rank                                x3.c:176-188,236-244
 *     model_match_size / chars  32 + 256 counters                                                      x3.c:259-267
 *     coder interval, pending bits, bit position                                                       ac.c:46-85, bio.c:49-72
 * so a slice costs what the SLICE costs, the slices of a stream follow each other through the stages on separate HIP streams (parse | features |
 * coder | bit emission), and a step's work arrays are temporaries of its slice (dense over the slice, reused by the next one).
 * The running counts at both ends of a slice (steps, hits, elements, fragment bytes, position) come from the checkpoint records the parse publishes
 * to host-mapped memory (X3ParseCkpt), so the host sizes every launch of a slice without a device round trip.
 *
 * Per slice:  x3s_tokens (records per hit / touch / fragment)  ->  [side stream: x3s_mtf (ranks)]  ->  arrangement by context1  ->  x3s_ctx<ord>  ->
 * x3s_pairs (ordinals of the new pairs, context0 of every hit)  ->  arrangement by context0  ->  x3s_ctx  ->  mode chain (code2.hip)  ->  x3s_idxstat,
 * x3s_order0  ->  symbol assembly  ->  [coder stream: x3_ac2_kernel on the new symbols]  ->  [emit stream: x3_emit_kernel, segment form].
 * Dictionaries of up to X3S_DMAX elements (the per-element tables of a wavefront live in LDS); api.hip falls back to the other schedules beyond.
 */
#include "k3_wave.h"
#include "k3_sym.h"

#include <stdlib.h>
#include <algorithm>
#include <vector>

static inline int bits_for64(uint64_t maxval) { int b = 1; while (b < 32 && (maxval >> b)) b++; return b; }

/* largest c with off[c] <= idx (off has n+1 non-decreasing entries, off[n] > idx) */
__device__ static __forceinline__ uint32_t s_find(const X3Slice *sl, uint32_t n, uint32_t idx)
{
	uint32_t lo = 0, hi = n; /* answer in [lo, hi): by slice-local step offset */
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (sl[mid].ss <= idx) lo = mid; else hi = mid;
	}
	return lo;
}

/* ... by slice-local hit */
__device__ static __forceinline__ uint32_t s_find_hit(const X3Slice *sl, uint32_t n, uint32_t idx)
{
	uint32_t lo = 0, hi = n;
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (sl[mid].sh <= idx) lo = mid; else hi = mid;
	}
	return lo;
}

/* ============================================================================================================
 * Token walk of a slice (x3.c:379-429 read back from K2's one word per step): running counts, per-hit / per-touch records, the values of the
 * new fragments.  One workgroup per stream; the counts at the slice's first step come from the checkpoint (X3Slice).
 * ============================================================================================================ */
#define X3S_TOK_THREADS 1024u
struct X3sTokArgs {
	const X3Chunk *chunks; const X3Slice *sl;
	const uint8_t *bytes; const uint32_t *tok_info; const uint8_t *dict_len;
	uint32_t *s_hb, *s_mb;                                 /* out per slice step: hits / fragment bytes before the step (stream-absolute) */
	uint32_t *h_tag, *h_c1, *h_pv, *h_dk, *h_step, *k1;    /* out per slice hit: tag, context1, "previous step was a hit", elements before, step, arrangement key (stream << kshift | context1) */
	uint32_t *e_tag, *e_hit;                               /* out per touch event (hit or insertion): tag, slice-local hit or NONE32 */
	uint32_t *lval, *bval;                                 /* out per new fragment / fragment byte: length - 1, byte */
	uint32_t kshift;
};

__device__ static void x3s_tokens_body(const X3sTokArgs &a)
{
	const uint32_t NW = X3S_TOK_THREADS / X3_WAVE;
	X3_LDS uint32_t s_w[3][X3S_TOK_THREADS / X3_WAVE];
	const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = x3_lane(), wave = tid / X3_WAVE;
	const X3Slice sl = a.sl[c];
	const uint64_t base = a.chunks[c].elem_off;
	const uint8_t *bytes = a.bytes + a.chunks[c].byte_off;
	const uint64_t below = ((uint64_t)1 << lane) - 1;
	uint32_t chb = sl.h0, cnb = sl.d0, cmb = sl.mb0, cpos = sl.p0; /* counts before the tile */
	uint32_t ninfo_ = sl.t0 < sl.t1 ? a.tok_info[base + (sl.t0 + tid < sl.t1 ? sl.t0 + tid : sl.t1 - 1)] : 0u; /* (one tile ahead, unconditional loads from clamped indices) */
	for (uint32_t tb = sl.t0; tb < sl.t1; tb += X3S_TOK_THREADS) {
		const uint32_t k = tb + tid;
		const bool in = k < sl.t1;
		const uint32_t info = in ? ninfo_ : X3_TOK_MISS;
		ninfo_ = a.tok_info[base + (k + X3S_TOK_THREADS < sl.t1 ? k + X3S_TOK_THREADS : sl.t1 - 1)];
		const bool hit = in && !(info & X3_TOK_MISS), nw = in && (info & X3_TOK_MISS) && !(info & X3_TOK_DUP), miss = in && (info & X3_TOK_MISS);
		const uint32_t mb = miss ? (info & 0x3Fu) : 0u;
		const uint32_t ln = hit ? (uint32_t)a.dict_len[base + info] : mb;
		const uint64_t Hm = x3_ballot(hit), Nm = x3_ballot(nw);
		const uint32_t pk_w = x3_wave_incl_scan_u32(mb | ln << 16); /* both sums stay below 2^16 inside a tile (1024 x 32) */
		if (lane == X3_WAVE - 1) { s_w[0][wave] = (uint32_t)x3_popc64(Hm); s_w[1][wave] = (uint32_t)x3_popc64(Nm); s_w[2][wave] = pk_w; }
		__syncthreads();
		uint32_t hbw = 0, nbw = 0, pkw = 0, hbt = 0, nbt = 0, pkt = 0;
		for (uint32_t w = 0; w < NW; w++) {
			if (w < wave) { hbw += s_w[0][w]; nbw += s_w[1][w]; pkw += s_w[2][w]; }
			hbt += s_w[0][w]; nbt += s_w[1][w]; pkt += s_w[2][w];
		}
		const uint32_t hb = chb + hbw + (uint32_t)x3_popc64(Hm & below), nb = cnb + nbw + (uint32_t)x3_popc64(Nm & below);
		const uint32_t pk = pkw + pk_w - (mb | ln << 16);
		const uint32_t mbb = cmb + (pk & 0xFFFFu), pos = cpos + (pk >> 16);
		uint32_t pinfo = x3_shfl_up_u32(info, 1); /* the step before mine: lane - 1, or (lane 0) read back */
		if (in) {
			if (lane == 0) pinfo = k == 0 ? X3_TOK_MISS : a.tok_info[base + k - 1];
			const uint32_t s = sl.ss + (k - sl.t0);
			a.s_hb[s] = hb; a.s_mb[s] = mbb;
			if (hit) {
				const uint32_t j = sl.sh + (hb - sl.h0), ev = sl.se + (hb - sl.h0) + (nb - sl.d0);
				const bool pv = !(pinfo & X3_TOK_MISS);
				const uint32_t c1 = pv ? pinfo : 0u; /* context1 (x3.c:390,425) */
				a.h_tag[j] = info; a.h_c1[j] = c1; a.h_pv[j] = pv ? 1u : 0u; a.h_dk[j] = nb; a.h_step[j] = k;
				a.k1[j] = (c << a.kshift) | c1;
				a.e_tag[ev] = info; a.e_hit[ev] = j;
			} else {
				const uint32_t mi = sl.sm + ((k - hb) - (sl.t0 - sl.h0)); /* new fragments before this one, inside the slice */
				a.lval[mi] = mb - 1;
				const uint32_t b0 = sl.sb + (mbb - sl.mb0);
				for (uint32_t q = 0; q < mb; q++) a.bval[b0 + q] = bytes[pos + q];
				if (nw) { const uint32_t ev = sl.se + (hb - sl.h0) + (nb - sl.d0); a.e_tag[ev] = nb; a.e_hit[ev] = NONE32; } /* the new element's tag (dict.c:100) */
			}
		}
		chb += hbt; cnb += nbt; cmb += pkt & 0xFFFFu; cpos += pkt >> 16;
		__syncthreads();
	}
}

/* ============================================================================================================
 * Move-to-front ranks of a slice (dict.c:132-146 == move-to-front; code3.hip x3_mtfrank_par_kernel for whole streams): the slice's touch events cut
 * into WAVES time ranges, one wavefront each.  The list at the start of a range is the elements that exist by then in order of their last touch --
 * and the last touches BEFORE the slice are the carried state: lt[tag] = 1 + (stream-absolute index of the element's last touch event), per stream.
 * ============================================================================================================ */
struct X3sMtfArgs {
	const X3Chunk *chunks; const X3Slice *sl; const uint32_t *e_tag, *e_hit; uint32_t *h_rank;
	const uint32_t *lt; /* carried per stream (at elem_off): per tag, last touch + 1 (stream-absolute index of the touch event), 0: never */
	uint32_t *lt_out;   /* ... as the NEXT slice finds it: a second array, because the ranges of this slice read `lt` whenever they get to run */
	uint32_t *scratch;  /* [stream][range][X3S_DMAX + 1]: last touch + 1 of every tag INSIDE the range (0: none), and the range's insertions in the last word */
	uint32_t nranges;   /* time ranges per stream, one wavefront (= workgroup) each */
};

/* pass 1: wavefront (stream, range) records the last touch of every tag inside its range, and how many elements the range inserts */
template <uint32_t DMAX>
__device__ static void x3s_mtf_scan_body(const X3sMtfArgs &a)
{
	X3_LDS uint32_t tab[DMAX];
	const uint32_t c = blockIdx.x / a.nranges, r = blockIdx.x % a.nranges, lane = x3_lane();
	const X3Slice sl = a.sl[c];
	const uint32_t e0 = sl.se, e1 = sl.se + (sl.h1 - sl.h0) + (sl.d1 - sl.d0);
	const uint32_t abs0 = sl.h0 + sl.d0; /* stream-absolute index of the slice's first touch event */
	const uint32_t per = (((e1 - e0 + a.nranges - 1) / a.nranges) + X3_WAVE - 1) & ~(X3_WAVE - 1);
	const uint32_t s0 = e0 + r * per < e1 ? e0 + r * per : e1, s1 = s0 + per < e1 ? s0 + per : e1;
	uint32_t *out = a.scratch + ((size_t)c * a.nranges + r) * (X3S_DMAX + 1);
	const uint32_t D = sl.d1;
	for (uint32_t i = lane; i < D; i += X3_WAVE) tab[i] = 0;
	x3_wave_order();
	uint32_t cnt_new = 0;
	uint32_t nt_ = 0, nh_ = 0;
	if (s0 < s1) { const uint32_t i0 = s0 + lane < s1 ? s0 + lane : s1 - 1; nt_ = a.e_tag[i0]; nh_ = a.e_hit[i0]; }
	for (uint32_t base = s0; base < s1; base += X3_WAVE) {
		const uint32_t i = base + lane;
		const bool valid = i < s1;
		const uint32_t t = nt_, hit = nh_;
		{ const uint32_t nx = i + X3_WAVE < s1 ? i + X3_WAVE : s1 - 1; nt_ = a.e_tag[nx]; nh_ = a.e_hit[nx]; }
		if (valid) atomicMax(&tab[t], abs0 + (i - e0) + 1);
		cnt_new += (uint32_t)x3_popc64(x3_ballot(valid && hit == NONE32));
	}
	x3_wave_order();
	for (uint32_t i = lane; i < D; i += X3_WAVE) out[i] = tab[i];
	if (lane == 0) out[X3S_DMAX] = cnt_new;
}

/* pass 2: wavefront (stream, range) builds the list as of its range's first event -- the elements that exist by then in order of their last touch, latest
 * first: last touch before the range = the maximum over the carried table and the earlier ranges' tables; an element's position = how many elements were
 * touched later (last touches are distinct events) -- and replays its events in tiles (x3_mtf_tiles).  The last range leaves the stream's table behind. */
template <uint32_t DMAX>
__device__ static void x3s_mtf_body(const X3sMtfArgs &a)
{
	X3_LDS uint32_t tab[DMAX];
	X3_LDS uint16_t lst[DMAX];
	X3_LDS uint16_t pos0[DMAX];
	const uint32_t c = blockIdx.x / a.nranges, r = blockIdx.x % a.nranges, lane = x3_lane();
	const X3Slice sl = a.sl[c];
	const uint32_t *lt = a.lt + a.chunks[c].elem_off;
	uint32_t *lt_out = a.lt_out + a.chunks[c].elem_off;
	const uint32_t e0 = sl.se, e1 = sl.se + (sl.h1 - sl.h0) + (sl.d1 - sl.d0);
	const uint32_t per = (((e1 - e0 + a.nranges - 1) / a.nranges) + X3_WAVE - 1) & ~(X3_WAVE - 1);
	const uint32_t s0 = e0 + r * per < e1 ? e0 + r * per : e1, s1 = s0 + per < e1 ? s0 + per : e1;
	const uint32_t *rows = a.scratch + (size_t)c * a.nranges * (X3S_DMAX + 1);
	const bool lastr = r + 1 == a.nranges;
	if (s0 >= s1 && !lastr) return;
	uint32_t Dcur = sl.d0;
	for (uint32_t w = 0; w < r; w++) Dcur += rows[(size_t)w * (X3S_DMAX + 1) + X3S_DMAX];
	const uint32_t Dend = lastr ? sl.d1 : Dcur; /* the last range also folds its own table in, for the next slice */
	for (uint32_t tb = 0; tb < Dend; tb += X3_WAVE) {
		const uint32_t t = tb + lane;
		uint32_t run = t < sl.d0 ? lt[t] : 0u;
		if (t < Dend) for (uint32_t w = 0; w < r; w++) { const uint32_t v = rows[(size_t)w * (X3S_DMAX + 1) + t]; run = v > run ? v : run; }
		if (t < Dcur) tab[t] = run;
		if (lastr && t < Dend) { const uint32_t v = rows[(size_t)r * (X3S_DMAX + 1) + t]; lt_out[t] = v > run ? v : run; }
	}
	x3_wave_order();
	if (s0 >= s1) return;
	for (uint32_t tb = 0; tb < Dcur; tb += X3_WAVE) {
		const uint32_t t = tb + lane;
		const uint32_t mine = t < Dcur ? tab[t] : 0xFFFFFFFFu;
		uint32_t later = 0;
		for (uint32_t u = 0; u < Dcur; u++) later += tab[u] > mine ? 1u : 0u;
		if (t < Dcur) { lst[later] = (uint16_t)t; pos0[t] = (uint16_t)later; }
	}
	x3_wave_order();
	X3MtfArgs m;
	m.eo = nullptr; m.dof = nullptr; m.e_tag = a.e_tag; m.e_hit = a.e_hit; m.h_rank = a.h_rank;
	x3_mtf_tiles(m, lst, pos0, s0, s1, Dcur, 0, lane);
}

/* ============================================================================================================
 * Context statistics of a slice with CARRIED item lists (context.c:7-56,88-133; ac.c:6-18).  The slice's hits arrive arranged by (context, time):
 * kA = stream << kshift | context, vA = slice-local hit.  A stream's range of the arrangement is cut at context boundaries over `nsub` wavefronts
 * (contexts are independent); a wavefront takes its contexts ONE AT A TIME:
 *   load   the context's header {offset, items, capacity, total} and its items {tag, freq} (first-seen order) from the stream's pool into LDS: tag ->
 *          position, position -> tag / freq / cum_freq (a wave scan)                                                 [empty for a context never seen]
 *   sweep  the context's hits of this slice in tiles of 64: a hit's freq / total / cum_freq as the reference's lists would hold them at that moment
 *          (ballots inside the tile, the LDS tables across tiles: x3_ctxseg_kernel's logic for its one open context), new tags appended
 *   store  the list back (into a block of twice the capacity if it outgrew its own: ctx_enlarge) and the header.
 * ORD (context1 family): an item also carries the ORDINAL of the tag pair (context1, tag) it stands for (x3.c:213-222 add the pair in the step that adds
 * the item), which is what the next step's context0 is (x3.c:139-147).  Items made in this slice get their ordinals from x3s_pairs_kernel (time order
 * over ALL contexts of the stream); until then the item holds 0x80000000 | the slice-local hit that made it, and newaddr[that hit] = its pool slot.
 * Per hit: stat = {freq (0: the tag is not in the context yet), total, cum_freq, ORD: the pair's ordinal or 0x80000000 | making hit}.
 * ============================================================================================================ */
#define X3S_SMALL_K 8u          /* a context with at most this many hits in a tile and items in its list (afterwards) is walked by one lane */
#define X3S_CHUNK 512u          /* entries a wavefront takes from its stream's pool at a time */
#define X3S_POOL_PER_BYTE 6u    /* pool entries per input byte: 4 bound the lists' blocks (every block at most twice its list, every abandoned block at most half the next), */
#define X3S_POOL_EXTRA 65536u   /* ... the rest and this much per stream is for the ends of chunks that were abandoned (a violated bound ends the sliced run, see X3_ST_POOL_FULL) */
struct X3sCtxArgs {
	const X3Chunk *chunks; const X3Slice *sl;
	const uint32_t *kA, *vA, *h_tag;
	uint4 *stat;
	X3CtxHdr *hdr;             /* per stream at elem_off: one header per context (tag / pair ordinal) */
	uint64_t *pool;            /* per stream at X3S_POOL_PER_BYTE * elem_off + stream * X3S_POOL_EXTRA: items tag << 32 | freq */
	uint32_t *pord;            /* ORD: same layout, the pair ordinal of the item */
	uint32_t *newaddr;         /* ORD: per slice hit that made an item: its slot in the stream's pool */
	uint32_t *top;             /* per stream: bump pointer of the pool */
	uint32_t *first00;         /* ORD: per stream, the slice-local hit that registers the pair (0, 0) in this slice (preset to NONE32 by the caller) */
	uint32_t *status;          /* per stream: X3_ST_POOL_FULL if the pool bound was violated (a sizing bug) */
	uint4 *pending;            /* [stream * nsub + wavefront] {context, offset, items | capacity log2 << 27, total}, context == NONE32: nothing.  The wavefront that ends a context which
	                            * was cut leaves the context's new header HERE and its list in a NEW block: the other parts of that context read the header and the old block whenever
	                            * they get to run.  x3s_ctx_apply moves the headers into place behind the kernel. */
	uint32_t *scratch;         /* [stream * nsub + wavefront][2 * sstride + 4]: per tag, the hits of the wavefront's LAST context inside its part of the range and the first of them
	                            * (slice-local hit, NONE32: none) -- published when that context goes on in the next wavefront's part (x3s_ctx_publish_kernel) */
	uint32_t sstride;          /* tags a row holds: the batch's largest dictionary, rounded up to 64 */
	uint32_t *dbg;             /* (experiments, X3H_CTX_DEBUG) per wavefront {kcycles in all, kcycles building the start state, contexts, tiles}; nullptr otherwise */
	uint32_t kshift, kmask, dbits, nsub, nc;
};

/* first index in [lo, hi) whose key is >= key (keys ascend) */
__device__ static __forceinline__ uint32_t s_lower_bound(const uint32_t *kA, uint32_t lo, uint32_t hi, uint32_t key)
{
	while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (kA[mid] < key) lo = mid + 1; else hi = mid; }
	return lo;
}

/* A stream's range of the arrangement is cut into `nsub` EQUAL parts, one wavefront each, wherever the cuts fall: a context that is cut goes through its parts
 * like a stream through time ranges.  Pass 1 (this kernel): a wavefront whose last context goes on behind its part publishes what the later parts need of it:
 * per tag the number of hits and the first hit inside the part.  Pass 2 (x3s_ctx_kernel): a wavefront that starts inside a context adds up the rows of the
 * parts before it -- counts are additive, new tags enter the list in the order of their first hits -- and continues from there. */
template <uint32_t DMAX>
__device__ static void x3s_ctx_publish_body(const X3sCtxArgs &a)
{
	X3_LDS uint32_t cnt[DMAX];
	X3_LDS uint32_t fst[DMAX];
	const uint32_t lane = x3_lane();
	const uint32_t c = blockIdx.x / a.nsub, sub = blockIdx.x % a.nsub;
	const X3Slice sl = a.sl[c];
	const uint32_t c0 = sl.sh, c1 = sl.sh + (sl.h1 - sl.h0);
	const uint32_t per = (c1 - c0 + a.nsub - 1) / a.nsub;
	const uint64_t n0 = (uint64_t)c0 + (uint64_t)sub * per, n1 = n0 + per;
	const uint32_t h0 = n0 < c1 ? (uint32_t)n0 : c1, h1 = n1 < c1 ? (uint32_t)n1 : c1;
	uint32_t *row = a.scratch + ((size_t)c * a.nsub + sub) * (2 * (size_t)a.sstride + 4);
	if (h0 >= h1 || h1 >= c1) { if (lane == 0) row[2 * a.sstride] = 0; return; }
	const uint32_t kl = a.kA[h1 - 1];
	if (a.kA[h1] != kl) { if (lane == 0) row[2 * a.sstride] = 0; return; } /* my last context ends with my part */
	const uint32_t ps = s_lower_bound(a.kA, h0, h1, kl); /* where it starts inside my part */
	const uint32_t D = a.sstride < DMAX ? a.sstride : DMAX;
	for (uint32_t i = lane; i < D; i += X3_WAVE) { cnt[i] = 0; fst[i] = NONE32; }
	x3_wave_order();
	uint32_t nj_ = a.vA[ps + lane < h1 ? ps + lane : h1 - 1], nj2_ = a.vA[ps + X3_WAVE + lane < h1 ? ps + X3_WAVE + lane : h1 - 1];
	uint32_t nt_ = a.h_tag[nj_];
	for (uint32_t base = ps; base < h1; base += X3_WAVE) {
		const bool valid = base + lane < h1;
		const uint32_t j = nj_, t = nt_;
		{
			const uint32_t nx2 = base + 2 * X3_WAVE + lane < h1 ? base + 2 * X3_WAVE + lane : h1 - 1;
			nj_ = nj2_; nt_ = a.h_tag[nj2_];
			nj2_ = a.vA[nx2];
		}
		if (valid) { atomicAdd(&cnt[t], 1u); atomicMin(&fst[t], j); }
	}
	x3_wave_order();
	for (uint32_t i = lane; i < D; i += X3_WAVE) { row[i] = cnt[i]; row[a.sstride + i] = fst[i]; }
	if (lane == 0) row[2 * a.sstride] = h1 - ps;
}

template <uint32_t DMAX, bool ORD>
__device__ static void x3s_ctx_body(const X3sCtxArgs &a)
{
	X3_LDS uint16_t tpos[DMAX];      /* tag -> list position, NONE16: absent */
	X3_LDS uint16_t ltag[DMAX];      /* position -> tag                      */
	X3_LDS uint32_t lfreq[DMAX];     /* position -> freq                     */
	X3_LDS uint32_t lpre[DMAX];      /* position -> cum_freq (valid when !stale) */
	X3_LDS uint32_t lord[ORD ? DMAX : 1]; /* position -> pair ordinal, or 0x80000000 | making hit */
	X3_LDS uint32_t fmin[DMAX];      /* (start inside a context) tag -> its first hit in the earlier parts */
	X3_LDS uint32_t s_t[ORD ? 1 : X3_WAVE], s_j[ORD ? 1 : X3_WAVE];   /* (small contexts) the tile's tags and hits */
	X3_LDS uint32_t s_small[ORD ? 1 : X3_WAVE * 2 * X3S_SMALL_K];     /* (small contexts) per lane: a list of up to X3S_SMALL_K items {tag, freq} */
	const uint32_t lane = x3_lane();
	const uint32_t c = blockIdx.x / a.nsub, sub = blockIdx.x % a.nsub;
	const X3Slice sl = a.sl[c];
	const uint32_t c0 = sl.sh, c1 = sl.sh + (sl.h1 - sl.h0);
	const uint64_t eoff = a.chunks[c].elem_off;
	X3CtxHdr *hdrs = a.hdr + eoff;
	uint64_t *pool = a.pool + X3S_POOL_PER_BYTE * eoff + (uint64_t)c * X3S_POOL_EXTRA;
	uint32_t *pord = ORD ? a.pord + X3S_POOL_PER_BYTE * eoff + (uint64_t)c * X3S_POOL_EXTRA : nullptr;
	const uint32_t pool_cap = X3S_POOL_PER_BYTE * (a.chunks[c].len + 16u) + X3S_POOL_EXTRA;
	const uint64_t bit = (uint64_t)1 << lane, below = bit - 1;
	const int tbits = (int)a.dbits;
	/* my part of the stream's range: equal parts, wherever the cuts fall (see x3s_ctx_publish_body) */
	const uint32_t per = (c1 - c0 + a.nsub - 1) / a.nsub;
	const uint64_t n0 = (uint64_t)c0 + (uint64_t)sub * per, n1 = n0 + per;
	const uint32_t h0 = n0 < c1 ? (uint32_t)n0 : c1, h1 = n1 < c1 ? (uint32_t)n1 : c1;
	if (h0 >= h1) { if (lane == 0) a.pending[(size_t)c * a.nsub + sub] = make_uint4(NONE32, 0, 0, 0); return; }
	const uint64_t dbg_t0 = x3_clock();
	uint32_t dbg_nseg = 0;
	uint64_t dbg_store = 0, dbg_load = 0, dbg_body = 0;
	for (uint32_t i = lane; i < DMAX; i += X3_WAVE) tpos[i] = NONE16;
	x3_wave_order();
	/* the OPEN context: its list is in the LDS tables (wave-uniform bookkeeping) */
	bool open = false, stale = false;
	uint32_t okey = 0, ok0 = 0, ooff = 0, ocap = 0, ok = 0, ototal = 0;
	/* blocks for new / outgrown lists come from a chunk this wavefront takes from the stream's pool in one go: one atomic per X3S_CHUNK entries instead of one
	 * per list (a returning atomic on ONE address per stream serialises: 15 000 of them per slice and stream were most of this kernel's time) */
	uint32_t chunk_at = 0, chunk_end = 0;
	const bool goes_on = h1 < c1 && a.kA[h1] == a.kA[h1 - 1]; /* my last context goes on in the next part: the wavefront that sees its end stores it */
	bool split = false; /* the open context was cut: I am not its first part */
	if (lane == 0) a.pending[(size_t)c * a.nsub + sub] = make_uint4(NONE32, 0, 0, 0);
	if (h0 > c0 && a.kA[h0 - 1] == a.kA[h0]) {
		split = true;
		/* ---- I start INSIDE a context: its list as of my first hit = the carried list + what the earlier parts did to it ---- */
		const uint32_t kf = x3_uniform(a.kA[h0]);
		const uint32_t cs = s_lower_bound(a.kA, c0, h0, kf);     /* the context's first hit of the slice */
		const uint32_t w0 = (cs - c0) / per;                      /* ... lies in this wavefront's part    */
		const X3CtxHdr hd0 = hdrs[kf & a.kmask];
		okey = kf; open = true; ok0 = x3_uniform(hd0.items); ooff = x3_uniform(hd0.off); ocap = x3_uniform(hd0.cap); ototal = x3_uniform(hd0.total) + (h0 - cs);
		for (uint32_t pb = 0; pb < ok0; pb += X3_WAVE) {
			const uint32_t p = pb + lane;
			if (p < ok0) {
				const uint64_t it = pool[(uint64_t)ooff + p];
				const uint32_t tg = (uint32_t)(it >> 32);
				ltag[p] = (uint16_t)tg; lfreq[p] = (uint32_t)it; tpos[tg] = (uint16_t)p; if (ORD) lord[p] = pord[(uint64_t)ooff + p];
			}
		}
		x3_wave_order();
		const uint32_t D = a.sstride < DMAX ? a.sstride : DMAX;
		const size_t rs = 2 * (size_t)a.sstride + 4;
		const uint32_t *rows = a.scratch + (size_t)c * a.nsub * rs;
		uint32_t nnew_total = 0;
		for (uint32_t tb = 0; tb < D; tb += X3_WAVE) { /* per tag: hits in the earlier parts, the first of them (lpre doubles as the counter array here) */
			const uint32_t t = tb + lane;
			uint32_t acc = 0, fj = NONE32;
			if (t < D) for (uint32_t w = w0; w < sub; w++) { const uint32_t *r = rows + (size_t)w * rs; acc += r[t]; const uint32_t f = r[a.sstride + t]; fj = f < fj ? f : fj; }
			const uint32_t cp = t < D ? (uint32_t)tpos[t] : (uint32_t)NONE16;
			if (t < D && acc && cp != NONE16) lfreq[cp] += acc;
			const bool isn = t < D && acc && cp == NONE16;
			if (t < D) { lpre[t] = isn ? acc : 0u; fmin[t] = isn ? fj : NONE32; }
			nnew_total += (uint32_t)x3_popc64(x3_ballot(isn));
		}
		x3_wave_order();
		if (nnew_total) { /* tags the earlier parts added: behind the carried ones, in the order of their first hits */
			for (uint32_t tb = 0; tb < D; tb += X3_WAVE) {
				const uint32_t t = tb + lane;
				const uint32_t mine = t < D ? fmin[t] : NONE32;
				uint32_t before = 0;
				if (mine != NONE32) for (uint32_t u = 0; u < D; u++) before += fmin[u] < mine ? 1u : 0u;
				if (mine != NONE32) { const uint32_t pos = ok0 + before; tpos[t] = (uint16_t)pos; ltag[pos] = (uint16_t)t; lfreq[pos] = lpre[t]; if (ORD) lord[pos] = 0x80000000u | mine; }
			}
		}
		ok = ok0 + nnew_total;
		stale = true; /* the first tile rescans the cum_freqs */
		x3_wave_order();
	}
	/* records of the range in tiles of 64, fetched ahead: keys and hits one tile, the gathered tags one tile (their hits two tiles), and every lane the
	 * header of ITS key's context -- up to 64 headers in flight at once instead of one dependent load per context */
	uint32_t nk_ = 0, nj_ = 0, nt_ = 0, nj2_ = 0;
	X3CtxHdr nhd_; nhd_.off = nhd_.items = nhd_.cap = nhd_.total = 0;
	/* (unconditional loads from clamped indices: behind a branch the compiler could only wait for them with vmcnt(0), i.e. for the loads just issued as well) */
	{
		const uint32_t i0 = h0 + lane < h1 ? h0 + lane : h1 - 1, i1 = h0 + X3_WAVE + lane < h1 ? h0 + X3_WAVE + lane : h1 - 1;
		nk_ = a.kA[i0]; nj_ = a.vA[i0]; nj2_ = a.vA[i1];
		nt_ = a.h_tag[nj_]; nhd_ = hdrs[nk_ & a.kmask];
	}
	for (uint32_t base = h0; base < h1; base += X3_WAVE) {
		const bool valid = base + lane < h1;
		const uint32_t key = nk_, j = nj_, t = valid ? nt_ : 0u;
		const X3CtxHdr hd = nhd_;
		{
			const uint32_t nx = base + X3_WAVE + lane < h1 ? base + X3_WAVE + lane : h1 - 1, nx2 = base + 2 * X3_WAVE + lane < h1 ? base + 2 * X3_WAVE + lane : h1 - 1;
			nk_ = a.kA[nx]; nj_ = nj2_; nt_ = a.h_tag[nj2_]; nhd_ = hdrs[nk_ & a.kmask];
			nj2_ = a.vA[nx2];
		}
		const uint64_t V = x3_ballot(valid);
		const uint32_t nvalid = (uint32_t)x3_popc64(V);
		const uint32_t kprev = wave_prev_u32(key);
		uint64_t S = x3_ballot(valid && (lane == 0 || key != kprev)); /* first lanes of the tile's contexts */
		const uint64_t Sall = S;
		/* the first items of EVERY context of the tile in one go: the i-th lane of a context's hits fetches its item i (most contexts of a tile are small: a
		 * dependent load per context was most of this kernel's time on its busiest wavefronts) */
		const uint32_t sl_ = 63u - (uint32_t)x3_clz64(S & (below | bit)); /* first lane of my context (valid lanes) */
		const uint32_t ii_ = lane - sl_;
		const uint32_t pix = hd.items ? (ii_ < hd.items ? ii_ : hd.items - 1u) : 0u; /* (always a slot of the pool: see above) */
		const uint64_t pit = pool[(uint64_t)hd.off + pix];
		const uint32_t pio = ORD ? pord[(uint64_t)hd.off + pix] : 0u;
		if (!ORD) {
			/* ---- SMALL contexts, all of the tile at once: a context with a few hits in this tile and a short list is walked by ONE LANE -- the first lane of its hits --
			 * in plain serial code (list in the lane's LDS slot, <= X3S_SMALL_K items, <= X3S_SMALL_K hits), every such context of the tile side by side.  The tail of
			 * the context0 arrangement is thousands of pairs with a hit or two each: through the wavefront-wide path below (about a thousand instructions per context,
			 * whatever its size) they made the busiest wavefronts of this kernel 40 times slower than the average one.  The tile's first and last context keep to
			 * that path (they may be the open one / go on in the next tile). ---- */
			const uint64_t Sab = S & ~(below | bit);
			const uint32_t e_ = Sab ? (uint32_t)x3_ctz64(Sab) : nvalid;
			const uint32_t seglen = e_ - sl_, lastseg = 63u - (uint32_t)x3_clz64(S);
			const bool fast = valid && lane == sl_ && sl_ != 0u && sl_ != lastseg && seglen <= X3S_SMALL_K && hd.items + seglen <= X3S_SMALL_K;
			const uint64_t F = x3_ballot(fast);
			if (F) {
				s_t[lane] = t; s_j[lane] = j;
				x3_wave_order();
				uint32_t k = hd.items, total = hd.total, off = hd.off, cap = hd.cap, need = 0;
				uint32_t *my = s_small + lane * 2u * X3S_SMALL_K;
				if (fast) {
					for (uint32_t p = 0; p < k; p++) { const uint64_t it = p == 0 ? pit : pool[(uint64_t)off + p]; my[2 * p] = (uint32_t)(it >> 32); my[2 * p + 1] = (uint32_t)it; }
					for (uint32_t h = 0; h < seglen; h++) {
						const uint32_t tt = s_t[sl_ + h];
						const uint32_t jj = s_j[sl_ + h];
						uint32_t pos = NONE32, cum = 0, fq = 0;
						for (uint32_t p = 0; p < k; p++) { if (my[2 * p] == tt) { pos = p; fq = my[2 * p + 1]; break; } cum += my[2 * p + 1]; }
						if (pos == NONE32) { cum = total; pos = k; my[2 * k] = tt; my[2 * k + 1] = 0; k++; } /* a new item stands behind every other one */
						uint4 r; r.x = fq; r.y = total; r.z = cum; r.w = 0u;
						a.stat[jj] = r;
						my[2 * pos + 1]++; total++;
					}
					if (k > cap) { need = cap ? cap : 2u; while (need < k) need <<= 1; }
				}
				/* blocks for the lists that outgrew theirs: one bump of the wavefront's chunk for the whole tile */
				const uint32_t incl = x3_wave_incl_scan_u32(need);
				const uint32_t tot_need = x3_readlane_u32(incl, X3_WAVE - 1);
				if (tot_need) {
					if (chunk_end - chunk_at < tot_need) {
						const uint32_t want = tot_need > X3S_CHUNK ? tot_need : X3S_CHUNK;
						uint32_t got = 0;
						if (lane == 0) got = atomicAdd(&a.top[c], want);
						chunk_at = x3_uniform(x3_bcast_u32(got, 0)); chunk_end = chunk_at + want;
					}
					if ((uint64_t)chunk_at + tot_need > pool_cap) { if (lane == 0) a.status[c] = X3_ST_POOL_FULL; }
					else if (need) { off = chunk_at + incl - need; cap = need; }
					chunk_at += tot_need;
				}
				if (fast && k <= cap) {
					for (uint32_t p = 0; p < k; p++) pool[(uint64_t)off + p] = ((uint64_t)my[2 * p] << 32) | my[2 * p + 1];
					X3CtxHdr nh; nh.off = off; nh.items = k; nh.cap = cap; nh.total = total;
					hdrs[key & a.kmask] = nh;
				}
				S &= ~F;
				x3_wave_order();
			}
		}
		while (S) {
			dbg_nseg++;
			const uint32_t s = (uint32_t)x3_ctz64(S);
			S &= S - 1;
			const uint64_t Snext = Sall & ~((((uint64_t)1 << s) - 1) | ((uint64_t)1 << s)); /* (contexts that took the small path are not in S any more, but they still end this one) */
			const uint32_t e = Snext ? (uint32_t)x3_ctz64(Snext) : nvalid;
			const uint64_t seg = (e >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << e) - 1)) & ~(((uint64_t)1 << s) - 1);
			const uint32_t key_s = x3_readlane_u32(key, s);
			const uint64_t dq0 = a.dbg ? x3_clock() : 0;
			uint64_t dq1 = dq0;
			if (!(open && key_s == okey)) {
				if (open) { /* ---- store the context that ended ---- */
					uint32_t off = ooff, cap = ocap;
					if (ok > ocap || split) { /* ctx_enlarge: a new block (the old one is abandoned, like a realloc that moved); a context that was cut always moves */
						cap = ocap ? ocap : 2u;
						while (cap < ok) cap <<= 1;
						if (chunk_end - chunk_at < cap) { /* (what is left of the old chunk is abandoned: < X3S_CHUNK entries) */
							const uint32_t want = cap > X3S_CHUNK ? cap : X3S_CHUNK;
							uint32_t got = 0;
							if (lane == 0) got = atomicAdd(&a.top[c], want);
							chunk_at = x3_uniform(x3_bcast_u32(got, 0)); chunk_end = chunk_at + want;
						}
						off = chunk_at; chunk_at += cap;
						if ((uint64_t)off + cap > pool_cap) { if (lane == 0) a.status[c] = X3_ST_POOL_FULL; off = 0; cap = 0; ok = 0; } /* the pool bound was violated: api.hip codes the batch again, stage after stage */
					}
					for (uint32_t p = lane; p < ok; p += X3_WAVE) {
						const uint32_t tg = ltag[p];
						pool[(uint64_t)off + p] = ((uint64_t)tg << 32) | lfreq[p];
						if (ORD) {
							const uint32_t o = lord[p];
							pord[(uint64_t)off + p] = o;
							if (o & 0x80000000u) a.newaddr[o & 0x7FFFFFFFu] = off + p; /* made in this slice: x3s_pairs_kernel writes its ordinal there */
						}
						tpos[tg] = NONE16;
					}
					if (lane == 0) {
						X3CtxHdr nh; nh.off = off; nh.items = ok; nh.cap = cap; nh.total = ototal;
						if (split) a.pending[(size_t)c * a.nsub + sub] = make_uint4(okey & a.kmask, off, ok | (cap ? (uint32_t)(31 - x3_clz32(cap)) << 27 : 0u), ototal);
						else hdrs[okey & a.kmask] = nh;
					}
					split = false; /* (only my first context can be one that was cut before me) */
					x3_wave_order();
				}
				if (a.dbg) dq1 = x3_clock();
				/* ---- load the context that starts (its header came with the tile) ---- */
				okey = key_s; open = true;
				ok0 = x3_readlane_u32(hd.items, s); ooff = x3_readlane_u32(hd.off, s); ocap = x3_readlane_u32(hd.cap, s); ototal = x3_readlane_u32(hd.total, s);
				ok = ok0;
				{
					const uint32_t npre = ok0 < e - s ? ok0 : e - s; /* items [0, npre) came with the tile, in lanes [s, s + npre) */
					if (lane >= s && lane < s + npre) {
						const uint32_t p = lane - s, tg = (uint32_t)(pit >> 32);
						ltag[p] = (uint16_t)tg; lfreq[p] = (uint32_t)pit; tpos[tg] = (uint16_t)p; if (ORD) lord[p] = pio;
					}
					for (uint32_t pb = npre; pb < ok0; pb += X3_WAVE) { /* a long list behind few hits: the rest */
						const uint32_t p = pb + lane;
						if (p < ok0) {
							const uint64_t it = pool[(uint64_t)ooff + p];
							const uint32_t tg = (uint32_t)(it >> 32);
							ltag[p] = (uint16_t)tg; lfreq[p] = (uint32_t)it; tpos[tg] = (uint16_t)p; if (ORD) lord[p] = pord[(uint64_t)ooff + p];
						}
					}
				}
				x3_wave_order();
				{ /* cum_freqs of the list */
					uint32_t carry = 0;
					for (uint32_t pb = 0; pb < ok0; pb += X3_WAVE) {
						const uint32_t p = pb + lane;
						const uint32_t v = p < ok0 ? lfreq[p] : 0u;
						const uint32_t incl = x3_wave_incl_scan_u32(v) + carry;
						if (p < ok0) lpre[p] = incl - v;
						carry = x3_readlane_u32(incl, X3_WAVE - 1);
					}
				}
				stale = false;
				x3_wave_order();
			} else if (stale) { /* the open context goes on in this tile: cum_freqs of its list as this tile sees them */
				uint32_t carry = 0;
				for (uint32_t pb = 0; pb < ok; pb += X3_WAVE) {
					const uint32_t p = pb + lane;
					const uint32_t v = p < ok ? lfreq[p] : 0u;
					const uint32_t incl = x3_wave_incl_scan_u32(v) + carry;
					if (p < ok) lpre[p] = incl - v;
					carry = x3_readlane_u32(incl, X3_WAVE - 1);
				}
				stale = false;
				x3_wave_order();
			}
			const uint64_t dq2 = a.dbg ? x3_clock() : 0;
			/* ---- the context's hits of this tile: lanes [s, e) ---- */
			const bool in = (seg >> lane) & 1u;
			const uint64_t B = seg & below;
			const uint64_t M = wave_same_mask(t, tbits, seg, in);
			const uint64_t E = M & below;
			const uint32_t fl = E ? (uint32_t)x3_ctz64(E) : lane;    /* first lane of my tag in the segment */
			const uint32_t cp = in ? (uint32_t)tpos[t] : (uint32_t)NONE16;
			const bool known = cp != NONE16;
			const bool isnew = in && fl == lane && !known;           /* this hit adds the tag to the context */
			const uint64_t N = x3_ballot(isnew);
			uint32_t pos = known ? cp : ok + (uint32_t)x3_popc64(N & below);
			const uint32_t pos_fl = x3_bcast_u32(pos, (int)fl);
			if (in && !known && fl != lane) pos = pos_fl;
			const uint32_t freq = (known ? lfreq[cp] : 0u) + (uint32_t)x3_popc64(E);
			const uint32_t cbase = known ? lpre[cp] : ototal; /* a new item stands behind every carried one */
			const uint32_t cum = cbase + wave_count_less(pos, pos, tbits, B);
			const uint32_t j_fl = x3_bcast_u32(j, (int)fl);
			if (in) {
				uint4 r;
				r.x = freq; r.y = ototal + (lane - s); r.z = cum;
				r.w = ORD ? (known ? lord[cp] : (0x80000000u | j_fl)) : 0u;
				a.stat[j] = r;
				if (ORD && isnew && t == 0u && (key_s & a.kmask) == 0u) a.first00[c] = j; /* x3.c:424-425: both contexts after a new fragment */
			}
			x3_wave_order();
			if (isnew) { tpos[t] = (uint16_t)pos; ltag[pos] = (uint16_t)t; lfreq[pos] = 0; if (ORD) lord[pos] = 0x80000000u | j; }
			x3_wave_order();
			if (in) atomicAdd(&lfreq[pos], 1u);
			ok += (uint32_t)x3_popc64(N);
			ototal += e - s;
			stale = true;
			x3_wave_order();
			if (a.dbg) { const uint64_t dq3 = x3_clock(); dbg_store += dq1 - dq0; dbg_load += dq2 - dq1; dbg_body += dq3 - dq2; }
		}
	}
	if (open && !goes_on) { /* the last context of my part ends there: its final list goes back to the pool */
		uint32_t off = ooff, cap = ocap;
		if (ok > ocap || split) {
			cap = ocap ? ocap : 2u;
			while (cap < ok) cap <<= 1;
			if (chunk_end - chunk_at < cap) {
				const uint32_t want = cap > X3S_CHUNK ? cap : X3S_CHUNK;
				uint32_t got = 0;
				if (lane == 0) got = atomicAdd(&a.top[c], want);
				chunk_at = x3_uniform(x3_bcast_u32(got, 0)); chunk_end = chunk_at + want;
			}
			off = chunk_at; chunk_at += cap;
			if ((uint64_t)off + cap > pool_cap) { if (lane == 0) a.status[c] = X3_ST_POOL_FULL; off = 0; cap = 0; ok = 0; }
		}
		for (uint32_t p = lane; p < ok; p += X3_WAVE) {
			const uint32_t tg = ltag[p];
			pool[(uint64_t)off + p] = ((uint64_t)tg << 32) | lfreq[p];
			if (ORD) {
				const uint32_t o = lord[p];
				pord[(uint64_t)off + p] = o;
				if (o & 0x80000000u) a.newaddr[o & 0x7FFFFFFFu] = off + p;
			}
		}
		if (lane == 0) {
			X3CtxHdr nh; nh.off = off; nh.items = ok; nh.cap = cap; nh.total = ototal;
			if (split) a.pending[(size_t)c * a.nsub + sub] = make_uint4(okey & a.kmask, off, ok | (cap ? (uint32_t)(31 - x3_clz32(cap)) << 27 : 0u), ototal);
			else hdrs[okey & a.kmask] = nh;
		}
	}
	(void)ok0;
	if (a.dbg && lane == 0) { uint32_t *d = a.dbg + 4 * ((size_t)c * a.nsub + sub); d[0] = (uint32_t)((x3_clock() - dbg_t0) >> 10); d[1] = (uint32_t)(dbg_store >> 10) | (uint32_t)(dbg_load >> 10) << 16; d[2] = dbg_nseg; d[3] = (uint32_t)(dbg_body >> 10); }
}

/* ============================================================================================================
 * Tag-pair ordinals of a slice (tag_pair.c:100-130: ordinal = order of registration) and the context0 of every hit (x3.c:139-147).  One workgroup per
 * stream: (1) the hits that registered a pair in this slice, counted in TIME order behind the pairs of the earlier slices; (2) the items they made get
 * their ordinals, every hit learns the ordinal of ITS pair (context1, tag); (3) context0 of a hit = the pair the previous hit registered or met
 * (prev_context1, context1), or -- after a new fragment -- the pair (0, 0) once that exists, else ordinal 0 ("not found: default to 0", x3.c:142-145).
 * Carried per stream: pairs so far, the ordinal of (0, 0) or NONE32, the ordinal of the last hit's pair.
 * ============================================================================================================ */
#define X3S_PAIR_THREADS 1024u
struct X3sPairArgs {
	const X3Chunk *chunks; const X3Slice *sl;
	uint4 *stat1;              /* in: .w as x3s_ctx left it; out: .w = the ordinal of the hit's pair */
	const uint32_t *newaddr, *h_pv;
	uint32_t *pord;            /* the context1 items' ordinals (pool layout) */
	const uint32_t *first00;   /* per stream: the slice-local hit that registered (0, 0) in this slice, or NONE32 */
	uint32_t *npairs, *ord00, *last_ord; /* carried per stream */
	uint32_t *k0;              /* out per slice hit: arrangement key stream << kshift | context0 */
	uint32_t kshift;
};

__device__ static void x3s_pairs_body(const X3sPairArgs &a)
{
	/* (1) one workgroup per stream: the slice's pair-registering hits numbered in time order (a running count carried from tile to tile) */
	X3_LDS uint32_t s_w[X3S_PAIR_THREADS / X3_WAVE];
	const uint32_t NW = X3S_PAIR_THREADS / X3_WAVE;
	const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = x3_lane(), wave = tid / X3_WAVE;
	const X3Slice sl = a.sl[c];
	const uint32_t j0 = sl.sh, j1 = sl.sh + (sl.h1 - sl.h0);
	uint32_t *pord = a.pord + X3S_POOL_PER_BYTE * a.chunks[c].elem_off + (uint64_t)c * X3S_POOL_EXTRA;
	const uint64_t below = ((uint64_t)1 << lane) - 1;
	uint32_t run = a.npairs[c];
	uint32_t nw_ = j0 < j1 ? a.stat1[j0 + tid < j1 ? j0 + tid : j1 - 1].w : 0u; /* the next tile's words are in flight while this one is counted */
	for (uint32_t tb = j0; tb < j1; tb += X3S_PAIR_THREADS) {
		const uint32_t j = tb + tid;
		const bool in = j < j1;
		const uint32_t w = nw_;
		nw_ = a.stat1[j + X3S_PAIR_THREADS < j1 ? j + X3S_PAIR_THREADS : j1 - 1].w;
		const bool isnew = in && w == (0x80000000u | j);
		const uint64_t Nm = x3_ballot(isnew);
		if (lane == 0) s_w[wave] = (uint32_t)x3_popc64(Nm);
		__syncthreads();
		uint32_t before = 0, tot = 0;
		for (uint32_t q = 0; q < NW; q++) { if (q < wave) before += s_w[q]; tot += s_w[q]; }
		if (isnew) {
			const uint32_t o = run + before + (uint32_t)x3_popc64(Nm & below);
			pord[a.newaddr[j]] = o;
			a.stat1[j].w = o;
		}
		run += tot;
		__syncthreads();
	}
	if (tid == 0) a.npairs[c] = run;
}

/* ============================================================================================================
 * model_index1 as the IDX1-coded hits of a slice see it (x3.c:187-188; code3.hip x3_idxstat_kernel for whole streams).  Carried per stream: one count
 * per rank (hist[elem_off + rank] = earlier IDX1 hits with that rank) and the number of IDX1 hits so far (mode chain: evfinal[4c + 3]).
 * ============================================================================================================ */
struct X3sIdxArgs {
	const X3Chunk *chunks; const X3Slice *sl;
	const uint32_t *nidx_before, *evfinal; /* per stream: IDX1 hits before the slice (saved by the caller before the mode chain ran) / after */
	const uint32_t *lrank, *lhit, *h_dk;
	uint32_t *rfreq, *rcum, *itot;
	uint32_t *hist;
	uint32_t dbits;
};

template <uint32_t DMAX>
__device__ static void x3s_idxstat_body(const X3sIdxArgs &a)
{
	X3_LDS uint32_t hist[DMAX];
	X3_LDS uint32_t pre[DMAX];
	const uint32_t c = blockIdx.x, lane = x3_lane();
	const X3Slice sl = a.sl[c];
	uint32_t *gh = a.hist + a.chunks[c].elem_off;
	const uint32_t n0 = a.nidx_before[c], n = a.evfinal[4 * c + 3] - n0, D = sl.d1;
	const uint64_t below = ((uint64_t)1 << lane) - 1;
	const int bits = (int)a.dbits;
	{
		uint32_t carry = 0;
		for (uint32_t pb = 0; pb < D; pb += X3_WAVE) {
			const uint32_t q = pb + lane;
			const uint32_t h = q < sl.d0 ? gh[q] : 0u;
			const uint32_t incl = x3_wave_incl_scan_u32(h) + carry;
			if (q < D) { hist[q] = h; pre[q] = incl - h; }
			carry = x3_readlane_u32(incl, X3_WAVE - 1);
		}
	}
	x3_wave_sync();
	for (uint32_t base = 0; base < n; base += X3_WAVE) {
		const bool valid = base + lane < n;
		const uint32_t r = valid ? a.lrank[sl.sh + base + lane] : 0u, hit = valid ? a.lhit[sl.sh + base + lane] : 0u;
		const uint64_t V = x3_ballot(valid);
		const uint64_t M = wave_same_mask(r, bits, V, valid);
		const uint32_t sm = wave_count_less(r, r, bits, below & V);
		if (valid) {
			a.rfreq[hit] = 1u + hist[r] + (uint32_t)x3_popc64(M & below);
			a.rcum[hit] = r + pre[r] + sm;
			a.itot[hit] = a.h_dk[hit] + n0 + base + lane;
		}
		x3_wave_sync();
		if (valid) atomicAdd(&hist[r], 1u);
		x3_wave_sync();
		if (base + X3_WAVE < n) {
			uint32_t carry = 0;
			for (uint32_t pb = 0; pb < D; pb += X3_WAVE) {
				const uint32_t q = pb + lane;
				const uint32_t h = q < D ? hist[q] : 0u;
				const uint32_t incl = x3_wave_incl_scan_u32(h) + carry;
				if (q < D) pre[q] = incl - h;
				carry = x3_readlane_u32(incl, X3_WAVE - 1);
			}
			x3_wave_sync();
		}
	}
	for (uint32_t q = lane; q < D; q += X3_WAVE) gh[q] = hist[q];
}

/* ============================================================================================================
 * The two adaptive order-0 models of new fragments (x3.c:259-267; code3.hip x3_order0_kernel for whole streams), one slice: per value the earlier
 * equal / smaller values of the STREAM.  Carried: the 32 + 256 counters per stream (o0hist[c * 288 ..]).  One wavefront per (stream, model).
 * ============================================================================================================ */
struct X3sOrder0Args { const X3Slice *sl; const uint32_t *lval, *bval; uint32_t *lsm, *leq, *bsm, *beq; uint32_t *o0hist; uint32_t nc; };

__device__ static void x3s_order0_body(const X3sOrder0Args &a)
{
	X3_LDS uint32_t hist[256];
	X3_LDS uint32_t pre[256];
	const uint32_t which = blockIdx.x >= a.nc ? 1u : 0u, c = blockIdx.x - which * a.nc, lane = x3_lane();
	const X3Slice sl = a.sl[c];
	const int bits = which ? 8 : 5;
	const uint32_t A = which ? 256u : 32u;
	const uint32_t i0 = which ? sl.sb : sl.sm;
	const uint32_t n = which ? sl.mb1 - sl.mb0 : (sl.t1 - sl.t0) - (sl.h1 - sl.h0);
	const uint32_t *val = which ? a.bval : a.lval;
	uint32_t *osm = which ? a.bsm : a.lsm, *oeq = which ? a.beq : a.leq;
	uint32_t *gh = a.o0hist + (size_t)c * 288u + (which ? 32u : 0u);
	const uint64_t below = ((uint64_t)1 << lane) - 1;
	{
		uint32_t carry = 0;
		for (uint32_t pb = 0; pb < 256; pb += X3_WAVE) {
			const uint32_t h = pb + lane < A ? gh[pb + lane] : 0u;
			const uint32_t incl = x3_wave_incl_scan_u32(h) + carry;
			hist[pb + lane] = h; pre[pb + lane] = incl - h;
			carry = x3_readlane_u32(incl, X3_WAVE - 1);
		}
	}
	x3_wave_sync();
	for (uint32_t base = 0; base < n; base += X3_WAVE) {
		const bool valid = base + lane < n;
		const uint32_t v = valid ? val[i0 + base + lane] : 0u;
		const uint64_t V = x3_ballot(valid);
		const uint64_t M = wave_same_mask(v, bits, V, valid);
		const uint32_t sm = wave_count_less(v, v, bits, below & V);
		if (valid) { oeq[i0 + base + lane] = hist[v] + (uint32_t)x3_popc64(M & below); osm[i0 + base + lane] = pre[v] + sm; }
		x3_wave_sync();
		if (valid) atomicAdd(&hist[v], 1u);
		x3_wave_sync();
		uint32_t carry = 0;
		for (uint32_t pb = 0; pb < A; pb += X3_WAVE) {
			const uint32_t h = hist[pb + lane];
			const uint32_t incl = x3_wave_incl_scan_u32(h) + carry;
			pre[pb + lane] = incl - h;
			carry = x3_readlane_u32(incl, X3_WAVE - 1);
		}
		x3_wave_sync();
	}
	for (uint32_t q = lane; q < A; q += X3_WAVE) gh[q] = hist[q];
}

#ifndef X3_EMU
__global__ void __launch_bounds__(X3S_TOK_THREADS) x3s_tokens_kernel(X3sTokArgs a) { x3s_tokens_body(a); }
__global__ void __launch_bounds__(X3_WAVE) x3s_mtf_scan_kernel_t(X3sMtfArgs a) { x3s_mtf_scan_body<512>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3s_mtf_scan_kernel_s(X3sMtfArgs a) { x3s_mtf_scan_body<2048>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3s_mtf_kernel_t(X3sMtfArgs a) { x3s_mtf_body<512>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3s_mtf_kernel_s(X3sMtfArgs a) { x3s_mtf_body<2048>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3s_ctx_publish_kernel_t(X3sCtxArgs a) { x3s_ctx_publish_body<512>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3s_ctx_publish_kernel_s(X3sCtxArgs a) { x3s_ctx_publish_body<2048>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3s_ctx1_kernel_t(X3sCtxArgs a) { x3s_ctx_body<512, true>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3s_ctx1_kernel_s(X3sCtxArgs a) { x3s_ctx_body<2048, true>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3s_ctx0_kernel_t(X3sCtxArgs a) { x3s_ctx_body<512, false>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3s_ctx0_kernel_s(X3sCtxArgs a) { x3s_ctx_body<2048, false>(a); }
__global__ void __launch_bounds__(X3S_PAIR_THREADS) x3s_pairs_kernel(X3sPairArgs a) { x3s_pairs_body(a); }
__global__ void __launch_bounds__(X3_WAVE) x3s_idxstat_kernel_t(X3sIdxArgs a) { x3s_idxstat_body<512>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3s_idxstat_kernel_s(X3sIdxArgs a) { x3s_idxstat_body<2048>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3s_order0_kernel(X3sOrder0Args a) { x3s_order0_body(a); }
#define X3S_LAUNCH(kern, args, grid, block, st) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, st, args)
#else
static void stok_tramp(void *p) { x3s_tokens_body(*(const X3sTokArgs *)p); }
static void smtfscan_tramp_t(void *p) { x3s_mtf_scan_body<512>(*(const X3sMtfArgs *)p); }
static void smtfscan_tramp_s(void *p) { x3s_mtf_scan_body<2048>(*(const X3sMtfArgs *)p); }
static void smtf_tramp_t(void *p) { x3s_mtf_body<512>(*(const X3sMtfArgs *)p); }
static void smtf_tramp_s(void *p) { x3s_mtf_body<2048>(*(const X3sMtfArgs *)p); }
static void sctxpub_tramp_t(void *p) { x3s_ctx_publish_body<512>(*(const X3sCtxArgs *)p); }
static void sctxpub_tramp_s(void *p) { x3s_ctx_publish_body<2048>(*(const X3sCtxArgs *)p); }
static void sctx1_tramp_t(void *p) { x3s_ctx_body<512, true>(*(const X3sCtxArgs *)p); }
static void sctx1_tramp_s(void *p) { x3s_ctx_body<2048, true>(*(const X3sCtxArgs *)p); }
static void sctx0_tramp_t(void *p) { x3s_ctx_body<512, false>(*(const X3sCtxArgs *)p); }
static void sctx0_tramp_s(void *p) { x3s_ctx_body<2048, false>(*(const X3sCtxArgs *)p); }
static void spairs_tramp(void *p) { x3s_pairs_body(*(const X3sPairArgs *)p); }
static void sidx_tramp_t(void *p) { x3s_idxstat_body<512>(*(const X3sIdxArgs *)p); }
static void sidx_tramp_s(void *p) { x3s_idxstat_body<2048>(*(const X3sIdxArgs *)p); }
static void sorder0_tramp(void *p) { x3s_order0_body(*(const X3sOrder0Args *)p); }
#define x3s_tokens_kernel stok_tramp
#define x3s_mtf_scan_kernel_t smtfscan_tramp_t
#define x3s_mtf_scan_kernel_s smtfscan_tramp_s
#define x3s_mtf_kernel_t smtf_tramp_t
#define x3s_mtf_kernel_s smtf_tramp_s
#define x3s_ctx_publish_kernel_t sctxpub_tramp_t
#define x3s_ctx_publish_kernel_s sctxpub_tramp_s
#define x3s_ctx1_kernel_t sctx1_tramp_t
#define x3s_ctx1_kernel_s sctx1_tramp_s
#define x3s_ctx0_kernel_t sctx0_tramp_t
#define x3s_ctx0_kernel_s sctx0_tramp_s
#define x3s_pairs_kernel spairs_tramp
#define x3s_idxstat_kernel_t sidx_tramp_t
#define x3s_idxstat_kernel_s sidx_tramp_s
#define x3s_order0_kernel sorder0_tramp
#define X3S_LAUNCH(kern, args, grid, block, st) x3emu_launch(kern, (void *)&(args), dim3(grid), dim3(block))
#endif

/* ============================================================================================================
 * Host side: the workspace of a sliced run and one slice through the feature stages.
 * ============================================================================================================ */
/* the headers of the contexts that were cut go into place behind the kernel (X3sCtxArgs::pending) */
static int x3s_ctx_apply(hipStream_t st, const X3sCtxArgs &ca, const X3Chunk *d_chunks)
{
	if (ca.nsub <= 1) return X3H_OK;
	const uint4 *pend = ca.pending;
	X3CtxHdr *hdr = ca.hdr;
	const uint32_t nsub = ca.nsub;
	x3_foreach((size_t)ca.nc * nsub, st, X3_LAMBDA(size_t i) {
		const uint4 p = pend[i];
		if (p.x == NONE32) return;
		X3CtxHdr nh; nh.off = p.y; nh.items = p.z & 0x07FFFFFFu; nh.cap = (p.z >> 27) ? 1u << (p.z >> 27) : (nh.items ? 1u : 0u); nh.total = p.w;
		hdr[d_chunks[i / nsub].elem_off + p.x] = nh;
	});
	HIPCHK(hipGetLastError());
	return X3H_OK;
}
int x3s_begin(X3SliceRun &R, hipStream_t st, uint32_t nc, const X3Chunk *h_chunks, uint64_t max_slice_steps, uint64_t max_slice_bytes)
{
	R.nc = nc;
	uint64_t elems = 0;
	for (uint32_t c = 0; c < nc; c++) elems = h_chunks[c].elem_off + h_chunks[c].len + 16;
	if (elems >= ((uint64_t)1 << 29)) return X3H_E_ARG; /* symbol slots 3 * elems stay below 2^31 */
	R.elems = elems;
	/* carried state */
	CHK(R.lt.reserve(elems * 4)); CHK(R.lt2.reserve(elems * 4)); R.lt_flip = false; CHK(R.idxfreq.reserve(elems * 4)); CHK(R.idxhist.reserve(elems * 4));
	CHK(R.hdr1.reserve(elems * sizeof(X3CtxHdr))); CHK(R.hdr0.reserve(elems * sizeof(X3CtxHdr)));
	const uint64_t pool_entries = X3S_POOL_PER_BYTE * elems + (uint64_t)nc * X3S_POOL_EXTRA;
	CHK(R.pool1.reserve(pool_entries * 8)); CHK(R.pord1.reserve(pool_entries * 4)); CHK(R.pool0.reserve(pool_entries * 8));
	CHK(R.sym.reserve((3 * elems + X3_SYM_PAD) * 16)); CHK(R.states.reserve((3 * elems + 8) * 8));
	CHK(R.small.reserve((size_t)nc * X3S_SMALL_WORDS * 4 + 64));
	CHK(R.mtf_scratch.reserve((size_t)nc * X3S_MTF_RANGES * (X3S_DMAX + 1) * 4));
	CHK(R.ctx_pending.reserve((size_t)nc * 1024 * 16));
	CHK(R.ctx_scratch.reserve(((size_t)max_slice_steps / 256 + 2 * (size_t)nc) * (2 * 512 + 4) * 4)); /* rows of x3s_ctx_publish_kernel for dictionaries up to 512 elements */
	HIPCHK(hipMemsetAsync(R.idxhist.p, 0, elems * 4, st)); /* a slice without an IDX1 hit leaves the new elements' counters untouched: they start here */
	{ uint32_t *f = R.idxfreq.as<uint32_t>(); x3_foreach((size_t)elems, st, X3_LAMBDA(size_t i) { f[i] = 1u; }); } /* model_enlarge: a new symbol has frequency 1 (ac.c:250-266) */
	HIPCHK(hipMemsetAsync(R.hdr1.p, 0, elems * sizeof(X3CtxHdr), st));
	HIPCHK(hipMemsetAsync(R.hdr0.p, 0, elems * sizeof(X3CtxHdr), st));
	HIPCHK(hipMemsetAsync(R.small.p, 0, (size_t)nc * X3S_SMALL_WORDS * 4, st));
	{
		uint32_t *sm = R.small.as<uint32_t>();
		const uint32_t n = nc;
		x3_foreach(nc, st, X3_LAMBDA(size_t c) {
			uint32_t *ev = sm + X3S_EVFINAL * n + 4 * c;
			ev[0] = 1024; ev[1] = 1024; ev[2] = 1; ev[3] = 0;                       /* create(), x3.c:236-244 */
			sm[X3S_ORD00 * n + c] = NONE32;
			sm[X3S_CODER * n + 2 * c] = 0u; sm[X3S_CODER * n + 2 * c + 1] = 0x80000000u; /* ac_init, ac.c:35-41 */
		});
	}
	/* temporaries of a slice: dense over the slice's steps */
	const size_t ns = (size_t)max_slice_steps + 64, nb = (size_t)max_slice_bytes + 64;
	for (int q = 0; q < 2; q++) {
		for (int i = 0; i < X3S_NARR; i++) CHK(R.a[q][i].reserve(ns * 4));
		CHK(R.stat1[q].reserve(ns * 16)); CHK(R.stat0[q].reserve(ns * 16));
		for (int i = 0; i < 3; i++) CHK(R.b[q][i].reserve(nb * 4));
	}
	CHK(R.est_val.reserve((3 * nb + 8) * 4)); CHK(R.est_cls.reserve(3 * nb + 8));
	CHK(R.tmp.reserve(ns * 16 + ((size_t)4 << 20)));
	CHK(R.tables.reserve((size_t)(X3S_MAX_SLICES + 2) * nc * (sizeof(X3Slice) + 8) + 64));
	R.slices.clear();
	return X3H_OK;
}

/* the feature stages of one slice, queued on `st` (the move-to-front ranks on `side`); on return the slice's new symbols are in the operand array and
 * R's per-stream {first, count} of the coder segment are set on the device: the caller queues the coder and the bit emission behind `st` */
int x3s_slice(X3SliceRun &R, hipStream_t st, hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join, const X3Chunk *d_chunks, const std::vector<X3Slice> &hs,
              uint64_t max_dict, const uint8_t *d_bytes, const uint32_t *tok_info, const uint8_t *dict_len, bool last, bool want_est, uint32_t **seg_off_out, uint32_t **seg_len_out, int phase)
{
	/* phase 1 = stage A (records, ranks, context statistics), phase 2 = stage B (mode chain, index / order-0 models, symbol assembly), 3 = both.  The two stages of a
	 * slice use one of TWO sets of slice temporaries (slice number & 1), so that stage A of slice k + 1 may run while stage B of slice k still reads its own set. */
	const uint32_t nc = R.nc;
	if (hs.size() != nc || max_dict > X3S_DMAX) return X3H_E_INTERNAL;
	if (phase & 1) R.slices.push_back(hs);            /* (kept alive: the copy below is asynchronous) */
	if (R.slices.empty()) return X3H_E_INTERNAL;
	const std::vector<X3Slice> &keep = R.slices.back();
	const size_t k = R.slices.size() - 1;
	const int set = (int)(k & 1);
	if (k >= X3S_MAX_SLICES + 2) return X3H_E_INTERNAL;
	if (R.tables.cap < (size_t)(X3S_MAX_SLICES + 2) * nc * (sizeof(X3Slice) + 8)) return X3H_E_INTERNAL; /* (x3s_begin sized it) */
	X3Slice *d_sl = R.tables.as<X3Slice>() + k * nc;
	/* {first symbol, count} of every stream's coder segment of THIS slice: a slot per slice (the coder and the bit emission of slice k read them on their own
	 * HIP streams while this stream is already assembling slice k + 1) */
	uint32_t *m_segoff = (uint32_t *)(R.tables.as<X3Slice>() + (size_t)(X3S_MAX_SLICES + 2) * nc) + 2 * k * nc, *m_seglen = m_segoff + nc;
	*seg_off_out = m_segoff; *seg_len_out = m_seglen;
	if (phase & 1) HIPCHK(hipMemcpyAsync(d_sl, keep.data(), (size_t)nc * sizeof(X3Slice), hipMemcpyHostToDevice, st));
	uint64_t nS = 0, nH = 0, nE = 0, nM = 0, nB = 0, maxH = 0;
	for (uint32_t c = 0; c < nc; c++) {
		const X3Slice &s = hs[c];
		nS += s.t1 - s.t0; nH += s.h1 - s.h0; nE += (s.h1 - s.h0) + (s.d1 - s.d0); nM += (s.t1 - s.t0) - (s.h1 - s.h0); nB += s.mb1 - s.mb0;
		if (s.h1 > maxH) maxH = s.h1;
	}
	uint32_t *A[X3S_NARR];
	for (int i = 0; i < X3S_NARR; i++) { if (R.a[set][i].cap < (nS + 8) * 4) return X3H_E_INTERNAL; A[i] = R.a[set][i].as<uint32_t>(); }
	if (R.b[set][0].cap < (nB + 8) * 4) return X3H_E_INTERNAL;
	uint32_t *s_hb = A[0], *s_mb = A[1], *h_tag = A[2], *h_c1 = A[3], *h_pv = A[4], *h_dk = A[5], *h_step = A[6], *k1 = A[7], *e_tag = A[8], *e_hit = A[9];
	uint32_t *lval = A[10], *lsm = A[11], *leq = A[12], *h_rank = A[13], *iota = A[14], *kA = A[15], *vA = A[16], *newaddr = A[17], *k0 = A[18];
	uint32_t *mode = A[19], *pe0 = A[20], *pe1 = A[21], *nzl = A[22], *il_rank = A[23], *il_hit = A[24], *rfreq = A[25], *rcum = A[26], *itot = A[27];
	uint32_t *bval = R.b[set][0].as<uint32_t>(), *bsm = R.b[set][1].as<uint32_t>(), *beq = R.b[set][2].as<uint32_t>();
	uint4 *stat1 = R.stat1[set].as<uint4>(), *stat0 = R.stat0[set].as<uint4>();
	uint32_t *sm = R.small.as<uint32_t>();
	uint32_t *m_evfinal = sm + X3S_EVFINAL * nc, *m_nnoop = sm + X3S_NNOOP * nc, *m_npairs = sm + X3S_NPAIRS * nc, *m_ord00 = sm + X3S_ORD00 * nc;
	uint32_t *m_lastord = sm + X3S_LASTORD * nc, *m_ycnt = sm + X3S_YCNT * nc, *m_ydone = sm + X3S_YDONE * nc, *m_top1 = sm + X3S_TOP1 * nc, *m_top0 = sm + X3S_TOP0 * nc;
	uint32_t *m_status = sm + X3S_STATUS * nc, *m_first00 = sm + X3S_FIRST00 * nc, *m_nidx0 = sm + X3S_NIDX0 * nc, *m_o0 = sm + X3S_O0HIST * nc;
	uint32_t *m_ntok = sm + X3S_NTOK * nc, *m_nhits = sm + X3S_NHITS * nc;
	uint32_t *m_estfirst = sm + X3S_ESTFIRST * nc, *m_estcnt = sm + X3S_ESTCNT * nc;
	const uint32_t dsh = (uint32_t)bits_for64(max_dict ? max_dict : 1);          /* bits of a local tag */
	const uint32_t psh = (uint32_t)bits_for64(maxH ? maxH : 1);                  /* bits of a pair ordinal (pairs <= hits) */
	const int cb = bits_for64(nc ? nc - 1 : 0);
	if (dsh + cb > 32 || psh + cb > 32) return X3H_E_ARG;
	const bool small = max_dict <= 512;

	if (phase & 1) {
	if (nS) {
		X3sTokArgs ta;
		ta.chunks = d_chunks; ta.sl = d_sl; ta.bytes = d_bytes; ta.tok_info = tok_info; ta.dict_len = dict_len; ta.s_hb = s_hb; ta.s_mb = s_mb;
		ta.h_tag = h_tag; ta.h_c1 = h_c1; ta.h_pv = h_pv; ta.h_dk = h_dk; ta.h_step = h_step; ta.k1 = k1; ta.e_tag = e_tag; ta.e_hit = e_hit; ta.lval = lval; ta.bval = bval;
		ta.kshift = dsh;
		X3S_LAUNCH(x3s_tokens_kernel, ta, nc, X3S_TOK_THREADS, st);
		HIPCHK(hipGetLastError());
	}
	if (nE) {
		/* move-to-front ranks beside the context statistics (a slice that only inserts elements still moves the list) */
		HIPCHK(hipEventRecord(ev_fork, st));
		HIPCHK(hipStreamWaitEvent(side, ev_fork, 0));
		X3sMtfArgs ma;
		ma.chunks = d_chunks; ma.sl = d_sl; ma.e_tag = e_tag; ma.e_hit = e_hit; ma.h_rank = h_rank; ma.scratch = R.mtf_scratch.as<uint32_t>();
		ma.lt = (R.lt_flip ? R.lt2 : R.lt).as<uint32_t>(); ma.lt_out = (R.lt_flip ? R.lt : R.lt2).as<uint32_t>(); R.lt_flip = !R.lt_flip;
		/* time ranges of ~1024 events, one wavefront each (a range costs its events plus building its list: elements^2 / 64 steps) */
		uint64_t rw = (nE / nc + 1023) / 1024;
		ma.nranges = rw < 1 ? 1u : rw > X3S_MTF_RANGES ? X3S_MTF_RANGES : (uint32_t)rw;
		if (const char *e = getenv("X3H_SLICE_MTF_RANGES")) { const int v = atoi(e); if (v >= 1 && v <= (int)X3S_MTF_RANGES) ma.nranges = (uint32_t)v; }
		if (small) { X3S_LAUNCH(x3s_mtf_scan_kernel_t, ma, nc * ma.nranges, X3_WAVE, side); X3S_LAUNCH(x3s_mtf_kernel_t, ma, nc * ma.nranges, X3_WAVE, side); }
		else { X3S_LAUNCH(x3s_mtf_scan_kernel_s, ma, nc * ma.nranges, X3_WAVE, side); X3S_LAUNCH(x3s_mtf_kernel_s, ma, nc * ma.nranges, X3_WAVE, side); }
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventRecord(ev_join, side));
	}
	if (nH) {
		/* context1: arrangement, statistics with carried lists */
		/* arrangement by (context1, time).  Small slices: one workgroup per stream, a counting sort on the stream-local key in LDS (x3_arrange_kernel, code3.hip: one
		 * launch per 11 key bits; a slice's arrays are L2-resident, so its scattered stores cost nothing -- on whole streams they made this kernel lose against
		 * the library sort).  Large slices: the chip-wide radix sort (a workgroup of four wavefronts per stream would be the bottleneck). */
		bool arrange = nH / nc <= 32768, segsorted = false, seg_big = false;
		if (const char *e = getenv("X3H_SLICE_SEGSORT")) seg_big = e[0] == '1';
		if (const char *e = getenv("X3H_SLICE_ARRANGE")) arrange = e[0] == '1';
		uint32_t *d_aho = A[30], *d_akb = A[31]; /* (nc + 1 <= steps + 8 entries each) */
		if ((uint64_t)nc + 1 > nS + 8) arrange = false;
		if (arrange) {
			const uint32_t nHs = (uint32_t)nH, ksh = dsh;
			x3_foreach((size_t)nc + 1, st, X3_LAMBDA(size_t c) { d_aho[c] = c < nc ? d_sl[c].sh : nHs; d_akb[c] = (uint32_t)c << ksh; });
			CHK(x3_arrange_run(st, nc, d_aho, d_akb, max_dict ? max_dict - 1 : 0, k1, nullptr, kA, vA, nullptr, A[28], A[29]));
		} else if (seg_big && (uint64_t)nc + 1 <= nS + 8) {
			/* X3H_SLICE_SEGSORT=1 (tests / A-B): one workgroup per stream sorts its segment on the stream-local key in tiles of 4096 (x3_segsort_kernel, code3.hip).  Not the
			 * default for large slices: a lone workgroup needs ~0.4 ms per 256 K hits and pass, and a slice of config 4's share is 2.9 M hits per stream -- the feature stages of a
			 * slice (96 ms) then outlast the coder segment before it (profiles/r05_config4_timeline.txt); the chip-wide sort of prims.hip does all 16 streams in ~2 ms */
			const uint32_t nHs = (uint32_t)nH, ksh = dsh;
			x3_foreach((size_t)nc + 1, st, X3_LAMBDA(size_t c) { d_aho[c] = c < nc ? d_sl[c].sh : nHs; d_akb[c] = (uint32_t)c << ksh; });
			CHK(x3_segsort_run(st, nc, d_aho, d_akb, max_dict ? max_dict - 1 : 0, k1, kA, vA, A[28], A[29], nullptr));
			segsorted = true;
		} else {
			x3_foreach(nH, st, X3_LAMBDA(size_t i) { iota[i] = (uint32_t)i; });
			CHK(x3p_sort_pairs(R.tmp, k1, kA, iota, vA, nH, (int)dsh + cb, st));
		}
		HIPCHK(hipMemsetAsync(m_first00, 0xFF, (size_t)nc * 4, st));
		X3sCtxArgs ca;
		ca.chunks = d_chunks; ca.sl = d_sl; ca.kA = kA; ca.vA = vA; ca.h_tag = h_tag; ca.stat = stat1; ca.hdr = R.hdr1.as<X3CtxHdr>(); ca.pool = R.pool1.as<uint64_t>();
		ca.pord = R.pord1.as<uint32_t>(); ca.newaddr = newaddr; ca.top = m_top1; ca.first00 = m_first00; ca.status = m_status;
		ca.kshift = dsh; ca.kmask = (dsh >= 32 ? 0xFFFFFFFFu : (1u << dsh) - 1u); ca.dbits = dsh; ca.nc = nc;
		uint64_t want = (nH / nc + 255) / 256;
		ca.nsub = want < 1 ? 1u : want > 512 ? 512u : (uint32_t)want;
		if (const char *e = getenv("X3H_SLICE_SUB")) { const int v = atoi(e); if (v >= 1 && v <= 1024) ca.nsub = (uint32_t)v; }
		ca.sstride = (uint32_t)((max_dict + 63) & ~(uint64_t)63);
		ca.dbg = nullptr;
		const bool ctx_debug = getenv("X3H_CTX_DEBUG") != nullptr;
		if (ctx_debug) { CHK(R.ctx_dbg.reserve((size_t)nc * 1024 * 16)); ca.dbg = R.ctx_dbg.as<uint32_t>(); HIPCHK(hipMemsetAsync(ca.dbg, 0, (size_t)nc * ca.nsub * 16, st)); }
		{
			const size_t need = (size_t)nc * ca.nsub * (2 * (size_t)ca.sstride + 4) * 4;
			if (R.ctx_scratch.cap < need) CHK(R.ctx_scratch.reserve(need)); /* (beyond x3s_begin's guess: a reallocation in mid-flight waits for the device, once) */
			ca.scratch = R.ctx_scratch.as<uint32_t>();
		}
		if (R.ctx_pending.cap < (size_t)nc * ca.nsub * 16) return X3H_E_INTERNAL; /* (x3s_begin sized it) */
		ca.pending = R.ctx_pending.as<uint4>();
		if (ca.nsub > 1) { if (small) X3S_LAUNCH(x3s_ctx_publish_kernel_t, ca, nc * ca.nsub, X3_WAVE, st); else X3S_LAUNCH(x3s_ctx_publish_kernel_s, ca, nc * ca.nsub, X3_WAVE, st); }
		if (small) X3S_LAUNCH(x3s_ctx1_kernel_t, ca, nc * ca.nsub, X3_WAVE, st); else X3S_LAUNCH(x3s_ctx1_kernel_s, ca, nc * ca.nsub, X3_WAVE, st);
		CHK(x3s_ctx_apply(st, ca, d_chunks));
		HIPCHK(hipGetLastError());
		if (ctx_debug) { /* (experiments) the slowest wavefronts of the context0 launch */
			std::vector<uint32_t> d((size_t)nc * ca.nsub * 4);
			HIPCHK(hipMemcpyAsync(d.data(), ca.dbg, d.size() * 4, hipMemcpyDeviceToHost, st));
			HIPCHK(hipStreamSynchronize(st));
			std::vector<size_t> idx((size_t)nc * ca.nsub);
			for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
			std::sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return d[4 * x] > d[4 * y]; });
			uint64_t sum = 0; for (size_t i = 0; i < idx.size(); i++) sum += d[4 * i];
			fprintf(stderr, "[x3h] ctx0 slice %zu: %zu wavefronts (nsub %u), mean %.0f kcycles; slowest:", k, idx.size(), ca.nsub, (double)sum / idx.size());
			for (size_t i = 0; i < 6 && i < idx.size(); i++) fprintf(stderr, " [stream %zu part %zu: %u kcyc = store %u + load %u + body %u + rest, contexts %u]", idx[i] / ca.nsub, idx[i] % ca.nsub, d[4 * idx[i]], d[4 * idx[i] + 1] & 0xFFFFu, d[4 * idx[i] + 1] >> 16, d[4 * idx[i] + 3], d[4 * idx[i] + 2]);
			fprintf(stderr, "\n");
		}
		/* pair ordinals, context0 */
		X3sPairArgs pa;
		pa.chunks = d_chunks; pa.sl = d_sl; pa.stat1 = stat1; pa.newaddr = newaddr; pa.h_pv = h_pv; pa.pord = R.pord1.as<uint32_t>(); pa.first00 = m_first00;
		pa.npairs = m_npairs; pa.ord00 = m_ord00; pa.last_ord = m_lastord; pa.k0 = k0; pa.kshift = psh;
		X3S_LAUNCH(x3s_pairs_kernel, pa, nc, X3S_PAIR_THREADS, st);
		HIPCHK(hipGetLastError());
		/* (2) hits that met a pair made earlier in this slice: the making hit holds the ordinal by now; (3) context0 of every hit (x3.c:139-147): the pair the
		 * previous hit registered or met, or -- after a new fragment -- the pair (0, 0) once it exists, else ordinal 0 */
		x3_foreach(nH, st, X3_LAMBDA(size_t j) {
			const uint32_t w = stat1[j].w;
			if (w & 0x80000000u) stat1[j].w = stat1[w & 0x7FFFFFFFu].w; /* (a making hit's own word has no flag any more) */
		});
		{
			const uint32_t kshift0 = psh;
			x3_foreach(nH, st, X3_LAMBDA(size_t j) {
				const uint32_t c = s_find_hit(d_sl, nc, (uint32_t)j);
				const uint32_t j0 = d_sl[c].sh;
				uint32_t g;
				if (h_pv[j]) g = j > j0 ? stat1[j - 1].w : m_lastord[c];
				else {
					const uint32_t ob = m_ord00[c], f00 = m_first00[c];
					g = ob != NONE32 ? ob : (f00 != NONE32 && f00 < j) ? stat1[f00].w : 0u;
				}
				k0[j] = (c << kshift0) | g;
			});
		}
		x3_foreach(nc, st, X3_LAMBDA(size_t c) {
			const X3Slice sl = d_sl[c];
			const uint32_t f00 = m_first00[c];
			if (m_ord00[c] == NONE32 && f00 != NONE32) m_ord00[c] = stat1[f00].w;
			if (sl.h1 > sl.h0) m_lastord[c] = stat1[sl.sh + (sl.h1 - sl.h0) - 1].w;
		});
		if (arrange && maxH <= X3_ARRANGE_MAX_LOCAL) {
			const uint32_t ksh = psh;
			x3_foreach((size_t)nc, st, X3_LAMBDA(size_t c) { d_akb[c] = (uint32_t)c << ksh; });
			CHK(x3_arrange_run(st, nc, d_aho, d_akb, maxH, k0, nullptr, kA, vA, nullptr, A[28], A[29]));
		} else if ((arrange || segsorted) && maxH <= X3_SEGSORT_MAX_LOCAL) {
			const uint32_t ksh = psh;
			x3_foreach((size_t)nc, st, X3_LAMBDA(size_t c) { d_akb[c] = (uint32_t)c << ksh; });
			CHK(x3_segsort_run(st, nc, d_aho, d_akb, maxH, k0, kA, vA, A[28], A[29], nullptr));
		} else {
			x3_foreach(nH, st, X3_LAMBDA(size_t i) { iota[i] = (uint32_t)i; });
			CHK(x3p_sort_pairs(R.tmp, k0, kA, iota, vA, nH, (int)psh + cb, st));
		}
		ca.stat = stat0; ca.hdr = R.hdr0.as<X3CtxHdr>(); ca.pool = R.pool0.as<uint64_t>(); ca.pord = nullptr; ca.newaddr = nullptr; ca.top = m_top0; ca.first00 = nullptr;
		ca.kshift = psh; ca.kmask = (psh >= 32 ? 0xFFFFFFFFu : (1u << psh) - 1u);
		if (ca.nsub > 1) { if (small) X3S_LAUNCH(x3s_ctx_publish_kernel_t, ca, nc * ca.nsub, X3_WAVE, st); else X3S_LAUNCH(x3s_ctx_publish_kernel_s, ca, nc * ca.nsub, X3_WAVE, st); }
		if (small) X3S_LAUNCH(x3s_ctx0_kernel_t, ca, nc * ca.nsub, X3_WAVE, st); else X3S_LAUNCH(x3s_ctx0_kernel_s, ca, nc * ca.nsub, X3_WAVE, st);
		CHK(x3s_ctx_apply(st, ca, d_chunks));
		HIPCHK(hipGetLastError());
		if (ctx_debug) { /* (experiments) the slowest wavefronts of the context0 launch */
			std::vector<uint32_t> d((size_t)nc * ca.nsub * 4);
			HIPCHK(hipMemcpyAsync(d.data(), ca.dbg, d.size() * 4, hipMemcpyDeviceToHost, st));
			HIPCHK(hipStreamSynchronize(st));
			std::vector<size_t> idx((size_t)nc * ca.nsub);
			for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
			std::sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return d[4 * x] > d[4 * y]; });
			uint64_t sum = 0; for (size_t i = 0; i < idx.size(); i++) sum += d[4 * i];
			fprintf(stderr, "[x3h] ctx0 slice %zu: %zu wavefronts (nsub %u), mean %.0f kcycles; slowest:", k, idx.size(), ca.nsub, (double)sum / idx.size());
			for (size_t i = 0; i < 6 && i < idx.size(); i++) fprintf(stderr, " [stream %zu part %zu: %u kcyc = store %u + load %u + body %u + rest, contexts %u]", idx[i] / ca.nsub, idx[i] % ca.nsub, d[4 * idx[i]], d[4 * idx[i] + 1] & 0xFFFFu, d[4 * idx[i] + 1] >> 16, d[4 * idx[i] + 3], d[4 * idx[i] + 2]);
			fprintf(stderr, "\n");
		}
	}
	if (nE) HIPCHK(hipStreamWaitEvent(st, ev_join, 0)); /* the ranks */
	}
	if (!(phase & 2)) return X3H_OK;
	/* mode chain: continues from the earlier slices' state (also for streams without a hit in this slice: nothing happens) */
	x3_foreach(nc, st, X3_LAMBDA(size_t c) { m_nidx0[c] = m_evfinal[4 * c + 3]; });
	if (nH) {
		X3ModesArgs ma;
		ma.parsed = nullptr; ma.ho = nullptr; ma.dof = nullptr;
		const uint32_t *r0 = (const uint32_t *)stat0, *r1 = (const uint32_t *)stat1;
		ma.f0 = r0; ma.t0 = r0 + 1; ma.f1 = r1; ma.t1 = r1 + 1; ma.fs = 4; ma.rank = h_rank; ma.dk = h_dk; ma.step = h_step;
		ma.idxfreq = R.idxfreq.as<uint32_t>(); ma.mode = mode; ma.pe0 = pe0; ma.pe1 = pe1; ma.ilist_rank = il_rank; ma.ilist_hit = il_hit; ma.evfinal = m_evfinal;
		ma.nzl = nzl; ma.nnoop = m_nnoop; ma.state = nullptr; ma.resume = 0; ma.slice = d_sl; ma.chunks = d_chunks;
		CHK(x3s_modes_launch(ma, nc, max_dict, st));
		X3sIdxArgs ia;
		ia.chunks = d_chunks; ia.sl = d_sl; ia.nidx_before = m_nidx0; ia.evfinal = m_evfinal; ia.lrank = il_rank; ia.lhit = il_hit; ia.h_dk = h_dk;
		ia.rfreq = rfreq; ia.rcum = rcum; ia.itot = itot; ia.hist = R.idxhist.as<uint32_t>(); ia.dbits = dsh;
		if (small) X3S_LAUNCH(x3s_idxstat_kernel_t, ia, nc, X3_WAVE, st); else X3S_LAUNCH(x3s_idxstat_kernel_s, ia, nc, X3_WAVE, st);
		HIPCHK(hipGetLastError());
	}
	if (nM) {
		X3sOrder0Args oa;
		oa.sl = d_sl; oa.lval = lval; oa.bval = bval; oa.lsm = lsm; oa.leq = leq; oa.bsm = bsm; oa.beq = beq; oa.o0hist = m_o0; oa.nc = nc;
		X3S_LAUNCH(x3s_order0_kernel, oa, 2 * nc, X3_WAVE, st);
		HIPCHK(hipGetLastError());
	}
#ifdef X3_EMU
	if (getenv("X3S_DUMP")) {
		for (uint32_t c = 0; c < nc; c++) { const X3Slice &q = hs[c]; fprintf(stderr, "[slice %zu] stream %u: t %u..%u h %u..%u d %u..%u mb %u..%u p0 %u last %u\n", k, c, q.t0, q.t1, q.h0, q.h1, q.d0, q.d1, q.mb0, q.mb1, q.p0, q.last); }
		for (size_t j = 0; j < nH; j++) fprintf(stderr, "  hit %zu: tag %u c1 %u pv %u dk %u step %u rank %u | s1 f %u t %u c %u ord %u | k0 %x s0 f %u t %u c %u | mode %u pe0 %u pe1 %u nzl %u\n", j, h_tag[j], h_c1[j], h_pv[j], h_dk[j], h_step[j], h_rank[j],
			stat1[j].x, stat1[j].y, stat1[j].z, stat1[j].w, k0[j], stat0[j].x, stat0[j].y, stat0[j].z, mode[j], pe0[j], pe1[j], nzl[j]);
	}
#endif
	/* ---- the coded symbols of the slice, in coding order, straight into the stream's operand array (no-op symbols dropped): the compacted index of step k's
	 *      first symbol = 2k + fragment bytes before k - no-op hits before k.  Size estimates: one term per slot of the slice's RAW symbol range. ---- */
	float *est_val = nullptr;
	uint8_t *est_cls = nullptr;
	const uint64_t nYraw = 2 * nS + nB + (last ? nc : 0);
	if (want_est) {
		if (R.est_val.cap < (nYraw + 4) * 4 || R.est_cls.cap < nYraw + 4) return X3H_E_INTERNAL;
		est_val = R.est_val.as<float>(); est_cls = R.est_cls.as<uint8_t>();
		HIPCHK(hipMemsetAsync(est_cls, X3_EST_NONE, nYraw + 4, st));
	}
	uint4 *sy = R.sym.as<uint4>();
	if (nS) x3_foreach(nS, st, X3_LAMBDA(size_t gs) {
		const uint32_t c = s_find(d_sl, nc, (uint32_t)gs);
		const X3Slice sl = d_sl[c];
		const uint32_t k = sl.t0 + ((uint32_t)gs - sl.ss);
		const uint64_t base = d_chunks[c].elem_off;
		const uint32_t info = tok_info[base + k], hb = s_hb[gs], mb = s_mb[gs];
		const uint32_t nh = sl.h1 - sl.h0, hl = hb - sl.h0;                 /* slice-local index of the next hit at or after this step */
		const uint32_t nz = hl < nh ? nzl[sl.sh + hl] : m_nnoop[c];         /* no-op hits of the stream before this step */
		const uint64_t yi = 3 * base + (2 * k + mb - nz);
		const uint32_t ye = sl.sy + 2 * (k - sl.t0) + (mb - sl.mb0);         /* slot of the step's first symbol in the slice's raw range (estimates) */
		const uint32_t evtotal = 2051u + k; /* model_events: 1024+1024+1+1+1 (x3.c:236-244), +1 per step */
		if (!(info & X3_TOK_MISS)) {
			const uint32_t j = sl.sh + hl, m = mode[j], e0 = pe0[j], e1 = pe1[j];
			const uint32_t e2 = 2049u + hb - e0 - e1; /* every hit bumps exactly one of the three */
			sy[yi] = x3_make_symbol(m == X3_E_CTX0 ? 0u : m == X3_E_CTX1 ? e0 : e0 + e1, m == X3_E_CTX0 ? e0 : m == X3_E_CTX1 ? e1 : e2, evtotal);
			uint32_t cu, fq, to;
			if (m == X3_E_CTX0) { const uint4 r = stat0[j]; cu = r.z; fq = r.x; to = r.y; }
			else if (m == X3_E_CTX1) { const uint4 r = stat1[j]; cu = r.z; fq = r.x; to = r.y; }
			else { cu = rcum[j]; fq = rfreq[j]; to = itot[j]; }
			if (to > 1u) sy[yi + 1] = x3_make_symbol(cu, fq, to);
			if (est_cls) { /* x3.c:152-172,192-193 */
				const float pe = (float)(m == X3_E_CTX0 ? e0 : m == X3_E_CTX1 ? e1 : e2) / (float)evtotal;
				est_val[ye] = x3_est_term(pe * ((float)fq / (float)to));
				est_cls[ye] = (uint8_t)m;
			}
		} else {
			const uint32_t len = info & 0x3Fu, mk = k - hb; /* mk = new fragments before this one */
			sy[yi] = x3_make_symbol(2049u + hb, 1u + mk, evtotal); /* E_NEW */
			const uint32_t gm = sl.sm + (mk - (sl.t0 - sl.h0));
			sy[yi + 1] = x3_make_symbol((len - 1) + lsm[gm], 1u + leq[gm], 32u + mk);
			if (est_cls) {
				est_val[ye] = x3_est_term((float)(1u + mk) / (float)evtotal); est_cls[ye] = X3_E_NEW;
				est_val[ye + 1] = x3_est_term((float)(1u + leq[gm]) / (float)(32u + mk)); est_cls[ye + 1] = X3_E_NEW;
			}
			for (uint32_t q = 0; q < len; q++) {
				const uint32_t gb = sl.sb + (mb - sl.mb0) + q;
				sy[yi + 2 + q] = x3_make_symbol(bval[gb] + bsm[gb], 1u + beq[gb], 256u + mb + q);
				if (est_cls) { est_val[ye + 2 + q] = x3_est_term((float)(1u + beq[gb]) / (float)(256u + mb + q)); est_cls[ye + 2 + q] = X3_E_NEW; }
			}
		}
	});
	/* per stream: E_EOF behind the last slice (x3.c:432-433), the symbols assembled so far, and the coder's segment -- whole groups of 8 until the end */
	x3_foreach(nc, st, X3_LAMBDA(size_t c) {
		const X3Slice sl = d_sl[c];
		const uint64_t base = d_chunks[c].elem_off;
		uint32_t ycnt = 2 * sl.t1 + sl.mb1 - m_nnoop[c];
		if (sl.last) { const uint32_t evtotal = 2051u + sl.t1; sy[3 * base + ycnt] = x3_make_symbol(evtotal - 1, 1, evtotal); }
		ycnt += sl.ended;
		const uint32_t done = m_ydone[c];
		uint32_t avail = ycnt - done;
		if (ycnt < done || avail > 3u * (d_chunks[c].len + 16u)) { m_status[c] = X3_ST_POOL_FULL; avail = 0; } /* (a bookkeeping bug must not become a runaway coder chain) */
		const uint32_t len = sl.ended ? avail : avail - avail % X3_AC2_G;
		m_ycnt[c] = ycnt; m_segoff[c] = (uint32_t)(3 * base) + done; m_seglen[c] = len; m_ydone[c] = done + len;
		m_ntok[c] = sl.t1; m_nhits[c] = sl.h1;
		m_estfirst[c] = sl.sy; m_estcnt[c] = 2 * (sl.t1 - sl.t0) + (sl.mb1 - sl.mb0);
	});
	HIPCHK(hipGetLastError());
	return X3H_OK;
}
