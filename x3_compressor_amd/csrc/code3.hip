/*
 * code3.hip -- K3 feature kernels for batches of MANY independent streams: one wavefront per stream sweeps that stream's records in
 * tiles of 64 with its adaptive tables in LDS, instead of chip-wide sort / scan / partition passes over (stream | value) keys.
 *
 * code2.hip turns everything the reference recomputes step by step into counting questions answered by global radix sorts and the
 * count-smaller-before partition; those passes move ~2.4 KB of HBM traffic per input byte of a many-chunk batch.  The alphabets
 * involved are SMALL per stream (a dictionary of a few hundred elements for a 256 KiB chunk), so a stream's tables fit in LDS and
 * the serial order of the reference only has to be respected BETWEEN tiles of 64 records; inside a tile the 64 lanes resolve their
 * mutual dependencies with ballots:
 *     same-value masks            : one ballot per value bit                               (wave_same_mask)
 *     "how many earlier lanes of my group hold a smaller value" : one ballot per value bit  (wave_count_less)
 * Kernels (each cites the reference code whose state it carries):
 *   x3_mtfrank_kernel : move-to-front rank of every hit        dict.c:132-146 (dict_update_costs + qsort == move-to-front)
 *   x3_ctxseg_kernel  : freq / total / cum_freq / first use of every hit's tag in its context   context.c:20-56,95-133
 * Used by x3_code_v2_run for the stage-after-stage schedule when every stream's dictionary fits the LDS tables; the generic
 * sort-based passes of code2.hip remain for anything else (and for the pipelined schedule of a few long streams).
 */
#include "k3_wave.h"
#include "seg_rank.h"
#include <stdlib.h>

template <uint32_t DMAX>
__device__ static void x3_mtfrank_body(const X3MtfArgs &a)
{
	X3_LDS uint16_t lst[DMAX];
	X3_LDS uint16_t pos0[DMAX];
	const uint32_t c = blockIdx.x, lane = x3_lane();
	x3_mtf_tiles(a, lst, pos0, a.eo[c], a.eo[c + 1], 0, a.dof[c], lane);
}

/* The same ranks with a stream's events cut into X3_MTFP_WAVES time ranges, one wavefront each (one workgroup per stream; dictionaries of at
 * most X3_MTFP_DMAX elements).  The list at the start of a range is the elements that exist by then in order of their last touch, latest
 * first: (1) every wavefront records the last touch (insertion included) of each element inside ITS range and counts its insertions;
 * (2) a running maximum over the earlier ranges gives each wavefront the last touch before its range, the running count the number of
 * elements; (3) an element's position = how many elements were touched later (last touches are distinct events); (4) the tile loop above.
 * One lone wavefront per stream walks ~1 500 tiles at ~5 us each (a chain of ballots and LDS round trips); the eight ranges overlap them. */
#define X3_MTFP_WAVES 8u
#define X3_MTFP_DMAX 512u
__device__ static void x3_mtfrank_par_body(const X3MtfArgs &a)
{
	X3_LDS uint32_t tab[X3_MTFP_WAVES][X3_MTFP_DMAX];   /* (1) last touch + 1 inside range w, 0: none; after (2): last touch + 1 before range w */
	X3_LDS uint16_t lstw[X3_MTFP_WAVES][X3_MTFP_DMAX];
	X3_LDS uint16_t pos0w[X3_MTFP_WAVES][X3_MTFP_DMAX];
	X3_LDS uint32_t nnew[X3_MTFP_WAVES];
	const uint32_t c = blockIdx.x, lane = x3_lane(), wv = threadIdx.x / X3_WAVE;
	const uint32_t e0 = a.eo[c], e1 = a.eo[c + 1], dof = a.dof[c];
	const uint32_t per = (((e1 - e0 + X3_MTFP_WAVES - 1) / X3_MTFP_WAVES) + X3_WAVE - 1) & ~(X3_WAVE - 1);
	const uint32_t s0 = e0 + wv * per < e1 ? e0 + wv * per : e1, s1 = s0 + per < e1 ? s0 + per : e1; /* (per * 8 < 2^32: a stream has < 2^28 events) */
	for (uint32_t i = lane; i < X3_MTFP_DMAX; i += X3_WAVE) tab[wv][i] = 0;
	x3_wave_sync();
	uint32_t cnt_new = 0;
	for (uint32_t base = s0; base < s1; base += X3_WAVE) {
		const uint32_t i = base + lane;
		const bool valid = i < s1;
		const uint32_t t = valid ? a.e_tag[i] - dof : 0u;
		const bool isnew = valid && a.e_hit[i] == NONE32;
		if (valid) atomicMax(&tab[wv][t], i - e0 + 1);
		cnt_new += (uint32_t)x3_popc64(x3_ballot(isnew));
	}
	if (lane == 0) nnew[wv] = cnt_new;
	__syncthreads();
	for (uint32_t t = threadIdx.x; t < X3_MTFP_DMAX; t += X3_MTFP_WAVES * X3_WAVE) { /* (2), in place */
		uint32_t run = 0;
		for (uint32_t w = 0; w < X3_MTFP_WAVES; w++) { const uint32_t own = tab[w][t]; tab[w][t] = run; run = own > run ? own : run; }
	}
	uint32_t Dcur = 0;
	for (uint32_t w = 0; w < wv; w++) Dcur += nnew[w];
	__syncthreads();
	uint16_t *const lst = lstw[wv], *const pos0 = pos0w[wv];
	if (s0 < s1) {
		for (uint32_t tb = 0; tb < Dcur; tb += X3_WAVE) { /* (3) */
			const uint32_t t = tb + lane;
			const uint32_t mine = t < Dcur ? tab[wv][t] : 0xFFFFFFFFu;
			uint32_t later = 0;
			for (uint32_t u = 0; u < Dcur; u++) later += tab[wv][u] > mine ? 1u : 0u;
			if (t < Dcur) { lst[later] = (uint16_t)t; pos0[t] = (uint16_t)later; }
		}
		x3_wave_sync();
		x3_mtf_tiles(a, lst, pos0, s0, s1, Dcur, dof, lane);
	}
}

/* ============================================================================================================
 * Context statistics.  Hits arranged by (context, time) -- arrangement A of code2.hip: kA = context key, vA = hit.  A stream's hits
 * are one contiguous range of the arrangement (context keys are stream-major), so one wavefront per stream sweeps it in tiles of 64.
 * A context's item list (context.c:42-56: tags in first-seen order, freq per item) only has to be CARRIED between tiles for the one
 * context that spans the tile boundary ("open"): LDS holds tag -> position, position -> tag / first hit / freq / cum_freq
 * (cum_freq of an item = sum of the freqs at smaller positions, count_cum_freqs ac.c:6-18).  Everything else is
 * resolved inside the tile.  The arrangement's tags are gathered by a chip-wide pass beforehand and the four results of a hit leave
 * as ONE 16-byte store (a stream's hit arrays do not fit any cache: a scattered 4-byte store costs a whole sector).  Per hit:
 *   freq  = earlier hits of the context with the same tag (0: the tag is not in the context yet, context.c:20-29)
 *   total = earlier hits of the context                     (calc_total_freq)
 *   cum   = earlier hits of the context whose tag stands before this tag in the list
 *   first = the hit that put the tag into the context, isfirst = this hit is that one   (-> tag-pair ordinals, tag_pair.c:100-130)
 * ============================================================================================================ */
struct X3CtxSegArgs {
	const uint32_t *ho;        /* nc+1: hit ranges (== ranges of the arrangement)         */
	const uint32_t *dof;       /* per chunk: first global tag id                          */
	const uint32_t *kA, *vA;   /* arrangement by (context, time): context key, hit        */
	const uint32_t *tA;        /* ... and the hit's global tag id; or nullptr and          */
	const uint32_t *h_tag;     /* ... the tags by hit: the kernel fetches h_tag[vA[.]] itself, vA two tiles ahead, the tags one (no gather pass, no tA array) */
	uint4 *stat;               /* out per hit: {freq, total, cum, first | isfirst << 31}  */
	uint32_t *first00;         /* context1 launch: per stream, the hit that first uses the pair (element 0 in context 0) -- x3.c:424-425; preset to NONE32.  Else nullptr */
	uint32_t dbits_max;        /* bits that cover every local tag of the batch            */
	uint32_t nsub;             /* wavefronts per stream: wavefront r takes the contexts that START in the r-th part of the range (contexts are independent) */
	uint32_t nc;               /* streams */
	uint32_t xcd;              /* 1: workgroup b works for stream 8 * (b / 8 / nsub) + b % 8 -- all wavefronts of a stream on ONE XCD (workgroups go to the XCDs
	                              round-robin), so the stream's scattered 16-byte results meet in that XCD's L2 and leave it as whole lines */
};

template <uint32_t DMAX>
__device__ static void x3_ctxseg_body(const X3CtxSegArgs &a)
{
	X3_LDS uint16_t tpos[DMAX];      /* tag -> list position in the open context, NONE16: absent */
	X3_LDS uint16_t ltag[DMAX];      /* position -> tag                                           */
	X3_LDS uint32_t lfirst[DMAX];    /* position -> hit that added the item                       */
	X3_LDS uint32_t lfreq[DMAX];     /* position -> freq                                          */
	X3_LDS uint32_t lpre[DMAX];      /* position -> cum_freq = sum of the freqs before it (count_cum_freqs, ac.c:6-18), as of the tile's start */
	const uint32_t lane = x3_lane();
	uint32_t c = blockIdx.x / a.nsub, sub = blockIdx.x % a.nsub;
	if (a.xcd) {
		const uint32_t q = blockIdx.x >> 3;
		c = ((q / a.nsub) << 3) + (blockIdx.x & 7u); sub = q % a.nsub;
		if (c >= a.nc) return;
	}
	const uint32_t c0 = a.ho[c], c1 = a.ho[c + 1], dof = a.dof[c];
	const uint64_t bit = (uint64_t)1 << lane, below = bit - 1;
	const int tbits = (int)a.dbits_max; /* covers tags and list positions alike (a list holds each tag once) */
	/* my part of the stream's range: from the first context boundary at or after the nominal cut to the first one at or after the next cut */
	uint32_t h0 = c0, h1 = c1;
	if (a.nsub > 1) {
		const uint32_t per = (c1 - c0 + a.nsub - 1) / a.nsub;
		uint32_t cut[2];
		for (int w = 0; w < 2; w++) {
			uint64_t nom = (uint64_t)c0 + (uint64_t)(sub + (uint32_t)w) * per;
			uint32_t p = nom >= c1 ? c1 : (uint32_t)nom;
			if (p > c0 && p < c1) { /* first i >= p with kA[i] != kA[i-1] */
				for (;;) {
					const uint32_t i = p + lane;
					const uint64_t bm = x3_ballot(i < c1 && a.kA[i] != a.kA[i - 1]);
					if (bm) { p += (uint32_t)x3_ctz64(bm); break; }
					p += X3_WAVE;
					if (p >= c1) { p = c1; break; }
				}
			}
			cut[w] = p;
		}
		h0 = cut[0]; h1 = cut[1];
	}
	for (uint32_t i = lane; i < DMAX; i += X3_WAVE) tpos[i] = NONE16;
	bool open = false;
	uint32_t open_g = 0, open_k = 0, open_total = 0;
	uint32_t ng_ = 0, nh_ = 0, nt_ = 0, nh2_ = 0;
	if (h0 + lane < h1) { ng_ = a.kA[h0 + lane]; nh_ = a.vA[h0 + lane]; if (a.tA) nt_ = a.tA[h0 + lane]; }
	if (!a.tA) {
		if (h0 + X3_WAVE + lane < h1) nh2_ = a.vA[h0 + X3_WAVE + lane];
		if (h0 + lane < h1) nt_ = a.h_tag[nh_];
	}
	x3_wave_sync();
	for (uint32_t base = h0; base < h1; base += X3_WAVE) {
		const bool valid = base + lane < h1;
		const uint32_t g = ng_, hit = nh_, t = valid ? nt_ - dof : 0u;
		{ /* the next tile's records are in flight while this one is resolved */
			const uint32_t nx = base + X3_WAVE + lane;
			if (a.tA) { if (nx < h1) { ng_ = a.kA[nx]; nh_ = a.vA[nx]; nt_ = a.tA[nx]; } }
			else {
				if (nx < h1) { ng_ = a.kA[nx]; nh_ = nh2_; nt_ = a.h_tag[nh2_]; } /* nh2_ was fetched a tile ago */
				if (nx + X3_WAVE < h1) nh2_ = a.vA[nx + X3_WAVE];
			}
		}
		const uint64_t V = x3_ballot(valid);
		const uint32_t nvalid = (uint32_t)x3_popc64(V);
		/* segments (contexts) of the tile */
		const uint32_t gprev = wave_prev_u32(g);
		const uint64_t S = x3_ballot(valid && (lane == 0 || g != gprev));
		const uint32_t s = 63u - (uint32_t)x3_clz64(S & (below | bit));                 /* first lane of my segment */
		const uint64_t Sabove = S & ~(below | bit);
		const uint32_t e = Sabove ? (uint32_t)x3_ctz64(Sabove) : nvalid;                /* one past its last lane    */
		const uint64_t seg = (e >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << e) - 1)) & ~(((uint64_t)1 << s) - 1);
		const uint64_t B = seg & below;                                                 /* earlier lanes of my segment */
		const uint32_t g0 = x3_readlane_u32(g, 0);
		const bool cont = open && g0 == open_g;
		if (open && !cont) { /* the open context ended with the previous tile */
			for (uint32_t p = lane; p < open_k; p += X3_WAVE) tpos[ltag[p]] = NONE16;
			open = false;
			x3_wave_sync();
		}
		const bool carried = valid && cont && s == 0; /* my context is the open one */
		const uint64_t M = wave_same_mask(t, tbits, V, valid) & seg;
		const uint64_t E = M & B;
		const uint32_t fl = E ? (uint32_t)x3_ctz64(E) : lane;    /* first lane of my (context, tag) in the tile */
		const uint32_t cp = carried ? (uint32_t)tpos[t] : (uint32_t)NONE16;
		const bool known = cp != NONE16;
		const bool isnew = valid && fl == lane && !known;        /* this hit adds the tag to its context */
		const uint64_t N = x3_ballot(isnew);
		uint32_t pos = known ? cp : (carried ? open_k : 0u) + (uint32_t)x3_popc64(N & B);
		const uint32_t pos_fl = x3_bcast_u32(pos, (int)fl);
		if (!known && fl != lane) pos = pos_fl;
		const uint32_t freq = (known ? lfreq[cp] : 0u) + (uint32_t)x3_popc64(E);
		const uint32_t total = (carried ? open_total : 0u) + (lane - s);
		const uint32_t cbase = !carried ? 0u : known ? lpre[cp] : open_total; /* a new item stands behind every carried one */
		const uint32_t cum = cbase + wave_count_less(pos, pos, tbits, B);
		const uint32_t hit_fl = x3_bcast_u32(hit, (int)fl);
		if (valid) {
			uint4 r;
			r.x = freq; r.y = total; r.z = cum; r.w = (known ? lfirst[cp] : hit_fl) | (isnew ? 0x80000000u : 0u);
			a.stat[hit] = r;
			if (a.first00 && isnew && t == 0u && g == dof) a.first00[c] = hit;
		}
		/* ---- carry the last context of the tile if it goes on in the next one ---- */
		const uint32_t sl = 63u - (uint32_t)x3_clz64(S);               /* first lane of the tile's last segment */
		const uint32_t glast = x3_readlane_u32(g, sl);
		const bool more = base + X3_WAVE < h1;
		const uint32_t gnext = more ? x3_readlane_u32(ng_, 0) : 0u;
		const bool goes_on = more && gnext == glast;
		const bool whole = cont && sl == 0;                             /* the open context covers the whole tile */
		x3_wave_sync();
		if (open && !whole) { /* it ended inside this tile */
			for (uint32_t p = lane; p < open_k; p += X3_WAVE) tpos[ltag[p]] = NONE16;
			open = false;
			x3_wave_sync();
		}
		if (goes_on) {
			if (!open) { open = true; open_g = glast; open_k = 0; open_total = 0; } /* a context that started in this tile becomes the open one */
			const bool mine = valid && s == sl;
			if (mine && isnew) { tpos[t] = (uint16_t)pos; ltag[pos] = (uint16_t)t; lfirst[pos] = hit; lfreq[pos] = 0; }
			x3_wave_sync();
			if (mine) atomicAdd(&lfreq[pos], 1u);
			open_k += (uint32_t)x3_popc64(N & ~(((uint64_t)1 << sl) - 1));
			open_total += nvalid - sl;
			x3_wave_sync();
			uint32_t carry = 0; /* cum_freqs of the list as the next tile will see them */
			for (uint32_t pb = 0; pb < open_k; pb += X3_WAVE) {
				const uint32_t p = pb + lane;
				const uint32_t v = p < open_k ? lfreq[p] : 0u;
				const uint32_t incl = x3_wave_incl_scan_u32(v) + carry;
				if (p < open_k) lpre[p] = incl - v;
				carry = x3_readlane_u32(incl, X3_WAVE - 1);
			}
			x3_wave_sync();
		}
	}
}

/* ============================================================================================================
 * The two adaptive order-0 models of a new fragment (x3.c:259-267): model_match_size over the 32 lengths, model_chars over the 256
 * byte values; every symbol is coded and then counted (+1).  For the i-th value v of a stream:
 *   equal   = earlier values == v  (freq = 1 + equal),   smaller = earlier values < v  (cum_freq = v + smaller).
 * One wavefront per (stream, model): 256 counters and their running sums in LDS, 64 values per trip.
 * ============================================================================================================ */
struct X3Order0Args {
	const uint32_t *off[2];    /* nc+1 ranges: fragment lengths, fragment bytes   */
	const uint32_t *val[2];    /* the values (length-1 / byte)                    */
	uint32_t *smaller[2], *equal[2];
	uint32_t nc;
};

__device__ static void x3_order0_body(const X3Order0Args &a)
{
	X3_LDS uint32_t hist[256];
	X3_LDS uint32_t pre[256];
	const uint32_t which = blockIdx.x >= a.nc ? 1u : 0u, c = blockIdx.x - which * a.nc, lane = x3_lane();
	const int bits = which ? 8 : 5;
	const uint32_t A = which ? 256u : 32u;
	const uint32_t i0 = a.off[which][c], i1 = a.off[which][c + 1];
	const uint32_t *val = a.val[which];
	uint32_t *osm = a.smaller[which], *oeq = a.equal[which];
	const uint64_t below = ((uint64_t)1 << lane) - 1;
	for (uint32_t i = lane; i < 256; i += X3_WAVE) { hist[i] = 0; pre[i] = 0; }
	x3_wave_sync();
	for (uint32_t base = i0; base < i1; base += X3_WAVE) {
		const bool valid = base + lane < i1;
		const uint32_t v = valid ? val[base + lane] : 0u;
		const uint64_t V = x3_ballot(valid);
		const uint64_t M = wave_same_mask(v, bits, V, valid);
		const uint32_t sm = wave_count_less(v, v, bits, below & V);
		if (valid) { oeq[base + lane] = hist[v] + (uint32_t)x3_popc64(M & below); osm[base + lane] = pre[v] + sm; }
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
}

/* ============================================================================================================
 * model_index1 as the IDX1-coded hits of a stream see it (x3.c:187-188): the symbol is the hit's move-to-front rank, its freq is
 * 1 + earlier IDX1 hits with that rank, its cum_freq = rank + earlier IDX1 hits with a smaller rank (every rank starts at freq 1,
 * model_enlarge ac.c:250-266), the total = dictionary elements + earlier IDX1 hits.  The mode kernel left the stream's IDX1 hits as a
 * list (rank, hit) in time order; one wavefront per stream, counters per rank and their running sums in LDS, 64 list entries per trip.
 * ============================================================================================================ */
struct X3IdxStatArgs {
	const uint32_t *ho;        /* nc+1: the list of stream c starts at ho[c]             */
	const uint32_t *evfinal;   /* per chunk, word 3: number of list entries              */
	const uint32_t *lrank, *lhit, *h_dk;
	uint32_t *rfreq, *rcum, *itot; /* out per (IDX1-coded) hit                           */
	uint32_t dbits_max;
};

/* list entries [lo, hi) of the stream whose list starts at i0, on counters (hist) and running sums (pre) that hold the entries before lo:
 * live up to rank rmax when `started`, untouched otherwise */
__device__ static __forceinline__ void x3_idxstat_tiles(const X3IdxStatArgs &a, uint32_t *hist, uint32_t *pre, const uint32_t i0, const uint32_t lo, const uint32_t hi,
                                                        uint32_t rmax, bool started, const uint32_t lane)
{
	const uint64_t below = ((uint64_t)1 << lane) - 1;
	const int bits = (int)a.dbits_max;
	for (uint32_t base = lo; base < hi; base += X3_WAVE) {
		const bool valid = base + lane < hi;
		const uint32_t r = valid ? a.lrank[i0 + base + lane] : 0u, hit = valid ? a.lhit[i0 + base + lane] : 0u;
		const uint32_t tmax = wave_max_u32(r);
		if (!started || tmax > rmax) { /* grow the live part (zero counters, running sums continue flat) */
			const uint32_t from = !started ? 0u : rmax + 1, flat = !started ? 0u : pre[rmax] + hist[rmax];
			x3_wave_sync();
			for (uint32_t q = from + lane; q <= tmax; q += X3_WAVE) { hist[q] = 0; pre[q] = flat; }
			if (tmax > rmax || !started) rmax = tmax;
			started = true;
			x3_wave_sync();
		}
		const uint64_t V = x3_ballot(valid);
		const uint64_t M = wave_same_mask(r, bits, V, valid);
		const uint32_t sm = wave_count_less(r, r, bits, below & V);
		if (valid) {
			a.rfreq[hit] = 1u + hist[r] + (uint32_t)x3_popc64(M & below);
			a.rcum[hit] = r + pre[r] + sm;
			a.itot[hit] = a.h_dk[hit] + base + lane;
		}
		x3_wave_sync();
		if (valid) atomicAdd(&hist[r], 1u);
		x3_wave_sync();
		uint32_t carry = 0;
		for (uint32_t pb = 0; pb <= rmax; pb += X3_WAVE) {
			const uint32_t q = pb + lane;
			const uint32_t h = q <= rmax ? hist[q] : 0u;
			const uint32_t incl = x3_wave_incl_scan_u32(h) + carry;
			if (q <= rmax) pre[q] = incl - h;
			carry = x3_readlane_u32(incl, X3_WAVE - 1);
		}
		x3_wave_sync();
	}
}

template <uint32_t DMAX>
__device__ static void x3_idxstat_body(const X3IdxStatArgs &a)
{
	X3_LDS uint32_t hist[DMAX];
	X3_LDS uint32_t pre[DMAX];
	const uint32_t c = blockIdx.x, lane = x3_lane();
	x3_idxstat_tiles(a, hist, pre, a.ho[c], 0, a.evfinal[4 * c + 3], 0, false, lane);
}

/* The same with a stream's list cut into X3_IDXP_WAVES time ranges, one wavefront each (ranks below X3_IDXP_DMAX, i.e. dictionaries of at most that
 * many elements): the counters are additive, so the state at the start of a range is the sum of the earlier ranges' counts -- every wavefront
 * counts its range, a running sum over the wavefronts (in place) turns the counts into start states, the running sums over the ranks follow,
 * then the tile loop above.  One lone wavefront per stream is a chain of LDS round trips per tile; the ranges overlap them. */
#define X3_IDXP_WAVES 8u
#define X3_IDXP_DMAX 512u
__device__ static void x3_idxstat_par_body(const X3IdxStatArgs &a)
{
	X3_LDS uint32_t histw[X3_IDXP_WAVES][X3_IDXP_DMAX];
	X3_LDS uint32_t prew[X3_IDXP_WAVES][X3_IDXP_DMAX];
	X3_LDS uint32_t wmax[X3_IDXP_WAVES]; /* largest rank inside the range + 1; 0: the range is empty */
	const uint32_t c = blockIdx.x, lane = x3_lane(), wv = threadIdx.x / X3_WAVE;
	const uint32_t i0 = a.ho[c], n = a.evfinal[4 * c + 3];
	const uint32_t per = (((n + X3_IDXP_WAVES - 1) / X3_IDXP_WAVES) + X3_WAVE - 1) & ~(X3_WAVE - 1);
	const uint32_t lo = wv * per < n ? wv * per : n, hi = lo + per < n ? lo + per : n;
	for (uint32_t i = lane; i < X3_IDXP_DMAX; i += X3_WAVE) histw[wv][i] = 0;
	x3_wave_sync();
	uint32_t mx = 0;
	for (uint32_t base = lo; base < hi; base += X3_WAVE) {
		const bool valid = base + lane < hi;
		const uint32_t r = valid ? a.lrank[i0 + base + lane] : 0u;
		if (valid) atomicAdd(&histw[wv][r], 1u);
		const uint32_t t = wave_max_u32(valid ? r + 1u : 0u);
		mx = t > mx ? t : mx;
	}
	if (lane == 0) wmax[wv] = mx;
	__syncthreads();
	for (uint32_t r = threadIdx.x; r < X3_IDXP_DMAX; r += X3_IDXP_WAVES * X3_WAVE) { /* counts of the earlier ranges, in place */
		uint32_t run = 0;
		for (uint32_t w = 0; w < X3_IDXP_WAVES; w++) { const uint32_t own = histw[w][r]; histw[w][r] = run; run += own; }
	}
	uint32_t seen = 0; /* largest rank before my range, + 1 */
	for (uint32_t w = 0; w < wv; w++) seen = wmax[w] > seen ? wmax[w] : seen;
	__syncthreads();
	if (lo >= hi) return;
	uint32_t *const hist = histw[wv], *const pre = prew[wv];
	if (seen) {
		uint32_t carry = 0;
		for (uint32_t pb = 0; pb < seen; pb += X3_WAVE) {
			const uint32_t q = pb + lane;
			const uint32_t h = q < seen ? hist[q] : 0u;
			const uint32_t incl = x3_wave_incl_scan_u32(h) + carry;
			if (q < seen) pre[q] = incl - h;
			carry = x3_readlane_u32(incl, X3_WAVE - 1);
		}
		x3_wave_sync();
	}
	x3_idxstat_tiles(a, hist, pre, i0, lo, hi, seen ? seen - 1u : 0u, seen != 0u, lane);
}

/* ============================================================================================================
 * Token walk.  K2 leaves one word per parse step (tag of the hit element, or fragment length + "already present" flag); the coding
 * stage indexes with running counts over that list: hits / inserted elements / new-fragment bytes / input position before each step
 * (x3.c:394,422: p += len), and needs per-hit and per-touch records (context1 = previous tag or 0 after a new fragment, x3.c:389-390,
 * 424-425).  One workgroup per stream walks its tokens in tiles of 1024 with the four counts carried from tile to tile -- instead of
 * four chip-wide scans plus two passes that find every step's stream by binary search.  Counts are stream-relative.
 * ============================================================================================================ */
#define X3_TOK_THREADS 1024u
struct X3TokArgs {
	const X3Chunk *chunks;
	const X3ParseResult *parsed;
	const uint32_t *tok_info;
	const uint8_t *dict_len;
	uint32_t *tok_pos, *tok_hb, *tok_nb, *tok_mb;        /* out per step (layout: chunk elem_off + step): tok_hb and tok_mb -- what the symbol assembly indexes with */
	const uint32_t *ho, *eo, *dof;                       /* per chunk: first hit / first touch event / first tag      */
	uint32_t *h_tag, *h_c1, *h_pv, *h_dk, *h_step;       /* out per hit                                               */
	uint32_t *e_tag, *e_hit;                             /* out per touch event (hit or insertion)                    */
	/* new fragments (x3.c:259-267): the length symbol and the bytes of every one, in order -- the walk knows where a fragment lies and how many came before it, so the
	 * order-0 models' inputs leave here instead of by one more pass over every step of the batch (1.8 ms of the 1024 x 256 KiB batch for 0.1-1.5 % of the steps) */
	const uint8_t *bytes;                                /* padded chunks                                             */
	const uint32_t *mo, *bo;                             /* per chunk: first new fragment / first new-fragment byte   */
	uint32_t *lval, *bval;                               /* out per new fragment: length - 1; per fragment byte: the byte */
};

__device__ static void x3_tokens_body(const X3TokArgs &a)
{
	const uint32_t NW = X3_TOK_THREADS / X3_WAVE;
	X3_LDS uint32_t s_w[3][X3_TOK_THREADS / X3_WAVE];
	const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = x3_lane(), wave = tid / X3_WAVE;
	const uint64_t base = a.chunks[c].elem_off;
	const uint32_t ntok = a.parsed[c].ntok;
	const uint32_t ho = a.ho[c], eo = a.eo[c], dof = a.dof[c];
	const uint64_t below = ((uint64_t)1 << lane) - 1;
	uint32_t chb = 0, cnb = 0, cmb = 0, cpos = 0; /* counts before the tile */
	uint32_t prev_info = X3_TOK_MISS;            /* the step before the tile's first (none: behaves like a new fragment) */
	for (uint32_t tb = 0; tb < ntok; tb += X3_TOK_THREADS) {
		const uint32_t k = tb + tid;
		const bool in = k < ntok;
		const uint32_t info = in ? a.tok_info[base + k] : X3_TOK_MISS;
		const bool hit = in && !(info & X3_TOK_MISS), nw = in && (info & X3_TOK_MISS) && !(info & X3_TOK_DUP);
		const uint32_t mb = in && (info & X3_TOK_MISS) ? (info & 0x3Fu) : 0u;
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
		/* the step before mine: lane - 1, the previous wave's last lane, or the previous tile's last step */
		uint32_t pinfo = x3_shfl_up_u32(info, 1);
		if (in) {
			if (lane == 0) pinfo = k == 0 ? X3_TOK_MISS : a.tok_info[base + k - 1];
			a.tok_hb[base + k] = hb; a.tok_mb[base + k] = mbb; /* (the running element count and the position of a step stay in registers: nothing behind this walk reads them per step) */
			if (hit) {
				const uint32_t gh = ho + hb, ev = eo + hb + nb;
				const bool pv = !(pinfo & X3_TOK_MISS);
				a.h_tag[gh] = dof + info;
				a.h_c1[gh] = dof + (pv ? pinfo : 0u); /* context1 (x3.c:390,425) */
				a.h_pv[gh] = pv ? 1u : 0u;
				a.h_dk[gh] = nb;
				a.h_step[gh] = k;
				a.e_tag[ev] = dof + info;
				a.e_hit[ev] = gh;
			} else if (nw) {
				const uint32_t ev = eo + hb + nb;
				a.e_tag[ev] = dof + nb; /* the new element's tag (dict.c:100) */
				a.e_hit[ev] = NONE32;
			}
			if (!hit) a.lval[a.mo[c] + (k - hb)] = mb - 1u; /* a new fragment (one that repeats an element too): its length symbol for the order-0 model */
		}
		/* ... and its bytes: the wavefront takes its new fragments one at a time, lane j copies byte j (a fragment is at most 32 bytes) -- a loop over the bytes in the
		 * fragment's own lane made every wavefront with a new fragment among its 64 steps (all of them on Zipf bytes) wait for 32 dependent trips */
		for (uint64_t Mm = x3_ballot(in && !hit); Mm; Mm &= Mm - 1) {
			const uint32_t l = (uint32_t)x3_ctz64(Mm);
			const uint32_t flen = x3_readlane_u32(mb, l), fpos = x3_readlane_u32(pos, l), fmb = x3_readlane_u32(mbb, l);
			if (lane < flen) a.bval[a.bo[c] + fmb + lane] = a.bytes[a.chunks[c].byte_off + fpos + lane];
		}
		chb += hbt; cnb += nbt; cmb += pkt & 0xFFFFu; cpos += pkt >> 16;
		(void)prev_info;
		__syncthreads();
	}
}


/* ============================================================================================================
 * Arranging a stream's hits by (context, time) -- what x3_ctxseg_kernel sweeps.  The context keys of a batch are stream-major (a stream's
 * tags / pair ordinals are one contiguous range of the global numbering), so the arrangement is a SEGMENTED sort: one workgroup of four
 * wavefronts per stream does one stable counting-sort pass on an 11-bit digit of the stream-local key.  Wavefront w takes the w-th quarter
 * of the stream's hits (time order); LDS holds one counter row per wavefront, so after a scan over (digit, wavefront) every wavefront
 * knows where its hits of each digit go and the rest is tiles of 64: lanes with the same digit found by ballots (wave_same_mask), rank =
 * earlier lanes of the group, the group's first lane moves the wavefront's cursor.  Local keys beyond 2048 (tag-pair ordinals) take two
 * passes, low digit first.  An alternative to the chip-wide iota + multi-pass radix sort of (key, index) pairs + gather of the hits' tags
 * (X3H_ARRANGE=1); measured slower than those on the 1024-chunk batch (features 48 against 40 ms), so not the default.
 * ============================================================================================================ */
#define X3_ARR_WAVES 4u
#define X3_ARR_THREADS (X3_ARR_WAVES * X3_WAVE)
#ifndef X3_ARR_DBITS
#define X3_ARR_DBITS 11 /* (the emulator build of the tests uses 8, so that small inputs take the two-pass form) */
#endif
#define X3_ARR_DIGITS (1u << X3_ARR_DBITS)
#define X3_ARR_PER (X3_ARR_DIGITS / X3_ARR_THREADS) /* digits per thread in the scan */
struct X3ArrangeArgs {
	const uint32_t *ho;        /* nc+1: hit ranges                                                                           */
	const uint32_t *kbase;     /* per stream: its first global key (first tag id / first pair ordinal)                       */
	const uint32_t *kin;       /* per input entry: global key                                                                */
	const uint32_t *vin;       /* per input entry: hit index; nullptr = the entry's own index (first pass)                   */
	const uint32_t *h_tag;     /* per hit: global tag id; nullptr = not the last pass                                        */
	uint32_t *kout, *vout, *tout; /* out, same ranges: key, hit index, and (last pass) the hit's tag                          */
	uint32_t shift;            /* digit = ((key - kbase) >> shift) & (X3_ARR_DIGITS - 1)                                     */
};

__device__ static void x3_arrange_body(const X3ArrangeArgs &a)
{
	X3_LDS uint32_t cnt[X3_ARR_WAVES][X3_ARR_DIGITS]; /* counts, then each wavefront's cursor per digit */
	X3_LDS uint32_t wtot[X3_ARR_WAVES];
	const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = x3_lane(), wv = tid / X3_WAVE;
	const uint32_t h0 = a.ho[c], h1 = a.ho[c + 1], kb = a.kbase[c];
	const uint32_t nh = h1 - h0;
	if (!nh) return;
	const uint32_t per = (nh + X3_ARR_WAVES - 1) / X3_ARR_WAVES;
	const uint32_t q0 = h0 + wv * per < h1 ? h0 + wv * per : h1, q1 = q0 + per < h1 ? q0 + per : h1; /* this wavefront's quarter */
	const uint64_t below = ((uint64_t)1 << lane) - 1;
	for (uint32_t i = tid; i < X3_ARR_WAVES * X3_ARR_DIGITS; i += X3_ARR_THREADS) (&cnt[0][0])[i] = 0u;
	__syncthreads();
	/* ---- histogram of the quarter (LDS atomics into the wavefront's own row) ---- */
	for (uint32_t i = q0 + lane; i < q1; i += X3_WAVE) atomicAdd(&cnt[wv][((a.kin[i] - kb) >> a.shift) & (X3_ARR_DIGITS - 1u)], 1u);
	__syncthreads();
	/* ---- exclusive scan in (digit, wavefront) order: thread t owns digits [t * PER, (t + 1) * PER) of all four rows ---- */
	{
		static_assert(X3_ARR_PER >= 1 && X3_ARR_PER * X3_ARR_THREADS == X3_ARR_DIGITS, "digits per thread");
		uint32_t v[X3_ARR_PER][X3_ARR_WAVES], s = 0;
#pragma unroll
		for (uint32_t d = 0; d < X3_ARR_PER; d++)
#pragma unroll
			for (uint32_t w = 0; w < X3_ARR_WAVES; w++) { v[d][w] = cnt[w][tid * X3_ARR_PER + d]; s += v[d][w]; }
		const uint32_t incl = x3_wave_incl_scan_u32(s);
		if (lane == X3_WAVE - 1u) wtot[wv] = incl;
		__syncthreads();
		uint32_t ex = h0 + incl - s;
		for (uint32_t w = 0; w < wv; w++) ex += wtot[w];
#pragma unroll
		for (uint32_t d = 0; d < X3_ARR_PER; d++)
#pragma unroll
			for (uint32_t w = 0; w < X3_ARR_WAVES; w++) { cnt[w][tid * X3_ARR_PER + d] = ex; ex += v[d][w]; }
	}
	__syncthreads();
	/* ---- stable scatter of the quarter, 64 entries per trip; the next trip's entries are in flight meanwhile ---- */
	uint32_t nk = 0, nv = 0;
	if (q0 + lane < q1) { nk = a.kin[q0 + lane]; nv = a.vin ? a.vin[q0 + lane] : q0 + lane; }
	for (uint32_t base = q0; base < q1; base += X3_WAVE) {
		const bool valid = base + lane < q1;
		const uint32_t k = nk, v = nv;
		{
			const uint32_t nx = base + X3_WAVE + lane;
			if (nx < q1) { nk = a.kin[nx]; nv = a.vin ? a.vin[nx] : nx; }
		}
		const uint32_t d = ((k - kb) >> a.shift) & (X3_ARR_DIGITS - 1u);
		const uint64_t V = x3_ballot(valid);
		const uint64_t M = wave_same_mask(d, X3_ARR_DBITS, V, valid);
		const uint32_t cur = valid ? cnt[wv][d] : 0u;
		x3_wave_order();
		if (valid && (M & below) == 0) cnt[wv][d] = cur + (uint32_t)x3_popc64(M);
		x3_wave_order();
		if (valid) {
			const uint32_t dst = cur + (uint32_t)x3_popc64(M & below);
			a.kout[dst] = k; a.vout[dst] = v;
			if (a.h_tag) a.tout[dst] = a.h_tag[v];
		}
	}
}

#ifndef X3_EMU
__global__ void __launch_bounds__(X3_WAVE) x3_mtfrank_kernel_s(X3MtfArgs a) { x3_mtfrank_body<2048>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_mtfrank_kernel_l(X3MtfArgs a) { x3_mtfrank_body<16384>(a); }
__global__ void __launch_bounds__(X3_MTFP_WAVES * X3_WAVE) x3_mtfrank_par_kernel(X3MtfArgs a) { x3_mtfrank_par_body(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_ctxseg_kernel_t(X3CtxSegArgs a) { x3_ctxseg_body<512>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_ctxseg_kernel_s(X3CtxSegArgs a) { x3_ctxseg_body<2048>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_ctxseg_kernel_l(X3CtxSegArgs a) { x3_ctxseg_body<8192>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_order0_kernel(X3Order0Args a) { x3_order0_body(a); }
__global__ void __launch_bounds__(X3_TOK_THREADS) x3_tokens_kernel(X3TokArgs a) { x3_tokens_body(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_idxstat_kernel_s(X3IdxStatArgs a) { x3_idxstat_body<2048>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_idxstat_kernel_l(X3IdxStatArgs a) { x3_idxstat_body<X3_STREAM_DMAX>(a); }
__global__ void __launch_bounds__(X3_IDXP_WAVES * X3_WAVE) x3_idxstat_par_kernel(X3IdxStatArgs a) { x3_idxstat_par_body(a); }
__global__ void __launch_bounds__(X3_ARR_THREADS) x3_arrange_kernel(X3ArrangeArgs a) { x3_arrange_body(a); }
#define X3_LAUNCH1(kern, args, nc, st) hipLaunchKernelGGL(kern, dim3(nc), dim3(X3_WAVE), 0, st, args)
#else
static void order0_tramp(void *p) { x3_order0_body(*(const X3Order0Args *)p); }
static void arrange_tramp(void *p) { x3_arrange_body(*(const X3ArrangeArgs *)p); }
static void tokens_tramp(void *p) { x3_tokens_body(*(const X3TokArgs *)p); }
static void idxstat_tramp_s(void *p) { x3_idxstat_body<2048>(*(const X3IdxStatArgs *)p); }
static void idxstat_tramp_l(void *p) { x3_idxstat_body<X3_STREAM_DMAX>(*(const X3IdxStatArgs *)p); }
static void idxstat_tramp_par(void *p) { x3_idxstat_par_body(*(const X3IdxStatArgs *)p); }
#define x3_idxstat_kernel_s idxstat_tramp_s
#define x3_idxstat_kernel_l idxstat_tramp_l
#define x3_order0_kernel order0_tramp
static void mtf_tramp_s(void *p) { x3_mtfrank_body<2048>(*(const X3MtfArgs *)p); }
static void mtf_tramp_l(void *p) { x3_mtfrank_body<16384>(*(const X3MtfArgs *)p); }
static void mtf_tramp_par(void *p) { x3_mtfrank_par_body(*(const X3MtfArgs *)p); }
static void ctx_tramp_t(void *p) { x3_ctxseg_body<512>(*(const X3CtxSegArgs *)p); }
static void ctx_tramp_s(void *p) { x3_ctxseg_body<2048>(*(const X3CtxSegArgs *)p); }
static void ctx_tramp_l(void *p) { x3_ctxseg_body<8192>(*(const X3CtxSegArgs *)p); }
#define x3_mtfrank_kernel_s mtf_tramp_s
#define x3_mtfrank_kernel_l mtf_tramp_l
#define x3_ctxseg_kernel_t ctx_tramp_t
#define x3_ctxseg_kernel_s ctx_tramp_s
#define x3_ctxseg_kernel_l ctx_tramp_l
#define X3_LAUNCH1(kern, args, nc, st) x3emu_launch(kern, (void *)&(args), dim3(nc), dim3(X3_WAVE))
#endif

/* the largest dictionary of the batch decides: small tables (several streams per CU), large tables, or "does not fit" (false) */
bool x3_stream_kernels_fit(uint64_t max_dict) { return max_dict <= X3_STREAM_DMAX; }

int x3_mtf_ranks_run(hipStream_t st, uint32_t nc, uint64_t max_dict, const uint32_t *d_eo, const uint32_t *d_dof, const uint32_t *e_tag,
                     const uint32_t *e_hit, uint32_t *h_rank)
{
	X3MtfArgs a;
	a.eo = d_eo; a.dof = d_dof; a.e_tag = e_tag; a.e_hit = e_hit; a.h_rank = h_rank;
	bool par = max_dict <= X3_MTFP_DMAX;
	if (const char *e = getenv("X3H_MTF_PAR")) par = par && atoi(e) != 0;
	if (par) {
#ifndef X3_EMU
		hipLaunchKernelGGL(x3_mtfrank_par_kernel, dim3(nc), dim3(X3_MTFP_WAVES * X3_WAVE), 0, st, a);
#else
		x3emu_launch(mtf_tramp_par, (void *)&a, dim3(nc), dim3(X3_MTFP_WAVES * X3_WAVE));
#endif
	} else if (max_dict <= 2048) X3_LAUNCH1(x3_mtfrank_kernel_s, a, nc, st);
	else X3_LAUNCH1(x3_mtfrank_kernel_l, a, nc, st);
	HIPCHK(hipGetLastError());
	return X3H_OK;
}

int x3_ctx_stats_run(hipStream_t st, uint32_t nc, uint64_t max_dict, uint64_t nhits, const uint32_t *d_ho, const uint32_t *d_dof, const uint32_t *kA,
                     const uint32_t *vA, const uint32_t *tA, const uint32_t *h_tag, uint4 *stat, uint32_t *first00)
{
	X3CtxSegArgs a;
	a.ho = d_ho; a.dof = d_dof; a.kA = kA; a.vA = vA; a.tA = tA; a.h_tag = h_tag; a.stat = stat; a.first00 = first00;
	uint32_t b = 1; while (b < 32 && (max_dict >> b)) b++;
	a.dbits_max = b;
	/* contexts are independent, so a stream's range is cut (at context boundaries) over several wavefronts -- MANY of them: ~512 hits each, and
	 * all wavefronts of a stream on one XCD (see X3CtxSegArgs::xcd).  The kernel's cost is its scattered 16-byte result stores (without them it
	 * takes half the time): with a handful of wavefronts per stream the chip works on every stream's 1-2 MB of results at once, no cache holds
	 * them and every store is its own HBM write; with a stream's results finished by ~100 wavefronts within microseconds the lines fill in L2
	 * (1024 x 256 KiB of text: features 28 -> 20 ms; profiles/r03_ctx_wavefronts_per_stream_sweep.txt) */
	uint64_t want = nc ? (nhits / nc + 511) / 512 : 1;
	uint32_t nsub = want < 1 ? 1u : want > 1024 ? 1024u : (uint32_t)want;
	if (const char *e = getenv("X3H_CTX_SUB")) { const int v = atoi(e); if (v >= 1 && v <= 1024) nsub = (uint32_t)v; }
	a.nsub = nsub; a.nc = nc; a.xcd = nc >= 8 ? 1u : 0u;
	if (const char *e = getenv("X3H_CTX_XCD")) a.xcd = atoi(e) ? 1u : 0u;
	const uint32_t nblk = a.xcd ? ((nc + 7u) & ~7u) * nsub : nc * nsub;
	if (max_dict <= 512) X3_LAUNCH1(x3_ctxseg_kernel_t, a, nblk, st);
	else if (max_dict <= 2048) X3_LAUNCH1(x3_ctxseg_kernel_s, a, nblk, st);
	else X3_LAUNCH1(x3_ctxseg_kernel_l, a, nblk, st);
	HIPCHK(hipGetLastError());
	return X3H_OK;
}

int x3_order0_run(hipStream_t st, uint32_t nc, const uint32_t *d_mo, const uint32_t *lval, uint32_t *lsm, uint32_t *leq,
                  const uint32_t *d_bo, const uint32_t *bval, uint32_t *bsm, uint32_t *beq)
{
	X3Order0Args a;
	a.off[0] = d_mo; a.val[0] = lval; a.smaller[0] = lsm; a.equal[0] = leq;
	a.off[1] = d_bo; a.val[1] = bval; a.smaller[1] = bsm; a.equal[1] = beq;
	a.nc = nc;
	X3_LAUNCH1(x3_order0_kernel, a, 2 * nc, st);
	HIPCHK(hipGetLastError());
	return X3H_OK;
}

int x3_idxstat_run(hipStream_t st, uint32_t nc, uint64_t max_dict, const uint32_t *d_ho, const uint32_t *evfinal, const uint32_t *lrank,
                   const uint32_t *lhit, const uint32_t *h_dk, uint32_t *rfreq, uint32_t *rcum, uint32_t *itot)
{
	X3IdxStatArgs a;
	a.ho = d_ho; a.evfinal = evfinal; a.lrank = lrank; a.lhit = lhit; a.h_dk = h_dk; a.rfreq = rfreq; a.rcum = rcum; a.itot = itot;
	uint32_t b = 1; while (b < 32 && (max_dict >> b)) b++;
	a.dbits_max = b;
	bool par = max_dict <= X3_IDXP_DMAX; /* (a rank is below the stream's dictionary size) */
	if (const char *e = getenv("X3H_IDX_PAR")) par = par && atoi(e) != 0;
	if (par) {
#ifndef X3_EMU
		hipLaunchKernelGGL(x3_idxstat_par_kernel, dim3(nc), dim3(X3_IDXP_WAVES * X3_WAVE), 0, st, a);
#else
		x3emu_launch(idxstat_tramp_par, (void *)&a, dim3(nc), dim3(X3_IDXP_WAVES * X3_WAVE));
#endif
	} else if (max_dict <= 2048) X3_LAUNCH1(x3_idxstat_kernel_s, a, nc, st);
	else X3_LAUNCH1(x3_idxstat_kernel_l, a, nc, st);
	HIPCHK(hipGetLastError());
	return X3H_OK;
}

int x3_tokens_run(hipStream_t st, uint32_t nc, const X3Chunk *d_chunks, const X3ParseResult *d_parsed, const uint32_t *tok_info, const uint8_t *dict_len,
                  uint32_t *tok_pos, uint32_t *tok_hb, uint32_t *tok_nb, uint32_t *tok_mb, const uint32_t *d_ho, const uint32_t *d_eo, const uint32_t *d_dof,
                  uint32_t *h_tag, uint32_t *h_c1, uint32_t *h_pv, uint32_t *h_dk, uint32_t *h_step, uint32_t *e_tag, uint32_t *e_hit,
                  const uint8_t *d_bytes, const uint32_t *d_mo, const uint32_t *d_bo, uint32_t *lval, uint32_t *bval)
{
	X3TokArgs a;
	a.bytes = d_bytes; a.mo = d_mo; a.bo = d_bo; a.lval = lval; a.bval = bval;
	a.chunks = d_chunks; a.parsed = d_parsed; a.tok_info = tok_info; a.dict_len = dict_len;
	a.tok_pos = tok_pos; a.tok_hb = tok_hb; a.tok_nb = tok_nb; a.tok_mb = tok_mb; a.ho = d_ho; a.eo = d_eo; a.dof = d_dof;
	a.h_tag = h_tag; a.h_c1 = h_c1; a.h_pv = h_pv; a.h_dk = h_dk; a.h_step = h_step; a.e_tag = e_tag; a.e_hit = e_hit;
#ifndef X3_EMU
	hipLaunchKernelGGL(x3_tokens_kernel, dim3(nc), dim3(X3_TOK_THREADS), 0, st, a);
#else
	x3emu_launch(tokens_tramp, (void *)&a, dim3(nc), dim3(X3_TOK_THREADS));
#endif
	HIPCHK(hipGetLastError());
	return X3H_OK;
}

/* hits of every stream arranged by (key, time): kA = key, vA = hit, tA = the hit's tag.  max_local = largest stream-local key of the batch
 * (decides one pass or two); tmpk / tmpv: nH-entry temporaries for the two-pass form */
int x3_arrange_run(hipStream_t st, uint32_t nc, const uint32_t *d_ho, const uint32_t *kbase, uint64_t max_local, const uint32_t *key, const uint32_t *h_tag,
                   uint32_t *kA, uint32_t *vA, uint32_t *tA, uint32_t *tmpk, uint32_t *tmpv)
{
	if (max_local >= (uint64_t)X3_ARR_DIGITS * X3_ARR_DIGITS) return X3H_E_INTERNAL; /* (the caller keeps the chip-wide sort for such a batch) */
	X3ArrangeArgs a;
	a.ho = d_ho; a.kbase = kbase;
	const bool two = max_local >= X3_ARR_DIGITS;
	for (int pass = 0; pass < (two ? 2 : 1); pass++) {
		const bool last = pass == (two ? 1 : 0);
		a.kin = pass == 0 ? key : tmpk; a.vin = pass == 0 ? nullptr : tmpv;
		a.kout = last ? kA : tmpk; a.vout = last ? vA : tmpv; a.tout = tA; a.h_tag = last ? h_tag : nullptr;
		a.shift = pass == 0 ? 0u : (uint32_t)X3_ARR_DBITS;
#ifndef X3_EMU
		hipLaunchKernelGGL(x3_arrange_kernel, dim3(nc), dim3(X3_ARR_THREADS), 0, st, a);
#else
		x3emu_launch(arrange_tramp, (void *)&a, dim3(nc), dim3(X3_ARR_THREADS));
#endif
		HIPCHK(hipGetLastError());
	}
	return X3H_OK;
}

/* ============================================================================================================
 * The hits of every stream grouped by (key, time), by ONE WORKGROUP PER STREAM with the tile machinery of scan3.hip (seg_rank.h): the two
 * arrangements of the context statistics (key = context1 = the previous tag, then key = the ordinal of the tag pair; reference: context.c:20-56
 * keeps one list per context and appends in time order -- the arrangement IS those lists laid end to end).  Keys are global numbers whose
 * stream-local part (key - kbase[stream]) decides; a stream's hits are a contiguous segment of the hit arrays, so the stable sort of a segment on
 * the local key equals what the chip-wide stable radix sort of all hits on the global key gives (rocPRIM onesweep, 3 + 4 passes of 8 bits over
 * ~100 M pairs for 1024 streams = 9.8 ms and 35.7 GB of HBM traffic on the many-chunk batch: scattered 8-byte stores into 256 runs spread
 * over the whole array).  Here a pass scatters into 256 runs inside the stream's own segment (a few hundred KB: they stay in the XCD's L2 between
 * tiles), and a stream needs only as many passes as ITS local keys have digits:
 *   sweep 0  one histogram per pass of the local keys (eight copies in LDS, as in scan3.hip);
 *   pass p   stable counting-sort pass on digit p of the local key in tiles of 4096: ballot ranking per wavefront, [digit][wave] counters, tile
 *            staged in LDS, out as runs.  The first pass makes the values (hit numbers) itself; the last adds kbase back.
 * x3_arrange_kernel above (four wavefronts per stream, 11-bit digits, each wavefront a chain of LDS round trips over its quarter) was slower than
 * rocPRIM and stays behind X3H_ARRANGE=1.
 * Measured (1024 streams, 106 M hits, nothing else on the chip: profiles/r04_many_chunks_pmc_traffic.json): by context1 (local keys < 282: one pass of 9 bits;
 * as 5 + 4 bits: 2.3 ms) 1.5 ms, by pair (local keys < 41 289: two passes of 8 bits) 2.7 ms, 11.9 GB of HBM traffic -- against 9.8 ms + two index fills and
 * 35.7 GB for the two rocPRIM sorts.
 * (In a kernel trace of the product the first sort shows 5 ms: the move-to-front ranks run beside it on their own stream, by design.)  One more pass
 * over all-zero digits costs 1.0 ms, the histogram sweep 0.3-0.7 ms; LDS bank conflicts were 74 % of the LDS cycles with the counters of a digit 16
 * words apart (every lane of a wavefront in one of two banks): X3_SEG_CS = 17. */
#define X3_SSORT_MAXPASS 3u
#define X3_SSORT_CS X3_SEG_CS
/* one more entry of digit d in a [copy][digit] histogram in LDS.  The lanes that hold the same digit as the wavefront's first valid lane are counted by
 * that lane alone: context1 of an incompressible stream is 0 for most hits (both contexts restart behind a new fragment, x3.c:424-425), a stream's
 * higher key digits are 0 for all -- and 64 atomics on one LDS word are executed one after the other (measured: the histogram sweep of such a batch
 * took longer than its two sorting passes). */
template <uint32_t ND>
__device__ static __forceinline__ void segsort_count(uint32_t *hist8, uint32_t d, bool valid, uint32_t lane)
{
	const uint64_t v = x3_ballot(valid);
	if (!v) return;
	const uint32_t first = (uint32_t)x3_ctz64(v), d0 = x3_readlane_u32(d, first);
	const uint64_t same = x3_ballot(valid && d == d0);
	if (lane == first) atomicAdd(&hist8[(lane & 7u) * ND + d0], (uint32_t)x3_popc64(same));
	else if (valid && d != d0) atomicAdd(&hist8[(lane & 7u) * ND + d], 1u);
}
struct X3SegSortArgs {
	const uint32_t *ho, *kbase; /* hit offsets (nstreams + 1) and key base per stream */
	const uint32_t *kin;        /* global keys per hit */
	uint32_t *kout, *vout;      /* sorted: global key, hit number */
	uint32_t *tk, *tv;          /* temporaries (more than one pass) */
	uint32_t npass, dbits;      /* passes, bits per digit (<= 8): the key's bits spread evenly over the passes */
	/* gen != nullptr: the keys are not read but MADE in sweep 0 and stored to gen (then read from there) -- the context0 group of every hit (x3.c:139-147):
	 * the ordinal of the pair (previous context1, context1), which is the pair the previous hit registered = the rank of the hit that first used it
	 * (P[first hit]); hits without a previous hit in their fragment: the stream's pair (0, 0) once it exists, else context 0.  Made here, the scattered
	 * reads of P stay inside the stream's own segment and the L2 of the one XCD its workgroup runs on (as a chip-wide element-wise pass: 19 GB, 5 ms). */
	uint32_t *gen;
	const uint32_t *h_pv, *P, *first00, *ord00;
	const uint4 *stat;
};
/* DB = 8: digits of up to 8 bits, up to three passes.  DB = 9: ONE pass of 9-bit digits -- context1 of streams with 256..511 dictionary elements (text at -t 256)
 * is sorted in one pass instead of two (the histogram's eight copies of 512 counters fill the staging buffer: no room for a second pass's) */
template <uint32_t DB>
__device__ static void x3_segsort_body(const X3SegSortArgs &a)
{
	constexpr uint32_t ND = 1u << DB, PER = ND * X3_SEG_WAVES / X3_SEG_THREADS; /* digits; counters per thread in the table scan (4 or 8) */
	X3_LDS uint32_t cnt[ND * X3_SSORT_CS]; /* [digit][wave], a digit's sixteen counters X3_SSORT_CS words apart: the lanes of a wavefront (one wave number, many digits) meet in every bank, not in two */
	X3_LDS uint32_t stk[X3_SEG_TILE], stv[X3_SEG_TILE]; /* the tile in sorted order (sweep 0: the histograms, 8 copies x 256 x passes) */
	X3_LDS uint32_t bbase[X3_SSORT_MAXPASS][ND], bcur[ND];
	X3_LDS __attribute__((aligned(16))) uint32_t wtot[X3_SEG_WAVES];
	const uint32_t tid = threadIdx.x, lane = x3_lane(), wv = tid / X3_WAVE;
	const uint32_t lo = a.ho[blockIdx.x], n = a.ho[blockIdx.x + 1] - lo, kb = a.kbase[blockIdx.x], npass = a.npass, db = a.dbits, dmask = (1u << db) - 1u;
	if (!n) return;
	/* sweep 0: [copy][digit] counters of pass 0 in stk[0..2047], of pass 1 in stk[2048..4095], of pass 2 in stv[0..2047] */
	for (uint32_t i = tid; i < 2048u; i += X3_SEG_THREADS) { stk[i] = 0u; stk[2048u + i] = 0u; stv[i] = 0u; } /* (DB = 9: pass 0 alone, all of stk) */
	__syncthreads();
	for (uint32_t b0 = 0; b0 < n; b0 += X3_SEG_THREADS) { /* (uniform trip count: wave operations inside) */
		const uint32_t i = b0 + tid;
		const bool valid = i < n;
		uint32_t k = 0u;
		if (valid) {
			if (a.gen) {
				const uint32_t gh = lo + i, f00 = a.first00[blockIdx.x];
				const uint32_t g = a.h_pv[gh] ? a.P[a.stat[gh - 1u].w & 0x7FFFFFFFu] : ((f00 != NONE32 && f00 < gh) ? a.ord00[blockIdx.x] : kb);
				a.gen[gh] = g;
				k = g - kb;
			} else k = a.kin[lo + i] - kb;
		}
		segsort_count<ND>(stk, k & dmask, valid, lane);
		if (npass > 1u) segsort_count<ND>(stk + 2048u, (k >> db) & dmask, valid, lane);
		if (npass > 2u) segsort_count<ND>(stv, (k >> (2u * db)) & dmask, valid, lane);
	}
	__syncthreads();
	for (uint32_t ps = 0; ps < npass; ps++) { /* bucket bases of every pass: exclusive scan of its histogram */
		const uint32_t *h8 = ps == 0u ? stk : ps == 1u ? stk + 2048u : stv;
		uint32_t h = 0, incl = 0;
		if (tid < ND) {
#pragma unroll
			for (uint32_t k = 0; k < 8; k++) h += h8[k * ND + tid];
			incl = x3_wave_incl_scan_u32(h);
			if (lane == X3_WAVE - 1u) wtot[wv] = incl;
		}
		__syncthreads();
		if (tid < ND) bbase[ps][tid] = seg_waves_before(wtot, wv) + incl - h;
		__syncthreads();
	}
	for (uint32_t ps = 0; ps < npass; ps++) {
		const bool first = ps == 0u, last = ps + 1u == npass;
		/* buffers: the last pass lands in (kout, vout); the ones before alternate so that no pass reads what it writes */
		const bool to_out = ((npass - 1u - ps) & 1u) == 0u;
		const uint32_t *ink = first ? (a.gen ? a.gen : a.kin) + lo : (to_out ? a.tk : a.kout) + lo, *inv = first ? nullptr : (to_out ? a.tv : a.vout) + lo;
		uint32_t *outk = (to_out ? a.kout : a.tk) + lo, *outv = (to_out ? a.vout : a.tv) + lo;
		const uint32_t sh = db * ps, ksub = first ? kb : 0u, kadd = last ? kb : 0u;
		if (tid < ND) bcur[tid] = bbase[ps][tid];
		uint32_t nk[X3_SEG_E], nv[X3_SEG_E];
		{
			const uint32_t i0 = wv * (X3_SEG_E * X3_WAVE) + lane;
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) {
				const uint32_t i = i0 + e * X3_WAVE;
				nk[e] = i < n ? ink[i] - ksub : 0u; nv[e] = i < n ? (first ? lo + i : inv[i]) : 0u;
			}
		}
		for (uint32_t t0 = 0; t0 < n; t0 += X3_SEG_TILE) {
			for (uint32_t i = tid; i < ND * X3_SSORT_CS; i += X3_SEG_THREADS) cnt[i] = 0u;
			uint32_t ik[X3_SEG_E], iv[X3_SEG_E], rk[X3_SEG_E];
			const uint32_t i0 = t0 + wv * (X3_SEG_E * X3_WAVE) + lane;
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) { ik[e] = nk[e]; iv[e] = nv[e]; }
			if (t0 + X3_SEG_TILE < n) { /* the next tile's entries are on their way while this one is ranked */
#pragma unroll
				for (uint32_t e = 0; e < X3_SEG_E; e++) {
					const uint32_t i = i0 + X3_SEG_TILE + e * X3_WAVE;
					nk[e] = i < n ? ink[i] - ksub : 0u; nv[e] = i < n ? (first ? lo + i : inv[i]) : 0u;
				}
			}
			__syncthreads();
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) {
				const bool valid = i0 + e * X3_WAVE < n;
				const uint32_t d = (ik[e] >> sh) & dmask;
				uint32_t mlo, mhi;
				seg_match<DB>(d, valid, mlo, mhi);
				const uint32_t lower = seg_lower(mlo, mhi);
				const uint32_t prev = valid ? cnt[d * X3_SSORT_CS + wv] : 0u;
				x3_wave_order();
				if (valid && lower == 0u) cnt[d * X3_SSORT_CS + wv] = prev + seg_size(mlo, mhi);
				x3_wave_order();
				rk[e] = prev + lower;
			}
			__syncthreads();
			/* exclusive scan of the counter table in (digit, wave) order = the tile-sorted order */
			uint32_t *const cp = &cnt[((tid * PER) / X3_SEG_WAVES) * X3_SSORT_CS + (tid * PER) % X3_SEG_WAVES]; /* PER waves' counters of one digit */
			uint32_t c[PER], s = 0;
#pragma unroll
			for (uint32_t k = 0; k < PER; k++) { c[k] = cp[k]; s += c[k]; }
			const uint32_t incl = x3_wave_incl_scan_u32(s);
			if (lane == X3_WAVE - 1u) wtot[wv] = incl;
			__syncthreads();
			uint32_t ex = incl - s + seg_waves_before(wtot, wv);
#pragma unroll
			for (uint32_t k = 0; k < PER; k++) { cp[k] = ex; ex += c[k]; }
			__syncthreads();
			const uint32_t tile_n = n - t0 < X3_SEG_TILE ? n - t0 : X3_SEG_TILE;
			uint32_t delta = 0; /* entries of digit `tid` in this tile */
			if (tid < ND) delta = (tid < ND - 1u ? cnt[(tid + 1u) * X3_SSORT_CS] : tile_n) - cnt[tid * X3_SSORT_CS];
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) {
				if (i0 + e * X3_WAVE < n) { const uint32_t at = cnt[((ik[e] >> sh) & dmask) * X3_SSORT_CS + wv] + rk[e]; stk[at] = ik[e]; stv[at] = iv[e]; }
			}
			__syncthreads();
#pragma unroll
			for (uint32_t e = 0; e < X3_SEG_E; e++) {
				const uint32_t i = e * X3_SEG_THREADS + tid;
				if (i < tile_n) {
					const uint32_t k = stk[i], d = (k >> sh) & dmask;
					const uint32_t dest = bcur[d] + (i - cnt[d * X3_SSORT_CS]);
					outk[dest] = k + kadd; outv[dest] = stv[i];
				}
			}
			__syncthreads();
			if (tid < ND) bcur[tid] += delta; /* (read again only behind the next tile's barriers) */
		}
		__syncthreads();
	}
}
#ifndef X3_EMU
__global__ void __launch_bounds__(X3_SEG_THREADS, 8) x3_segsort_kernel(X3SegSortArgs a) { x3_segsort_body<8>(a); }
__global__ void __launch_bounds__(X3_SEG_THREADS, 8) x3_segsort9_kernel(X3SegSortArgs a) { x3_segsort_body<9>(a); }
#else
static void segsort_tramp(void *p) { x3_segsort_body<8>(*(const X3SegSortArgs *)p); }
static void segsort9_tramp(void *p) { x3_segsort_body<9>(*(const X3SegSortArgs *)p); }
#endif
/* -> X3H_OK; the caller keeps the chip-wide sort when max_local needs more than three 8-bit passes */
int x3_segsort_run(hipStream_t st, uint32_t nc, const uint32_t *d_ho, const uint32_t *kbase, uint64_t max_local, const uint32_t *key,
                   uint32_t *kA, uint32_t *vA, uint32_t *tmpk, uint32_t *tmpv, const X3SegSortGen *gen)
{
	if (max_local >= ((uint64_t)1 << (8u * X3_SSORT_MAXPASS))) return X3H_E_INTERNAL;
	X3SegSortArgs a;
	a.ho = d_ho; a.kbase = kbase; a.kin = key; a.kout = kA; a.vout = vA; a.tk = tmpk; a.tv = tmpv;
	a.gen = nullptr; a.h_pv = a.P = a.first00 = a.ord00 = nullptr; a.stat = nullptr;
	if (gen) { a.gen = gen->out; a.h_pv = gen->h_pv; a.P = gen->P; a.first00 = gen->first00; a.ord00 = gen->ord00; a.stat = gen->stat; }
	a.npass = max_local < 256u ? 1u : max_local < 65536u ? 2u : 3u;
	if (const char *e = getenv("X3H_SEGSORT_PASSES")) { const int v = atoi(e); if (v > (int)a.npass && v <= (int)X3_SSORT_MAXPASS) a.npass = (uint32_t)v; } /* (tests: more passes than the keys need) */
	/* the key's bits in equal shares: 9 bits are two passes of 5 and 4, not of 8 and 1 -- a pass of 2^5 runs keeps its partly written lines in L2 (header) */
	uint32_t kbits = 1; while (kbits < 24u && (max_local >> kbits)) kbits++;
	a.dbits = (kbits + a.npass - 1u) / a.npass;
	if (getenv("X3H_SEGSORT_DBITS8")) a.dbits = 8u; /* (tests / measurements: whole bytes) */
	bool nine = kbits == 9u && a.npass == 2u && !getenv("X3H_SEGSORT_DBITS8"); /* one pass of nine bits instead of 5 + 4 */
	if (const char *e = getenv("X3H_SEGSORT_NINE")) nine = e[0] == '1' ? (kbits <= 9u && a.npass <= 2u) : false; /* (1: also for shorter keys -- tests; 0: never) */
	if (nine) { a.npass = 1u; a.dbits = 9u; }
	if (getenv("X3H_DEBUG")) fprintf(stderr, "[x3h] per-stream sort: %u streams, largest local key %llu, %u passes of %u bits\n", nc, (unsigned long long)max_local, a.npass, a.dbits);
#ifndef X3_EMU
	if (nine) hipLaunchKernelGGL(x3_segsort9_kernel, dim3(nc), dim3(X3_SEG_THREADS), 0, st, a);
	else hipLaunchKernelGGL(x3_segsort_kernel, dim3(nc), dim3(X3_SEG_THREADS), 0, st, a);
#else
	x3emu_launch(nine ? segsort9_tramp : segsort_tramp, (void *)&a, dim3(nc), dim3(X3_SEG_THREADS));
#endif
	HIPCHK(hipGetLastError());
	return X3H_OK;
}
