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
#include "x3_host.h"

#define NONE32 0xFFFFFFFFu
#define NONE16 0xFFFFu

/* lanes (among `valid`) that hold the same `v` as this lane; `bits` covers every value */
__device__ static __forceinline__ uint64_t wave_same_mask(uint32_t v, int bits, uint64_t valid, bool me_valid)
{
	uint64_t m = valid;
	for (int b = 0; b < bits; b++) {
		const uint64_t B = x3_ballot(me_valid && ((v >> b) & 1u));
		m &= ((v >> b) & 1u) ? B : ~B;
	}
	return me_valid ? m : 0;
}

/* #{ j in cand : v_j < x }  where v_j is lane j's `v` and x, cand are this lane's; `bits` covers every v and x */
__device__ static __forceinline__ uint32_t wave_count_less(uint32_t v, uint32_t x, int bits, uint64_t cand)
{
	uint32_t cnt = 0;
	uint64_t A = cand;
	for (int b = bits - 1; b >= 0; b--) {
		const uint64_t B = x3_ballot((v >> b) & 1u);
		if ((x >> b) & 1u) { cnt += (uint32_t)x3_popc64(A & ~B); A &= B; }
		else A &= ~B;
	}
	return cnt;
}

__device__ static __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#ifndef X3_EMU
	/* inclusive max-scan by DPP row shifts + row broadcasts (the sequence of x3_wave_incl_scan_u32), then lane 63 */
	int x = (int)v;
#define X3_MAX_STEP(ctrl, rows) { const int y = __builtin_amdgcn_update_dpp(x, x, ctrl, rows, 0xf, false); x = (uint32_t)y > (uint32_t)x ? y : x; }
	X3_MAX_STEP(0x111, 0xf) X3_MAX_STEP(0x112, 0xf) X3_MAX_STEP(0x114, 0xf) X3_MAX_STEP(0x118, 0xf) X3_MAX_STEP(0x142, 0xa) X3_MAX_STEP(0x143, 0xc)
#undef X3_MAX_STEP
	return (uint32_t)__builtin_amdgcn_readlane(x, 63);
#else
	for (int d = 1; d < X3_WAVE; d <<= 1) { const uint32_t u = x3_shfl_xor_u32(v, d); v = u > v ? u : v; }
	return v;
#endif
}

__device__ static __forceinline__ int dev_bits_for(uint32_t maxval) { return maxval ? 32 - x3_clz32(maxval) : 1; }

/* ============================================================================================================
 * Move-to-front ranks.  Events of a stream in time order: a hit touches its element, a new fragment inserts one at the front
 * (x3.c:392-397,414-427 -> dict_update_costs).  The `index` model_index1 codes for a hit is the element's position in that list.
 * LDS: lst[pos] = tag, pos0[tag] = pos, both as of the tile's first event.  For the 64 events of a tile:
 *   first touch of its tag inside the tile : rank = pos0 + #{distinct tags touched earlier in the tile that stood BEHIND it}
 *   repeated touch (previous one at lane p) : rank = #{distinct tags touched in lanes (p, i)} = #{ j in (p, i) : prev_j < p }
 * then the list is rebuilt: the tile's tags in order of their last touch, the untouched ones behind them in their old order.
 * ============================================================================================================ */
struct X3MtfArgs {
	const uint32_t *eo;        /* nc+1: event ranges                                     */
	const uint32_t *dof;       /* per chunk: first global tag id                         */
	const uint32_t *e_tag;     /* per event: global tag id                               */
	const uint32_t *e_hit;     /* per event: hit index, NONE32 for an insertion          */
	uint32_t *h_rank;          /* out per hit                                            */
};

template <uint32_t DMAX>
__device__ static void x3_mtfrank_body(const X3MtfArgs &a)
{
	X3_LDS uint16_t lst[DMAX];
	X3_LDS uint16_t pos0[DMAX];
	const uint32_t c = blockIdx.x, lane = x3_lane();
	const uint32_t e0 = a.eo[c], e1 = a.eo[c + 1], dof = a.dof[c];
	const uint64_t bit = (uint64_t)1 << lane, below = bit - 1, above = ~(below | bit);
	uint32_t Dcur = 0;
	uint32_t nt_ = 0, nh_ = 0;
	if (e0 + lane < e1) { nt_ = a.e_tag[e0 + lane] - dof; nh_ = a.e_hit[e0 + lane]; }
	for (uint32_t base = e0; base < e1; base += X3_WAVE) {
		const bool valid = base + lane < e1;
		const uint32_t t = nt_, hit = nh_;
		{ /* next tile's records are in flight while this one is resolved */
			const uint32_t nx = base + X3_WAVE + lane;
			if (nx < e1) { nt_ = a.e_tag[nx] - dof; nh_ = a.e_hit[nx]; }
		}
		const uint64_t V = x3_ballot(valid);
		const bool isnew = valid && hit == NONE32;
		const uint64_t NEW = x3_ballot(isnew);
		const uint32_t n_new = (uint32_t)x3_popc64(NEW);
		const int tbits = dev_bits_for(Dcur + n_new); /* tags of this tile are < Dcur + n_new */
		const uint64_t M = wave_same_mask(t, tbits, V, valid);
		const uint64_t E = M & below;
		const uint32_t pl = E ? 64u - (uint32_t)x3_clz64(E) : 0u; /* lane of the previous touch in the tile + 1; 0: none */
		const bool first = valid && E == 0;
		/* position at the tile's start; an element inserted in this tile stands behind every existing one, in insertion order */
		const uint32_t p0 = !valid ? 0u : isnew ? Dcur + (uint32_t)x3_popc64(NEW & below) : (first ? (uint32_t)pos0[t] : 0u);
		const uint64_t F = x3_ballot(first);
		const int pbits = dev_bits_for(Dcur + n_new);
		const uint32_t less_first = wave_count_less(first ? p0 : 0xFFFFFFFFu >> (32 - pbits), p0, pbits, F & below);
		const uint32_t rank_first = p0 + (uint32_t)x3_popc64(F & below) - less_first;
		const uint64_t W = below & ~(pl >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << pl) - 1));
		const uint32_t rank_rep = wave_count_less(pl, pl, 7, W & V);
		if (valid && !isnew) a.h_rank[hit] = first ? rank_first : rank_rep;
		/* ---- rebuild the list ---- */
		const bool last = valid && (M & above) == 0;
		const uint64_t L = x3_ballot(last);
		const uint32_t nt = (uint32_t)x3_popc64(L);
		const bool last_old = last && !x3_popc64(M & NEW); /* an element that existed at the tile's start */
		const uint32_t t_old = nt - n_new;                  /* how many of those were touched */
		/* every lane of a tag saw the same pos0; the last-touch lane needs it too */
		const uint32_t p0_tag = valid && !x3_popc64(M & NEW) ? (uint32_t)pos0[t] : 0u;
		uint32_t qmax = wave_max_u32(last_old ? p0_tag : 0u);
		if (n_new && Dcur) qmax = Dcur - 1;
		x3_wave_sync();
		if (last_old) pos0[t] = (uint16_t)(p0_tag | 0x8000u); /* mark: this position moves to the front */
		x3_wave_sync();
		if (t_old || (n_new && Dcur)) {
			uint32_t after = 0; /* touched old positions in the blocks above the current one */
			for (int blk = (int)(qmax / X3_WAVE); blk >= 0; blk--) {
				const uint32_t q = (uint32_t)blk * X3_WAVE + lane;
				const bool in = q <= qmax && q < Dcur;
				const uint32_t tq = in ? (uint32_t)lst[q] : 0u;
				const bool moved = in && (pos0[tq] & 0x8000u);
				const uint64_t mm = x3_ballot(moved);
				const uint32_t ge = after + (uint32_t)x3_popc64(mm & (above | bit)); /* touched old positions >= q */
				const uint32_t np = nt + q - (t_old - ge);
				x3_wave_sync(); /* the block is read before any lane writes into it */
				if (in && !moved) { lst[np] = (uint16_t)tq; pos0[tq] = (uint16_t)np; }
				after += (uint32_t)x3_popc64(mm);
				x3_wave_sync();
			}
		}
		if (last) { const uint32_t np = (uint32_t)x3_popc64(L & above); lst[np] = (uint16_t)t; pos0[t] = (uint16_t)np; }
		Dcur += n_new;
		x3_wave_sync();
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
	const uint32_t *tA;        /* ... and the hit's global tag id                         */
	uint4 *stat;               /* out per hit: {freq, total, cum, first | isfirst << 31}  */
	uint32_t dbits_max;        /* bits that cover every local tag of the batch            */
};

/* lane i gets lane i-1's value (lane 0: its own) */
__device__ static __forceinline__ uint32_t wave_prev_u32(uint32_t v)
{
#ifndef X3_EMU
	return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
#else
	return x3_shfl_up_u32(v, 1);
#endif
}

template <uint32_t DMAX>
__device__ static void x3_ctxseg_body(const X3CtxSegArgs &a)
{
	X3_LDS uint16_t tpos[DMAX];      /* tag -> list position in the open context, NONE16: absent */
	X3_LDS uint16_t ltag[DMAX];      /* position -> tag                                           */
	X3_LDS uint32_t lfirst[DMAX];    /* position -> hit that added the item                       */
	X3_LDS uint32_t lfreq[DMAX];     /* position -> freq                                          */
	X3_LDS uint32_t lpre[DMAX];      /* position -> cum_freq = sum of the freqs before it (count_cum_freqs, ac.c:6-18), as of the tile's start */
	const uint32_t c = blockIdx.x, lane = x3_lane();
	const uint32_t h0 = a.ho[c], h1 = a.ho[c + 1], dof = a.dof[c];
	const uint64_t bit = (uint64_t)1 << lane, below = bit - 1;
	const int tbits = (int)a.dbits_max; /* covers tags and list positions alike (a list holds each tag once) */
	for (uint32_t i = lane; i < DMAX; i += X3_WAVE) tpos[i] = NONE16;
	bool open = false;
	uint32_t open_g = 0, open_k = 0, open_total = 0;
	uint32_t ng_ = 0, nh_ = 0, nt_ = 0;
	if (h0 + lane < h1) { ng_ = a.kA[h0 + lane]; nh_ = a.vA[h0 + lane]; nt_ = a.tA[h0 + lane]; }
	x3_wave_sync();
	for (uint32_t base = h0; base < h1; base += X3_WAVE) {
		const bool valid = base + lane < h1;
		const uint32_t g = ng_, hit = nh_, t = valid ? nt_ - dof : 0u;
		{ /* the next tile's records are in flight while this one is resolved */
			const uint32_t nx = base + X3_WAVE + lane;
			if (nx < h1) { ng_ = a.kA[nx]; nh_ = a.vA[nx]; nt_ = a.tA[nx]; }
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

#ifndef X3_EMU
__global__ void __launch_bounds__(X3_WAVE) x3_mtfrank_kernel_s(X3MtfArgs a) { x3_mtfrank_body<2048>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_mtfrank_kernel_l(X3MtfArgs a) { x3_mtfrank_body<16384>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_ctxseg_kernel_s(X3CtxSegArgs a) { x3_ctxseg_body<2048>(a); }
__global__ void __launch_bounds__(X3_WAVE) x3_ctxseg_kernel_l(X3CtxSegArgs a) { x3_ctxseg_body<8192>(a); }
#define X3_LAUNCH1(kern, args, nc, st) hipLaunchKernelGGL(kern, dim3(nc), dim3(X3_WAVE), 0, st, args)
#else
static void mtf_tramp_s(void *p) { x3_mtfrank_body<2048>(*(const X3MtfArgs *)p); }
static void mtf_tramp_l(void *p) { x3_mtfrank_body<16384>(*(const X3MtfArgs *)p); }
static void ctx_tramp_s(void *p) { x3_ctxseg_body<2048>(*(const X3CtxSegArgs *)p); }
static void ctx_tramp_l(void *p) { x3_ctxseg_body<8192>(*(const X3CtxSegArgs *)p); }
#define x3_mtfrank_kernel_s mtf_tramp_s
#define x3_mtfrank_kernel_l mtf_tramp_l
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
	if (max_dict <= 2048) X3_LAUNCH1(x3_mtfrank_kernel_s, a, nc, st);
	else X3_LAUNCH1(x3_mtfrank_kernel_l, a, nc, st);
	HIPCHK(hipGetLastError());
	return X3H_OK;
}

int x3_ctx_stats_run(hipStream_t st, uint32_t nc, uint64_t max_dict, const uint32_t *d_ho, const uint32_t *d_dof, const uint32_t *kA,
                     const uint32_t *vA, const uint32_t *tA, uint4 *stat)
{
	X3CtxSegArgs a;
	a.ho = d_ho; a.dof = d_dof; a.kA = kA; a.vA = vA; a.tA = tA; a.stat = stat;
	uint32_t b = 1; while (b < 32 && (max_dict >> b)) b++;
	a.dbits_max = b;
	if (max_dict <= 2048) X3_LAUNCH1(x3_ctxseg_kernel_s, a, nc, st);
	else X3_LAUNCH1(x3_ctxseg_kernel_l, a, nc, st);
	HIPCHK(hipGetLastError());
	return X3H_OK;
}
