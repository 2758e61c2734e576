/*
 * k3_wave.h -- wave-level building blocks shared by the per-stream K3 kernels (code3.hip: whole streams; code4.hip: slices of streams with
 * carried state): ballot-based "same value" / "count smaller before" inside a tile of 64 records, and the move-to-front tile loop.
 */
#ifndef X3_K3_WAVE_H
#define X3_K3_WAVE_H

#include "x3_host.h"

#ifndef NONE32
#define NONE32 0xFFFFFFFFu
#endif
#ifndef NONE16
#define NONE16 0xFFFFu
#endif

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

/* the events [e0, e1) of one stream, 64 per trip, on the list (lst, pos0) that holds Dcur elements when the range starts.  lst / pos0 are PRIVATE to the
 * calling wavefront (every user gives each wavefront its own tables), so a compiler barrier orders its LDS accesses (x3_wave_order, simt.h): the fence
 * of x3_wave_sync would also wait for the records in flight for the next tile and for the rank stores of this one -- two memory round trips per tile */
__device__ static __forceinline__ void x3_mtf_tiles(const X3MtfArgs &a, uint16_t *lst, uint16_t *pos0, const uint32_t e0, const uint32_t e1, uint32_t Dcur,
                                                    const uint32_t dof, const uint32_t lane)
{
	const uint64_t bit = (uint64_t)1 << lane, below = bit - 1, above = ~(below | bit);
	uint32_t nt_ = 0, nh_ = 0;
	if (e0 < e1) { const uint32_t i0 = e0 + lane < e1 ? e0 + lane : e1 - 1; nt_ = a.e_tag[i0] - dof; nh_ = a.e_hit[i0]; } /* (unconditional loads from clamped indices: see below) */
	for (uint32_t base = e0; base < e1; base += X3_WAVE) {
		const bool valid = base + lane < e1;
		const uint32_t t = nt_, hit = nh_;
		{ /* next tile's records are in flight while this one is resolved (no branch around the loads: behind one the compiler waits for them with vmcnt(0) at once) */
			const uint32_t nx = base + X3_WAVE + lane < e1 ? base + X3_WAVE + lane : e1 - 1;
			nt_ = a.e_tag[nx] - dof; nh_ = a.e_hit[nx];
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
		x3_wave_order();
		if (last_old) pos0[t] = (uint16_t)(p0_tag | 0x8000u); /* mark: this position moves to the front */
		x3_wave_order();
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
				x3_wave_order(); /* the block is read before any lane writes into it */
				if (in && !moved) { lst[np] = (uint16_t)tq; pos0[tq] = (uint16_t)np; }
				after += (uint32_t)x3_popc64(mm);
				x3_wave_order();
			}
		}
		if (last) { const uint32_t np = (uint32_t)x3_popc64(L & above); lst[np] = (uint16_t)t; pos0[t] = (uint16_t)np; }
		Dcur += n_new;
		x3_wave_order();
	}
}


/* lane i gets lane i-1's value (lane 0: its own) */
__device__ static __forceinline__ uint32_t wave_prev_u32(uint32_t v)
{
#ifndef X3_EMU
	return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
#else
	return x3_shfl_up_u32(v, 1);
#endif
}

#endif /* X3_K3_WAVE_H */
