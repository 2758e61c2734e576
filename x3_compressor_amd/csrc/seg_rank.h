/* seg_rank.h -- the workgroup-per-segment counting sort's building blocks, shared by scan3.hip (K1 of many-chunk batches: the n-gram lists of a
 * chunk) and code3.hip (K3 of many-stream batches: a stream's hits grouped by context): tile geometry, the ballot ranking of a wavefront's
 * digits, the per-wave totals. */
#ifndef X3_SEG_RANK_H
#define X3_SEG_RANK_H
#include "x3_host.h"

#define X3_SEG_WAVES (X3_SEG_THREADS / X3_WAVE)
#define X3_SEG_E     4u                            /* elements per thread and tile */
#define X3_SEG_TILE  (X3_SEG_THREADS * X3_SEG_E)
#define X3_SEG_CS    (X3_SEG_WAVES + 1u)            /* words between two digits' counters in the [digit][wave] tables */


/* the lanes (among `valid` ones) that hold the same NB-bit digit as this lane, as the two halves of a lane mask: one ballot per digit bit.  Per
 * bit and half ONE three-input operation  m & ~(ballot ^ -bit)  (gfx950: v_bitop3_b32); written on 64-bit values with a per-lane select the
 * compiler spends eleven vector instructions per bit, and the ranking was half of a sorting pass's instructions. */
template <uint32_t NB>
__device__ static __forceinline__ void seg_match(uint32_t d, bool valid, uint32_t &mlo, uint32_t &mhi)
{
	const uint64_t v = x3_ballot(valid);
	mlo = (uint32_t)v; mhi = (uint32_t)(v >> 32);
#pragma unroll
	for (uint32_t b = 0; b < NB; b++) {
#ifndef X3_EMU
		const uint32_t sgn = (uint32_t)__builtin_amdgcn_sbfe((int)d, b, 1u);
		const uint64_t bal = x3_ballot(sgn != 0u);
		mlo = __builtin_amdgcn_bitop3_b32(mlo, (uint32_t)bal, sgn, 0x90); /* 0x90: a & ~(b ^ c) */
		mhi = __builtin_amdgcn_bitop3_b32(mhi, (uint32_t)(bal >> 32), sgn, 0x90);
#else
		const uint32_t sgn = 0u - ((d >> b) & 1u);
		const uint64_t bal = x3_ballot(sgn != 0u);
		mlo &= ~((uint32_t)bal ^ sgn);
		mhi &= ~((uint32_t)(bal >> 32) ^ sgn);
#endif
	}
}
/* this lane's rank among the lanes of the mask, and the mask's size */
__device__ static __forceinline__ uint32_t seg_lower(uint32_t mlo, uint32_t mhi)
{
#ifndef X3_EMU
	return __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
#else
	const uint64_t m = ((uint64_t)mhi << 32) | mlo;
	return (uint32_t)x3_popc64(m & (((uint64_t)1 << x3_lane()) - 1u));
#endif
}
__device__ static __forceinline__ uint32_t seg_size(uint32_t mlo, uint32_t mhi) { return (uint32_t)__builtin_popcount(mlo) + (uint32_t)__builtin_popcount(mhi); }

/* sum of the first `wv` of the workgroup's per-wave totals (wt: X3_SEG_WAVES = 16 words, 16-byte aligned): four wide reads in flight together --
 * a loop over wv words is a chain of up to fifteen LDS round trips on every tile */
__device__ static __forceinline__ uint32_t seg_waves_before(const uint32_t *wt, uint32_t wv)
{
	static_assert(X3_SEG_WAVES == 16u, "four uint4");
	const uint4 a = ((const uint4 *)wt)[0], b = ((const uint4 *)wt)[1], c = ((const uint4 *)wt)[2], d = ((const uint4 *)wt)[3];
	const uint32_t t[16] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w };
	uint32_t s = 0;
#pragma unroll
	for (uint32_t w = 0; w < 16u; w++) s += w < wv ? t[w] : 0u;
	return s;
}

#endif
