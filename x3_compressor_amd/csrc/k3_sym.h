/*
 * k3_sym.h -- what the symbol assembly passes of K3 (code2.hip: whole streams / growing prefixes; code4.hip: slices) share: the operand format of
 * the coder chain (x3_ac2_kernel) and the terms of the size estimates.
 */
#ifndef X3_K3_SYM_H
#define X3_K3_SYM_H

#include "x3_host.h"
#include <math.h>

#define X3_AC2_G 8u   /* symbols per stored chain state */
#define X3_SYM_PAD 72 /* readable operand entries behind the last symbol (the chain fetches one group of 8 ahead) */

/* range / total as a multiply-shift (Granlund-Montgomery, N = 31): L = ceil(log2 total) >= 1, m = ceil(2^(31+L)/total) in [2^31, 2^32),
 * floor(range*m / 2^(31+L)) == floor(range/total) for every range <= 2^31 because m*total - 2^(31+L) < total <= 2^L.  The chain takes the
 * high product word (s_mul_hi_u32) and shifts it by L-1: two instructions.  total == 1 (a context or index model with a single symbol of
 * frequency 1) would need m = 2^32 -- but such a symbol is a NO-OP for the coder (step = range, the interval does not change, nothing is
 * renormalised): it is marked (w = X3_SYM_NOOP) and dropped from the chain's input by the compaction pass of x3_code_v2_run.
 * Computed per symbol by the parallel assembly kernels, off the serial chain. */
#define X3_SYM_NOOP 0xFFFFFFFFu
__device__ static __forceinline__ uint4 x3_make_symbol(uint32_t cum, uint32_t freq, uint32_t total)
{
	uint4 q;
	q.x = cum; q.y = freq;
	if (total <= 1) { q.z = 0; q.w = X3_SYM_NOOP; return q; }
	const uint32_t L = 32u - (uint32_t)x3_clz32(total - 1);
	q.z = (uint32_t)((((uint64_t)1 << (31 + L)) + total - 1) / total);
	q.w = L - 1;
	return q;
}


/* one term of the reference's size estimates (x3.c:52-55: prob_to_bits = -log2f): -log2 in double, rounded to single */
#define X3_EST_NONE 0xFFu
__device__ static __forceinline__ float x3_est_term(float prob) { return (float)(-log2((double)prob)); }

#endif /* X3_K3_SYM_H */
