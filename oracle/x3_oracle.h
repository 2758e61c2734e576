/*
 * x3_oracle.h -- CPU restatement of the x3 hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle for the MI355X build: an independently written, instance-based (no
 * globals, re-entrant) plain-C restatement of the reference algorithm.  It is NOT part of the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity status: PINNED.  The reference ships no golden vectors of its own, so
 * the oracle is pinned against (a) the real reference compiled from /root/reference into
 * oracle/_ref/x3 (oracle/Makefile, target `ref`) and (b) the committed fixtures under tests/golden/
 * that were generated with that binary (tests/golden/make_golden.py).
 *
 * Every function cites the reference file:line it restates (paths relative to the reference root).
 */
#ifndef X3_ORACLE_H
#define X3_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define X3O_MAX_MATCH 32 /* backend.h:7-10, MAX_MATCH_LEN */

/* event alphabet, x3.c:33-40 */
enum { X3O_E_CTX0 = 0, X3O_E_CTX1 = 1, X3O_E_IDX1 = 2, X3O_E_NEW = 3, X3O_E_EOF = 4, X3O_E_LAST = 5 };

/* tunables held in file-scope globals by the reference: backend.c:8,21,33-34 and x3.c:355 */
typedef struct {
	uint32_t window_bytes;    /* -w N  => N*1024 (x3.c:503); default 8192 (backend.c:8)       */
	int32_t  max_match_count; /* -t N; default 15 (backend.c:21)                               */
	uint32_t factor1;         /* -m N; default 4  (backend.c:33)                               */
	uint32_t factor2;         /* -n N; default 0  (backend.c:34)                               */
	int32_t  nl_mode;         /* -x  ; default 0  (x3.c:355-370)                               */
} x3o_params;

typedef struct {
	uint64_t events[X3O_E_LAST]; /* x3.c:42 (E_EOF slot stays 0 as in the reference)           */
	uint64_t dict_elems;         /* dict_get_elems(), x3.c:693                                  */
	uint64_t ctx0_entries;       /* tag_pair_get_elems(), x3.c:693                              */
	uint64_t steps;              /* parse steps = hits + misses (x3.c:669)                      */
	float    sizes[4];           /* x3.c:43: estimated bits per event class, accumulated in IEEE single IN CODING ORDER:
	                              * sizes[mode] += -log2f(prob) per hit (x3.c:192-193), per coded symbol of a new fragment (x3.c:253-266) */
} x3o_stats;

/* Token trace of the parse (one entry per parse step, x3.c:379-429):
 *   info <  0x80000000 : dictionary hit, info = tag of the element
 *   info >= 0x80000000 : new fragment; bits 0..5 = length (1..32), bit 30 = fragment was already in
 *                        the dictionary (x3.c:412 guard false), so nothing was inserted.            */
#define X3O_TOK_MISS 0x80000000u
#define X3O_TOK_DUP  0x40000000u

void   x3o_default_params(x3o_params *prm);
size_t x3o_compress_bound(size_t n);

/* status codes */
enum { X3O_OK = 0, X3O_E_ARG = -1, X3O_E_NOMEM = -2, X3O_E_FULL = -3, X3O_E_CORRUPT = -4 };

/* Whole-path compress: restates create() + compress() + ac_encode_flush() + bio_close()
 * (x3.c:225-249, 372-434, 603-604).  `in` need not be padded (padding is internal).
 * tok_pos/tok_info (optional, capacity tok_cap) receive the token trace; *ntok the step count. */
int x3o_compress(const x3o_params *prm, const uint8_t *in, size_t n,
                 uint8_t *out, size_t cap, size_t *out_len, x3o_stats *stats);

int x3o_compress_trace(const x3o_params *prm, const uint8_t *in, size_t n,
                       uint8_t *out, size_t cap, size_t *out_len, x3o_stats *stats,
                       uint32_t *tok_pos, uint32_t *tok_info, size_t tok_cap, size_t *ntok);

/* Same stream, but find_best_match is evaluated through the closed form of SURVEY.md 7.1(1)
 * from a precomputed m[] array (x3o_scan_m).  Used to prove the closed form == backend.c:76-97. */
int x3o_compress_via_m(const x3o_params *prm, const uint8_t *in, size_t n, const uint8_t *m,
                       uint8_t *out, size_t cap, size_t *out_len, x3o_stats *stats);

/* Whole-path decompress: decompress() (x3.c:285-353) with bounds checks instead of the
 * reference's unchecked 64x buffer (x3.c:621). */
int x3o_decompress(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len);

/* Stage oracles for kernel-level parity.
 * x3o_count : backend.c:56-74 -- count[i] for one position of the ZERO-PADDED input
 *             (`padded` must hold n + window_bytes readable bytes).
 * x3o_scan_m: m[p] for every p in [0,n): m = max{ i : count[i] > min(T, count[0]-1) }, or 0 when
 *             T <= 0 or count[0] < 2  (closed form of backend.c:76-97 without the dictionary filters;
 *             find_best_match(p) == 1 + max{ i <= m[p] : filters pass }).                       */
void x3o_count(const uint8_t *padded, size_t pos, uint32_t window_bytes, uint32_t count[X3O_MAX_MATCH]);
int  x3o_scan_m(const x3o_params *prm, const uint8_t *in, size_t n, uint8_t *m_out);

/* x3o_ac_chain: the arithmetic coder alone (ac.c:35-85): interval [lo_out[i], hi_out[i]] after symbol i = (cum, freq, total),
 * from ac_init's state; `bits` receives the bit stream so far (no flush), *nbits its length. */
int x3o_ac_chain(const uint32_t *cum, const uint32_t *freq, const uint32_t *total, size_t n, uint32_t *lo_out, uint32_t *hi_out,
                 uint8_t *bits, size_t cap, size_t *nbits);

#ifdef __cplusplus
}
#endif
#endif /* X3_ORACLE_H */
