/*
 * x3_kernels.h -- argument blocks and constants shared by the HIP kernels and their host launcher.
 *
 * Pipeline per stream (chunk), all stages on the GPU:
 *   K1 x3_scan   : forward-window prefix histogram -> one byte m[p] per position     (backend.c:56-74, 76-78)
 *   K2 x3_parse  : dictionary longest-match + best-match selection + parse loop      (dict.c:105-157, backend.c:76-99, x3.c:372-429)
 *   K3 x3_code   : move-to-front rank, context models, mode choice, arithmetic coder, bit output
 *                                                                                    (dict.c:132-146, context.c, tag_pair.c, x3.c:132-270, ac.c:46-126, bio.c:49-112)
 */
#ifndef X3_KERNELS_H
#define X3_KERNELS_H

#include "simt.h"

#define X3_MAXLEN        32u     /* MAX_MATCH_LEN, backend.h:7-10 */
#define X3_PAD_EXTRA     4096u   /* readable zero bytes after n + window: scan over-reads and the parse block stage stay in bounds */

/* event alphabet, x3.c:33-40 */
#define X3_E_CTX0 0
#define X3_E_CTX1 1
#define X3_E_IDX1 2
#define X3_E_NEW  3
#define X3_E_EOF  4

/* token trace produced by K2, consumed by K3 */
#define X3_TOK_MISS 0x80000000u
#define X3_TOK_DUP  0x40000000u

/* device status words */
#define X3_ST_OK        0u
#define X3_ST_OUT_FULL  1u   /* output buffer too small                                  */
#define X3_ST_POOL_FULL 2u   /* a workspace bound was violated (would be a sizing bug)    */

struct X3Chunk {           /* one independent x3 stream */
	uint64_t byte_off;     /* offset of the chunk in the padded byte buffer (and in m[])                 */
	uint32_t len;          /* n: input bytes                                                             */
	uint32_t ht_log2_max;  /* K2: log2 of the hash-table slots reserved for this chunk                   */
	uint64_t elem_off;     /* K2: offset (elements) into dict_pos/dict_len/tok_pos/tok_info              */
	uint64_t ht_off;       /* K2: offset (slots) into ht                                                 */
	uint64_t out_off;      /* byte offset of the chunk's stream in the output buffer                     */
	uint64_t out_cap;      /* bytes reserved (multiple of 4)                                             */
};

struct X3ParseResult { uint32_t ntok, dict_elems, hits, status, miss_bytes, _r0 /* kcycles: step table */, kcyc_fill, kcyc_patch, kcyc_walk, _r1, _r2, _r3; };
struct X3CodeResult  { uint32_t out_len, status, pairs, _r; uint32_t events[8]; };

struct X3CtxHdr { uint32_t off, items, cap, total; };   /* one context: items live in the pool at [off, off+items) */

/* ---- K1 ------------------------------------------------------------------------------------------- */
#define X3_SCAN_TP      32     /* positions per workgroup tile                                    */
#define X3_SCAN_THREADS 256

struct X3ScanArgs {
	const uint8_t *bytes;       /* padded chunks                                                  */
	const X3Chunk *chunks;
	uint8_t *m;                 /* out, same layout as bytes                                      */
	uint32_t *counts;           /* optional (tests): [chunk 0 only][position][32] full histogram  */
	uint32_t window;            /* W bytes (x3.c:503)                                             */
	int32_t  max_match_count;   /* T                                                              */
};

/* ---- K2 ------------------------------------------------------------------------------------------- */
#define X3_PARSE_THREADS 1024 /* the block fill (hash probes of every cached position) is the parallel part */
#define X3_PARSE_PB      2048  /* positions whose dictionary matches are cached in LDS            */
#define X3_HT_LOG2_MIN   10

/* Progress checkpoints of a running parse (pipelined schedule, api.hip): whenever the parse pointer of a chunk crosses one of its
 * marks, the kernel makes every token of that chunk written so far visible device-wide and publishes the chunk's counters to
 * host-mapped memory; the host starts the coding stage of the prefixes while the parse continues.  seq is written last (mark + 1).
 * Slot X3_MAX_CKPT of a chunk is its "done" record (the final counters). */
#define X3_MAX_CKPT 16
#define X3_CKPT_SLOTS (X3_MAX_CKPT + 1)
struct X3ParseCkpt { volatile uint32_t seq, p, ntok, hits, dict_elems, miss_bytes, _r0, _r1; };

struct X3ParseArgs {
	X3ParseCkpt *ckpt;          /* host-mapped, X3_CKPT_SLOTS entries per chunk; nullptr: no checkpoints                */
	const uint32_t *ckpt_pos;   /* device, X3_MAX_CKPT marks per chunk (0xFFFFFFFF: unused)                            */
	uint32_t nckpt;
	const uint8_t *bytes;
	const X3Chunk *chunks;
	const uint8_t *m;
	uint32_t *dict_pos;         /* element e of a chunk is bytes[dict_pos[e] .. +dict_len[e]) ; tag == e */
	uint8_t  *dict_len;
	uint32_t *ht;               /* open addressing, slot = tag+1, 0 = empty                       */
	uint32_t *tok_info;         /* the only per-step output of K2: tag of the hit, or X3_TOK_MISS | dup | length               */
	X3ParseResult *result;
	uint32_t factor1, factor2;
	int32_t  nl_mode;
};

/* ---- K3 ------------------------------------------------------------------------------------------- */
#define X3_CODE_THREADS 64

/* ---- decoder ------------------------------------------------------------------------------------------ */
#define X3_ST_CORRUPT 3u     /* not an x3 stream (the reference abort()s, ac.c:178)                 */

struct X3DecChunk {          /* one stream to decode; workspace sized from the output capacity       */
	uint64_t in_off;
	uint32_t in_len, out_cap;
	uint64_t out_off;
	uint64_t tag_off;        /* per dictionary element: dict_pos, dict_len, mtf, idxfreq, c1off (capacity out_cap + 8)          */
	uint64_t item_off, item_cap; /* the stream's share of the context pool, in 8-byte units                                     */
	uint64_t ht_off;
	uint64_t tok_off;        /* the chain's output: one tag per parse step (capacity out_cap + 8 tokens)                        */
	uint64_t lit_off;        /* bytes of the dictionary elements, in order of creation (capacity out_cap + 64)                  */
	uint32_t ht_log2, _pad;
};

/* The decoder is two stages.  The CHAIN (x3_decode_kernel, one wavefront per stream) decodes events and tags and keeps the models; it writes
 * one tag per parse step (a new fragment is the tag of the element it becomes or repeats) and the bytes of every new element -- it never
 * touches the output.  The second stage turns tags into bytes in parallel: lengths -> positions (prefix sum) -> copy. */
#define X3_DEC_TILE 2048u    /* tokens per workgroup of the second stage */

struct X3DecArgs {
	const uint8_t *in;
	const X3DecChunk *chunks;
	uint8_t *out;
	uint32_t *dict_pos;         /* element = lit[dict_pos .. +dict_len)                             */
	uint8_t  *dict_len;
	uint32_t *ht;
	uint32_t *mtf, *idxfreq, *c1off; /* per element once the dictionary has outgrown the LDS tables: recency list, index model, offset of the element's context1 block */
	uint64_t *pool;             /* context blocks, see decode.hip                                   */
	uint32_t *tokens;
	uint8_t  *lit;
	uint32_t *tile_sum;         /* second stage: output bytes per tile, then their exclusive prefix sum per stream */
	const uint32_t *tile_first; /* second stage: [nchunks + 1] first tile of every stream (from the chain's token counts) */
	uint32_t nchunks, _pad;
	X3CodeResult *result;       /* chain: out_len = tokens, _r = dictionary elements; second stage: out_len = decoded bytes */
};

#endif /* X3_KERNELS_H */
