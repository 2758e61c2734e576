/*
 * x3hip.h -- C ABI of libx3hip.so: the MI355X (gfx950) implementation of the x3 hot path.
 *
 * The reference has no plugin/FFI surface; its hot path is a set of static-linkage C functions driven by
 * main().  This header is the drop-in boundary for that path: plain C, caller-owned buffers,
 * integer status codes instead of abort(), one opaque handle per GPU, thread-safe across handles.  Each entry
 * point names the reference interface it replaces (file:line in the reference tree).
 *
 * The library has NO CPU implementation: every call needs a gfx950 device and returns X3H_E_NO_DEVICE
 * (or the HIP error) otherwise.
 */
#ifndef X3HIP_H
#define X3HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define X3H_ABI_VERSION 6 /* 2: x3h_stats grew (mode_iters, chain_symbols, pipelined); 3: X3C1 container + multi-device batch entries; 4: x3h_decompress_chunks_dev; 5: x3h_compress_container_rccl; 6: x3h_stats.est_bits + x3h_ctx_set_estimates */

/* status codes (the reference abort()s on every error: file.c:9-18, x3.c:515,547,554,583) */
enum {
	X3H_OK            = 0,
	X3H_E_ARG         = -1, /* bad argument (NULL buffer, chunk > X3H_MAX_CHUNK, ...)             */
	X3H_E_NOMEM       = -2, /* host or device allocation failed                                  */
	X3H_E_OUTPUT_FULL = -3, /* caller's output capacity too small (see x3h_compress_bound)        */
	X3H_E_CORRUPT     = -4, /* decoder: not an x3 stream                                          */
	X3H_E_NO_DEVICE   = -5, /* no usable HIP device                                               */
	X3H_E_HIP         = -6, /* a HIP runtime call failed (x3h_last_hip_error() has the code)      */
	X3H_E_INTERNAL    = -7, /* a device-side workspace bound was violated (bug)                   */
	X3H_E_RCCL        = -8  /* librccl could not be loaded, or a send / receive of the gather failed */
};

/* The longest ONE stream: 2^28 - 4096 bytes (256 MiB less a page).  Why not more: every adaptive model counts one per coded symbol on top of at most 2051 initial counts
 * (x3.c:225-249), and a stream of n bytes codes at most n + 1 events, n fragment lengths, n literal bytes, n tags -- so n <= 2^28 - 4096 keeps every model total
 * < 2^28.  After a renormalisation the range is > 2^29 (ac.c:46-75), hence step = range / total >= 2 and the narrowed interval is never a single value: the one
 * assumption the closed-form renormalisation of the coder and decoder chains makes (code2.hip x3_ac2_sym, decode.hip ac_narrow: D = step * freq - 1 != 0).  The
 * reference itself stops working when a total passes the range (> 2^29, SURVEY trap 9: ~2^29 parse steps).  Longer inputs are coded as X3C1 containers. */
#define X3H_MAX_CHUNK (((size_t)1 << 28) - 4096)

/* The tunables the reference keeps in file-scope globals.  Defaults: x3h_default_params().
 *   window_bytes    set_forward_window()   backend.c:8-18   (CLI -w N means N*1024, x3.c:503)
 *   max_match_count set_max_match_count()  backend.c:21-31  (CLI -t)
 *   factor1/factor2 set_magic_factor1/2()  backend.c:33-54  (CLI -m / -n)
 *   nl_mode         g_nl                   x3.c:355-370     (CLI -x)                                   */
typedef struct x3h_params {
	uint32_t window_bytes;
	int32_t  max_match_count;
	uint32_t factor1;
	uint32_t factor2;
	int32_t  nl_mode;
} x3h_params;

/* What the reference prints on stderr after a run (x3.c:662-693), plus device timings. */
typedef struct x3h_stats {
	uint64_t events[5];     /* E_CTX0, E_CTX1, E_IDX1, E_NEW, E_EOF counts (x3.c:42,684)               */
	uint64_t dict_elems;    /* dict_get_elems()      x3.c:693                                          */
	uint64_t ctx0_entries;  /* tag_pair_get_elems()  x3.c:693                                          */
	uint64_t steps;         /* parse steps = hits + misses (x3.c:669)                                   */
	double   ms_total;      /* device time of the whole call (inputs resident -> outputs resident)      */
	double   ms_scan;       /* K1: window scan                                                          */
	double   ms_parse;      /* K2: dictionary + parse                                                   */
	double   ms_code;       /* K3: models + arithmetic coder + bit output                               */
	double   ms_copy;       /* host<->device staging (0 for device-resident calls)                      */
	double   ms_features;   /* K3: parallel feature extraction (sorts, scans, count-smaller-before)     */
	double   ms_modes;      /* K3: x3_modes_kernel (mode choice, x3.c:152-172)                           */
	double   ms_coder;      /* K3: x3_ac2_kernel (arithmetic-coder interval recurrence, ac.c:46-85)      */
	double   ms_emit;       /* K3: symbol assembly + bit emission                                        */
	uint64_t coded_symbols; /* arithmetic-coder symbols (ac_encode calls) of the batch                   */
	int64_t  mode_iters;    /* K3: fixed-point iterations of the mode choice (0: the serial kernel decided;  */
	                        /*     < 0: no fixed point within the cap, the serial kernel ran after all)       */
	uint64_t chain_symbols; /* symbols the coder recurrence processed (coded_symbols minus the no-op ones)   */
	uint64_t pipelined;     /* 1: overlapped stages, the coding stage re-run on growing prefixes of one stream; 2: K3 in slices (carried   */
	                        /*    model state, a few long streams).  ms_parse, ms_features, ms_modes, ms_coder are then per-stage sums that   */
	                        /*    overlap inside ms_total.  0: stage after stage                                                              */
	double   est_bits[4];   /* sizes[E_CTX0..E_NEW] of x3.c:43: the estimated code length per event class, -log2f(prob) summed
	                         * per stream in IEEE single IN CODING ORDER like the reference (x3.c:52-55,192-193,253-266), the streams of a
	                         * batch added up in double.  Only filled after x3h_ctx_set_estimates(ctx, 1); zeros otherwise.
	                         * Tolerance: each term is -log2 evaluated in double and rounded to single, which differs from glibc's log2f by an
	                         * ulp now and then -- the sums equal the reference's up to float rounding (tests: 1e-6 relative), not bit for bit, so
	                         * a printed size can differ by one near a ceil() boundary; for chunked input they are sums the reference never forms. */
	uint64_t coder_launches; /* launches of the coder recurrence behind ms_coder (1 stage after stage; one per prefix / per slice otherwise) */
} x3h_stats;

typedef struct x3h_ctx x3h_ctx; /* one per GPU: device, stream, growable workspace */

int          x3h_abi_version(void);
const char  *x3h_strerror(int status);
int          x3h_last_hip_error(void);
int          x3h_device_count(void);
void         x3h_default_params(x3h_params *prm);           /* backend.c:8,21,33-34 ; x3.c:355            */
size_t       x3h_compress_bound(size_t n);                  /* replaces the unchecked 2*isize of x3.c:580 */

/* create()/destroy() of the reference (x3.c:225-249, 436-458) become handle life time. */
int  x3h_ctx_create(x3h_ctx **ctx, int device);
void x3h_ctx_destroy(x3h_ctx *ctx);
/* A batch of chunks is coded in consecutive sub-batches of at most `input_bytes` (default 512 MiB, >= 1 MiB): the workspace is ~350 bytes of
 * HBM per input byte of a sub-batch.  A short-lived process (the CLI) wants it small: the driver hands out memory that an earlier process
 * freed at ~40 GB/s only, so a 90 GB workspace costs two seconds right behind another process.  Output does not depend on it.        */
int  x3h_ctx_set_batch_bytes(x3h_ctx *ctx, uint64_t input_bytes);
/* on != 0: compress calls on this handle also fill x3h_stats.est_bits (the float size estimates behind the "output stream size",
 * "codestream size", "est. compression ratio" and "event sizes" lines of x3.c:664-691).  Off by default: the sums are one more
 * per-stream chain (a float accumulator per event class, in coding order), which runs beside the coder on its own HIP stream.  */
int  x3h_ctx_set_estimates(x3h_ctx *ctx, int on);

/* Whole path, host buffers.  Replaces   create(); bio_open(); ac_init(); compress(ptr,size,&bio);
 * ac_encode_flush(); bio_close();   (x3.c:562,593-604).  `in` needs no padding (the W zero bytes of
 * x3.c:579,590 are added on the device).  *out_len is a multiple of 4 (bio.c:105-112).               */
int x3h_compress(x3h_ctx *ctx, const x3h_params *prm, const uint8_t *in, size_t n,
                 uint8_t *out, size_t cap, size_t *out_len, x3h_stats *stats);

/* Independent chunks, each coded as its own x3 stream (== `x3 -z` on that chunk alone).
 * Chunk c is in[offsets[c] .. offsets[c+1]); its stream is written at out + c*out_stride (out_stride a
 * multiple of 4, >= the capacity wanted per chunk) and its length to out_lens[c].
 * `in`/`out` are HOST pointers here.                                                                   */
int x3h_compress_chunks(x3h_ctx *ctx, const x3h_params *prm, const uint8_t *in, const uint64_t *offsets,
                        int nchunks, uint8_t *out, uint64_t out_stride, uint64_t *out_lens, x3h_stats *stats);

/* Same, but `d_in` and `d_out` are DEVICE pointers on ctx's GPU (inputs already resident in HBM, outputs left
 * there); offsets/out_lens stay host arrays.  This is the call bench.py times.                          */
int x3h_compress_chunks_dev(x3h_ctx *ctx, const x3h_params *prm, const void *d_in, const uint64_t *offsets,
                            int nchunks, void *d_out, uint64_t out_stride, uint64_t *out_lens, x3h_stats *stats);

/* Decoder.  Replaces   bio_open(READ); ac_init(); ac_decode_init(); decompress(optr,&bio); bio_close();   (x3.c:635-647).
 * The stream carries neither its length nor -w/-t: `cap` bounds the output (the reference assumes 64x the
 * input, unchecked, x3.c:621); X3H_E_OUTPUT_FULL asks for a larger buffer, X3H_E_CORRUPT replaces the abort() of ac.c:178.
 * Workspace: ~200 bytes of HBM per byte of `cap` (the context pool is sized for the worst case), so size `cap` to the data, not to 64x.
 * x3h_stats of a decode: ms_code = the per-stream chains (one tag per parse step), ms_emit = tags -> bytes, ms_total = both + staging. */
int x3h_decompress(x3h_ctx *ctx, const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len, x3h_stats *stats);

/* Independent streams (e.g. the chunks of an X3C1 container): stream c is in[in_offsets[c] .. in_offsets[c+1]) and decodes
 * into out[out_offsets[c] .. out_offsets[c+1]) (that span is its capacity); out_lens[c] receives the decoded size. Host pointers. */
int x3h_decompress_chunks(x3h_ctx *ctx, const uint8_t *in, const uint64_t *in_offsets, int nchunks,
                          uint8_t *out, const uint64_t *out_offsets, uint64_t *out_lens, x3h_stats *stats);

/* Same, but `d_in` and `d_out` are DEVICE pointers on ctx's GPU: the streams are read and the bytes written where they are (no copy).
 * d_in is 4-byte aligned and every in_offsets[c] a multiple of 4 (x3 streams are whole 32-bit words, so streams stored back to back
 * qualify); in_offsets / out_offsets / out_lens stay host arrays. */
int x3h_decompress_chunks_dev(x3h_ctx *ctx, const void *d_in, const uint64_t *in_offsets, int nchunks,
                              void *d_out, const uint64_t *out_offsets, uint64_t *out_lens, x3h_stats *stats);

/* ---- independent chunks across several GPUs + the X3C1 chunk container (SURVEY.md 8(b) batch form, 8(e), 8(f).2) --------------
 * The reference has one stream per file (x3.c:599-611) and no container.  Independent chunks are the only way the path shards
 * (SURVEY.md 8(e)): chunk c of a batch is coded exactly like `x3 -z` on that chunk alone.  `ctxs[0..ndevices)` are handles on the
 * GPUs to use (several handles on ONE GPU are allowed); chunks are dealt out in contiguous blocks, device d gets
 * [d*nchunks/ndevices ...) like BASELINE config 4 (chunk c on GPU c / 16), one host thread per device, no exchange between devices.
 * Host pointers.  *stats (optional) sums the counters; its ms_* fields are those of the slowest device.                        */
int x3h_compress_chunks_multi(x3h_ctx *const *ctxs, int ndevices, const x3h_params *prm, const uint8_t *in, const uint64_t *offsets,
                              int nchunks, uint8_t *out, uint64_t out_stride, uint64_t *out_lens, x3h_stats *stats);
int x3h_decompress_chunks_multi(x3h_ctx *const *ctxs, int ndevices, const uint8_t *in, const uint64_t *in_offsets, int nchunks,
                                uint8_t *out, const uint64_t *out_offsets, uint64_t *out_lens, x3h_stats *stats);

/* X3C1 container (little-endian):  "X3C1" | u32 version=1 | u32 window_bytes | i32 max_match_count | u32 factor1 | u32 factor2 |
 * i32 nl_mode | u32 nchunks | nchunks x { u64 raw_len, u64 comp_len } | the chunk streams back to back.
 * ONE chunk is never wrapped: it stays the raw x3 code stream of x3.c:603-611, so the CLI remains a drop-in.                  */
#define X3H_CONTAINER_VERSION 1
#define X3H_NOT_A_CONTAINER   1 /* x3h_container_probe: no magic -- treat the bytes as one raw x3 stream */
size_t x3h_container_header_bytes(int nchunks);
int    x3h_container_write_header(uint8_t *dst, size_t cap, const x3h_params *prm, int nchunks,
                                  const uint64_t *raw_lens, const uint64_t *comp_lens);
/* X3H_OK: a well-formed container (prm, *nchunks, *raw_total filled; every length checked against n);
 * X3H_NOT_A_CONTAINER: no magic;  X3H_E_CORRUPT: magic but a bad version / table / total length.                               */
int    x3h_container_probe(const uint8_t *blob, size_t n, x3h_params *prm, int *nchunks, uint64_t *raw_total);
/* after an X3H_OK probe: raw_lens[nchunks], and comp_offsets[nchunks+1] = byte offsets of the chunk streams inside blob        */
int    x3h_container_table(const uint8_t *blob, size_t n, uint64_t *raw_lens, uint64_t *comp_offsets);

/* Whole files.  x3h_compress_container: cut `in` into chunks of chunk_bytes (the last one shorter), code them on the given
 * devices and write the container (or, for a single chunk, the raw stream) into out[0..cap).  x3h_container_bound(n, chunk_bytes)
 * is a sufficient `cap`.  x3h_decompress_container accepts either form; for a raw stream `cap` bounds the output as in
 * x3h_decompress, for a container X3H_E_OUTPUT_FULL is returned at once when cap < raw_total.                                  */
size_t x3h_container_bound(size_t n, size_t chunk_bytes);
int x3h_compress_container(x3h_ctx *const *ctxs, int ndevices, const x3h_params *prm, const uint8_t *in, size_t n, size_t chunk_bytes,
                           uint8_t *out, size_t cap, size_t *out_len, x3h_stats *stats);
int x3h_decompress_container(x3h_ctx *const *ctxs, int ndevices, const uint8_t *in, size_t n,
                             uint8_t *out, size_t cap, size_t *out_len, x3h_stats *stats);

/* The same container, with the final concat as ONE RCCL exchange over xGMI (BASELINE north star; the reference has no counterpart: one
 * stream per file, x3.c:599-611): every device keeps its block of the input and its streams in its own HBM, packs the streams back to
 * back, one ncclGroupStart .. ncclSend / ncclRecv .. ncclGroupEnd moves every block to ctxs[0]'s GPU behind the header laid down there,
 * and the finished container crosses PCIe once.  The handles must sit on DISTINCT GPUs (one rank per GPU; X3H_E_ARG otherwise); with one
 * handle the block is sent to itself through RCCL (so the leg can be exercised on a one-GPU machine).  librccl is loaded on first use;
 * X3H_E_RCCL if that fails.  Output is byte-identical to x3h_compress_container's.  x3h_rccl_release() destroys the cached
 * communicators (call it before the handles' GPUs go away; harmless otherwise).                                                        */
int  x3h_compress_container_rccl(x3h_ctx *const *ctxs, int ndevices, const x3h_params *prm, const uint8_t *in, size_t n, size_t chunk_bytes,
                                 uint8_t *out, size_t cap, size_t *out_len, x3h_stats *stats);
void x3h_rccl_release(void);

/* Stage-level entry points (kernel parity tests; the seams named in SURVEY.md 8(b)).
 *  x3h_scan_m      : K1 alone.  m_out[p] = max{ i : count[i] > min(T, count[0]-1) } (0 if T<=0 or count[0]<2) with
 *                    count[] of backend.c:56-74; find_best_match(p) = 1 + max{ i <= m[p] : filters of backend.c:79-90 }.
 *  x3h_scan_counts : K1's raw histogram, counts_out[p*32+i] = count[i] at p (backend.c:62-74).
 *  x3h_parse       : K1+K2.  Token list of the parse loop x3.c:379-429: tok_info < 2^31 is a hit on that tag;
 *                    otherwise bit31 set, bits0..5 = fragment length, bit30 = fragment already in the dictionary.
 *  x3h_coder_chain : the coder recurrence alone (x3_ac2_kernel, ac.c:46-85): n symbols (cum_lo, freq, total) coded from the initial
 *                    interval of ac_init; states_out[2g], states_out[2g+1] = (mLow, range = mHigh - mLow + 1) BEFORE symbol 8g (one state per group
 *                    of 8 symbols, what the kernel stores), *final_lo = mLow after the last symbol.  total in [2, 2^28).               */
int x3h_scan_m(x3h_ctx *ctx, const x3h_params *prm, const uint8_t *in, size_t n, uint8_t *m_out);
int x3h_coder_chain(x3h_ctx *ctx, const uint32_t *cum, const uint32_t *freq, const uint32_t *total, size_t n,
                    uint32_t *states_out, uint32_t *final_lo);
int x3h_scan_counts(x3h_ctx *ctx, const x3h_params *prm, const uint8_t *in, size_t n, uint32_t *counts_out);
int x3h_parse(x3h_ctx *ctx, const x3h_params *prm, const uint8_t *in, size_t n,
              uint32_t *tok_pos, uint32_t *tok_info, size_t tok_cap, size_t *ntok, uint64_t *dict_elems);

#ifdef __cplusplus
}
#endif
#endif /* X3HIP_H */
