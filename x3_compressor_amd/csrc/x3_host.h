/*
 * x3_host.h -- host-side helpers shared by api.hip / code2.hip / prims.hip: growable device buffers, status macros,
 * an element-wise launcher (x3_foreach) and the three library primitives of the v2 coding stage
 * (stable radix sort of (key,value) pairs, exclusive sum scan, inclusive max scan -- hand-written kernels, prims.hip).
 */
#ifndef X3_HOST_H
#define X3_HOST_H

#include "x3_kernels.h"
#include "../../include/x3hip.h"

#ifdef X3_EMU
#include "hip_shim.h"
#endif

#include <stddef.h>
#include <stdint.h>
#include <vector>

extern thread_local int x3_last_hip;
extern thread_local double x3_alloc_ms;   /* time spent in hipMalloc / hipFree by this thread's DevBuf::reserve calls (X3H_DEBUG reports it per call) */
extern thread_local unsigned long long x3_alloc_bytes, x3_alloc_calls;
double x3_now_ms();

#define HIPCHK(expr)                                                   \
	do {                                                               \
		hipError_t e__ = (expr);                                       \
		if (e__ != hipSuccess) { x3_last_hip = (int)e__; return e__ == hipErrorOutOfMemory ? X3H_E_NOMEM : X3H_E_HIP; } \
	} while (0)
#define CHK(expr) do { int r__ = (expr); if (r__ != X3H_OK) return r__; } while (0)

struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	int reserve(size_t bytes)
	{
		if (bytes <= cap) return X3H_OK;
		const double t0 = x3_now_ms();
		if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
		size_t want = bytes + bytes / 8 + 4096;
		hipError_t e = hipMalloc(&p, want);
		x3_alloc_ms += x3_now_ms() - t0; x3_alloc_bytes += want; x3_alloc_calls++;
		if (e != hipSuccess) { p = nullptr; x3_last_hip = (int)e; return X3H_E_NOMEM; }
		cap = want;
		return X3H_OK;
	}
	void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
	template <typename T> T *as() const { return (T *)p; }
};

/* ---- element-wise launcher: f(i) for i in [0,n) ---------------------------------------------------------- */
#ifndef X3_EMU
template <class F> __global__ void __launch_bounds__(256) x3_foreach_kernel(size_t n, F f)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) f(i);
}
template <class F> static inline void x3_foreach(size_t n, hipStream_t st, F f)
{
	if (!n) return;
	hipLaunchKernelGGL(x3_foreach_kernel<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, f);
}
#define X3_LAMBDA [=] __device__
#else
template <class F> static inline void x3_foreach(size_t n, hipStream_t, F f) { for (size_t i = 0; i < n; i++) f(i); }
#define X3_LAMBDA [=]
#endif

/* ---- library primitives (prims.hip) ---------------------------------------------------------------------- */
/* stable LSD radix sort on key bits [0,bits) */
int x3p_sort_pairs(DevBuf &tmp, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, int bits, hipStream_t st);
/* one stable pass on key bits [begin_bit, end_bit) only */
int x3p_sort_pairs_bits(DevBuf &tmp, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit, int end_bit, hipStream_t st);
/* out[i] = sum in[0..i) for i in [0,n]; `in` must have n+1 readable entries (in[n] is ignored), out n+1 writable */
int x3p_excl_scan(DevBuf &tmp, const uint32_t *in, uint32_t *out, size_t n, hipStream_t st);
int x3p_excl_scan_top_bit_w(DevBuf &tmp, const uint4 *rec, uint32_t *out, size_t n, hipStream_t st); /* out[i] = #{ j < i : rec[j].w >> 31 }, i <= n (rec[n] is read, not used) */
int x3p_check_error(hipStream_t st); /* X3H_E_INTERNAL if a chained scan gave up its bounded wait since the last check (cannot happen; checked behind every call) */
/* out[i] = max in[0..i] */
int x3p_incl_max_scan(DevBuf &tmp, const uint32_t *in, uint32_t *out, size_t n, hipStream_t st);

/* ---- K1 v2 (scan2.hip) ----------------------------------------------------------------------------------- */
struct X3Scan2Bufs { DevBuf a[24]; DevBuf misc; };
int x3_scan_v2_run(X3Scan2Bufs &B, DevBuf &tmp, hipStream_t st, int nchunks, const X3Chunk *h_chunks, const X3Chunk *d_chunks,
                   const uint8_t *d_bytes, uint8_t *d_m, uint64_t total, uint32_t window, int32_t T);

/* ---- K1 for many chunks: one workgroup per chunk (scan3.hip) ------------------------------------------------- */
#define X3_SEG_THREADS 1024u
#ifndef X3_SEG_MAXLEN
#define X3_SEG_MAXLEN (256u << 10)   /* longest chunk whose level counters (2 bits per position) and small-K marks (1 bit) fit the workgroup's LDS: 96 KiB */
#endif
#ifndef X3_SEG_MIN_STREAMS
#define X3_SEG_MIN_STREAMS 48u       /* fewer chunks: the chip-wide sort of scan2.hip */
#endif
#define X3_SEG_MAXLEN_BIG (1u << 20)     /* longer chunks (up to this) keep those counters in global memory */
#define X3_SEG_MIN_STREAMS_BIG 224u      /* ... and need nearly a chunk per CU to beat the chip-wide sort (a 1 MiB chunk keeps its CU busy for 27 ms) */
struct X3SegArgs {
	const uint8_t *bytes;        /* padded chunks */
	const X3Chunk *chunks;
	uint2 *la, *lb;              /* ping-pong lists of (key, END position), one entry per position of the padded layout; chunk c's list starts at entry byte_off */
	uint32_t *S4, *K4;           /* out: list 4 (END positions ordered by 4-gram) and its keys, same layout; slot tails hold fillers */
	uint8_t *m;                  /* out: m[p] for levels 0..3 (the walk kernel raises it where the 4-gram repeats often enough) */
	uint32_t *rare, *kexact;     /* out: positions with K < T+1 (bitmap, zeroed by the caller) and their K */
	uint32_t *gmf;               /* nullptr: level counters in LDS; else 2 bits per position of the padded layout, zeroed by the caller (long chunks) */
	uint32_t *act, *act_k, *act_j, *nact; /* out: positions for the walk kernel; nact[0] = their number, nact[1] = a dense class was met, nact[2] = the per-chunk refinement gave up */
	uint32_t *dense_chunk;       /* out, per chunk (zeroed by the caller): a class of this chunk is dense -> x3_segrefine_kernel */
	uint32_t window, ncand, Tu, dense_at;
	uint32_t la_lds, _pad;       /* 1: entry j + T + 1 of a tile's entries is read out of LDS where it lies inside the tile (X3H_SEG_LA_LDS; scan3.hip) */
	uint64_t *prof;              /* nullptr, or 16 cycle counters: phase 0, passes 1-4, levels 1-4 (X3H_SEG_PROF, debugging) */
};
int x3_scan_seg_applies(uint32_t nchunks, uint64_t max_len, uint64_t padded_total = 0); /* 0: no, 1: counters in LDS, 2: counters in global memory */
int x3_scan_seg_launch(const X3SegArgs &a, uint32_t nchunks, hipStream_t st);
int x3_scan_seg_refine_launch(const X3SegArgs &a, uint32_t nchunks, hipStream_t st); /* dense classes of every marked chunk, by the chunk's own workgroup */

/* ---- v2 coding stage (code2.hip) ------------------------------------------------------------------------- */
struct X3Code2Stats { double ms_features, ms_modes, ms_coder, ms_emit; uint64_t symbols, chain_symbols; int mode_iters; /* fixed-point iterations of the mode choice (0: serial kernel, < 0: not converged, serial kernel ran) */ };

struct X3Code2Bufs {
	DevBuf tmp, offs, chunkmeta;
	DevBuf a[48]; /* u32 work arrays of max(hits, events)+4 entries */
	DevBuf idxfreq, hsym, maxred, csbsmall;
	DevBuf stat, stat0;  /* per hit {freq, total, cum, first} in context1 / context0: what a per-stream context kernel (code3.hip) stores in one go */
	hipEvent_t ev[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
	hipStream_t side = nullptr;                       /* batches of many streams: the move-to-front ranks run beside the context statistics */
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	X3Code2Stats last = { 0, 0, 0, 0, 0, 0, 0 };
	DevBuf y[12]; /* u32 arrays over coded symbols */
	DevBuf yraw;  /* symbol operands before the no-op symbols are dropped */
	DevBuf yfin, yfinrec; /* pipelined schedule, final call: operands / chain states gathered into the final symbol layout */
	DevBuf pp[4]; /* token post-pass temporaries */
	DevBuf ms[16]; /* u32 arrays over new-fragment lengths / bytes */
	/* size estimates (x3h_ctx_set_estimates; x3.c:43,192-193,253-266): one float term and one class byte per coded symbol, summed per stream and
	 * class in coding order by x3_est_kernel on its own HIP stream (beside the coder); est_out = 4 floats per stream */
	bool want_est = false, est_pending = false;
	DevBuf est_val, est_cls, est_out;
	hipStream_t est_stream = nullptr;
	hipEvent_t ev_est_fork = nullptr, ev_est_done = nullptr;
};


int x3_token_postpass(X3Code2Bufs &B, hipStream_t st, int nchunks, const X3Chunk *h_chunks, const X3Chunk *d_chunks,
                      const X3ParseResult *d_parsed, const uint32_t *tok_info, const uint8_t *dict_len,
                      uint32_t *tok_pos, uint32_t *tok_hb, uint32_t *tok_nb, uint32_t *tok_mb, size_t prefix_tokens = 0, bool rebase = false);
/* (rebase = false: the four arrays stay batch-wide prefix sums and X3Code2Bufs::pp[0] holds, per stream, the four values at its first
 * token -- x3_code_v2_run subtracts them where it reads; rebase = true rewrites the arrays stream-relative, for the v1 kernel) */

/* Coding GROWING PREFIXES of a few long streams while their parse is still running (api.hip, pipelined schedule): every call
 * recomputes the (parallel) features of the whole prefixes, puts the operands of the NEW chain symbols of every stream into a ring
 * (bump-allocated, never moved) and queues the coder recurrence for them on its own HIP stream, so the recurrence of one segment
 * overlaps the parse and the feature passes of the next.  The final call gathers the chain states of all segments into the final
 * symbol layout and emits the bit streams. */
struct X3CodeSegCall { uint64_t base, total; std::vector<uint32_t> off /* nc+1, ring coordinates */, len, before; };
struct X3CodeSeg {
	bool final = false;               /* the prefixes are the whole streams: E_EOF, flush and bit emission happen in this call      */
	hipStream_t coder_stream = nullptr;
	hipEvent_t ev_ready = nullptr, ev_coder_begin = nullptr, ev_coder_end = nullptr; /* symbols assembled (feature stream); around the recurrence (coder stream) */
	uint32_t *coder_state = nullptr;  /* device: {lo, R} per stream, carried from segment to segment; {0, 0x80000000} before the first */
	hipStream_t emit_stream = nullptr; /* the bits of a segment are written behind its recurrence, beside the recurrence of the next segment (nullptr: all of them after the last) */
	hipEvent_t ev_emit_done = nullptr;
	DevBuf emit_state;                /* per stream {pending bits, bit position lo, hi, -}: carried from segment to segment */
	std::vector<uint32_t> y_done;     /* per stream: chain symbols already handed to the coder (a multiple of 8 until the final call)  */
	uint64_t ring_top = 0;            /* bump pointer of the operand / state rings, in symbols                                     */
	std::vector<X3CodeSegCall> calls; /* what every call put where                                                                 */
	DevBuf meta;                      /* device copy of (off, len, before) of every call                                           */
	DevBuf modes_state, mode_prev;    /* serial mode kernel: saved chain state per stream; the previous call's modes while they are re-laid out */
	std::vector<uint32_t> prev_ho;    /* hit offsets of the previous call (its layout of the per-hit arrays)                      */
	bool prev_serial = false;         /* the previous call ran the serial mode kernel (its saved state is valid)                  */
	size_t res_steps = 0, res_hits = 0, res_elems = 0, res_mbytes = 0, res_bytes = 0; /* sizing estimates for the whole batch: nothing is reallocated while other streams run */
};


/* ---- K3 in slices (code4.hip) ------------------------------------------------------------------------------ */
/* One stream's part of one slice: the parse steps [t0, t1) between two checkpoints of the running parse (X3ParseCkpt), with the running counts at
 * both ends -- all known to the host from the checkpoint records, so every launch of a slice is sized without a device round trip.  The per-hit /
 * per-step work arrays of a slice are DENSE over the slice (stream c's entries start at sh / se / sm / sb / ss): they are temporaries of the slice,
 * what persists from slice to slice is the adaptive state (recency order, context lists, model counters, coder interval). */
struct X3Slice {
	uint32_t t0, t1;           /* steps                                             */
	uint32_t h0, h1;           /* hits before t0 / t1                               */
	uint32_t d0, d1;           /* dictionary elements before t0 / t1                */
	uint32_t mb0, mb1;         /* new-fragment bytes before t0 / t1                 */
	uint32_t p0;               /* input position of step t0                         */
	uint32_t sh, se, sm, sb, ss, sy; /* first slice-local hit / touch event / new fragment / fragment byte / step / (raw) symbol of this stream */
	uint32_t last;             /* 1: the stream ends with this slice                */
	uint32_t ended;            /* 1: the stream has ended, with this slice or an earlier one (its E_EOF symbol exists) */
};

#define X3S_DMAX 2048u          /* largest dictionary of a stream the sliced kernels hold in their LDS tables (api.hip falls back to the other schedules beyond) */
#define X3S_MAX_SLICES 24u
#define X3S_MTF_RANGES 128u      /* most time ranges per stream and slice of the move-to-front kernel */
#define X3S_NARR 32
/* per-stream carried scalars, one array of nc (x multiplicity) words each inside X3SliceRun::small, in this order */
enum { X3S_EVFINAL = 0 /* x4: model_events freqs of E_CTX0 / E_CTX1 / E_IDX1, IDX1 uses */, X3S_NNOOP = 4, X3S_NPAIRS = 5, X3S_ORD00 = 6, X3S_LASTORD = 7, X3S_YCNT = 8, X3S_YDONE = 9,
       X3S_TOP1 = 10, X3S_TOP0 = 11, X3S_STATUS = 12, X3S_FIRST00 = 13, X3S_NIDX0 = 14, X3S_SEGOFF = 15, X3S_SEGLEN = 16, X3S_NTOK = 17, X3S_NHITS = 18, X3S_ESTFIRST = 19,
       X3S_ESTCNT = 20, X3S_CODER = 21 /* x2: {lo, R} */, X3S_EMITCARRY = 23 /* x4 */, X3S_FINALLO = 27, X3S_EST = 28 /* x4 floats */, X3S_O0HIST = 32 /* x288: model_match_size + model_chars counters */,
       X3S_SMALL_WORDS = 32 + 288 };
struct X3SliceRun {
	uint32_t nc = 0;
	uint64_t elems = 0;
	/* carried state (per stream at elem_off / 4 * elem_off / 3 * elem_off) */
	DevBuf lt, idxfreq, idxhist, hdr1, hdr0, pool1, pord1, pool0, sym, states, small, mtf_scratch, ctx_scratch, ctx_pending, ctx_dbg, lt2;
	bool lt_flip = false;
	/* temporaries of a slice */
	DevBuf a[2][X3S_NARR], b[2][3], stat1[2], stat0[2], est_val, est_cls, tmp, tables; /* two sets: stage A of a slice beside stage B of the slice before */
	std::vector<std::vector<X3Slice>> slices; /* host copies of the slice tables, kept until the run ends (their H2D copies are asynchronous) */
	void release()
	{
		DevBuf *all[] = { &lt, &idxfreq, &idxhist, &hdr1, &hdr0, &pool1, &pord1, &pool0, &sym, &states, &small, &mtf_scratch, &ctx_scratch, &ctx_pending, &ctx_dbg, &lt2, &est_val, &est_cls, &tmp, &tables };
		for (DevBuf *d : all) d->release();
		for (int q = 0; q < 2; q++) { for (DevBuf &d : a[q]) d.release(); for (DevBuf &d : b[q]) d.release(); stat1[q].release(); stat0[q].release(); }
	}
};
int x3s_begin(X3SliceRun &R, hipStream_t st, uint32_t nc, const X3Chunk *h_chunks, uint64_t max_slice_steps, uint64_t max_slice_bytes);
int x3s_slice(X3SliceRun &R, hipStream_t st, hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join, const X3Chunk *d_chunks, const std::vector<X3Slice> &hs,
              uint64_t max_dict, const uint8_t *d_bytes, const uint32_t *tok_info, const uint8_t *dict_len, bool last, bool want_est, uint32_t **seg_off_out, uint32_t **seg_len_out, int phase = 3);
int x3_zero_output_slots(hipStream_t st, uint32_t nc, const X3Chunk *h_chunks, const X3Chunk *d_chunks, uint8_t *d_out);

/* argument blocks of kernels that code2.hip defines and code4.hip launches too */
#ifndef X3_IDXF_LDS
#define X3_IDXF_LDS 32768u /* ranks whose model_index1 frequency lives in LDS (128 KiB); beyond that: global memory */
#endif
#ifndef X3_IDXF_LDS_SMALL
#define X3_IDXF_LDS_SMALL 2048u /* ... in batches of many streams: 8 KiB per stream, so the whole batch is resident instead of one stream per CU */
#endif
struct X3ModesArgs {
	const X3ParseResult *parsed;
	const uint32_t *ho, *dof;          /* per chunk: first hit, first tag */
	const uint32_t *f0, *t0, *f1, *t1; /* per hit: freq/total in ctx0 and ctx1 (freq 0 == tag absent) */
	uint32_t fs;                       /* stride of those four in words: 1 = plain arrays, 4 = fields of the per-hit records the context kernel stores */
	const uint32_t *rank, *dk, *step;  /* per hit: MTF rank, dictionary size at that step, step index */
	uint32_t *idxfreq;                 /* per tag slot (by rank), pre-set to 1: spill area for ranks >= X3_IDXF_LDS */
	uint32_t *mode;                    /* out per hit: the chosen event (E_CTX0 / E_CTX1 / E_IDX1) */
	/* optional (pe0 != nullptr; batches of many streams): the model state every hit is coded under, straight from the chain's counters --
	 * otherwise x3_code_v2_run recovers it from the modes with scans, sorts and a count-smaller-before over the IDX1 hits */
	uint32_t *pe0, *pe1;               /* out per hit: model_events freq of E_CTX0 / E_CTX1 before the hit (x3.c:176-177)            */
	uint32_t *ilist_rank, *ilist_hit;  /* out: the IDX1-coded hits of stream c in time order, at [ho[c], ho[c] + evfinal[4c+3])          */
	uint32_t *evfinal;                 /* out per chunk: the three model_events freqs after the last hit, and the number of IDX1 hits */
	uint32_t *nzl, *nnoop;             /* optional (with pe0): per hit, the stream's earlier hits whose tag / index symbol is a no-op for the coder (model total 1:
	                                    * x3_make_symbol); per chunk, how many there are -- the compacted symbol index of every step follows without a scan */
	/* optional (state != nullptr; growing prefixes of a few long streams): the chain's state after the last hit is saved per stream
	 * {E0, E1, E2, nidx, hits done, table entries, 0, 0, table...} (stride X3_MODES_STATE_STRIDE words), and a later call on a longer prefix
	 * resumes behind the hits already decided instead of starting over (their modes are in `mode` already) */
	uint32_t *state;
	uint32_t resume;                   /* 1: continue from the saved state where it is valid (hits done <= this prefix's hits) */
	/* optional (slice != nullptr; K3 in slices, code4.hip): the hits are those of ONE SLICE of every stream (slice-local arrays, stream c's start at
	 * slice[c].sh), the chain continues from evfinal[4c ..] / nnoop[c] / the per-rank table idxfreq[chunks[c].elem_off + .] of the earlier slices and
	 * leaves its state there; parsed / ho / dof are not read */
	const struct X3Slice *slice;
	const X3Chunk *chunks;
};
#define X3_MODES_STATE_STRIDE (X3_IDXF_LDS + 8u)
struct X3EmitArgs {
	const uint32_t *yoc;        /* nc+1: symbol ranges (no-op symbols already dropped) */
	const uint4 *sym;           /* per symbol: {cum, freq, magic, shift}               */
	const uint32_t *state;      /* per symbol slot: {lo, R} at the first symbol of every group of X3_AC2_G */
	const uint32_t *final_lo;   /* per stream: lo after the last symbol                */
	const X3Chunk *chunks;      /* out_off / out_cap                                   */
	const X3ParseResult *parsed;
	const uint32_t *npairs, *evfinal;
	uint8_t *out;               /* streams are assembled with ORs into pre-zeroed slots */
	X3CodeResult *result;
	/* segment form (pipelined schedule of a few long streams): the symbols of stream c are [seg_off[c], seg_off[c] + seg_len[c]) of the operand /
	 * state rings, the pending-bit count and the bit position come from and go back to carry[4c ..], flush + result only when last */
	const uint32_t *seg_off, *seg_len;
	uint32_t *carry;
	uint32_t last;
	uint32_t compact;           /* the states lie in compact slots (X3Ac2Args::compact) */
	const uint32_t *ntok, *nhits; /* sliced schedule: steps / hits per stream for the result record (nullptr: parsed[c]) */
};
struct X3EstArgs { const uint32_t *range; const float *val; const uint8_t *cls; float *out;
	const uint32_t *seg_first, *seg_count; /* sliced schedule (nullptr otherwise): stream c's terms are [seg_first[c], + seg_count[c]) and out[4c ..] carries the four sums from slice to slice */
};

int x3s_modes_launch(const X3ModesArgs &a, uint32_t nc, uint64_t max_dict, hipStream_t st);   /* the mode chain of one slice (a.slice != nullptr) */
int x3s_ac2_launch(const uint4 *sym, uint32_t *states, uint32_t *final_lo, const uint32_t *seg_off, const uint32_t *seg_len, uint32_t *seg_state, uint32_t nc, hipStream_t st);
int x3s_emit_launch(const X3EmitArgs &a, uint32_t nc, hipStream_t st);
int x3s_est_launch(const X3EstArgs &a, uint32_t nc, hipStream_t st);

int x3_code_v2_run(X3Code2Bufs &B, hipStream_t st, int nchunks, const X3Chunk *h_chunks, const X3Chunk *d_chunks,
                   const X3ParseResult *h_parsed, const X3ParseResult *d_parsed,
                   const uint8_t *d_bytes, const uint32_t *tok_pos, const uint32_t *tok_info, const uint32_t *tok_hb,
                   const uint32_t *tok_nb, const uint32_t *tok_mb, uint8_t *d_out, X3CodeResult *d_result, X3CodeSeg *seg = nullptr,
                   const uint8_t *dict_len = nullptr);
/* dict_len != nullptr: x3_token_postpass has NOT run (api.hip leaves it out for batches of many streams): x3_code_v2_run derives the token
 * prefix sums itself -- one workgroup per stream (code3.hip) when the per-stream kernels apply, else by calling x3_token_postpass */

/* ---- per-stream feature kernels for batches of many streams (code3.hip) -------------------------------------- */
#define X3_STREAM_DMAX 8192u        /* largest per-stream dictionary the LDS tables of those kernels hold */
#define X3_STREAM_MIN_STREAMS 48u   /* batches with fewer streams keep the chip-wide passes (one wavefront per stream would leave the chip idle) */
bool x3_stream_kernels_fit(uint64_t max_dict);
int x3_mtf_ranks_run(hipStream_t st, uint32_t nc, uint64_t max_dict, const uint32_t *d_eo, const uint32_t *d_dof, const uint32_t *e_tag,
                     const uint32_t *e_hit, uint32_t *h_rank);
int x3_arrange_run(hipStream_t st, uint32_t nc, const uint32_t *d_ho, const uint32_t *kbase, uint64_t max_local, const uint32_t *key, const uint32_t *h_tag,
                   uint32_t *kA, uint32_t *vA, uint32_t *tA, uint32_t *tmpk, uint32_t *tmpv);
struct X3SegSortGen { uint32_t *out; const uint32_t *h_pv, *P, *first00, *ord00; const uint4 *stat; }; /* the keys (context0 groups) made by the sort itself: X3SegSortArgs::gen */
int x3_segsort_run(hipStream_t st, uint32_t nc, const uint32_t *d_ho, const uint32_t *kbase, uint64_t max_local, const uint32_t *key,
                   uint32_t *kA, uint32_t *vA, uint32_t *tmpk, uint32_t *tmpv, const X3SegSortGen *gen = nullptr); /* one workgroup per stream, <= 8-bit passes (code3.hip) */
#define X3_SEGSORT_MAX_LOCAL (((uint64_t)1 << 24) - 1)
#define X3_SEGSORT_MIN_STREAMS 128u  /* fewer streams: the chip-wide sort (a stream's passes are a chain on one CU) */
#ifndef X3_ARR_DBITS
#define X3_ARR_DBITS 11
#endif
#define X3_ARRANGE_MAX_LOCAL (((uint64_t)1 << (2 * X3_ARR_DBITS)) - 1) /* largest stream-local key two passes cover */
int x3_ctx_stats_run(hipStream_t st, uint32_t nc, uint64_t max_dict, uint64_t nhits, const uint32_t *d_ho, const uint32_t *d_dof, const uint32_t *kA,
                     const uint32_t *vA, const uint32_t *tA, const uint32_t *h_tag, uint4 *stat, uint32_t *first00 /* per hit: {freq, total, cum, first hit | isfirst << 31} */);
int x3_order0_run(hipStream_t st, uint32_t nc, const uint32_t *d_mo, const uint32_t *lval, uint32_t *lsm, uint32_t *leq,
                  const uint32_t *d_bo, const uint32_t *bval, uint32_t *bsm, uint32_t *beq);
int x3_idxstat_run(hipStream_t st, uint32_t nc, uint64_t max_dict, const uint32_t *d_ho, const uint32_t *evfinal, const uint32_t *lrank,
                   const uint32_t *lhit, const uint32_t *h_dk, uint32_t *rfreq, uint32_t *rcum, uint32_t *itot);
int x3_tokens_run(hipStream_t st, uint32_t nc, const X3Chunk *d_chunks, const X3ParseResult *d_parsed, const uint32_t *tok_info, const uint8_t *dict_len,
                  uint32_t *tok_pos, uint32_t *tok_hb, uint32_t *tok_nb, uint32_t *tok_mb, const uint32_t *d_ho, const uint32_t *d_eo, const uint32_t *d_dof,
                  uint32_t *h_tag, uint32_t *h_c1, uint32_t *h_pv, uint32_t *h_dk, uint32_t *h_step, uint32_t *e_tag, uint32_t *e_hit,
                  const uint8_t *d_bytes, const uint32_t *d_mo, const uint32_t *d_bo, uint32_t *lval, uint32_t *bval);
/* stage seam for tests: x3_ac2_kernel on a caller-given symbol sequence (code2.hip) */
int x3_coder_chain_run(X3Code2Bufs &B, hipStream_t st, const uint32_t *h_cum, const uint32_t *h_freq, const uint32_t *h_total, size_t n,
                       uint32_t *h_states, uint32_t *h_final_lo);

#endif /* X3_HOST_H */
