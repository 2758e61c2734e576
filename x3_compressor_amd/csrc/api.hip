/*
 * api.hip -- host side of libx3hip.so: the C ABI of include/x3hip.h on top of the three stages (K1 scan2.hip, K2 parse.hip, K3 code2.hip)
 * and the decoder, plus the two schedules of a compress call (run_one: stage after stage; run_pipelined: overlapped).
 *
 * Data layout in HBM for a batch of independent chunks (streams):
 *   pad    : every chunk copied to a 256-byte aligned slot of  n + W + X3_PAD_EXTRA  bytes, tail zeroed -- the
 *            W zero bytes of x3.c:579,590 that take part in the scan and in dictionary compares (never the next chunk)
 *   m      : one byte per position, same slots (K1 -> K2)
 *   dict_pos/dict_len/tok_pos/tok_info : n+16 entries per chunk (worst case: every byte its own fragment)
 *   ht     : 2^ceil(log2(2(n+1))) slots per chunk, K2 uses the first 2^hlog of them
 *   mtf/idxfreq/ctx1 (per tag), ctx0 (per pair), item pool, pair map : sized AFTER K2 from the exact number of
 *            elements D and hits H of each chunk: pool 8H+64 items (doubling arrays waste < 4x of <= 2H items),
 *            pair map 2^ceil(log2(2(H+2))) slots -- provable bounds, no device-side allocation failure path.
 */
#include "x3_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <new>

extern "C" void x3k_launch_scan(const X3ScanArgs *a, uint32_t max_len, uint32_t nchunks, hipStream_t st);
extern "C" void x3k_launch_parse(const X3ParseArgs *a, uint32_t nchunks, hipStream_t st);
struct x3h_ctx;
static bool sliced_masked_setup(x3h_ctx *c);
extern "C" void x3k_launch_decode(const X3DecArgs *a, uint32_t nchunks, hipStream_t st);
extern "C" void x3k_launch_decode_bytes(const X3DecArgs *a, uint32_t nchunks, uint32_t ntiles, hipStream_t st);

thread_local int x3_last_hip = 0;
thread_local double x3_alloc_ms = 0;
thread_local unsigned long long x3_alloc_bytes = 0, x3_alloc_calls = 0;
#include <time.h>
double x3_now_ms() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
#define g_last_hip x3_last_hip

struct x3h_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	hipEvent_t ev[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
	DevBuf pad, m, dict_pos, dict_len, ht, tok_pos, tok_info, tok_hb, tok_nb, tok_mb, chunks, presult, cresult;
	X3Code2Bufs c2;
	X3Scan2Bufs s2;
	int scan_v1 = 0;
	/* single-stream pipelining (run_pipelined): the parse on its own stream publishes checkpoints, the coding stage of every prefix
	 * runs while the parse continues, the coder recurrence of a segment on a third stream */
	int pipe_max_streams = 32; /* X3H_PIPE_STREAMS */
	double pipe_marks[X3_MAX_CKPT] = { 0.02, 0.08, 0.26, 0.62 }; /* where the parse publishes checkpoints (X3H_PIPE_MARKS: up to 8 ascending fractions) */
	uint32_t pipe_nmarks = 4;
	uint64_t pipe_min = (uint64_t)256 << 10; /* streams at least this long are pipelined (X3H_PIPE_MIN; 0: never): measured faster from 256 KiB up */
	hipStream_t s_parse = nullptr, s_coder = nullptr, s_emit = nullptr;
	hipEvent_t ev_emit = nullptr;
	int seg_emit = -1; /* pipelined schedule: 1 = each coder segment's bits are written behind it (one workgroup per stream), 0 = every bit after the last
	                    * segment by chip-wide passes, -1 = by the batch: 8 or more streams of at most 16 MiB (config 4's shape) measured 8 ms faster the
	                    * first way, one or a few longer streams 6-60 ms faster the second (X3H_SEG_EMIT overrides) */
	hipEvent_t ev_p0 = nullptr, ev_p1 = nullptr, ev_ready = nullptr, ev_cb[X3_MAX_CKPT + 1] = {}, ev_ce[X3_MAX_CKPT + 1] = {};
	/* K3 in slices (code4.hip; run_sliced): the default overlapped schedule wherever every stream's dictionary fits the per-stream LDS tables */
	int sliced = 1;                               /* X3H_SLICED=0: never */
	uint64_t sliced_min = (uint64_t)96 << 10;     /* X3H_SLICED_MIN: longest stream of the batch at least this long */
	int sliced_max_streams = 192;                 /* X3H_SLICED_STREAMS: beyond, every CU has a stream of every stage anyway (stage after stage) */
	/* X3H_SLICE_MARKS: small slices first (the coder starts after 2 % of the input), then each about 1.7 times the one before: the feature stages of a slice take
	 * 0.35-0.6 of its coder time and the parse has to get there first, so the coder never runs dry -- and FEW slices: every slice is one coder launch for all
	 * streams, which ends with its longest segment (eleven equal slices cost config 3, whose streams differ in symbols per byte, 19 % of coder time).
	 * Round 5: nine marks instead of six (0.02 .. 0.60) -- a smaller first slice (the coder starts after 1 %) and, above all, a smaller LAST one: behind the last coder
	 * segment only its own bits remain to be written, and with a 40 % tail that was 10 ms of config 4's share (441.6 -> 419.3 ms; config 3 2 006 -> 1 957 ms;
	 * profiles/r05_slice_marks.txt) */
	double slice_marks[X3_MAX_CKPT] = { 0.01, 0.03, 0.07, 0.14, 0.25, 0.40, 0.58, 0.76, 0.90 };
	uint32_t slice_nmarks = 9;
	bool slice_marks_fixed = false;
	X3SliceRun sr;
	hipEvent_t ev_sf[X3S_MAX_SLICES + 2] = {}, ev_sb[X3S_MAX_SLICES + 2] = {}, ev_se[X3S_MAX_SLICES + 2] = {}, ev_sc[X3S_MAX_SLICES + 2] = {}, ev_sa[X3S_MAX_SLICES + 2] = {}, ev_s0 = nullptr;
	/* Mid-size batches (every stream short enough that the parse is a few milliseconds): the coder's lone wavefronts get CUs of their own (the top 32 of the CU mask),
	 * parse and feature kernels the rest.  A coder wavefront slows the feature workgroups that share its CU (and the neighbour CU: "odd / even" masks do not help) by a
	 * factor of two to six -- measured with the coder serialised, with a memory-free kernel in its place and with disjoint masks (profiles/r04_midsize_marks_and_timeline.txt);
	 * what exactly they compete for is not known (neither scalar loads nor scalar stores nor a saturated scalar unit alone reproduce it).  Three more HIP streams,
	 * created on first use; X3H_SLICE_CUMASK=0: never. */
	int slice_cumask = 1;
	uint64_t slice_cumask_max_len = (uint64_t)2 << 20; /* X3H_SLICE_CUMASK_LEN: longest stream of the batch at most this long (long parses stay on the three plain streams: see sliced_setup) */
	uint64_t slice_cumask_min_len = 0; /* X3H_SLICE_CUMASK_MIN: ... and at least this long (tuning) */
	uint32_t slice_cumask_min_streams = 4; /* X3H_SLICE_CUMASK_STREAMS: batches of at least this many streams */
	hipStream_t sm_feat = nullptr, sm_parse = nullptr, sm_coder = nullptr;
	int sm_state = 0; /* 0: not tried, 1: ready, -1: not available on this device */
	int slice_bstream = 0;                        /* X3H_SLICE_BSTREAM=1: stage B of a slice (mode chain, models, assembly) on the parse stream once the parse is done, beside stage A of the
	                                               * next slice.  Measured (round 4): no gain -- stage A, not B, is what a small slice costs (~1.3 ms of launches and latency-bound
	                                               * per-stream kernels whatever its size), and config 4's share got 3 % slower (437 against 423 ms): off by default */
	hipEvent_t ev_sfork = nullptr, ev_sjoin = nullptr;
	X3ParseCkpt *ckpt = nullptr; /* host-mapped, X3_CKPT_SLOTS per stream */
	uint32_t ckpt_cap = 0;
	X3CodeSeg seg;
	DevBuf coder_state, prefix_result, srcoff, ckpt_pos;
	uint64_t batch_bytes = (uint64_t)512 << 20; /* X3H_BATCH_BYTES: input bytes of one sub-batch (the workspace is ~350 B per input byte) */
	uint64_t dec_batch_bytes = (uint64_t)512 << 20; /* X3H_DEC_BATCH_BYTES */
	bool dec_batch_from_env = false;
	uint64_t batch_pad_bytes = (uint64_t)2 << 30; /* X3H_BATCH_PAD_BYTES: padded layout of one sub-batch (K1 needs < 2^32 - 256) */
	uint64_t pad_total = 0;
	DevBuf mtf, idxfreq, items, pkey, pval, out, counts;
	DevBuf din, dchunks, dc1, dtok, dlit, dtile, dtfirst; /* decoder: input streams, stream table, context1 block per element, token trace, element bytes, second-stage tiles */
	std::vector<uint32_t> htfirst;
	DevBuf g_in, g_out, g_pack, g_final; /* x3h_compress_container_rccl: this device's input block, its strided streams, the packed block, the finished container (root) */
	std::vector<X3Chunk> hchunks;
	std::vector<X3ParseResult> hparse;
	std::vector<X3CodeResult> hcode;
};

static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }
static inline uint32_t ceil_log2(uint64_t v) { uint32_t l = 0; while (((uint64_t)1 << l) < v) l++; return l; }

extern "C" int x3h_abi_version(void) { return X3H_ABI_VERSION; }
extern "C" int x3h_last_hip_error(void) { return g_last_hip; }

extern "C" const char *x3h_strerror(int s)
{
	switch (s) {
		case X3H_OK: return "ok";
		case X3H_E_ARG: return "bad argument";
		case X3H_E_NOMEM: return "out of memory";
		case X3H_E_OUTPUT_FULL: return "output buffer too small";
		case X3H_E_CORRUPT: return "corrupt stream";
		case X3H_E_NO_DEVICE: return "no HIP device";
		case X3H_E_HIP: return "HIP runtime error";
		case X3H_E_INTERNAL: return "internal workspace bound violated";
		case X3H_E_RCCL: return "RCCL unavailable or a send/receive failed";
		default: return "unknown status";
	}
}

extern "C" int x3h_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" void x3h_default_params(x3h_params *p)
{
	p->window_bytes = 8 * 1024;
	p->max_match_count = 15;
	p->factor1 = 4;
	p->factor2 = 0;
	p->nl_mode = 0;
}

extern "C" size_t x3h_compress_bound(size_t n)
{
	/* a 1-byte fragment costs three coder symbols of < 31 bits each; + EOF, flush, word padding */
	return 12 * n + 64;
}

extern "C" int x3h_ctx_create(x3h_ctx **out, int device)
{
	if (!out) return X3H_E_ARG;
	*out = nullptr;
	int n = 0;
	const double t_c0 = x3_now_ms();
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return X3H_E_NO_DEVICE;
	if (device < 0 || device >= n) return X3H_E_ARG;
	const double t_c1 = x3_now_ms();
	HIPCHK(hipSetDevice(device));
	x3h_ctx *c = new (std::nothrow) x3h_ctx();
	if (!c) return X3H_E_NOMEM;
	c->device = device;
	{ const char *e = getenv("X3H_SCAN_V1"); c->scan_v1 = e && *e && *e != '0'; }
	{ const char *e = getenv("X3H_BATCH_BYTES"); if (e && atoll(e) > 0) c->batch_bytes = (uint64_t)atoll(e); }
	{ const char *e = getenv("X3H_DEC_BATCH_BYTES"); if (e && atoll(e) > 0) { c->dec_batch_bytes = (uint64_t)atoll(e); c->dec_batch_from_env = true; } }
	{ const char *e = getenv("X3H_BATCH_PAD_BYTES"); if (e && atoll(e) > 0) c->batch_pad_bytes = (uint64_t)atoll(e); }
	if (c->batch_pad_bytes > 0xF0000000ull) c->batch_pad_bytes = 0xF0000000ull;
	{ const char *e = getenv("X3H_PIPE_MIN"); if (e && *e) c->pipe_min = (uint64_t)atoll(e); }
	{ const char *e = getenv("X3H_PIPE_MARKS");
	  if (e && *e) { /* e.g. "0.02,0.08,0.26,0.62,0.85": anything malformed keeps the default */
		double m[X3_MAX_CKPT]; uint32_t k = 0; const char *q = e; bool ok = true;
		while (*q && k < X3_MAX_CKPT) { char *end = nullptr; const double v = strtod(q, &end); if (end == q || v <= (k ? m[k - 1] : 0.0) || v >= 1.0) { ok = false; break; } m[k++] = v; q = end; if (*q == ',') q++; }
		if (ok && k && !*q) { for (uint32_t i = 0; i < k; i++) c->pipe_marks[i] = m[i]; c->pipe_nmarks = k; }
	  } }
	{ const char *e = getenv("X3H_PIPE_STREAMS"); if (e && *e) c->pipe_max_streams = atoi(e); }
	{ const char *e = getenv("X3H_SLICED"); if (e && *e) c->sliced = *e != '0'; }
	{ const char *e = getenv("X3H_SLICED_MIN"); if (e && *e) c->sliced_min = (uint64_t)atoll(e); }
	{ const char *e = getenv("X3H_SLICE_BSTREAM"); if (e && *e) c->slice_bstream = *e != '0'; }
	{ const char *e = getenv("X3H_SLICE_CUMASK"); if (e && *e) c->slice_cumask = *e != '0'; }
	{ const char *e = getenv("X3H_SLICE_CUMASK_LEN"); if (e && *e) c->slice_cumask_max_len = (uint64_t)atoll(e); }
	{ const char *e = getenv("X3H_SLICE_CUMASK_MIN"); if (e && *e) c->slice_cumask_min_len = (uint64_t)atoll(e); }
	{ const char *e = getenv("X3H_SLICE_CUMASK_STREAMS"); if (e && atoi(e) >= 1) c->slice_cumask_min_streams = (uint32_t)atoi(e); }
	{ const char *e = getenv("X3H_SLICED_STREAMS"); if (e && *e) c->sliced_max_streams = atoi(e); }
	{ const char *e = getenv("X3H_SLICE_MARKS");
	  if (e && *e) {
		double m[X3_MAX_CKPT]; uint32_t k = 0; const char *q = e; bool ok = true;
		while (*q && k < X3_MAX_CKPT) { char *end = nullptr; const double v = strtod(q, &end); if (end == q || v <= (k ? m[k - 1] : 0.0) || v >= 1.0) { ok = false; break; } m[k++] = v; q = end; if (*q == ',') q++; }
		if (ok && k && !*q) { for (uint32_t i = 0; i < k; i++) c->slice_marks[i] = m[i]; c->slice_nmarks = k; c->slice_marks_fixed = true; }
	  } }
	{ const char *e = getenv("X3H_SEG_EMIT"); if (e && *e) c->seg_emit = *e != '0' ? 1 : 0; }
	const double t_c2 = x3_now_ms();
	if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return X3H_E_HIP; }
	/* the three CU-masked streams of the mid-size layout are made NOW, right behind the handle's own stream: made later (on first use, behind the plain parse and coder streams
	 * and whatever else the process has created meanwhile) they separated coder and feature kernels only in some processes -- bench.py, which makes its handle before torch
	 * touches the device, lost what tools/chunked_dickens.py gained (profiles/r04_midsize_marks_and_timeline.txt: creation orders); made here they work in both */
	const double t_c2b = x3_now_ms();
	if (c->slice_cumask) (void)sliced_masked_setup(c);
	const double t_c3 = x3_now_ms();
	for (int i = 0; i < 6; i++)
		if (hipEventCreate(&c->ev[i]) != hipSuccess) { x3h_ctx_destroy(c); return X3H_E_HIP; }
	if (getenv("X3H_DEBUG")) fprintf(stderr, "[x3h] handle: device count (runtime start) %.1f ms, set device %.1f, stream %.1f, three masked streams %.1f, events %.1f\n", t_c1 - t_c0, t_c2 - t_c1, t_c2b - t_c2, t_c3 - t_c2b, x3_now_ms() - t_c3);
	*out = c;
	return X3H_OK;
}

extern "C" int x3h_ctx_set_batch_bytes(x3h_ctx *c, uint64_t input_bytes)
{
	if (!c || input_bytes < ((uint64_t)1 << 20)) return X3H_E_ARG;
	if (input_bytes > ((uint64_t)1 << 40)) input_bytes = (uint64_t)1 << 40; /* more than any GPU holds: "no sub-batches" */
	c->batch_bytes = input_bytes;
	/* the decoder cuts on output capacity, at ~200 B of workspace per byte (the context pool is sized for the worst case: 192 B); its rate is streams in flight x the per-stream rate.  An explicit
	 * X3H_DEC_BATCH_BYTES from the environment stays in force. */
	if (!c->dec_batch_from_env) c->dec_batch_bytes = 2 * input_bytes;
	return X3H_OK;
}

extern "C" int x3h_ctx_set_estimates(x3h_ctx *c, int on)
{
	if (!c) return X3H_E_ARG;
	c->c2.want_est = on != 0;
	return X3H_OK;
}

extern "C" void x3h_ctx_destroy(x3h_ctx *c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	DevBuf *bufs[] = { &c->pad, &c->m, &c->dict_pos, &c->dict_len, &c->ht, &c->tok_pos, &c->tok_info, &c->tok_hb, &c->tok_nb, &c->tok_mb, &c->chunks, &c->presult,
		               &c->c2.tmp, &c->c2.offs, &c->c2.chunkmeta, &c->c2.idxfreq, &c->c2.hsym, &c->c2.maxred, &c->c2.pp[0], &c->c2.pp[1], &c->c2.pp[2], &c->c2.pp[3],
		               &c->cresult, &c->mtf, &c->idxfreq, &c->items, &c->pkey, &c->pval, &c->out, &c->counts, &c->din, &c->dchunks, &c->dc1, &c->dtok, &c->dlit, &c->dtile, &c->dtfirst,
		               &c->g_in, &c->g_out, &c->g_pack, &c->g_final };
	for (DevBuf *b : bufs) b->release();
	c->coder_state.release(); c->prefix_result.release(); c->srcoff.release(); c->ckpt_pos.release(); c->seg.meta.release(); c->seg.modes_state.release(); c->seg.mode_prev.release();
	c->c2.yfin.release(); c->c2.yfinrec.release();
	c->sr.release();
	for (uint32_t i = 0; i < X3S_MAX_SLICES + 2; i++) { if (c->ev_sf[i]) (void)hipEventDestroy(c->ev_sf[i]); if (c->ev_sb[i]) (void)hipEventDestroy(c->ev_sb[i]); if (c->ev_se[i]) (void)hipEventDestroy(c->ev_se[i]); if (c->ev_sc[i]) (void)hipEventDestroy(c->ev_sc[i]); if (c->ev_sa[i]) (void)hipEventDestroy(c->ev_sa[i]); }
	if (c->ev_s0) (void)hipEventDestroy(c->ev_s0);
	if (c->ev_sfork) (void)hipEventDestroy(c->ev_sfork);
	if (c->ev_sjoin) (void)hipEventDestroy(c->ev_sjoin);
	c->c2.est_val.release(); c->c2.est_cls.release(); c->c2.est_out.release();
	if (c->c2.est_stream) { (void)hipStreamSynchronize(c->c2.est_stream); (void)hipStreamDestroy(c->c2.est_stream); (void)hipEventDestroy(c->c2.ev_est_fork); (void)hipEventDestroy(c->c2.ev_est_done); }
	if (c->ckpt) (void)hipHostFree((void *)c->ckpt);
	if (c->sm_feat) (void)hipStreamDestroy(c->sm_feat);
	if (c->sm_parse) (void)hipStreamDestroy(c->sm_parse);
	if (c->sm_coder) (void)hipStreamDestroy(c->sm_coder);
	if (c->s_parse) (void)hipStreamDestroy(c->s_parse);
	if (c->s_coder) (void)hipStreamDestroy(c->s_coder);
	if (c->s_emit) (void)hipStreamDestroy(c->s_emit);
	if (c->ev_emit) (void)hipEventDestroy(c->ev_emit);
	c->seg.emit_state.release();
	hipEvent_t pev[] = { c->ev_p0, c->ev_p1, c->ev_ready };
	for (hipEvent_t e : pev) if (e) (void)hipEventDestroy(e);
	for (int i = 0; i <= X3_MAX_CKPT; i++) { if (c->ev_cb[i]) (void)hipEventDestroy(c->ev_cb[i]); if (c->ev_ce[i]) (void)hipEventDestroy(c->ev_ce[i]); }
	for (int i = 0; i < 5; i++) if (c->c2.ev[i]) (void)hipEventDestroy(c->c2.ev[i]);
	if (c->c2.side) { (void)hipStreamDestroy(c->c2.side); (void)hipEventDestroy(c->c2.ev_fork); (void)hipEventDestroy(c->c2.ev_join); }
	for (DevBuf &b : c->c2.a) b.release();
	for (DevBuf &b : c->s2.a) b.release();
	c->s2.misc.release();
	for (DevBuf &b : c->c2.y) b.release();
	c->c2.yraw.release(); c->c2.csbsmall.release(); c->c2.stat.release(); c->c2.stat0.release();
	for (DevBuf &b : c->c2.ms) b.release();
	for (int i = 0; i < 6; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

/* ------------------------------------------------------------------------------------------------------------ */
enum Stage { STAGE_SCAN = 1, STAGE_PARSE = 2, STAGE_CODE = 3 };

struct RunIO {
	const uint8_t *src; bool src_dev;   /* chunk c = src[offsets[c] .. offsets[c+1]) */
	const uint64_t *offsets; int nchunks;
	uint8_t *dst; bool dst_dev; uint64_t dst_stride; uint64_t *out_lens; /* STAGE_CODE only */
	bool want_counts;
};

static int stage_inputs(x3h_ctx *c, const x3h_params *prm, const RunIO &io, uint64_t *max_len_out)
{
	const int nc = io.nchunks;
	c->hchunks.assign((size_t)nc, X3Chunk());
	uint64_t boff = 0, eoff = 0, hoff = 0, max_len = 0;
	for (int i = 0; i < nc; i++) {
		if (io.offsets[i + 1] < io.offsets[i]) return X3H_E_ARG;
		const uint64_t len = io.offsets[i + 1] - io.offsets[i];
		if (len > X3H_MAX_CHUNK) return X3H_E_ARG;
		X3Chunk &k = c->hchunks[(size_t)i];
		k.byte_off = boff;
		k.len = (uint32_t)len;
		k.elem_off = eoff;
		k.ht_log2_max = ceil_log2(2 * (len + 1));
		if (k.ht_log2_max < X3_HT_LOG2_MIN) k.ht_log2_max = X3_HT_LOG2_MIN;
		k.ht_off = hoff;
		boff += align_up(len + prm->window_bytes + X3_PAD_EXTRA, 256);
		eoff += len + 16;
		hoff += (uint64_t)1 << k.ht_log2_max;
		if (len > max_len) max_len = len;
	}
	*max_len_out = max_len;
	c->pad_total = boff;
	CHK(c->pad.reserve(boff + 256));
	CHK(c->m.reserve(boff + 256));
	CHK(c->dict_pos.reserve(eoff * 4));
	CHK(c->dict_len.reserve(eoff));
	CHK(c->tok_pos.reserve((eoff + 4) * 4));
	CHK(c->tok_info.reserve((eoff + 4) * 4));
	CHK(c->tok_hb.reserve((eoff + 4) * 4));
	CHK(c->tok_nb.reserve((eoff + 4) * 4));
	CHK(c->tok_mb.reserve((eoff + 4) * 4));
	CHK(c->ht.reserve(hoff * 4));
	CHK(c->chunks.reserve((size_t)nc * sizeof(X3Chunk)));
	CHK(c->presult.reserve((size_t)nc * sizeof(X3ParseResult)));
	CHK(c->cresult.reserve((size_t)nc * sizeof(X3CodeResult)));

	HIPCHK(hipMemsetAsync(c->pad.p, 0, boff + 256, c->stream));
	HIPCHK(hipMemcpyAsync(c->chunks.p, c->hchunks.data(), (size_t)nc * sizeof(X3Chunk), hipMemcpyHostToDevice, c->stream));
	if (io.src_dev && nc > 4) {
		/* device-resident batch: one launch moves every chunk into its padded slot (a copy call per chunk costs more in launches than in bytes) */
		CHK(c->srcoff.reserve((size_t)(nc + 1) * 8));
		HIPCHK(hipMemcpyAsync(c->srcoff.p, io.offsets, (size_t)(nc + 1) * 8, hipMemcpyHostToDevice, c->stream));
		const uint64_t *so = c->srcoff.as<uint64_t>();
		const X3Chunk *dck = c->chunks.as<X3Chunk>();
		const uint8_t *src = io.src;
		uint8_t *pad = c->pad.as<uint8_t>();
		const uint64_t per = (max_len + 15) / 16;
		x3_foreach((size_t)(per * (uint64_t)nc), c->stream, X3_LAMBDA(size_t i) {
			const uint32_t ci = (uint32_t)(i / per);
			const uint64_t o = (uint64_t)(i % per) * 16, len = dck[ci].len;
			if (o >= len) return;
			const uint8_t *s = src + so[ci] + o;
			uint8_t *d = pad + dck[ci].byte_off + o; /* 16-byte aligned: slots are 256-byte aligned */
			const uint32_t nb = len - o >= 16 ? 16u : (uint32_t)(len - o);
			if (nb == 16 && (((uintptr_t)s) & 15) == 0) { *(uint4 *)d = *(const uint4 *)s; return; }
			for (uint32_t k = 0; k < nb; k++) d[k] = s[k];
		});
		return X3H_OK;
	}
	for (int i = 0; i < nc; i++) {
		const X3Chunk &k = c->hchunks[(size_t)i];
		if (!k.len) continue;
		HIPCHK(hipMemcpyAsync(c->pad.as<uint8_t>() + k.byte_off, io.src + io.offsets[i], k.len,
		                      io.src_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
	}
	return X3H_OK;
}

/* ------------------------------------------------------------------------------------------------------------
 * K2 + K3 of ONE long stream, pipelined.  The stages of a single stream are chains (parse: one workgroup; coder: one wavefront), but
 * they are DIFFERENT chains: the parse never reads coder state (parse.hip), and everything between them is prefix-causal counting
 * (code2.hip).  So the parse runs on its own HIP stream and publishes a checkpoint when it crosses 2 %, 8 %, 26 % and 62 % of the
 * input; for every checkpoint the host runs the coding stage of that PREFIX (features and modes recomputed chip-wide, new symbols
 * assembled) and queues the recurrence of the new symbols on a third stream.  The recurrence is ~3x slower per byte than the parse,
 * so after the first 2 % it never waits: a step costs about scan + coder instead of scan + parse + features + modes + coder.
 * Results are bit-identical by construction (same kernels, same order of symbols; only WHEN they run changes).
 * ------------------------------------------------------------------------------------------------------------ */
struct PipeStats { double ms_parse = 0, ms_features = 0, ms_modes = 0, ms_coder = 0; long long mode_iters = 0; unsigned launches = 0; };

static int pipe_setup(x3h_ctx *c)
{
	if (c->s_emit) return X3H_OK;
	if (!c->s_parse) HIPCHK(hipStreamCreate(&c->s_parse));
	if (!c->s_coder) HIPCHK(hipStreamCreate(&c->s_coder));
	HIPCHK(hipStreamCreate(&c->s_emit));
	if (!c->ev_emit) HIPCHK(hipEventCreate(&c->ev_emit));
	if (!c->ev_p0) { HIPCHK(hipEventCreate(&c->ev_p0)); HIPCHK(hipEventCreate(&c->ev_p1)); HIPCHK(hipEventCreate(&c->ev_ready)); }
	for (int i = 0; i <= X3_MAX_CKPT; i++) { HIPCHK(hipEventCreate(&c->ev_cb[i])); HIPCHK(hipEventCreate(&c->ev_ce[i])); }
	return X3H_OK;
}

static int run_pipelined_body(x3h_ctx *c, X3ParseArgs &pa, const uint8_t *d_bytes, uint32_t *tok_pos, uint32_t *tok_hb, uint32_t *tok_nb, uint32_t *tok_mb,
                              uint8_t *d_out, PipeStats *ps);

static int run_pipelined(x3h_ctx *c, X3ParseArgs &pa, const uint8_t *d_bytes, uint32_t *tok_pos, uint32_t *tok_hb, uint32_t *tok_nb, uint32_t *tok_mb,
                         uint8_t *d_out, PipeStats *ps)
{
	CHK(pipe_setup(c));
	const int rc = run_pipelined_body(c, pa, d_bytes, tok_pos, tok_hb, tok_nb, tok_mb, d_out, ps);
	if (rc != X3H_OK) { /* never return with the parse or a coder segment still running on the side streams */
		(void)hipStreamSynchronize(c->s_parse); (void)hipStreamSynchronize(c->s_coder); (void)hipStreamSynchronize(c->s_emit); (void)hipStreamSynchronize(c->stream);
	}
	return rc;
}

static int run_pipelined_body(x3h_ctx *c, X3ParseArgs &pa, const uint8_t *d_bytes, uint32_t *tok_pos, uint32_t *tok_hb, uint32_t *tok_nb, uint32_t *tok_mb,
                              uint8_t *d_out, PipeStats *ps)
{
	const uint32_t nc = (uint32_t)c->hchunks.size();
	const double *marks = c->pipe_marks;
	const uint32_t nmarks = c->pipe_nmarks;
	/* checkpoint memory (host-mapped) and the marks of every stream */
	if (c->ckpt_cap < nc) {
		if (c->ckpt) (void)hipHostFree((void *)c->ckpt);
		c->ckpt = nullptr; c->ckpt_cap = 0;
		HIPCHK(hipHostMalloc((void **)&c->ckpt, sizeof(X3ParseCkpt) * X3_CKPT_SLOTS * nc, hipHostMallocMapped | hipHostMallocCoherent));
		c->ckpt_cap = nc;
	}
	memset((void *)c->ckpt, 0, sizeof(X3ParseCkpt) * X3_CKPT_SLOTS * nc);
	std::vector<uint32_t> pos((size_t)nc * X3_MAX_CKPT, 0xFFFFFFFFu);
	uint64_t nsum = 0;
	for (uint32_t i = 0; i < nc; i++) {
		const uint32_t n = c->hchunks[i].len;
		nsum += n;
		for (uint32_t k = 0; k < nmarks; k++) { const uint32_t q = (uint32_t)((double)n * marks[k]); if (q > 0 && q < n) pos[(size_t)i * X3_MAX_CKPT + k] = q; }
	}
	CHK(c->ckpt_pos.reserve(pos.size() * 4));
	HIPCHK(hipMemcpyAsync(c->ckpt_pos.p, pos.data(), pos.size() * 4, hipMemcpyHostToDevice, c->stream));
	CHK(c->coder_state.reserve((size_t)nc * 8));
	CHK(c->prefix_result.reserve((size_t)nc * sizeof(X3ParseResult)));
	std::vector<uint32_t> init((size_t)nc * 2);
	for (uint32_t i = 0; i < nc; i++) { init[2 * i] = 0u; init[2 * i + 1] = 0x80000000u; } /* ac_init, ac.c:35-41 */
	HIPCHK(hipMemcpyAsync(c->coder_state.p, init.data(), init.size() * 4, hipMemcpyHostToDevice, c->stream));
	void *dck = nullptr;
	HIPCHK(hipHostGetDevicePointer(&dck, (void *)c->ckpt, 0));
	pa.ckpt = (X3ParseCkpt *)dck; pa.ckpt_pos = c->ckpt_pos.as<uint32_t>(); pa.nckpt = nmarks;
	/* the parse starts behind the scan (main stream) and runs on its own stream from here on */
	HIPCHK(hipEventRecord(c->ev_ready, c->stream));
	HIPCHK(hipStreamWaitEvent(c->s_parse, c->ev_ready, 0));
	HIPCHK(hipEventRecord(c->ev_p0, c->s_parse));
	x3k_launch_parse(&pa, nc, c->s_parse);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(c->ev_p1, c->s_parse));

	X3CodeSeg &seg = c->seg;
	seg.final = false; seg.coder_stream = c->s_coder; seg.ev_ready = c->ev_ready; seg.coder_state = c->coder_state.as<uint32_t>();
	{
		uint64_t longest = 0;
		for (uint32_t i = 0; i < nc; i++) if (c->hchunks[i].len > longest) longest = c->hchunks[i].len;
		const bool by_segment = c->seg_emit < 0 ? (nc >= 8 && longest <= ((uint64_t)16 << 20)) : c->seg_emit != 0;
		seg.emit_stream = by_segment ? c->s_emit : nullptr; seg.ev_emit_done = c->ev_emit;
	}
	seg.y_done.assign(nc, 0u); seg.ring_top = 0; seg.calls.clear(); seg.prev_ho.clear(); seg.prev_serial = false;
	/* the workspace is sized ONCE, for the worst case (steps, hits + elements and new-fragment bytes are each <= input bytes): a
	 * reallocation in mid-flight would wait for the running parse and coder (hipFree synchronises the device), and a token density
	 * extrapolated from the first 2 % is wrong as soon as the data changes character.  run_one only pipelines batches this fits. */
	seg.res_steps = seg.res_hits = seg.res_mbytes = seg.res_elems = (size_t)nsum + 16 * (size_t)nc; seg.res_bytes = (size_t)nsum;
	std::vector<X3ParseResult> pr(nc);
	std::vector<int> avail(nc, -1); /* newest record seen per stream: -1 none, mark index, X3_MAX_CKPT = done */
	int next = 0, nseg = 0;
	bool parse_done = false;
	for (;;) {
		/* the next record of every stream (a long step may skip a mark); a prefix call is due when EVERY stream has passed the next mark */
		int lowest = X3_MAX_CKPT;
		for (uint32_t i = 0; i < nc; i++) {
			const X3ParseCkpt *ck = c->ckpt + (size_t)i * X3_CKPT_SLOTS;
			if (avail[i] < X3_MAX_CKPT) {
				int best = avail[i];
				/* the marks are taken ONE BY ONE even when the parse is already further: a prefix call costs time in proportion to its
				 * prefix, and the coder must not run dry while a long call is under way (16 x 8 MiB: the parse is done after 100 ms, the
				 * coder needs 420 ms -- skipping from the 26 % mark straight to the end left it idle for 150 ms) */
				for (int k = avail[i] + 1; k < (int)nmarks; k++) { if (ck[k].seq == (uint32_t)k + 1) { best = k; if (k >= next) break; } }
				if ((best < next || best == avail[i]) && ck[X3_MAX_CKPT].seq == X3_MAX_CKPT + 1) best = X3_MAX_CKPT; /* no mark left to take: the stream is done */
				if (best != avail[i]) {
					avail[i] = best;
					pr[i] = X3ParseResult();
					pr[i].ntok = ck[best].ntok; pr[i].hits = ck[best].hits; pr[i].dict_elems = ck[best].dict_elems; pr[i].miss_bytes = ck[best].miss_bytes;
					pr[i]._r0 = ck[best].p; /* (scratch) parse position of the record */
				}
			}
			if (avail[i] < lowest) lowest = avail[i];
		}
		bool final = false;
		if (lowest == X3_MAX_CKPT) {
			/* every stream is parsed: final call with the kernel's own results */
			HIPCHK(hipStreamSynchronize(c->s_parse));
			c->hparse.resize(nc);
			HIPCHK(hipMemcpyAsync(c->hparse.data(), c->presult.p, (size_t)nc * sizeof(X3ParseResult), hipMemcpyDeviceToHost, c->stream));
			HIPCHK(hipStreamSynchronize(c->stream));
			pr = c->hparse;
			final = true;
		} else if (lowest < next) {
			if (!parse_done) {
				const hipError_t q = hipEventQuery(c->ev_p1);
				if (q == hipSuccess) parse_done = true; /* the done records are (about to be) visible */
				else if (q != hipErrorNotReady) { x3_last_hip = (int)q; return X3H_E_HIP; } /* the parse kernel faulted: do not spin on it */
			}
			continue; /* spin: a checkpoint should be picked up at once (the waits are milliseconds) */
		}
		next = lowest + 1;
		for (uint32_t i = 0; i < nc; i++) pr[i]._r0 = 0;
		seg.final = final;
		seg.ev_coder_begin = c->ev_cb[nseg]; seg.ev_coder_end = c->ev_ce[nseg];
		HIPCHK(hipMemcpyAsync(c->prefix_result.p, pr.data(), (size_t)nc * sizeof(X3ParseResult), hipMemcpyHostToDevice, c->stream));
		const X3ParseResult *d_pr = c->prefix_result.as<X3ParseResult>();
		CHK(x3_token_postpass(c->c2, c->stream, (int)nc, c->hchunks.data(), c->chunks.as<X3Chunk>(), d_pr, pa.tok_info, pa.dict_len,
		                      tok_pos, tok_hb, tok_nb, tok_mb, nc == 1 ? (pr[0].ntok ? pr[0].ntok : 1) : 0));
		CHK(x3_code_v2_run(c->c2, c->stream, (int)nc, c->hchunks.data(), c->chunks.as<X3Chunk>(), pr.data(), d_pr, d_bytes, tok_pos, pa.tok_info,
		                   tok_hb, tok_nb, tok_mb, d_out, c->cresult.as<X3CodeResult>(), &seg));
		nseg++;
		HIPCHK(hipStreamSynchronize(c->stream)); /* this stream only: the coder and the parse keep running */
		float ms = 0;
		(void)hipEventElapsedTime(&ms, c->c2.ev[0], c->c2.ev[1]); ps->ms_features += ms;
		(void)hipEventElapsedTime(&ms, c->c2.ev[1], c->c2.ev[2]); ps->ms_modes += ms;
		ps->mode_iters += c->c2.last.mode_iters;
		if (final) break;
	}
	HIPCHK(hipStreamSynchronize(c->s_coder));
	float ms = 0;
	(void)hipEventElapsedTime(&ms, c->ev_p0, c->ev_p1); ps->ms_parse = ms;
	for (int i = 0; i < nseg; i++) { (void)hipEventElapsedTime(&ms, c->ev_cb[i], c->ev_ce[i]); ps->ms_coder += ms; }
	ps->launches = (unsigned)nseg;
	if (getenv("X3H_DEBUG")) { /* where the coder segments sit on the time line of the call (ms after the parse started) */
		(void)hipStreamSynchronize(c->stream);
		fprintf(stderr, "[x3h] pipelined: parse %.1f ms;", ps->ms_parse);
		for (int i = 0; i < nseg; i++) { float b = 0, e = 0; (void)hipEventElapsedTime(&b, c->ev_p0, c->ev_cb[i]); (void)hipEventElapsedTime(&e, c->ev_p0, c->ev_ce[i]); fprintf(stderr, " coder segment %d: %.1f .. %.1f;", i, b, e); }
		if (c->seg.emit_stream) { float e = 0; (void)hipEventElapsedTime(&e, c->ev_p0, c->ev_emit); fprintf(stderr, " last bits written %.1f", e); }
		fprintf(stderr, "\n");
	}
	return X3H_OK;
}

/* ------------------------------------------------------------------------------------------------------------
 * K2 + K3 in SLICES (code4.hip): the parse of every stream runs on its own HIP stream and publishes checkpoints (host-mapped records) as it crosses its
 * marks; for every mark -- taken one by one -- the host queues the feature stages of the SLICE between the previous mark and this one (carried state, no
 * recomputation: a slice costs what the slice costs), the coder recurrence of the slice's symbols on a third stream and their bit emission on a fourth.
 * The host never waits for the device inside the loop: every launch is sized from the checkpoint records.
 * Returns X3S_FALLBACK (> 0) when a dictionary outgrew the sliced kernels' LDS tables: the parse has finished by then, nothing of the slices is kept, the
 * caller runs the coding stage of the whole batch stage after stage.
 * ------------------------------------------------------------------------------------------------------------ */
#define X3S_FALLBACK 1000

/* HIP streams of the sliced schedule: the handle's main stream (feature stages), a parse stream and a coder stream -- THREE, on purpose.  The ROCm
 * runtime multiplexes a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and two streams that share a queue run one
 * after the other: with five streams the feature stages of slice 0 sat behind the 90 ms parse kernel (profiles/r04_stream_queue_aliasing.txt).  The bit
 * emission of a slice needs its own queue least (it only has to be done by the end), so it is queued on the parse stream: behind the parse kernel its
 * launches run as their coder segments complete.  The move-to-front ranks run on the feature stream. */
static int sliced_setup(x3h_ctx *c)
{
	if (c->ev_s0) return X3H_OK;
	if (!c->s_parse) HIPCHK(hipStreamCreate(&c->s_parse));
	if (!c->s_coder) HIPCHK(hipStreamCreate(&c->s_coder));
	if (!c->ev_emit) HIPCHK(hipEventCreate(&c->ev_emit));
	if (!c->ev_p0) { HIPCHK(hipEventCreate(&c->ev_p0)); HIPCHK(hipEventCreate(&c->ev_p1)); HIPCHK(hipEventCreate(&c->ev_ready)); }
	/* (each event is made once: a call that failed halfway is completed by the next one, nothing leaks) */
	if (!c->ev_sfork) HIPCHK(hipEventCreate(&c->ev_sfork));
	if (!c->ev_sjoin) HIPCHK(hipEventCreate(&c->ev_sjoin));
	for (uint32_t i = 0; i < X3S_MAX_SLICES + 2; i++) {
		hipEvent_t *evs[5] = { &c->ev_sf[i], &c->ev_sb[i], &c->ev_se[i], &c->ev_sc[i], &c->ev_sa[i] };
		for (hipEvent_t *e : evs) if (!*e) HIPCHK(hipEventCreate(e));
	}
	HIPCHK(hipEventCreate(&c->ev_s0));
	return X3H_OK;
}

/* the three masked streams of the mid-size layout: coder = the top 32 CUs of the device's CU mask, parse and features = all the others */
static bool sliced_masked_setup(x3h_ctx *c)
{
	if (c->sm_state) return c->sm_state > 0;
	c->sm_state = -1;
	int ncu = 0;
	if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || ncu < 128 || ncu > 1024) return false;
	const uint32_t words = ((uint32_t)ncu + 31) / 32;
	std::vector<uint32_t> top(words, 0), rest(words, 0);
	for (int i = 0; i < ncu; i++) (i >= ncu - 32 ? top : rest)[(size_t)i / 32] |= 1u << (i % 32);
	if (hipExtStreamCreateWithCUMask(&c->sm_feat, words, rest.data()) != hipSuccess || hipExtStreamCreateWithCUMask(&c->sm_parse, words, rest.data()) != hipSuccess ||
	    hipExtStreamCreateWithCUMask(&c->sm_coder, words, top.data()) != hipSuccess) { (void)hipGetLastError(); return false; }
	c->sm_state = 1;
	return true;
}

static int run_sliced_body(x3h_ctx *c, X3ParseArgs &pa, const uint8_t *d_bytes, uint8_t *d_out, PipeStats *ps, std::vector<float> *est)
{
	const uint32_t nc = (uint32_t)c->hchunks.size();
	/* the three streams of the slices: features, parse (+ bit emission), coder -- plain ones, or the masked ones of the mid-size layout */
	uint64_t longest_len = 0;
	for (uint32_t i = 0; i < nc; i++) if (c->hchunks[i].len > longest_len) longest_len = c->hchunks[i].len;
	/* (round 5: streams longer than X3H_SLICE_CUMASK_LEN take the masked layout too while the coder's 32 CUs hold at most two chains each -- config 4's share: feature stages 147 -> 55 ms
	 * beside the coder, whose own lane shrinks 395 -> 383 ms, the call 427.8 -> 408.0 ms; config 3 1 957 -> 1 941 ms; profiles/r05_slice_marks.txt) */
	const bool masked = c->slice_cumask && nc >= c->slice_cumask_min_streams && (longest_len <= c->slice_cumask_max_len || nc <= 64) && longest_len >= c->slice_cumask_min_len && sliced_masked_setup(c);
	hipStream_t sF = masked ? c->sm_feat : c->stream, sP = masked ? c->sm_parse : c->s_parse, sC = masked ? c->sm_coder : c->s_coder;
	/* how many slices: a slice costs a few hundred microseconds of launches and of latency-bound per-stream kernels whatever its size, so the longest stream is cut
	 * about every 150 KB, into at most the full set of marks (small slices first: the coder starts early); X3H_SLICE_MARKS fixes the set */
	uint32_t nmarks = c->slice_nmarks;
	double marks[X3_MAX_CKPT];
	for (uint32_t k = 0; k < nmarks; k++) marks[k] = c->slice_marks[k];
	if (!c->slice_marks_fixed) {
		if (nc <= 2) { /* one or two streams: no other stream's segment to wait for in a coder launch, so more and smaller slices (297 against 299 ms on the dickens-sized stream) */
			/* (round 5: a first mark after 0.4 % -- the coder's first symbol waits for scan + parse to the first mark + that slice's stages: 296.9 -> 295.2 ms) */
			static const double fine[11] = { 0.004, 0.02, 0.05, 0.10, 0.17, 0.26, 0.36, 0.47, 0.59, 0.72, 0.86 };
			nmarks = 11;
			for (uint32_t k = 0; k < nmarks; k++) marks[k] = fine[k];
		}
		const uint64_t longest = longest_len;
		const uint64_t want = longest / (150u << 10);
		if (masked && longest < (900u << 10)) {
			/* mid-size layout: with the coder on CUs of its own the feature kernels of a slice run at full speed beside it, so more and smaller slices pay (the dickens-sized
			 * bytes as 24 / 40 / 64 / 96 chunks: 20.6 -> 17.6, 14.0 -> 12.9, 11.1 -> 10.2, 9.5 -> 8.4 ms; profiles/r04_midsize_marks_and_timeline.txt) */
			static const double m3[3] = { 0.13, 0.40, 0.70 }, m4[4] = { 0.10, 0.25, 0.45, 0.70 }, m5[5] = { 0.08, 0.20, 0.35, 0.55, 0.78 };
			const double *m = longest < (240u << 10) ? m3 : longest < (400u << 10) ? m4 : m5;
			nmarks = longest < (240u << 10) ? 3 : longest < (400u << 10) ? 4 : 5;
			for (uint32_t k = 0; k < nmarks; k++) marks[k] = m[k];
		} else if (want < nmarks + 1) {
			nmarks = want >= 2 ? (uint32_t)want - 1 : 1u;
			for (uint32_t k = 0; k < nmarks; k++) marks[k] = (double)(k + 1) / (double)(nmarks + 1);
		}
	}
	if (c->ckpt_cap < nc) {
		if (c->ckpt) (void)hipHostFree((void *)c->ckpt);
		c->ckpt = nullptr; c->ckpt_cap = 0;
		HIPCHK(hipHostMalloc((void **)&c->ckpt, sizeof(X3ParseCkpt) * X3_CKPT_SLOTS * nc, hipHostMallocMapped | hipHostMallocCoherent));
		c->ckpt_cap = nc;
	}
	memset((void *)c->ckpt, 0, sizeof(X3ParseCkpt) * X3_CKPT_SLOTS * nc);
	/* marks: fractions of every stream's length, at least 8 KiB apart and 8 KiB from either end -- the parse looks at its marks once per block of 2048
	 * positions, so every mark is published by its own block (none is skipped, none falls together with the end of the stream) and a slice is bounded
	 * by the distance of two marks + two blocks */
	std::vector<uint32_t> pos((size_t)nc * X3_MAX_CKPT, 0xFFFFFFFFu);
	uint64_t slice_bytes = 0;
	uint32_t min_gap = 8192;
	if (const char *e = getenv("X3H_SLICE_GAP")) { const int v = atoi(e); if (v > 0) min_gap = (uint32_t)v; } /* (tests: small inputs in many slices; a skipped mark merges two slices, a slice beyond the bound falls back) */
	for (uint32_t i = 0; i < nc; i++) {
		const uint32_t n = c->hchunks[i].len;
		uint32_t prev = 0, gap = 0, used = 0; /* the marks a stream keeps are its marks 0, 1, ...: the parse waits for them in that order */
		for (uint32_t k = 0; k < nmarks; k++) {
			const uint32_t q = (uint32_t)((double)n * marks[k]);
			if (q >= prev + min_gap && (uint64_t)q + min_gap <= n) { pos[(size_t)i * X3_MAX_CKPT + used++] = q; if (q - prev > gap) gap = q - prev; prev = q; }
		}
		if (n - prev > gap) gap = n - prev;
		slice_bytes += (uint64_t)gap + 2 * (X3_PARSE_PB + 64) + 16;
	}
	CHK(c->ckpt_pos.reserve(pos.size() * 4));
	HIPCHK(hipMemcpyAsync(c->ckpt_pos.p, pos.data(), pos.size() * 4, hipMemcpyHostToDevice, c->stream));
	CHK(x3s_begin(c->sr, c->stream, nc, c->hchunks.data(), slice_bytes, slice_bytes));
	CHK(x3_zero_output_slots(c->stream, nc, c->hchunks.data(), c->chunks.as<X3Chunk>(), d_out));
	void *dck = nullptr;
	HIPCHK(hipHostGetDevicePointer(&dck, (void *)c->ckpt, 0));
	pa.ckpt = (X3ParseCkpt *)dck; pa.ckpt_pos = c->ckpt_pos.as<uint32_t>(); pa.nckpt = nmarks;
	HIPCHK(hipEventRecord(c->ev_ready, c->stream));
	HIPCHK(hipStreamWaitEvent(sP, c->ev_ready, 0));
	HIPCHK(hipStreamWaitEvent(sC, c->ev_ready, 0));
	if (sF != c->stream) HIPCHK(hipStreamWaitEvent(sF, c->ev_ready, 0));
	HIPCHK(hipEventRecord(c->ev_p0, sP));
	x3k_launch_parse(&pa, nc, sP);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(c->ev_p1, sP));
	hipStream_t s_emit = sP; /* see sliced_setup */

	X3SliceRun &R = c->sr;
	uint32_t *sm = R.small.as<uint32_t>();
	struct Bound { uint32_t t, h, d, mb, p; };
	std::vector<Bound> prev(nc, Bound{ 0, 0, 0, 0, 0 }), cur(nc);
	std::vector<int> avail(nc, -1); /* newest record taken per stream: -1 none, mark index, X3_MAX_CKPT = done */
	std::vector<char> ended(nc, 0);
	std::vector<X3Slice> hs(nc);
	int next = 0, nslice = 0;
	bool parse_done = false, fallback = false;
	uint64_t max_dict = 1;
	uint32_t *emit_off[X3S_MAX_SLICES + 2] = {}, *emit_len[X3S_MAX_SLICES + 2] = {};
	auto queue_emit = [&](int k, bool last) -> int {
		HIPCHK(hipStreamWaitEvent(s_emit, c->ev_se[k], 0));
		X3EmitArgs ea;
		ea.yoc = nullptr; ea.sym = R.sym.as<uint4>(); ea.state = R.states.as<uint32_t>(); ea.final_lo = sm + X3S_FINALLO * nc; ea.chunks = c->chunks.as<X3Chunk>(); ea.parsed = nullptr;
		ea.npairs = sm + X3S_NPAIRS * nc; ea.evfinal = sm + X3S_EVFINAL * nc; ea.out = d_out; ea.result = c->cresult.as<X3CodeResult>();
		ea.seg_off = emit_off[k]; ea.seg_len = emit_len[k]; ea.carry = sm + X3S_EMITCARRY * nc; ea.last = last ? 1u : 0u; ea.compact = 0;
		ea.ntok = sm + X3S_NTOK * nc; ea.nhits = sm + X3S_NHITS * nc;
		return x3s_emit_launch(ea, nc, s_emit);
	};
	for (;;) {
		int lowest = X3_MAX_CKPT;
		for (uint32_t i = 0; i < nc; i++) {
			const X3ParseCkpt *ck = c->ckpt + (size_t)i * X3_CKPT_SLOTS;
			if (avail[i] < X3_MAX_CKPT && avail[i] < next) { /* (a stream that has its record for this slice takes no further one while the others are waited for: a slice
				                                                   * is ONE step of every stream from mark to mark -- that is what the slice buffers are sized for) */
				int best = avail[i];
				for (int k = avail[i] + 1; k < (int)nmarks; k++) {
					if (ck[k].seq != (uint32_t)k + 1) break; /* marks are published in order: behind the first unpublished one nothing is taken (a later one seen here means
					                                           * this thread was descheduled between two reads, and taking it would make ONE slice of two gaps) */
					best = k;
					if (k >= next) break;
				}
				if ((best < next || best == avail[i]) && ck[X3_MAX_CKPT].seq == X3_MAX_CKPT + 1) best = X3_MAX_CKPT; /* no mark left to take: the stream is done */
				avail[i] = best;
			}
			if (avail[i] < lowest) lowest = avail[i];
		}
		if (lowest < next && lowest < X3_MAX_CKPT) {
			if (!parse_done) {
				const hipError_t q = hipEventQuery(c->ev_p1);
				if (q == hipSuccess) parse_done = true;
				else if (q != hipErrorNotReady) { x3_last_hip = (int)q; return X3H_E_HIP; } /* the parse kernel faulted: do not spin on it */
			}
			continue; /* spin: a checkpoint is picked up at once (the waits are milliseconds) */
		}
		const bool final = lowest == X3_MAX_CKPT;
		next = lowest + 1;
		/* this slice of every stream: from where the previous one ended to the newest record taken */
		uint32_t sh = 0, se = 0, smi = 0, sb = 0, ss = 0, sy = 0;
		for (uint32_t i = 0; i < nc; i++) {
			const X3ParseCkpt *ck = c->ckpt + (size_t)i * X3_CKPT_SLOTS;
			Bound b = prev[i];
			if (avail[i] >= 0) { const X3ParseCkpt &r = ck[avail[i]]; b.t = r.ntok; b.h = r.hits; b.d = r.dict_elems; b.mb = r.miss_bytes; b.p = r.p; }
			cur[i] = b;
			if (b.d > max_dict) max_dict = b.d;
			X3Slice &s = hs[i];
			s.t0 = prev[i].t; s.t1 = b.t; s.h0 = prev[i].h; s.h1 = b.h; s.d0 = prev[i].d; s.d1 = b.d; s.mb0 = prev[i].mb; s.mb1 = b.mb; s.p0 = prev[i].p;
			s.sh = sh; s.se = se; s.sm = smi; s.sb = sb; s.ss = ss; s.sy = sy;
			s.last = (avail[i] == X3_MAX_CKPT && !ended[i]) ? 1u : 0u;
			if (s.last) ended[i] = 1;
			s.ended = ended[i];
			sh += s.h1 - s.h0; se += (s.h1 - s.h0) + (s.d1 - s.d0); smi += (s.t1 - s.t0) - (s.h1 - s.h0); sb += s.mb1 - s.mb0; ss += s.t1 - s.t0;
			sy += 2 * (s.t1 - s.t0) + (s.mb1 - s.mb0) + s.last;
		}
		if (max_dict > X3S_DMAX || ss > slice_bytes || sb > slice_bytes) { /* (the last two: a safety net for the slice buffers -- tokens and new-fragment bytes; slices are single mark-to-mark steps, see above) */
			if (getenv("X3H_DEBUG")) fprintf(stderr, "[x3h] sliced: falling back in slice %d: dictionary %llu elements (limit %u), slice of %u tokens (limit %llu)\n", nslice, (unsigned long long)max_dict, (unsigned)X3S_DMAX, ss, (unsigned long long)slice_bytes);
			if (getenv("X3H_DEBUG")) for (uint32_t i = 0; i < nc; i++) fprintf(stderr, "[x3h]   stream %u (%u bytes): tokens %u..%u, record taken %d\n", i, c->hchunks[i].len, hs[i].t0, hs[i].t1, avail[i]);
			fallback = true; break;
		}
		if (nslice >= (int)X3S_MAX_SLICES + 1) return X3H_E_INTERNAL;
		uint32_t *d_segoff = nullptr, *d_seglen = nullptr;
		/* stage A (records, ranks, context statistics) on the handle's stream; its slice temporaries are one of two sets, the one stage B of slice k - 2 has read */
		if (nslice >= 2) HIPCHK(hipStreamWaitEvent(sF, c->ev_sf[nslice - 2], 0));
		HIPCHK(hipEventRecord(c->ev_sb[nslice], sF));
		CHK(x3s_slice(R, sF, sF, c->ev_sfork, c->ev_sjoin, c->chunks.as<X3Chunk>(), hs, max_dict, d_bytes, pa.tok_info, pa.dict_len, final,
		              c->c2.want_est, &d_segoff, &d_seglen, 1));
		HIPCHK(hipEventRecord(c->ev_sa[nslice], sF));
		/* stage B (mode chain, index / order-0 models, symbol assembly): once the parse kernel has finished its HIP stream is free, and stage B of this slice runs
		 * there beside stage A of the next one; while the parse is still running (long streams) it stays behind stage A on this stream */
		if (!parse_done) { const hipError_t q = hipEventQuery(c->ev_p1); if (q == hipSuccess) parse_done = true; }
		hipStream_t sB = parse_done && c->slice_bstream ? sP : sF;
		if (sB != sF) {
			HIPCHK(hipStreamWaitEvent(sB, c->ev_sa[nslice], 0));
			if (nslice >= 1) HIPCHK(hipStreamWaitEvent(sB, c->ev_sf[nslice - 1], 0));
		} else if (nslice >= 1) HIPCHK(hipStreamWaitEvent(sB, c->ev_sf[nslice - 1], 0)); /* (stage B of the slice before may have run on the other stream) */
		CHK(x3s_slice(R, sB, sB, c->ev_sfork, c->ev_sjoin, c->chunks.as<X3Chunk>(), hs, max_dict, d_bytes, pa.tok_info, pa.dict_len, final,
		              c->c2.want_est, &d_segoff, &d_seglen, 2));
		if (c->c2.want_est) { /* the reference's float accumulators (x3.c:43), continued in coding order: one more chain per stream, behind stage B */
			X3EstArgs ea;
			ea.range = nullptr; ea.val = R.est_val.as<float>(); ea.cls = R.est_cls.as<uint8_t>(); ea.out = (float *)(sm + X3S_EST * nc);
			ea.seg_first = sm + X3S_ESTFIRST * nc; ea.seg_count = sm + X3S_ESTCNT * nc;
			CHK(x3s_est_launch(ea, nc, sB));
		}
		HIPCHK(hipEventRecord(c->ev_sf[nslice], sB));
		/* coder recurrence of the slice's symbols, then their bits (both carry their state per stream from launch to launch) */
		HIPCHK(hipStreamWaitEvent(sC, c->ev_sf[nslice], 0));
		HIPCHK(hipEventRecord(c->ev_sc[nslice], sC));
		CHK(x3s_ac2_launch(R.sym.as<uint4>(), R.states.as<uint32_t>(), sm + X3S_FINALLO * nc, d_segoff, d_seglen, sm + X3S_CODER * nc, nc, sC));
		HIPCHK(hipEventRecord(c->ev_se[nslice], sC));
		/* the bit emission of a slice is queued on the parse stream ONE SLICE LATE (and the last ones behind the loop): it waits for its coder segment, and
		 * whatever is queued behind it on that stream -- stage B of a later slice -- must not wait that long */
		emit_off[nslice] = d_segoff; emit_len[nslice] = d_seglen;
		if (nslice >= 1) CHK(queue_emit(nslice - 1, false));
		if (final) CHK(queue_emit(nslice, true));
		prev = cur;
		nslice++;
		if (final) break;
	}
	if (fallback) { /* nothing of the slices is kept: wait for everything in flight, the caller codes the batch stage after stage */
		HIPCHK(hipStreamSynchronize(sP)); HIPCHK(hipStreamSynchronize(sF)); HIPCHK(hipStreamSynchronize(sC)); HIPCHK(hipStreamSynchronize(c->stream));
		return X3S_FALLBACK;
	}
	HIPCHK(hipEventRecord(c->ev_emit, s_emit));
	HIPCHK(hipStreamWaitEvent(c->stream, c->ev_emit, 0));
	/* results: what the parse counted (its own records) and, if asked for, the size estimates */
	c->hparse.resize(nc);
	HIPCHK(hipStreamSynchronize(sP));
	HIPCHK(hipMemcpyAsync(c->hparse.data(), c->presult.p, (size_t)nc * sizeof(X3ParseResult), hipMemcpyDeviceToHost, c->stream));
	std::vector<uint32_t> hstat(nc);
	HIPCHK(hipMemcpyAsync(hstat.data(), sm + X3S_STATUS * nc, (size_t)nc * 4, hipMemcpyDeviceToHost, c->stream));
	if (est && c->c2.want_est) { est->resize((size_t)nc * 4); HIPCHK(hipMemcpyAsync(est->data(), sm + X3S_EST * nc, (size_t)nc * 16, hipMemcpyDeviceToHost, c->stream)); }
	HIPCHK(hipStreamSynchronize(c->stream));
	for (uint32_t i = 0; i < nc; i++) if (hstat[i] != X3_ST_OK) {
		if (getenv("X3H_DEBUG")) fprintf(stderr, "[x3h] sliced: stream %u ended with status %u: falling back\n", i, hstat[i]);
		return X3S_FALLBACK;
	} /* a list pool's bound was violated (adversarial input): every stream is idle, the caller codes the batch stage after stage */
	float ms = 0;
	(void)hipEventElapsedTime(&ms, c->ev_p0, c->ev_p1); ps->ms_parse = ms;
	for (int i = 0; i < nslice; i++) {
		(void)hipEventElapsedTime(&ms, c->ev_sb[i], c->ev_sa[i]); ps->ms_features += ms; /* stage A */
		(void)hipEventElapsedTime(&ms, c->ev_sa[i], c->ev_sf[i]); ps->ms_modes += ms;    /* stage B (mode chain, models, assembly; its wait for the stream it runs on included) */
		(void)hipEventElapsedTime(&ms, c->ev_sf[i], c->ev_se[i]); /* (includes the wait of a segment for its predecessor) */
	}
	{ /* coder: first segment's begin to last segment's end minus nothing -- the segments of a stream follow each other without a gap when the pipeline is fed */
		float tot = 0;
		for (int i = 0; i < nslice; i++) { (void)hipEventElapsedTime(&ms, c->ev_sc[i], c->ev_se[i]); tot += ms; }
		ps->ms_coder = tot;
	}
	ps->mode_iters = 0; ps->launches = (unsigned)nslice;
	c->c2.last.symbols = 0; c->c2.last.chain_symbols = 0;
	{
		std::vector<uint32_t> yc(nc);
		HIPCHK(hipMemcpy(yc.data(), sm + X3S_YCNT * nc, (size_t)nc * 4, hipMemcpyDeviceToHost));
		uint64_t y = 0, raw = 0;
		for (uint32_t i = 0; i < nc; i++) { y += yc[i]; raw += 2ull * c->hparse[i].ntok + c->hparse[i].miss_bytes + 1; }
		c->c2.last.symbols = raw; c->c2.last.chain_symbols = y;
	}
	if (getenv("X3H_DEBUG")) {
		fprintf(stderr, "[x3h] sliced: %d slices, parse %.1f ms;", nslice, ps->ms_parse);
		for (int i = 0; i < nslice; i++) { float f0 = 0, f1 = 0, b = 0, e = 0; (void)hipEventElapsedTime(&f0, c->ev_p0, c->ev_sb[i]); (void)hipEventElapsedTime(&f1, c->ev_p0, c->ev_sf[i]);
			(void)hipEventElapsedTime(&b, c->ev_p0, c->ev_sc[i]); (void)hipEventElapsedTime(&e, c->ev_p0, c->ev_se[i]); fprintf(stderr, " [%d] features %.2f..%.2f coder %.2f..%.2f;", i, f0, f1, b, e); }
		{ float e = 0; (void)hipEventElapsedTime(&e, c->ev_p0, c->ev_emit); fprintf(stderr, " last bits written %.2f\n", e); }
	}
	return X3H_OK;
}

static int run_sliced(x3h_ctx *c, X3ParseArgs &pa, const uint8_t *d_bytes, uint8_t *d_out, PipeStats *ps, std::vector<float> *est)
{
	CHK(sliced_setup(c));
	const int rc = run_sliced_body(c, pa, d_bytes, d_out, ps, est);
	if (rc != X3H_OK && rc != X3S_FALLBACK) { /* never return with work in flight on the side streams */
		(void)hipStreamSynchronize(c->s_parse); (void)hipStreamSynchronize(c->s_coder); (void)hipStreamSynchronize(c->stream);
		if (c->sm_state > 0) { (void)hipStreamSynchronize(c->sm_parse); (void)hipStreamSynchronize(c->sm_coder); (void)hipStreamSynchronize(c->sm_feat); }
	}
	return rc;
}

static int run_one(x3h_ctx *c, const x3h_params *prm_in, const RunIO &io, Stage upto, x3h_stats *stats)
{
	if (!c || !io.offsets || io.nchunks <= 0 || (!io.src && io.offsets[io.nchunks] != io.offsets[0])) return X3H_E_ARG;
	x3h_params dp;
	if (!prm_in) { x3h_default_params(&dp); prm_in = &dp; }
	const x3h_params prm = *prm_in;
	if (prm.window_bytes > (1u << 24)) return X3H_E_ARG; /* -w up to 16384 (KiB) */
	HIPCHK(hipSetDevice(c->device));
	const int nc = io.nchunks;
	uint64_t max_len = 0;
	const double dbg_t0 = x3_now_ms(), dbg_a0 = x3_alloc_ms;
	const unsigned long long dbg_b0 = x3_alloc_bytes, dbg_c0 = x3_alloc_calls;
	struct DbgAlloc { double t0, a0; unsigned long long b0, c0; ~DbgAlloc() { if (getenv("X3H_DEBUG")) fprintf(stderr, "[x3h] call: %.1f ms wall, of which %.1f ms in %llu hipMalloc/hipFree calls for %.2f GB of workspace\n",
		x3_now_ms() - t0, x3_alloc_ms - a0, x3_alloc_calls - c0, (double)(x3_alloc_bytes - b0) / 1e9); } } dbg_alloc = { dbg_t0, dbg_a0, dbg_b0, dbg_c0 };

	HIPCHK(hipEventRecord(c->ev[0], c->stream));
	CHK(stage_inputs(c, &prm, io, &max_len));
	HIPCHK(hipEventRecord(c->ev[1], c->stream));

	/* ---- K1 ---- */
	X3ScanArgs sa;
	sa.bytes = c->pad.as<uint8_t>();
	sa.chunks = c->chunks.as<X3Chunk>();
	sa.m = c->m.as<uint8_t>();
	sa.counts = nullptr;
	sa.window = prm.window_bytes;
	sa.max_match_count = prm.max_match_count;
	if (io.want_counts) {
		CHK(c->counts.reserve((size_t)c->hchunks[0].len * 32 * 4 + 256));
		sa.counts = c->counts.as<uint32_t>();
	}
	if (io.want_counts || c->scan_v1) {
		/* brute-force sweep (scan.hip): the raw histogram for tests, and the A/B reference of the v2 scan (X3H_SCAN_V1=1) */
		x3k_launch_scan(&sa, (uint32_t)max_len, (uint32_t)nc, c->stream);
		HIPCHK(hipGetLastError());
	} else {
		CHK(x3_scan_v2_run(c->s2, c->c2.tmp, c->stream, nc, c->hchunks.data(), c->chunks.as<X3Chunk>(), sa.bytes, sa.m, c->pad_total,
		                   prm.window_bytes, prm.max_match_count));
	}
	HIPCHK(hipEventRecord(c->ev[2], c->stream));
	if (upto == STAGE_SCAN) { HIPCHK(hipStreamSynchronize(c->stream)); return x3p_check_error(c->stream); }

	/* ---- K2 ---- */
	/* pipelined schedule: a few long streams (the serial chains dominate); many short ones run stage after stage (the chip-wide passes dominate) */
	/* K3 in slices (code4.hip) wherever it applies: up to a few streams per CU, the longest one long enough for a few slices (dictionaries beyond the sliced
	 * kernels' LDS tables fall back inside run_sliced) */
	uint64_t elems_total = 0;
	for (int i = 0; i < nc; i++) elems_total += (uint64_t)c->hchunks[(size_t)i].len + 16;
	bool sliced = upto == STAGE_CODE && c->sliced && c->pipe_min /* (X3H_PIPE_MIN=0: stage after stage, whatever the batch) */ && nc <= c->sliced_max_streams && max_len >= c->sliced_min && elems_total < ((uint64_t)1 << 29);
	const bool pipe = !sliced && upto == STAGE_CODE && nc <= c->pipe_max_streams && c->pipe_min && max_len >= c->pipe_min &&
	                  c->pad_total <= ((uint64_t)320 << 20); /* worst-case workspace of the pipelined schedule: ~330 B per input byte */
	X3ParseArgs pa;
	pa.ckpt = nullptr; pa.nckpt = 0;
	pa.bytes = sa.bytes; pa.chunks = sa.chunks; pa.m = sa.m;
	pa.dict_pos = c->dict_pos.as<uint32_t>(); pa.dict_len = c->dict_len.as<uint8_t>();
	pa.ht = c->ht.as<uint32_t>();
	pa.tok_info = c->tok_info.as<uint32_t>();
	uint32_t *tok_pos = c->tok_pos.as<uint32_t>(), *tok_hb = c->tok_hb.as<uint32_t>(), *tok_nb = c->tok_nb.as<uint32_t>(), *tok_mb = c->tok_mb.as<uint32_t>();
	pa.result = c->presult.as<X3ParseResult>();
	pa.factor1 = prm.factor1; pa.factor2 = prm.factor2; pa.nl_mode = prm.nl_mode;
	/* batches of many streams: the coding stage walks the tokens itself (one workgroup per stream, code3.hip) */
	const bool tokens_in_code = !pipe && upto == STAGE_CODE && (uint32_t)nc >= X3_STREAM_MIN_STREAMS;
	if (!pipe && !sliced) {
		x3k_launch_parse(&pa, (uint32_t)nc, c->stream);
		HIPCHK(hipGetLastError());
		if (!tokens_in_code)
			CHK(x3_token_postpass(c->c2, c->stream, nc, c->hchunks.data(), c->chunks.as<X3Chunk>(), pa.result, pa.tok_info, pa.dict_len,
			                      tok_pos, tok_hb, tok_nb, tok_mb, 0, upto == STAGE_PARSE));
		HIPCHK(hipEventRecord(c->ev[3], c->stream));
		c->hparse.resize((size_t)nc);
		HIPCHK(hipMemcpyAsync(c->hparse.data(), c->presult.p, (size_t)nc * sizeof(X3ParseResult), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		if (getenv("X3H_DEBUG")) for (int i = 0; i < nc && i < 4; i++) { const X3ParseResult &r = c->hparse[(size_t)i];
			fprintf(stderr, "[x3h] chunk %d: steps %u hits %u D %u missbytes %u | parse kcycles: fill %u patch %u table %u walk %u\n", i, r.ntok, r.hits, r.dict_elems, r.miss_bytes, r.kcyc_fill, r.kcyc_patch, r._r0, r.kcyc_walk); }
		if (upto == STAGE_PARSE) return X3H_OK;
	} else {
		HIPCHK(hipEventRecord(c->ev[3], c->stream)); /* the parse overlaps the coding stage (run_pipelined): its time is reported from its own events */
		c->hparse.assign((size_t)nc, X3ParseResult());
	}

	/* ---- where every chunk's stream goes ---- */
	uint64_t ooff = 0;
	for (int i = 0; i < nc; i++) {
		X3Chunk &k = c->hchunks[(size_t)i];
		if (io.dst_dev) {
			k.out_off = (uint64_t)i * io.dst_stride;
			k.out_cap = io.dst_stride & ~(uint64_t)3;
		} else {
			uint64_t cap = io.dst_stride & ~(uint64_t)3, bound = align_up(x3h_compress_bound(k.len), 4);
			if (cap > bound) cap = bound;
			k.out_off = ooff;
			k.out_cap = cap;
			ooff += align_up(cap, 256);
		}
	}
	if (!io.dst_dev) CHK(c->out.reserve(ooff + 256));
	uint8_t *d_out = io.dst_dev ? io.dst : c->out.as<uint8_t>();
	HIPCHK(hipMemcpyAsync(c->chunks.p, c->hchunks.data(), (size_t)nc * sizeof(X3Chunk), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipEventRecord(c->ev[4], c->stream));
	PipeStats ps;
	std::vector<float> hest;
	bool fell_back = false;
	if (sliced) {
		const int r = run_sliced(c, pa, sa.bytes, d_out, &ps, &hest);
		if (r == X3S_FALLBACK) { /* a dictionary outgrew the sliced kernels: the parse is complete, the coding stage runs stage after stage */
			fell_back = true; sliced = false; hest.clear();
			c->sr.release(); /* the carried state of the slices (~240 B per input byte) makes room for the stage-after-stage workspace */
			c->hparse.resize((size_t)nc);
			HIPCHK(hipMemcpyAsync(c->hparse.data(), c->presult.p, (size_t)nc * sizeof(X3ParseResult), hipMemcpyDeviceToHost, c->stream));
			HIPCHK(hipStreamSynchronize(c->stream));
			pa.ckpt = nullptr; pa.nckpt = 0;
			if (!tokens_in_code)
				CHK(x3_token_postpass(c->c2, c->stream, nc, c->hchunks.data(), c->chunks.as<X3Chunk>(), pa.result, pa.tok_info, pa.dict_len, tok_pos, tok_hb, tok_nb, tok_mb, 0, false));
			CHK(x3_code_v2_run(c->c2, c->stream, nc, c->hchunks.data(), c->chunks.as<X3Chunk>(), c->hparse.data(), pa.result,
			                   sa.bytes, tok_pos, pa.tok_info, tok_hb, tok_nb, tok_mb, d_out, c->cresult.as<X3CodeResult>(), nullptr,
			                   tokens_in_code ? pa.dict_len : nullptr));
		} else CHK(r);
	} else if (pipe) {
		CHK(run_pipelined(c, pa, sa.bytes, tok_pos, tok_hb, tok_nb, tok_mb, d_out, &ps));
	} else {
		/* parallel feature extraction (sorts / scans / count-smaller-before) + two thin serial passes (code2.hip) */
		CHK(x3_code_v2_run(c->c2, c->stream, nc, c->hchunks.data(), c->chunks.as<X3Chunk>(), c->hparse.data(), pa.result,
		                   sa.bytes, tok_pos, pa.tok_info, tok_hb, tok_nb, tok_mb, d_out, c->cresult.as<X3CodeResult>(), nullptr,
		                   tokens_in_code ? pa.dict_len : nullptr));
	}
	if (c->c2.est_pending) { /* the size estimates ran beside the coder on their own stream */
		HIPCHK(hipStreamWaitEvent(c->stream, c->c2.ev_est_done, 0));
		c->c2.est_pending = false;
		hest.resize((size_t)nc * 4);
	}
	HIPCHK(hipEventRecord(c->ev[5], c->stream));
	c->hcode.resize((size_t)nc);
	HIPCHK(hipMemcpyAsync(c->hcode.data(), c->cresult.p, (size_t)nc * sizeof(X3CodeResult), hipMemcpyDeviceToHost, c->stream));
	if (!hest.empty() && !sliced) HIPCHK(hipMemcpyAsync(hest.data(), c->c2.est_out.p, hest.size() * 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));

	const int scan_rc = x3p_check_error(c->stream); /* (a chained scan that gave up its wait: never seen, never to be returned as a result) */
	int rc = X3H_OK;
	for (int i = 0; i < nc; i++) {
		const X3CodeResult &r = c->hcode[(size_t)i];
		if (r.status == X3_ST_OUT_FULL) rc = X3H_E_OUTPUT_FULL;
		else if (r.status != X3_ST_OK && rc == X3H_OK) rc = X3H_E_INTERNAL;
		if (io.out_lens) io.out_lens[i] = r.out_len;
	}
	if (scan_rc != X3H_OK) rc = scan_rc;
	if (rc == X3H_OK && !io.dst_dev) {
		for (int i = 0; i < nc; i++) {
			const X3Chunk &k = c->hchunks[(size_t)i];
			HIPCHK(hipMemcpyAsync(io.dst + (uint64_t)i * io.dst_stride, c->out.as<uint8_t>() + k.out_off, c->hcode[(size_t)i].out_len,
			                      hipMemcpyDeviceToHost, c->stream));
		}
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	if (stats) {
		memset(stats, 0, sizeof(*stats));
		for (int i = 0; i < nc; i++) {
			for (int e = 0; e < 5; e++) stats->events[e] += c->hcode[(size_t)i].events[e];
			stats->dict_elems += c->hparse[(size_t)i].dict_elems;
			stats->ctx0_entries += c->hcode[(size_t)i].pairs;
			stats->steps += c->hparse[(size_t)i].ntok;
			if (!hest.empty()) for (int e = 0; e < 4; e++) stats->est_bits[e] += (double)hest[(size_t)i * 4 + e];
		}
		float ms = 0;
		(void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]); stats->ms_copy = ms;
		(void)hipEventElapsedTime(&ms, c->ev[1], c->ev[2]); stats->ms_scan = ms;
		(void)hipEventElapsedTime(&ms, c->ev[2], c->ev[3]); stats->ms_parse = ms;
		(void)hipEventElapsedTime(&ms, c->ev[4], c->ev[5]); stats->ms_code = ms;
		(void)hipEventElapsedTime(&ms, c->ev[0], c->ev[5]); stats->ms_total = ms;
		if (!pipe && !sliced && c->c2.ev[4]) {
			(void)hipEventElapsedTime(&ms, c->c2.ev[0], c->c2.ev[1]); stats->ms_features = ms;
			(void)hipEventElapsedTime(&ms, c->c2.ev[1], c->c2.ev[2]); stats->ms_modes = ms;
			(void)hipEventElapsedTime(&ms, c->c2.ev[3], c->c2.ev[4]); stats->ms_coder = ms;
			stats->ms_emit = stats->ms_code - stats->ms_features - stats->ms_modes - stats->ms_coder;
			stats->coded_symbols = c->c2.last.symbols;
			stats->mode_iters = c->c2.last.mode_iters;
			stats->chain_symbols = c->c2.last.chain_symbols;
			stats->coder_launches = 1;
		}
		if (pipe || sliced) { /* overlapped stages: each from its own events (their sum exceeds ms_total) */
			stats->ms_parse = ps.ms_parse; stats->ms_features = ps.ms_features; stats->ms_modes = ps.ms_modes; stats->ms_coder = ps.ms_coder;
			stats->ms_emit = 0; stats->mode_iters = ps.mode_iters; stats->coded_symbols = c->c2.last.symbols;
			stats->chain_symbols = c->c2.last.chain_symbols; stats->pipelined = sliced ? 2 : 1; stats->coder_launches = ps.launches;
		}
		(void)fell_back;
	}
	return rc;
}

/* Workspace is ~150 bytes per input byte (+ ~32 per PADDED byte in K1), so very large batches are coded as consecutive sub-batches:
 * at most `batch_bytes` input bytes (X3H_BATCH_BYTES, default 512 MiB ~ 190 GB of HBM) and at most `batch_pad_bytes` bytes of padded
 * layout (every chunk occupies len + W + X3_PAD_EXTRA there, so many small chunks under a large window are bounded by THIS: K1 indexes
 * the padded layout with 32 bits).  Chunks of a sub-batch still run concurrently, streams are independent, so the output is
 * identical to the unsplit run.  A sub-batch that does not fit its output capacity does not stop the others: every out_lens entry is
 * written (0 for chunks that were never coded after a hard error) and the first error is returned after the loop. */
static void stats_add(x3h_stats &acc, const x3h_stats &part)
{
	for (int e = 0; e < 5; e++) acc.events[e] += part.events[e];
	acc.dict_elems += part.dict_elems; acc.ctx0_entries += part.ctx0_entries; acc.steps += part.steps; acc.coded_symbols += part.coded_symbols;
	acc.ms_total += part.ms_total; acc.ms_scan += part.ms_scan; acc.ms_parse += part.ms_parse; acc.ms_code += part.ms_code; acc.ms_copy += part.ms_copy;
	acc.ms_features += part.ms_features; acc.ms_modes += part.ms_modes; acc.ms_coder += part.ms_coder; acc.ms_emit += part.ms_emit;
	acc.mode_iters += part.mode_iters; acc.chain_symbols += part.chain_symbols; acc.pipelined |= part.pipelined;
	for (int e = 0; e < 4; e++) acc.est_bits[e] += part.est_bits[e];
	acc.coder_launches += part.coder_launches;
}

static int run(x3h_ctx *c, const x3h_params *prm_in, const RunIO &io, Stage upto, x3h_stats *stats)
{
	if (!c || !io.offsets || io.nchunks <= 0) return X3H_E_ARG;
	const uint64_t limit = c->batch_bytes, pad_limit = c->batch_pad_bytes;
	const uint64_t window = prm_in ? prm_in->window_bytes : 8 * 1024;
	auto padded = [&](int i) { return io.offsets[i + 1] < io.offsets[i] ? (uint64_t)0 : align_up(io.offsets[i + 1] - io.offsets[i] + window + X3_PAD_EXTRA, 256); };
	uint64_t pad_total = 0;
	for (int i = 0; i < io.nchunks; i++) pad_total += padded(i);
	const uint64_t total = io.offsets[io.nchunks] - io.offsets[0];
	if (upto != STAGE_CODE || (total <= limit && pad_total <= pad_limit) || io.nchunks == 1) return run_one(c, prm_in, io, upto, stats);
	x3h_stats acc, part;
	memset(&acc, 0, sizeof acc);
	int first = 0, rc = X3H_OK;
	while (first < io.nchunks) {
		int last = first + 1;
		uint64_t pad = padded(first);
		while (last < io.nchunks && io.offsets[last + 1] - io.offsets[first] <= limit && pad + padded(last) <= pad_limit) { pad += padded(last); last++; }
		RunIO sub = io;
		sub.offsets = io.offsets + first;
		sub.nchunks = last - first;
		sub.dst = io.dst ? io.dst + (uint64_t)first * io.dst_stride : nullptr;
		sub.out_lens = io.out_lens ? io.out_lens + first : nullptr;
		memset(&part, 0, sizeof part);
		const int r = run_one(c, prm_in, sub, upto, &part);
		stats_add(acc, part);
		if (r != X3H_OK && rc == X3H_OK) rc = r;
		if (r != X3H_OK && r != X3H_E_OUTPUT_FULL) { /* hard error: the remaining chunks are not attempted */
			if (io.out_lens) for (int i = first; i < io.nchunks; i++) io.out_lens[i] = 0;
			break;
		}
		first = last;
	}
	if (stats) *stats = acc;
	return rc;
}

/* ------------------------------------------------------------------------------------------------------------ */
extern "C" int x3h_compress(x3h_ctx *ctx, const x3h_params *prm, const uint8_t *in, size_t n,
                            uint8_t *out, size_t cap, size_t *out_len, x3h_stats *stats)
{
	if (!ctx || !out || !out_len || (!in && n) || cap < 4) return X3H_E_ARG;
	uint64_t off[2] = { 0, n }, len = 0;
	RunIO io = { in, false, off, 1, out, false, cap, &len, false };
	int rc = run(ctx, prm, io, STAGE_CODE, stats);
	*out_len = (size_t)len;
	return rc;
}

extern "C" int x3h_compress_chunks(x3h_ctx *ctx, const x3h_params *prm, const uint8_t *in, const uint64_t *offsets,
                                   int nchunks, uint8_t *out, uint64_t out_stride, uint64_t *out_lens, x3h_stats *stats)
{
	if (!ctx || !out || !out_lens || out_stride < 4) return X3H_E_ARG;
	RunIO io = { in, false, offsets, nchunks, out, false, out_stride, out_lens, false };
	return run(ctx, prm, io, STAGE_CODE, stats);
}

extern "C" int x3h_compress_chunks_dev(x3h_ctx *ctx, const x3h_params *prm, const void *d_in, const uint64_t *offsets,
                                       int nchunks, void *d_out, uint64_t out_stride, uint64_t *out_lens, x3h_stats *stats)
{
	if (!ctx || !d_out || !out_lens || out_stride < 4 || (out_stride & 3)) return X3H_E_ARG;
	RunIO io = { (const uint8_t *)d_in, true, offsets, nchunks, (uint8_t *)d_out, true, out_stride, out_lens, false };
	return run(ctx, prm, io, STAGE_CODE, stats);
}

extern "C" int x3h_scan_m(x3h_ctx *ctx, const x3h_params *prm, const uint8_t *in, size_t n, uint8_t *m_out)
{
	if (!ctx || (!in && n) || (!m_out && n)) return X3H_E_ARG;
	uint64_t off[2] = { 0, n };
	RunIO io = { in, false, off, 1, nullptr, false, 0, nullptr, false };
	CHK(run(ctx, prm, io, STAGE_SCAN, nullptr));
	if (n) {
		HIPCHK(hipMemcpy(m_out, ctx->m.as<uint8_t>() + ctx->hchunks[0].byte_off, n, hipMemcpyDeviceToHost));
	}
	return X3H_OK;
}

extern "C" int x3h_coder_chain(x3h_ctx *ctx, const uint32_t *cum, const uint32_t *freq, const uint32_t *total, size_t n,
                               uint32_t *states_out, uint32_t *final_lo)
{
	if (!ctx || !cum || !freq || !total || !n || !states_out || !final_lo) return X3H_E_ARG;
	HIPCHK(hipSetDevice(ctx->device));
	return x3_coder_chain_run(ctx->c2, ctx->stream, cum, freq, total, n, states_out, final_lo);
}

extern "C" int x3h_scan_counts(x3h_ctx *ctx, const x3h_params *prm, const uint8_t *in, size_t n, uint32_t *counts_out)
{
	if (!ctx || (!in && n) || (!counts_out && n)) return X3H_E_ARG;
	uint64_t off[2] = { 0, n };
	RunIO io = { in, false, off, 1, nullptr, false, 0, nullptr, true };
	CHK(run(ctx, prm, io, STAGE_SCAN, nullptr));
	if (n) {
		HIPCHK(hipMemcpy(counts_out, ctx->counts.p, n * 32 * sizeof(uint32_t), hipMemcpyDeviceToHost));
	}
	return X3H_OK;
}

extern "C" int x3h_parse(x3h_ctx *ctx, const x3h_params *prm, const uint8_t *in, size_t n,
                         uint32_t *tok_pos, uint32_t *tok_info, size_t tok_cap, size_t *ntok, uint64_t *dict_elems)
{
	if (!ctx || (!in && n) || !ntok) return X3H_E_ARG;
	uint64_t off[2] = { 0, n };
	RunIO io = { in, false, off, 1, nullptr, false, 0, nullptr, false };
	CHK(run(ctx, prm, io, STAGE_PARSE, nullptr));
	const X3ParseResult &r = ctx->hparse[0];
	*ntok = r.ntok;
	if (dict_elems) *dict_elems = r.dict_elems;
	size_t k = r.ntok < tok_cap ? r.ntok : tok_cap;
	if (k && tok_pos) HIPCHK(hipMemcpy(tok_pos, ctx->tok_pos.as<uint32_t>() + ctx->hchunks[0].elem_off, k * 4, hipMemcpyDeviceToHost));
	if (k && tok_info) HIPCHK(hipMemcpy(tok_info, ctx->tok_info.as<uint32_t>() + ctx->hchunks[0].elem_off, k * 4, hipMemcpyDeviceToHost));
	return X3H_OK;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* dev: `in` / `out` are device pointers on c's GPU -- the kernel reads the streams and writes the bytes where they are, no copy at all
 * (stream c at in + in_offsets[c], a multiple of 4 from a 4-byte aligned base; its bytes at out + out_offsets[c]) */
static int decompress_batch(x3h_ctx *c, const uint8_t *in, const uint64_t *in_offsets, int nchunks,
                            uint8_t *out, const uint64_t *out_offsets, uint64_t *out_lens, x3h_stats *stats, bool dev = false)
{
	if (!c || !in_offsets || !out_offsets || !out_lens || nchunks <= 0 || (!in && in_offsets[nchunks] != in_offsets[0])) return X3H_E_ARG;
	HIPCHK(hipSetDevice(c->device));
	const int nc = nchunks;
	std::vector<X3DecChunk> dk((size_t)nc);
	uint64_t ioff = 0, ooff = 0, toff = 0, itoff = 0, hoff = 0, koff = 0, loff = 0, max_tiles = 0;
	for (int i = 0; i < nc; i++) {
		if (in_offsets[i + 1] < in_offsets[i] || out_offsets[i + 1] < out_offsets[i]) return X3H_E_ARG;
		const uint64_t ilen = in_offsets[i + 1] - in_offsets[i], cap = out_offsets[i + 1] - out_offsets[i];
		if (ilen > 0xFFFFFFF0ull || cap > X3H_MAX_CHUNK) return X3H_E_ARG;
		X3DecChunk &k = dk[(size_t)i];
		k.in_off = dev ? in_offsets[i] : ioff; k.in_len = (uint32_t)ilen; k.out_cap = (uint32_t)cap; k.out_off = dev ? out_offsets[i] : ooff;
		if (dev && (in_offsets[i] & 3)) return X3H_E_ARG;
		/* the context pool in 8-byte units, worst case per parse step: an item more in a context0 list (<= 5 units with the blocks it outgrew), a new pair (3),
		 * an item more in a context1 list (<= 11); a new element costs 7.  Every block is read 64 entries at a time: two of those as slack. */
		k.tag_off = toff; k.item_off = itoff; k.item_cap = 24 * cap + 512; /* (pool units of 8 bytes per step, worst case: an item appended to the context1 list, <= 8 with doubling and the blocks left behind, and a
		                                  * new pair's context0 block of 1 + X3_DEC_CAP0 = 9 -- or an item appended to a context0 list, <= 5.2) */
		k.ht_log2 = ceil_log2(2 * (cap + 1)); if (k.ht_log2 < 4) k.ht_log2 = 4; k.ht_off = hoff;
		k.tok_off = koff; k.lit_off = loff; k._pad = 0;
		ioff += align_up(ilen, 16) + 16; ooff += align_up(cap, 256) + 256;
		toff += cap + 8; itoff += k.item_cap; hoff += (uint64_t)1 << k.ht_log2; koff += cap + 8; loff += align_up(cap + 64, 64);
		max_tiles += cap / X3_DEC_TILE + 1;
	}
	if (dev && (((uintptr_t)in & 3) || (!out && out_offsets[nc] != out_offsets[0]))) return X3H_E_ARG;
	if (itoff > 0xFFFFFFFF00ull || max_tiles > 0x7FFFFFFFull) return X3H_E_ARG;
	if (!dev) { CHK(c->din.reserve(ioff + 64)); CHK(c->out.reserve(ooff + 256)); }
	CHK(c->dchunks.reserve((size_t)nc * sizeof(X3DecChunk)));
	CHK(c->cresult.reserve((size_t)nc * sizeof(X3CodeResult)));
	CHK(c->dict_pos.reserve(toff * 4)); CHK(c->dict_len.reserve(toff));
	CHK(c->mtf.reserve(toff * 4)); CHK(c->idxfreq.reserve(toff * 4)); CHK(c->dc1.reserve(toff * 4));
	CHK(c->items.reserve(itoff * 8)); CHK(c->ht.reserve(hoff * 4));
	CHK(c->dtok.reserve(koff * 4)); CHK(c->dlit.reserve(loff + 64));
	CHK(c->dtile.reserve((max_tiles + 1) * 4)); CHK(c->dtfirst.reserve(((size_t)nc + 1) * 4));
	HIPCHK(hipEventRecord(c->ev[0], c->stream));
	for (int i = 0; i < nc && !dev; i++)
		if (dk[(size_t)i].in_len)
			HIPCHK(hipMemcpyAsync(c->din.as<uint8_t>() + dk[(size_t)i].in_off, in + in_offsets[i], dk[(size_t)i].in_len, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(c->dchunks.p, dk.data(), (size_t)nc * sizeof(X3DecChunk), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemsetAsync(c->ht.p, 0, hoff * 4, c->stream));
	X3DecArgs da;
	da.in = dev ? in : c->din.as<uint8_t>(); da.chunks = c->dchunks.as<X3DecChunk>(); da.out = dev ? out : c->out.as<uint8_t>();
	da.dict_pos = c->dict_pos.as<uint32_t>(); da.dict_len = c->dict_len.as<uint8_t>(); da.ht = c->ht.as<uint32_t>();
	da.mtf = c->mtf.as<uint32_t>(); da.idxfreq = c->idxfreq.as<uint32_t>(); da.c1off = c->dc1.as<uint32_t>();
	da.pool = c->items.as<uint64_t>(); da.tokens = c->dtok.as<uint32_t>(); da.lit = c->dlit.as<uint8_t>();
	da.tile_sum = c->dtile.as<uint32_t>(); da.tile_first = c->dtfirst.as<uint32_t>(); da.nchunks = (uint32_t)nc; da._pad = 0;
	da.result = c->cresult.as<X3CodeResult>();
	/* stage 1: the chains -- one tag per parse step */
	HIPCHK(hipEventRecord(c->ev[4], c->stream));
	x3k_launch_decode(&da, (uint32_t)nc, c->stream);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(c->ev[5], c->stream));
	c->hcode.resize((size_t)nc);
	HIPCHK(hipMemcpyAsync(c->hcode.data(), c->cresult.p, (size_t)nc * sizeof(X3CodeResult), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	/* stage 2: tags -> bytes, one workgroup per X3_DEC_TILE tokens */
	c->htfirst.resize((size_t)nc + 1);
	uint64_t ntiles = 0;
	for (int i = 0; i < nc; i++) {
		const X3CodeResult &r = c->hcode[(size_t)i];
		c->htfirst[(size_t)i] = (uint32_t)ntiles;
		if (r.status == X3_ST_OK) ntiles += ((uint64_t)r.events[7] + X3_DEC_TILE - 1) / X3_DEC_TILE;
	}
	c->htfirst[(size_t)nc] = (uint32_t)ntiles;
	if (ntiles > max_tiles) return X3H_E_INTERNAL;
	HIPCHK(hipMemcpyAsync(c->dtfirst.p, c->htfirst.data(), ((size_t)nc + 1) * 4, hipMemcpyHostToDevice, c->stream));
	x3k_launch_decode_bytes(&da, (uint32_t)nc, (uint32_t)ntiles, c->stream);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(c->ev[1], c->stream));
	HIPCHK(hipMemcpyAsync(c->hcode.data(), c->cresult.p, (size_t)nc * sizeof(X3CodeResult), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	if (getenv("X3H_DEBUG")) { const X3CodeResult &r = c->hcode[0];
		fprintf(stderr, "[x3h] decode stream 0: out %u tokens %u D %u pairs %u events %u %u %u %u status %u | kcycles (profile builds): waiting for blocks %u, request to use %u, use to request %u\n",
		        r.out_len, r.events[7], r._r, r.pairs, r.events[0], r.events[1], r.events[2], r.events[3], r.status, r.events[4], r.events[5], r.events[6]); }
	int rc = X3H_OK;
	for (int i = 0; i < nc; i++) {
		const X3CodeResult &r = c->hcode[(size_t)i];
		out_lens[i] = r.out_len;
		if (r.status == X3_ST_CORRUPT) rc = X3H_E_CORRUPT;
		else if (r.status == X3_ST_OUT_FULL && rc == X3H_OK) rc = X3H_E_OUTPUT_FULL;
		else if (r.status != X3_ST_OK && rc == X3H_OK) rc = X3H_E_INTERNAL;
	}
	if (rc == X3H_OK && !dev) {
		for (int i = 0; i < nc; i++)
			if (c->hcode[(size_t)i].out_len)
				HIPCHK(hipMemcpyAsync(out + out_offsets[i], c->out.as<uint8_t>() + dk[(size_t)i].out_off, c->hcode[(size_t)i].out_len, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	if (stats) {
		memset(stats, 0, sizeof(*stats));
		for (int i = 0; i < nc; i++) {
			for (int e = 0; e < 4; e++) { stats->events[e] += c->hcode[(size_t)i].events[e]; stats->steps += c->hcode[(size_t)i].events[e]; }
			stats->dict_elems += c->hcode[(size_t)i]._r;
			stats->ctx0_entries += c->hcode[(size_t)i].pairs;
		}
		float ms = 0;
		(void)hipEventElapsedTime(&ms, c->ev[4], c->ev[5]); stats->ms_code = ms;  /* stage 1: the chains */
		(void)hipEventElapsedTime(&ms, c->ev[5], c->ev[1]); stats->ms_emit = ms;  /* stage 2: tags -> bytes (the read-back of the token counts included) */
		(void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]); stats->ms_total = ms;
	}
	return rc;
}

/* The decoder's tables are sized from the output capacity (~200 B per byte: context pool 160, token trace 4, hash table 8-16, element tables 17, element bytes 1), so a batch whose capacities add up to more than
 * `dec_batch_bytes` (X3H_DEC_BATCH_BYTES, default 512 MiB) is decoded as consecutive sub-batches; streams are independent. */
static int decompress_chunks(x3h_ctx *c, const uint8_t *in, const uint64_t *in_offsets, int nchunks,
                             uint8_t *out, const uint64_t *out_offsets, uint64_t *out_lens, x3h_stats *stats, bool dev)
{
	if (!c || !in_offsets || !out_offsets || !out_lens || nchunks <= 0) return X3H_E_ARG;
	const uint64_t limit = c->dec_batch_bytes;
	if (out_offsets[nchunks] < out_offsets[0]) return X3H_E_ARG;
	if (nchunks == 1 || out_offsets[nchunks] - out_offsets[0] <= limit) return decompress_batch(c, in, in_offsets, nchunks, out, out_offsets, out_lens, stats, dev);
	x3h_stats acc, part;
	memset(&acc, 0, sizeof acc);
	int first = 0, rc = X3H_OK;
	while (first < nchunks) {
		int last = first + 1;
		while (last < nchunks && out_offsets[last + 1] >= out_offsets[first] && out_offsets[last + 1] - out_offsets[first] <= limit) last++;
		memset(&part, 0, sizeof part);
		const int r = decompress_batch(c, in, in_offsets + first, last - first, out, out_offsets + first, out_lens + first, &part, dev);
		stats_add(acc, part);
		if (r != X3H_OK && rc == X3H_OK) rc = r;
		if (r != X3H_OK && r != X3H_E_OUTPUT_FULL && r != X3H_E_CORRUPT) { for (int i = first; i < nchunks; i++) out_lens[i] = 0; break; }
		first = last;
	}
	if (stats) *stats = acc;
	return rc;
}

extern "C" int x3h_decompress_chunks(x3h_ctx *c, const uint8_t *in, const uint64_t *in_offsets, int nchunks,
                                     uint8_t *out, const uint64_t *out_offsets, uint64_t *out_lens, x3h_stats *stats)
{
	return decompress_chunks(c, in, in_offsets, nchunks, out, out_offsets, out_lens, stats, false);
}
extern "C" int x3h_decompress_chunks_dev(x3h_ctx *c, const void *d_in, const uint64_t *in_offsets, int nchunks,
                                         void *d_out, const uint64_t *out_offsets, uint64_t *out_lens, x3h_stats *stats)
{
	return decompress_chunks(c, (const uint8_t *)d_in, in_offsets, nchunks, (uint8_t *)d_out, out_offsets, out_lens, stats, true);
}

extern "C" int x3h_decompress(x3h_ctx *ctx, const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len, x3h_stats *stats)
{
	if (!ctx || !out_len || (!in && n) || (!out && cap)) return X3H_E_ARG;
	uint64_t io[2] = { 0, n }, oo[2] = { 0, cap }, len = 0;
	int rc = x3h_decompress_chunks(ctx, in, io, 1, out, oo, &len, stats);
	*out_len = (size_t)len;
	return rc;
}

/* ------------------------------------------------------------------------------------------------------------
 * Independent chunks over several devices (SURVEY.md 8(b) batch form / 8(e)): contiguous blocks of chunks per handle, one host
 * thread per handle, no exchange between devices.  Then the X3C1 container on top (x3_container.c has the format).
 * ------------------------------------------------------------------------------------------------------------ */
#include <thread>

static inline void shard(int n, int parts, int d, int *lo, int *hi) /* == dist.shard_range: config 4 puts chunk c on GPU c / (n / parts) */
{
	const int base = n / parts, rem = n % parts;
	*lo = d * base + (d < rem ? d : rem);
	*hi = *lo + base + (d < rem ? 1 : 0);
}

static void stats_merge_devices(x3h_stats *dst, const std::vector<x3h_stats> &parts)
{
	memset(dst, 0, sizeof *dst);
	for (const x3h_stats &p : parts) {
		x3h_stats t = *dst;
		stats_add(*dst, p);
		/* devices run side by side: times are those of the slowest one, not sums */
		if (p.ms_total < t.ms_total) { dst->ms_total = t.ms_total; dst->ms_scan = t.ms_scan; dst->ms_parse = t.ms_parse; dst->ms_code = t.ms_code; dst->ms_copy = t.ms_copy;
			dst->ms_features = t.ms_features; dst->ms_modes = t.ms_modes; dst->ms_coder = t.ms_coder; dst->ms_emit = t.ms_emit; }
		else { dst->ms_total = p.ms_total; dst->ms_scan = p.ms_scan; dst->ms_parse = p.ms_parse; dst->ms_code = p.ms_code; dst->ms_copy = p.ms_copy;
			dst->ms_features = p.ms_features; dst->ms_modes = p.ms_modes; dst->ms_coder = p.ms_coder; dst->ms_emit = p.ms_emit; }
	}
}

template <class F> static int run_on_devices(int ndevices, int nchunks, x3h_stats *stats, F call)
{
	const int nd = ndevices < nchunks ? ndevices : nchunks;
	std::vector<int> rc((size_t)nd, X3H_OK), hip((size_t)nd, 0);
	std::vector<x3h_stats> st((size_t)nd);
	auto work = [&](int d) {
		int lo, hi;
		shard(nchunks, nd, d, &lo, &hi);
		memset(&st[(size_t)d], 0, sizeof(x3h_stats));
		rc[(size_t)d] = call(d, lo, hi, &st[(size_t)d]);
		hip[(size_t)d] = x3_last_hip; /* thread-local in the worker */
	};
	const char *ser = getenv("X3H_MULTI_SERIAL"); /* debugging aid: one device after the other on the calling thread */
	if (nd == 1 || (ser && *ser && *ser != '0')) { for (int d = 0; d < nd; d++) work(d); }
	else {
		std::vector<std::thread> th;
		for (int d = 0; d < nd; d++) th.emplace_back(work, d);
		for (std::thread &t : th) t.join();
	}
	if (stats) stats_merge_devices(stats, st);
	for (int d = 0; d < nd; d++) if (rc[(size_t)d] != X3H_OK) { x3_last_hip = hip[(size_t)d]; return rc[(size_t)d]; }
	return X3H_OK;
}

extern "C" int x3h_compress_chunks_multi(x3h_ctx *const *ctxs, int ndevices, const x3h_params *prm, const uint8_t *in, const uint64_t *offsets,
                                         int nchunks, uint8_t *out, uint64_t out_stride, uint64_t *out_lens, x3h_stats *stats)
{
	if (!ctxs || ndevices <= 0 || !offsets || nchunks <= 0 || !out || !out_lens || out_stride < 4) return X3H_E_ARG;
	for (int d = 0; d < ndevices; d++) if (!ctxs[d]) return X3H_E_ARG;
	return run_on_devices(ndevices, nchunks, stats, [&](int d, int lo, int hi, x3h_stats *st) {
		return x3h_compress_chunks(ctxs[d], prm, in, offsets + lo, hi - lo, out + (uint64_t)lo * out_stride, out_stride, out_lens + lo, st);
	});
}

extern "C" int x3h_decompress_chunks_multi(x3h_ctx *const *ctxs, int ndevices, const uint8_t *in, const uint64_t *in_offsets, int nchunks,
                                           uint8_t *out, const uint64_t *out_offsets, uint64_t *out_lens, x3h_stats *stats)
{
	if (!ctxs || ndevices <= 0 || !in_offsets || !out_offsets || !out_lens || nchunks <= 0) return X3H_E_ARG;
	for (int d = 0; d < ndevices; d++) if (!ctxs[d]) return X3H_E_ARG;
	return run_on_devices(ndevices, nchunks, stats, [&](int d, int lo, int hi, x3h_stats *st) {
		return x3h_decompress_chunks(ctxs[d], in, in_offsets + lo, hi - lo, out, out_offsets + lo, out_lens + lo, st);
	});
}

extern "C" int x3h_compress_container(x3h_ctx *const *ctxs, int ndevices, const x3h_params *prm_in, const uint8_t *in, size_t n, size_t chunk_bytes,
                                      uint8_t *out, size_t cap, size_t *out_len, x3h_stats *stats)
{
	if (!ctxs || ndevices <= 0 || !ctxs[0] || !out || !out_len || (!in && n) || cap < 4) return X3H_E_ARG;
	*out_len = 0;
	if (!chunk_bytes || chunk_bytes > X3H_MAX_CHUNK) chunk_bytes = n <= X3H_MAX_CHUNK ? (n ? n : 1) : X3H_MAX_CHUNK;
	const uint64_t nch64 = n ? ((uint64_t)n + chunk_bytes - 1) / chunk_bytes : 1;
	if (nch64 > 0x7FFFFFF0ull) return X3H_E_ARG;
	const int nch = (int)nch64;
	if (nch == 1) return x3h_compress(ctxs[0], prm_in, in, n, out, cap, out_len, stats); /* one chunk: the raw stream of x3.c:603-611, no frame */
	x3h_params prm;
	if (prm_in) prm = *prm_in; else x3h_default_params(&prm);
	const size_t head = x3h_container_header_bytes(nch);
	if (cap < head + 4 * (size_t)nch) return X3H_E_OUTPUT_FULL;
	std::vector<uint64_t> off((size_t)nch + 1), raw((size_t)nch), lens((size_t)nch, 0);
	for (int i = 0; i <= nch; i++) { const uint64_t o = (uint64_t)i * chunk_bytes; off[(size_t)i] = o < n ? o : n; }
	for (int i = 0; i < nch; i++) raw[(size_t)i] = off[(size_t)i + 1] - off[(size_t)i];
	/* streams land in a strided staging buffer first: a stride that holds any sane stream (incompressible data grows by a few percent),
	 * then the provable bound if some chunk did not fit */
	uint64_t stride = ((uint64_t)chunk_bytes + chunk_bytes / 4 + 4096 + 3) & ~(uint64_t)3;
	const uint64_t bound = ((uint64_t)x3h_compress_bound(chunk_bytes) + 3) & ~(uint64_t)3;
	if (stride > bound) stride = bound;
	for (;;) {
		uint8_t *stage = (uint8_t *)malloc((size_t)(stride * (uint64_t)nch));
		if (!stage) return X3H_E_NOMEM;
		int rc = x3h_compress_chunks_multi(ctxs, ndevices, &prm, in, off.data(), nch, stage, stride, lens.data(), stats);
		if (rc == X3H_E_OUTPUT_FULL && stride < bound) { free(stage); stride = bound; continue; }
		if (rc == X3H_OK) {
			uint64_t total = head;
			for (int i = 0; i < nch; i++) total += lens[(size_t)i];
			if (total > cap) rc = X3H_E_OUTPUT_FULL;
			else {
				rc = x3h_container_write_header(out, cap, &prm, nch, raw.data(), lens.data());
				uint64_t o = head;
				for (int i = 0; i < nch && rc == X3H_OK; i++) { memcpy(out + o, stage + (uint64_t)i * stride, (size_t)lens[(size_t)i]); o += lens[(size_t)i]; }
				if (rc == X3H_OK) *out_len = (size_t)total;
			}
		}
		free(stage);
		return rc;
	}
}

extern "C" int x3h_decompress_container(x3h_ctx *const *ctxs, int ndevices, const uint8_t *in, size_t n,
                                        uint8_t *out, size_t cap, size_t *out_len, x3h_stats *stats)
{
	if (!ctxs || ndevices <= 0 || !ctxs[0] || !out_len || (!in && n) || (!out && cap)) return X3H_E_ARG;
	*out_len = 0;
	int nch = 0;
	uint64_t raw_total = 0;
	const int pr = x3h_container_probe(in, n, nullptr, &nch, &raw_total);
	if (pr == X3H_NOT_A_CONTAINER) return x3h_decompress(ctxs[0], in, n, out, cap, out_len, stats);
	if (pr != X3H_OK) return pr;
	if (raw_total > cap) return X3H_E_OUTPUT_FULL;
	std::vector<uint64_t> raw((size_t)nch), coff((size_t)nch + 1), ooff((size_t)nch + 1, 0), lens((size_t)nch, 0);
	CHK(x3h_container_table(in, n, raw.data(), coff.data()));
	for (int i = 0; i < nch; i++) ooff[(size_t)i + 1] = ooff[(size_t)i] + raw[(size_t)i];
	int rc = x3h_decompress_chunks_multi(ctxs, ndevices, in, coff.data(), nch, out, ooff.data(), lens.data(), stats);
	if (rc == X3H_E_OUTPUT_FULL) rc = X3H_E_CORRUPT; /* a chunk decodes to more than its table entry says */
	if (rc != X3H_OK) return rc;
	for (int i = 0; i < nch; i++) if (lens[(size_t)i] != raw[(size_t)i]) return X3H_E_CORRUPT;
	*out_len = (size_t)raw_total;
	return X3H_OK;
}

/* ------------------------------------------------------------------------------------------------------------
 * The final bitstream concat of a multi-GPU batch as ONE RCCL exchange, behind the C boundary (BASELINE north star: "independent input
 * chunks partition across the GPUs of one node with a single RCCL gather over xGMI for the final bitstream concat").  Every device
 * codes its contiguous block of chunks with inputs and streams resident in ITS HBM, packs its streams back to back, and one
 * ncclGroupStart .. ncclSend / ncclRecv .. ncclGroupEnd brings every block to ctxs[0]'s GPU, directly behind the X3C1 header that
 * is laid down there; the finished container crosses PCIe once.  No host staging of the streams (x3h_compress_container goes through a
 * host buffer per device).  librccl is loaded on first use (dlopen): a process that never calls this never pays for it.
 * ------------------------------------------------------------------------------------------------------------ */
#include <mutex>
#ifndef X3_EMU
#include <dlfcn.h>
#else
#include "rccl_stub.h" /* tests/emu: sends and receives recorded inside the group, executed as memcpy at its end -- the code below runs with 2..8 emulated devices */
#endif

/* The six entry points of librccl this file uses, declared here (the library is dlopen'ed: building libx3hip.so needs no RCCL headers).
 * Values as in rccl.h: ncclSuccess == 0, ncclUint8 == 1. */
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
static const ncclResult_t ncclSuccess = 0;
static const ncclDataType_t ncclUint8 = 1;

namespace {
struct RcclApi {
	void *lib = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	std::vector<int> devs;          /* the communicators in hand were made for these devices, in this order */
	std::vector<ncclComm_t> comms;
	std::mutex mu;
};
RcclApi g_rccl;

#ifndef X3_EMU
bool rccl_load()
{
	if (g_rccl.lib) return true;
	void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
	if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
	if (!h) return false;
#define X3_SYM(field, name) *(void **)&g_rccl.field = dlsym(h, name); if (!g_rccl.field) { dlclose(h); return false; }
	X3_SYM(CommInitAll, "ncclCommInitAll") X3_SYM(CommDestroy, "ncclCommDestroy") X3_SYM(Send, "ncclSend") X3_SYM(Recv, "ncclRecv")
	X3_SYM(GroupStart, "ncclGroupStart") X3_SYM(GroupEnd, "ncclGroupEnd")
#undef X3_SYM
	g_rccl.lib = h;
	return true;
}
#else
bool rccl_load()
{
	g_rccl.lib = (void *)&g_rccl;
	g_rccl.CommInitAll = x3emu_rccl::CommInitAll; g_rccl.CommDestroy = x3emu_rccl::CommDestroy; g_rccl.Send = x3emu_rccl::Send; g_rccl.Recv = x3emu_rccl::Recv;
	g_rccl.GroupStart = x3emu_rccl::GroupStart; g_rccl.GroupEnd = x3emu_rccl::GroupEnd;
	return true;
}
#endif

void rccl_drop_comms()
{
	for (ncclComm_t c : g_rccl.comms) if (c) (void)g_rccl.CommDestroy(c);
	g_rccl.comms.clear(); g_rccl.devs.clear();
}
} // namespace

extern "C" void x3h_rccl_release(void)
{
	std::lock_guard<std::mutex> lk(g_rccl.mu);
	if (g_rccl.lib) rccl_drop_comms();
}

struct GatherDev { /* what one device contributes */
	int lo = 0, hi = 0;
	uint64_t payload = 0;           /* bytes of its packed streams */
	uint8_t *packed = nullptr;      /* where they are (device memory of that GPU) */
};

extern "C" int x3h_compress_container_rccl(x3h_ctx *const *ctxs, int ndevices, const x3h_params *prm_in, const uint8_t *in, size_t n, size_t chunk_bytes,
                                           uint8_t *out, size_t cap, size_t *out_len, x3h_stats *stats)
{
	if (!ctxs || ndevices <= 0 || !out || !out_len || (!in && n) || cap < 4) return X3H_E_ARG;
	for (int d = 0; d < ndevices; d++) {
		if (!ctxs[d]) return X3H_E_ARG;
		for (int e = 0; e < d; e++) if (ctxs[e]->device == ctxs[d]->device) return X3H_E_ARG; /* one rank per GPU */
	}
	*out_len = 0;
	if (!chunk_bytes || chunk_bytes > X3H_MAX_CHUNK) chunk_bytes = n <= X3H_MAX_CHUNK ? (n ? n : 1) : X3H_MAX_CHUNK;
	const uint64_t nch64 = n ? ((uint64_t)n + chunk_bytes - 1) / chunk_bytes : 1;
	if (nch64 > 0x7FFFFFF0ull) return X3H_E_ARG;
	const int nch = (int)nch64;
	if (nch == 1) return x3h_compress(ctxs[0], prm_in, in, n, out, cap, out_len, stats); /* one chunk: the raw stream, nothing to gather */
	x3h_params prm;
	if (prm_in) prm = *prm_in; else x3h_default_params(&prm);
	const int nd = ndevices < nch ? ndevices : nch;
	const size_t head = x3h_container_header_bytes(nch);
	if (cap < head + 4 * (size_t)nch) return X3H_E_OUTPUT_FULL;

	/* the mutex covers the communicators (lookup / creation here, the exchange below): independent callers still code their chunks side by side */
	{
		std::lock_guard<std::mutex> lk(g_rccl.mu);
		if (!rccl_load()) return X3H_E_RCCL;
		std::vector<int> devs((size_t)nd);
		for (int d = 0; d < nd; d++) devs[(size_t)d] = ctxs[d]->device;
		if (g_rccl.devs != devs) {
			rccl_drop_comms();
			g_rccl.comms.assign((size_t)nd, nullptr);
			if (g_rccl.CommInitAll(g_rccl.comms.data(), nd, devs.data()) != ncclSuccess) { g_rccl.comms.clear(); return X3H_E_RCCL; }
			g_rccl.devs = devs;
		}
	}

	std::vector<uint64_t> off((size_t)nch + 1), raw((size_t)nch), lens((size_t)nch, 0);
	for (int i = 0; i <= nch; i++) { const uint64_t o = (uint64_t)i * chunk_bytes; off[(size_t)i] = o < n ? o : n; }
	for (int i = 0; i < nch; i++) raw[(size_t)i] = off[(size_t)i + 1] - off[(size_t)i];
	uint64_t stride = ((uint64_t)chunk_bytes + chunk_bytes / 4 + 4096 + 3) & ~(uint64_t)3;
	const uint64_t bound = ((uint64_t)x3h_compress_bound(chunk_bytes) + 3) & ~(uint64_t)3;
	if (stride > bound) stride = bound;
	std::vector<GatherDev> gd((size_t)nd);
	for (;;) {
		/* every device: H2D of its block of the input, the three stages with streams left in its HBM, streams packed back to back */
		int rc = run_on_devices(nd, nch, stats, [&](int d, int lo, int hi, x3h_stats *st) -> int {
			x3h_ctx *c = ctxs[d];
			GatherDev &g = gd[(size_t)d];
			g.lo = lo; g.hi = hi; g.payload = 0; g.packed = nullptr;
			HIPCHK(hipSetDevice(c->device));
			const int k = hi - lo;
			const uint64_t bytes = off[(size_t)hi] - off[(size_t)lo];
			CHK(c->g_in.reserve(bytes + 16));
			CHK(c->g_out.reserve(stride * (uint64_t)k + 16));
			HIPCHK(hipMemcpyAsync(c->g_in.p, in + off[(size_t)lo], bytes, hipMemcpyHostToDevice, c->stream));
			std::vector<uint64_t> rel((size_t)k + 1);
			for (int i = 0; i <= k; i++) rel[(size_t)i] = off[(size_t)(lo + i)] - off[(size_t)lo];
			RunIO io = { c->g_in.as<uint8_t>(), true, rel.data(), k, c->g_out.as<uint8_t>(), true, stride, lens.data() + lo, false };
			CHK(run(c, &prm, io, STAGE_CODE, st));
			std::vector<uint64_t> po((size_t)k + 1, 0); /* packed offsets (x3 streams are whole 32-bit words) */
			for (int i = 0; i < k; i++) po[(size_t)i + 1] = po[(size_t)i] + lens[(size_t)(lo + i)];
			g.payload = po[(size_t)k];
			/* device 0's block is the first of the container: it is packed straight behind the header */
			const uint64_t lead = d == 0 ? head : 0;
			CHK(c->g_pack.reserve(lead + g.payload + 16));
			CHK(c->srcoff.reserve((size_t)(k + 1) * 8));
			HIPCHK(hipMemcpy(c->srcoff.p, po.data(), (size_t)(k + 1) * 8, hipMemcpyHostToDevice)); /* (synchronous: `po` is a local) */
			const uint64_t *dpo = c->srcoff.as<uint64_t>();
			const uint32_t *src = c->g_out.as<uint32_t>();
			uint32_t *dst = (uint32_t *)(c->g_pack.as<uint8_t>() + lead);
			const uint64_t wstride = stride / 4;
			uint64_t maxw = 0;
			for (int i = 0; i < k; i++) if (lens[(size_t)(lo + i)] / 4 > maxw) maxw = lens[(size_t)(lo + i)] / 4;
			x3_foreach((size_t)(maxw * (uint64_t)k), c->stream, X3_LAMBDA(size_t t) {
				const uint32_t ci = (uint32_t)(t / maxw);
				const uint64_t w = t % maxw, b0 = dpo[ci] / 4, nw = dpo[ci + 1] / 4 - b0;
				if (w < nw) dst[b0 + w] = src[(uint64_t)ci * wstride + w];
			});
			HIPCHK(hipGetLastError());
			HIPCHK(hipStreamSynchronize(c->stream));
			g.packed = c->g_pack.as<uint8_t>() + lead;
			return X3H_OK;
		});
		if (rc == X3H_E_OUTPUT_FULL && stride < bound) { stride = bound; continue; }
		if (rc != X3H_OK) return rc;
		break;
	}
	uint64_t total = head;
	for (int i = 0; i < nch; i++) total += lens[(size_t)i];
	if (total > cap) return X3H_E_OUTPUT_FULL;

	/* root: header + its own block are in place; the other blocks arrive behind them.  ndevices == 1: the block makes the round trip
	 * through RCCL all the same (send to self), so that this leg is exercised on a one-GPU machine. */
	x3h_ctx *root = ctxs[0];
	std::vector<uint8_t> hdr(head);
	CHK(x3h_container_write_header(hdr.data(), head, &prm, nch, raw.data(), lens.data()));
	/* the exchange and what follows: whatever fails in here, no stream of any device is left with work in flight when this function returns
	 * (the header travels by a synchronous copy: its source is a local) */
	auto exchange = [&]() -> int {
		HIPCHK(hipSetDevice(root->device));
		uint8_t *final_buf = root->g_pack.as<uint8_t>();
		if (nd == 1) { /* self-send: the packed block moves to a second buffer behind a copy of the header */
			CHK(root->g_final.reserve(total + 16));
			final_buf = root->g_final.as<uint8_t>();
		} else {
			uint64_t need = total + 16; /* g_pack of the root was sized for its own block only */
			if (root->g_pack.cap < need) {
				CHK(root->g_final.reserve(need));
				HIPCHK(hipMemcpyAsync(root->g_final.p, root->g_pack.p, head + gd[0].payload, hipMemcpyDeviceToDevice, root->stream));
				HIPCHK(hipStreamSynchronize(root->stream));
				final_buf = root->g_final.as<uint8_t>();
			}
		}
		HIPCHK(hipMemcpy(final_buf, hdr.data(), head, hipMemcpyHostToDevice));
		std::lock_guard<std::mutex> lk(g_rccl.mu);
		std::vector<int> devs((size_t)nd);
		for (int d = 0; d < nd; d++) devs[(size_t)d] = ctxs[d]->device;
		if (g_rccl.devs != devs || g_rccl.comms.size() != (size_t)nd) return X3H_E_RCCL; /* another caller replaced the communicators meanwhile */
		bool ok = g_rccl.GroupStart() == ncclSuccess;
		uint64_t at = head + (nd == 1 ? 0 : gd[0].payload);
		for (int d = (nd == 1 ? 0 : 1); d < nd && ok; d++) {
			const GatherDev &g = gd[(size_t)d];
			if (g.payload) {
				ok = ok && g_rccl.Send(g.packed, (size_t)g.payload, ncclUint8, 0, g_rccl.comms[(size_t)d], ctxs[d]->stream) == ncclSuccess;
				ok = ok && g_rccl.Recv(final_buf + at, (size_t)g.payload, ncclUint8, d, g_rccl.comms[0], root->stream) == ncclSuccess;
			}
			at += g.payload;
		}
		ok = (g_rccl.GroupEnd() == ncclSuccess) && ok;
		if (!ok) { rccl_drop_comms(); return X3H_E_RCCL; }
		for (int d = 1; d < nd; d++) { HIPCHK(hipSetDevice(ctxs[d]->device)); HIPCHK(hipStreamSynchronize(ctxs[d]->stream)); }
		HIPCHK(hipSetDevice(root->device));
		HIPCHK(hipMemcpyAsync(out, final_buf, (size_t)total, hipMemcpyDeviceToHost, root->stream)); /* the ONE transfer of the container over PCIe */
		HIPCHK(hipStreamSynchronize(root->stream));
		return X3H_OK;
	};
	const int xrc = exchange();
	if (xrc != X3H_OK) { /* drain every stream before the caller (or a fallback path) touches these handles again */
		const int keep = x3_last_hip;
		for (int d = 0; d < nd; d++) { (void)hipSetDevice(ctxs[d]->device); (void)hipStreamSynchronize(ctxs[d]->stream); }
		x3_last_hip = keep;
		return xrc;
	}
	*out_len = (size_t)total;
	return X3H_OK;
}
