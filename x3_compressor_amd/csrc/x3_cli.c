/*
 * x3_cli.c -- the x3 command line, host side in C, on top of the C ABI of include/x3hip.h.
 *
 * Same letters, defaults and file-name rules as the reference's main() (x3.c:479-548): -z -d -f -k -h -t N -w N(KiB)
 * -m N -n N -x, 0/1/2 positional arguments, "<in>.x3" default output, no clobber without -f.  Compressed output
 * is the raw x3 code stream (no header), bit-identical to the reference for the same input and parameters.
 * The statistics block on stderr follows x3.c:662-693 line by line, the float size estimates included (x3h_stats.est_bits: the same
 * single-precision accumulators in coding order).  Errors print a message and exit(1) instead of abort().
 *
 * Additive options (new, none of them changes what the reference's letters do):
 *   -g N             use GPU N (default 0)
 *   --gpus a,b,...   use these GPUs side by side (chunks are dealt out in contiguous blocks); the finished streams are
 *                    concatenated on the host (x3h_compress_container)
 *   --rccl           with --gpus: the streams stay in HBM and ONE RCCL send/receive group concatenates them on the first GPU
 *                    (x3h_compress_container_rccl; also X3_RCCL=1).  Same bytes; any failure of that path falls back to the host concat.
 *   --chunk-kib N    cut the input into independent chunks of N KiB, each coded as its own x3 stream, and write the X3C1
 *                    container (include/x3hip.h); an input of one chunk is still written as the raw stream.  Inputs above
 *                    2^28 - 4096 bytes (X3H_MAX_CHUNK) are always chunked -- the reference codes them as ONE stream (x3.c:577-611), so that
 *                    output is NOT readable by the reference's `x3 -d`: the CLI says so on stderr.
 *   --batch-mib N    chunks are coded in sub-batches of at most N MiB of input: bounds the workspace in HBM (~350 B per byte).  Default: 64 MiB,
 *                    but never fewer than 32 chunks side by side (a chunk is one serial coder chain: long chunks need company)
 * -d recognises a container by its magic and decodes the chunks as one batch; anything else is a raw x3 stream.
 */
#define _GNU_SOURCE /* getopt_long */
#include <execinfo.h>
#include <getopt.h>
#include <math.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "../../include/x3hip.h"

static void die(const char *msg) { fprintf(stderr, "%s\n", msg); exit(1); }

/* A fatal signal says where it struck (phase of main() + the faulting thread's stack) and is then delivered again with its default
 * action, so the exit status still names it.  The reference abort()s on every error (file.c:9-18); a drop-in must at least say where. */
static volatile sig_atomic_t g_phase; /* 1 options, 2 handles, 3 input read, 4 library call, 5 output written, 6 handles released, 7 leaving */
static void on_fatal(int sig)
{
	/* async-signal-safe only: no stdio; the handlers were installed with SA_RESETHAND, so a second fault in here terminates at once, and an
	 * alarm bounds a hang.  backtrace() was called once at start-up, so libgcc is loaded already (its first call would dlopen + malloc). */
	char msg[48] = "x3: fatal signal 00 in phase 0\n";
	msg[17] = (char)('0' + sig / 10 % 10); msg[18] = (char)('0' + sig % 10); msg[29] = (char)('0' + (int)g_phase % 10);
	alarm(5);
	if (write(2, msg, 31) < 0) { /* nothing left to do about it */ }
	void *bt[48];
	backtrace_symbols_fd(bt, backtrace(bt, 48), 2);
	raise(sig); /* default action now (SA_RESETHAND): the exit status still names the signal */
}

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

static void print_help(const char *path) /* x3.c:465-477 */
{
	fprintf(stderr, "Usage :\n\t%s [arguments] [input-file] [output-file]\n\n", path);
	fprintf(stderr, "Arguments :\n");
	fprintf(stderr, " -d     : force decompression\n");
	fprintf(stderr, " -z     : force compression\n");
	fprintf(stderr, " -f     : overwrite existing output file\n");
	fprintf(stderr, " -k     : keep (don't delete) input file (default)\n");
	fprintf(stderr, " -h     : print this message\n");
	fprintf(stderr, " -t NUM : maximum number of matches (affects compression ratio and speed)\n");
	fprintf(stderr, " -w NUM : window size (in kilobytes, affects compression ratio and speed)\n");
	fprintf(stderr, " -m NUM : magic factor (affects compression ratio and speed)\n");
	fprintf(stderr, " -g NUM : GPU to use (default 0)\n");
	fprintf(stderr, " --gpus A,B,...  : GPUs to use side by side\n");
	fprintf(stderr, " --chunk-kib NUM : code independent chunks of NUM KiB (X3C1 container output)\n");
	fprintf(stderr, " --batch-mib NUM : code at most NUM MiB of chunks at a time (default 64, at least 32 chunks; bounds the GPU workspace)\n");
}

static FILE *open_output(const char *path, int force) /* force_fopen, file.c:47-55 */
{
	if (!force && access(path, F_OK) != -1) die("File already exists");
	return fopen(path, "w");
}

static unsigned char *read_all(FILE *f, size_t *n) /* fsize + fload, file.c:7-45 */
{
	long begin = ftell(f);
	if (begin == -1L) die("Stream is not seekable");
	if (fseek(f, 0, SEEK_END)) die("seek failed");
	long end = ftell(f);
	if (end == -1L || fseek(f, begin, SEEK_SET)) die("seek failed");
	*n = (size_t)(end - begin);
	unsigned char *p = malloc(*n ? *n : 1);
	if (!p) die("out of memory");
	if (fread(p, 1, *n, f) < *n) die("short read");
	return p;
}

#define MAX_GPUS 64

/* The HIP runtime needs 150-250 ms to start and to make a stream (profiles/r03_cli_first_call.txt); reading the input needs 50-70 ms for
 * 256 MiB.  Neither needs the other: the handles are made on a thread while main() reads the file. */
struct handle_job { int ngpu; const int *gpus; x3h_ctx **ctxs; uint64_t batch_mib; int set_batch; int rc; int hip; };
static void *make_handles(void *arg)
{
	struct handle_job *j = arg;
	j->rc = X3H_OK;
	for (int i = 0; i < j->ngpu && j->rc == X3H_OK; i++) j->rc = x3h_ctx_create(&j->ctxs[i], j->gpus[i]);
	/* a command-line process lives for one call: keep its workspace small (the sub-batches of a big input follow each other, x3hip.h) */
	for (int i = 0; j->set_batch && i < j->ngpu && j->rc == X3H_OK; i++) j->rc = x3h_ctx_set_batch_bytes(j->ctxs[i], j->batch_mib << 20);
	for (int i = 0; i < j->ngpu && j->rc == X3H_OK; i++) j->rc = x3h_ctx_set_estimates(j->ctxs[i], 1); /* the statistics block prints them (x3.c:664-691) */
	j->hip = x3h_last_hip_error(); /* thread-local in the library: main() cannot ask for it later */
	return NULL;
}

int main(int argc, char *argv[])
{
	int decompress = 0, force = 0, o, ngpu = 1, gpus[MAX_GPUS] = { 0 }, use_rccl = 0;
	size_t chunk_bytes = 0;
	uint64_t batch_mib = 0; /* 0: by the chunk size (below) */
	x3h_params prm;
	const int fatal[] = { SIGSEGV, SIGBUS, SIGILL, SIGFPE, SIGABRT };
	{ void *warm[4]; (void)backtrace(warm, 4); } /* loads libgcc's unwinder now, not inside a signal handler */
	for (size_t i = 0; i < sizeof fatal / sizeof fatal[0]; i++) {
		struct sigaction sa;
		memset(&sa, 0, sizeof sa);
		sa.sa_handler = on_fatal; sa.sa_flags = (int)SA_RESETHAND | SA_NODEFER;
		sigemptyset(&sa.sa_mask);
		sigaction(fatal[i], &sa, NULL);
	}
	g_phase = 1;
	x3h_default_params(&prm);
	{ const char *e = getenv("X3_RCCL"); use_rccl = e && *e && *e != '0'; }
	/* a short-lived process: the three CU-masked streams a handle makes for mid-size batches (~30 ms of queue creation) would cost more than they can save here */
	(void)setenv("X3H_SLICE_CUMASK", "0", 0);
	static const struct option longopts[] = { { "chunk-kib", required_argument, NULL, 1000 }, { "gpus", required_argument, NULL, 1001 }, { "batch-mib", required_argument, NULL, 1002 },
	                                          { "rccl", no_argument, NULL, 1003 }, { NULL, 0, NULL, 0 } };

	while ((o = getopt_long(argc, argv, "zdfkht:w:m:n:xg:", longopts, NULL)) != -1) { /* x3.c:484 */
		switch (o) {
			case 'z': decompress = 0; break;
			case 'd': decompress = 1; break;
			case 'f': force = 1; break;
			case 'k': break;
			case 'h': print_help(argv[0]); return 0;
			case 't': prm.max_match_count = atoi(optarg); break;
			case 'w': prm.window_bytes = (uint32_t)atoi(optarg) * 1024u; break;
			case 'm': prm.factor1 = (uint32_t)atoi(optarg); break;
			case 'n': prm.factor2 = (uint32_t)atoi(optarg); break;
			case 'x': prm.nl_mode = 1; break;
			case 'g': ngpu = 1; gpus[0] = atoi(optarg); break;
			case 1000: {
				long k = atol(optarg);
				if (k <= 0 || (size_t)k * 1024 > X3H_MAX_CHUNK) die("--chunk-kib: between 1 and 262140");
				chunk_bytes = (size_t)k * 1024;
				break;
			}
			case 1001: {
				ngpu = 0;
				for (char *tok = strtok(optarg, ","); tok; tok = strtok(NULL, ",")) {
					if (ngpu == MAX_GPUS) die("--gpus: too many devices");
					gpus[ngpu++] = atoi(tok);
				}
				if (!ngpu) die("--gpus: empty list");
				break;
			}
			case 1002: {
				long k = atol(optarg);
				if (k < 1 || k > 4096) die("--batch-mib: between 1 and 4096");
				batch_mib = (uint64_t)k;
				break;
			}
			case 1003: use_rccl = 1; break;
			default: die("Unexpected argument");
		}
	}

	FILE *istream = NULL, *ostream = NULL;
	switch (argc - optind) { /* x3.c:522-548 */
		case 0: istream = stdin; ostream = stdout; break;
		case 1: {
			istream = fopen(argv[optind], "r");
			if (!decompress) {
				char path[4096 + 8];
				snprintf(path, sizeof path, "%s.x3", argv[optind]);
				ostream = open_output(path, force);
			} else {
				char *dot = strrchr(argv[optind], '.');
				if (dot) *dot = 0;
				ostream = open_output(argv[optind], force);
			}
			break;
		}
		case 2:
			istream = fopen(argv[optind], "r");
			ostream = open_output(argv[optind + 1], force);
			break;
		default: die("Unexpected argument");
	}
	fprintf(stderr, "%s\n", decompress ? "Decompressing..." : "Compressing...");
	if (!istream) die("Cannot open input file");
	if (!ostream) die("Cannot open output file");

	x3h_ctx *ctxs[MAX_GPUS] = { NULL };
	int rc = X3H_OK;
	g_phase = 2;
	const double t_start = now_ms();
	/* sub-batch of a short-lived process: small (the workspace is ~350 B per input byte, and memory an earlier process freed comes back slowly:
	 * profiles/r03_cli_first_call.txt), but a chunk is ONE serial coder chain, so long chunks need at least 32 of them side by side per GPU:
	 * 64 MiB for chunks up to 2 MiB, 32 chunks beyond (config 4's 8 MiB chunks: 256 MiB, all 16 of a GPU's share in one go) */
	if (!batch_mib) {
		const uint64_t cb = chunk_bytes ? chunk_bytes : X3H_MAX_CHUNK; /* no --chunk-kib: one stream, or chunks of the longest stream (X3H_MAX_CHUNK) for a larger input */
		batch_mib = 64;
		if (((32 * cb) >> 20) > batch_mib) batch_mib = (32 * cb) >> 20;
		if (batch_mib > 512) batch_mib = 512; /* the library's own default bound */
	}
	struct handle_job job = { ngpu, gpus, ctxs, batch_mib, getenv("X3H_BATCH_BYTES") == NULL, X3H_OK, 0 };
	pthread_t th;
	const int threaded = pthread_create(&th, NULL, make_handles, &job) == 0;
	if (!threaded) make_handles(&job);

	size_t isize = 0, osize = 0;
	g_phase = 3;
	unsigned char *iptr = read_all(istream, &isize), *optr = NULL;
	const double t_read = now_ms();
	if (threaded) pthread_join(th, NULL);
	rc = job.rc;
	if (rc != X3H_OK) { fprintf(stderr, "x3: %s (hip error %d; the hot path only exists as gfx950 HIP kernels; no CPU fallback)\n", x3h_strerror(rc), job.hip); return 1; }
	const double t_ctx = now_ms();
	g_phase = 4;
	x3h_stats st;
	memset(&st, 0, sizeof st);

	if (!decompress) {
		fprintf(stderr, "max match count: %i\n", prm.max_match_count);
		fprintf(stderr, "forward window: %zu\n", (size_t)prm.window_bytes);
		fprintf(stderr, "magic factor 1: %zu\n", (size_t)prm.factor1);
		fprintf(stderr, "magic factor 2: %zu\n", (size_t)prm.factor2);
		if (chunk_bytes) fprintf(stderr, "chunk size: %zu (independent x3 streams, X3C1 container)\n", chunk_bytes);
		else if (isize > X3H_MAX_CHUNK)
			fprintf(stderr, "x3: NOTE: the input is larger than %zu bytes (the longest single stream this build codes): writing an X3C1 container of %zu-byte chunks.  "
			        "The reference codes such an input as ONE stream (x3.c:577-611): its `x3 -d` cannot read this output; this `x3 -d` can.\n",
			        (size_t)X3H_MAX_CHUNK, (size_t)X3H_MAX_CHUNK);
		size_t cap = x3h_container_bound(isize, chunk_bytes);
		optr = malloc(cap);
		if (!optr) die("out of memory");
		/* several GPUs: host-staged concat by default; --rccl / X3_RCCL=1: the chunk streams stay in HBM and ONE RCCL exchange concatenates them
		 * on the first GPU (x3h_compress_container_rccl).  Same bytes; whatever that path fails with (except a too small output buffer, which
		 * the host path would hit as well) the host-staged concat takes over. */
		rc = X3H_E_RCCL;
		if (ngpu > 1 && use_rccl) {
			rc = x3h_compress_container_rccl(ctxs, ngpu, &prm, iptr, isize, chunk_bytes, optr, cap, &osize, &st);
			x3h_rccl_release();
			if (rc == X3H_OK) fprintf(stderr, "final concat: one RCCL gather over %d GPUs\n", ngpu);
			else if (rc != X3H_E_OUTPUT_FULL) { fprintf(stderr, "x3: RCCL gather failed (%s): falling back to the host-staged concat\n", x3h_strerror(rc)); rc = X3H_E_RCCL; }
		}
		if (rc == X3H_E_RCCL) rc = x3h_compress_container(ctxs, ngpu, &prm, iptr, isize, chunk_bytes, optr, cap, &osize, &st);
		if (rc != X3H_OK) { fprintf(stderr, "x3: compress failed: %s\n", x3h_strerror(rc)); return 1; }
		fprintf(stderr, "elapsed time: %f\n", st.ms_total / 1000.0);
		fprintf(stderr, "  device ms: scan %.3f parse %.3f code %.3f copy %.3f\n", st.ms_scan, st.ms_parse, st.ms_code, st.ms_copy);
	} else {
		int nch = 0;
		uint64_t raw_total = 0;
		rc = x3h_container_probe(iptr, isize, NULL, &nch, &raw_total);
		if (rc == X3H_OK) { /* X3C1: the table carries every chunk's size */
			optr = malloc(raw_total ? raw_total : 1);
			if (!optr) die("out of memory");
			rc = x3h_decompress_container(ctxs, ngpu, iptr, isize, optr, raw_total, &osize, &st);
		} else if (rc == X3H_NOT_A_CONTAINER) {
			/* a raw stream carries no length.  The reference assumes <= 64:1 (x3.c:621, unchecked); here the capacity is checked, so
			 * start at 16:1 and grow until the stream fits, up to the largest stream the library codes (X3H_MAX_CHUNK) */
			size_t cap = isize * 16 + 65536;
			if (cap > X3H_MAX_CHUNK) cap = X3H_MAX_CHUNK;
			for (;;) {
				optr = malloc(cap);
				if (!optr) die("out of memory");
				rc = x3h_decompress(ctxs[0], iptr, isize, optr, cap, &osize, &st);
				if (rc != X3H_E_OUTPUT_FULL) break;
				free(optr);
				if (cap == X3H_MAX_CHUNK) die("x3: decompress failed: the stream decodes to more than X3H_MAX_CHUNK (2^28 - 4096) bytes");
				cap = cap > X3H_MAX_CHUNK / 4 ? X3H_MAX_CHUNK : cap * 4;
			}
		}
		if (rc != X3H_OK) { fprintf(stderr, "x3: decompress failed: %s\n", x3h_strerror(rc)); return 1; }
		fprintf(stderr, "elapsed time: %f\n", st.ms_total / 1000.0);
	}
	const double t_call = now_ms();
	if (fwrite(optr, 1, osize, ostream) < osize) die("short write");
	g_phase = 5;
	const double t_write = now_ms();

	/* the statistics block of x3.c:662-693, same lines, same arithmetic (single-precision sums and quotients).  The GPU decoder accumulates no size
	 * estimates (the reference's decode_tag does, x3.c:96-97): after -d the four lines that print them are left out. */
	size_t size = decompress ? osize : isize, asize = decompress ? isize : osize;
	const float sizes[4] = { (float)st.est_bits[0], (float)st.est_bits[1], (float)st.est_bits[2], (float)st.est_bits[3] };
	size_t dict_hit_count = (size_t)(st.events[0] + st.events[1] + st.events[2]);
	size_t stream_size_dict = (size_t)ceilf(sizes[0] + sizes[1] + sizes[2]);
	size_t stream_size = (size_t)ceilf(sizes[0] + sizes[1] + sizes[2] + sizes[3]);
	fprintf(stderr, "input stream size: %zu\n", size);
	if (!decompress) fprintf(stderr, "output stream size: %zu\n", (stream_size + 7) / 8);
	fprintf(stderr, "dictionary: hit %zu, miss %zu\n", dict_hit_count, (size_t)st.events[3]);
	if (!decompress) fprintf(stderr, "codestream size: dictionary %zu / %f%%, new fragment %zu / %f%%\n",
	        (stream_size_dict + 7) / 8, 100.f * stream_size_dict / stream_size,
	        ((size_t)ceilf(sizes[3]) + 7) / 8, 100.f * (size_t)ceilf(sizes[3]) / stream_size);
	if (!decompress) fprintf(stderr, "\x1b[37;1mest. compression ratio: %f\x1b[0m\n", size / (float)((stream_size + 7) / 8));
	fprintf(stderr, "\x1b[37;1mreal compression ratio: %f\x1b[0m\n", size / (float)asize);
	fprintf(stderr, "number of events: ctx0 %zu, ctx1 %zu, miss1 %zu, new %zu\n", (size_t)st.events[0], (size_t)st.events[1], (size_t)st.events[2], (size_t)st.events[3]);
	if (!decompress) fprintf(stderr, "event sizes: ctx0 %f%%, ctx1 %f%%, miss1 %f%%, new %f%%\n",
	        100.f * (size_t)ceilf(sizes[0]) / stream_size, 100.f * (size_t)ceilf(sizes[1]) / stream_size,
	        100.f * (size_t)ceilf(sizes[2]) / stream_size, 100.f * (size_t)ceilf(sizes[3]) / stream_size);
	fprintf(stderr, "context entries: ctx0 %zu, ctx1 %zu\n", (size_t)st.ctx0_entries, (size_t)st.dict_elems);

	for (int i = 0; i < ngpu; i++) x3h_ctx_destroy(ctxs[i]);
	g_phase = 6;
	free(iptr);
	free(optr);
	fclose(istream);
	if (fclose(ostream)) die("short write");
	/* everything is written and every handle is released: leave without the HIP runtime's exit-time teardown (nothing of ours is left
	 * for it to release) */
	fflush(NULL);
	g_phase = 7;
	if (getenv("X3H_DEBUG")) fprintf(stderr, "[x3] ms: input read %.1f beside the handles (ready after %.1f), library call %.1f, write %.1f, release + close %.1f\n", t_read - t_start, t_ctx - t_start,
	                                 t_call - t_ctx, t_write - t_call, now_ms() - t_write);
	if (getenv("X3_CLI_RUNTIME_TEARDOWN")) return 0; /* (diagnostics) leave through exit(): atexit handlers and the HIP runtime's static destructors run */
	_exit(0);
}
