/*
 * x3_cli.c -- the x3 command line, host side in C, on top of the C ABI of include/x3hip.h.
 *
 * Same letters, defaults and file-name rules as the reference's main() (x3.c:479-548): -z -d -f -k -h -t N -w N(KiB)
 * -m N -n N -x, 0/1/2 positional arguments, "<in>.x3" default output, no clobber without -f.  Compressed output
 * is the raw x3 code stream (no header), bit-identical to the reference for the same input and parameters.
 * The statistics block on stderr follows x3.c:662-693 for the integer fields (the float size estimates are not
 * reproduced).  Errors print a message and exit(1) instead of abort().
 *
 * Additive options (new, none of them changes what the reference's letters do):
 *   -g N             use GPU N (default 0)
 *   --gpus a,b,...   use these GPUs side by side (chunks are dealt out in contiguous blocks); the finished streams are
 *                    concatenated by ONE RCCL send/receive group on the first GPU (x3h_compress_container_rccl)
 *   --chunk-kib N    cut the input into independent chunks of N KiB, each coded as its own x3 stream, and write the X3C1
 *                    container (include/x3hip.h); an input of one chunk is still written as the raw stream.  Inputs above
 *                    128 MiB (X3H_MAX_CHUNK) are always chunked.
 *   --batch-mib N    chunks are coded in sub-batches of at most N MiB of input (default 64): bounds the workspace in HBM (~350 B per byte)
 * -d recognises a container by its magic and decodes the chunks as one batch; anything else is a raw x3 stream.
 */
#define _GNU_SOURCE /* getopt_long */
#include <execinfo.h>
#include <getopt.h>
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
	char msg[64];
	int k = snprintf(msg, sizeof msg, "x3: fatal signal %d in phase %d\n", sig, (int)g_phase);
	if (k > 0 && write(2, msg, (size_t)k) < 0) { /* nothing left to do about it */ }
	void *bt[48];
	backtrace_symbols_fd(bt, backtrace(bt, 48), 2);
	signal(sig, SIG_DFL);
	raise(sig);
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
	fprintf(stderr, " --batch-mib NUM : code at most NUM MiB of chunks at a time (default 64; bounds the GPU workspace)\n");
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
struct handle_job { int ngpu; const int *gpus; x3h_ctx **ctxs; uint64_t batch_mib; int set_batch; int rc; };
static void *make_handles(void *arg)
{
	struct handle_job *j = arg;
	j->rc = X3H_OK;
	for (int i = 0; i < j->ngpu && j->rc == X3H_OK; i++) j->rc = x3h_ctx_create(&j->ctxs[i], j->gpus[i]);
	/* a command-line process lives for one call: keep its workspace small (the sub-batches of a big input follow each other, x3hip.h) */
	for (int i = 0; j->set_batch && i < j->ngpu && j->rc == X3H_OK; i++) j->rc = x3h_ctx_set_batch_bytes(j->ctxs[i], j->batch_mib << 20);
	return NULL;
}

int main(int argc, char *argv[])
{
	int decompress = 0, force = 0, o, ngpu = 1, gpus[MAX_GPUS] = { 0 };
	size_t chunk_bytes = 0;
	uint64_t batch_mib = 64;
	x3h_params prm;
	const int fatal[] = { SIGSEGV, SIGBUS, SIGILL, SIGFPE, SIGABRT };
	for (size_t i = 0; i < sizeof fatal / sizeof fatal[0]; i++) signal(fatal[i], on_fatal);
	g_phase = 1;
	x3h_default_params(&prm);
	static const struct option longopts[] = { { "chunk-kib", required_argument, NULL, 1000 }, { "gpus", required_argument, NULL, 1001 }, { "batch-mib", required_argument, NULL, 1002 }, { NULL, 0, NULL, 0 } };

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
				if (k <= 0 || (size_t)k * 1024 > X3H_MAX_CHUNK) die("--chunk-kib: between 1 and 131072");
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
	struct handle_job job = { ngpu, gpus, ctxs, batch_mib, getenv("X3H_BATCH_BYTES") == NULL, X3H_OK };
	pthread_t th;
	const int threaded = pthread_create(&th, NULL, make_handles, &job) == 0;
	if (!threaded) make_handles(&job);

	size_t isize = 0, osize = 0;
	g_phase = 3;
	unsigned char *iptr = read_all(istream, &isize), *optr = NULL;
	const double t_read = now_ms();
	if (threaded) pthread_join(th, NULL);
	rc = job.rc;
	if (rc != X3H_OK) { fprintf(stderr, "x3: %s (the hot path only exists as gfx950 HIP kernels; no CPU fallback)\n", x3h_strerror(rc)); return 1; }
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
		size_t cap = x3h_container_bound(isize, chunk_bytes);
		optr = malloc(cap);
		if (!optr) die("out of memory");
		/* several GPUs: the chunk streams stay in HBM and ONE RCCL exchange concatenates them on the first GPU (x3h_compress_container_rccl);
		 * X3_NO_RCCL=1, the same GPU named twice, or a machine without librccl: host-staged concat, same bytes */
		rc = X3H_E_RCCL;
		if (ngpu > 1 && !getenv("X3_NO_RCCL")) {
			rc = x3h_compress_container_rccl(ctxs, ngpu, &prm, iptr, isize, chunk_bytes, optr, cap, &osize, &st);
			x3h_rccl_release();
			if (rc == X3H_OK) fprintf(stderr, "final concat: one RCCL gather over %d GPUs\n", ngpu);
		}
		if (rc == X3H_E_RCCL || rc == X3H_E_ARG) rc = x3h_compress_container(ctxs, ngpu, &prm, iptr, isize, chunk_bytes, optr, cap, &osize, &st);
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
				if (cap == X3H_MAX_CHUNK) die("x3: decompress failed: the stream decodes to more than 128 MiB");
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

	size_t size = decompress ? osize : isize, asize = decompress ? isize : osize;
	fprintf(stderr, "input stream size: %zu\n", size);
	fprintf(stderr, "dictionary: hit %llu, miss %llu\n", (unsigned long long)(st.events[0] + st.events[1] + st.events[2]),
	        (unsigned long long)st.events[3]);
	fprintf(stderr, "real compression ratio: %f\n", asize ? size / (float)asize : 0.f);
	fprintf(stderr, "number of events: ctx0 %llu, ctx1 %llu, miss1 %llu, new %llu\n", (unsigned long long)st.events[0],
	        (unsigned long long)st.events[1], (unsigned long long)st.events[2], (unsigned long long)st.events[3]);
	fprintf(stderr, "context entries: ctx0 %llu, ctx1 %llu\n", (unsigned long long)st.ctx0_entries, (unsigned long long)st.dict_elems);

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
