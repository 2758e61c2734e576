/* debugging aid: x3h_compress on a zero-filled buffer, create/destroy per run, with a SIGSEGV backtrace */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "../../include/x3hip.h"
static void on_segv(int sig) { void *bt[64]; int n = backtrace(bt, 64); backtrace_symbols_fd(bt, n, 2); _exit(100 + sig); }
int main(int argc, char **argv)
{
	signal(SIGSEGV, on_segv);
	size_t n = argc > 1 ? (size_t)atol(argv[1]) : 1024;
	int reps = argc > 2 ? atoi(argv[2]) : 1;
	for (int r = 0; r < reps; r++) {
		x3h_ctx *c = NULL;
		if (x3h_ctx_create(&c, 0)) return 2;
		unsigned char *in = calloc(n, 1), *out = malloc(12 * n + 64);
		size_t ol = 0; x3h_stats st; x3h_params p; x3h_default_params(&p);
		int rc = x3h_compress(c, &p, in, n, out, 12 * n + 64, &ol, &st);
		fprintf(stderr, "run %d rc %d out %zu\n", r, rc, ol);
		x3h_ctx_destroy(c);
		free(in); free(out);
	}
	return 0;
}
