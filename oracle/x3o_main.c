/* x3o_main.c -- tiny file front-end for the oracle (TEST INFRASTRUCTURE ONLY).
 * usage: x3o (-z|-d) [-w KiB] [-t N] [-m N] [-n N] [-x] [-M] in out      (-M: go through m[] closed form)
 * Prints "elapsed <seconds>" for the codec call on stderr, like the reference's timing of x3.c:597-601. */
#define _POSIX_C_SOURCE 200809L
#include "x3_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
	x3o_params prm; x3o_default_params(&prm);
	int mode = 'z', via_m = 0, o;
	while ((o = getopt(argc, argv, "zdw:t:m:n:xM")) != -1) {
		switch (o) {
			case 'z': case 'd': mode = o; break;
			case 'w': prm.window_bytes = (uint32_t)atoi(optarg) * 1024u; break;
			case 't': prm.max_match_count = atoi(optarg); break;
			case 'm': prm.factor1 = (uint32_t)atoi(optarg); break;
			case 'n': prm.factor2 = (uint32_t)atoi(optarg); break;
			case 'x': prm.nl_mode = 1; break;
			case 'M': via_m = 1; break;
			default: return 2;
		}
	}
	if (argc - optind != 2) { fprintf(stderr, "usage: x3o (-z|-d) [opts] in out\n"); return 2; }
	FILE *f = fopen(argv[optind], "rb");
	if (!f) { perror("open input"); return 1; }
	fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
	uint8_t *in = malloc(n ? n : 1);
	if (fread(in, 1, n, f) != (size_t)n) { perror("read"); return 1; }
	fclose(f);
	size_t cap = mode == 'z' ? x3o_compress_bound(n) : (size_t)n * 4096 + 65536, out_len = 0;
	uint8_t *out = malloc(cap);
	int rc;
	double t0 = now();
	if (mode == 'z') {
		x3o_stats st;
		if (via_m) {
			uint8_t *m = malloc(n ? n : 1);
			x3o_scan_m(&prm, in, n, m);
			rc = x3o_compress_via_m(&prm, in, n, m, out, cap, &out_len, &st);
			free(m);
		} else rc = x3o_compress(&prm, in, n, out, cap, &out_len, &st);
		fprintf(stderr, "elapsed %f\nsteps %llu dict %llu ctx0 %llu\n", now() - t0,
		        (unsigned long long)st.steps, (unsigned long long)st.dict_elems, (unsigned long long)st.ctx0_entries);
	} else {
		rc = x3o_decompress(in, n, out, cap, &out_len);
		fprintf(stderr, "elapsed %f\n", now() - t0);
	}
	if (rc) { fprintf(stderr, "oracle error %d\n", rc); return 1; }
	f = fopen(argv[optind + 1], "wb");
	if (!f || fwrite(out, 1, out_len, f) != out_len) { perror("write"); return 1; }
	fclose(f);
	return 0;
}
