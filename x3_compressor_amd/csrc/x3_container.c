/*
 * x3_container.c -- the X3C1 chunk container of include/x3hip.h (host side, plain C; also compiles as C++ for the emulator build).
 *
 * The reference writes ONE raw code stream per file (x3.c:603-611: no header, no length, no parameters).  Independent chunks are
 * the only way its path shards, so a multi-chunk output needs a frame: magic, parameter echo, per-chunk
 * (raw_len, comp_len), then the chunk streams, each byte-identical to `x3 -z` of that chunk alone.  A single chunk is never
 * wrapped.  All fields little-endian, written and read byte by byte (no struct punning, no host-endianness dependence).
 */
#include <string.h>
#include "../../include/x3hip.h"

#define HDR_BYTES 32u
#define ENT_BYTES 16u

static void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static void put64(uint8_t *p, uint64_t v) { put32(p, (uint32_t)v); put32(p + 4, (uint32_t)(v >> 32)); }
static uint32_t get32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static uint64_t get64(const uint8_t *p) { return (uint64_t)get32(p) | (uint64_t)get32(p + 4) << 32; }

#ifdef __cplusplus
extern "C" {
#endif

size_t x3h_container_header_bytes(int nchunks)
{
	return nchunks > 1 ? (size_t)HDR_BYTES + (size_t)ENT_BYTES * (size_t)nchunks : 0;
}

size_t x3h_container_bound(size_t n, size_t chunk_bytes)
{
	if (!chunk_bytes || chunk_bytes > X3H_MAX_CHUNK) chunk_bytes = X3H_MAX_CHUNK;
	const size_t nch = n ? (n + chunk_bytes - 1) / chunk_bytes : 1;
	/* x3h_compress_bound is affine: 12 n + 64 per chunk */
	return x3h_container_header_bytes((int)nch) + 12 * n + 64 * nch;
}

int x3h_container_write_header(uint8_t *dst, size_t cap, const x3h_params *prm, int nchunks,
                               const uint64_t *raw_lens, const uint64_t *comp_lens)
{
	if (!dst || !prm || nchunks < 2 || !raw_lens || !comp_lens) return X3H_E_ARG;
	if (cap < x3h_container_header_bytes(nchunks)) return X3H_E_OUTPUT_FULL;
	memcpy(dst, "X3C1", 4);
	put32(dst + 4, X3H_CONTAINER_VERSION);
	put32(dst + 8, prm->window_bytes);
	put32(dst + 12, (uint32_t)prm->max_match_count);
	put32(dst + 16, prm->factor1);
	put32(dst + 20, prm->factor2);
	put32(dst + 24, (uint32_t)prm->nl_mode);
	put32(dst + 28, (uint32_t)nchunks);
	for (int i = 0; i < nchunks; i++) {
		put64(dst + HDR_BYTES + (size_t)ENT_BYTES * (size_t)i, raw_lens[i]);
		put64(dst + HDR_BYTES + (size_t)ENT_BYTES * (size_t)i + 8, comp_lens[i]);
	}
	return X3H_OK;
}

int x3h_container_probe(const uint8_t *blob, size_t n, x3h_params *prm, int *nchunks, uint64_t *raw_total)
{
	if (!blob && n) return X3H_E_ARG;
	/* a raw x3 stream can in principle begin with these four bytes too; the frame is only accepted when everything else adds up */
	if (n < 4 || memcmp(blob, "X3C1", 4) != 0) return X3H_NOT_A_CONTAINER;
	if (n < HDR_BYTES) return X3H_E_CORRUPT;
	if (get32(blob + 4) != X3H_CONTAINER_VERSION) return X3H_E_CORRUPT;
	const uint32_t nch = get32(blob + 28);
	if (nch < 2 || nch > 0x7FFFFFFFu || (n - HDR_BYTES) / ENT_BYTES < nch) return X3H_E_CORRUPT;
	uint64_t comp = 0, raw = 0;
	const uint64_t payload = (uint64_t)n - HDR_BYTES - (uint64_t)ENT_BYTES * nch;
	for (uint32_t i = 0; i < nch; i++) {
		const uint64_t r = get64(blob + HDR_BYTES + (size_t)ENT_BYTES * i), c = get64(blob + HDR_BYTES + (size_t)ENT_BYTES * i + 8);
		/* every x3 stream is a positive number of 32-bit words (bio.c:105-112) and codes at most X3H_MAX_CHUNK bytes here */
		if (r > X3H_MAX_CHUNK || c < 4 || (c & 3) || c > payload - comp) return X3H_E_CORRUPT;
		comp += c;
		raw += r;
	}
	if (comp != payload) return X3H_E_CORRUPT;
	if (prm) {
		prm->window_bytes = get32(blob + 8);
		prm->max_match_count = (int32_t)get32(blob + 12);
		prm->factor1 = get32(blob + 16);
		prm->factor2 = get32(blob + 20);
		prm->nl_mode = (int32_t)get32(blob + 24);
	}
	if (nchunks) *nchunks = (int)nch;
	if (raw_total) *raw_total = raw;
	return X3H_OK;
}

int x3h_container_table(const uint8_t *blob, size_t n, uint64_t *raw_lens, uint64_t *comp_offsets)
{
	int nch = 0;
	const int rc = x3h_container_probe(blob, n, NULL, &nch, NULL);
	if (rc != X3H_OK) return rc == X3H_NOT_A_CONTAINER ? X3H_E_CORRUPT : rc;
	if (!raw_lens || !comp_offsets) return X3H_E_ARG;
	uint64_t off = HDR_BYTES + (uint64_t)ENT_BYTES * (uint64_t)nch;
	for (int i = 0; i < nch; i++) {
		raw_lens[i] = get64(blob + HDR_BYTES + (size_t)ENT_BYTES * (size_t)i);
		comp_offsets[i] = off;
		off += get64(blob + HDR_BYTES + (size_t)ENT_BYTES * (size_t)i + 8);
	}
	comp_offsets[nch] = off;
	return X3H_OK;
}

#ifdef __cplusplus
}
#endif
