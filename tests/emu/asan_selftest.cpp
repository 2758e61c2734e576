/* asan_selftest.cpp -- TEST INFRASTRUCTURE ONLY (host sanitizer target).
 * The kernel + host SOURCES of libx3hip.so compiled for the SIMT emulator (X3_EMU) together with the oracle, all under
 * -fsanitize=address,undefined, driven through the C ABI of include/x3hip.h on small deterministic inputs:
 * compress == oracle stream, decompress == input, chunked batch, container round trip, error paths.
 * GPU sanitizers are not available on the pool, so this CPU build is where out-of-bounds indexing in the kernels' logic shows. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/x3hip.h"
#include "../../oracle/x3_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 32); }

static std::vector<uint8_t> make_input(int kind, size_t n)
{
	std::vector<uint8_t> v(n);
	static const char *words[] = { "the ", "of ", "and ", "window ", "match ", "dictionary ", "x3 ", "coder ", "a ", "context " };
	size_t i = 0;
	switch (kind) {
		case 0: while (i < n) { const char *w = words[rnd() % 10]; for (; *w && i < n; w++) v[i++] = (uint8_t)*w; } break; /* text */
		case 1: for (; i < n; i++) v[i] = (uint8_t)(rnd() & 0xFF); break;                                          /* random */
		case 2: break;                                                                                             /* zeros  */
		case 3: for (; i < n; i++) v[i] = (uint8_t)("abcdefg"[i % 7]); break;                                       /* periodic */
		default: for (; i < n; i++) v[i] = (uint8_t)((rnd() & 3) * 17); break;                                      /* 4 symbols */
	}
	return v;
}

static int same(const uint8_t *a, const uint8_t *b, size_t n) { return n == 0 || memcmp(a, b, n) == 0; }
#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "asan_selftest: %s:%d: %s failed\n", __FILE__, __LINE__, #c); exit(1); } } while (0)

int main()
{
	x3h_ctx *ctx = nullptr, *ctx2 = nullptr;
	REQUIRE(x3h_ctx_create(&ctx, 0) == X3H_OK && x3h_ctx_create(&ctx2, 0) == X3H_OK);
	setenv("X3H_MULTI_SERIAL", "1", 1); /* the emulator is single-threaded */
	struct { int kind; size_t n; uint32_t w; int t; uint32_t f2; int nl; } cases[] = {
		{ 0, 0, 8192, 15, 0, 0 }, { 0, 1, 8192, 15, 0, 0 }, { 0, 2200, 1024, 4, 0, 0 } /* crosses the 2048-position parse block */, { 1, 260, 1024, 2, 0, 0 },
		{ 2, 500, 1024, 15, 0, 0 }, { 3, 400, 1024, 3, 0, 0 }, { 4, 450, 2048, 8, 0, 0 }, { 0, 600, 1024, 3, 2, 0 }, { 0, 600, 1024, 3, 0, 1 },
	};
	for (auto &c : cases) {
		if (getenv("X3_SELFTEST_VERBOSE")) fprintf(stderr, "case kind %d n %zu w %u t %d\n", c.kind, c.n, c.w, c.t);
		std::vector<uint8_t> in = make_input(c.kind, c.n);
		x3h_params p; x3h_default_params(&p); p.window_bytes = c.w; p.max_match_count = c.t; p.factor2 = c.f2; p.nl_mode = c.nl;
		x3o_params op = { c.w, c.t, 4, c.f2, c.nl };
		std::vector<uint8_t> want(x3o_compress_bound(c.n)), got(x3h_compress_bound(c.n)), back(c.n + 16);
		size_t wl = 0, gl = 0, bl = 0;
		x3o_stats os; x3h_stats hs;
		REQUIRE(x3o_compress(&op, in.data(), c.n, want.data(), want.size(), &wl, &os) == X3O_OK);
		REQUIRE(x3h_compress(ctx, &p, in.data(), c.n, got.data(), got.size(), &gl, &hs) == X3H_OK);
		REQUIRE(gl == wl && same(got.data(), want.data(), wl));
		REQUIRE(hs.steps == os.steps && hs.dict_elems == os.dict_elems && hs.ctx0_entries == os.ctx0_entries);
		REQUIRE(x3h_decompress(ctx, got.data(), gl, back.data(), back.size(), &bl, nullptr) == X3H_OK);
		REQUIRE(bl == c.n && same(back.data(), in.data(), c.n));
		if (c.n > 100) REQUIRE(x3h_decompress(ctx, got.data(), gl, back.data(), c.n - 1, &bl, nullptr) == X3H_E_OUTPUT_FULL);
		if (c.n == 2200) REQUIRE(x3h_compress(ctx, &p, in.data(), c.n, got.data(), 8, &gl, nullptr) == X3H_E_OUTPUT_FULL); /* (one case: a whole second run of the pipeline) */
	}
	/* pipelined schedule + container over two handles */
	setenv("X3H_PIPE_MIN", "1", 1);
	x3h_ctx *pctx = nullptr;
	REQUIRE(x3h_ctx_create(&pctx, 0) == X3H_OK);
	{
		std::vector<uint8_t> in = make_input(0, 2500);
		x3h_params p; x3h_default_params(&p); p.window_bytes = 2048; p.max_match_count = 8;
		x3o_params op = { 2048, 8, 4, 0, 0 };
		std::vector<uint8_t> want(x3o_compress_bound(in.size())), got(x3h_compress_bound(in.size()));
		size_t wl = 0, gl = 0;
		REQUIRE(x3o_compress(&op, in.data(), in.size(), want.data(), want.size(), &wl, nullptr) == X3O_OK);
		REQUIRE(x3h_compress(pctx, &p, in.data(), in.size(), got.data(), got.size(), &gl, nullptr) == X3H_OK);
		REQUIRE(gl == wl && same(got.data(), want.data(), wl));
		x3h_ctx *two[2] = { ctx, ctx2 };
		std::vector<uint8_t> box(x3h_container_bound(in.size(), 1024)), back(in.size());
		size_t cl = 0, bl = 0;
		REQUIRE(x3h_compress_container(two, 2, &p, in.data(), in.size(), 1024, box.data(), box.size(), &cl, nullptr) == X3H_OK);
		int nch = 0; uint64_t raw = 0;
		REQUIRE(x3h_container_probe(box.data(), cl, nullptr, &nch, &raw) == X3H_OK && nch == 3 && raw == in.size());
		REQUIRE(x3h_decompress_container(two, 2, box.data(), cl, back.data(), back.size(), &bl, nullptr) == X3H_OK);
		REQUIRE(bl == in.size() && same(back.data(), in.data(), bl));
		REQUIRE(x3h_container_probe(box.data(), cl - 4, nullptr, &nch, &raw) == X3H_E_CORRUPT);
		uint8_t junk[64]; for (int i = 0; i < 64; i++) junk[i] = (uint8_t)(i * 37 + 11);
		REQUIRE(x3h_decompress(ctx, junk, 64, back.data(), back.size(), &bl, nullptr) != X3H_OK);
	}
	/* round 3: K1 by one workgroup per chunk (scan3.hip) in both of its forms with its dense-class refinement (X3_WALK_DENSE is 6 in this build),
	 * bits written behind every coder segment, hits arranged by one workgroup per stream -- a ragged batch, each stream against the oracle */
	{
		struct { const char *k, *v; } envs[][3] = {
			{ { "X3H_SEG_MIN", "1" }, { "X3H_SEG_EMIT", "1" }, { "X3H_PIPE_MIN", "1" } },
			{ { "X3H_SEG_MIN", "1" }, { "X3H_SEG_SMALL_MAX", "0" }, { "X3H_PIPE_MIN", "0" } },
			{ { "X3H_STREAM_KERNELS", "1" }, { "X3H_ARRANGE", "1" }, { "X3H_PIPE_MIN", "0" } },
		};
		const int kinds[] = { 0, 2, 4, 0 };
		const size_t lens[] = { 450, 300, 300, 0 };
		const int NS = 4;
		std::vector<uint8_t> all;
		std::vector<uint64_t> off(1, 0);
		for (int i = 0; i < NS; i++) { std::vector<uint8_t> v = make_input(kinds[i], lens[i]); all.insert(all.end(), v.begin(), v.end()); off.push_back(all.size()); }
		x3h_params p; x3h_default_params(&p); p.window_bytes = 1024; p.max_match_count = 3;
		x3o_params op = { 1024, 3, 4, 0, 0 };
		const int env_streams[] = { 1, 1, 3 }; /* (K1 by an emulated 1024-thread workgroup per chunk under ASan costs ~10 s a chunk) */
		int env_i = 0;
		for (auto &env : envs) {
			const int ns = env_streams[env_i++];
			unsetenv("X3H_PIPE_MIN");
			for (auto &kv : env) if (kv.k) setenv(kv.k, kv.v, 1);
			x3h_ctx *e = nullptr;
			REQUIRE(x3h_ctx_create(&e, 0) == X3H_OK);
			const uint64_t stride = 8192;
			std::vector<uint8_t> out(stride * NS);
			uint64_t lens_out[NS];
			REQUIRE(x3h_compress_chunks(e, &p, all.data(), off.data(), ns, out.data(), stride, lens_out, nullptr) == X3H_OK);
			for (int i = 0; i < ns; i++) {
				std::vector<uint8_t> want(x3o_compress_bound(lens[i]));
				size_t wl = 0;
				REQUIRE(x3o_compress(&op, all.data() + off[(size_t)i], lens[i], want.data(), want.size(), &wl, nullptr) == X3O_OK);
				REQUIRE(lens_out[i] == wl && same(out.data() + (size_t)i * stride, want.data(), wl));
			}
			x3h_ctx_destroy(e);
			for (auto &kv : env) if (kv.k) unsetenv(kv.k);
		}
	}
	x3h_ctx_destroy(pctx); x3h_ctx_destroy(ctx2); x3h_ctx_destroy(ctx);
	printf("asan_selftest ok\n");
	return 0;
}
