/*
 * prims.hip -- the chip-wide primitives of the generic paths (K1 of a few long streams: scan2.hip; K3 stage after stage and the prefix-wise
 * schedule: code2.hip; slices whose local keys outgrow the per-stream sort: code4.hip): a stable LSD radix sort of (key, value) pairs and three
 * prefix scans.  Hand-written for gfx950 since round 5 (rounds 1-4 called rocPRIM here); the same sources run on the SIMT emulator (tests/emu).
 *
 * Radix sort, one pass per 8 key bits, three launches per pass:
 *   x3p_hist_kernel    one workgroup per tile of 4096 pairs counts its digits in LDS -> counts[digit][tile]                       (4 B read per pair)
 *   exclusive scan     of counts in [digit][tile] order = where each tile's run of each digit starts in the output            (256 x tiles words)
 *   x3p_scatter_kernel the tile again: every wavefront ranks its lanes' digits with one ballot per digit bit (seg_rank.h, the tile machinery of
 *                      scan3.hip), a [digit][wave] counter table and one workgroup scan turn that into the tile-sorted order, the tile is staged
 *                      in LDS in that order and leaves as one contiguous run per digit                                 (8 B read + 8 B written per pair)
 *   Stable: tile order is (wave, element, lane), runs of a digit are laid out tile after tile.  Bound: HBM, 20 B per pair and pass (a one-sweep sort
 *   with decoupled look-back moves 16 B; the offsets here come from the chained scan below, the passes themselves never wait for one another).
 * Scans: one pass, tiles of 4096 chained by decoupled look-back with a bounded wait (below).
 */
#include "x3_host.h"
#include "seg_rank.h"

#define X3P_THREADS 512u
#define X3P_WAVES   (X3P_THREADS / X3_WAVE)
#define X3P_E       8u
#define X3P_TILE    (X3P_THREADS * X3P_E)
#define X3P_CS      (X3P_WAVES + 1u) /* words between two digits' per-wave counters (an odd stride: a wavefront's lanes -- one wave number, many digits -- meet in every LDS bank) */

/* ---- scans ----------------------------------------------------------------------------------------------------------------------------------- */
/* ONE pass over the data: a tile takes a ticket (tiles are numbered in the order their workgroups START, so every tile with a smaller number is running or done),
 * scans itself, publishes its total and looks back over its predecessors' status words -- {state, value} in one 64-bit word, so a reader never sees half of one --
 * adding totals until it meets a tile that already knows its prefix (decoupled look-back).  A predecessor publishes its total without waiting for anybody, so the
 * wait is short and cannot deadlock; it is bounded all the same (X3P_SPIN_MAX polls, then the error word is set and the tile goes on): a kernel of this library never
 * spins without an exit.  Loads and stores are striped (row k of a tile = 512 consecutive entries, one per thread): a wavefront's access is one contiguous run. */
enum { X3P_SUM = 0, X3P_TOPBIT = 1, X3P_MAX = 2 };
#define X3P_ST_EMPTY 0u
#define X3P_ST_AGG   1u
#define X3P_ST_INCL  2u
#define X3P_SPIN_MAX (1u << 22)
struct X3pScanArgs {
	const uint32_t *in;     /* X3P_SUM / X3P_MAX: values; X3P_TOPBIT: uint4 records, the value is bit 31 of .w */
	uint32_t *out;
	uint64_t *status;       /* one word per tile, zeroed by the caller; behind them (status[ntiles]): the ticket counter (low word) and the error word (high word) */
	size_t n;               /* entries that carry a value */
	size_t nout;            /* entries written: n + 1 for the exclusive forms (out[n] = total), n for the inclusive maximum */
	uint32_t ntiles, _pad;
};

/* set (and never cleared by a kernel) when a tile gave up waiting for a predecessor: the forward-progress argument above says this cannot happen, so if it ever does the
 * call must fail loudly instead of returning a wrong prefix -- api.hip reads the word behind every call (x3p_check_error) */
#ifndef X3_EMU
__device__ uint32_t x3p_dev_error;
#else
uint32_t x3p_dev_error;
#endif

template <int MODE> __device__ static __forceinline__ uint32_t x3p_scan_load(const X3pScanArgs &a, size_t i)
{
	if (i >= a.n) return 0u;
	return MODE == X3P_TOPBIT ? ((const uint4 *)a.in)[i].w >> 31 : a.in[i];
}
template <int MODE> __device__ static __forceinline__ uint32_t x3p_op(uint32_t x, uint32_t y) { return MODE == X3P_MAX ? (x > y ? x : y) : x + y; }
/* inclusive scan over the wavefront under MODE's operator */
template <int MODE> __device__ static __forceinline__ uint32_t x3p_wave_incl(uint32_t v)
{
	if (MODE != X3P_MAX) return x3_wave_incl_scan_u32(v);
	for (uint32_t d = 1; d < X3_WAVE; d <<= 1) { const uint32_t u = x3_shfl_up_u32(v, d); if (x3_lane() >= d) v = v > u ? v : u; }
	return v;
}
/* the workgroup's threads each hold `mine`: -> the operator over the threads before this one (identity 0), and the workgroup's total in *total */
template <int MODE> __device__ static __forceinline__ uint32_t x3p_block_excl(uint32_t mine, uint32_t *s_w, uint32_t *total)
{
	const uint32_t lane = x3_lane(), wv = threadIdx.x / X3_WAVE;
	const uint32_t incl = x3p_wave_incl<MODE>(mine);
	if (lane == X3_WAVE - 1) s_w[wv] = incl;
	__syncthreads();
	uint32_t before = 0, tot = 0;
	for (uint32_t w = 0; w < X3P_WAVES; w++) { const uint32_t t = s_w[w]; if (w < wv) before = x3p_op<MODE>(before, t); tot = x3p_op<MODE>(tot, t); }
	const uint32_t up = x3_shfl_up_u32(incl, 1);
	if (lane) before = x3p_op<MODE>(before, up);
	*total = tot;
	__syncthreads(); /* s_w is free again */
	return before;
}

__device__ static __forceinline__ uint64_t x3p_status_load(const uint64_t *p)
{
#ifndef X3_EMU
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
	return *p;
#endif
}
__device__ static __forceinline__ void x3p_status_store(uint64_t *p, uint32_t state, uint32_t value)
{
	const uint64_t w = ((uint64_t)state << 32) | value;
#ifndef X3_EMU
	__hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
	*p = w;
#endif
}

template <int MODE> __device__ static void x3p_scan_body(const X3pScanArgs &a)
{
	static_assert(X3P_E * X3P_WAVES == X3_WAVE, "the (row, wave) totals of a tile are scanned by one wavefront");
	X3_LDS uint32_t s_tab[X3_WAVE]; /* [row][wave]: total of that wavefront's 64 entries of that row, then the operator over everything of the tile before them */
	X3_LDS uint32_t s_tile, s_prefix;
	const uint32_t lane = x3_lane(), wv = threadIdx.x / X3_WAVE;
	if (threadIdx.x == 0) s_tile = atomicAdd((uint32_t *)(a.status + a.ntiles), 1u);
	__syncthreads();
	const uint32_t tile = s_tile;
	const size_t base = (size_t)tile * X3P_TILE + threadIdx.x;
	uint32_t v[X3P_E], incl[X3P_E];
	for (uint32_t k = 0; k < X3P_E; k++) v[k] = x3p_scan_load<MODE>(a, base + (size_t)k * X3P_THREADS);
	for (uint32_t k = 0; k < X3P_E; k++) {
		incl[k] = x3p_wave_incl<MODE>(v[k]);
		if (lane == X3_WAVE - 1) s_tab[k * X3P_WAVES + wv] = incl[k];
	}
	__syncthreads();
	if (wv == 0) {
		const uint32_t mine = s_tab[lane];
		const uint32_t inc = x3p_wave_incl<MODE>(mine);
		const uint32_t agg = x3_bcast_u32(inc, X3_WAVE - 1);
		const uint32_t up = x3_shfl_up_u32(inc, 1);
		s_tab[lane] = lane ? up : 0u;
		uint32_t excl = 0;
		if (tile == 0) { if (lane == 0) x3p_status_store(a.status, X3P_ST_INCL, agg); }
		else {
			if (lane == 0) x3p_status_store(a.status + tile, X3P_ST_AGG, agg);
			int64_t idx = (int64_t)tile - 1;
			uint32_t polls = 0;
			for (;;) {
				const int64_t mi = idx - (int64_t)lane; /* lane 0: the nearest predecessor */
				uint64_t w = mi >= 0 ? x3p_status_load(a.status + mi) : ((uint64_t)X3P_ST_INCL << 32);
				while (x3_ballot((uint32_t)(w >> 32) == X3P_ST_EMPTY) && polls < X3P_SPIN_MAX) { /* a predecessor has not published yet (it is running: tiles are numbered as they start) */
					polls++;
#ifndef X3_EMU
					__builtin_amdgcn_s_sleep(1);
#endif
					if ((uint32_t)(w >> 32) == X3P_ST_EMPTY) w = x3p_status_load(a.status + mi);
				}
				if (x3_ballot((uint32_t)(w >> 32) == X3P_ST_EMPTY)) { if (lane == 0) { atomicOr((uint32_t *)(a.status + a.ntiles) + 1, 1u); atomicOr(&x3p_dev_error, 1u); } break; } /* gave up (never seen): results are wrong, nothing hangs */
				const uint64_t known = x3_ballot((uint32_t)(w >> 32) == X3P_ST_INCL);
				const uint32_t upto = known ? (uint32_t)x3_ctz64(known) : X3_WAVE - 1; /* lanes 0..upto contribute */
				const uint32_t part = x3p_wave_incl<MODE>(lane <= upto ? (uint32_t)w : 0u);
				excl = x3p_op<MODE>(excl, x3_bcast_u32(part, X3_WAVE - 1));
				if (known) break;
				idx -= X3_WAVE;
			}
			if (lane == 0) x3p_status_store(a.status + tile, X3P_ST_INCL, x3p_op<MODE>(excl, agg));
		}
		if (lane == 0) s_prefix = excl;
	}
	__syncthreads();
	const uint32_t pre = s_prefix;
	for (uint32_t k = 0; k < X3P_E; k++) {
		const size_t i = base + (size_t)k * X3P_THREADS;
		if (i >= a.nout) continue;
		const uint32_t before = x3p_op<MODE>(pre, s_tab[k * X3P_WAVES + wv]);
		if (MODE == X3P_MAX) a.out[i] = x3p_op<MODE>(before, incl[k]);
		else a.out[i] = before + incl[k] - v[k];
	}
}

/* ---- radix sort -------------------------------------------------------------------------------------------------------------------------------- */
struct X3pSortArgs {
	const uint32_t *kin, *vin;
	uint32_t *kout, *vout;
	uint32_t *counts;       /* [digit][tile]: counted by the histogram kernel, scanned in place (exclusive, one more entry behind the last) */
	size_t n;
	uint32_t shift, mask, ntiles, ndig;
};

__device__ static void x3p_hist_body(const X3pSortArgs &a)
{
	X3_LDS uint32_t s_h[256];
	if (threadIdx.x < 256u) s_h[threadIdx.x] = 0u;
	__syncthreads();
	const size_t base = (size_t)blockIdx.x * X3P_TILE;
	for (uint32_t e = 0; e < X3P_E; e++) {
		const size_t i = base + (size_t)e * X3P_THREADS + threadIdx.x;
		if (i < a.n) atomicAdd(&s_h[(a.kin[i] >> a.shift) & a.mask], 1u);
	}
	__syncthreads();
	if (threadIdx.x < a.ndig) a.counts[(size_t)threadIdx.x * a.ntiles + blockIdx.x] = s_h[threadIdx.x];
}

__device__ static void x3p_scatter_body(const X3pSortArgs &a)
{
	X3_LDS uint32_t s_cnt[256u * X3P_CS]; /* [digit][wave]: the wave's count of the digit, then the count of the waves before it */
	X3_LDS uint32_t s_start[256];          /* first tile-sorted slot of the digit */
	X3_LDS uint32_t s_gbase[256];          /* output index of tile-sorted slot j of digit d = s_gbase[d] + j */
	X3_LDS uint32_t s_w[X3P_WAVES];
	X3_LDS uint32_t s_k[X3P_TILE], s_v[X3P_TILE];
	const uint32_t lane = x3_lane(), wv = threadIdx.x / X3_WAVE;
	const size_t base = (size_t)blockIdx.x * X3P_TILE;
	const uint32_t cnt = a.n - base < X3P_TILE ? (uint32_t)(a.n - base) : X3P_TILE;
	for (uint32_t i = threadIdx.x; i < 256u * X3P_CS; i += X3P_THREADS) s_cnt[i] = 0u;
	/* tile order: (wave, element, lane) -- a wavefront owns X3P_E * 64 consecutive pairs, 64 consecutive ones per element slot: coalesced loads */
	uint32_t k[X3P_E], v[X3P_E], rank[X3P_E];
	const uint32_t w0 = wv * (X3P_E * X3_WAVE) + lane;
	for (uint32_t e = 0; e < X3P_E; e++) {
		const uint32_t t = w0 + e * X3_WAVE;
		k[e] = t < cnt ? a.kin[base + t] : 0u;
		v[e] = t < cnt ? a.vin[base + t] : 0u;
	}
	__syncthreads();
	for (uint32_t e = 0; e < X3P_E; e++) {
		const bool valid = w0 + e * X3_WAVE < cnt;
		const uint32_t d = (k[e] >> a.shift) & a.mask;
		uint32_t mlo, mhi;
		seg_match<8>(d, valid, mlo, mhi);
		const uint32_t lower = seg_lower(mlo, mhi), size = seg_size(mlo, mhi);
		const uint32_t old = valid ? s_cnt[d * X3P_CS + wv] : 0u;
		rank[e] = old + lower;
		x3_wave_sync(); /* every lane has read the counter ... */
		if (valid && lower == 0) s_cnt[d * X3P_CS + wv] = old + size;
		x3_wave_sync(); /* ... and the next element slot sees the update */
	}
	__syncthreads();
	/* per digit: the waves' counts become "pairs of this digit in the waves before", the digit's total goes through a workgroup scan */
	uint32_t tot = 0;
	if (threadIdx.x < 256u) for (uint32_t w = 0; w < X3P_WAVES; w++) { const uint32_t c = s_cnt[threadIdx.x * X3P_CS + w]; s_cnt[threadIdx.x * X3P_CS + w] = tot; tot += c; }
	uint32_t all;
	const uint32_t start = x3p_block_excl<X3P_SUM>(tot, s_w, &all);
	if (threadIdx.x < 256u) {
		s_start[threadIdx.x] = start;
		s_gbase[threadIdx.x] = (threadIdx.x < a.ndig ? a.counts[(size_t)threadIdx.x * a.ntiles + blockIdx.x] : 0u) - start;
	}
	__syncthreads();
	for (uint32_t e = 0; e < X3P_E; e++) {
		if (w0 + e * X3_WAVE >= cnt) continue;
		const uint32_t d = (k[e] >> a.shift) & a.mask;
		const uint32_t slot = s_start[d] + s_cnt[d * X3P_CS + wv] + rank[e];
		s_k[slot] = k[e]; s_v[slot] = v[e];
	}
	__syncthreads();
	for (uint32_t j = threadIdx.x; j < cnt; j += X3P_THREADS) {
		const uint32_t kk = s_k[j], d = (kk >> a.shift) & a.mask;
		const size_t o = (size_t)(s_gbase[d] + j); /* (32-bit wrap-around arithmetic: base - start + j) */
		a.kout[o] = kk; a.vout[o] = s_v[j];
	}
}

#ifndef X3_EMU
template <int MODE> __global__ void __launch_bounds__(X3P_THREADS) x3p_scan_kernel(X3pScanArgs a) { x3p_scan_body<MODE>(a); }
__global__ void __launch_bounds__(X3P_THREADS) x3p_hist_kernel(X3pSortArgs a) { x3p_hist_body(a); }
__global__ void __launch_bounds__(X3P_THREADS) x3p_scatter_kernel(X3pSortArgs a) { x3p_scatter_body(a); }
template <int MODE> static void x3p_scan_launch(const X3pScanArgs &a, hipStream_t st) { hipLaunchKernelGGL(x3p_scan_kernel<MODE>, dim3(a.ntiles), dim3(X3P_THREADS), 0, st, a); }
static void x3p_hist_launch(const X3pSortArgs &a, hipStream_t st) { hipLaunchKernelGGL(x3p_hist_kernel, dim3(a.ntiles), dim3(X3P_THREADS), 0, st, a); }
static void x3p_scatter_launch(const X3pSortArgs &a, hipStream_t st) { hipLaunchKernelGGL(x3p_scatter_kernel, dim3(a.ntiles), dim3(X3P_THREADS), 0, st, a); }
#else
template <int MODE> static void x3p_scan_tramp(void *p) { x3p_scan_body<MODE>(*(const X3pScanArgs *)p); }
static void x3p_hist_tramp(void *p) { x3p_hist_body(*(const X3pSortArgs *)p); }
static void x3p_scatter_tramp(void *p) { x3p_scatter_body(*(const X3pSortArgs *)p); }
template <int MODE> static void x3p_scan_launch(const X3pScanArgs &a, hipStream_t) { x3emu_launch(x3p_scan_tramp<MODE>, (void *)&a, dim3(a.ntiles), dim3(X3P_THREADS)); }
static void x3p_hist_launch(const X3pSortArgs &a, hipStream_t) { x3emu_launch(x3p_hist_tramp, (void *)&a, dim3(a.ntiles), dim3(X3P_THREADS)); }
static void x3p_scatter_launch(const X3pSortArgs &a, hipStream_t) { x3emu_launch(x3p_scatter_tramp, (void *)&a, dim3(a.ntiles), dim3(X3P_THREADS)); }
#endif

#ifdef X3_EMU
/* The emulator runs a 512-fiber workgroup per tile and launch: with ~40 sorts and scans per coding stage the CPU suite would spend its time here.  So the emulated
 * library answers these calls with the obvious host loops unless X3_EMU_PRIM_KERNELS=1 asks for the kernels above (tests/test_emu_kernels.py runs both). */
#include <algorithm>
static bool x3p_emu_kernels() { const char *e = getenv("X3_EMU_PRIM_KERNELS"); return e && e[0] == '1'; }
static void x3p_host_sort(const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit, int end_bit)
{
	const uint32_t mask = (end_bit - begin_bit) >= 32 ? 0xFFFFFFFFu : ((1u << (end_bit - begin_bit)) - 1);
	std::vector<size_t> idx(n);
	for (size_t i = 0; i < n; i++) idx[i] = i;
	std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return ((kin[x] >> begin_bit) & mask) < ((kin[y] >> begin_bit) & mask); });
	std::vector<uint32_t> k(n), v(n);
	for (size_t i = 0; i < n; i++) { k[i] = kin[idx[i]]; v[i] = vin[idx[i]]; }
	for (size_t i = 0; i < n; i++) { kout[i] = k[i]; vout[i] = v[i]; }
}
#endif

static inline size_t x3p_tiles(size_t n) { return (n + X3P_TILE - 1) / X3P_TILE; }
static inline size_t x3p_al(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

/* `status` needs x3p_status_bytes(nout) bytes */
static inline size_t x3p_status_bytes(size_t nout) { return x3p_al((x3p_tiles(nout) + 1) * 8); }
template <int MODE> static int x3p_scan_run(const uint32_t *in, uint32_t *out, size_t n, size_t nout, void *status, hipStream_t st)
{
	if (!nout) return X3H_OK;
	if (x3p_tiles(nout) > 0x7FFFFFFFull) return X3H_E_ARG;
	X3pScanArgs a;
	a.in = in; a.out = out; a.status = (uint64_t *)status; a.n = n; a.nout = nout; a.ntiles = (uint32_t)x3p_tiles(nout); a._pad = 0;
	HIPCHK(hipMemsetAsync(status, 0, ((size_t)a.ntiles + 1) * 8, st));
	x3p_scan_launch<MODE>(a, st);
	HIPCHK(hipGetLastError());
	return X3H_OK;
}

int x3p_excl_scan(DevBuf &tmp, const uint32_t *in, uint32_t *out, size_t n, hipStream_t st)
{
#ifdef X3_EMU
	if (!x3p_emu_kernels()) { uint32_t acc = 0; for (size_t i = 0; i <= n; i++) { const uint32_t v = i < n ? in[i] : 0; out[i] = acc; acc += v; } return X3H_OK; }
#endif
	CHK(tmp.reserve(x3p_status_bytes(n + 1)));
	return x3p_scan_run<X3P_SUM>(in, out, n, n + 1, tmp.p, st);
}

int x3p_excl_scan_top_bit_w(DevBuf &tmp, const uint4 *rec, uint32_t *out, size_t n, hipStream_t st)
{
#ifdef X3_EMU
	if (!x3p_emu_kernels()) { uint32_t acc = 0; for (size_t i = 0; i <= n; i++) { out[i] = acc; if (i < n) acc += rec[i].w >> 31; } return X3H_OK; }
#endif
	CHK(tmp.reserve(x3p_status_bytes(n + 1)));
	return x3p_scan_run<X3P_TOPBIT>((const uint32_t *)rec, out, n, n + 1, tmp.p, st);
}

int x3p_incl_max_scan(DevBuf &tmp, const uint32_t *in, uint32_t *out, size_t n, hipStream_t st)
{
	if (!n) return X3H_OK;
#ifdef X3_EMU
	if (!x3p_emu_kernels()) { uint32_t acc = 0; for (size_t i = 0; i < n; i++) { acc = in[i] > acc ? in[i] : acc; out[i] = acc; } return X3H_OK; }
#endif
	CHK(tmp.reserve(x3p_status_bytes(n)));
	return x3p_scan_run<X3P_MAX>(in, out, n, n, tmp.p, st);
}

/* stable sort on key bits [begin_bit, end_bit): ceil(bits / 8) passes; the pairs travel kin -> (scratch <->) kout so that the last pass writes kout / vout */
int x3p_sort_pairs_bits(DevBuf &tmp, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit, int end_bit, hipStream_t st)
{
	if (!n) return X3H_OK;
	if (begin_bit < 0) begin_bit = 0;
	if (end_bit > 32) end_bit = 32;
	if (end_bit <= begin_bit) end_bit = begin_bit + 1;
#ifdef X3_EMU
	if (!x3p_emu_kernels()) { x3p_host_sort(kin, kout, vin, vout, n, begin_bit, end_bit); return X3H_OK; }
#endif
	if (n >= ((size_t)1 << 32) - X3P_TILE) return X3H_E_ARG; /* output indices are 32-bit */
	const uint32_t npass = ((uint32_t)(end_bit - begin_bit) + 7u) / 8u;
	const size_t nt = x3p_tiles(n), ncounts = 256 * nt + 1;
	const size_t o_counts = 0, o_part = o_counts + x3p_al(ncounts * 4), o_k = o_part + x3p_status_bytes(ncounts), o_v = o_k + x3p_al(npass > 1 ? n * 4 : 0);
	CHK(tmp.reserve(o_v + x3p_al(npass > 1 ? n * 4 : 0)));
	uint8_t *t8 = tmp.as<uint8_t>();
	uint32_t *counts = (uint32_t *)(t8 + o_counts); void *part = t8 + o_part; uint32_t *sk = (uint32_t *)(t8 + o_k), *sv = (uint32_t *)(t8 + o_v);
	const uint32_t *ck = kin, *cv = vin;
	for (uint32_t p = 0; p < npass; p++) {
		const uint32_t lo = (uint32_t)begin_bit + 8u * p, width = (uint32_t)end_bit - lo < 8u ? (uint32_t)end_bit - lo : 8u;
		const bool to_out = ((npass - 1u - p) & 1u) == 0u;
		X3pSortArgs a;
		a.kin = ck; a.vin = cv; a.kout = to_out ? kout : sk; a.vout = to_out ? vout : sv; a.counts = counts; a.n = n;
		a.shift = lo; a.mask = (1u << width) - 1u; a.ntiles = (uint32_t)nt; a.ndig = 1u << width;
		x3p_hist_launch(a, st);
		HIPCHK(hipGetLastError());
		const size_t used = (size_t)a.ndig * nt; /* [digit][tile], scanned in place */
		CHK(x3p_scan_run<X3P_SUM>(counts, counts, used, used + 1, part, st));
		x3p_scatter_launch(a, st);
		HIPCHK(hipGetLastError());
		ck = a.kout; cv = a.vout;
	}
	return X3H_OK;
}

int x3p_sort_pairs(DevBuf &tmp, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, int bits, hipStream_t st)
{
	if (bits < 1) bits = 1;
	return x3p_sort_pairs_bits(tmp, kin, kout, vin, vout, n, 0, bits > 32 ? 32 : bits, st);
}

/* X3H_OK, or X3H_E_INTERNAL if a scan tile of any call since the last check gave up its bounded wait (the word is cleared again).  Synchronises `st`. */
int x3p_check_error(hipStream_t st)
{
	uint32_t e = 0;
#ifndef X3_EMU
	HIPCHK(hipMemcpyFromSymbolAsync(&e, HIP_SYMBOL(x3p_dev_error), 4, 0, hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	if (e) { const uint32_t z = 0; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(x3p_dev_error), &z, 4, 0, hipMemcpyHostToDevice)); }
#else
	(void)st; e = x3p_dev_error; x3p_dev_error = 0;
#endif
	return e ? X3H_E_INTERNAL : X3H_OK;
}
