/*
 * simt.h -- the one include of every kernel source.
 *
 * Product build (hipcc --offload-arch=gfx950): this is <hip/hip_runtime.h> plus a few helpers.
 *
 * X3_EMU build (plain g++, TESTS ONLY, tests/emu/): a small functional SIMT emulator so that the very
 * same kernel sources can be executed, single-stepped and sanitised on a machine without a GPU.
 * One fiber (ucontext) per GPU thread, 64 consecutive fibers form a wave; __ballot/__shfl/__syncthreads
 * are rendezvous points.  It models semantics only (no timing, no memory model): it exists to catch logic
 * errors before a kernel is sent to a real MI355X.  Nothing in the shipped library is compiled with X3_EMU,
 * and the emulator is never a fallback: libx3hip.so has no CPU path.
 */
#ifndef X3_SIMT_H
#define X3_SIMT_H

#include <stdint.h>
#include <stddef.h>

#define X3_WAVE 64

#ifndef X3_EMU
/* ------------------------------------------------------------------------------------------------ */
#include <hip/hip_runtime.h>

#define X3_LDS __shared__

__device__ __forceinline__ unsigned x3_lane() { return threadIdx.x & (X3_WAVE - 1); }
__device__ __forceinline__ uint64_t x3_ballot(int p) { return __ballot(p); }
__device__ __forceinline__ uint32_t x3_bcast_u32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, X3_WAVE); }
/* tell the compiler a value is wave-uniform (it then lives in an SGPR and control flow on it is scalar) */
__device__ __forceinline__ uint32_t x3_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
/* v[lane] for a wave-uniform lane index: one v_readlane_b32 (SGPR result) instead of a ds_bpermute round trip */
__device__ __forceinline__ uint32_t x3_readlane_u32(uint32_t v, uint32_t lane_uniform) { return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane((int)lane_uniform)); }
/* copy of `old` with lane `lane_uniform` replaced by the wave-uniform value `val`.  (ROCm 7.2's clang has no writelane
 * builtin and v_writelane_b32 with an SGPR lane select needs M0, which the compiler reserves: v_cmp + v_cndmask it is.) */
__device__ __forceinline__ uint32_t x3_writelane_u32(uint32_t old, uint32_t val, uint32_t lane_uniform) { return x3_lane() == lane_uniform ? val : old; }
__device__ __forceinline__ uint32_t x3_shfl_up_u32(uint32_t v, unsigned d) { return (uint32_t)__shfl_up((int)v, d, X3_WAVE); }
__device__ __forceinline__ uint32_t x3_shfl_xor_u32(uint32_t v, int m) { return (uint32_t)__shfl_xor((int)v, m, X3_WAVE); }
/* inclusive prefix sum / total over the 64 lanes with DPP row shifts + row broadcasts (six VALU adds; the same sequence LLVM's
 * atomic optimizer emits for gfx9) instead of six ds_bpermute round trips */
__device__ __forceinline__ uint32_t x3_wave_incl_scan_u32(uint32_t v)
{
	int x = (int)v;
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false); /* row_shr:1 */
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false); /* row_shr:2 */
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false); /* row_shr:4 */
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false); /* row_shr:8 */
	x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); /* row_bcast:15 -> rows 1,3 */
	x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); /* row_bcast:31 -> rows 2,3 */
	return (uint32_t)x;
}
__device__ __forceinline__ uint32_t x3_wave_sum_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)x3_wave_incl_scan_u32(v), 63); }
/* v of the lane below (lane 0: 0) as one DPP move across the whole wave (wave_shr:1, a GFX9 control) */
__device__ __forceinline__ uint32_t x3_wave_shr1_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }
/* inclusive prefix sum that is only valid in lanes 0..7 (three row shifts): models with a handful of symbols */
__device__ __forceinline__ uint32_t x3_row8_incl_scan_u32(uint32_t v)
{
	int x = (int)v;
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false); /* row_shr:1 */
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false); /* row_shr:2 */
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false); /* row_shr:4 */
	return (uint32_t)x;
}
/* Orders this wave's earlier LDS/global accesses before its later ones when different lanes touch the same
 * address (the compiler only tracks per-lane dependencies). */
__device__ __forceinline__ void x3_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
/* The same ordering for a kernel whose work-group is ONE wave, as a compiler barrier only: the memory instructions of a wave are issued in
 * program order and performed in that order per address, so no s_waitcnt is needed -- while the fence above makes the compiler wait for
 * every outstanding load AND store (vmcnt(0)), i.e. a full memory round trip, wherever it stands. */
__device__ __forceinline__ void x3_wave_order() { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }
__device__ __forceinline__ uint64_t x3_clock() { return (uint64_t)clock64(); } /* s_memtime: shader cycles */
__device__ __forceinline__ int x3_popc64(uint64_t v) { return __popcll(v); }
__device__ __forceinline__ int x3_ctz64(uint64_t v) { return __ffsll((long long)v) - 1; }           /* v != 0 */
__device__ __forceinline__ int x3_ctz32(uint32_t v) { return __ffs((int)v) - 1; }                   /* v != 0 */
__device__ __forceinline__ int x3_clz64(uint64_t v) { return __clzll((long long)v); }               /* v != 0 */
__device__ __forceinline__ int x3_clz32(uint32_t v) { return __clz((int)v); }                       /* 32 for v == 0 */

#else
/* ------------------------------------------------------------------------------------------------ */
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)
#define X3_LDS static

struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 r; r.x = x; r.y = y; return r; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
struct x3emu_dim3 { unsigned x, y, z; x3emu_dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef x3emu_dim3 dim3;

extern x3emu_dim3 threadIdx, blockIdx, blockDim, gridDim; /* switched per fiber by the scheduler */

void     x3emu_syncthreads();
uint64_t x3emu_ballot(int p);
uint32_t x3emu_shfl(uint32_t v, int src_lane);

#define __syncthreads() x3emu_syncthreads()

static inline unsigned x3_lane() { return threadIdx.x & (X3_WAVE - 1); }
static inline uint64_t x3_ballot(int p) { return x3emu_ballot(p); }
static inline uint32_t x3_bcast_u32(uint32_t v, int src) { return x3emu_shfl(v, src); }
static inline uint32_t x3_uniform(uint32_t v) { return v; }
static inline uint32_t x3_readlane_u32(uint32_t v, uint32_t lane_uniform) { return x3emu_shfl(v, (int)lane_uniform); }
static inline uint32_t x3_writelane_u32(uint32_t old, uint32_t val, uint32_t lane_uniform) { return x3_lane() == lane_uniform ? val : old; }
static inline uint32_t x3_shfl_up_u32(uint32_t v, unsigned d) { int l = (int)x3_lane(); return x3emu_shfl(v, l >= (int)d ? l - (int)d : l); }
static inline uint32_t x3_shfl_xor_u32(uint32_t v, int m) { return x3emu_shfl(v, (int)x3_lane() ^ m); }
static inline uint32_t x3_wave_incl_scan_u32(uint32_t v)
{
	const int l = (int)x3_lane();
	for (int d = 1; d < X3_WAVE; d <<= 1) { const uint32_t u = x3emu_shfl(v, l >= d ? l - d : l); if (l >= d) v += u; }
	return v;
}
static inline uint32_t x3_row8_incl_scan_u32(uint32_t v) { return x3_wave_incl_scan_u32(v); }
static inline uint32_t x3_wave_shr1_u32(uint32_t v) { const int l = (int)x3_lane(); const uint32_t u = x3emu_shfl(v, l ? l - 1 : 0); return l ? u : 0u; }
static inline uint32_t x3_wave_sum_u32(uint32_t v) { return x3emu_shfl(x3_wave_incl_scan_u32(v), X3_WAVE - 1); }
static inline void x3_wave_sync() { (void)x3emu_ballot(0); }
static inline void x3_wave_order() { (void)x3emu_ballot(0); }
static inline uint64_t x3_clock() { return 0; }
static inline int x3_popc64(uint64_t v) { return __builtin_popcountll(v); }
static inline int x3_ctz64(uint64_t v) { return __builtin_ctzll(v); }
static inline int x3_ctz32(uint32_t v) { return __builtin_ctz(v); }
static inline int x3_clz64(uint64_t v) { return __builtin_clzll(v); }
static inline int x3_clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }

template <typename T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicCAS(T *p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline void __threadfence_system() {}

/* run `fn(arg)` as a kernel: grid x block fibers, blocks one after another */
void x3emu_launch(void (*fn)(void *), void *arg, dim3 grid, dim3 block);

#endif /* X3_EMU */
#endif /* X3_SIMT_H */
