/*
 * prims.hip -- the three device-wide library primitives the v2 coding stage is built from (rocPRIM, gfx950):
 * stable radix sort of (key,value) pairs, exclusive sum scan, inclusive max scan.  Plain library calls for plain
 * library jobs; every x3-specific step is a hand-written kernel in code2.hip.
 * (X3_EMU test builds replace them with the obvious host loops so the rest of code2.hip can run on the CPU emulator.)
 */
#include "x3_host.h"

#ifndef X3_EMU
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

int x3p_sort_pairs(DevBuf &tmp, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, int bits, hipStream_t st)
{
	if (!n) return X3H_OK;
	if (bits < 1) bits = 1;
	if (bits > 32) bits = 32;
	size_t need = 0;
	HIPCHK(rocprim::radix_sort_pairs(nullptr, need, kin, kout, vin, vout, n, 0u, (unsigned)bits, st));
	CHK(tmp.reserve(need));
	HIPCHK(rocprim::radix_sort_pairs(tmp.p, need, kin, kout, vin, vout, n, 0u, (unsigned)bits, st));
	return X3H_OK;
}

int x3p_sort_pairs_bits(DevBuf &tmp, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit, int end_bit, hipStream_t st)
{
	if (!n) return X3H_OK;
	size_t need = 0;
	HIPCHK(rocprim::radix_sort_pairs(nullptr, need, kin, kout, vin, vout, n, (unsigned)begin_bit, (unsigned)end_bit, st));
	CHK(tmp.reserve(need));
	HIPCHK(rocprim::radix_sort_pairs(tmp.p, need, kin, kout, vin, vout, n, (unsigned)begin_bit, (unsigned)end_bit, st));
	return X3H_OK;
}

int x3p_excl_scan(DevBuf &tmp, const uint32_t *in, uint32_t *out, size_t n, hipStream_t st)
{
	size_t need = 0;
	HIPCHK(rocprim::exclusive_scan(nullptr, need, in, out, 0u, n + 1, rocprim::plus<uint32_t>(), st));
	CHK(tmp.reserve(need));
	HIPCHK(rocprim::exclusive_scan(tmp.p, need, in, out, 0u, n + 1, rocprim::plus<uint32_t>(), st));
	return X3H_OK;
}

struct X3TopBitW { __device__ __host__ uint32_t operator()(const uint4 &r) const { return r.w >> 31; } };
int x3p_excl_scan_top_bit_w(DevBuf &tmp, const uint4 *rec, uint32_t *out, size_t n, hipStream_t st)
{
	auto in = rocprim::make_transform_iterator(rec, X3TopBitW());
	size_t need = 0;
	HIPCHK(rocprim::exclusive_scan(nullptr, need, in, out, 0u, n + 1, rocprim::plus<uint32_t>(), st));
	CHK(tmp.reserve(need));
	HIPCHK(rocprim::exclusive_scan(tmp.p, need, in, out, 0u, n + 1, rocprim::plus<uint32_t>(), st));
	return X3H_OK;
}

int x3p_incl_max_scan(DevBuf &tmp, const uint32_t *in, uint32_t *out, size_t n, hipStream_t st)
{
	if (!n) return X3H_OK;
	size_t need = 0;
	HIPCHK(rocprim::inclusive_scan(nullptr, need, in, out, n, rocprim::maximum<uint32_t>(), st));
	CHK(tmp.reserve(need));
	HIPCHK(rocprim::inclusive_scan(tmp.p, need, in, out, n, rocprim::maximum<uint32_t>(), st));
	return X3H_OK;
}

#else
#include <algorithm>
#include <vector>

int x3p_sort_pairs(DevBuf &, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, int bits, hipStream_t)
{
	const uint32_t mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1);
	std::vector<size_t> idx(n);
	for (size_t i = 0; i < n; i++) idx[i] = i;
	std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return (kin[x] & mask) < (kin[y] & mask); });
	std::vector<uint32_t> k(n), v(n);
	for (size_t i = 0; i < n; i++) { k[i] = kin[idx[i]]; v[i] = vin[idx[i]]; }
	for (size_t i = 0; i < n; i++) { kout[i] = k[i]; vout[i] = v[i]; }
	return X3H_OK;
}

int x3p_sort_pairs_bits(DevBuf &, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit, int end_bit, hipStream_t)
{
	const uint32_t mask = (end_bit - begin_bit) >= 32 ? 0xFFFFFFFFu : ((1u << (end_bit - begin_bit)) - 1);
	std::vector<size_t> idx(n);
	for (size_t i = 0; i < n; i++) idx[i] = i;
	std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return ((kin[x] >> begin_bit) & mask) < ((kin[y] >> begin_bit) & mask); });
	std::vector<uint32_t> k(n), v(n);
	for (size_t i = 0; i < n; i++) { k[i] = kin[idx[i]]; v[i] = vin[idx[i]]; }
	for (size_t i = 0; i < n; i++) { kout[i] = k[i]; vout[i] = v[i]; }
	return X3H_OK;
}

int x3p_excl_scan(DevBuf &, const uint32_t *in, uint32_t *out, size_t n, hipStream_t)
{
	uint32_t acc = 0;
	for (size_t i = 0; i <= n; i++) { uint32_t v = i < n ? in[i] : 0; out[i] = acc; acc += v; }
	return X3H_OK;
}

int x3p_excl_scan_top_bit_w(DevBuf &, const uint4 *rec, uint32_t *out, size_t n, hipStream_t)
{
	uint32_t acc = 0;
	for (size_t i = 0; i <= n; i++) { out[i] = acc; if (i < n) acc += rec[i].w >> 31; }
	return X3H_OK;
}

int x3p_incl_max_scan(DevBuf &, const uint32_t *in, uint32_t *out, size_t n, hipStream_t)
{
	uint32_t acc = 0;
	for (size_t i = 0; i < n; i++) { acc = in[i] > acc ? in[i] : acc; out[i] = acc; }
	return X3H_OK;
}
#endif
