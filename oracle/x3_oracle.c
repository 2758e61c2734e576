/*
 * x3_oracle.c -- CPU restatement of the x3 hot path.  TEST INFRASTRUCTURE ONLY (see x3_oracle.h).
 *
 * Parity status: PINNED against the real reference (oracle/_ref/x3, built from /root/reference by
 * oracle/Makefile) and against the fixtures it generated under tests/golden/.
 *
 * Written from the algorithm, not from the reference text: instance state instead of globals, a
 * move-to-front array instead of a re-sort, on-the-fly cumulative frequencies instead of cached
 * tables, a hash map instead of an unbalanced BST.  Each block cites what it restates.
 */
#include "x3_oracle.h"

#include <stdlib.h>
#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * bit I/O -- bio.c:49-72 (write), bio.c:30-42,74-103 (read), bio.c:105-112 (close)
 * Bit k of the stream is bit (k mod 32), LSB first, of little-endian 32-bit word k/32.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
	uint8_t *ptr, *end;
	uint32_t acc;
	unsigned cnt;
	int full;
} bitw;

static void bw_flush_word(bitw *w)
{
	if (w->end - w->ptr < 4) { w->full = 1; w->acc = 0; w->cnt = 0; return; }
	w->ptr[0] = (uint8_t)(w->acc);
	w->ptr[1] = (uint8_t)(w->acc >> 8);
	w->ptr[2] = (uint8_t)(w->acc >> 16);
	w->ptr[3] = (uint8_t)(w->acc >> 24);
	w->ptr += 4;
	w->acc = 0;
	w->cnt = 0;
}

static void bw_put(bitw *w, unsigned bit)
{
	w->acc |= (uint32_t)(bit & 1u) << w->cnt;
	if (++w->cnt == 32) bw_flush_word(w);
}

static void bw_close(bitw *w) /* bio.c:105-112: a partial word is written whole */
{
	if (w->cnt > 0) bw_flush_word(w);
}

typedef struct {
	const uint8_t *ptr, *end;
	uint32_t acc;
	unsigned cnt;
} bitr;

static unsigned br_get(bitr *r)
{
	if (r->cnt == 32) {
		/* bio.c:10 keeps end-3, bio.c:35-39: past the last whole word the reader feeds 0x80000000 */
		if (r->end - r->ptr >= 4) {
			r->acc = (uint32_t)r->ptr[0] | (uint32_t)r->ptr[1] << 8 | (uint32_t)r->ptr[2] << 16 | (uint32_t)r->ptr[3] << 24;
			r->ptr += 4;
		} else {
			r->acc = 0x80000000u;
		}
		r->cnt = 0;
	}
	unsigned b = r->acc & 1u;
	r->acc >>= 1;
	r->cnt++;
	return b;
}

/* ------------------------------------------------------------------------------------------------
 * arithmetic coder -- ac.c:31-41 (constants/init), 46-75 (encoder renormalisation), 77-85 (encode),
 * 115-126 (flush), 128-198 (decoder)
 * ---------------------------------------------------------------------------------------------- */
#define AC_Q1   0x20000000ull
#define AC_HALF 0x40000000ull
#define AC_Q3   0x60000000ull

typedef struct {
	uint64_t lo, hi, pending, buf;
} accoder;

static void ac_reset(accoder *a) { a->lo = 0; a->hi = 0x7FFFFFFFull; a->pending = 0; a->buf = 0; }

static void ac_enc(accoder *a, bitw *w, uint64_t cum_lo, uint64_t cum_hi, uint64_t total)
{
	uint64_t step = (a->hi - a->lo + 1) / total;
	a->hi = a->lo + step * cum_hi - 1;
	a->lo = a->lo + step * cum_lo;
	for (;;) { /* E1 / E2 */
		if (a->hi < AC_HALF) {
			bw_put(w, 0);
			a->lo = 2 * a->lo;
			a->hi = 2 * a->hi + 1;
			for (; a->pending > 0; a->pending--) bw_put(w, 1);
		} else if (a->lo >= AC_HALF) {
			bw_put(w, 1);
			a->lo = 2 * (a->lo - AC_HALF);
			a->hi = 2 * (a->hi - AC_HALF) + 1;
			for (; a->pending > 0; a->pending--) bw_put(w, 0);
		} else {
			break;
		}
	}
	while (a->lo >= AC_Q1 && a->hi < AC_Q3) { /* E3 */
		a->pending++;
		a->lo = 2 * (a->lo - AC_Q1);
		a->hi = 2 * (a->hi - AC_Q1) + 1;
	}
}

static void ac_enc_flush(accoder *a, bitw *w) /* ac.c:115-126; pending bits are dropped in the else arm */
{
	if (a->lo < AC_Q1) {
		bw_put(w, 0);
		for (uint64_t i = 0; i < a->pending + 1; i++) bw_put(w, 1);
	} else {
		bw_put(w, 1);
	}
}

/* Stage oracle for the coder recurrence alone: the interval (mLow, mHigh) after each of n symbols given as (cum_lo, freq, total),
 * starting from ac_init's [0, 0x7FFFFFFF] (ac.c:35-41,46-85), and the bits the symbols shift out (pending bits included). */
int x3o_ac_chain(const uint32_t *cum, const uint32_t *freq, const uint32_t *total, size_t n, uint32_t *lo_out, uint32_t *hi_out,
                 uint8_t *bits, size_t cap, size_t *nbits)
{
	accoder a;
	bitw w;
	ac_reset(&a);
	w.ptr = bits; w.end = bits + cap; w.acc = 0; w.cnt = 0; w.full = 0;
	for (size_t i = 0; i < n; i++) {
		if (!total[i] || !freq[i] || (uint64_t)cum[i] + freq[i] > total[i]) return X3O_E_ARG;
		ac_enc(&a, &w, cum[i], (uint64_t)cum[i] + freq[i], total[i]);
		lo_out[i] = (uint32_t)a.lo;
		hi_out[i] = (uint32_t)a.hi;
	}
	if (nbits) *nbits = (size_t)(w.ptr - bits) * 8 + w.cnt;
	bw_close(&w);
	return w.full ? X3O_E_FULL : X3O_OK;
}

static void ac_dec_start(accoder *a, bitr *r) /* ac.c:133-140 */
{
	a->buf = 0;
	for (int i = 0; i < 31; i++) a->buf = (a->buf << 1) | br_get(r);
}

static void ac_dec_narrow(accoder *a, bitr *r, uint64_t step, uint64_t cum_lo, uint64_t cum_hi) /* ac.c:192-195,142-165 */
{
	a->hi = a->lo + step * cum_hi - 1;
	a->lo = a->lo + step * cum_lo;
	for (;;) {
		if (a->hi < AC_HALF) {
			a->lo = 2 * a->lo;
			a->hi = 2 * a->hi + 1;
			a->buf = 2 * a->buf + br_get(r);
		} else if (a->lo >= AC_HALF) {
			a->lo = 2 * (a->lo - AC_HALF);
			a->hi = 2 * (a->hi - AC_HALF) + 1;
			a->buf = 2 * (a->buf - AC_HALF) + br_get(r);
		} else {
			break;
		}
	}
	while (a->lo >= AC_Q1 && a->hi < AC_Q3) {
		a->lo = 2 * (a->lo - AC_Q1);
		a->hi = 2 * (a->hi - AC_Q1) + 1;
		a->buf = 2 * (a->buf - AC_Q1) + br_get(r);
	}
}

/* ------------------------------------------------------------------------------------------------
 * adaptive order-0 frequency model -- ac.c:215-266 (struct model).  Symbols are 0..count-1, every
 * symbol starts at freq 1, +1 per use, never rescaled.  cum_freq is recomputed on demand.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
	uint32_t *freq;
	size_t count, cap;
	uint64_t total;
} fmodel;

static int fm_init(fmodel *m, size_t count)
{
	m->cap = count > 16 ? count : 16;
	m->freq = (uint32_t *)malloc(m->cap * sizeof(uint32_t));
	if (!m->freq) return -1;
	for (size_t i = 0; i < count; i++) m->freq[i] = 1;
	m->count = count;
	m->total = count;
	return 0;
}

static int fm_grow(fmodel *m) /* model_enlarge, ac.c:250-266 */
{
	if (m->count == m->cap) {
		size_t ncap = m->cap * 2;
		uint32_t *nf = (uint32_t *)realloc(m->freq, ncap * sizeof(uint32_t));
		if (!nf) return -1;
		m->freq = nf;
		m->cap = ncap;
	}
	m->freq[m->count++] = 1;
	m->total += 1;
	return 0;
}

static uint64_t fm_cum(const fmodel *m, size_t sym)
{
	uint64_t c = 0;
	for (size_t i = 0; i < sym; i++) c += m->freq[i];
	return c;
}

static void fm_inc(fmodel *m, size_t sym) { m->freq[sym] += 1; m->total += 1; } /* inc_model, ac.c:215-228 */

static float fm_prob(const fmodel *m, size_t sym) /* ac.c:108-113: (float)freq / total */
{
	return (float)m->freq[sym] / (float)m->total;
}

static void fm_encode(accoder *a, bitw *w, const fmodel *m, size_t sym)
{
	uint64_t c = fm_cum(m, sym);
	ac_enc(a, w, c, c + m->freq[sym], m->total);
}

static size_t fm_decode(accoder *a, bitr *r, const fmodel *m, int *err) /* ac.c:167-198 */
{
	uint64_t step = (a->hi - a->lo + 1) / m->total;
	uint64_t value = (a->buf - a->lo) / step;
	uint64_t c = 0;
	for (size_t i = 0; i < m->count; i++) {
		if (value >= c && value < c + m->freq[i]) {
			ac_dec_narrow(a, r, step, c, c + m->freq[i]);
			return i;
		}
		c += m->freq[i];
	}
	*err = 1; /* the reference abort()s here (ac.c:178) */
	return 0;
}

/* ------------------------------------------------------------------------------------------------
 * context = list of (tag,freq) in first-seen order -- context.c:20-56,88-152
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t tag, freq; } citem;
typedef struct { citem *arr; uint32_t items, cap; uint64_t total; } cctx;

static long cx_find(const cctx *c, uint32_t tag)
{
	for (uint32_t i = 0; i < c->items; i++)
		if (c->arr[i].tag == tag) return (long)i;
	return -1;
}

static int cx_touch(cctx *c, uint32_t tag) /* x3.c:197-209: add with freq 1, else freq++ */
{
	long i = cx_find(c, tag);
	if (i >= 0) {
		c->arr[i].freq++;
	} else {
		if (c->items == c->cap) {
			uint32_t ncap = c->cap ? c->cap * 2 : 4;
			citem *na = (citem *)realloc(c->arr, ncap * sizeof(citem));
			if (!na) return -1;
			c->arr = na;
			c->cap = ncap;
		}
		c->arr[c->items].tag = tag;
		c->arr[c->items].freq = 1;
		c->items++;
	}
	c->total++;
	return 0;
}

static uint64_t cx_cum(const cctx *c, uint32_t pos)
{
	uint64_t s = 0;
	for (uint32_t i = 0; i < pos; i++) s += c->arr[i].freq;
	return s;
}

/* a growable vector of contexts, zero-initialised on growth (ctx_enlarge, context.c:7-18) */
typedef struct { cctx *v; size_t size; } cvec;

static cctx *cv_at(cvec *cv, size_t idx)
{
	if (idx >= cv->size) {
		size_t nsize = cv->size ? cv->size : 2;
		while (nsize <= idx) nsize *= 2;
		cctx *nv = (cctx *)realloc(cv->v, nsize * sizeof(cctx));
		if (!nv) return NULL;
		memset(nv + cv->size, 0, (nsize - cv->size) * sizeof(cctx));
		cv->v = nv;
		cv->size = nsize;
	}
	return &cv->v[idx];
}

static void cv_free(cvec *cv)
{
	for (size_t i = 0; i < cv->size; i++) free(cv->v[i].arr);
	free(cv->v);
	cv->v = NULL;
	cv->size = 0;
}

/* ------------------------------------------------------------------------------------------------
 * tag-pair map: (tag0,tag1) -> dense ordinal in insertion order -- tag_pair.c:67-84,100-130
 * (exact map; the reference's BST shape is irrelevant to the stream)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint64_t *key; uint32_t *val; size_t cap, n; } pairmap;

static size_t pm_slot(const pairmap *pm, uint64_t k)
{
	uint64_t h = k * 0x9E3779B97F4A7C15ull;
	size_t i = (size_t)(h >> 20) & (pm->cap - 1);
	while (pm->key[i] != 0 && pm->key[i] != k) i = (i + 1) & (pm->cap - 1);
	return i;
}

static int pm_reserve(pairmap *pm)
{
	if (pm->cap && (pm->n + 1) * 2 <= pm->cap) return 0;
	size_t ncap = pm->cap ? pm->cap * 2 : 1024;
	pairmap np;
	np.key = (uint64_t *)calloc(ncap, sizeof(uint64_t));
	np.val = (uint32_t *)malloc(ncap * sizeof(uint32_t));
	np.cap = ncap;
	np.n = pm->n;
	if (!np.key || !np.val) { free(np.key); free(np.val); return -1; }
	for (size_t i = 0; i < pm->cap; i++) {
		if (pm->key[i]) {
			size_t s = pm_slot(&np, pm->key[i]);
			np.key[s] = pm->key[i];
			np.val[s] = pm->val[i];
		}
	}
	free(pm->key);
	free(pm->val);
	*pm = np;
	return 0;
}

#define PM_KEY(t0, t1) ((((uint64_t)(t0)) << 32 | (uint64_t)(t1)) + 1) /* never 0 for tags < 2^32-1 */

static long pm_get(const pairmap *pm, uint32_t t0, uint32_t t1)
{
	if (!pm->cap) return -1;
	size_t s = pm_slot(pm, PM_KEY(t0, t1));
	return pm->key[s] ? (long)pm->val[s] : -1;
}

static int pm_add_if_absent(pairmap *pm, uint32_t t0, uint32_t t1)
{
	if (pm_reserve(pm)) return -1;
	size_t s = pm_slot(pm, PM_KEY(t0, t1));
	if (!pm->key[s]) {
		pm->key[s] = PM_KEY(t0, t1);
		pm->val[s] = (uint32_t)pm->n++;
	}
	return 0;
}

/* ------------------------------------------------------------------------------------------------
 * dictionary -- dict.h:7-13, dict.c:82-157.  The array is kept in recency order: index 0 is the
 * most recently used/inserted element, which is what the reference's qsort by cost
 * (dict.c:132-146) produces because all costs are distinct.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint8_t s[X3O_MAX_MATCH]; uint32_t len, tag; } delem;
typedef struct { delem *e; size_t n, cap; } dict;

static long dict_longest(const dict *d, const uint8_t *p) /* dict_find_match, dict.c:105-130 */
{
	size_t best = 0;
	long besti = -1;
	for (size_t i = 0; i < d->n; i++) {
		if (d->e[i].len > best && memcmp(p, d->e[i].s, d->e[i].len) == 0) {
			best = d->e[i].len;
			besti = (long)i;
		}
	}
	return besti;
}

static int dict_has(const dict *d, const uint8_t *p, size_t len) /* dict_query_elem, dict.c:148-157 */
{
	for (size_t i = 0; i < d->n; i++)
		if (d->e[i].len == len && memcmp(d->e[i].s, p, len) == 0) return 1;
	return 0;
}

static void dict_to_front(dict *d, size_t idx) /* dict_update_costs after dict_set_last_pos == move-to-front */
{
	if (idx == 0) return;
	delem t = d->e[idx];
	memmove(&d->e[1], &d->e[0], idx * sizeof(delem));
	d->e[0] = t;
}

static int dict_push_front(dict *d, const uint8_t *p, size_t len) /* dict_insert_elem + re-sort: new element has the smallest cost */
{
	if (d->n == d->cap) {
		size_t ncap = d->cap ? d->cap * 2 : 64;
		delem *ne = (delem *)realloc(d->e, ncap * sizeof(delem));
		if (!ne) return -1;
		d->e = ne;
		d->cap = ncap;
	}
	memmove(&d->e[1], &d->e[0], d->n * sizeof(delem));
	memset(d->e[0].s, 0, X3O_MAX_MATCH);
	memcpy(d->e[0].s, p, len);
	d->e[0].len = (uint32_t)len;
	d->e[0].tag = (uint32_t)d->n; /* tag = insertion ordinal, dict.c:100 */
	d->n++;
	return 0;
}

static long dict_index_of_tag(const dict *d, uint32_t tag) /* dict.c:174-183 */
{
	for (size_t i = 0; i < d->n; i++)
		if (d->e[i].tag == tag) return (long)i;
	return -1;
}

/* ------------------------------------------------------------------------------------------------
 * codec state -- the reference's globals (x3.c:19-20,45-50; dict.c:8-14; tag_pair.c:7-9)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
	x3o_params prm;
	dict d;
	cvec ctx0, ctx1;
	pairmap pairs;
	fmodel m_events, m_len, m_chars, m_index;
	accoder ac;
	uint64_t events[X3O_E_LAST];
	float sizes[4]; /* x3.c:43 */
	int oom;
} codec;

static int codec_init(codec *c, const x3o_params *prm) /* create(), x3.c:225-249 */
{
	memset(c, 0, sizeof(*c));
	if (prm) c->prm = *prm; else x3o_default_params(&c->prm);
	if (fm_init(&c->m_events, X3O_E_LAST)) return -1;
	c->m_events.freq[X3O_E_CTX0] = 1024; /* x3.c:239-242 */
	c->m_events.freq[X3O_E_CTX1] = 1024;
	c->m_events.freq[X3O_E_IDX1] = 1;
	c->m_events.freq[X3O_E_NEW] = 1;
	c->m_events.total = 1024 + 1024 + 1 + 1 + 1;
	if (fm_init(&c->m_len, X3O_MAX_MATCH)) return -1;
	if (fm_init(&c->m_chars, 256)) return -1;
	if (fm_init(&c->m_index, 0)) return -1;
	if (!cv_at(&c->ctx0, 1) || !cv_at(&c->ctx1, 1)) return -1; /* both start with 2 zeroed contexts */
	ac_reset(&c->ac);
	return 0;
}

static void codec_free(codec *c)
{
	free(c->d.e);
	cv_free(&c->ctx0);
	cv_free(&c->ctx1);
	free(c->pairs.key);
	free(c->pairs.val);
	free(c->m_events.freq);
	free(c->m_len.freq);
	free(c->m_chars.freq);
	free(c->m_index.freq);
}

/* after coding a hit: both contexts learn the tag, and (context1,tag) becomes a known pair -- x3.c:195-222 */
static void learn_tag(codec *c, cctx *c0, cctx *c1, uint32_t context1, uint32_t tag)
{
	if (cx_touch(c0, tag)) c->oom = 1;
	if (cx_touch(c1, tag)) c->oom = 1;
	if (pm_add_if_absent(&c->pairs, context1, tag)) c->oom = 1;
}

static void pick_contexts(codec *c, uint32_t prev_context1, uint32_t context1, cctx **c0, cctx **c1)
{
	long id = pm_get(&c->pairs, prev_context1, context1); /* x3.c:141-145: unknown pair -> context 0 */
	if (id < 0) id = 0;
	*c0 = cv_at(&c->ctx0, (size_t)id);
	*c1 = cv_at(&c->ctx1, context1);
	if (!*c0 || !*c1) c->oom = 1;
}

/* encode_tag, x3.c:132-223 */
static void put_hit(codec *c, bitw *w, uint32_t prev_context1, uint32_t context1, size_t index)
{
	uint32_t tag = c->d.e[index].tag;
	cctx *c0, *c1;
	pick_contexts(c, prev_context1, context1, &c0, &c1);
	if (c->oom) return;

	long i0 = cx_find(c0, tag), i1 = cx_find(c1, tag);

	/* x3.c:152-160 -- IEEE single: (float)freq/total, then one multiply */
	float p_ctx0 = 0.f, p_ctx1 = 0.f;
	if (i0 >= 0) p_ctx0 = fm_prob(&c->m_events, X3O_E_CTX0) * ((float)c0->arr[i0].freq / (float)c0->total);
	if (i1 >= 0) p_ctx1 = fm_prob(&c->m_events, X3O_E_CTX1) * ((float)c1->arr[i1].freq / (float)c1->total);
	float p_idx1 = fm_prob(&c->m_events, X3O_E_IDX1) * fm_prob(&c->m_index, index);

	int mode = X3O_E_IDX1; /* x3.c:162-172: order IDX1 -> CTX0 -> CTX1, strict > */
	float best = p_idx1;
	if (p_ctx0 > best) { mode = X3O_E_CTX0; best = p_ctx0; }
	if (p_ctx1 > best) { mode = X3O_E_CTX1; best = p_ctx1; }

	fm_encode(&c->ac, w, &c->m_events, (size_t)mode);
	fm_inc(&c->m_events, (size_t)mode);

	if (mode == X3O_E_CTX0) { /* context.c:95-112: symbol = list position under the item freqs */
		uint64_t cum = cx_cum(c0, (uint32_t)i0);
		ac_enc(&c->ac, w, cum, cum + c0->arr[i0].freq, c0->total);
	} else if (mode == X3O_E_CTX1) {
		uint64_t cum = cx_cum(c1, (uint32_t)i1);
		ac_enc(&c->ac, w, cum, cum + c1->arr[i1].freq, c1->total);
	} else {
		fm_encode(&c->ac, w, &c->m_index, index);
		fm_inc(&c->m_index, index);
	}
	c->events[mode]++;
	c->sizes[mode] += -log2f(best); /* x3.c:52-55,192-193: prob_to_bits of the chosen product, summed in float */
	learn_tag(c, c0, c1, context1, tag);
}

/* encode_match, x3.c:251-270 */
static void put_new(codec *c, bitw *w, const uint8_t *p, size_t len)
{
	c->sizes[X3O_E_NEW] += -log2f(fm_prob(&c->m_events, X3O_E_NEW)); /* x3.c:253,259,264: one term per coded symbol, before the model learns it */
	fm_encode(&c->ac, w, &c->m_events, X3O_E_NEW);
	fm_inc(&c->m_events, X3O_E_NEW);
	c->sizes[X3O_E_NEW] += -log2f(fm_prob(&c->m_len, len - 1));
	fm_encode(&c->ac, w, &c->m_len, len - 1);
	fm_inc(&c->m_len, len - 1);
	for (size_t k = 0; k < len; k++) {
		c->sizes[X3O_E_NEW] += -log2f(fm_prob(&c->m_chars, p[k]));
		fm_encode(&c->ac, w, &c->m_chars, p[k]);
		fm_inc(&c->m_chars, p[k]);
	}
	c->events[X3O_E_NEW]++;
}

/* ------------------------------------------------------------------------------------------------
 * match finder
 * ---------------------------------------------------------------------------------------------- */
void x3o_count(const uint8_t *padded, size_t pos, uint32_t window_bytes, uint32_t count[X3O_MAX_MATCH])
{
	/* backend.c:56-74: candidates s in [p+1, p+W-33], common prefix capped at 32 */
	const uint8_t *p = padded + pos;
	for (int i = 0; i < X3O_MAX_MATCH; i++) count[i] = 0;
	if (window_bytes <= X3O_MAX_MATCH + 1) return;
	const uint8_t *end = p + window_bytes - X3O_MAX_MATCH;
	for (const uint8_t *s = p + 1; s < end; s++) {
		for (int i = 0; i < X3O_MAX_MATCH; i++) {
			if (p[i] != s[i]) break;
			count[i]++;
		}
	}
}

static uint32_t longest_len_at(const codec *c, const uint8_t *q) /* length of dict_find_match(q), 0 if none */
{
	long i = dict_longest(&c->d, q);
	return i < 0 ? 0 : c->d.e[i].len;
}

/* the dictionary-aware filters of backend.c:79-90 for candidate length i+1 at p */
static int passes_filters(const codec *c, const uint8_t *p, int i)
{
	uint32_t f1 = c->prm.factor1, f2 = c->prm.factor2;
	if (i >= 2 && f1 > 0) {
		uint64_t l = longest_len_at(c, p + i);
		if (l != 0 && l * (uint64_t)f1 > (uint64_t)(i + 1)) return 0;
	}
	if (i >= 1 && f2 > 0) {
		for (int o = 1; o <= i; o++) {
			uint32_t l = longest_len_at(c, p + o);
			if (l != 0 && ((int)l - o) * (int)f2 > i + 1) return 0;
		}
	}
	return 1;
}

/* find_best_match, backend.c:56-100, literally (double loop over tc and i) */
static size_t best_match_faithful(const codec *c, const uint8_t *padded, size_t pos)
{
	uint32_t count[X3O_MAX_MATCH];
	x3o_count(padded, pos, c->prm.window_bytes, count);
	for (int tc = c->prm.max_match_count; tc > 0; tc--) {
		for (int i = X3O_MAX_MATCH - 1; i >= 0; i--) {
			if (count[i] > (uint32_t)tc && passes_filters(c, padded + pos, i)) return (size_t)i + 1;
		}
	}
	return 1;
}

/* closed form: 1 + max{ i <= m : filters pass } (i = 0 always passes) */
static size_t best_match_from_m(const codec *c, const uint8_t *padded, size_t pos, uint8_t m)
{
	for (int i = m; i > 0; i--)
		if (passes_filters(c, padded + pos, i)) return (size_t)i + 1;
	return 1;
}

static uint8_t m_from_counts(const uint32_t count[X3O_MAX_MATCH], int32_t T)
{
	if (T <= 0 || count[0] < 2) return 0;
	uint32_t thr = count[0] - 1 < (uint32_t)T ? count[0] - 1 : (uint32_t)T;
	int m = 0;
	for (int i = 1; i < X3O_MAX_MATCH; i++)
		if (count[i] > thr) m = i;
	return (uint8_t)m;
}

static uint8_t *make_padded(const uint8_t *in, size_t n, uint32_t window)
{
	/* x3.c:579,590: the input is followed by W zero bytes that take part in every comparison.
	 * (+64 so that dictionary probes near the end stay in bounds even for tiny windows.) */
	size_t pad = (size_t)window + 64;
	uint8_t *b = (uint8_t *)malloc(n + pad);
	if (!b) return NULL;
	if (n) memcpy(b, in, n);
	memset(b + n, 0, pad);
	return b;
}

int x3o_scan_m(const x3o_params *prm, const uint8_t *in, size_t n, uint8_t *m_out)
{
	x3o_params dp;
	if (!prm) { x3o_default_params(&dp); prm = &dp; }
	uint8_t *b = make_padded(in, n, prm->window_bytes);
	if (!b) return X3O_E_NOMEM;
	uint32_t count[X3O_MAX_MATCH];
	for (size_t p = 0; p < n; p++) {
		x3o_count(b, p, prm->window_bytes, count);
		m_out[p] = m_from_counts(count, prm->max_match_count);
	}
	free(b);
	return X3O_OK;
}

static size_t nl_map(const codec *c, size_t len) /* nl(), x3.c:357-370 */
{
	if (!c->prm.nl_mode) return len;
	switch (len - 1) {
		case 0: return 1;
		case 1: return 4;
		case 2: return 6;
		case 3: return 8;
		default: return 9999;
	}
}

/* ------------------------------------------------------------------------------------------------
 * compress -- x3.c:372-434, then ac_encode_flush + bio_close (x3.c:603-604)
 * ---------------------------------------------------------------------------------------------- */
static int compress_impl(const x3o_params *prm, const uint8_t *in, size_t n, const uint8_t *m,
                         uint8_t *out, size_t cap, size_t *out_len, x3o_stats *stats,
                         uint32_t *tok_pos, uint32_t *tok_info, size_t tok_cap, size_t *ntok)
{
	if ((!in && n) || !out || !out_len) return X3O_E_ARG;
	codec c;
	if (codec_init(&c, prm)) { codec_free(&c); return X3O_E_NOMEM; }
	uint8_t *b = make_padded(in, n, c.prm.window_bytes);
	if (!b) { codec_free(&c); return X3O_E_NOMEM; }

	bitw w = { out, out + cap, 0, 0, 0 };
	uint32_t prev_context1 = 0, context1 = 0;
	size_t steps = 0;

	for (size_t p = 0; p < n && !c.oom && !w.full;) {
		long index = dict_longest(&c.d, b + p);
		size_t fbm = 0;
		int hit = 0;
		if (index >= 0) { /* x3.c:383: the window scan only runs if the dictionary matched */
			fbm = m ? best_match_from_m(&c, b, p, m[p]) : best_match_faithful(&c, b, p);
			size_t len = c.d.e[index].len;
			hit = nl_map(&c, len) >= fbm && p + len <= n;
		}
		if (hit) {
			size_t len = c.d.e[index].len;
			uint32_t tag = c.d.e[index].tag;
			if (tok_pos && steps < tok_cap) { tok_pos[steps] = (uint32_t)p; tok_info[steps] = tag; }
			put_hit(&c, &w, prev_context1, context1, (size_t)index);
			prev_context1 = context1;
			context1 = tag;
			dict_to_front(&c.d, (size_t)index);
			p += len;
		} else {
			size_t len = index >= 0 ? fbm : (m ? best_match_from_m(&c, b, p, m[p]) : best_match_faithful(&c, b, p));
			if (p + len > n) len = n - p; /* x3.c:402-404 */
			put_new(&c, &w, b + p, len);
			int dup = dict_has(&c.d, b + p, len); /* x3.c:412 */
			if (tok_pos && steps < tok_cap) {
				tok_pos[steps] = (uint32_t)p;
				tok_info[steps] = X3O_TOK_MISS | (dup ? X3O_TOK_DUP : 0) | (uint32_t)len;
			}
			if (!dup) {
				if (dict_push_front(&c.d, b + p, len)) c.oom = 1;
				if (fm_grow(&c.m_index)) c.oom = 1; /* x3.c:419 */
			}
			p += len;
			prev_context1 = 0; /* x3.c:424-425 */
			context1 = 0;
		}
		steps++;
	}

	fm_encode(&c.ac, &w, &c.m_events, X3O_E_EOF); /* x3.c:432-433 */
	fm_inc(&c.m_events, X3O_E_EOF);
	ac_enc_flush(&c.ac, &w);
	bw_close(&w);

	int rc = c.oom ? X3O_E_NOMEM : (w.full ? X3O_E_FULL : X3O_OK);
	*out_len = (size_t)(w.ptr - out);
	if (ntok) *ntok = steps;
	if (stats) {
		for (int i = 0; i < X3O_E_LAST; i++) stats->events[i] = c.events[i];
		stats->dict_elems = c.d.n;
		stats->ctx0_entries = c.pairs.n;
		stats->steps = steps;
		for (int i = 0; i < 4; i++) stats->sizes[i] = c.sizes[i];
	}
	free(b);
	codec_free(&c);
	return rc;
}

int x3o_compress(const x3o_params *prm, const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len, x3o_stats *stats)
{
	return compress_impl(prm, in, n, NULL, out, cap, out_len, stats, NULL, NULL, 0, NULL);
}

int x3o_compress_trace(const x3o_params *prm, const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len,
                       x3o_stats *stats, uint32_t *tok_pos, uint32_t *tok_info, size_t tok_cap, size_t *ntok)
{
	return compress_impl(prm, in, n, NULL, out, cap, out_len, stats, tok_pos, tok_info, tok_cap, ntok);
}

int x3o_compress_via_m(const x3o_params *prm, const uint8_t *in, size_t n, const uint8_t *m,
                       uint8_t *out, size_t cap, size_t *out_len, x3o_stats *stats)
{
	if (!m && n) return X3O_E_ARG;
	return compress_impl(prm, in, n, m, out, cap, out_len, stats, NULL, NULL, 0, NULL);
}

/* ------------------------------------------------------------------------------------------------
 * decompress -- x3.c:285-353 with decode_tag (x3.c:58-129) and decode_match (x3.c:272-283)
 * ---------------------------------------------------------------------------------------------- */
static long cx_decode(accoder *a, bitr *r, const cctx *c, int *err) /* context.c:135-152 */
{
	if (c->items == 0 || c->total == 0) { *err = 1; return -1; }
	uint64_t step = (a->hi - a->lo + 1) / c->total;
	if (step == 0) { *err = 1; return -1; }
	uint64_t value = (a->buf - a->lo) / step;
	uint64_t cum = 0;
	for (uint32_t i = 0; i < c->items; i++) {
		if (value >= cum && value < cum + c->arr[i].freq) {
			ac_dec_narrow(a, r, step, cum, cum + c->arr[i].freq);
			return (long)i;
		}
		cum += c->arr[i].freq;
	}
	*err = 1;
	return -1;
}

int x3o_decompress(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len)
{
	if ((!in && n) || (!out && cap) || !out_len) return X3O_E_ARG;
	codec c;
	if (codec_init(&c, NULL)) { codec_free(&c); return X3O_E_NOMEM; }
	bitr r = { in, in + n, 0, 32 };
	ac_dec_start(&c.ac, &r);

	uint32_t prev_context1 = 0, context1 = 0;
	size_t p = 0;
	int err = 0, rc = X3O_OK;

	for (;;) {
		size_t decision = fm_decode(&c.ac, &r, &c.m_events, &err);
		if (err) { rc = X3O_E_CORRUPT; break; }
		fm_inc(&c.m_events, decision);
		if (decision == X3O_E_EOF) break;

		if (decision == X3O_E_NEW) {
			size_t len = fm_decode(&c.ac, &r, &c.m_len, &err) + 1;
			if (err) { rc = X3O_E_CORRUPT; break; }
			fm_inc(&c.m_len, len - 1);
			if (p + len > cap) { rc = X3O_E_FULL; break; }
			for (size_t k = 0; k < len; k++) {
				size_t ch = fm_decode(&c.ac, &r, &c.m_chars, &err);
				if (err) break;
				out[p + k] = (uint8_t)ch;
				fm_inc(&c.m_chars, ch);
			}
			if (err) { rc = X3O_E_CORRUPT; break; }
			if (!dict_has(&c.d, out + p, len)) {
				if (dict_push_front(&c.d, out + p, len) || fm_grow(&c.m_index)) { rc = X3O_E_NOMEM; break; }
			}
			p += len;
			prev_context1 = 0;
			context1 = 0;
			c.events[X3O_E_NEW]++;
		} else {
			if (c.d.n == 0) { rc = X3O_E_CORRUPT; break; }
			cctx *c0, *c1;
			pick_contexts(&c, prev_context1, context1, &c0, &c1);
			if (c.oom) { rc = X3O_E_NOMEM; break; }
			long index;
			uint32_t tag;
			if (decision == X3O_E_CTX0 || decision == X3O_E_CTX1) {
				const cctx *cc = decision == X3O_E_CTX0 ? c0 : c1;
				long pos = cx_decode(&c.ac, &r, cc, &err);
				if (err) { rc = X3O_E_CORRUPT; break; }
				tag = cc->arr[pos].tag;
				index = dict_index_of_tag(&c.d, tag);
				if (index < 0) { rc = X3O_E_CORRUPT; break; }
			} else {
				if (c.m_index.count == 0) { rc = X3O_E_CORRUPT; break; }
				index = (long)fm_decode(&c.ac, &r, &c.m_index, &err);
				if (err) { rc = X3O_E_CORRUPT; break; }
				fm_inc(&c.m_index, (size_t)index);
				tag = c.d.e[index].tag;
			}
			c.events[decision]++;
			learn_tag(&c, c0, c1, context1, tag);
			if (c.oom) { rc = X3O_E_NOMEM; break; }
			size_t len = c.d.e[index].len;
			if (p + len > cap) { rc = X3O_E_FULL; break; }
			memcpy(out + p, c.d.e[index].s, len);
			prev_context1 = context1;
			context1 = tag;
			dict_to_front(&c.d, (size_t)index);
			p += len;
		}
	}
	*out_len = p;
	codec_free(&c);
	return rc;
}

/* ------------------------------------------------------------------------------------------------ */
void x3o_default_params(x3o_params *prm)
{
	prm->window_bytes = 8 * 1024; /* backend.c:8  */
	prm->max_match_count = 15;    /* backend.c:21 */
	prm->factor1 = 4;             /* backend.c:33 */
	prm->factor2 = 0;             /* backend.c:34 */
	prm->nl_mode = 0;             /* x3.c:355     */
}

size_t x3o_compress_bound(size_t n)
{
	/* The reference assumes 2n (x3.c:580) and overruns on tiny inputs; a 1-byte fragment costs three
	 * coder symbols, each below 31 bits, so 12 bytes per input byte + flush/padding always suffices. */
	return 12 * n + 64;
}
