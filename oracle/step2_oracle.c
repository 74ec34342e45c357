/* oracle/step2_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see nd_oracle.h).
 *
 * CPU restatement of the dovetail / contained filter of the `--step 2` overlapper:
 *   nd_s2_filter   lib/ovl.c:449-563  filter_ovl: per-read running state (end depths, best end identities / lengths, contained
 *                                     count, covered intervals), verdict 1 = keep the overlap (dovetail, or a read covered end to end
 *                                     within maxhan1), 0 = drop
 *   nd_s2_out_bl   lib/ovl.c:339-362  out_bl: the `.bl` table written when the run ends, one line per read, in the order the
 *                                     reference's hash table (util/khash.h 0.2.8, identity hash, triangular probing, in-place
 *                                     rehash at 77 % load) iterates -- the table below lays its keys out the same way
 *   interval lists lib/ovl.c:255-337  init_aln / fill_aln / find_alni / merge_aln / fill_alnl
 * Pinned against the compiled reference (oracle/_ref/ovlseq.so exports filter_ovl and out_bl) by
 * tests/test_oracle.py::test_step2_filter_and_bl_vs_reference.  No device path exists for this row yet.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CON_MAX 2     /* MAX_CON */
#define EDGE_BACK 10  /* EDGEBACKLEN */
#define LIST_STEP 5   /* INIT_ALNM */

typedef struct { uint32_t s, e; } span_t;

typedef struct {
	uint16_t lc, rc, cur, cap;       /* 5' / 3' depth, slot last written, list capacity (alni, alnm) */
	uint32_t con, lim, rim;          /* 2-bit / 15-bit / 15-bit fields of the reference */
	uint32_t llm, rlm, len;
	span_t longest;                  /* alnl */
	span_t *list;                    /* alns */
} read_info;

typedef struct {
	uint32_t n_buckets, size, n_occupied, upper;
	uint8_t *used;
	uint32_t *keys;
	read_info *vals;
} table_t;

typedef struct { uint32_t rev, qname, qs, qe, qlen, tname, ts, te, tlen, identity; } nd_s2_ovl;

/* ---------------------------------------------------------------- the hash table's layout */

static uint32_t table_find(const table_t *h, uint32_t key)
{
	uint32_t mask, i, last, step = 0;
	if (!h->n_buckets) return 0;
	mask = h->n_buckets - 1, i = key & mask, last = i;
	while (h->used[i] && h->keys[i] != key) {
		i = (i + (++step)) & mask;
		if (i == last) return h->n_buckets;
	}
	return h->used[i] ? i : h->n_buckets;
}

static void table_grow(table_t *h, uint32_t want)
{
	uint32_t nb = want, j;
	uint8_t *nused;
	--nb, nb |= nb >> 1, nb |= nb >> 2, nb |= nb >> 4, nb |= nb >> 8, nb |= nb >> 16, ++nb;
	if (nb < 4) nb = 4;
	if (h->size >= (uint32_t)(nb * 0.77 + 0.5)) return;
	nused = (uint8_t*)calloc(nb, 1);
	h->keys = (uint32_t*)realloc(h->keys, sizeof(uint32_t) * nb);
	h->vals = (read_info*)realloc(h->vals, sizeof(read_info) * nb);
	for (j = 0; j != h->n_buckets; ++j) {
		if (h->used[j] == 1) { /* still sitting at its old position */
			uint32_t key = h->keys[j];
			read_info val = h->vals[j];
			h->used[j] = 2; /* moved out */
			for (;;) {
				uint32_t i = key & (nb - 1), step = 0;
				while (nused[i]) i = (i + (++step)) & (nb - 1);
				nused[i] = 1;
				if (i < h->n_buckets && h->used[i] == 1) { /* an unmoved element lives here: it is displaced and placed next */
					uint32_t tk = h->keys[i];
					read_info tv = h->vals[i];
					h->keys[i] = key, h->vals[i] = val;
					key = tk, val = tv;
					h->used[i] = 2;
				} else {
					h->keys[i] = key, h->vals[i] = val;
					break;
				}
			}
		}
	}
	free(h->used);
	h->used = nused;
	h->n_buckets = nb;
	h->n_occupied = h->size;
	h->upper = (uint32_t)(nb * 0.77 + 0.5);
}

static uint32_t table_insert(table_t *h, uint32_t key)
{
	uint32_t mask, i, step = 0;
	if (h->n_occupied >= h->upper) table_grow(h, h->n_buckets > (h->size << 1) ? h->n_buckets - 1 : h->n_buckets + 1);
	mask = h->n_buckets - 1, i = key & mask;
	while (h->used[i] && h->keys[i] != key) i = (i + (++step)) & mask;
	if (!h->used[i]) {
		h->used[i] = 1, h->keys[i] = key;
		++h->size, ++h->n_occupied;
	}
	return i;
}

/* ---------------------------------------------------------------- interval lists */

static void list_merge(read_info *s)
{
	uint16_t i, j;
	for (i = 1; i < s->cap; i++) { /* by start; equal starts keep their order */
		span_t t = s->list[i];
		for (j = i; j > 0 && s->list[j - 1].s > t.s; j--) s->list[j] = s->list[j - 1];
		s->list[j] = t;
	}
	i = 0;
	while (i < s->cap - 1) {
		if (!s->list[i].e) { i++; continue; }
		for (j = i + 1; j < s->cap; j++) {
			if (s->list[j].e <= s->list[i].e) s->list[j].e = 0;
			else if (s->list[j].s <= s->list[i].e && s->list[j].e >= s->list[i].e) s->list[i].e = s->list[j].e, s->list[j].e = 0;
			else break;
		}
		i = j;
	}
}

static uint16_t list_free_slot(read_info *s)
{
	uint16_t i;
	if (s->cur != s->cap - 1)
		for (i = 0; i < s->cap; i++) if (!s->list[i].e) return i;
	list_merge(s);
	for (i = 0; i < s->cap; i++) if (!s->list[i].e) return i;
	s->cap = (uint16_t)(s->cap + LIST_STEP);
	s->list = (span_t*)realloc(s->list, sizeof(span_t) * s->cap);
	memset(s->list + s->cap - LIST_STEP, 0, sizeof(span_t) * LIST_STEP);
	return i;
}

static void cover(read_info *s, uint32_t a, uint32_t b)
{
	if (s->con >= CON_MAX) return;
	s->cur = list_free_slot(s);
	s->list[s->cur].s = a + EDGE_BACK, s->list[s->cur].e = b - EDGE_BACK;
}

static void remember_longest(read_info *s, uint32_t a, uint32_t b)
{
	if (s->con < CON_MAX && b - a > s->longest.e - s->longest.s) s->longest.s = a, s->longest.e = b;
}

static read_info *touch(table_t *h, uint32_t name, uint32_t len, uint32_t lo_gap, uint32_t hi_gap, uint32_t han2, int target_side)
{
	uint32_t k = table_find(h, name);
	read_info *r;
	if (k != h->n_buckets) {
		r = &h->vals[k];
		if (r->con < CON_MAX) {
			/* the target side tests rc before it bumps lc (lib/ovl.c:480) */
			if (lo_gap <= han2 && (target_side ? r->rc : r->lc) < UINT16_MAX) r->lc++;
			if (hi_gap <= han2 && r->rc < UINT16_MAX) r->rc++;
		}
		return r;
	}
	k = table_insert(h, name);
	r = &h->vals[k];
	memset(r, 0, sizeof(*r));
	r->len = len;
	r->cap = LIST_STEP;
	r->list = (span_t*)calloc(LIST_STEP, sizeof(span_t));
	if (lo_gap <= han2) r->lc++;
	if (hi_gap <= han2) r->rc++;
	return r;
}

static int bump_contained(read_info *r)
{
	if (++r->con >= CON_MAX) { free(r->list); r->list = NULL, r->cap = 0; }
	return 0;
}

/* ---------------------------------------------------------------- the filter */

void *nd_s2_new(void) { return calloc(1, sizeof(table_t)); }

void nd_s2_free(void *state)
{
	table_t *h = (table_t*)state;
	uint32_t i;
	for (i = 0; i < h->n_buckets; ++i) if (h->used[i]) free(h->vals[i].list);
	free(h->used); free(h->keys); free(h->vals); free(h);
}

int nd_s2_filter(void *state, const nd_s2_ovl *o, int32_t maxhan1, int32_t maxhan2)
{
	table_t *h = (table_t*)state;
	const uint32_t h1 = (uint32_t)maxhan1, h2 = (uint32_t)maxhan2; /* the reference compares uint32 with int32: unsigned */
	read_info *q, *t;
	uint32_t span;
	touch(h, o->qname, o->qlen, o->qs, o->qlen - o->qe, h2, 0);
	t = touch(h, o->tname, o->tlen, o->ts, o->tlen - o->te, h2, 1);
	q = &h->vals[table_find(h, o->qname)]; /* inserting the target may have moved the table */
	cover(q, o->qs, o->qe);
	cover(t, o->ts, o->te);
	if (q->con < CON_MAX && o->qs <= h2 && o->qe + h2 >= o->qlen) return bump_contained(q);
	if (t->con < CON_MAX && o->ts <= h2 && o->te + h2 >= o->tlen) return bump_contained(t);
	if (q->con >= CON_MAX || t->con >= CON_MAX) return 0;
	span = o->qe - o->qs > o->te - o->ts ? o->qe - o->qs : o->te - o->ts;
	{
		/* which end of each read the overlap reaches: 0 = 5', 1 = 3' */
		int q_end = -1, t_end = -1;
		const uint32_t q_lo = o->qs, q_hi = o->qlen - o->qe, t_lo = o->ts, t_hi = o->tlen - o->te;
		uint32_t gq = 0, gt = 0;
		if (o->rev) {
			if (q_lo <= h1 && t_lo <= h1) q_end = 0, t_end = 0, gq = q_lo, gt = t_lo;
			else if (q_hi <= h1 && t_hi <= h1) q_end = 1, t_end = 1, gq = q_hi, gt = t_hi;
		} else {
			if (q_hi <= h1 && t_lo <= h1) q_end = 1, t_end = 0, gq = q_hi, gt = t_lo;
			else if (q_lo <= h1 && t_hi <= h1) q_end = 0, t_end = 1, gq = q_lo, gt = t_hi;
		}
		if (q_end >= 0) {
			if (gq <= h2 && gt <= h2) {
				uint32_t *ql = q_end ? &q->rlm : &q->llm, *tl = t_end ? &t->rlm : &t->llm;
				uint32_t *qi = q_end ? &q->rim : &q->lim, *ti = t_end ? &t->rim : &t->lim;
				if (span > *ql) *ql = span;
				if (span > *tl) *tl = span;
				if (o->identity > *qi) *qi = o->identity & 0x7fff; /* 15-bit fields */
				if (o->identity > *ti) *ti = o->identity & 0x7fff;
			}
			return 1;
		}
	}
	if (o->qs <= h1 && o->qe + h1 >= o->qlen) return 1; /* contained once the read ends are clipped: kept */
	if (o->ts <= h1 && o->te + h1 >= o->tlen) return 1;
	remember_longest(q, o->qs, o->qe);
	remember_longest(t, o->ts, o->te);
	return 0;
}

/* the `.bl` text; the per-read lists are released as the reference does (the state is spent afterwards) */
int64_t nd_s2_out_bl(void *state, char *out, int64_t cap)
{
	table_t *h = (table_t*)state;
	int64_t n = 0;
	uint32_t k;
	uint16_t i;
	for (k = 0; k < h->n_buckets; ++k) {
		read_info *r;
		if (!h->used[k]) continue;
		r = &h->vals[k];
		if (n + 64 + 24 * (int64_t)r->cap > cap) return -1;
		if (r->con < CON_MAX) {
			n += sprintf(out + n, "%u\t%u\t%hu\t%hu\t%u\t%u\t%u\t%u\t%u\t%u\t%u", h->keys[k], r->con, r->lc, r->rc, r->lim, r->rim, r->llm, r->rlm,
			             r->len, r->longest.s, r->longest.e);
			list_merge(r);
			for (i = 0; i < r->cap; i++)
				if (r->list[i].e) n += sprintf(out + n, "\t%u\t%u", r->list[i].s - EDGE_BACK, r->list[i].e + EDGE_BACK);
			n += sprintf(out + n, "\n");
			free(r->list);
			r->list = NULL;
		} else n += sprintf(out + n, "%u\t%u\n", h->keys[k], r->con);
	}
	return n;
}
