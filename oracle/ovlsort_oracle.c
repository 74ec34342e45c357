/* ovlsort_oracle.c -- CPU restatement of util/ovl_sort.c for raw-read data (no -H), in-memory case
 * (TEST INFRASTRUCTURE ONLY: loaded by tests/, never by the product).
 *
 * Follows, for the case where every input fits the sort buffers (no temporary files):
 *   nd_os_expand   util/ovl_sort.c:980-1037  (sort_ovl_file: pre-filters, both directions, seed lookup,
 *                                             the "5 misses then stop" counters per input file)
 *   nd_os_order    util/ovl_sort.c:246-261 (cmp_ovl) + :876-925 (merge_ovl_from_sort): seed asc, match desc,
 *                  span asc; equal keys keep input order (glibc qsort is a merge sort; buffers merge in order).
 *                  One case in which the reference does not define the order at all: equal keys in DIFFERENT input
 *                  files (duplicated read segments).  The merge takes the sort buffer with the lower index (:883-893),
 *                  and a reader thread takes its buffer when it reaches its file's first kept record (:933-936): a
 *                  race between the -t threads.  On an idle machine buffer i goes to file i -- file order, as here
 *                  and on the device -- and on a loaded one it does not (tools/fuzz_stage.py met it once in 50 sets;
 *                  14 reruns of the compiled ovl_sort on that input gave file order every time).
 *   admit()        util/ovl_sort.c:675-741   (encode_ovl_filter: 64-base coverage bins)
 *   finish_seed()  util/ovl_sort.c:433-571   (ovl_filter: chimera / low-coverage trimming, .bl verdict)
 *                  with check_chimer :316-336 and check_chimer2 :339-383
 *   nd_os_run      the whole run -> records of sorted.ovl (self record first per seed, inclusive ends) + .bl lines
 *
 * Pinned against the compiled reference (oracle/_ref/ovl_sort) by tests/test_ovlsort_oracle.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

typedef struct { uint32_t rev, qname, qs, qe, tname, ts, te, match; } nd_ovl;

#define BIN_SHIFT 6
#define COV_CAP 150        /* MAX_OVL_COV */
#define EDGE_TOL 50        /* BIN_TOLERANCE_EDGE */
#define EDGE_COUNT 5       /* BIN_TOLERANCE_COUNT */
#define CONTAINED_MIN 2    /* MIN_CONTAINTED_COUNT */

#define MAXV(x, y) ((x) > (y) ? (x) : (y))
#define MINV(x, y) ((x) > (y) ? (y) : (x))

/* ---------------------------------------------------------------- expand */

/* raw: n x 8 in decode_ovl order [qname, rev, qs, qe, tname, ts, te, match] of ONE input file.
 * seed_len[id] = length of seed `id` in this seed file, 0 when id is not one of its seeds.
 * out needs room for 2n records.  Returns the number of candidates written. */
int64_t nd_os_expand(const uint32_t *raw, int64_t n, const uint32_t *seed_len, uint32_t n_ids, nd_ovl *out)
{
	int left_q = 5, left_t = 5; /* "max error record" */
	int64_t m = 0, i;
	for (i = 0; i < n; ++i) {
		const uint32_t *r = raw + 8 * i;
		if (r[0] == r[4]) continue;
		if (r[3] - r[2] < 500 || r[6] - r[5] < 500) continue;
		if (left_q && r[0] < n_ids && seed_len[r[0]] && seed_len[r[0]] >= r[3]) {
			nd_ovl *o = &out[m++];
			o->qname = r[0], o->rev = r[1], o->qs = r[2], o->qe = r[3] - 1, o->tname = r[4], o->ts = r[5], o->te = r[6] - 1, o->match = r[7];
		} else if (left_q) --left_q;
		if (left_t && r[4] < n_ids && seed_len[r[4]] && seed_len[r[4]] >= r[6]) {
			nd_ovl *o = &out[m++];
			o->qname = r[4], o->rev = r[1], o->qs = r[5], o->qe = r[6] - 1, o->tname = r[0], o->ts = r[2], o->te = r[3] - 1, o->match = r[7];
		} else if (left_t) --left_t;
	}
	return m;
}

/* ---------------------------------------------------------------- order */

static const nd_ovl *g_base;
static int cmp_idx(const void *pa, const void *pb)
{
	const uint32_t ia = *(const uint32_t*)pa, ib = *(const uint32_t*)pb;
	const nd_ovl *a = g_base + ia, *b = g_base + ib;
	if (a->qname != b->qname) return a->qname < b->qname ? -1 : 1;
	if (a->match != b->match) return a->match > b->match ? -1 : 1;
	{
		const uint32_t sa = a->qe - a->qs, sb = b->qe - b->qs;
		if (sa != sb) return sa < sb ? -1 : 1;
	}
	return ia < ib ? -1 : ia > ib; /* stable */
}

void nd_os_order(const nd_ovl *c, int64_t n, uint32_t *perm)
{
	int64_t i;
	for (i = 0; i < n; ++i) perm[i] = (uint32_t)i;
	g_base = c;
	qsort(perm, n, 4, cmp_idx);
}

/* ---------------------------------------------------------------- per-seed state */

typedef struct {
	uint8_t repeat_run;              /* pcount */
	uint32_t n_kept;                 /* ovl_i (uint16 in the reference; capped below 64535 by the admission test) */
	uint16_t *bins;
	nd_ovl *kept;
	uint32_t cap_kept;
	uint32_t last_qs, last_qe, qlen, qcap, qcov;
	uint32_t bins_touched, bins_sum; /* binlen, bincount */
	uint32_t contained, chimera;
	uint32_t n_bins, cap_bins;
} seed_state;

static void seed_begin(seed_state *s, uint32_t qlen, int hq)
{
	s->qlen = qlen;
	s->n_bins = (qlen >> BIN_SHIFT) + (hq ? 2 : 1); /* ovl_sort.c:622 vs :663 */
	if (s->n_bins > s->cap_bins) {
		s->cap_bins = s->n_bins;
		s->bins = (uint16_t*)realloc(s->bins, 2 * (size_t)s->cap_bins);
	}
	memset(s->bins, 0, 2 * (size_t)s->n_bins);
	s->repeat_run = 1;
	s->qcov = s->bins_sum = s->bins_touched = s->contained = s->chimera = s->n_kept = 0;
	s->qcap = qlen * COV_CAP * (hq ? 6 : 1);        /* ovl_sort.c:630 */
}

static void keep(seed_state *s, const nd_ovl *o, int flank)
{
	s->last_qs = o->qs, s->last_qe = o->qe;
	s->qcov += o->qe - o->qs + 1;
	if (o->qname != o->tname && o->qs <= (uint32_t)flank && o->qe + (uint32_t)flank >= s->qlen) s->contained++;
	if (s->n_kept >= s->cap_kept) {
		s->cap_kept = s->cap_kept ? s->cap_kept * 2 : 1024;
		s->kept = (nd_ovl*)realloc(s->kept, sizeof(nd_ovl) * s->cap_kept);
	}
	s->kept[s->n_kept++] = *o;
}

/* coverage-bin admission of one candidate of the current seed (not the self record) */
static void admit(seed_state *s, const nd_ovl *o, int max_bin_cov, int flank)
{
	int label = 1, lowest = 200;
	if (s->qcov > s->qcap || s->n_kept > 65535 - 1000) return;
	{
		int i, sum = 0, fresh = 0;
		const int j = (int)((o->qs + 10) >> BIN_SHIFT), k = (int)((o->qe - 10) >> BIN_SHIFT);
		if ((j > 15 || k < (int)s->n_bins - 16) && abs((int)(o->qs - s->last_qs)) < EDGE_TOL && abs((int)(o->qe - s->last_qe)) < EDGE_TOL)
			label = s->repeat_run++ < EDGE_COUNT ? 2 : 0;
		if (!label) return;
		for (i = j + 1; i <= k; ++i) {
			if (!s->bins[i]) ++fresh;
			if (++s->bins[i] < lowest) lowest = s->bins[i];
			if (s->bins[i] > 65535 - 1000) s->bins[i]--;
			sum += s->bins[i];
		}
		if ((lowest > max_bin_cov ||
		     (float)sum / (k - j) > 1.3 * MINV(MAXV((float)s->bins_sum / s->bins_touched, 10), max_bin_cov)) &&
		    (o->qe - o->qs <= s->qlen * 0.8)) {
			for (i = j + 1; i <= k; ++i) s->bins[i]--;
			return;
		}
		if (label != 2) s->repeat_run = 1;
		s->bins_touched += (uint32_t)fresh;
		s->bins_sum += (uint32_t)(k - j);
	}
	keep(s, o, flank);
}

/* -H (high-quality reads), ovl_sort.c:616-655 encode_ovl_filter_hq: every candidate is collected (up to the caps); the two
 * halves of bins[] count alignment starts and ends per 128 bases */
static void admit_hq(seed_state *s, const nd_ovl *o, int flank)
{
	if (s->qcov > s->qcap || s->n_kept > 65535 - 1000) return;
	{
		const int off = 1 + (int)(s->qlen >> (BIN_SHIFT + 1));
		s->bins[(o->qs + 10) >> (BIN_SHIFT + 1)]++;
		s->bins[((o->qe - 10) >> (BIN_SHIFT + 1)) + off]++;
	}
	keep(s, o, flank);
}

/* ovl_sort.c:389-431 del_repeat_alns: overlaps that start AND end where >= 5 overlaps start / end are repeat-induced;
 * the 64-base coverage is then rebuilt over the rest, dropping what would push a whole span above 2 x max_bin_cov */
static void drop_repeat_alignments(seed_state *s, int max_bin_cov, int flank)
{
	int i, t;
	const int off = 1 + (int)(s->qlen >> (BIN_SHIFT + 1)), hot = 5;
	const uint32_t fl = flank > 100 ? (uint32_t)flank * 3 : 300;
	for (i = 1; i < (int)s->n_kept; ++i) {
		nd_ovl *o = &s->kept[i];
		if (o->qs <= fl && o->qe + fl >= s->qlen) continue;
		if (s->bins[(o->qs + 10) >> (BIN_SHIFT + 1)] >= hot && s->bins[((o->qe - 10) >> (BIN_SHIFT + 1)) + off] >= hot) o->qe = 0;
	}
	memset(s->bins, 0, 2 * (size_t)s->n_bins);
	for (i = 1; i < (int)s->n_kept; ++i) {
		nd_ovl *o = &s->kept[i];
		int lowest = UINT16_MAX;
		const int j = (int)((o->qs + 10) >> BIN_SHIFT), k = (int)((o->qe - 10) >> BIN_SHIFT);
		if (!o->qe) continue;
		for (t = j + 1; t <= k; ++t) {
			if (++s->bins[t] > UINT16_MAX - 1000) s->bins[t]--;
			if (s->bins[t] < lowest) lowest = s->bins[t];
		}
		if (lowest > 2 * max_bin_cov) {
			for (t = j + 1; t <= k; ++t) s->bins[t]--;
			o->qe = 0;
		}
	}
}

/* ovl_sort.c:287-314 check_chimer_hq: a bin covered at most once, inside the covered part, that no kept overlap spans
 * with 15 bins to spare on both sides */
static int chimera_hq(const seed_state *s)
{
	int i, j, l = 0, r = (int)s->n_bins;
	const int pad = 15;
	while (l < (int)s->n_bins && s->bins[l] < 2) ++l;
	while (r > 0 && s->bins[r - 1] < 2) --r;
	for (i = l + 1; i < r - 1; ++i) {
		if (s->bins[i] <= 1) {
			const int lo = i > l + pad ? (i - pad) << BIN_SHIFT : l << BIN_SHIFT;
			const int hi = i + pad < r ? (i + pad) << BIN_SHIFT : r << BIN_SHIFT;
			for (j = 1; j < (int)s->n_kept; ++j)
				if (s->kept[j].qs < (uint32_t)lo && s->kept[j].qe > (uint32_t)hi) break;
			if (j >= (int)s->n_kept) return i;
		}
	}
	return 0;
}

static int chimera_by_coverage(const seed_state *s)
{
	int i, l, r, label = 0, llabel = 0, rlabel = 0;
	const int n = (int)s->n_bins;
	for (i = 1; i < n - 1; ++i) {
		if (s->bins[i] > 20 && ++llabel) {
			if (label && ++rlabel >= 5) break;
		} else {
			l = MAXV(i - 5, 0); r = MINV(i + 5, n - 1);
			if (llabel > 5 && (s->bins[l] > 20 || s->bins[r] > 20) && s->bins[i] <= MAXV(3, MINV(s->bins[l], s->bins[r]) / 5))
				label = i;
		}
	}
	if (rlabel < 5) label = 0;
	return label;
}

/* hot break ends: a pile-up of alignment ends well inside the read */
static int chimera_by_ends(seed_state *s)
{
	int i, lo, hi, t, c = 0;
	const int sh = BIN_SHIFT + 1;
	memset(s->bins, 0, 2 * ((size_t)s->n_bins / 2 + 1));
	lo = (int)s->n_bins, hi = 0;
	for (i = 1; i < (int)s->n_kept; ++i) {
		const nd_ovl *o = &s->kept[i];
		if (!o->qe) continue;
		++c;
		t = (int)((o->qs + 10) >> sh);
		if (t < lo) lo = t;
		s->bins[t]++;
		t = (int)((o->qe - 10) >> sh);
		if (t > hi) hi = t;
		s->bins[t]++;
	}
	t = 0;
	if (c > 20) {
		int ms, me, m;
		while (lo < hi && s->bins[lo] < 4) ++lo;
		while (hi > lo && s->bins[hi] < 4) --hi;
		for (m = 0, ms = s->bins[lo], me = s->bins[hi], i = lo; i < hi + 1; ++i) {
			if (i < lo + 5 && s->bins[i] > ms) ms = s->bins[i];
			if (i > hi - 5 && s->bins[i] > me) me = s->bins[i];
			if (s->bins[i] > s->bins[m]) m = i;
		}
		if (m > lo + 5 && m < hi - 5 && s->bins[m] > 1.f * MAXV(ms, me) && ((c > 75 && m > c / 5) || (c < 75 && m > c / 2)))
			t = m << sh;
	}
	return t;
}

/* end of a seed: trimming decisions; survivors (qe != 0) go to `out`, a verdict may go to `bl` */
static int64_t finish_seed(seed_state *s, int max_bin_cov, int flank, int min_seed_len, nd_ovl *out, uint32_t *bl_id, uint8_t *bl_kind,
                           int64_t *n_bl, int hq)
{
	int i, lo = 0, hi = 0;
	int64_t n = 0;
	const int nb = (int)s->n_bins;
	if (hq) drop_repeat_alignments(s, max_bin_cov, flank);
	s->chimera = (uint32_t)(hq ? chimera_hq(s) : chimera_by_coverage(s));
	if (s->chimera || !(s->contained || hq)) {
		int j = 0, k, m;
		uint16_t *b = s->bins; /* the list of (first, last) low-coverage bin runs overwrites the front of bins[] */
		if (s->qcov > s->qlen * 10) {
			for (i = 1; i < nb - 1; ++i) {
				if (b[i] < MINV(4, max_bin_cov / 10)) {
					if (lo == 0) lo = i;
					hi = i;
				} else if (lo) {
					if (s->chimera && s->chimera < (uint32_t)lo && ((!j) || s->chimera > b[j - 1])) b[j++] = (uint16_t)s->chimera, b[j++] = (uint16_t)s->chimera;
					b[j++] = (uint16_t)lo, b[j++] = (uint16_t)hi;
					lo = hi = 0;
				}
			}
			if (lo) {
				if (s->chimera && s->chimera < (uint32_t)lo && ((!j) || s->chimera > b[j - 1])) b[j++] = (uint16_t)s->chimera, b[j++] = (uint16_t)s->chimera;
				b[j++] = (uint16_t)lo, b[j++] = (uint16_t)hi;
			}
			if (s->chimera && (j == 0 || s->chimera > b[j - 1])) b[j++] = (uint16_t)s->chimera, b[j++] = (uint16_t)s->chimera;
		} else if (s->chimera) b[j++] = (uint16_t)s->chimera, b[j++] = (uint16_t)s->chimera;
		if (j) {
			m = j;
			if (b[0] < 5) m -= 2;
			if (b[j - 1] > nb - 5) m -= 2;
			if (m > 0) {
				k = 0;
				m = b[0];
				for (i = 2; i < j; i += 2)
					if (b[i] - b[i - 1] > m) m = b[i] - b[i - 1], k = i;
				if (nb - b[i - 1] > m) {
					m = nb - b[i - 1];
					lo = b[i - 1], hi = nb;
				} else if (b[k + 1] > nb - 5) {
					lo = b[k - 1], hi = nb;
				} else if (k == 0 || b[k - 2] < 5) {
					lo = 0, hi = b[k];
				} else {
					lo = b[k - 1], hi = b[k];
				}
				lo = lo > 5 ? (lo - 5) << BIN_SHIFT : 0;
				hi = (hi + 5) << BIN_SHIFT;
				if (m > (min_seed_len >> BIN_SHIFT) * 2 / 3) {
					s->chimera = 0;
					for (i = 1; i < (int)s->n_kept; ++i)
						if (s->kept[i].qs < (uint32_t)lo || s->kept[i].qe > (uint32_t)hi) s->kept[i].qe = 0;
				} else s->chimera = 1;
			} else lo = hi = 0;
		}
	}
	if (!hq && s->qcov > s->qlen * 20 && !s->chimera && s->contained < CONTAINED_MIN) {
		s->chimera = (uint32_t)chimera_by_ends(s);
		if (!hi) hi = (int)s->qlen;
		if (s->chimera <= (uint32_t)(lo + (15 << BIN_SHIFT)) || s->chimera + (15 << BIN_SHIFT) >= (uint32_t)hi) s->chimera = 0;
	}
	s->contained = 0;
	for (i = 0; i < (int)s->n_kept; ++i) {
		const nd_ovl *o = &s->kept[i];
		if (!o->qe) continue;
		out[n++] = *o;
		if (o->qname != o->tname && o->qs <= (uint32_t)flank && o->qe + (uint32_t)flank >= s->qlen) {
			if (!hq) s->contained++;
			else if (o->match >= (o->qe - o->qs + 1) * 0.9) s->contained++; /* a containing overlap of HQ reads must be >= 90 % matches */
		}
	}
	if (s->contained >= CONTAINED_MIN) bl_id[*n_bl] = s->kept[0].qname, bl_kind[(*n_bl)++] = 'c';
	else if (s->chimera) bl_id[*n_bl] = s->kept[0].qname, bl_kind[(*n_bl)++] = 'k';
	return n;
}

/* ---------------------------------------------------------------- whole run */

/* cand[perm[0..n)] = candidates in merge order.  out: room for n + number of seeds records; bl_*: room for the
 * number of seeds.  Returns records written. */
int64_t nd_os_filter2(const nd_ovl *cand, const uint32_t *perm, int64_t n, const uint32_t *seed_len, int max_bin_cov, int flank,
                      int min_seed_len, nd_ovl *out, uint32_t *bl_id, uint8_t *bl_kind, int64_t *n_bl, int hq);

int64_t nd_os_filter(const nd_ovl *cand, const uint32_t *perm, int64_t n, const uint32_t *seed_len, int max_bin_cov, int flank,
                     int min_seed_len, nd_ovl *out, uint32_t *bl_id, uint8_t *bl_kind, int64_t *n_bl)
{
	return nd_os_filter2(cand, perm, n, seed_len, max_bin_cov, flank, min_seed_len, out, bl_id, bl_kind, n_bl, 0);
}

/* hq != 0: the -H variant (ovl_sort.c:27,1045) */
int64_t nd_os_filter2(const nd_ovl *cand, const uint32_t *perm, int64_t n, const uint32_t *seed_len, int max_bin_cov, int flank,
                      int min_seed_len, nd_ovl *out, uint32_t *bl_id, uint8_t *bl_kind, int64_t *n_bl, int hq)
{
	seed_state s;
	int64_t i, n_out = 0;
	uint32_t cur = UINT32_MAX;
	int open = 0;
	memset(&s, 0, sizeof(s));
	*n_bl = 0;
	for (i = 0; i < n; ++i) {
		const nd_ovl *o = &cand[perm[i]];
		if (!open || o->qname != cur) {
			nd_ovl self;
			if (open) n_out += finish_seed(&s, max_bin_cov, flank, min_seed_len, out + n_out, bl_id, bl_kind, n_bl, hq);
			cur = o->qname, open = 1;
			seed_begin(&s, seed_len[cur], hq);
			memset(&self, 0, sizeof(self));
			self.qname = self.tname = cur;
			self.qe = self.te = seed_len[cur] - 1;
			keep(&s, &self, flank);
		}
		if (hq) admit_hq(&s, o, flank);
		else admit(&s, o, max_bin_cov, flank);
	}
	if (open) n_out += finish_seed(&s, max_bin_cov, flank, min_seed_len, out + n_out, bl_id, bl_kind, n_bl, hq);
	free(s.bins); free(s.kept);
	return n_out;
}

/* lib/ovl.c:109-150 (encode_ovl) over the output records, delta state starting at {0,0}; out: 40 bytes per record */
int64_t nd_os_encode(const nd_ovl *r, int64_t n, uint8_t *out)
{
	uint32_t pq = 0, pt = 0;
	int64_t nb = 0, i;
	int f, sh;
	for (i = 0; i < n; ++i, ++r) {
		uint32_t fld[8], flags = r->rev;
		const uint32_t qspan = r->qe - r->qs, tspan = r->te - r->ts;
		fld[3] = qspan;
		if (r->qname >= pq) fld[0] = r->qname - pq; else flags |= 2, fld[0] = pq - r->qname;
		pq = r->qname;
		if (r->tname >= pt) fld[4] = r->tname - pt; else flags |= 4, fld[4] = pt - r->tname;
		pt = r->tname;
		if (qspan >= tspan) fld[6] = qspan - tspan; else flags |= 8, fld[6] = tspan - qspan;
		fld[1] = flags & 0xff, fld[2] = r->qs, fld[5] = r->ts, fld[7] = r->match;
		for (f = 0; f < 8; ++f) {
			const uint32_t v = fld[f];
			if (v <= 127) { out[nb++] = (uint8_t)v; continue; }
			{
				int m = 0;
				for (sh = 28; sh >= 0; sh -= 7) {
					const uint32_t g = v >> sh & 127;
					if (g > 0 || m > 0) out[nb + m++] = (uint8_t)(g | 128);
				}
				out[nb + m - 1] &= 127;
				nb += m;
			}
		}
	}
	return nb;
}
