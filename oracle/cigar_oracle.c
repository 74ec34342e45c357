/* oracle/cigar_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see nd_oracle.h).  Compiled as part of mm_oracle.c (it uses that
 * file's static helpers: the mapping of one read up to the chains, the record encoder).
 *
 * CPU restatement of `minimap2-nd --step 1 -c` between chaining and the writer -- the base-level alignment through every chain:
 *     mm_align_skeleton   minimap2/align.c:857-913      mm_align1       :565-788      mm_align1_inv    :790-845
 *     mm_test_zdrop       :47-89                         mm_fix_cigar    :91-166       mm_update_extra  :240-286
 *     mm_append_cigar     :288-311                       mm_adjust_minier / mm_get_hplen_back  :341-365
 *     collect_long_gaps, mm_filter_bad_seeds, mm_filter_bad_seeds_alt, mm_fix_bad_ends          :367-493
 *     mm_split_reg, mm_filter_regs, mm_hit_sort          minimap2/hit.c:90-107,257-276,169-201
 *     the step-1 writer's filter                          minimap2/map.c:1297-1304
 * for the presets nextDenovo runs on raw reads (ava-ont / ava-pb: every chain kept, no long join, no splicing, no short-read mode),
 * ONE read and ONE chain at a time, in the reference's order, with the scalar kernels of oracle/ksw2_oracle.c (nd_oracle_ksw_extd2,
 * nd_oracle_ksw_ll_i16).  The product (nextdenovo_amd/csrc/ovl_cigar.cpp) computes the same in batches on the device.
 * Pinned by the compiled reference binary's golden `.ovl` files (tests/golden/cigar/, tests/test_overlap_oracle.py) and live. */

typedef struct {
	int32_t a, b, q, e, q2, e2, sc_ambi, zdrop, zdrop_inv, end_bonus, min_dp_max, min_ksw_len;
	int64_t max_sw_mat;
} nd_aln_opt;

typedef struct {
	int32_t id, cnt, rid, score, qs, qe, rs, re, parent, as, mlen, blen;
	uint32_t hash;
	int rev, inv, split, split_inv;
	int has_p, dp_max, n_cigar, m_cigar;
	uint32_t *cigar;
} c_reg;

typedef struct {
	int32_t max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, n_cigar, reach_end;
} c_ksw_result; /* = nd_ksw_result of ksw2_oracle.c */
int nd_oracle_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat, int8_t q, int8_t e,
                        int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, c_ksw_result *ez, uint32_t *cigar, int cigar_cap);
int nd_oracle_ksw_ll_i16(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int gapo, int gape, int *qe, int *te);

#define C_SEED_LONG_JOIN (1ULL << 40)
#define C_SEED_IGNORE (1ULL << 41)
#define C_EZ_RIGHT 0x02
#define C_EZ_APPROX_MAX 0x08
#define C_EZ_EXTZ_ONLY 0x40
#define C_EZ_REV_CIGAR 0x80

typedef struct {
	const nd_mm_opt *opt;
	const nd_aln_opt *ao;
	const nd_mm_index *ix;
	const uint8_t *tcodes;
	const uint64_t *toff;
	int8_t mat[25];
	int qlen;
	uint8_t *qseq0[2];
	nd_mm128 *a;
	int n_a;
	uint32_t *cg; /* scratch CIGAR of one kernel call */
	int cg_cap;
} c_ctx;

static void c_reg_set_coor(c_reg *r, int qlen, const nd_mm128 *a) /* mm_reg_set_coor + mm_cal_fuzzy_len, hit.c:8-38 */
{
	int k = r->as, span = (int)(a[k].y >> 32 & 0xff), i;
	r->rev = (int)(a[k].x >> 63), r->rid = (int32_t)(a[k].x << 1 >> 33);
	r->rs = (int32_t)a[k].x + 1 > span ? (int32_t)a[k].x + 1 - span : 0;
	r->re = (int32_t)a[k + r->cnt - 1].x + 1;
	if (!r->rev) r->qs = (int32_t)a[k].y + 1 - span, r->qe = (int32_t)a[k + r->cnt - 1].y + 1;
	else r->qs = qlen - ((int32_t)a[k + r->cnt - 1].y + 1), r->qe = qlen - ((int32_t)a[k].y + 1 - span);
	r->mlen = r->blen = span;
	for (i = r->as + 1; i < r->as + r->cnt; ++i) {
		int sp = (int)(a[i].y >> 32 & 0xff), tl = (int32_t)a[i].x - (int32_t)a[i - 1].x, ql = (int32_t)a[i].y - (int32_t)a[i - 1].y;
		r->blen += tl > ql ? tl : ql;
		r->mlen += tl > sp && ql > sp ? sp : tl < ql ? tl : ql;
	}
}

static void c_append(c_reg *r, int n, const uint32_t *c) /* mm_append_cigar */
{
	if (n == 0) return;
	r->has_p = 1;
	if (r->n_cigar + n > r->m_cigar) {
		r->m_cigar = (r->n_cigar + n) * 2 + 16;
		r->cigar = (uint32_t*)realloc(r->cigar, 4 * (size_t)r->m_cigar);
	}
	if (r->n_cigar > 0 && (r->cigar[r->n_cigar - 1] & 0xf) == (c[0] & 0xf)) {
		r->cigar[r->n_cigar - 1] += c[0] >> 4 << 4;
		if (n > 1) memcpy(r->cigar + r->n_cigar, c + 1, 4 * (size_t)(n - 1));
		r->n_cigar += n - 1;
	} else {
		memcpy(r->cigar + r->n_cigar, c, 4 * (size_t)n);
		r->n_cigar += n;
	}
}

static void c_getseq(const c_ctx *C, int rid, int st, int en, uint8_t *out) /* mm_idx_getseq */
{
	memcpy(out, C->tcodes + C->toff[rid] + st, (size_t)(en > st ? en - st : 0));
}

static void c_adjust_minier(const c_ctx *C, const nd_mm128 *a, int32_t *r, int32_t *q) /* align.c:341-365 */
{
	if (C->opt->hpc) {
		const uint8_t *qseq = C->qseq0[a->x >> 63], *T = C->tcodes + C->toff[a->x << 1 >> 33];
		int i, c;
		int64_t j;
		*q = (int32_t)a->y;
		for (i = *q - 1, c = qseq[*q]; i > 0; --i)
			if (qseq[i] != c) break;
		*q = i + 1;
		c = T[(int32_t)a->x];
		for (j = (int64_t)(int32_t)a->x - 1; j >= 0; --j)
			if (T[j] != c) break;
		*r = (int32_t)a->x + 1 - (int)((int64_t)(int32_t)a->x - j);
	} else {
		*r = (int32_t)a->x - (C->opt->k >> 1);
		*q = (int32_t)a->y - (C->opt->k >> 1);
	}
}

static int *c_long_gaps(int as1, int cnt1, const nd_mm128 *a, int min_gap, int *n_) /* collect_long_gaps */
{
	int i, n, *K;
	*n_ = 0;
	for (i = 1, n = 0; i < cnt1; ++i) {
		int gap = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - (int32_t)a[as1 + i - 1].x);
		if (gap < -min_gap || gap > min_gap) ++n;
	}
	if (n <= 1) return 0;
	K = (int*)malloc(sizeof(int) * (size_t)n);
	for (i = 1, n = 0; i < cnt1; ++i) {
		int gap = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - (int32_t)a[as1 + i - 1].x);
		if (gap < -min_gap || gap > min_gap) K[n++] = i;
	}
	*n_ = n;
	return K;
}

static void c_filter_bad_seeds(int as1, int cnt1, nd_mm128 *a, int min_gap, int diff_thres, int max_ext_len, int max_ext_cnt)
{
	int max_st, max_en, n, i, k, max, *K = c_long_gaps(as1, cnt1, a, min_gap, &n);
	if (K == 0) return;
	max = 0, max_st = max_en = -1;
	for (k = 0;; ++k) {
		int gap, l, n_ins = 0, n_del = 0, qs, rs, max_diff = 0, max_diff_l = -1;
		if (k == n || k >= max_en) {
			if (max_en > 0)
				for (i = K[max_st]; i < K[max_en]; ++i) a[as1 + i].y |= C_SEED_IGNORE;
			max = 0, max_st = max_en = -1;
			if (k == n) break;
		}
		i = K[k];
		gap = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - (int32_t)(a[as1 + i].x - a[as1 + i - 1].x);
		if (gap > 0) n_ins += gap; else n_del += -gap;
		qs = (int32_t)a[as1 + i - 1].y, rs = (int32_t)a[as1 + i - 1].x;
		for (l = k + 1; l < n && l <= k + max_ext_cnt; ++l) {
			int j = K[l], diff;
			if ((int32_t)a[as1 + j].y - qs > max_ext_len || (int32_t)a[as1 + j].x - rs > max_ext_len) break;
			gap = ((int32_t)a[as1 + j].y - (int32_t)a[as1 + j - 1].y) - (int32_t)(a[as1 + j].x - a[as1 + j - 1].x);
			if (gap > 0) n_ins += gap; else n_del += -gap;
			diff = n_ins + n_del - abs(n_ins - n_del);
			if (max_diff < diff) max_diff = diff, max_diff_l = l;
		}
		if (max_diff > diff_thres && max_diff > max) max = max_diff, max_st = k, max_en = max_diff_l;
	}
	free(K);
}

static void c_filter_bad_seeds_alt(int as1, int cnt1, nd_mm128 *a, int min_gap, int max_ext)
{
	int n, k, *K = c_long_gaps(as1, cnt1, a, min_gap, &n);
	if (K == 0) return;
	for (k = 0; k < n;) {
		int i = K[k], l;
		int gap1 = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - (int32_t)a[as1 + i - 1].x);
		int re1 = (int32_t)a[as1 + i].x, qe1 = (int32_t)a[as1 + i].y;
		gap1 = gap1 > 0 ? gap1 : -gap1;
		for (l = k + 1; l < n; ++l) {
			int j = K[l], gap2, q_span_pre, rs2, qs2, m;
			if ((int32_t)a[as1 + j].y - qe1 > max_ext || (int32_t)a[as1 + j].x - re1 > max_ext) break;
			gap2 = ((int32_t)a[as1 + j].y - (int32_t)a[as1 + j - 1].y) - (int32_t)(a[as1 + j].x - a[as1 + j - 1].x);
			q_span_pre = (int)(a[as1 + j - 1].y >> 32 & 0xff);
			rs2 = (int32_t)a[as1 + j - 1].x + q_span_pre, qs2 = (int32_t)a[as1 + j - 1].y + q_span_pre;
			m = rs2 - re1 < qs2 - qe1 ? rs2 - re1 : qs2 - qe1;
			gap2 = gap2 > 0 ? gap2 : -gap2;
			if (m > gap1 + gap2) break;
			re1 = (int32_t)a[as1 + j].x, qe1 = (int32_t)a[as1 + j].y;
			gap1 = gap2;
		}
		if (l > k + 1) {
			int j, end = K[l - 1];
			for (j = K[k]; j < end; ++j) a[as1 + j].y |= C_SEED_IGNORE;
			a[as1 + end].y |= C_SEED_LONG_JOIN;
		}
		k = l;
	}
	free(K);
}

static void c_fix_bad_ends(const c_reg *r, const nd_mm128 *a, int bw, int min_match, int32_t *as, int32_t *cnt)
{
	int32_t i, l, m;
	*as = r->as, *cnt = r->cnt;
	if (r->cnt < 3) return;
	m = l = (int32_t)(a[r->as].y >> 32 & 0xff);
	for (i = r->as + 1; i < r->as + r->cnt - 1; ++i) {
		int32_t lq, lr, mn, mx, q_span = (int32_t)(a[i].y >> 32 & 0xff);
		if (a[i].y & C_SEED_LONG_JOIN) break;
		lr = (int32_t)a[i].x - (int32_t)a[i - 1].x, lq = (int32_t)a[i].y - (int32_t)a[i - 1].y;
		mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
		if (mx - mn > l >> 1) *as = i;
		l += mn;
		m += mn < q_span ? mn : q_span;
		if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r->mlen >> 1) break;
	}
	*cnt = r->as + r->cnt - *as;
	m = l = (int32_t)(a[r->as + r->cnt - 1].y >> 32 & 0xff);
	for (i = r->as + r->cnt - 2; i > *as; --i) {
		int32_t lq, lr, mn, mx, q_span = (int32_t)(a[i + 1].y >> 32 & 0xff);
		if (a[i + 1].y & C_SEED_LONG_JOIN) break;
		lr = (int32_t)a[i + 1].x - (int32_t)a[i].x, lq = (int32_t)a[i + 1].y - (int32_t)a[i].y;
		mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
		if (mx - mn > l >> 1) *cnt = i + 1 - *as;
		l += mn;
		m += mn < q_span ? mn : q_span;
		if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r->mlen >> 1) break;
	}
}

static void c_fix_cigar(c_reg *r, const uint8_t *qseq, const uint8_t *tseq, int *qshift, int *tshift) /* mm_fix_cigar */
{
	int32_t toff = 0, qoff = 0, to_shrink = 0;
	uint32_t k, *cg = r->cigar, n = (uint32_t)r->n_cigar;
	*qshift = *tshift = 0;
	if (n <= 1) return;
	for (k = 0; k < n; ++k) {
		uint32_t op = cg[k] & 0xf, len = cg[k] >> 4;
		if (len == 0) to_shrink = 1;
		if (op == 0) toff += len, qoff += len;
		else if (op == 1 || op == 2) {
			if (k > 0 && k < n - 1 && (cg[k - 1] & 0xf) == 0 && (cg[k + 1] & 0xf) == 0) {
				int l, prev_len = (int)(cg[k - 1] >> 4);
				if (op == 1) { for (l = 0; l < prev_len; ++l) if (qseq[qoff - 1 - l] != qseq[qoff + len - 1 - l]) break; }
				else { for (l = 0; l < prev_len; ++l) if (tseq[toff - 1 - l] != tseq[toff + len - 1 - l]) break; }
				if (l > 0) cg[k - 1] -= (uint32_t)l << 4, cg[k + 1] += (uint32_t)l << 4, qoff -= l, toff -= l;
				if (l == prev_len) to_shrink = 1;
			}
			if (op == 1) qoff += len; else toff += len;
		} else if (op == 3) toff += len;
	}
	for (k = 0; k + 2 < n; ++k) {
		if ((cg[k] & 0xf) > 0 && (cg[k] & 0xf) + (cg[k + 1] & 0xf) == 3) {
			uint32_t l, s[3] = {0, 0, 0};
			for (l = k; l < n; ++l) {
				uint32_t op = cg[l] & 0xf;
				if (op == 1 || op == 2 || cg[l] >> 4 == 0) s[op] += cg[l] >> 4;
				else break;
			}
			if (s[1] > 0 && s[2] > 0 && l - k > 2) {
				cg[k] = s[1] << 4 | 1, cg[k + 1] = s[2] << 4 | 2;
				for (k += 2; k < l; ++k) cg[k] &= 0xf;
				to_shrink = 1;
			}
			k = l;
		}
	}
	if (to_shrink) {
		uint32_t l = 0;
		for (k = 0; k < n; ++k) if (cg[k] >> 4 != 0) cg[l++] = cg[k];
		n = l;
		for (k = l = 0; k < n; ++k)
			if (k == n - 1 || (cg[k] & 0xf) != (cg[k + 1] & 0xf)) cg[l++] = cg[k];
			else cg[k + 1] += cg[k] >> 4 << 4;
		n = l;
	}
	if ((cg[0] & 0xf) == 1 || (cg[0] & 0xf) == 2) {
		int32_t l = (int32_t)(cg[0] >> 4);
		if ((cg[0] & 0xf) == 1) { if (r->rev) r->qe -= l; else r->qs += l; *qshift = l; }
		else r->rs += l, *tshift = l;
		--n;
		memmove(cg, cg + 1, 4 * (size_t)n);
	}
	r->n_cigar = (int)n;
}

static void c_update_extra(const c_ctx *C, c_reg *r, const uint8_t *qseq, const uint8_t *tseq) /* mm_update_extra */
{
	int qshift, tshift, k;
	int32_t s = 0, max = 0, toff = 0, qoff = 0;
	if (!r->has_p) return;
	c_fix_cigar(r, qseq, tseq, &qshift, &tshift);
	qseq += qshift, tseq += tshift;
	r->blen = r->mlen = 0;
	for (k = 0; k < r->n_cigar; ++k) {
		uint32_t op = r->cigar[k] & 0xf, len = r->cigar[k] >> 4, l;
		if (op == 0) {
			int n_ambi = 0, n_diff = 0;
			for (l = 0; l < len; ++l) {
				int cq = qseq[qoff + l], ct = tseq[toff + l];
				if (ct > 3 || cq > 3) ++n_ambi;
				else if (ct != cq) ++n_diff;
				s += C->mat[ct * 5 + cq];
				if (s < 0) s = 0; else max = max > s ? max : s;
			}
			r->blen += len - n_ambi, r->mlen += len - (n_ambi + n_diff);
			toff += len, qoff += len;
		} else if (op == 1 || op == 2) {
			int n_ambi = 0;
			for (l = 0; l < len; ++l) if ((op == 1 ? qseq[qoff + l] : tseq[toff + l]) > 3) ++n_ambi;
			r->blen += len - n_ambi;
			s -= C->ao->q + C->ao->e * (int32_t)len;
			if (s < 0) s = 0;
			if (op == 1) qoff += len; else toff += len;
		} else if (op == 3) toff += len;
	}
	r->dp_max = max;
}

/* mm_align_pair: max_sw_mat, then ksw_extd2_sse (the two gap pieces differ in every preset of this path) */
static void c_align_pair(c_ctx *C, int qlen, const uint8_t *qseq, int tlen, const uint8_t *tseq, int w, int end_bonus, int zdrop, int flag, c_ksw_result *ez)
{
	const nd_aln_opt *o = C->ao;
	if (qlen + tlen + 4 > C->cg_cap) { C->cg_cap = (qlen + tlen + 4) * 2; C->cg = (uint32_t*)realloc(C->cg, 4 * (size_t)C->cg_cap); }
	if (o->max_sw_mat > 0 && (int64_t)tlen * qlen > o->max_sw_mat) {
		memset(ez, 0, sizeof(*ez));
		ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1, ez->score = ez->mqe = ez->mte = -0x40000000, ez->zdropped = 1;
		return;
	}
	nd_oracle_ksw_extd2(qlen, qseq, tlen, tseq, 5, C->mat, (int8_t)o->q, (int8_t)o->e, (int8_t)o->q2, (int8_t)o->e2, w, zdrop, end_bonus, flag, ez, C->cg, C->cg_cap);
	if (getenv("ND_ORACLE_DBG")) { /* the format of minimap2's --print-aln-seq (align.c:315-322,333-339), for a diff against it */
		int i;
		fprintf(stderr, "===> q=(%d,%d), e=(%d,%d), bw=%d, flag=%d, zdrop=%d <===\n", o->q, o->q2, o->e, o->e2, w, flag, o->zdrop);
		for (i = 0; i < tlen; ++i) fputc("ACGTN"[tseq[i]], stderr);
		fputc('\n', stderr);
		for (i = 0; i < qlen; ++i) fputc("ACGTN"[qseq[i]], stderr);
		fputc('\n', stderr);
		fprintf(stderr, "score=%d, cigar=", ez->score);
		for (i = 0; i < ez->n_cigar; ++i) fprintf(stderr, "%d%c", C->cg[i] >> 4, "MIDN"[C->cg[i] & 0xf]);
		fprintf(stderr, "\n");
	}
}

static int c_test_zdrop(c_ctx *C, const uint8_t *qseq, const uint8_t *tseq, int n_cigar, const uint32_t *cigar) /* mm_test_zdrop */
{
	const nd_aln_opt *o = C->ao;
	int k, pos[2][2] = {{-1, -1}, {-1, -1}}, q_len, t_len;
	int32_t score = 0, max = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
#define C_UPD(sc_, ii_, jj_) do { int32_t sc = (sc_); int ii = (ii_), jj = (jj_); \
		if (sc < max) { int li = ii - max_i, lj = jj - max_j, diff = li > lj ? li - lj : lj - li, z = max - sc - diff * o->e; \
			if (z > max_zdrop) max_zdrop = z, pos[0][0] = max_i, pos[0][1] = ii + 1, pos[1][0] = max_j, pos[1][1] = jj + 1; \
		} else max = sc, max_i = ii, max_j = jj; } while (0)
	for (k = 0; k < n_cigar; ++k) {
		uint32_t l, op = cigar[k] & 0xf, len = cigar[k] >> 4;
		if (op == 0) {
			for (l = 0; l < len; ++l) {
				score += C->mat[tseq[i + l] * 5 + qseq[j + l]];
				C_UPD(score, i + (int)l, j + (int)l);
			}
			i += len, j += len;
		} else if (op == 1 || op == 2 || op == 3) {
			score -= o->q + o->e * (int32_t)len;
			if (op == 1) j += len; else i += len;
			C_UPD(score, i, j);
		}
	}
#undef C_UPD
	q_len = pos[1][1] - pos[1][0], t_len = pos[0][1] - pos[0][0];
	if (max_zdrop > o->zdrop_inv && q_len < C->opt->max_gap && t_len < C->opt->max_gap) {
		uint8_t *q2 = (uint8_t*)malloc((size_t)(q_len > 0 ? q_len : 1));
		int qe, te, sc;
		for (i = 0; i < q_len; ++i) { int c = qseq[pos[1][1] - i - 1]; q2[i] = (uint8_t)(c >= 4 ? 4 : 3 - c); }
		sc = nd_oracle_ksw_ll_i16(q_len, q2, t_len, tseq + pos[0][0], C->mat, o->q, o->e, &qe, &te);
		free(q2);
		if (sc >= C->opt->min_sc * o->a && sc >= o->min_dp_max) return 2;
	}
	return max_zdrop > o->zdrop ? 1 : 0;
}

static void c_split_reg(c_reg *r, c_reg *r2, int n, int qlen, const nd_mm128 *a) /* mm_split_reg */
{
	if (n <= 0 || n >= r->cnt) return;
	*r2 = *r;
	r2->id = -1, r2->has_p = 0, r2->cigar = 0, r2->n_cigar = r2->m_cigar = 0, r2->dp_max = 0, r2->split_inv = 0;
	r2->cnt = r->cnt - n;
	r2->score = (int32_t)(r->score * ((float)r2->cnt / r->cnt) + .499);
	r2->as = r->as + n;
	if (r->parent == r->id) r2->parent = -2;
	c_reg_set_coor(r2, qlen, a);
	r->cnt -= r2->cnt;
	r->score -= r2->score;
	c_reg_set_coor(r, qlen, a);
	r->split |= 1, r2->split |= 2;
}

static void c_rev(int len, uint8_t *s) { int i; for (i = 0; i < len >> 1; ++i) { uint8_t t = s[i]; s[i] = s[len - 1 - i], s[len - 1 - i] = t; } }

static void c_align1(c_ctx *C, c_reg *r, c_reg *r2) /* mm_align1, neither short reads nor splicing */
{
	const nd_mm_opt *opt = C->opt;
	const nd_aln_opt *o = C->ao;
	nd_mm128 *a = C->a;
	const int qlen = C->qlen, n_a = C->n_a;
	int32_t rid, rev, as1, cnt1, i, l, bw, dropped = 0, rs0, re0, qs0, qe0, rs, re, qs, qe, rs1, qs1, re1, qe1, tlen;
	uint8_t *tseq, *qseq;
	c_ksw_result ez;
	r2->cnt = 0;
	if (r->cnt == 0) return;
	rid = (int32_t)(a[r->as].x << 1 >> 33), rev = (int32_t)(a[r->as].x >> 63), tlen = (int32_t)C->ix->len[rid];
	bw = (int)(opt->bw * 1.5 + 1.);
	c_fix_bad_ends(r, a, opt->bw, opt->min_sc * 2, &as1, &cnt1);
	c_filter_bad_seeds(as1, cnt1, a, 10, 40, opt->max_gap >> 1, 10);
	c_filter_bad_seeds_alt(as1, cnt1, a, 30, opt->max_gap >> 1);
	c_adjust_minier(C, &a[as1], &rs, &qs);
	c_adjust_minier(C, &a[as1 + cnt1 - 1], &re, &qe);
	rs0 = (int32_t)a[r->as].x + 1 - (int32_t)(a[r->as].y >> 32 & 0xff);
	qs0 = (int32_t)a[r->as].y + 1 - (int32_t)(a[r->as].y >> 32 & 0xff);
	if (rs0 < 0) rs0 = 0;
	rs1 = qs1 = 0;
	for (i = r->as - 1, l = 0; i >= 0 && a[i].x >> 32 == a[r->as].x >> 32; --i) {
		int32_t x = (int32_t)a[i].x + 1 - (int32_t)(a[i].y >> 32 & 0xff), y = (int32_t)a[i].y + 1 - (int32_t)(a[i].y >> 32 & 0xff);
		if (x < rs0 && y < qs0) {
			if (++l > opt->min_cnt) {
				l = rs0 - x > qs0 - y ? rs0 - x : qs0 - y;
				rs1 = rs0 - l, qs1 = qs0 - l;
				if (rs1 < 0) rs1 = 0;
				break;
			}
		}
	}
	if (qs > 0 && rs > 0) {
		l = qs < opt->max_gap ? qs : opt->max_gap;
		qs1 = qs1 > qs - l ? qs1 : qs - l;
		qs0 = qs0 < qs1 ? qs0 : qs1;
		l += l * o->a > o->q ? (l * o->a - o->q) / o->e : 0;
		l = l < opt->max_gap ? l : opt->max_gap;
		l = l < rs ? l : rs;
		rs1 = rs1 > rs - l ? rs1 : rs - l;
		rs0 = rs0 < rs1 ? rs0 : rs1;
		rs0 = rs0 < rs ? rs0 : rs;
	} else rs0 = rs, qs0 = qs;
	re0 = (int32_t)a[r->as + r->cnt - 1].x + 1;
	qe0 = (int32_t)a[r->as + r->cnt - 1].y + 1;
	re1 = tlen, qe1 = qlen;
	for (i = r->as + r->cnt, l = 0; i < n_a && a[i].x >> 32 == a[r->as].x >> 32; ++i) {
		int32_t x = (int32_t)a[i].x + 1, y = (int32_t)a[i].y + 1;
		if (x > re0 && y > qe0) {
			if (++l > opt->min_cnt) {
				l = x - re0 > y - qe0 ? x - re0 : y - qe0;
				re1 = re0 + l, qe1 = qe0 + l;
				break;
			}
		}
	}
	if (qe < qlen && re < tlen) {
		l = qlen - qe < opt->max_gap ? qlen - qe : opt->max_gap;
		qe1 = qe1 < qe + l ? qe1 : qe + l;
		qe0 = qe0 > qe1 ? qe0 : qe1;
		l += l * o->a > o->q ? (l * o->a - o->q) / o->e : 0;
		l = l < opt->max_gap ? l : opt->max_gap;
		l = l < tlen - re ? l : tlen - re;
		re1 = re1 < re + l ? re1 : re + l;
		re0 = re0 > re1 ? re0 : re1;
	} else re0 = re, qe0 = qe;
	if (a[r->as].y & SEED_SELF) {
		int max_ext = r->qs > r->rs ? r->qs - r->rs : r->rs - r->qs;
		if (r->rs - rs0 > max_ext) rs0 = r->rs - max_ext;
		if (r->qs - qs0 > max_ext) qs0 = r->qs - max_ext;
		max_ext = r->qe > r->re ? r->qe - r->re : r->re - r->qe;
		if (re0 - r->re > max_ext) re0 = r->re + max_ext;
		if (qe0 - r->qe > max_ext) qe0 = r->qe + max_ext;
	}
	if (re0 <= rs0) return;
	tseq = (uint8_t*)malloc((size_t)(re0 - rs0) + 16);

	if (qs > 0 && rs > 0) { /* left extension */
		qseq = &C->qseq0[rev][qs0];
		c_getseq(C, rid, rs0, rs, tseq);
		c_rev(qs - qs0, qseq);
		c_rev(rs - rs0, tseq);
		c_align_pair(C, qs - qs0, qseq, rs - rs0, tseq, bw, o->end_bonus, r->split_inv ? o->zdrop_inv : o->zdrop, C_EZ_EXTZ_ONLY | C_EZ_RIGHT | C_EZ_REV_CIGAR, &ez);
		if (ez.n_cigar > 0) c_append(r, ez.n_cigar, C->cg);
		rs1 = rs - (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
		qs1 = qs - (ez.reach_end ? qs - qs0 : ez.max_q + 1);
		c_rev(qs - qs0, qseq);
	} else rs1 = rs, qs1 = qs;
	re1 = rs, qe1 = qs;

	for (i = 1; i < cnt1; ++i) { /* gap filling */
		if ((a[as1 + i].y & (C_SEED_IGNORE | SEED_TANDEM)) && i != cnt1 - 1) continue;
		c_adjust_minier(C, &a[as1 + i], &re, &qe);
		re1 = re, qe1 = qe;
		if (i == cnt1 - 1 || (a[as1 + i].y & C_SEED_LONG_JOIN) || (qe - qs >= o->min_ksw_len && re - rs >= o->min_ksw_len)) {
			int j, bw1 = bw, zdrop_code;
			if (a[as1 + i].y & C_SEED_LONG_JOIN) bw1 = qe - qs > re - rs ? qe - qs : re - rs;
			qseq = &C->qseq0[rev][qs];
			c_getseq(C, rid, rs, re, tseq);
			c_align_pair(C, qe - qs, qseq, re - rs, tseq, bw1, -1, o->zdrop, C_EZ_APPROX_MAX, &ez);
			if ((zdrop_code = c_test_zdrop(C, qseq, tseq, ez.n_cigar, C->cg)) != 0)
				c_align_pair(C, qe - qs, qseq, re - rs, tseq, bw1, -1, zdrop_code == 2 ? o->zdrop_inv : o->zdrop, 0, &ez);
			if (ez.n_cigar > 0) c_append(r, ez.n_cigar, C->cg);
			if (ez.zdropped) {
				for (j = i - 1; j >= 0; --j)
					if ((int32_t)a[as1 + j].x <= rs + ez.max_t) break;
				dropped = 1;
				if (j < 0) j = 0;
				re1 = rs + (ez.max_t + 1);
				qe1 = qs + (ez.max_q + 1);
				if (cnt1 - (j + 1) >= opt->min_cnt) {
					c_split_reg(r, r2, as1 + j + 1 - r->as, qlen, a);
					if (r2->cnt > 0 && zdrop_code == 2) r2->split_inv = 1;
				}
				break;
			}
			rs = re, qs = qe;
		}
	}

	if (!dropped && qe < qe0 && re < re0) { /* right extension */
		qseq = &C->qseq0[rev][qe];
		c_getseq(C, rid, re, re0, tseq);
		c_align_pair(C, qe0 - qe, qseq, re0 - re, tseq, bw, o->end_bonus, o->zdrop, C_EZ_EXTZ_ONLY, &ez);
		if (ez.n_cigar > 0) c_append(r, ez.n_cigar, C->cg);
		re1 = re + (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
		qe1 = qe + (ez.reach_end ? qe0 - qe : ez.max_q + 1);
	}
	r->rs = rs1, r->re = re1;
	if (rev) r->qs = qlen - qe1, r->qe = qlen - qs1;
	else r->qs = qs1, r->qe = qe1;
	if (r->has_p) {
		uint8_t *t2 = (uint8_t*)malloc((size_t)(re1 > rs1 ? re1 - rs1 : 0) + 16);
		c_getseq(C, rid, rs1, re1, t2);
		c_update_extra(C, r, &C->qseq0[r->rev][qs1], t2);
		free(t2);
	}
	free(tseq);
}

static int c_align1_inv(c_ctx *C, const c_reg *r1, const c_reg *r2, c_reg *r_inv) /* mm_align1_inv */
{
	const nd_mm_opt *opt = C->opt;
	const nd_aln_opt *o = C->ao;
	int tl, ql, score, ret = 0, q_off, t_off;
	uint8_t *tseq, *qseq;
	c_ksw_result ez;
	memset(r_inv, 0, sizeof(*r_inv));
	if (!(r1->split & 1) || !(r2->split & 2)) return 0;
	if (r1->id != r1->parent && r1->parent != -2) return 0;
	if (r2->id != r2->parent && r2->parent != -2) return 0;
	if (r1->rid != r2->rid || r1->rev != r2->rev) return 0;
	ql = r1->rev ? r1->qs - r2->qe : r2->qs - r1->qe;
	tl = r2->rs - r1->re;
	if (ql < opt->min_sc || ql > opt->max_gap) return 0;
	if (tl < opt->min_sc || tl > opt->max_gap) return 0;
	tseq = (uint8_t*)malloc((size_t)tl + 16);
	c_getseq(C, r1->rid, r1->re, r2->rs, tseq);
	qseq = r1->rev ? &C->qseq0[0][r2->qe] : &C->qseq0[1][C->qlen - r2->qs];
	c_rev(ql, qseq);
	c_rev(tl, tseq);
	score = nd_oracle_ksw_ll_i16(ql, qseq, tl, tseq, C->mat, o->q, o->e, &q_off, &t_off);
	c_rev(ql, qseq);
	c_rev(tl, tseq);
	if (score < o->min_dp_max) goto end;
	q_off = ql - (q_off + 1), t_off = tl - (t_off + 1);
	c_align_pair(C, ql - q_off, qseq + q_off, tl - t_off, tseq + t_off, (int)(opt->bw * 1.5), -1, o->zdrop, C_EZ_EXTZ_ONLY, &ez);
	if (ez.n_cigar == 0) goto end;
	c_append(r_inv, ez.n_cigar, C->cg);
	r_inv->id = -1, r_inv->parent = -1, r_inv->inv = 1, r_inv->rev = !r1->rev, r_inv->rid = r1->rid;
	if (r_inv->rev == 0) r_inv->qs = r2->qe + q_off, r_inv->qe = r_inv->qs + ez.max_q + 1;
	else r_inv->qe = r2->qs - q_off, r_inv->qs = r_inv->qe - (ez.max_q + 1);
	r_inv->rs = r1->re + t_off, r_inv->re = r_inv->rs + ez.max_t + 1;
	c_update_extra(C, r_inv, &qseq[q_off], &tseq[t_off]);
	ret = 1;
end:
	free(tseq);
	return ret;
}

static c_reg *c_insert_reg(const c_reg *r, int i, int *n_regs, c_reg *regs) /* mm_insert_reg */
{
	regs = (c_reg*)realloc(regs, ((size_t)*n_regs + 1) * sizeof(c_reg));
	if (i + 1 != *n_regs) memmove(&regs[i + 2], &regs[i + 1], sizeof(c_reg) * (size_t)(*n_regs - i - 1));
	regs[i + 1] = *r;
	++*n_regs;
	return regs;
}

/* one read: chains -> mm_align_skeleton -> writer; appends its records to out.  Returns the bytes written. */
static int64_t c_read(c_ctx *C, uint32_t qid, const uint8_t *qcodes, int qlen, int mid_occ, const uint32_t *tids, uint32_t *prev, uint8_t *out)
{
	const nd_mm_opt *opt = C->opt;
	char qname[12];
	nd_mm128 *mv, *a;
	uint64_t *u;
	int64_t n_mv, n_a = 0, i, n_b, n = 0;
	int n_u, n_regs, k;
	nd_mm_reg *g;
	c_reg *regs;
	nd_mm128 *aux;
	if (qlen <= 0) return 0;
	sprintf(qname, "%u", qid);
	mv = (nd_mm128*)malloc(sizeof(nd_mm128) * ((size_t)qlen + 1));
	n_mv = nd_mm_sketch(qcodes, qlen, opt->w, opt->k, 0, opt->hpc, mv);
	for (i = 0; i < n_mv; ++i) { int n_occ; index_get(C->ix, mv[i].x >> 8, &n_occ); if (n_occ < mid_occ) n_a += n_occ; }
	a = (nd_mm128*)malloc(sizeof(nd_mm128) * (size_t)(n_a > 0 ? n_a : 1));
	n_a = nd_mm_seeds(C->ix, opt, qname, qlen, mid_occ, mv, n_mv, a, 1);
	u = (uint64_t*)malloc(8 * (size_t)(n_a > 0 ? n_a : 1));
	n_u = nd_mm_chain(opt, n_a, a, u, &n_b, 0, 0);
	if (n_u == 0 && nd_mm_rechain_wanted(C->ix, opt, mid_occ, mv, n_mv)) { /* map.c:553-575 */
		free(a); free(u);
		for (i = 0, n_a = 0; i < n_mv; ++i) { int n_occ; index_get(C->ix, mv[i].x >> 8, &n_occ); if (n_occ < opt->max_occ) n_a += n_occ; }
		a = (nd_mm128*)malloc(sizeof(nd_mm128) * (size_t)(n_a > 0 ? n_a : 1));
		n_a = nd_mm_seeds(C->ix, opt, qname, qlen, opt->max_occ, mv, n_mv, a, 1);
		u = (uint64_t*)malloc(8 * (size_t)(n_a > 0 ? n_a : 1));
		n_u = nd_mm_chain(opt, n_a, a, u, &n_b, 0, 0);
	}
	free(mv);
	if (n_u <= 0) { free(a); free(u); return 0; }
	g = (nd_mm_reg*)malloc(sizeof(nd_mm_reg) * (size_t)n_u);
	nd_mm_gen_regs(nd_mm_read_hash(qname, qlen, opt->seed), qlen, n_u, u, a, g);
	n_regs = n_u;
	regs = (c_reg*)calloc((size_t)n_regs, sizeof(c_reg));
	for (k = 0; k < n_regs; ++k) {
		c_reg *r = &regs[k];
		r->id = k, r->parent = -1, r->cnt = g[k].cnt, r->as = g[k].as, r->score = g[k].score, r->hash = g[k].hash;
		c_reg_set_coor(r, qlen, a);
	}
	free(g);
	C->qlen = qlen, C->a = a, C->n_a = (int)n_b;
	C->qseq0[0] = (uint8_t*)malloc((size_t)qlen * 2 + 16);
	C->qseq0[1] = C->qseq0[0] + qlen;
	memset(C->qseq0[0] + 2 * (size_t)qlen, 0, 16);
	for (k = 0; k < qlen; ++k) C->qseq0[0][k] = qcodes[k], C->qseq0[1][qlen - 1 - k] = (uint8_t)(qcodes[k] < 4 ? 3 - qcodes[k] : 4);
	for (k = 0; k < n_regs; ++k) { /* mm_align_skeleton's loop */
		c_reg r2;
		memset(&r2, 0, sizeof(r2));
		c_align1(C, &regs[k], &r2);
		if (r2.cnt > 0) regs = c_insert_reg(&r2, k, &n_regs, regs);
		if (k > 0 && regs[k].split_inv) {
			c_reg rv;
			if (c_align1_inv(C, &regs[k - 1], &regs[k], &rv)) {
				regs = c_insert_reg(&rv, k, &n_regs, regs);
				++k;
			}
		}
	}
	{ /* mm_filter_regs */
		int m = 0;
		for (k = 0; k < n_regs; ++k) {
			c_reg *r = &regs[k];
			int flt = 0;
			if (!r->inv && r->cnt < opt->min_cnt) flt = 1;
			if (r->has_p) {
				if (r->mlen < opt->min_sc) flt = 1;
				else if (r->dp_max < C->ao->min_dp_max) flt = 1;
			}
			if (flt) free(r->cigar);
			else regs[m++] = *r;
		}
		n_regs = m;
	}
	if (n_regs > 1) { /* mm_hit_sort */
		int n_aux = 0;
		c_reg *t = (c_reg*)malloc(sizeof(c_reg) * (size_t)n_regs);
		aux = (nd_mm128*)malloc(sizeof(nd_mm128) * (size_t)n_regs);
		for (k = 0; k < n_regs; ++k)
			if (regs[k].inv || regs[k].cnt > 0) {
				aux[n_aux].x = (uint64_t)(uint32_t)(regs[k].has_p ? regs[k].dp_max : regs[k].score) << 32 | regs[k].hash;
				aux[n_aux++].y = (uint64_t)k;
			} else free(regs[k].cigar);
		nd_mm_rs_sort128(aux, n_aux);
		for (k = n_aux - 1; k >= 0; --k) t[n_aux - 1 - k] = regs[aux[k].y];
		memcpy(regs, t, sizeof(c_reg) * (size_t)n_aux);
		n_regs = n_aux;
		free(aux); free(t);
	}
	{ /* the writer: the fields nd_mm_encode reads */
		nd_mm_reg *w = (nd_mm_reg*)calloc((size_t)(n_regs > 0 ? n_regs : 1), sizeof(nd_mm_reg));
		for (k = 0; k < n_regs; ++k) {
			w[k].rev = regs[k].rev, w[k].rid = regs[k].rid, w[k].qs = regs[k].qs, w[k].qe = regs[k].qe, w[k].rs = regs[k].rs, w[k].re = regs[k].re;
			w[k].mlen = regs[k].mlen, w[k].blen = regs[k].blen;
			free(regs[k].cigar);
		}
		n = nd_mm_encode(C->ix, opt, qid, qlen, tids, w, n_regs, prev, out);
		free(w);
	}
	free(regs); free(C->qseq0[0]); free(a); free(u);
	return n;
}

/* `minimap2-nd --step 1 -c target query` for ONE index part; arguments as nd_mm_step1 */
int64_t nd_mm_step1_cigar(const nd_mm_opt *opt, const nd_aln_opt *ao, float mid_occ_frac, int mid_occ_fixed,
                          int32_t n_t, const uint8_t *tcodes, const uint64_t *toff, const uint32_t *tlen, const uint32_t *tids,
                          int32_t n_q, const uint8_t *qcodes, const uint64_t *qoff, const uint32_t *qlen, const uint32_t *qids,
                          uint8_t *out, int64_t out_cap, int32_t *mid_occ_out, uint32_t *prev_io)
{
	nd_mm_index *ix = nd_mm_index_build(n_t, tcodes, toff, tlen, tids, opt->w, opt->k, opt->hpc);
	int mid_occ = mid_occ_fixed > 0 ? mid_occ_fixed : nd_mm_index_mid_occ(ix, mid_occ_frac);
	uint32_t prev[2] = { prev_io ? prev_io[0] : 0, prev_io ? prev_io[1] : 0 };
	int64_t n = 0;
	int i, j;
	c_ctx C;
	memset(&C, 0, sizeof(C));
	C.opt = opt, C.ao = ao, C.ix = ix, C.tcodes = tcodes, C.toff = toff;
	{ /* ksw_gen_simple_mat, align.c:9-22 */
		int8_t a = (int8_t)(ao->a < 0 ? -ao->a : ao->a), b = (int8_t)(ao->b > 0 ? -ao->b : ao->b), amb = (int8_t)(ao->sc_ambi > 0 ? -ao->sc_ambi : ao->sc_ambi);
		for (i = 0; i < 4; ++i) { for (j = 0; j < 4; ++j) C.mat[i * 5 + j] = i == j ? a : b; C.mat[i * 5 + 4] = amb; }
		for (j = 0; j < 5; ++j) C.mat[20 + j] = amb;
	}
	if (mid_occ_out) *mid_occ_out = mid_occ;
	for (i = 0; i < n_q; ++i) {
		/* (a read's records: at most one per chain piece; pieces <= anchors / min_cnt) */
		if (n + 40LL * ((int64_t)qlen[i] / 4 + 1024) > out_cap) { n = -(n + 40LL * ((int64_t)qlen[i] / 4 + 1024)); break; }
		n += c_read(&C, qids[i], qcodes + qoff[i], (int)qlen[i], mid_occ, tids, prev, out + n);
	}
	free(C.cg);
	nd_mm_index_free(ix);
	if (prev_io && n >= 0) prev_io[0] = prev[0], prev_io[1] = prev[1];
	return n;
}
