/* oracle/ksw2_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see nd_oracle.h).
 *
 * CPU restatement, in plain scalar C, of minimap2's two-piece affine-gap extension kernel
 *     ksw_extd2_sse      minimap2/ksw2_extd2_sse.c:26-399   (caller: mm_align_pair, minimap2/align.c:331; only with -c / -a)
 * and of the helpers it uses from minimap2/ksw2.h: ksw_reset_extz (:161-166), ksw_apply_zdrop (:168-184), ksw_backtrack (:119-159),
 * ksw_push_cigar (:101-112).
 *
 * The algorithm is Suzuki & Kasahara's difference recurrence: anti-diagonal r = i + j of the (target i, query j) matrix is computed
 * from anti-diagonal r - 1 with 8-bit differences u, v (of H), x, y (first gap piece), x2, y2 (second gap piece), all indexed by the
 * target position t.  What makes a bit-exact restatement more than the recurrence:
 *   * the reference works on 16-byte blocks: the cells of a diagonal are t in [st, en] with st rounded down and en rounded up to
 *     multiples of 16 around the true range [st0, en0].  The extra cells are computed too, from whatever the arrays hold, and the
 *     next diagonal may take them as its boundary ("(r-1, s-1) calculated in the last round"), so they are part of the result;
 *   * the per-cell scores s[] are refreshed in runs of 16 starting at st0, so the extra cells below st0 see scores of earlier
 *     diagonals and those above en0 see scores of positions that are not in the matrix (target bases paired with the zero padding
 *     behind the reversed query);
 *   * all arithmetic on u, v, x, y, x2, y2, s is 8-bit and wraps;
 *   * the exact maximum is searched four positions at a time, which fixes which of several equal maxima wins.
 * Pinned against the compiled reference function (oracle/_ref/libksw2ref.so) on fuzzed problems and against committed vectors
 * (tests/golden/ksw2.npz) by tests/test_oracle_ksw2.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NEG_INF (-0x40000000) /* KSW_NEG_INF */
#define F_SCORE_ONLY 0x01
#define F_RIGHT 0x02
#define F_GENERIC_SC 0x04
#define F_APPROX_MAX 0x08
#define F_APPROX_DROP 0x10
#define F_EXTZ_ONLY 0x40
#define F_REV_CIGAR 0x80

typedef struct {
	int32_t max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, n_cigar, reach_end;
} nd_ksw_result;

static int zdrop_test(nd_ksw_result *ez, int32_t H, int r, int t, int zdrop, int8_t e) /* ksw_apply_zdrop, rotated form */
{
	if (H > ez->max) {
		ez->max = H, ez->max_t = t, ez->max_q = r - t;
	} else if (t >= ez->max_t && r - t >= ez->max_q) {
		const int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l = tl > ql ? tl - ql : ql - tl;
		if (zdrop >= 0 && ez->max - H > zdrop + l * e) { ez->zdropped = 1; return 1; }
	}
	return 0;
}

static int push_op(uint32_t *cigar, int n, int cap, uint32_t op, int len)
{
	if (n == 0 || op != (cigar[n - 1] & 0xf)) {
		if (n < cap) cigar[n] = (uint32_t)len << 4 | op;
		return n + 1;
	}
	if (n <= cap) cigar[n - 1] += (uint32_t)len << 4;
	return n;
}

/* ksw_backtrack with is_rot = 1, min_intron_len = 0 */
static int backtrack(int is_rev, const uint8_t *p, const int *off, const int *off_end, int n_col, int i0, int j0, uint32_t *cigar, int cap)
{
	int n = 0, i = i0, j = j0, state = 0, k;
	while (i >= 0 && j >= 0) {
		const int r = i + j;
		int force = -1;
		uint32_t tmp;
		if (i < off[r]) force = 2;
		if (i > off_end[r]) force = 1;
		tmp = force < 0 ? p[(size_t)r * n_col + i - off[r]] : 0;
		if (state == 0) state = tmp & 7;
		else if (!(tmp >> (state + 2) & 1)) state = 0;
		if (state == 0) state = tmp & 7;
		if (force >= 0) state = force;
		if (state == 0) n = push_op(cigar, n, cap, 0, 1), --i, --j;
		else if (state == 1 || state == 3) n = push_op(cigar, n, cap, 2, 1), --i;
		else n = push_op(cigar, n, cap, 1, 1), --j;
	}
	if (i >= 0) n = push_op(cigar, n, cap, 2, i + 1);
	if (j >= 0) n = push_op(cigar, n, cap, 1, j + 1);
	if (!is_rev && n <= cap)
		for (k = 0; k < n >> 1; ++k) { const uint32_t t = cigar[k]; cigar[k] = cigar[n - 1 - k], cigar[n - 1 - k] = t; }
	return n;
}

/* Returns 0 (result in *ez, CIGAR in cigar[0 .. ez->n_cigar); n_cigar > cigar_cap means the buffer was too small). */
int nd_oracle_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat, int8_t q, int8_t e,
                        int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, nd_ksw_result *ez, uint32_t *cigar, int cigar_cap)
{
	const int with_cigar = !(flag & F_SCORE_ONLY), approx_max = !!(flag & F_APPROX_MAX);
	/* (the reference initialises `qe` where it is declared, BEFORE the two gap pieces may be swapped, and uses that value for the
	 * score of the first cell only: with q2 + e2 < q + e every score is off by the difference -- kept, it is what callers get) */
	const int qe_first = q + e;
	int r, t, qe, qe2, n_col, tl16, last_st = -1, last_en = -1, max_sc, min_sc, long_thres, long_diff;
	int8_t *u, *v, *x, *y, *x2, *y2, *s, sc_mch, sc_mis, sc_N;
	uint8_t *sf, *qr, *p = 0;
	int32_t *H = 0, H0 = 0, last_H0_t = 0;
	int *off = 0, *off_end = 0;

	ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
	ez->max = 0, ez->score = ez->mqe = ez->mte = NEG_INF;
	ez->n_cigar = 0, ez->zdropped = 0, ez->reach_end = 0;
	if (m <= 1 || qlen <= 0 || tlen <= 0) return 0;
	if (q2 + e2 < q + e) { int8_t z_ = q; q = q2, q2 = z_, z_ = e, e = e2, e2 = z_; }
	qe = q + e, qe2 = q2 + e2;
	sc_mch = mat[0], sc_mis = mat[1], sc_N = mat[m * m - 1] == 0 ? (int8_t)-e2 : mat[m * m - 1];
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	tl16 = (tlen + 15) / 16 * 16;
	n_col = qlen < tlen ? qlen : tlen;
	n_col = (((n_col < w + 1 ? n_col : w + 1) + 15) / 16 + 1) * 16; /* bytes per row of the backtrack matrix */
	for (t = 1, max_sc = mat[0], min_sc = mat[1]; t < m * m; ++t) {
		max_sc = max_sc > mat[t] ? max_sc : mat[t];
		min_sc = min_sc < mat[t] ? min_sc : mat[t];
	}
	(void)max_sc;
	if (-min_sc > 2 * (q + e)) return 0;
	long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	long_diff = long_thres * (e - e2) - (q2 - q) - e2;

	u = (int8_t*)malloc((size_t)tl16 * 7 + 32);
	v = u + tl16, x = v + tl16, y = x + tl16, x2 = y + tl16, y2 = x2 + tl16, s = y2 + tl16;
	memset(u, -q - e, (size_t)tl16 * 4);
	memset(x2, -q2 - e2, (size_t)tl16 * 2);
	memset(s, 0, (size_t)tl16 + 32);
	sf = (uint8_t*)calloc((size_t)tl16 + 32, 1);           /* the target, zero behind it */
	qr = (uint8_t*)calloc((size_t)qlen + 48, 1);           /* the query reversed, zero behind it */
	memcpy(sf, target, (size_t)tlen);
	for (t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
	if (!approx_max) {
		H = (int32_t*)malloc(sizeof(int32_t) * (size_t)tl16);
		for (t = 0; t < tl16; ++t) H[t] = NEG_INF;
	}
	if (with_cigar) {
		p = (uint8_t*)calloc((size_t)(qlen + tlen - 1) * n_col + 16, 1);
		off = (int*)malloc(sizeof(int) * 2 * (size_t)(qlen + tlen - 1));
		off_end = off + qlen + tlen - 1;
	}

	for (r = 0; r < qlen + tlen - 1; ++r) {
		int st = 0, en = tlen - 1, st0, en0;
		int8_t x1, x21, v1;
		const uint8_t *qrr = qr + (qlen - 1 - r); /* qrr[t] = query[r - t] */
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
		if (en > (r + w) >> 1) en = (r + w) >> 1;
		if (st > en) { ez->zdropped = 1; break; }
		st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) x1 = x[st - 1], x21 = x2[st - 1], v1 = v[st - 1];
			else x1 = (int8_t)(-q - e), x21 = (int8_t)(-q2 - e2), v1 = (int8_t)(-q - e);
		} else {
			x1 = (int8_t)(-q - e), x21 = (int8_t)(-q2 - e2);
			v1 = (int8_t)(r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2);
		}
		if (en >= r) {
			y[r] = (int8_t)(-q - e), y2[r] = (int8_t)(-q2 - e2);
			u[r] = (int8_t)(r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2);
		}
		/* scores: runs of 16 from st0 (positions past the end of the arrays are never looked at again) */
		if (!(flag & F_GENERIC_SC)) {
			for (t = st0; t <= en0; t += 16) {
				int k;
				for (k = 0; k < 16 && t + k < tl16; ++k) {
					const uint8_t a = sf[t + k], b = qrr[t + k];
					s[t + k] = (a == (uint8_t)(m - 1) || b == (uint8_t)(m - 1)) ? sc_N : a == b ? sc_mch : sc_mis;
				}
			}
		} else {
			for (t = st0; t <= en0; ++t) s[t] = mat[sf[t] * m + qrr[t]];
		}
		if (with_cigar) off[r] = st, off_end[r] = en;
		/* cells, highest first: cell t reads what diagonal r - 1 left at t - 1 and at t */
		for (t = en; t >= st; --t) {
			int8_t z = s[t];
			const int8_t xt1 = t > st ? x[t - 1] : x1, vt1 = t > st ? v[t - 1] : v1, x2t1 = t > st ? x2[t - 1] : x21, ut = u[t];
			int8_t a = (int8_t)(xt1 + vt1), b = (int8_t)(y[t] + ut), a2 = (int8_t)(x2t1 + vt1), b2 = (int8_t)(y2[t] + ut), tmp;
			uint8_t d = 0;
			if (!(flag & F_RIGHT) || !with_cigar) {
				d = a > z ? 1 : 0;  z = z > a ? z : a;
				d = b > z ? 2 : d;  z = z > b ? z : b;
				d = a2 > z ? 3 : d; z = z > a2 ? z : a2;
				d = b2 > z ? 4 : d; z = z > b2 ? z : b2;
			} else {
				d = z > a ? 0 : 1;  z = z > a ? z : a;
				d = z > b ? d : 2;  z = z > b ? z : b;
				d = z > a2 ? d : 3; z = z > a2 ? z : a2;
				d = z > b2 ? d : 4; z = z > b2 ? z : b2;
			}
			z = z < sc_mch ? z : sc_mch;
			u[t] = (int8_t)(z - vt1), v[t] = (int8_t)(z - ut);
			tmp = (int8_t)(z - q);  a = (int8_t)(a - tmp), b = (int8_t)(b - tmp);
			tmp = (int8_t)(z - q2); a2 = (int8_t)(a2 - tmp), b2 = (int8_t)(b2 - tmp);
			if (!(flag & F_RIGHT) || !with_cigar) {
				x[t] = (int8_t)((a > 0 ? a : 0) - qe);    if (a > 0) d |= 0x08;
				y[t] = (int8_t)((b > 0 ? b : 0) - qe);    if (b > 0) d |= 0x10;
				x2[t] = (int8_t)((a2 > 0 ? a2 : 0) - qe2); if (a2 > 0) d |= 0x20;
				y2[t] = (int8_t)((b2 > 0 ? b2 : 0) - qe2); if (b2 > 0) d |= 0x40;
			} else {
				x[t] = (int8_t)((0 > a ? 0 : a) - qe);    if (!(0 > a)) d |= 0x08;
				y[t] = (int8_t)((0 > b ? 0 : b) - qe);    if (!(0 > b)) d |= 0x10;
				x2[t] = (int8_t)((0 > a2 ? 0 : a2) - qe2); if (!(0 > a2)) d |= 0x20;
				y2[t] = (int8_t)((0 > b2 ? 0 : b2) - qe2); if (!(0 > b2)) d |= 0x40;
			}
			if (with_cigar) p[(size_t)r * n_col + (t - st)] = d;
		}
		if (!approx_max) {
			int32_t max_H, max_t;
			if (r > 0) {
				const int en1 = st0 + (en0 - st0) / 4 * 4;
				int32_t HH[4], tt[4];
				int i;
				max_H = H[en0] = en0 > 0 ? H[en0 - 1] + u[en0] : H[en0] + v[en0];
				max_t = en0;
				for (i = 0; i < 4; ++i) HH[i] = max_H, tt[i] = max_t;
				for (t = st0; t < en1; t += 4)
					for (i = 0; i < 4; ++i) {
						H[t + i] += v[t + i];
						if (H[t + i] > HH[i]) HH[i] = H[t + i], tt[i] = t;
					}
				for (i = 0; i < 4; ++i)
					if (max_H < HH[i]) max_H = HH[i], max_t = tt[i] + i;
				for (t = en1; t < en0; ++t) {
					H[t] += (int32_t)v[t];
					if (H[t] > max_H) max_H = H[t], max_t = t;
				}
			} else H[0] = v[0] - qe_first, max_H = H[0], max_t = 0;
			if (en0 == tlen - 1 && H[en0] > ez->mte) ez->mte = H[en0], ez->mte_q = r - en;
			if (r - st0 == qlen - 1 && H[st0] > ez->mqe) ez->mqe = H[st0], ez->mqe_t = st0;
			if (zdrop_test(ez, max_H, r, max_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H[tlen - 1];
		} else {
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					const int32_t d0 = v[last_H0_t], d1 = u[last_H0_t + 1];
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) {
					H0 += v[last_H0_t];
				} else {
					++last_H0_t, H0 += u[last_H0_t];
				}
			} else H0 = v[0] - qe_first, last_H0_t = 0;
			if ((flag & F_APPROX_DROP) && zdrop_test(ez, H0, r, last_H0_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H0;
		}
		last_st = st, last_en = en;
	}
	if (with_cigar) {
		const int rev = !!(flag & F_REV_CIGAR);
		if (!ez->zdropped && !(flag & F_EXTZ_ONLY))
			ez->n_cigar = backtrack(rev, p, off, off_end, n_col, tlen - 1, qlen - 1, cigar, cigar_cap);
		else if (!ez->zdropped && (flag & F_EXTZ_ONLY) && ez->mqe + end_bonus > ez->max) {
			ez->reach_end = 1;
			ez->n_cigar = backtrack(rev, p, off, off_end, n_col, ez->mqe_t, qlen - 1, cigar, cigar_cap);
		} else if (ez->max_t >= 0 && ez->max_q >= 0)
			ez->n_cigar = backtrack(rev, p, off, off_end, n_col, ez->max_t, ez->max_q, cigar, cigar_cap);
	}
	free(u); free(sf); free(qr); free(H); free(p); free(off);
	return 0;
}

/* ksw_ll_i16 with the query profile of ksw_ll_qinit (minimap2/ksw2_ll_sse.c:32-83 with size 2, :85-156): the local-alignment score
 * behind the inversion test of minimap2's -c path (mm_test_zdrop, minimap2/align.c:71-87) and behind mm_align1_inv (:790-845).
 * Farrar's striped Smith-Waterman on eight 16-bit lanes, restated lane by lane in scalar C: stripe k holds the query positions
 * j + k * slen; the values are those of the striped schedule (E is opened from H as it stands before the lazy-F pass; the zero-score
 * padding columns behind the query take part in the maximum; of equal maxima the last row and the last cell in memory order win).
 * Pinned against the compiled reference function (oracle/_ref/libksw2llref.so) by tests/test_oracle_ksw2.py. */
static int ll_sat16(int v) { return v > 32767 ? 32767 : v < -32768 ? -32768 : v; }
static int ll_subs(int a, int b) { return a > b ? a - b : 0; } /* _mm_subs_epu16 on non-negative values */
int nd_oracle_ksw_ll_i16(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat /* 5 x 5 */, int gapo,
                         int gape, int *qe, int *te)
{
	const int slen = (qlen + 7) / 8, gapoe = gapo + gape;
	int i, j, k, gmax = 0;
	int16_t *H0, *H1, *E, *Hmax, *tmp;
	*qe = *te = -1;
	H0 = (int16_t*)calloc((size_t)slen * 8 * 4 + 8, 2);
	H1 = H0 + (size_t)slen * 8, E = H1 + (size_t)slen * 8, Hmax = E + (size_t)slen * 8;
	for (i = 0; i < tlen; ++i) {
		const int8_t *ma = mat + target[i] * 5;
		int h[8], f[8], mx[8], done = 0, imax = 0;
		for (k = 0; k < 8; ++k) f[k] = 0, mx[k] = 0;
		for (k = 7; k >= 1; --k) h[k] = slen > 0 ? H0[(slen - 1) * 8 + k - 1] : 0; /* _mm_slli_si128(h, 2) */
		h[0] = 0;
		for (j = 0; j < slen; ++j)
			for (k = 0; k < 8; ++k) {
				const int pos = j + k * slen, sc = pos < qlen ? ma[query[pos]] : 0;
				int hh = ll_sat16(h[k] + sc), e = E[j * 8 + k];
				hh = hh > e ? hh : e;
				hh = hh > f[k] ? hh : f[k];
				mx[k] = mx[k] > hh ? mx[k] : hh;
				H1[j * 8 + k] = (int16_t)hh;
				hh = ll_subs(hh, gapoe);
				e = ll_subs(e, gape);
				e = e > hh ? e : hh;
				E[j * 8 + k] = (int16_t)e;
				f[k] = ll_subs(f[k], gape);
				f[k] = f[k] > hh ? f[k] : hh;
				h[k] = H0[j * 8 + k];
			}
		for (k = 0; k < 8 && !done; ++k) { /* the lazy-F pass */
			int l, any;
			for (l = 7; l >= 1; --l) f[l] = f[l - 1];
			f[0] = 0;
			for (j = 0; j < slen; ++j) {
				any = 0;
				for (l = 0; l < 8; ++l) {
					int hh = H1[j * 8 + l];
					hh = hh > f[l] ? hh : f[l];
					H1[j * 8 + l] = (int16_t)hh;
					hh = ll_subs(hh, gapoe);
					f[l] = ll_subs(f[l], gape);
					if (f[l] > hh) any = 1;
				}
				if (!any) { done = 1; break; }
			}
		}
		for (k = 0; k < 8; ++k) imax = imax > mx[k] ? imax : mx[k];
		if (imax >= gmax) {
			gmax = imax, *te = i;
			memcpy(Hmax, H1, (size_t)slen * 8 * 2);
		}
		tmp = H1, H1 = H0, H0 = tmp;
	}
	for (i = 0; i < slen * 8; ++i)
		if ((int)(uint16_t)Hmax[i] == gmax) *qe = i / 8 + i % 8 * slen;
	free(H0 < H1 ? (H0 < Hmax ? H0 : Hmax) : (H1 < Hmax ? H1 : Hmax));
	return gmax;
}
