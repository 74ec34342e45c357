/* oracle/ond_ext_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see nd_oracle.h).
 *
 * CPU restatement of the prefix / extension members of NextDenovo's greedy O(ND) family, the functions the HiFi
 * `--mode 3` overlap path is built on (callers minimap2/map.c:385-482, 919-987):
 *   nd_oracle_ide      lib/align.c:80-141    `ide`: edit steps until either sequence is exhausted -> (matches, block length)
 *   nd_oracle_alnpos   lib/align.c:146-253   `alnpos`: the same forward pass + traceback -> column / match counts and start coordinates
 *   nd_oracle_extend   lib/align.c:256-340 (`extend_fwd`), :343-426 (`extend_rev`): the forward pass with a running
 *                      score (x + y) * d_factor - d; the coordinates of its peak are the extension, 30 below the peak ends it
 * All four share the forward recurrence and the band re-centring of `core()` (lib/align.c:428-490); storage is ours: the
 * furthest-reaching x per diagonal in fr[], one row of move bits per edit step covering the live band only.
 * Pinned against the compiled reference (oracle/_ref/nextcorrect.so exports ide / alnpos / extend_fwd / extend_rev) by
 * tests/test_oracle.py::test_prefix_and_extension_variants_vs_reference.
 */
#include "nd_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef struct {
	int lo, n;
	uint8_t *from_left;
} ext_row;

enum { M_IDE, M_ALNPOS, M_EXT };

typedef struct {
	const char *q, *t;
	int ql, tl, rev;
} seqs_t;

static int same_base(const seqs_t *s, int x, int y)
{
	return s->rev ? s->q[s->ql - x - 1] == s->t[s->tl - y - 1] : s->q[x] == s->t[y];
}

/* returns 1 when an end condition fired (outputs written), 0 when the budget / band ran out */
static int forward(int mode, const seqs_t *s, int max_d, int band, float d_factor, int *o1, int *o2, unsigned *pos)
{
	const int off = max_d + 2;
	int *fr = (int*)calloc((size_t)(2 * off + 2), sizeof(int));
	ext_row *rows = mode == M_ALNPOS ? (ext_row*)calloc((size_t)(max_d > 0 ? max_d : 1), sizeof(ext_row)) : NULL;
	int lo = 0, hi = 0, reach = -1, d, k, done = 0, x = 0, y = 0, fin_k = 0;
	float peak = 0;
	for (d = 0; d < max_d && hi - lo <= band && !done; ++d) {
		if (rows) {
			rows[d].lo = lo, rows[d].n = (hi - lo) / 2 + 1;
			rows[d].from_left = (uint8_t*)calloc((size_t)rows[d].n, 1);
		}
		for (k = lo; k <= hi; k += 2) {
			int left;
			if (k == lo || (k != hi && fr[k - 1 + off] < fr[k + 1 + off])) x = fr[k + 1 + off], left = 0;
			else x = fr[k - 1 + off] + 1, left = 1;
			if (rows) rows[d].from_left[(k - lo) / 2] = (uint8_t)left;
			y = x - k;
			while (x < s->ql && y < s->tl && same_base(s, x, y)) ++x, ++y;
			fr[k + off] = x;
			if (x + y > reach) {
				reach = x + y;
				if (mode == M_EXT) {
					const float score = (x + y) * d_factor - d;
					if (score > peak) peak = score, *o1 = x, *o2 = y;
					else if (score < peak - 30) { done = 2; break; }
				}
			}
			if (x >= s->ql || y >= s->tl) {
				if (mode == M_IDE) *o1 = x - (k + d) / 2, *o2 = y + (k + d) / 2;
				else if (mode == M_EXT) {
					const float score = (x + y) * d_factor - d;
					if (score > 0) *o1 = x, *o2 = y;
				}
				done = 1, fin_k = k;
				break;
			}
		}
		if (done) break;
		{ /* band re-centring, lib/align.c:473-489 (the same in every member of the family) */
			int nlo = hi, nhi = lo, k2;
			for (k2 = lo; k2 < nlo; k2 += 2) if (fr[k2 + off] * 2 - k2 >= reach - 150) nlo = k2;
			for (k2 = hi; k2 > nhi; k2 -= 2) if (fr[k2 + off] * 2 - k2 >= reach - 150) nhi = k2;
			hi = nhi + 1, lo = nlo - 1;
		}
	}
	if (mode == M_ALNPOS && done == 1) {
		int cols = 0, gaps = 0;
		const unsigned q_e = (unsigned)x, t_e = (unsigned)y;
		k = fin_k;
		--x;
		for (;;) {
			while (x >= 0 && x >= k && s->q[x] == s->t[x - k]) --x, ++cols;
			if (x < 0 || x - k < 0) break;
			if (x < k || rows[d].from_left[(k - rows[d].lo) / 2]) --k, --x;
			else ++k;
			++cols, ++gaps, --d;
		}
		pos[0] = (unsigned)cols, pos[1] = (unsigned)(cols - gaps), pos[2] = (unsigned)(x + 1 - k), pos[3] = t_e, pos[4] = (unsigned)(x + 1), pos[5] = q_e;
	}
	if (rows) {
		int i;
		for (i = 0; i < max_d; ++i) free(rows[i].from_left);
		free(rows);
	}
	free(fr);
	return done == 1;
}

/* *mlen / *blen are left untouched when the budget or the band runs out first (as the reference leaves them) */
void nd_oracle_ide(const char *q, int q_len, const char *t, int t_len, int max_d, int band, int *mlen, int *blen)
{
	seqs_t s = { q, t, q_len, t_len, 0 };
	forward(M_IDE, &s, max_d, band, 0.f, mlen, blen, NULL);
}

/* pos[6] = aln_len, aln_mlen, aln_t_s, aln_t_e, aln_q_s, aln_q_e (the fields of `alignpos`, lib/align.h:36-43); untouched when
 * no end was reached */
void nd_oracle_alnpos(const char *q, int q_len, const char *t, int t_len, int max_d, int band, unsigned pos[6])
{
	seqs_t s = { q, t, q_len, t_len, 0 };
	int a = 0, b = 0;
	forward(M_ALNPOS, &s, max_d, band, 0.f, &a, &b, pos);
}

/* rev = 0: extend_fwd (5' -> 3' from the starts of q and t); rev = 1: extend_rev (3' -> 5' from their ends) */
void nd_oracle_extend(const char *q, int q_len, const char *t, int t_len, int max_d, int band, float d_factor, int rev, int *bstx, int *bsty)
{
	seqs_t s = { q, t, q_len, t_len, rev };
	*bstx = *bsty = 0;
	forward(M_EXT, &s, max_d, band, d_factor, bstx, bsty, NULL);
}
