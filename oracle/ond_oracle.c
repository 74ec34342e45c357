/* oracle/ond_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see nd_oracle.h).
 *
 * CPU restatement of the banded greedy O(ND) global aligner that NextDenovo's
 * consensus is built on: `core()` reached through `align` / `align_hq`
 * (reference lib/align.c:428-578) and of `get_align_shift`
 * (lib/nextcorrect.c:102-154).  Written from the algorithm description in
 * SURVEY.md Appendix B; structure, storage and names are ours:
 *   - furthest-reaching x per diagonal lives in `fr[]` (reference: V[k + max_d]);
 *   - the move table is one byte row per edit step, each row covering only the
 *     live band [lo, hi] (reference: triangular D[d][|k|]).
 * Pinned against the compiled reference by tests/test_oracle_vs_ref.py.
 */
#include "nd_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef struct {
    int lo;            /* first diagonal evaluated in this step */
    int n;             /* number of same-parity diagonals evaluated */
    uint8_t *from_left;/* 1: reached from diagonal k-1 (consumes a query base) */
} step_row;

static void reverse_bytes(char *s, int n)
{
    int a = 0, b = n - 1;
    while (a < b) {
        char c = s[a];
        s[a++] = s[b];
        s[b--] = c;
    }
}

/* lib/align.c:563-578: edit budget and band cap as functions of q_len + t_len. */
static void limits(int total, int hq, int *max_d, int *band)
{
    if (hq) {
        *max_d = (int)((total > 1000 ? 0.1 : 0.5) * total);
        *band = (int)((total > 1000 ? 0.03 : 0.3) * total);
    } else {
        *max_d = (int)(0.4 * total);
        *band = (int)((total > 5000 ? 0.1 : 1) * total);
    }
}

void nd_oracle_align(const char *q, int q_len, const char *t, int t_len, int hq,
                     nd_oracle_aln *res, char *t_str, char *q_str, uint8_t *ops)
{
    int max_d, band;
    limits(q_len + t_len, hq, &max_d, &band);
    memset(res, 0, sizeof(*res));
    res->d_final = -1;
    if (t_str) t_str[0] = 0;
    if (q_str) q_str[0] = 0;

    /* fr[k + off]: one slack slot each side because step d reads k-1 / k+1 */
    const int off = max_d + 2;
    int *fr = (int *)calloc((size_t)(2 * off + 1), sizeof(int));
    step_row *rows = (step_row *)calloc((size_t)(max_d > 0 ? max_d : 1), sizeof(step_row));
    int lo = 0, hi = 0, reach = -1, d, k;
    int fin_k = 0, fin_x = 0, fin_y = 0, finished = 0;

    /* ---- forward sweep (lib/align.c:440-489) ---- */
    for (d = 0; d < max_d && hi - lo <= band; d++) {
        const int n = hi >= lo ? (hi - lo) / 2 + 1 : 0;
        step_row *row = &rows[d];
        row->lo = lo;
        row->n = n;
        row->from_left = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
        res->d_steps++;
        if (hi - lo > res->max_band) res->max_band = hi - lo;
        for (k = lo; k <= hi; k += 2) {
            int x, y, left;
            /* lib/align.c:443: take the k+1 neighbour on the lower band edge, or when it is
             * strictly further than the k-1 neighbour (and we are not on the upper edge) */
            if (k == lo || (k != hi && fr[k - 1 + off] < fr[k + 1 + off])) {
                x = fr[k + 1 + off];
                left = 0;
            } else {
                x = fr[k - 1 + off] + 1;
                left = 1;
            }
            row->from_left[(k - lo) / 2] = (uint8_t)left;
            y = x - k;
            while (x < q_len && y < t_len && q[x] == t[y]) {
                x++;
                y++;
            }
            fr[k + off] = x;
            res->cells++;
            if (x + y > reach) reach = x + y;
            if (x >= q_len && y >= t_len) { /* global end; the smallest such k wins (:467-470) */
                finished = 1;
                fin_k = k;
                fin_x = x;
                fin_y = y;
                break;
            }
        }
        if (finished) break;
        /* band re-centring (lib/align.c:473-489): keep diagonals within 150 of the best
         * anti-diagonal, scanning inward from both edges, then widen by one each side */
        {
            int nlo = hi, nhi = lo, kk;
            for (kk = lo; kk < nlo; kk += 2)
                if (fr[kk + off] * 2 - kk >= reach - 150) nlo = kk;
            for (kk = hi; kk > nhi; kk -= 2)
                if (fr[kk + off] * 2 - kk >= reach - 150) nhi = kk;
            hi = nhi + 1;
            lo = nlo - 1;
        }
    }

    if (finished) {
        /* ---- traceback (lib/align.c:491-558), columns collected back to front ---- */
        const int cap = q_len + t_len + 1;
        char *tr = (char *)malloc((size_t)cap), *qr = (char *)malloc((size_t)cap);
        uint8_t *kr = (uint8_t *)malloc((size_t)cap);
        int n = 0, gap = 0, x = fin_x - 1, dd = d, aborted = 0;
        k = fin_k;
        res->d_final = d;
        res->k_final = fin_k;
        res->t_used = fin_y;
        res->q_used = fin_x;
        for (;;) {
            int nk, nx;
            while (x >= 0 && x >= k && q[x] == t[x - k]) {
                tr[n] = qr[n] = q[x];
                kr[n++] = 0;
                x--;
                gap = 0;
            }
            if (x < 0 && x - k < 0) break;
            if (x < k || (x >= 0 && rows[dd].from_left[(k - rows[dd].lo) / 2])) {
                nk = k - 1; /* query base against a gap */
                nx = x - 1;
                if (x < 0) gap = 260;
                else {
                    qr[n] = q[x];
                    tr[n] = '-';
                    kr[n++] = 1;
                }
            } else {
                nk = k + 1; /* target base against a gap */
                nx = x;
                if (x - k < 0) gap = 260;
                else {
                    qr[n] = '-';
                    tr[n] = t[x - k];
                    kr[n++] = 2;
                }
            }
            if (gap++ > 250) { /* runs of more than 250 gap columns abort (:542-545) */
                aborted = 1;
                break;
            }
            dd--;
            k = nk;
            x = nx;
        }
        if (aborted) {
            res->status = 2;
            res->aln_len = 2;
            n = 2;
        } else {
            res->status = 1;
            res->aln_len = n;
        }
        reverse_bytes(tr, n);
        reverse_bytes(qr, n);
        reverse_bytes((char *)kr, n);
        if (t_str) { memcpy(t_str, tr, (size_t)n); t_str[n] = 0; }
        if (q_str) { memcpy(q_str, qr, (size_t)n); q_str[n] = 0; }
        if (ops && !aborted) memcpy(ops, kr, (size_t)n);
        free(tr);
        free(qr);
        free(kr);
    }
    for (k = 0; k < max_d; k++) free(rows[k].from_left);
    free(rows);
    free(fr);
}

int nd_oracle_shift(const uint8_t *ops, int n, int k, unsigned *aln_t_s, unsigned *aln_t_e, int *shift)
{
    int i, run = 0, tcols = 0, first = -1;
    for (i = 0; i < n; i++) {
        run = ops[i] == 0 ? run + 1 : 0;
        if (ops[i] != 1) tcols++;
        if (run == k) { first = i - k + 1; break; }
    }
    if (first < 0) { *shift = 0; return 0; }
    *aln_t_s += (unsigned)(tcols - k);
    run = 0;
    tcols = 0;
    for (i = n - 1; i >= 0; i--) {
        run = ops[i] == 0 ? run + 1 : 0;
        if (ops[i] != 1) tcols++;
        if (run == k) break;
    }
    *aln_t_e = *aln_t_e - (unsigned)tcols + (unsigned)k;
    *shift = first;
    return i + k - first;
}
