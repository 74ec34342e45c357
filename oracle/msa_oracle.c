/* oracle/msa_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see nd_oracle.h).
 *
 * CPU restatement of the consensus MSA of NextDenovo's nextCorrect():
 *   update_msa                 reference lib/nextcorrect.c:212-250
 *   scoring DP + global pick   reference lib/nextcorrect.c:2149-2202
 *   best_pp backtrack walk     reference lib/nextcorrect.c:1907-1982 (cell order only)
 * Storage is ours: one growable entry array per (column, delta, symbol) cell,
 * cells addressed through a per-column offset table.
 */
#include "nd_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef struct {
    nd_oracle_tag pp, ppp;
    int64_t score;
    uint16_t links;
} entry;

typedef struct {
    entry *e;
    int n, cap;
    int64_t best;
    nd_oracle_tag best_pp;
    uint16_t best_links;
} cell;

static int same(const nd_oracle_tag *a, const nd_oracle_tag *b)
{
    return a->t_pos == b->t_pos && a->delta == b->delta && a->base == b->base;
}

long nd_oracle_msa_path(const nd_oracle_tag *const *tags, const uint32_t *len, int n_reads,
                        int n_cols, int factor, nd_oracle_path *path, long cap)
{
    static const nd_oracle_tag HEAD = {-1, 0, 0};
    uint16_t *width = (uint16_t *)calloc((size_t)n_cols, sizeof(uint16_t));
    uint16_t *cover = (uint16_t *)calloc((size_t)n_cols, sizeof(uint16_t));
    uint32_t *first = (uint32_t *)calloc((size_t)n_cols + 1, sizeof(uint32_t));
    int r, p;
    uint32_t i;

    /* column depth / coverage as get_align_tags accumulates them (nextcorrect.c:1512-1517) */
    for (r = 0; r < n_reads; r++)
        for (i = 0; i < len[r]; i++) {
            const nd_oracle_tag *g = &tags[r][i];
            if (g->delta == 0 && g->base != 6) cover[g->t_pos]++;
            if (g->delta >= width[g->t_pos]) width[g->t_pos] = (uint16_t)(g->delta + 1);
        }
    for (p = 0; p < n_cols; p++) first[p + 1] = first[p] + (uint32_t)width[p] * 6u;
    cell *cells = (cell *)calloc(first[n_cols] ? first[n_cols] : 1, sizeof(cell));
#define CELL(t, d, b) (&cells[first[(t)] + (uint32_t)(d) * 6u + (b)])

    /* link counting, first-seen order per cell (nextcorrect.c:215-245) */
    for (r = 0; r < n_reads; r++)
        for (i = 0; i < len[r]; i++) {
            const nd_oracle_tag *cur = &tags[r][i];
            const nd_oracle_tag *pp = i > 0 ? &tags[r][i - 1] : &HEAD;
            const nd_oracle_tag *ppp = i > 1 ? &tags[r][i - 2] : &HEAD;
            int m, hit = 0;
            cell *c;
            if (cur->base == 6 || pp->base == 6) continue;
            c = CELL(cur->t_pos, cur->delta, cur->base);
            for (m = 0; m < c->n; m++)
                if (same(&c->e[m].pp, pp) && same(&c->e[m].ppp, ppp)) {
                    c->e[m].links++;
                    hit = 1;
                    break;
                }
            if (!hit) {
                if (c->n == c->cap) {
                    c->cap = c->cap ? c->cap * 2 : 4;
                    c->e = (entry *)realloc(c->e, sizeof(entry) * (size_t)c->cap);
                }
                c->e[c->n].pp = *pp;
                c->e[c->n].ppp = *ppp;
                c->e[c->n].links = 1;
                c->e[c->n].score = 0;
                c->n++;
            }
        }

    /* scoring DP (nextcorrect.c:2149-2202) */
    int64_t gbest = -10;
    nd_oracle_tag origin = {-1, 0, 0};
    for (p = 0; p < n_cols; p++) {
        int d, b, m, n;
        for (d = 0; d < width[p]; d++)
            for (b = 0; b < 5; b++) {
                cell *c = CELL(p, d, b);
                int64_t via = INT64_MIN, via_next = INT64_MIN;
                c->best = -10;
                c->best_pp.t_pos = -1;
                for (m = 0; m < c->n; m++) {
                    entry *em = &c->e[m];
                    if (em->pp.t_pos == -1) {
                        em->score = 10 * (int64_t)em->links - (int64_t)factor * cover[p];
                    } else {
                        cell *pc = CELL(em->pp.t_pos, em->pp.delta, em->pp.base);
                        for (n = 0; n < pc->n; n++) {
                            entry *en = &pc->e[n];
                            int64_t s;
                            if (!same(&en->pp, &em->ppp)) continue;
                            s = en->score + 10 * (int64_t)em->links - (int64_t)factor * cover[p];
                            if (s > em->score) {
                                em->score = s;
                                via_next = en->score;
                            }
                            if (en->score > via && (em->pp.base == 4 || em->pp.base == b)) {
                                via = en->score;
                                c->best = em->score;
                                c->best_pp = em->pp;
                                c->best_links = em->links;
                            }
                        }
                    }
                    if (em->score > c->best || (em->score == c->best && em->pp.base != 4)) {
                        via = via_next;
                        c->best = em->score;
                        c->best_pp = em->pp;
                        c->best_links = em->links;
                    }
                }
                if (c->best >= gbest - 3000) {
                    origin.t_pos = p;
                    origin.delta = (uint16_t)d;
                    origin.base = (uint8_t)b;
                    if (c->best > gbest) gbest = c->best;
                }
            }
    }

    /* walk best_pp from the origin */
    long np = 0;
    if (origin.t_pos >= 0) {
        nd_oracle_tag cur = origin;
        for (;;) {
            cell *c = CELL(cur.t_pos, cur.delta, cur.base);
            if (np >= cap) { np = -1; break; }
            path[np].t_pos = cur.t_pos;
            path[np].delta = cur.delta;
            path[np].base = cur.base;
            path[np].link_count = c->best_links;
            path[np].coverage = cover[cur.t_pos];
            np++;
            cur = c->best_pp;
            if (cur.t_pos == -1) break;
        }
    }
    for (i = 0; i < first[n_cols]; i++) free(cells[i].e);
    free(cells);
    free(first);
    free(cover);
    free(width);
    return np;
#undef CELL
}
