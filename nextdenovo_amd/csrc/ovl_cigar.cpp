// ovl_cigar.cpp -- `minimap2-nd --step 1 -c`: base-level alignment through the chains, then the step-1 writer.
//
//   replaces   mm_align_skeleton      minimap2/align.c:857-913   (called from align_regs, minimap2/map.c:484-503)
//              mm_align1              minimap2/align.c:565-788   (left extension, gap filling between anchors with the
//                                                                 approximate-then-exact z-drop passes, right extension)
//              mm_align1_inv          minimap2/align.c:790-845   (the inversion between two pieces of a z-dropped chain)
//              mm_test_zdrop, mm_fix_cigar, mm_update_extra, mm_append_cigar, the seed filters  align.c:47-166,240-311,341-493
//              mm_split_reg, mm_filter_regs, mm_hit_sort           minimap2/hit.c:90-107,257-276,169-201
//              the step-1 writer's filter                           minimap2/map.c:1297-1304
//
// The reference aligns one piece after the other on one thread per read.  Here every dynamic-programming problem of a round
// goes to the device in one batch (csrc/ksw2_kernels.hip: one wavefront per problem): the gaps between the anchors of a chain
// do not depend on each other's results, only the z-drop of one cuts the chain short, so a chain's left extension, all its gap
// fills (first pass) and its right extension are one batch; the gaps whose first-pass alignment shows a z-drop are aligned again
// (second pass) in a second batch, after the local-alignment scores of the inversion test (a third kind of problem, ksw_ll_i16)
// have come back.  The bookkeeping between the batches -- which anchors bound a gap, how CIGARs are joined and trimmed, where a
// z-dropped chain is split -- is the reference's, per chain, on host threads.  The pieces a split leaves behind are the next
// round's chains; inversions between two pieces follow in a round of their own.
//
// Built for the presets nextDenovo uses with raw reads (ava-ont, ava-pb: every chain is kept, MM_F_ALL_CHAINS, no long joins, no
// splicing, no short-read mode).  The compiled reference aborts on `-x ava-hifi -c` (k = 51), so there is nothing to match there.
// There is no CPU path: the alignments themselves run on the device or the call fails.
#include <chrono>
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ndgpu_overlap.h"

// wall time of the device batches of one ndgpu_ovl_map_cigar call (ndgpu_ovl_cigar_stats: the split the -c timing reports)
static std::atomic<uint64_t> g_t_ksw_ns{0}, g_t_ll_ns{0};


namespace {

constexpr uint64_t kSeedLongJoin = 1ULL << 40, kSeedIgnore = 1ULL << 41, kSeedTandem = 1ULL << 42, kSeedSelf = 1ULL << 43;  // mmpriv.h:18-21
constexpr int kNegInf = -0x40000000;
constexpr int kParentUnset = -1, kParentTmpPri = -2;  // mmpriv.h:10-11
enum { EZ_RIGHT = 0x02, EZ_APPROX_MAX = 0x08, EZ_EXTZ_ONLY = 0x40, EZ_REV_CIGAR = 0x80 };  // ksw2.h:9-16

template <class F> void par_for(size_t n, int threads, F f) {
    if (n == 0) return;
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, threads), n));
    if (nt == 1) {
        for (size_t i = 0; i < n; i++) f(i);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    std::exception_ptr failed;  // the first exception of a worker, rethrown in the caller
    std::mutex mu;
    for (int t = 0; t < nt; t++)
        th.emplace_back([&] {
            try {
                for (;;) {
                    const size_t i = next.fetch_add(1);
                    if (i >= n) break;
                    f(i);
                }
            } catch (...) {
                std::lock_guard<std::mutex> g(mu);
                if (!failed) failed = std::current_exception();
                next = n;
            }
        });
    for (auto &t : th) t.join();
    if (failed) std::rethrow_exception(failed);
}

struct Anchor { uint64_t x, y; };

struct Reg {  // mm_reg1_t (minimap.h:84-99), the fields this path touches
    int32_t id = 0, cnt = 0, rid = 0, score = 0, qs = 0, qe = 0, rs = 0, re = 0, parent = kParentUnset, as = 0, mlen = 0, blen = 0;
    uint32_t hash = 0;
    uint8_t rev = 0, inv = 0, split = 0, split_inv = 0;
    bool has_p = false;  // mm_extra_t: dp_max and the CIGAR
    int32_t dp_score = 0, dp_max = 0;
    std::vector<uint32_t> cigar;
    // driver state
    bool aligned = false;
    std::unique_ptr<Reg> r2;  // the piece a z-drop split off (inserted behind this one by the read's loop)
};

struct Ez {  // ksw_extz_t (ksw2.h:23-32)
    int32_t max = 0, zdropped = 0, max_q = -1, max_t = -1, mqe = kNegInf, mqe_t = -1, mte = kNegInf, mte_q = -1, score = kNegInf, reach_end = 0;
    std::vector<uint32_t> cigar;
};

struct Opt {
    int32_t a, b, q, e, q2, e2, sc_ambi, zdrop, zdrop_inv, end_bonus, min_dp_max, min_ksw_len;
    int64_t max_sw_mat;
    int32_t k, hpc, min_cnt, min_chain_score, bw, max_gap, minlen, dvt, maxhan1, maxhan2;
    int8_t mat[25];
};

struct Targets {  // the index's reads as the caller holds them (.2bit layout: 16 bases a word, the first in the top bits)
    const uint32_t *words;
    const uint64_t *word_off;
    const uint32_t *lens, *ids;
    uint8_t base(uint32_t rid, uint32_t pos) const { return (uint8_t)(words[word_off[rid] + (pos >> 4)] >> (30 - 2 * (pos & 15)) & 3u); }
    void get(uint32_t rid, int32_t st, int32_t en, uint8_t *out) const {  // mm_idx_getseq
        for (int32_t p = st; p < en; p++) out[p - st] = base(rid, (uint32_t)p);
    }
};

struct ReadCtx {
    uint32_t qid = 0;
    int32_t qlen = 0, n_a = 0;
    std::vector<uint8_t> qbuf;     // forward codes, then the reverse complement, back to back as the reference allocates them
    const uint8_t *qseq[2] = {nullptr, nullptr};  // (mm_align_skeleton, align.c:864-870)
    std::vector<Anchor> a;         // the read's chained anchors; the seed filters set flag bits in y
    std::vector<std::unique_ptr<Reg>> regs;
    size_t cur = 0;
    int stage = 0;
    // inversion result of regs[cur] (stage 1)
    bool inv_done = false, inv_ok = false;
    std::unique_ptr<Reg> inv_reg;
};

// ---- small pieces of minimap2/hit.c ----

void cal_fuzzy_len(Reg &r, const Anchor *a) {  // mm_cal_fuzzy_len, hit.c:8-21
    r.mlen = r.blen = 0;
    if (r.cnt <= 0) return;
    r.mlen = r.blen = (int32_t)(a[r.as].y >> 32 & 0xff);
    for (int i = r.as + 1; i < r.as + r.cnt; ++i) {
        const int span = (int)(a[i].y >> 32 & 0xff);
        const int tl = (int32_t)a[i].x - (int32_t)a[i - 1].x, ql = (int32_t)a[i].y - (int32_t)a[i - 1].y;
        r.blen += tl > ql ? tl : ql;
        r.mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
    }
}

void reg_set_coor(Reg &r, int32_t qlen, const Anchor *a) {  // mm_reg_set_coor, hit.c:23-38
    const int32_t k = r.as, q_span = (int32_t)(a[k].y >> 32 & 0xff);
    r.rev = (uint8_t)(a[k].x >> 63);
    r.rid = (int32_t)(a[k].x << 1 >> 33);
    r.rs = (int32_t)a[k].x + 1 > q_span ? (int32_t)a[k].x + 1 - q_span : 0;
    r.re = (int32_t)a[k + r.cnt - 1].x + 1;
    if (!r.rev) r.qs = (int32_t)a[k].y + 1 - q_span, r.qe = (int32_t)a[k + r.cnt - 1].y + 1;
    else r.qs = qlen - ((int32_t)a[k + r.cnt - 1].y + 1), r.qe = qlen - ((int32_t)a[k].y + 1 - q_span);
    cal_fuzzy_len(r, a);
}

void copy_plain(Reg &dst, const Reg &src) {  // `*r2 = *r` without the driver state
    dst.id = src.id, dst.cnt = src.cnt, dst.rid = src.rid, dst.score = src.score, dst.qs = src.qs, dst.qe = src.qe, dst.rs = src.rs, dst.re = src.re;
    dst.parent = src.parent, dst.as = src.as, dst.mlen = src.mlen, dst.blen = src.blen, dst.hash = src.hash, dst.rev = src.rev, dst.inv = src.inv;
    dst.split = src.split, dst.split_inv = src.split_inv;
}

void split_reg(Reg &r, Reg &r2, int n, int qlen, const Anchor *a) {  // mm_split_reg, hit.c:90-107
    if (n <= 0 || n >= r.cnt) return;
    copy_plain(r2, r);
    r2.id = -1;
    r2.has_p = false, r2.cigar.clear(), r2.dp_score = r2.dp_max = 0;
    r2.split_inv = 0;
    r2.cnt = r.cnt - n;
    r2.score = (int32_t)(r.score * ((float)r2.cnt / r.cnt) + .499);
    r2.as = r.as + n;
    if (r.parent == r.id) r2.parent = kParentTmpPri;
    reg_set_coor(r2, qlen, a);
    r.cnt -= r2.cnt;
    r.score -= r2.score;
    reg_set_coor(r, qlen, a);
    r.split |= 1, r2.split |= 2;
}

// radix_sort_128x (ksort.h:101-151 instantiated in misc.c:156): in-place most-significant-digit radix sort on x, insertion sort
// below 65 elements -- not stable, so equal keys come out in ITS order
struct X128 { uint64_t x, y; };
void rs_insertsort(X128 *beg, X128 *end) {
    for (X128 *i = beg + 1; i < end; ++i)
        if (i->x < (i - 1)->x) {
            X128 *j, tmp = *i;
            for (j = i; j > beg && tmp.x < (j - 1)->x; --j) *j = *(j - 1);
            *j = tmp;
        }
}
void rs_sort(X128 *beg, X128 *end, int n_bits, int s) {
    struct Bucket { X128 *b, *e; };
    const int size = 1 << n_bits, m = size - 1;
    Bucket b[256], *be = b + size, *k;
    for (k = b; k != be; ++k) k->b = k->e = beg;
    for (X128 *i = beg; i != end; ++i) ++b[i->x >> s & m].e;
    for (k = b + 1; k != be; ++k) k->e += (k - 1)->e - beg, k->b = (k - 1)->e;
    for (k = b; k != be;) {
        if (k->b != k->e) {
            Bucket *l;
            if ((l = b + (k->b->x >> s & m)) != k) {
                X128 tmp = *k->b, swap;
                do {
                    swap = tmp, tmp = *l->b, *l->b++ = swap;
                    l = b + (tmp.x >> s & m);
                } while (l != k);
                *k->b++ = tmp;
            } else ++k->b;
        } else ++k;
    }
    for (b->b = beg, k = b + 1; k != be; ++k) k->b = (k - 1)->e;
    if (s) {
        s = s > n_bits ? s - n_bits : 0;
        for (k = b; k != be; ++k)
            if (k->e - k->b > 64) rs_sort(k->b, k->e, n_bits, s);
            else if (k->e - k->b > 1) rs_insertsort(k->b, k->e);
    }
}
void radix_sort_128x(X128 *beg, X128 *end) {
    if (end - beg <= 64) rs_insertsort(beg, end);
    else rs_sort(beg, end, 8, 56);
}

// ---- the seed filters and coordinates of minimap2/align.c ----

int hplen_back(const Targets &T, uint32_t rid, uint32_t x) {  // mm_get_hplen_back, align.c:341-348
    const int c = T.base(rid, x);
    int64_t i;
    for (i = (int64_t)x - 1; i >= 0; --i)
        if (T.base(rid, (uint32_t)i) != c) break;
    return (int)((int64_t)x - i);
}

void adjust_minier(const Opt &o, const Targets &T, const ReadCtx &R, const Anchor &a, int32_t *r, int32_t *q) {  // mm_adjust_minier, align.c:350-365
    if (o.hpc) {
        const uint8_t *qseq = R.qseq[a.x >> 63];
        int i, c;
        *q = (int32_t)a.y;
        for (i = *q - 1, c = qseq[*q]; i > 0; --i)
            if (qseq[i] != c) break;
        *q = i + 1;
        c = hplen_back(T, (uint32_t)(a.x << 1 >> 33), (uint32_t)(int32_t)a.x);
        *r = (int32_t)a.x + 1 - c;
    } else {
        *r = (int32_t)a.x - (o.k >> 1);
        *q = (int32_t)a.y - (o.k >> 1);
    }
}

std::vector<int> collect_long_gaps(int as1, int cnt1, const Anchor *a, int min_gap) {  // align.c:367-384
    std::vector<int> K;
    for (int i = 1; i < cnt1; ++i) {
        const int gap = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - (int32_t)a[as1 + i - 1].x);
        if (gap < -min_gap || gap > min_gap) K.push_back(i);
    }
    if (K.size() <= 1) K.clear();
    return K;
}

void filter_bad_seeds(int as1, int cnt1, Anchor *a, int min_gap, int diff_thres, int max_ext_len, int max_ext_cnt) {  // align.c:386-421
    const std::vector<int> K = collect_long_gaps(as1, cnt1, a, min_gap);
    const int n = (int)K.size();
    if (n == 0) return;
    int max = 0, max_st = -1, max_en = -1;
    for (int k = 0;; ++k) {
        int gap, l, n_ins = 0, n_del = 0, qs, rs, max_diff = 0, max_diff_l = -1;
        if (k == n || k >= max_en) {
            if (max_en > 0)
                for (int i = K[max_st]; i < K[max_en]; ++i) a[as1 + i].y |= kSeedIgnore;
            max = 0, max_st = max_en = -1;
            if (k == n) break;
        }
        const int i = K[k];
        gap = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - (int32_t)(a[as1 + i].x - a[as1 + i - 1].x);
        if (gap > 0) n_ins += gap;
        else n_del += -gap;
        qs = (int32_t)a[as1 + i - 1].y;
        rs = (int32_t)a[as1 + i - 1].x;
        for (l = k + 1; l < n && l <= k + max_ext_cnt; ++l) {
            const int j = K[l];
            if ((int32_t)a[as1 + j].y - qs > max_ext_len || (int32_t)a[as1 + j].x - rs > max_ext_len) break;
            gap = ((int32_t)a[as1 + j].y - (int32_t)a[as1 + j - 1].y) - (int32_t)(a[as1 + j].x - a[as1 + j - 1].x);
            if (gap > 0) n_ins += gap;
            else n_del += -gap;
            const int diff = n_ins + n_del - abs(n_ins - n_del);
            if (max_diff < diff) max_diff = diff, max_diff_l = l;
        }
        if (max_diff > diff_thres && max_diff > max) max = max_diff, max_st = k, max_en = max_diff_l;
    }
}

void filter_bad_seeds_alt(int as1, int cnt1, Anchor *a, int min_gap, int max_ext) {  // align.c:423-457
    const std::vector<int> K = collect_long_gaps(as1, cnt1, a, min_gap);
    const int n = (int)K.size();
    for (int k = 0; k < n;) {
        const int i = K[k];
        int l;
        int gap1 = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - (int32_t)a[as1 + i - 1].x);
        int re1 = (int32_t)a[as1 + i].x, qe1 = (int32_t)a[as1 + i].y;
        gap1 = gap1 > 0 ? gap1 : -gap1;
        for (l = k + 1; l < n; ++l) {
            const int j = K[l];
            if ((int32_t)a[as1 + j].y - qe1 > max_ext || (int32_t)a[as1 + j].x - re1 > max_ext) break;
            int gap2 = ((int32_t)a[as1 + j].y - (int32_t)a[as1 + j - 1].y) - (int32_t)(a[as1 + j].x - a[as1 + j - 1].x);
            const int q_span_pre = (int)(a[as1 + j - 1].y >> 32 & 0xff);
            const int rs2 = (int32_t)a[as1 + j - 1].x + q_span_pre, qs2 = (int32_t)a[as1 + j - 1].y + q_span_pre;
            const int m = rs2 - re1 < qs2 - qe1 ? rs2 - re1 : qs2 - qe1;
            gap2 = gap2 > 0 ? gap2 : -gap2;
            if (m > gap1 + gap2) break;
            re1 = (int32_t)a[as1 + j].x, qe1 = (int32_t)a[as1 + j].y;
            gap1 = gap2;
        }
        if (l > k + 1) {
            const int end = K[l - 1];
            for (int j = K[k]; j < end; ++j) a[as1 + j].y |= kSeedIgnore;
            a[as1 + end].y |= kSeedLongJoin;
        }
        k = l;
    }
}

void fix_bad_ends(const Reg &r, const Anchor *a, int bw, int min_match, int32_t *as, int32_t *cnt) {  // mm_fix_bad_ends, align.c:459-493
    *as = r.as, *cnt = r.cnt;
    if (r.cnt < 3) return;
    int32_t i, l, m;
    m = l = (int32_t)(a[r.as].y >> 32 & 0xff);
    for (i = r.as + 1; i < r.as + r.cnt - 1; ++i) {
        const int32_t q_span = (int32_t)(a[i].y >> 32 & 0xff);
        if (a[i].y & kSeedLongJoin) break;
        const int32_t lr = (int32_t)a[i].x - (int32_t)a[i - 1].x, lq = (int32_t)a[i].y - (int32_t)a[i - 1].y;
        const int32_t mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
        if (mx - mn > l >> 1) *as = i;
        l += mn;
        m += mn < q_span ? mn : q_span;
        if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r.mlen >> 1) break;
    }
    *cnt = r.as + r.cnt - *as;
    m = l = (int32_t)(a[r.as + r.cnt - 1].y >> 32 & 0xff);
    for (i = r.as + r.cnt - 2; i > *as; --i) {
        const int32_t q_span = (int32_t)(a[i + 1].y >> 32 & 0xff);
        if (a[i + 1].y & kSeedLongJoin) break;
        const int32_t lr = (int32_t)a[i + 1].x - (int32_t)a[i].x, lq = (int32_t)a[i + 1].y - (int32_t)a[i].y;
        const int32_t mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
        if (mx - mn > l >> 1) *cnt = i + 1 - *as;
        l += mn;
        m += mn < q_span ? mn : q_span;
        if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r.mlen >> 1) break;
    }
}

// ---- CIGAR bookkeeping of minimap2/align.c ----

void append_cigar(Reg &r, const std::vector<uint32_t> &c) {  // mm_append_cigar, align.c:288-311
    if (c.empty()) return;
    r.has_p = true;
    if (!r.cigar.empty() && (r.cigar.back() & 0xf) == (c[0] & 0xf)) {
        r.cigar.back() += c[0] >> 4 << 4;
        r.cigar.insert(r.cigar.end(), c.begin() + 1, c.end());
    } else r.cigar.insert(r.cigar.end(), c.begin(), c.end());
}

void fix_cigar(Reg &r, const uint8_t *qseq, const uint8_t *tseq, int *qshift, int *tshift) {  // mm_fix_cigar, align.c:91-166
    std::vector<uint32_t> &cg = r.cigar;
    int32_t toff = 0, qoff = 0, to_shrink = 0;
    *qshift = *tshift = 0;
    uint32_t n = (uint32_t)cg.size(), k;
    if (n <= 1) return;
    for (k = 0; k < n; ++k) {  // indel left alignment
        const uint32_t op = cg[k] & 0xf, len = cg[k] >> 4;
        if (len == 0) to_shrink = 1;
        if (op == 0) toff += len, qoff += len;
        else if (op == 1 || op == 2) {
            if (k > 0 && k < n - 1 && (cg[k - 1] & 0xf) == 0 && (cg[k + 1] & 0xf) == 0) {
                int l;
                const int prev_len = (int)(cg[k - 1] >> 4);
                if (op == 1) {
                    for (l = 0; l < prev_len; ++l)
                        if (qseq[qoff - 1 - l] != qseq[qoff + len - 1 - l]) break;
                } else {
                    for (l = 0; l < prev_len; ++l)
                        if (tseq[toff - 1 - l] != tseq[toff + len - 1 - l]) break;
                }
                if (l > 0) cg[k - 1] -= (uint32_t)l << 4, cg[k + 1] += (uint32_t)l << 4, qoff -= l, toff -= l;
                if (l == prev_len) to_shrink = 1;
            }
            if (op == 1) qoff += len;
            else toff += len;
        } else if (op == 3) toff += len;
    }
    for (k = 0; k + 2 < n; ++k) {  // CIGARs like 5I6D7I
        if ((cg[k] & 0xf) > 0 && (cg[k] & 0xf) + (cg[k + 1] & 0xf) == 3) {
            uint32_t l, s[3] = {0, 0, 0};
            for (l = k; l < n; ++l) {
                const uint32_t op = cg[l] & 0xf;
                if (op == 1 || op == 2 || cg[l] >> 4 == 0) s[op] += cg[l] >> 4;
                else break;
            }
            if (s[1] > 0 && s[2] > 0 && l - k > 2) {
                cg[k] = s[1] << 4 | 1;
                cg[k + 1] = s[2] << 4 | 2;
                for (k += 2; k < l; ++k) cg[k] &= 0xf;
                to_shrink = 1;
            }
            k = l;
        }
    }
    if (to_shrink) {
        uint32_t l = 0;
        for (k = 0; k < n; ++k)
            if (cg[k] >> 4 != 0) cg[l++] = cg[k];
        n = l;
        for (k = l = 0; k < n; ++k)
            if (k == n - 1 || (cg[k] & 0xf) != (cg[k + 1] & 0xf)) cg[l++] = cg[k];
            else cg[k + 1] += cg[k] >> 4 << 4;
        n = l;
        cg.resize(n);
    }
    if ((cg[0] & 0xf) == 1 || (cg[0] & 0xf) == 2) {  // leading I or D
        const int32_t l = (int32_t)(cg[0] >> 4);
        if ((cg[0] & 0xf) == 1) {
            if (r.rev) r.qe -= l;
            else r.qs += l;
            *qshift = l;
        } else r.rs += l, *tshift = l;
        cg.erase(cg.begin());
    }
}

void update_extra(Reg &r, const uint8_t *qseq, const uint8_t *tseq, const Opt &o) {  // mm_update_extra, align.c:240-286 (no =/X)
    if (!r.has_p) return;
    int qshift, tshift;
    fix_cigar(r, qseq, tseq, &qshift, &tshift);
    qseq += qshift, tseq += tshift;
    int32_t s = 0, max = 0, toff = 0, qoff = 0;
    r.blen = r.mlen = 0;
    for (uint32_t c : r.cigar) {
        const uint32_t op = c & 0xf, len = c >> 4;
        if (op == 0) {
            int n_ambi = 0, n_diff = 0;
            for (uint32_t l = 0; l < len; ++l) {
                const int cq = qseq[qoff + l], ct = tseq[toff + l];
                if (ct > 3 || cq > 3) ++n_ambi;
                else if (ct != cq) ++n_diff;
                s += o.mat[ct * 5 + cq];
                if (s < 0) s = 0;
                else max = max > s ? max : s;
            }
            r.blen += len - n_ambi, r.mlen += len - (n_ambi + n_diff);
            toff += len, qoff += len;
        } else if (op == 1) {
            int n_ambi = 0;
            for (uint32_t l = 0; l < len; ++l)
                if (qseq[qoff + l] > 3) ++n_ambi;
            r.blen += len - n_ambi;
            s -= o.q + o.e * (int32_t)len;
            if (s < 0) s = 0;
            qoff += len;
        } else if (op == 2) {
            int n_ambi = 0;
            for (uint32_t l = 0; l < len; ++l)
                if (tseq[toff + l] > 3) ++n_ambi;
            r.blen += len - n_ambi;
            s -= o.q + o.e * (int32_t)len;
            if (s < 0) s = 0;
            toff += len;
        } else if (op == 3) toff += len;
    }
    r.dp_max = max;
}

// the walk of mm_test_zdrop (align.c:32-69) up to the point where it may ask for a local alignment
struct ZdropWalk { int32_t max_zdrop = 0; int pos[2][2] = {{-1, -1}, {-1, -1}}; };
ZdropWalk zdrop_walk(const Opt &o, const uint8_t *qseq, const uint8_t *tseq, const std::vector<uint32_t> &cigar) {
    ZdropWalk w;
    int32_t score = 0, max = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0;
    auto upd = [&](int32_t sc, int ii, int jj) {  // update_max_zdrop
        if (sc < max) {
            const int li = ii - max_i, lj = jj - max_j, diff = li > lj ? li - lj : lj - li, z = max - sc - diff * o.e;
            if (z > w.max_zdrop) {
                w.max_zdrop = z;
                w.pos[0][0] = max_i, w.pos[0][1] = ii + 1;
                w.pos[1][0] = max_j, w.pos[1][1] = jj + 1;
            }
        } else max = sc, max_i = ii, max_j = jj;
    };
    for (uint32_t c : cigar) {
        const uint32_t op = c & 0xf, len = c >> 4;
        if (op == 0) {
            for (uint32_t l = 0; l < len; ++l) {
                score += o.mat[tseq[i + l] * 5 + qseq[j + l]];
                upd(score, i + (int)l, j + (int)l);
            }
            i += len, j += len;
        } else if (op == 1 || op == 2 || op == 3) {
            score -= o.q + o.e * (int32_t)len;
            if (op == 1) j += len;
            else i += len;
            upd(score, i, j);
        }
    }
    return w;
}

// ---- one chain's alignment (mm_align1) as a task over three device batches ----

struct Seg {  // one gap-filling problem: the stretch between two anchors of the chain
    int i;  // the closing anchor's index in [as1, as1 + cnt1)
    int32_t rs, qs, re, qe, bw1;
    int job1 = -1, job2 = -1, ll = -1, code = 0;
    ZdropWalk walk;
    std::vector<uint8_t> ll_q;
};

struct Task {
    const Targets *tg = nullptr;
    ReadCtx *R = nullptr;
    Reg *reg = nullptr;
    bool skip = false;
    int32_t rid = 0, rev = 0, as1 = 0, cnt1 = 0, bw = 0;
    int32_t rs = 0, qs = 0, re = 0, qe = 0, rs0 = 0, qs0 = 0, re0 = 0, qe0 = 0;
    std::vector<uint8_t> tseq;    // target bases [rs0, re0)
    std::vector<uint8_t> lq, lt;  // the left extension's sequences, reversed
    bool left = false, right = false;
    int left_job = -1, right_job = -1;
    Ez left_local, right_local;
    std::vector<Seg> segs;
};

struct JobList {
    std::vector<ndgpu_ksw_job> jobs;
    int add(const Opt &o, const uint8_t *q, int ql, const uint8_t *t, int tl, int w, int zdrop, int end_bonus, int flag) {
        ndgpu_ksw_job j;
        j.query = q, j.target = t, j.mat = o.mat, j.qlen = ql, j.tlen = tl, j.w = w, j.zdrop = zdrop, j.end_bonus = end_bonus, j.flag = flag;
        j.m = 5, j.gapo = (int8_t)o.q, j.gape = (int8_t)o.e, j.gapo2 = (int8_t)o.q2, j.gape2 = (int8_t)o.e2;
        jobs.push_back(j);
        return (int)jobs.size() - 1;
    }
};

// mm_align_pair's cases that never reach the kernel: a matrix beyond max_sw_mat counts as z-dropped (align.c:323-325); an empty
// sequence leaves the result reset (ksw_extd2_sse returns before anything, ksw2_extd2_sse.c:52-53)
bool local_result(const Opt &o, int ql, int tl, Ez *ez) {
    if (o.max_sw_mat > 0 && (int64_t)tl * ql > o.max_sw_mat) {
        *ez = Ez();
        ez->zdropped = 1;
        return true;
    }
    if (ql <= 0 || tl <= 0) {
        *ez = Ez();
        return true;
    }
    return false;
}

Ez to_ez(const ndgpu_ksw_result &r) {
    Ez e;
    e.max = r.max, e.zdropped = r.zdropped, e.max_q = r.max_q, e.max_t = r.max_t, e.mqe = r.mqe, e.mqe_t = r.mqe_t, e.mte = r.mte, e.mte_q = r.mte_q;
    e.score = r.score, e.reach_end = r.reach_end;
    if (r.n_cigar > 0 && r.cigar) e.cigar.assign(r.cigar, r.cigar + r.n_cigar);
    return e;
}

// phase 0: the chain's geometry (align.c:575-678) and which problems it poses
void plan(Task &T, const Opt &o, const Targets &tg) {
    ReadCtx &R = *T.R;
    Reg &r = *T.reg;
    Anchor *a = R.a.data();
    T.tg = &tg;
    if (r.cnt == 0) {
        T.skip = true;
        return;
    }
    const int32_t qlen = R.qlen, n_a = R.n_a, tlen = (int32_t)tg.lens[(uint32_t)(a[r.as].x << 1 >> 33)];
    T.rid = (int32_t)(a[r.as].x << 1 >> 33), T.rev = (int32_t)(a[r.as].x >> 63);
    T.bw = (int)(o.bw * 1.5 + 1.);
    int32_t as1, cnt1, rs, qs, re, qe, rs0, qs0, re0, qe0, rs1, qs1, re1, qe1, i, l;
    fix_bad_ends(r, a, o.bw, o.min_chain_score * 2, &as1, &cnt1);
    filter_bad_seeds(as1, cnt1, a, 10, 40, o.max_gap >> 1, 10);
    filter_bad_seeds_alt(as1, cnt1, a, 30, o.max_gap >> 1);
    adjust_minier(o, tg, R, a[as1], &rs, &qs);
    adjust_minier(o, tg, R, a[as1 + cnt1 - 1], &re, &qe);
    // where the dynamic programming may start and end (align.c:615-674)
    rs0 = (int32_t)a[r.as].x + 1 - (int32_t)(a[r.as].y >> 32 & 0xff);
    qs0 = (int32_t)a[r.as].y + 1 - (int32_t)(a[r.as].y >> 32 & 0xff);
    if (rs0 < 0) rs0 = 0;
    rs1 = qs1 = 0;
    for (i = r.as - 1, l = 0; i >= 0 && a[i].x >> 32 == a[r.as].x >> 32; --i) {
        const int32_t x = (int32_t)a[i].x + 1 - (int32_t)(a[i].y >> 32 & 0xff), y = (int32_t)a[i].y + 1 - (int32_t)(a[i].y >> 32 & 0xff);
        if (x < rs0 && y < qs0) {
            if (++l > o.min_cnt) {
                l = rs0 - x > qs0 - y ? rs0 - x : qs0 - y;
                rs1 = rs0 - l, qs1 = qs0 - l;
                if (rs1 < 0) rs1 = 0;
                break;
            }
        }
    }
    if (qs > 0 && rs > 0) {
        l = qs < o.max_gap ? qs : o.max_gap;
        qs1 = qs1 > qs - l ? qs1 : qs - l;
        qs0 = qs0 < qs1 ? qs0 : qs1;
        l += l * o.a > o.q ? (l * o.a - o.q) / o.e : 0;
        l = l < o.max_gap ? l : o.max_gap;
        l = l < rs ? l : rs;
        rs1 = rs1 > rs - l ? rs1 : rs - l;
        rs0 = rs0 < rs1 ? rs0 : rs1;
        rs0 = rs0 < rs ? rs0 : rs;
    } else rs0 = rs, qs0 = qs;
    re0 = (int32_t)a[r.as + r.cnt - 1].x + 1;
    qe0 = (int32_t)a[r.as + r.cnt - 1].y + 1;
    re1 = tlen, qe1 = qlen;
    for (i = r.as + r.cnt, l = 0; i < n_a && a[i].x >> 32 == a[r.as].x >> 32; ++i) {
        const int32_t x = (int32_t)a[i].x + 1, y = (int32_t)a[i].y + 1;
        if (x > re0 && y > qe0) {
            if (++l > o.min_cnt) {
                l = x - re0 > y - qe0 ? x - re0 : y - qe0;
                re1 = re0 + l, qe1 = qe0 + l;
                break;
            }
        }
    }
    if (qe < qlen && re < tlen) {
        l = qlen - qe < o.max_gap ? qlen - qe : o.max_gap;
        qe1 = qe1 < qe + l ? qe1 : qe + l;
        qe0 = qe0 > qe1 ? qe0 : qe1;
        l += l * o.a > o.q ? (l * o.a - o.q) / o.e : 0;
        l = l < o.max_gap ? l : o.max_gap;
        l = l < tlen - re ? l : tlen - re;
        re1 = re1 < re + l ? re1 : re + l;
        re0 = re0 > re1 ? re0 : re1;
    } else re0 = re, qe0 = qe;
    if (a[r.as].y & kSeedSelf) {
        int max_ext = r.qs > r.rs ? r.qs - r.rs : r.rs - r.qs;
        if (r.rs - rs0 > max_ext) rs0 = r.rs - max_ext;
        if (r.qs - qs0 > max_ext) qs0 = r.qs - max_ext;
        max_ext = r.qe > r.re ? r.qe - r.re : r.re - r.qe;
        if (re0 - r.re > max_ext) re0 = r.re + max_ext;
        if (qe0 - r.qe > max_ext) qe0 = r.qe + max_ext;
    }
    T.as1 = as1, T.cnt1 = cnt1, T.rs = rs, T.qs = qs, T.rs0 = rs0, T.qs0 = qs0, T.re0 = re0, T.qe0 = qe0;
    if (re0 <= rs0) {  // (the reference asserts re0 > rs0)
        T.skip = true;
        return;
    }
    T.tseq.resize((size_t)(re0 - rs0));
    tg.get((uint32_t)T.rid, rs0, re0, T.tseq.data());
    const uint8_t *q0 = R.qseq[T.rev];
    if (qs > 0 && rs > 0) {  // left extension: both sequences reversed (align.c:684-701)
        T.left = true;
        T.lq.assign(q0 + qs0, q0 + qs);
        T.lt.assign(T.tseq.begin(), T.tseq.begin() + (rs - rs0));
        std::reverse(T.lq.begin(), T.lq.end());
        std::reverse(T.lt.begin(), T.lt.end());
    }
    // gap filling (align.c:703-757): which stretches are aligned follows from the anchors alone
    int32_t crs = rs, cqs = qs;
    for (i = 1; i < cnt1; ++i) {
        if ((a[as1 + i].y & (kSeedIgnore | kSeedTandem)) && i != cnt1 - 1) continue;
        adjust_minier(o, tg, R, a[as1 + i], &re, &qe);
        if (i == cnt1 - 1 || (a[as1 + i].y & kSeedLongJoin) || (qe - cqs >= o.min_ksw_len && re - crs >= o.min_ksw_len)) {
            Seg s;
            s.i = i, s.rs = crs, s.qs = cqs, s.re = re, s.qe = qe, s.bw1 = T.bw;
            if (a[as1 + i].y & kSeedLongJoin) s.bw1 = qe - cqs > re - crs ? qe - cqs : re - crs;
            T.segs.push_back(std::move(s));
            crs = re, cqs = qe;
        }
    }
    T.re = re, T.qe = qe;  // the last anchor's (re, qe from the second adjust_minier call when the loop did not run)
    T.right = qe < qe0 && re < re0;
}

void pose_first(Task &T, const Opt &o, JobList &J) {
    if (T.skip) return;
    const ReadCtx &R = *T.R;
    const uint8_t *q0 = R.qseq[T.rev];
    if (T.left) {
        const int ql = T.qs - T.qs0, tl = T.rs - T.rs0;
        if (!local_result(o, ql, tl, &T.left_local))
            T.left_job = J.add(o, T.lq.data(), ql, T.lt.data(), tl, T.bw, T.reg->split_inv ? o.zdrop_inv : o.zdrop, o.end_bonus,
                               EZ_EXTZ_ONLY | EZ_RIGHT | EZ_REV_CIGAR);
    }
    for (Seg &s : T.segs) {
        Ez tmp;
        if (local_result(o, s.qe - s.qs, s.re - s.rs, &tmp)) continue;  // (finish() recomputes it)
        s.job1 = J.add(o, q0 + s.qs, s.qe - s.qs, T.tseq.data() + (s.rs - T.rs0), s.re - s.rs, s.bw1, o.zdrop, -1, EZ_APPROX_MAX);
    }
    if (T.right) {
        const int ql = T.qe0 - T.qe, tl = T.re0 - T.re;
        if (!local_result(o, ql, tl, &T.right_local))
            T.right_job = J.add(o, q0 + T.qe, ql, T.tseq.data() + (T.re - T.rs0), tl, T.bw, o.zdrop, o.end_bonus, EZ_EXTZ_ONLY);
    }
}

// after the first pass: the walk of mm_test_zdrop over every gap's CIGAR; a deep enough drop asks for the inversion test
void judge(Task &T, const Opt &o, const std::vector<ndgpu_ksw_result> &res) {
    if (T.skip) return;
    const uint8_t *q0 = T.R->qseq[T.rev];
    for (Seg &s : T.segs) {
        std::vector<uint32_t> cg;
        if (s.job1 >= 0 && res[(size_t)s.job1].n_cigar > 0) cg.assign(res[(size_t)s.job1].cigar, res[(size_t)s.job1].cigar + res[(size_t)s.job1].n_cigar);
        const uint8_t *qseq = q0 + s.qs, *tseq = T.tseq.data() + (s.rs - T.rs0);
        s.walk = zdrop_walk(o, qseq, tseq, cg);
        const int q_len = s.walk.pos[1][1] - s.walk.pos[1][0], t_len = s.walk.pos[0][1] - s.walk.pos[0][0];
        if (s.walk.max_zdrop > o.zdrop_inv && q_len < o.max_gap && t_len < o.max_gap) {
            s.ll_q.resize((size_t)(q_len > 0 ? q_len : 0));
            for (int k = 0; k < q_len; ++k) {
                const int c = qseq[s.walk.pos[1][1] - k - 1];
                s.ll_q[(size_t)k] = (uint8_t)(c >= 4 ? 4 : 3 - c);
            }
            s.ll = 1;  // wanted
        }
    }
}

// the last step of mm_align1 (align.c:691-787): join the pieces in chain order, stop at the first z-drop
void finish(Task &T, const Opt &o, const std::vector<ndgpu_ksw_result> &res1, const std::vector<ndgpu_ksw_result> &res2) {
    ReadCtx &R = *T.R;
    Reg &r = *T.reg;
    r.aligned = true;
    if (T.skip) return;
    const Anchor *a = R.a.data();
    const int32_t qlen = R.qlen;
    const uint8_t *q0 = R.qseq[T.rev];
    int32_t rs1, qs1, re1, qe1, dropped = 0;
    r.has_p = false, r.cigar.clear(), r.dp_score = 0, r.dp_max = 0;
    if (T.left) {
        const Ez ez = T.left_job >= 0 ? to_ez(res1[(size_t)T.left_job]) : T.left_local;
        if (!ez.cigar.empty()) {
            append_cigar(r, ez.cigar);
            r.dp_score += ez.max;
        }
        rs1 = T.rs - (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
        qs1 = T.qs - (ez.reach_end ? T.qs - T.qs0 : ez.max_q + 1);
    } else rs1 = T.rs, qs1 = T.qs;
    re1 = T.rs, qe1 = T.qs;
    if (T.cnt1 > 1) re1 = T.re, qe1 = T.qe;  // (the loop's last assignment when nothing z-drops)
    std::unique_ptr<Reg> r2;
    for (const Seg &s : T.segs) {
        Ez ez;
        if (s.code != 0 && s.job2 >= 0) ez = to_ez(res2[(size_t)s.job2]);
        else if (s.job1 >= 0) ez = to_ez(res1[(size_t)s.job1]);
        else local_result(o, s.qe - s.qs, s.re - s.rs, &ez);
        if (!ez.cigar.empty()) append_cigar(r, ez.cigar);
        if (ez.zdropped) {
            int j;
            for (j = s.i - 1; j >= 0; --j)
                if ((int32_t)a[T.as1 + j].x <= s.rs + ez.max_t) break;
            dropped = 1;
            if (j < 0) j = 0;
            r.dp_score += ez.max;
            re1 = s.rs + (ez.max_t + 1);
            qe1 = s.qs + (ez.max_q + 1);
            if (T.cnt1 - (j + 1) >= o.min_cnt) {
                r2.reset(new Reg());
                split_reg(r, *r2, T.as1 + j + 1 - r.as, qlen, a);
                if (r2->cnt > 0 && s.code == 2) r2->split_inv = 1;
                if (r2->cnt <= 0) r2.reset();
            }
            break;
        } else r.dp_score += ez.score;
    }
    if (!dropped && T.right) {
        const Ez ez = T.right_job >= 0 ? to_ez(res1[(size_t)T.right_job]) : T.right_local;
        if (!ez.cigar.empty()) {
            append_cigar(r, ez.cigar);
            r.dp_score += ez.max;
        }
        re1 = T.re + (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
        qe1 = T.qe + (ez.reach_end ? T.qe0 - T.qe : ez.max_q + 1);
    }
    r.rs = rs1, r.re = re1;
    if (T.rev) r.qs = qlen - qe1, r.qe = qlen - qs1;
    else r.qs = qs1, r.qe = qe1;
    if (r.has_p) {
        if (rs1 >= T.rs0 && re1 <= T.re0) update_extra(r, q0 + qs1, T.tseq.data() + (rs1 - T.rs0), o);
        else {  // (the reference asserts the window holds it)
            std::vector<uint8_t> t((size_t)std::max(0, re1 - rs1));
            T.tg->get((uint32_t)T.rid, rs1, re1, t.data());
            update_extra(r, q0 + qs1, t.data(), o);
        }
    }
    r.r2 = std::move(r2);
}

// ---- the inversion between two pieces (mm_align1_inv, align.c:790-845) ----

struct InvTask {
    ReadCtx *R = nullptr;
    const Reg *r1 = nullptr, *r2 = nullptr;
    bool go = false;
    int32_t ql = 0, tl = 0, q_off = 0, t_off = 0;
    std::vector<uint8_t> tseq, rq, rt;  // target [r1.re, r2.rs); both sequences reversed (the local alignment runs from the far end)
    const uint8_t *qseq = nullptr;
    int ll = -1, job = -1;
};

void inv_plan(InvTask &I, const Opt &o, const Targets &tg) {
    const Reg &r1 = *I.r1, &r2 = *I.r2;
    I.go = false;
    if (!(r1.split & 1) || !(r2.split & 2)) return;
    if (r1.id != r1.parent && r1.parent != kParentTmpPri) return;
    if (r2.id != r2.parent && r2.parent != kParentTmpPri) return;
    if (r1.rid != r2.rid || r1.rev != r2.rev) return;
    I.ql = r1.rev ? r1.qs - r2.qe : r2.qs - r1.qe;
    I.tl = r2.rs - r1.re;
    if (I.ql < o.min_chain_score || I.ql > o.max_gap) return;
    if (I.tl < o.min_chain_score || I.tl > o.max_gap) return;
    I.tseq.resize((size_t)I.tl);
    tg.get((uint32_t)r1.rid, r1.re, r2.rs, I.tseq.data());
    I.qseq = r1.rev ? I.R->qseq[0] + r2.qe : I.R->qseq[1] + (I.R->qlen - r2.qs);
    I.rq.assign(I.qseq, I.qseq + I.ql);
    I.rt = I.tseq;
    std::reverse(I.rq.begin(), I.rq.end());
    std::reverse(I.rt.begin(), I.rt.end());
    I.go = true;
}

// ---- the driver ----

bool dovetail(int rev, uint32_t qs, uint32_t qe, uint32_t qlen, uint32_t ts, uint32_t te, uint32_t tlen, int32_t h1, int32_t h2) {
    // check_realign_nextdenovo as the step-1 writer calls it (map.c:1301-1303)
    const uint32_t a = (uint32_t)h1, b = (uint32_t)h2;
    if (rev) {
        if (qs <= a && ts <= a) return true;
        else if (qlen - qe <= a && tlen - te <= a) return true;
    } else {
        if (qlen - qe <= a && ts <= a) return true;
        else if (qs <= a && tlen - te <= a) return true;
    }
    if (h2 > 0) {
        if (qs <= b && qe + b >= qlen) return true;
        if (ts <= b && te + b >= tlen) return true;
    }
    return false;
}

void free_results(std::vector<ndgpu_ksw_result> &r) {
    for (auto &x : r)
        if (x.cigar) free(x.cigar), x.cigar = nullptr;
}

// one round over a set of chains: three device batches (first pass, inversion tests, second pass)
int run_tasks(std::vector<Task> &tasks, const Opt &o, const Targets &tg, int threads, ndgpu_ovl_cigar_stats *st) {
    if (tasks.empty()) return 0;
    par_for(tasks.size(), threads, [&](size_t i) { plan(tasks[i], o, tg); });
    JobList J1;
    for (Task &T : tasks) pose_first(T, o, J1);
    std::vector<ndgpu_ksw_result> res1(J1.jobs.size());
    if (!J1.jobs.empty() && ([&] { const auto t0_ = std::chrono::steady_clock::now(); const int rc_ = ndgpu_ksw_extd2_batch(J1.jobs.data(), (int)J1.jobs.size(), res1.data()); g_t_ksw_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0_).count(); return rc_; }()) != 0) return -1;
    par_for(tasks.size(), threads, [&](size_t i) { judge(tasks[i], o, res1); });
    std::vector<ndgpu_ll_job> L;
    std::vector<Seg *> l_seg;
    for (Task &T : tasks)
        for (Seg &s : T.segs)
            if (s.ll == 1) {
                ndgpu_ll_job j;
                const int q_len = s.walk.pos[1][1] - s.walk.pos[1][0], t_len = s.walk.pos[0][1] - s.walk.pos[0][0];
                j.query = s.ll_q.data(), j.qlen = q_len, j.target = T.tseq.data() + (s.rs - T.rs0) + s.walk.pos[0][0], j.tlen = t_len;
                j.mat = o.mat, j.gapo = o.q, j.gape = o.e;
                s.ll = (int)L.size();
                L.push_back(j);
                l_seg.push_back(&s);
            }
    std::vector<ndgpu_ll_result> lres(L.size());
    if (!L.empty() && ([&] { const auto t0_ = std::chrono::steady_clock::now(); const int rc_ = ndgpu_ksw_ll_batch(L.data(), (int)L.size(), lres.data()); g_t_ll_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0_).count(); return rc_; }()) != 0) {
        free_results(res1);
        return -1;
    }
    JobList J2;
    for (Task &T : tasks) {
        if (T.skip) continue;
        const uint8_t *q0 = T.R->qseq[T.rev];
        for (Seg &s : T.segs) {
            s.code = s.walk.max_zdrop > o.zdrop ? 1 : 0;  // mm_test_zdrop's verdict (align.c:71-89)
            if (s.ll >= 0 && !L.empty()) {
                const int score = lres[(size_t)s.ll].score;
                if (score >= o.min_chain_score * o.a && score >= o.min_dp_max) s.code = 2;
            }
            if (s.code != 0 && s.job1 >= 0)
                s.job2 = J2.add(o, q0 + s.qs, s.qe - s.qs, T.tseq.data() + (s.rs - T.rs0), s.re - s.rs, s.bw1, s.code == 2 ? o.zdrop_inv : o.zdrop, -1, 0);
        }
    }
    std::vector<ndgpu_ksw_result> res2(J2.jobs.size());
    if (!J2.jobs.empty() && ([&] { const auto t0_ = std::chrono::steady_clock::now(); const int rc_ = ndgpu_ksw_extd2_batch(J2.jobs.data(), (int)J2.jobs.size(), res2.data()); g_t_ksw_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0_).count(); return rc_; }()) != 0) {
        free_results(res1);
        return -1;
    }
    par_for(tasks.size(), threads, [&](size_t i) { finish(tasks[i], o, res1, res2); });
    if (st) {
        st->chains += tasks.size(), st->first_pass += J1.jobs.size(), st->second_pass += J2.jobs.size(), st->inversion_tests += L.size();
        for (const auto &j : J1.jobs) st->cells += (uint64_t)j.qlen * (uint64_t)j.tlen;
        for (const auto &j : J2.jobs) st->cells += (uint64_t)j.qlen * (uint64_t)j.tlen;
        for (const Task &T : tasks) st->splits += T.reg->r2 != nullptr;
    }
    free_results(res1);
    free_results(res2);
    return 0;
}

int run_inversions(std::vector<InvTask> &inv, const Opt &o, const Targets &tg, ndgpu_ovl_cigar_stats *st) {
    if (inv.empty()) return 0;
    for (InvTask &I : inv) inv_plan(I, o, tg);
    std::vector<ndgpu_ll_job> L;
    for (InvTask &I : inv)
        if (I.go) {
            ndgpu_ll_job j;
            j.query = I.rq.data(), j.qlen = I.ql, j.target = I.rt.data(), j.tlen = I.tl, j.mat = o.mat, j.gapo = o.q, j.gape = o.e;
            I.ll = (int)L.size();
            L.push_back(j);
        }
    std::vector<ndgpu_ll_result> lres(L.size());
    if (!L.empty() && ([&] { const auto t0_ = std::chrono::steady_clock::now(); const int rc_ = ndgpu_ksw_ll_batch(L.data(), (int)L.size(), lres.data()); g_t_ll_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0_).count(); return rc_; }()) != 0) return -1;
    JobList J;
    for (InvTask &I : inv) {
        if (!I.go) continue;
        const ndgpu_ll_result &lr = lres[(size_t)I.ll];
        if (lr.score < o.min_dp_max) {
            I.go = false;
            continue;
        }
        I.q_off = I.ql - (lr.qe + 1), I.t_off = I.tl - (lr.te + 1);
        Ez tmp;
        if (local_result(o, I.ql - I.q_off, I.tl - I.t_off, &tmp)) {
            I.go = false;  // (nothing to align: the reference's "should never be here")
            continue;
        }
        I.job = J.add(o, I.qseq + I.q_off, I.ql - I.q_off, I.tseq.data() + I.t_off, I.tl - I.t_off, (int)(o.bw * 1.5), o.zdrop, -1, EZ_EXTZ_ONLY);
    }
    std::vector<ndgpu_ksw_result> res(J.jobs.size());
    if (!J.jobs.empty() && ([&] { const auto t0_ = std::chrono::steady_clock::now(); const int rc_ = ndgpu_ksw_extd2_batch(J.jobs.data(), (int)J.jobs.size(), res.data()); g_t_ksw_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0_).count(); return rc_; }()) != 0) return -1;
    for (InvTask &I : inv) {
        ReadCtx &R = *I.R;
        R.inv_done = true, R.inv_ok = false;
        if (!I.go || I.job < 0) continue;
        const Ez ez = to_ez(res[(size_t)I.job]);
        if (ez.cigar.empty()) continue;
        std::unique_ptr<Reg> v(new Reg());
        append_cigar(*v, ez.cigar);
        v->dp_score = ez.max;
        v->id = -1, v->parent = kParentUnset, v->inv = 1, v->rev = !I.r1->rev, v->rid = I.r1->rid;
        if (v->rev == 0) v->qs = I.r2->qe + I.q_off, v->qe = v->qs + ez.max_q + 1;
        else v->qe = I.r2->qs - I.q_off, v->qs = v->qe - (ez.max_q + 1);
        v->rs = I.r1->re + I.t_off, v->re = v->rs + ez.max_t + 1;
        v->aligned = true;
        update_extra(*v, I.qseq + I.q_off, I.tseq.data() + I.t_off, o);
        R.inv_ok = true;
        R.inv_reg = std::move(v);
    }
    if (st) st->inversions += inv.size(), st->first_pass += J.jobs.size(), st->inversion_tests += L.size(), st->inversions_aligned += J.jobs.size();
    free_results(res);
    return 0;
}

// the loop of mm_align_skeleton over one read's chains (align.c:875-903), resumable: returns when the chain it stands on has no
// result yet (*want_task) or when the inversion behind a split has not been looked at (*want_inv)
void advance(ReadCtx &R, Reg **want_task, bool *want_inv) {
    *want_task = nullptr, *want_inv = false;
    while (R.cur < R.regs.size()) {
        Reg &r = *R.regs[R.cur];
        if (R.stage == 0) {
            if (!r.aligned) {
                *want_task = &r;
                return;
            }
            if (r.r2 && r.r2->cnt > 0) R.regs.insert(R.regs.begin() + (long)R.cur + 1, std::move(r.r2));  // mm_insert_reg
            r.r2.reset();
            R.stage = 1;
            R.inv_done = false;
        }
        Reg &rc = *R.regs[R.cur];
        if (R.cur > 0 && rc.split_inv) {
            if (!R.inv_done) {
                *want_inv = true;
                return;
            }
            if (R.inv_ok) {
                R.regs.insert(R.regs.begin() + (long)R.cur + 1, std::move(R.inv_reg));
                ++R.cur;  // skip the inserted inversion
            }
        }
        R.stage = 0;
        ++R.cur;
    }
}

}  // namespace

extern "C" void ndgpu_ovl_aln_opt_default(ndgpu_ovl_aln_opt *o, int32_t min_chain_score) {  // mm_mapopt_init, options.c:36-43
    o->a = 2, o->b = 4, o->q = 4, o->e = 2, o->q2 = 24, o->e2 = 1, o->sc_ambi = 1, o->zdrop = 400, o->zdrop_inv = 200, o->end_bonus = -1;
    o->min_dp_max = 40 * 2;  // min_chain_score * a at the time mm_mapopt_init runs (40, before the preset raises it)
    (void)min_chain_score;
    o->min_ksw_len = 200, o->max_sw_mat = 0;  // (this minimap2 never sets a cap: mm_mapopt_init leaves it 0, --cap-sw-mem sets it, main.c:305)
    o->host_threads = 0;
}

extern "C" int64_t ndgpu_ovl_map_cigar(ndgpu_ovl_index *idx, const ndgpu_ovl_opt *opt, const ndgpu_ovl_aln_opt *aopt, int32_t mid_occ,
                                       uint32_t n_reads, const uint32_t *words, uint64_t n_words, const uint64_t *word_off, const uint32_t *lens,
                                       const uint32_t *ids, const uint32_t *t_words, const uint64_t *t_word_off, const uint32_t *t_lens,
                                       const uint32_t *t_ids, ndgpu_ovl_rec **recs, ndgpu_ovl_cigar_stats *stats) {
    *recs = nullptr;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!idx || !opt || !aopt || !t_words || !t_word_off || !t_lens || !t_ids) return -1;
    if (opt->step != 1 || opt->mode == 3) {
        fprintf(stderr, "[ndgpu_overlap] -c is built for --step 1 without --mode 3\n");
        return -1;
    }
    if (opt->k > 28) {
        fprintf(stderr, "[ndgpu_overlap] -c with k > 28 (ava-hifi): the compiled reference aborts there; not built\n");
        return -1;
    }
    // One gap piece (-O a -E b: q == q2, e == e2): the reference takes ksw_extz2_sse there (minimap2/align.c:313-331).  With the four
    // flag sets this path passes (EXTZ_ONLY | RIGHT | REV_CIGAR, APPROX_MAX, none, EXTZ_ONLY: align.c:697,733,745,764,821) that kernel
    // and ksw_extd2_sse with two equal pieces return the same scores, end points and CIGARs -- the second piece never beats the first,
    // so the two extra backtrack states never occur -- which tests/test_oracle_ksw2.py holds against the compiled ksw2_extz2_sse.c on
    // 12,000 fuzzed problems (they differ under APPROX_DROP, which this path does not use).  The two-piece kernel serves both.
    Opt o;
    o.a = aopt->a, o.b = aopt->b, o.q = aopt->q, o.e = aopt->e, o.q2 = aopt->q2, o.e2 = aopt->e2, o.sc_ambi = aopt->sc_ambi, o.zdrop = aopt->zdrop;
    o.zdrop_inv = aopt->zdrop_inv, o.end_bonus = aopt->end_bonus, o.min_dp_max = aopt->min_dp_max, o.min_ksw_len = aopt->min_ksw_len;
    o.max_sw_mat = aopt->max_sw_mat;
    o.k = opt->k, o.hpc = opt->hpc, o.min_cnt = opt->min_cnt, o.min_chain_score = opt->min_chain_score, o.bw = opt->bw, o.max_gap = opt->max_gap;
    o.minlen = opt->minlen, o.dvt = opt->dvt, o.maxhan1 = opt->maxhan1, o.maxhan2 = opt->maxhan2;
    {  // ksw_gen_simple_mat(5, mat, a, b, sc_ambi), align.c:9-22
        int8_t a = (int8_t)(o.a < 0 ? -o.a : o.a), b = (int8_t)(o.b > 0 ? -o.b : o.b), amb = (int8_t)(o.sc_ambi > 0 ? -o.sc_ambi : o.sc_ambi);
        for (int i = 0; i < 4; ++i) {
            for (int j = 0; j < 4; ++j) o.mat[i * 5 + j] = i == j ? a : b;
            o.mat[i * 5 + 4] = amb;
        }
        for (int j = 0; j < 5; ++j) o.mat[4 * 5 + j] = amb;
    }
    int threads = aopt->host_threads > 0 ? aopt->host_threads : (int)std::max(1u, std::thread::hardware_concurrency());
    const Targets tg{t_words, t_word_off, t_lens, t_ids};
    try {

    ndgpu_ovl_rec *chains = nullptr;
    uint32_t *counts = nullptr;
    uint64_t *ax = nullptr, *ay = nullptr, *a_off = nullptr;
    const auto t_call0 = std::chrono::steady_clock::now();
    g_t_ksw_ns = 0, g_t_ll_ns = 0;
    const int64_t n_ch = ndgpu_ovl_map_chains(idx, opt, mid_occ, n_reads, words, n_words, word_off, lens, ids, &chains, &counts, &ax, &ay, &a_off);
    const uint64_t chains_ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_call0).count();
    struct Free { void *p[5]; ~Free() { for (void *q : p) free(q); } };
    Free fr{{chains, counts, ax, ay, a_off}};
    if (n_ch < 0) return n_ch;

    std::vector<ndgpu_ovl_rec> out;
    // groups of reads whose chains are aligned together: bounded by the target windows the tasks hold decoded
    uint64_t group_budget = 1ull << 30;
    if (const char *e = getenv("NDGPU_CIGAR_GROUP_BASES")) group_budget = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
    uint32_t g0 = 0;
    std::vector<uint64_t> c_off((size_t)n_reads + 1, 0);
    for (uint32_t i = 0; i < n_reads; ++i) c_off[i + 1] = c_off[i] + counts[i];
    while (g0 < n_reads) {
        uint32_t g1 = g0;
        uint64_t cost = 0;
        while (g1 < n_reads && (g1 == g0 || cost < group_budget)) {
            for (uint64_t c = c_off[g1]; c < c_off[g1 + 1]; ++c) {
                const ndgpu_ovl_rec &h = chains[c];
                const uint64_t *x = ax + a_off[g1] + h.qs;
                cost += (uint64_t)((uint32_t)x[h.qe - 1] - (uint32_t)x[0]) + 2ull * (uint64_t)opt->max_gap;
            }
            ++g1;
        }
        std::vector<ReadCtx> reads(g1 - g0);
        par_for(reads.size(), threads, [&](size_t k) {
            const uint32_t i = g0 + (uint32_t)k;
            ReadCtx &R = reads[k];
            R.qid = ids[i], R.qlen = (int32_t)lens[i], R.n_a = (int32_t)(a_off[i + 1] - a_off[i]);
            if (counts[i] == 0) return;
            R.qbuf.assign(2 * (size_t)R.qlen + 16, 0);
            R.qseq[0] = R.qbuf.data(), R.qseq[1] = R.qbuf.data() + R.qlen;
            for (int32_t p = 0; p < R.qlen; ++p) {
                const uint8_t c = (uint8_t)(words[word_off[i] + ((uint32_t)p >> 4)] >> (30 - 2 * (p & 15)) & 3u);
                R.qbuf[(size_t)p] = c, R.qbuf[(size_t)R.qlen + (size_t)(R.qlen - 1 - p)] = (uint8_t)(3 - c);
            }
            R.a.resize((size_t)R.n_a);
            for (int32_t p = 0; p < R.n_a; ++p) R.a[(size_t)p].x = ax[a_off[i] + (uint64_t)p], R.a[(size_t)p].y = ay[a_off[i] + (uint64_t)p];
            for (uint64_t c = c_off[i]; c < c_off[i + 1]; ++c) {  // mm_gen_regs' population (hit.c:73-84)
                const ndgpu_ovl_rec &h = chains[c];
                std::unique_ptr<Reg> r(new Reg());
                r->id = (int32_t)(c - c_off[i]), r->parent = kParentUnset, r->score = (int32_t)h.tname, r->hash = h.ts, r->cnt = (int32_t)h.qe, r->as = (int32_t)h.qs;
                reg_set_coor(*r, R.qlen, R.a.data());
                R.regs.push_back(std::move(r));
            }
        });
        // round 1: every chain of every read; later rounds: the pieces splits left behind, then the inversions between pieces
        std::vector<Task> tasks;
        for (ReadCtx &R : reads)
            for (auto &r : R.regs) {
                Task T;
                T.R = &R, T.reg = r.get();
                tasks.push_back(std::move(T));
            }
        for (;;) {
            if (run_tasks(tasks, o, tg, threads, stats) != 0) return -2;
            tasks.clear();
            std::vector<InvTask> inv;
            for (ReadCtx &R : reads) {  // every read moves on as far as its results reach
                Reg *want = nullptr;
                bool want_inv = false;
                advance(R, &want, &want_inv);
                if (want) {
                    Task T;
                    T.R = &R, T.reg = want;
                    tasks.push_back(std::move(T));
                } else if (want_inv) {
                    InvTask I;
                    I.R = &R, I.r1 = R.regs[R.cur - 1].get(), I.r2 = R.regs[R.cur].get();
                    inv.push_back(std::move(I));
                }
            }
            if (run_inversions(inv, o, tg, stats) != 0) return -2;
            if (tasks.empty() && inv.empty()) break;
        }
        // mm_filter_regs, mm_hit_sort, the writer (hit.c:257-276,169-201; map.c:1297-1304)
        std::vector<std::vector<ndgpu_ovl_rec>> per(reads.size());
        par_for(reads.size(), threads, [&](size_t k) {
            ReadCtx &R = reads[k];
            std::vector<Reg *> v;
            for (auto &r : R.regs) {
                bool flt = false;
                if (!r->inv && r->cnt < o.min_cnt) flt = true;
                if (r->has_p) {
                    if (r->mlen < o.min_chain_score) flt = true;
                    else if (r->dp_max < o.min_dp_max) flt = true;
                }
                if (!flt) v.push_back(r.get());
            }
            if (v.size() > 1) {
                std::vector<X128> aux;
                for (size_t i = 0; i < v.size(); ++i)
                    if (v[i]->inv || v[i]->cnt > 0) {
                        X128 z;
                        z.x = (uint64_t)(uint32_t)(v[i]->has_p ? v[i]->dp_max : v[i]->score) << 32 | v[i]->hash;
                        z.y = i;
                        aux.push_back(z);
                    }
                radix_sort_128x(aux.data(), aux.data() + aux.size());
                std::vector<Reg *> t(aux.size());
                for (size_t i = 0; i < aux.size(); ++i) t[aux.size() - 1 - i] = v[(size_t)aux[i].y];
                v.swap(t);
            }
            for (Reg *r : v) {
                const uint32_t tid = tg.ids[(uint32_t)r->rid];
                if (tid == R.qid) continue;
                if (r->qe - r->qs < o.minlen) continue;
                if (o.dvt && !dovetail((int)r->rev, (uint32_t)r->qs, (uint32_t)r->qe, (uint32_t)R.qlen, (uint32_t)r->rs, (uint32_t)r->re,
                                       tg.lens[(uint32_t)r->rid], o.maxhan1, o.maxhan2)) continue;
                ndgpu_ovl_rec x;
                x.rev = r->rev, x.qname = R.qid, x.qs = (uint32_t)r->qs, x.qe = (uint32_t)r->qe, x.tname = tid, x.ts = (uint32_t)r->rs, x.te = (uint32_t)r->re;
                x.match = (uint32_t)r->mlen;
                per[k].push_back(x);
            }
        });
        for (auto &p : per) out.insert(out.end(), p.begin(), p.end());
        g0 = g1;
    }
    *recs = (ndgpu_ovl_rec *)malloc(sizeof(ndgpu_ovl_rec) * (out.size() ? out.size() : 1));
    if (!*recs) return -2;
    if (!out.empty()) memcpy(*recs, out.data(), sizeof(ndgpu_ovl_rec) * out.size());
    if (stats) {
        stats->overlaps = out.size();
        stats->chains_ns = chains_ns, stats->ksw_ns = g_t_ksw_ns.load(), stats->ksw_ll_ns = g_t_ll_ns.load();
        stats->total_ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_call0).count();
    }
    return (int64_t)out.size();
    } catch (const std::exception &e) {  // (host memory: nothing of this may cross the C boundary)
        fprintf(stderr, "[ndgpu_overlap] ndgpu_ovl_map_cigar: %s\n", e.what());
        return -2;
    }
}
