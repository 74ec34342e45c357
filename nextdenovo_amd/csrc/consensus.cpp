// Per-seed consensus engine (host side).
//
// Everything the reference does inside nextCorrect() (lib/nextcorrect.c:2219-2305)
// EXCEPT the pairwise O(ND) alignments, which are requested as AlnJob batches and
// executed by the HIP kernels (csrc/ond_kernels.hip).  The logic here reproduces
// the reference's observable behaviour bit for bit (tie-breaks, integer/float
// conversions, first-seen ordering) but on our own data layout:
//   * alignments arrive as column-kind streams (match / query-only / target-only)
//     instead of two gapped strings,
//   * the MSA link graph is one flat cell table + one link arena per pile
//     (index-linked lists) instead of per-column malloc blocks,
//   * low-quality regions keep std::string sequences instead of 10 kB slots,
//   * sorting uses std::stable_sort, which yields the same permutation as glibc's
//     merge-sort qsort() on the reference's comparators.
// Reference locations are cited at each step.
#include "nd_host.h"

#include <algorithm>
#include <cctype>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <cmath>
#include <sched.h>

namespace ndgpu {
static inline uint64_t now_ns();

ConsensusTrimed *make_error_seed(unsigned len) {
    // lib/nextcorrect.c:261-266.  The reference leaves the buffer uninitialised;
    // callers only look at `len` (lib/nextcorrect.py:236,255).  We zero it.
    ConsensusTrimed *c = (ConsensusTrimed *)malloc(sizeof(ConsensusTrimed));
    c->len = len;
    c->identity = 0;
    c->seq = (char *)calloc(len + 1, 1);
    return c;
}

namespace {

// lib/nextcorrect.c:42-62
const unsigned char kIntToBase[7] = {'A', 'T', 'G', 'C', '-', 'N', 'M'};
struct BaseLut {
    uint8_t v[256];
    BaseLut() {
        memset(v, 4, sizeof(v));
        v['A'] = v['a'] = 0;
        v['T'] = v['t'] = 1;
        v['G'] = v['g'] = 2;
        v['C'] = v['c'] = 3;
        v['N'] = 5;
        v['M'] = 6;
    }
};
const BaseLut kLut;
inline uint8_t base_code(char c) { return kLut.v[(unsigned char)c]; }

struct Tag {
    int32_t t_pos;
    uint16_t delta;
    uint8_t base;
};

struct TagList {
    unsigned aln_t_s = 0;
    std::vector<Tag> tags;
};

struct Link {
    int32_t pp_t;
    int32_t ppp_t;
    uint16_t pp_d;
    uint16_t ppp_d;
    uint8_t pp_b;
    uint8_t ppp_b;
    uint16_t count;
    int32_t next;
    int64_t score;
};

struct Cell {
    int32_t head = -1, tail = -1;
    uint32_t len = 0;
    int64_t best_score = 0;
    int32_t best_t = 0;  // calloc-zero initial state (lib/nextcorrect.c:181)
    uint16_t best_d = 0;
    uint8_t best_b = 0;
    uint16_t best_link = 0;
};

struct Pos {
    int32_t t;
    uint16_t d;
    uint8_t b;
};

// Flat MSA link graph of one (pseudo-)seed.
struct Msa {
    std::vector<uint16_t> max_size, coverage;
    std::vector<uint32_t> cell_base;
    std::vector<Cell> cells;
    std::vector<Link> links;

    explicit Msa(size_t ncol) : max_size(ncol, 0), coverage(ncol, 0) {}
    size_t ncol() const { return max_size.size(); }
    Cell &cell(int32_t t, unsigned d, unsigned b) { return cells[cell_base[t] + d * 6 + b]; }

    // lib/nextcorrect.c:175-198
    void allocate() {
        cell_base.resize(ncol() + 1);
        uint32_t acc = 0;
        for (size_t p = 0; p < ncol(); p++) {
            cell_base[p] = acc;
            acc += (uint32_t)max_size[p] * 6;
        }
        cell_base[ncol()] = acc;
        cells.assign(acc, Cell());
    }

    // lib/nextcorrect.c:212-250: count (pp,ppp) predecessor pairs per cell,
    // keeping first-seen order inside each cell.
    void count_links(const std::vector<TagList> &reads) {
        size_t total = 0;
        for (const TagList &r : reads) total += r.tags.size();
        links.reserve(total / 4 + 16);
        const Tag head{-1, 0, 0};
        for (const TagList &r : reads) {
            const size_t n = r.tags.size();
            for (size_t i = 0; i < n; i++) {
                const Tag &cur = r.tags[i];
                const Tag &pp = i > 0 ? r.tags[i - 1] : head;
                const Tag &ppp = i > 1 ? r.tags[i - 2] : head;
                if (cur.base == 6 || pp.base == 6) continue;
                Cell &c = cell(cur.t_pos, cur.delta, cur.base);
                int32_t at = c.head;
                while (at != -1) {
                    Link &l = links[at];
                    if (l.pp_t == pp.t_pos && l.pp_d == pp.delta && l.pp_b == pp.base && l.ppp_t == ppp.t_pos &&
                        l.ppp_d == ppp.delta && l.ppp_b == ppp.base) {
                        l.count++;
                        break;
                    }
                    at = l.next;
                }
                if (at == -1) {
                    Link l;
                    l.pp_t = pp.t_pos; l.pp_d = pp.delta; l.pp_b = pp.base;
                    l.ppp_t = ppp.t_pos; l.ppp_d = ppp.delta; l.ppp_b = ppp.base;
                    l.count = 1; l.score = 0; l.next = -1;
                    int32_t id = (int32_t)links.size();
                    links.push_back(l);
                    if (c.tail == -1) c.head = id;
                    else links[c.tail].next = id;
                    c.tail = id;
                    c.len++;
                }
            }
        }
    }

    // lib/nextcorrect.c:1263-1300 (second, low-quality-region MSA): 6 symbols,
    // plain max, origin = last cell.
    Pos score_lq(int factor) {
        Pos origin{-1, 0, 0};
        const int ncols = (int)ncol();
        for (int p = 0; p < ncols; p++) {
            const int64_t penalty = (int64_t)factor * coverage[p];
            for (unsigned d = 0; d < max_size[p]; d++) {
                for (unsigned b = 0; b < 6; b++) {
                    Cell &c = cell(p, d, b);
                    c.best_score = -10;
                    c.best_t = -1;
                    for (int32_t m = c.head; m != -1; m = links[m].next) {
                        Link &lm = links[m];
                        if (lm.pp_t == -1) {
                            lm.score = 10 * (int64_t)lm.count - penalty;
                        } else {
                            const Cell &pc = cell(lm.pp_t, lm.pp_d, lm.pp_b);
                            for (int32_t n = pc.head; n != -1; n = links[n].next) {
                                const Link &ln = links[n];
                                if (ln.pp_t != lm.ppp_t || ln.pp_d != lm.ppp_d || ln.pp_b != lm.ppp_b) continue;
                                int64_t s = ln.score + 10 * (int64_t)lm.count - penalty;
                                if (s > lm.score) lm.score = s;
                            }
                        }
                        if (lm.score > c.best_score || (lm.score == c.best_score && lm.pp_b != 4)) {
                            c.best_score = lm.score;
                            c.best_t = lm.pp_t; c.best_d = lm.pp_d; c.best_b = lm.pp_b;
                            c.best_link = lm.count;
                        }
                    }
                    origin = Pos{p, (uint16_t)d, (uint8_t)b};
                }
            }
        }
        return origin;
    }
};

// get_align_tags (lib/nextcorrect.c:1485-1536) for an explicit pair of gapped strings.
void tags_from_strings(const std::string &t_str, const std::string &q_str, unsigned aln_t_s, TagList &out, Msa &msa) {
    const size_t n = t_str.size();
    out.aln_t_s = aln_t_s;
    out.tags.resize(n);
    int32_t t = (int32_t)aln_t_s - 1;
    uint16_t delta = 0;
    for (size_t i = 0; i < n; i++) {
        if (t_str[i] != '-') {
            t++;
            delta = 0;
        }
        Tag &g = out.tags[i];
        g.t_pos = t;
        g.delta = delta++;
        g.base = base_code(q_str[i]);
        if (g.delta == 0 && q_str[i] != 'M') msa.coverage[t]++;
        if (g.delta >= msa.max_size[t]) msa.max_size[t] = g.delta + 1;
    }
}

struct LqSeq {
    uint16_t order = 0, kscore = 0, len = 0;
    std::string seq;
    std::vector<uint32_t> packed;  // seq in the device's 2-bit form, made once for both low-quality-region rounds
};

struct LqRegion {
    unsigned start = 0, end = 0;
    int len = 0;
    uint8_t indexs = 0, indexe = 0;
    unsigned lqcount = 0;
    unsigned sudoseed_len = 0;
    bool has_seed = false;
    std::string sudoseed;  // sudoseed_len valid bytes
    std::vector<LqSeq> seqs;
};

struct CnsBase {
    unsigned pos;
    char base;
};

struct CnsData {
    unsigned len = 0, uncorrected_len = 0, lstrip = 0, rstrip = 0;
    std::vector<CnsBase> bases;
};

struct LqReg {
    unsigned start = 0, end = 0, lqlen = 0, lq_total_len = 0;
};
constexpr int kLqRegMax = 10;  // LQREG_MAX_COUNT

// lib/nextcorrect.c:1340-1363
int update_lqreg(LqReg *lq, const std::string &seq, unsigned p, int i, unsigned *hq_m, unsigned *lq_m) {
    if (seq[p] >= 'a') {
        if (!lq[i].lqlen) lq[i].start = p;
        if ((*lq_m)++ > 2) *hq_m = 0;
        lq[i].end = p;
        lq[i].lqlen++;
        lq[i].lq_total_len++;
    } else {
        if (lq[i].lqlen && lq[i].start == 0) {
            i++;
            *hq_m = 0;
        } else if (*hq_m + lq[i].start > lq[i].end || (*hq_m)++ > 10) {
            if (lq[i].end > lq[i].start + 100) i++;
            else lq[i].lqlen = lq[i].end = 0;
            *hq_m = 0;
        } else if (*hq_m >= lq[i].lqlen) {
            lq[i].lqlen = lq[i].end = 0;
            *hq_m = 0;
        }
        *lq_m = 0;
    }
    return i;
}

// ---- 8-mer ranking of low-quality-region candidates (lib/nextcorrect.c:281-337)
constexpr int kKmerLen = 8, kKmerRange = 40, kKmerBins = 65536, kKmerMaxSeq = 10;
constexpr int kLqCanMax = 40, kLqSeqMax = 30, kLqRevLen = 2000;

// The 8-mer histogram of the reference lives on the stack and is cleared per call (lib/nextcorrect.c:281-300: 128 KB); a call touches
// at most 40 x 32 of its 65,536 bins, so the clearing -- three to four times per region, tens of thousands of regions per step -- was
// gigabytes of memset.  The bins a call touched are remembered and only those are cleared by the next one.
struct KmerBins {
    std::vector<uint16_t> v = std::vector<uint16_t>(kKmerBins, 0);
    std::vector<uint16_t> touched;
    uint16_t *data() { return v.data(); }
};

void count_kmers(const LqRegion &lq, KmerBins &kb, int c, int from_tail) {
    uint16_t *bins = kb.v.data();
    for (uint16_t t : kb.touched) bins[t] = 0;
    kb.touched.clear();
    const int lim = std::min(lq.len, c);
    for (int j = 0; j < lim; j++) {
        const LqSeq &s = lq.seqs[j];
        if (s.len < kKmerLen) continue;
        const int off = from_tail && s.len > kKmerRange ? s.len - kKmerRange : 0;
        const int n = std::min<int>(s.len, kKmerRange) - kKmerLen;
        uint16_t km = 0;
        for (int k = 0; k < n; k++) {
            if (k) km = (uint16_t)(km << 2 | base_code(s.seq[off + k + kKmerLen - 1]));
            else
                for (int x = 0; x < kKmerLen; x++) km = (uint16_t)(km << 2 | base_code(s.seq[off + x]));
            if (bins[km]++ == 0) kb.touched.push_back(km);
        }
    }
}

void count_kscore(LqRegion &lq, const uint16_t *bins, int from_tail) {
    for (int j = 0; j < lq.len; j++) {
        LqSeq &s = lq.seqs[j];
        s.kscore = 0;
        if (s.len < kKmerLen) continue;
        const int off = from_tail && s.len > kKmerRange ? s.len - kKmerRange : 0;
        const int n = std::min<int>(s.len, kKmerRange) - kKmerLen;
        uint16_t km = 0;
        for (int k = 0; k < n; k++) {
            if (k) km = (uint16_t)(km << 2 | base_code(s.seq[off + k + kKmerLen - 1]));
            else
                for (int x = 0; x < kKmerLen; x++) km = (uint16_t)(km << 2 | base_code(s.seq[off + x]));
            s.kscore = (uint16_t)(s.kscore + bins[km]);
        }
    }
}

void sort_by_kscore_desc(LqRegion &lq) {
    // qsort(compare_seq_by_kscore) (lib/nextcorrect.c:254-258,413); glibc qsort is a
    // stable merge sort, so the permutation equals std::stable_sort's.
    std::stable_sort(lq.seqs.begin(), lq.seqs.begin() + lq.len,
                     [](const LqSeq &a, const LqSeq &b) { return a.kscore > b.kscore; });
}

// ---- terminal SSR clipping (lib/nextcorrect.c:2008-2128)
int terminal_ssr(int *bins, int range, int klen, const char *seq, int s) {
    memset(bins, 0, sizeof(int) * 256);
    uint8_t km = 0;
    for (int i = 0; i < range; i++) {
        if (i) km = (uint8_t)(km << 2 | base_code(seq[s + i + klen - 1]));
        else
            for (int k = 0; k < klen; k++) km = (uint8_t)(km << 2 | base_code(seq[s + k]));
        bins[km]++;
    }
    int best = 0;
    for (int i = 0; i < 256; i++)
        if (bins[i] > best) {
            best = bins[i];
            km = (uint8_t)i;
        }
    return km;
}

int clip_ssr(const char *seq, int seq_len, int klen, int kmer, int from_end) {
    const int gap = 20;
    int i, p = 0, p1 = 0, p2 = 0;
    uint8_t cur = 0;
    if (from_end) {
        uint8_t rk = 0;
        for (i = 0; i < 8; i += 2) rk = (uint8_t)(rk << 2 | (kmer >> i & 3));
        seq_len--;
        kmer = rk;
        for (i = 0; i < seq_len - klen; i++) {
            if (i) cur = (uint8_t)(cur << 2 | base_code(seq[seq_len - i - klen + 1]));
            else
                for (int k = 0; k < klen; k++) cur = (uint8_t)(cur << 2 | base_code(seq[seq_len - k]));
            if (cur != kmer) {
                if (i - p > gap) {
                    if (!p1) p1 = p;
                    else if (p2) {
                        if (i - p2 < 100) { p = p1; break; }
                        else p1 = p2 = 0;
                    }
                }
            } else {
                p = i;
                if (p1 && p2 == 0) p2 = p;
            }
        }
    } else {
        for (i = 0; i < seq_len - klen; i++) {
            if (i) cur = (uint8_t)(cur << 2 | base_code(seq[i + klen - 1]));
            else
                for (int k = 0; k < klen; k++) cur = (uint8_t)(cur << 2 | base_code(seq[k]));
            if (cur != kmer) {
                if (i - p > gap) {
                    if (!p1) p1 = p;
                    else if (p2) {
                        if (i - p2 < 100) { p = p1; break; }
                        else p1 = p2 = 0;
                    }
                }
            } else {
                p = i;
                if (p1 && p2 == 0) p2 = p;
            }
        }
    }
    return p > 100 ? p + klen : 0;
}

struct Result {
    unsigned len = 0;
    float identity = 0;
    std::string seq;
};

void trim_terminal_ssr(Result &r) {
    int bins[256];
    const int range = 24, klen = 4;
    int clip_s = 0, clip_e = 0;
    const int L = (int)r.len;
    int km = terminal_ssr(bins, range, klen, r.seq.c_str(), 0);
    if (bins[km] >= 4) {
        clip_s = clip_ssr(r.seq.c_str(), L, klen, km, 0);
        while (clip_s < L && r.seq[clip_s] >= 'a') clip_s++;
    }
    km = terminal_ssr(bins, range, klen, r.seq.c_str(), L - range - klen + 1);
    if (bins[km] >= 4) {
        clip_e = clip_ssr(r.seq.c_str(), L, klen, km, 1);
        while (clip_e < L && r.seq[L - clip_e - 1] >= 'a') clip_e++;
    }
    if (clip_s + clip_e < L - 10) {
        r.seq = r.seq.substr(clip_s, L - clip_s - clip_e);
        r.len = (unsigned)(L - clip_s - clip_e);
    } else {
        r.len = 4;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------

class PileImpl {
  public:
    typedef PileEngine::Phase Phase;

    CorrectParams prm;
    unsigned lq_max_len;
    std::vector<std::string> own_seqs;      // ASCII form only
    std::vector<const char *> seq_ptr;
    std::vector<unsigned> seq_len, aln_start, aln_end;
    std::vector<int64_t> dev_off;
    int seed_len = 0;
    Phase phase = PileEngine::MAIN;
    MainPile main;
    ExtractPile extract;
    std::vector<AlnJob> jobs;
    Result result;
    bool error2 = false;
    unsigned n_aligned = 0;

    // state carried through the low-quality-region rounds
    std::vector<LqRegion> regions;
    CnsData cns;
    int lq_iter = 0;          // 1-based round being aligned
    int lq_max_aln_length = 0;
    int lq_max_dif_len = 0;
    struct LqSlot { int i, j, job; };  // (row, region) -> job index or -1 ('M' fill)
    std::vector<LqSlot> lq_slots;
    LqRound lq;               // the round as a device request (HipBackend: K12); lq.ok: its result is in lq.lqc

    std::string seed_copy;  // DB form, HiFi only: the seed's own bases

    PileImpl(const char *const *s, const unsigned *len, const int64_t *dev, const unsigned *st, const unsigned *en,
             unsigned n, const CorrectParams &p, const char *seed_ascii)
        : prm(p) {
        if (seed_ascii) seed_copy = seed_ascii;
        lq_max_len = p.lqseq_max_length > 10000 ? 10000 : p.lqseq_max_length;  // DAG_MAX_LENGTH, nextcorrect.c:2231
        aln_start.assign(st, st + n);
        aln_end.assign(en, en + n);
        if (s) {
            own_seqs.reserve(n);
            for (unsigned i = 0; i < n; i++) own_seqs.emplace_back(s[i]);
            for (unsigned i = 0; i < n; i++) {
                seq_ptr.push_back(own_seqs[i].c_str());
                seq_len.push_back((unsigned)own_seqs[i].size());
            }
        } else {
            seq_len.assign(len, len + n);
            dev_off.assign(dev, dev + n);
        }
        if (n == 0) { finish_error(2); return; }
        seed_len = (int)en[0] + 1;
        // the seed is its own alignment over [aln_start[0], aln_end[0]] (nextcorrect.c:2279-2282);
        // lib/nextcorrect.py always passes 0 .. len-1
        if (st[0] != 0 || seq_len[0] != (unsigned)seed_len) {
            fprintf(stderr, "[ndgpu] seed record must span the whole seed (aln_start 0, aln_end len-1)\n");
            finish_error(2);
            return;
        }
        for (unsigned i = 1; i < n; i++)
            if (en[i] < st[i] || en[i] >= (unsigned)seed_len) {
                fprintf(stderr, "[ndgpu] overlap window outside the seed\n");
                finish_error(2);
                return;
            }
        main.n = n;
        main.seqs = s ? seq_ptr.data() : nullptr;
        main.seq_len = seq_len.data();
        main.aln_start = aln_start.data();
        main.aln_end = aln_end.data();
        main.dev_off = s ? nullptr : dev_off.data();
        main.min_len_aln = prm.min_len_aln;
        main.max_cov_aln = prm.max_cov_aln;
        main.factor = prm.read_type == 3 ? 4 : 3;  // nextcorrect.c:2147
        main.hq = prm.read_type == 3;              // align_hq for HiFi, nextcorrect.c:2269
    }

    void finish_error(unsigned code) {
        result = Result();
        result.len = code;
        error2 = true;
        phase = PileEngine::DONE;
    }

    void collect(std::vector<AlnJob *> &out) {
        if (phase != PileEngine::LQ_ROUND || lq.ok) return;
        for (AlnJob &j : jobs) out.push_back(&j);
    }

    void advance() {
        const uint64_t t0 = now_ns();
        const int ph = phase == PileEngine::MAIN ? 0 : phase == PileEngine::EXTRACT ? 1 : lq_iter <= 1 ? 2 : 3;
        if (phase == PileEngine::MAIN) after_main();
        else if (phase == PileEngine::EXTRACT) after_extract();
        else if (phase == PileEngine::LQ_ROUND) after_lq_round();
        g_prof.adv_ns[ph] += now_ns() - t0;  // per-thread sums: after main / after extract / after LQ round 1 / after round 2 + splice
    }

    // -- main phase: the device returns the best_pp walk --------------------------------
    void after_main() {
        std::vector<PathStep> path;
        path.swap(main.path);
        if (path.empty()) {  // no column carries a tag (reference: undefined behaviour)
            finish_error(2);
            return;
        }
        if (prm.fast) {
            cns_fast(path);
            phase = PileEngine::DONE;
            return;
        }
        n_aligned = main.n_aligned;
        const bool ok = prm.read_type == 3 ? cns_kmer(path) : cns_from_best_score(path);
        if (!ok) {
            finish_error(2);
            return;
        }
        lq_iter = 0;
        lq_max_dif_len = 0;
        if (regions.empty()) {
            lq_max_aln_length = 0;
            start_lq_round();
            return;
        }
        extract.slot = main.slot;
        extract.regions.resize(regions.size());
        for (size_t i = 0; i < regions.size(); i++) {
            extract.regions[i].start = regions[i].start;
            extract.regions[i].end = regions[i].end;
            extract.regions[i].max_len = lq_max_len;
            extract.regions[i].max_len0 = prm.read_type == 3 ? 10000 : 0;  // DAG_MAX_LENGTH, nextcorrect.c:765
        }
        phase = PileEngine::EXTRACT;
    }

    uint64_t poa_ns_local = 0;
    void after_extract() {
        const uint64_t t0 = now_ns();
        poa_ns_local = 0;
        lq_max_aln_length = prm.read_type == 3 ? lqseqs_from_candidates_kmer() : lqseqs_from_candidates();
        extract.regions.clear();
        const uint64_t t1 = now_ns();
        start_lq_round();
        g_prof.poa_ns += poa_ns_local;
        g_prof.rank_ns += t1 - t0 - poa_ns_local;
        g_prof.lqstart_ns += now_ns() - t1;
    }

    // lib/nextcorrect.c:1717-1784
    void cns_fast(const std::vector<PathStep> &path) {
        std::string out;
        LqReg lq[kLqRegMax];
        int lq_i = 0;
        const int min_cov = (int)prm.min_cov;
        for (const PathStep &cur : path) {
            if (cur.base != 4) {
                if ((int)cur.cov > min_cov) {
                    out.push_back((char)kIntToBase[cur.base]);
                    if (lq[lq_i].end >= lq[lq_i].start + 50 || !lq_i) {
                        if (++lq_i >= kLqRegMax) break;
                    } else lq[lq_i].end = 0;
                } else {
                    out.push_back((char)tolower(kIntToBase[cur.base]));
                    if (!lq[lq_i].end) {
                        lq[lq_i].start = (unsigned)out.size() - 1;
                        lq[lq_i].lqlen = 0;
                    }
                    lq[lq_i].end = (unsigned)out.size() - 1;
                    lq[lq_i].lq_total_len++;
                    lq[lq_i].lqlen++;
                }
            }
        }
        int i, lq_m = 0, hq_m = (int)lq[0].start, l = hq_m;
        unsigned lq_total_len = lq[0].lq_total_len - lq[0].lqlen;
        for (i = 1; i < kLqRegMax && lq[i].end; i++) {
            if (lq[i].start - lq[i - 1].end > (unsigned)l) {
                lq_m = (int)lq[i - 1].end + 1;
                hq_m = (int)lq[i].start;
                lq_total_len = lq[i].lq_total_len - lq[i].lqlen;
                l = hq_m - lq_m;
            }
        }
        if (i < kLqRegMax && (unsigned)out.size() - lq[i - 1].end > (unsigned)l) {
            lq_m = (int)lq[i - 1].end + 1;
            hq_m = (int)out.size();
            lq_total_len = lq[i].lq_total_len;
        }
        result.len = (unsigned)(hq_m - lq_m);
        result.identity = 1 - (float)lq_total_len / result.len;
        result.seq = (size_t)lq_m <= out.size() ? out.substr((size_t)lq_m, result.len) : std::string();
        std::reverse(result.seq.begin(), result.seq.end());
    }

    // lib/nextcorrect.c:1885-2006.  Returns false for the error_seed(2) outcome.
    bool cns_from_best_score(const std::vector<PathStep> &path) {
        int p = 0, lable = 1;
        const int lq_min_length = 8;
        int qv = 0, pqv, hq = 0, lq = 0, lq_l = 0, lq_s = -1, lq_e = -1;
        int lqseq_total_length = 0;
        const int min_cov = (int)prm.min_cov;
        cns = CnsData();
        cns.bases.reserve(path.size());
        regions.clear();
        for (const PathStep &cur : path) {
            if (cur.base == 4) continue;
            cns.bases.push_back(CnsBase{(unsigned)cur.t_pos, 0});
            const int cov = cur.cov;
            pqv = 100 * (int)cur.link / cov;
            if (pqv > 40) hq++;
            else {
                hq = 0;
                lqseq_total_length++;
            }
            if (hq > lq_min_length / 2 && lq_e - lq_s < lq_min_length / 2) {
                qv = lq_l = lq = 0;
                lq_s = -1;
            }
            if ((qv + pqv) / (lq_l + 1) < 40) {
                if (lq_s == -1) lq_s = p;
                lq_e = p;
                lq = 1;
                lq_l++;
                qv += pqv;
            } else if (lq && p - lq_e > 2 * lq_min_length && cns.bases[p].pos != cns.bases[p - 1].pos) {
                if (lq_e - lq_s + 1 > lq_min_length && (unsigned)(lq_e - lq_s + 1) < lq_max_len) {
                    lq_e = p - lq_min_length - 1;
                    lq_s = lq_s > lq_min_length ? lq_s - lq_min_length : 1;
                    LqRegion r;
                    r.end = cns.bases[lq_s].pos;
                    r.start = cns.bases[lq_e].pos;
                    if (!regions.empty() && r.end == regions.back().start) {
                        while (r.end == regions.back().start && lq_s < p - 4) r.end = cns.bases[++lq_s].pos;
                    }
                    regions.push_back(std::move(r));
                }
                qv = lq_l = lq = 0;
                lq_s = -1;
            } else if (lq && cns.bases[p].pos != cns.bases[p - 1].pos) {
                qv = lq_l = 0;
            }
            if (cov > min_cov && pqv > 20) {
                cns.bases[p].base = (char)kIntToBase[cur.base];
                lable = 0;
                cns.lstrip = 0;
            } else {
                cns.bases[p].base = (char)tolower(kIntToBase[cur.base]);
                cns.uncorrected_len++;
                cns.lstrip++;
                if (lable) cns.rstrip++;
            }
            p++;
        }
        cns.len = (unsigned)p;
        const float lhs = (float)(cns.uncorrected_len - cns.lstrip - cns.rstrip);
        const float rhs = (float)(cns.len - cns.lstrip - cns.rstrip) * (1 - prm.min_error_corrected_ratio);
        if (!(cns.len > 2 && (double)lqseq_total_length < cns.len * 0.8 && lhs < rhs)) return false;
        std::reverse(cns.bases.begin(), cns.bases.end());
        return true;
    }

    // HiFi backtrack + LQ windows: lib/nextcorrect.c:1786-1883
    bool cns_kmer(const std::vector<PathStep> &path) {
        const char *seed = !seq_ptr.empty() ? seq_ptr[0] : (seed_copy.empty() ? nullptr : seed_copy.c_str());
        if (!seed) {
            fprintf(stderr, "[ndgpu] HiFi consensus needs the seed sequence on the host\n");
            return false;
        }
        const int lq_min_length = 2, dag_min_qv = 80;
        int p = 0, qv, lq = 0, lq_s = -1, lq_e = -1, lqseq_total_length = 0;
        const int min_cov = (int)prm.min_cov;
        cns = CnsData();
        cns.bases.reserve(path.size());
        regions.clear();
        for (const PathStep &cur : path) {
            if (cur.base == 4) continue;
            cns.bases.push_back(CnsBase{(unsigned)cur.t_pos, 0});
            const int cov = cur.cov;
            qv = 100 * (int)cur.link / cov;
            if (cov < 4) {
                lq = 0;
                lq_s = -1;
                lqseq_total_length++;
            } else if (qv < dag_min_qv || cur.base != base_code(seed[cur.t_pos])) {
                if (lq_s == -1) lq_s = p;
                lq_e = p;
                lq = 1;
                lqseq_total_length++;
            } else if (lq && p - lq_e > 2 * lq_min_length && cns.bases[p].pos != cns.bases[p - 1].pos) {
                lq_e = p - lq_min_length - 1;
                lq_s = lq_s > lq_min_length ? lq_s - lq_min_length : 1;
                if (!regions.empty() && cns.bases[lq_s].pos >= regions.back().start) {
                    regions.back().start = cns.bases[lq_e].pos;
                } else {
                    LqRegion r;
                    r.end = cns.bases[lq_s].pos;
                    r.start = cns.bases[lq_e].pos;
                    regions.push_back(std::move(r));
                }
                lq = 0;
                lq_s = -1;
            }
            cns.bases[p].base = cov > min_cov ? (char)kIntToBase[cur.base] : (char)tolower(kIntToBase[cur.base]);
            p++;
        }
        cns.len = (unsigned)p;  // uncorrected_len, lstrip, rstrip stay 0 in this variant
        const float rhs = (float)cns.len * (1 - prm.min_error_corrected_ratio);
        if (!(cns.len > 2 && (double)lqseq_total_length < cns.len * 0.8 && 0.0f < rhs)) return false;
        std::reverse(cns.bases.begin(), cns.bases.end());
        return true;
    }

    // ---- helpers of the HiFi candidate phasing (lib/nextcorrect.c:512-737)
    struct Phs { uint16_t d = 0, s = 0; int del = 0; };
    static bool same_seq(const LqSeq &a, const LqSeq &b) { return a.len == b.len && a.seq == b.seq; }

    static int remove_differ_len(LqRegion &lq) {  // :512-539
        int s, j, k = (int)(lq.end - lq.start + 1);
        const int offset = std::min(std::max(30, k / 10), k / 3);
        int8_t dif[kLqCanMax] = {0};
        for (j = s = 0; j < lq.len; j++) {
            if (lq.seqs[j].len + offset >= k && lq.seqs[j].len <= k + offset) s++;
            else dif[j] = 1;
        }
        if (s != lq.len && (s >= lq.len / 2 || (s >= lq.len / 3 && s >= 3))) {
            k = lq.len;
            for (j = 0; j < lq.len && j < k; j++)
                if (dif[j])
                    for (k--; k > j; k--)
                        if (!dif[k]) {
                            std::swap(lq.seqs[j], lq.seqs[k]);
                            break;
                        }
            lq.len = k;
        }
        return s;
    }

    static void compact_flagged(LqRegion &lq, const int8_t *dif) {  // :599-609
        int j, k;
        for (k = lq.len, j = 0; j < lq.len && j < k; j++)
            if (dif[j])
                for (k--; k > j; k--)
                    if (!dif[k]) {
                        std::swap(lq.seqs[j], lq.seqs[k]);
                        break;
                    }
        lq.len = k;
    }

    static void select_most2(LqRegion &lq, int len, int *m1, int *m2) {  // :632-653
        int j, k;
        int8_t used[kLqCanMax] = {0};
        for (*m1 = *m2 = j = 0; j < std::min(lq.len, len); j++) {
            lq.seqs[j].kscore = 1;
            if (used[j]) continue;
            for (k = j + 1; k < lq.len; k++)
                if (same_seq(lq.seqs[j], lq.seqs[k])) {
                    used[k] = 1;
                    lq.seqs[j].kscore++;
                }
            if (lq.seqs[j].kscore > lq.seqs[*m1].kscore ||
                (lq.seqs[j].kscore == lq.seqs[*m1].kscore && lq.seqs[j].order < lq.seqs[*m1].order)) {
                *m2 = *m1;
                *m1 = j;
            } else if (*m2 == *m1 || lq.seqs[j].kscore > lq.seqs[*m2].kscore) {
                *m2 = j;
            }
        }
    }

    static void select_most2_with_kscore(LqRegion &lq, int len, int *m1, int *m2) {  // :656-668
        int j;
        for (*m1 = *m2 = j = 0; j < std::min(lq.len, len); j++) {
            if (lq.seqs[j].kscore > lq.seqs[*m1].kscore ||
                (lq.seqs[j].kscore == lq.seqs[*m1].kscore && lq.seqs[j].order < lq.seqs[*m1].order)) {
                *m2 = *m1;
                *m1 = j;
            } else if (*m2 == *m1 || lq.seqs[j].kscore > lq.seqs[*m2].kscore) {
                *m2 = j;
            }
        }
    }

    static void run_ends(const std::string &q, int *s, int *e) {  // :678-683
        const int n = (int)q.size();
        *s = 0;
        while (*s + 1 < n && q[*s] == q[*s + 1]) (*s)++;
        *e = n - 1;
        while (*e > 0 && q[*e - 1] == q[*e]) (*e)--;
    }
    static int homo_end_same(const std::string &a, const std::string &b) {  // :684-697
        int as, ae, bs, be;
        run_ends(a, &as, &ae);
        run_ends(b, &bs, &be);
        if (ae <= as && be <= bs) return 1;
        if (ae - as != be - bs) return 0;
        for (int i = 0; i <= ae - as; i++)
            if (a[i + as] != b[i + bs]) return 0;
        return 1;
    }
    static int prefixhomo_same(const std::string &a, const std::string &b) {  // :699-714
        int i = 0, j = 0;
        const int na = (int)a.size(), nb = (int)b.size();
        while (i < na && j < nb) {
            if (a[i] != b[j]) return 0;
            while (i + 1 < na && a[i] == a[i + 1]) i++;
            while (j + 1 < nb && b[j] == b[j + 1]) j++;
            i++;
            j++;
        }
        return 1;
    }
    static int trim_endssr_same(const std::string &x, const std::string &y) {  // :716-737
        const std::string *a = &x, *b = &y;
        if (a->size() < b->size()) std::swap(a, b);
        const int na = (int)a->size(), nb = (int)b->size();
        int i;
        for (i = 0; i < nb; i++)
            if ((*a)[i] != (*b)[i]) return 0;
        for (int j = na - 1; j >= i; j--)
            if ((*a)[j] != (*b)[nb - (na - j)]) return 0;
        return 1;
    }

    // HiFi: generate_lqseqs_from_tags_kmer, lib/nextcorrect.c:740-1008, on the extracted candidates
    int lqseqs_from_candidates_kmer() {
        int max_aln_length = 0, max_aln_lqseq_len = 0;
        int s = 0, k = 0, j, index;
        for (size_t ri = 0; ri < regions.size(); ri++) {
            LqRegion &lq = regions[ri];
            RegionReq &rq = extract.regions[ri];
            lq.len = 0;
            lq.seqs.clear();
            lq.has_seed = false;
            for (size_t c = 0; c < rq.cands.size(); c++) {
                LqSeq q;
                q.len = (uint16_t)rq.cands[c].size();
                q.order = rq.cand_rank[c];  // source read
                q.kscore = 0;
                if ((int)q.len > max_aln_lqseq_len) max_aln_lqseq_len = q.len;
                q.seq = std::move(rq.cands[c]);
                lq.seqs.push_back(std::move(q));
                lq.len++;
            }
        }
        std::vector<Phs> phase(n_aligned ? n_aligned : 1);
        int has_heter = 0;
        for (LqRegion &lq : regions) {  // heterozygous sites first (:789-810)
            if (!lq.len) continue;
            select_most2(lq, lq.len, &s, &k);
            if (s != k && lq.seqs[k].kscore >= 3 && lq.seqs[s].len == lq.seqs[k].len) {
                if (s == 0 || k == 0) {
                    const int heter = s == 0 ? k : s;
                    for (j = 0; j < lq.len; j++) {
                        index = lq.seqs[j].order;
                        if (same_seq(lq.seqs[0], lq.seqs[j])) phase[index].s++;
                        else if (same_seq(lq.seqs[heter], lq.seqs[j])) phase[index].d++;
                    }
                }
                lq.indexs = 1;
            } else lq.indexs = 0;
            if (!has_heter && (lq.indexs == 1 ||
                               (s != k && lq.seqs[k].kscore >= 5 &&
                                lq.seqs[s].kscore + lq.seqs[k].kscore >= lq.len * 0.8 &&
                                !prefixhomo_same(lq.seqs[s].seq, lq.seqs[k].seq))))
                has_heter = 1;
        }
        if (has_heter && !phase[0].s) {  // :812-854
            for (LqRegion &lq : regions) {
                if (!lq.len) continue;
                select_most2_with_kscore(lq, lq.len, &s, &k);
                if (s != k && lq.seqs[k].kscore >= 5 && (lq.seqs[s].kscore + lq.seqs[k].kscore) >= lq.len * 0.8 &&
                    (lq.seqs[s].len >= lq.seqs[k].len + 5 || lq.seqs[k].len >= lq.seqs[s].len + 5 ||
                     !prefixhomo_same(lq.seqs[s].seq, lq.seqs[k].seq))) {
                    int s_, k_;
                    if (s == 0) { s_ = 1; k_ = 0; }
                    else if (k == 0) { s_ = 0; k_ = 1; }
                    else {
                        s_ = homo_end_same(lq.seqs[s].seq, lq.seqs[0].seq) || trim_endssr_same(lq.seqs[s].seq, lq.seqs[0].seq) ||
                             prefixhomo_same(lq.seqs[s].seq, lq.seqs[0].seq);
                        k_ = homo_end_same(lq.seqs[k].seq, lq.seqs[0].seq) || trim_endssr_same(lq.seqs[k].seq, lq.seqs[0].seq) ||
                             prefixhomo_same(lq.seqs[k].seq, lq.seqs[0].seq);
                    }
                    int same, heter;
                    if (s_ && !k_) { same = s; heter = k; }
                    else if (k_ && !s_) { same = k; heter = s; }
                    else continue;
                    for (j = 0; j < lq.len; j++) {
                        index = lq.seqs[j].order;
                        if (same_seq(lq.seqs[same], lq.seqs[j])) phase[index].s++;
                        else if (same_seq(lq.seqs[heter], lq.seqs[j])) phase[index].d++;
                    }
                    lq.indexs = 2;
                } else lq.indexs = 0;
            }
        }
        for (LqRegion &lq : regions) {  // mark_del_lqseq, :570-588
            if (!lq.len) continue;
            int kk = 0;
            for (j = 1; j < lq.len; j++) {
                const int i = lq.seqs[j].order;
                if (phase[i].s >= 3 && !phase[i].d) kk++;
            }
            for (j = 0; j < lq.len; j++) {
                const int i = lq.seqs[j].order;
                if (kk >= 2) {
                    if (phase[i].d) phase[i].del = 1;
                } else if (phase[i].s < phase[i].d || phase[i].d >= 3) phase[i].del = 1;
            }
        }
        for (LqRegion &lq : regions) {  // remove_differ_phase_lqseq, :590-610
            if (!lq.len) continue;
            int8_t dif[kLqCanMax] = {0};
            for (j = 0; j < lq.len; j++)
                if (phase[lq.seqs[j].order].del) dif[j] = 1;
            compact_flagged(lq, dif);
        }
        KmerBins bins;
        for (LqRegion &lq : regions) {  // :874-984
            if (!lq.len) continue;
            select_most2(lq, lq.len, &s, &k);
            index = lq.seqs[s].order;
            if (lq.indexs && s != k && s != 0 && lq.seqs[k].kscore >= 3 && phase[index].s >= phase[index].d + 3) {
                int sp = 0, kp = 0;
                for (j = 1; j < lq.len; j++) {
                    index = lq.seqs[j].order;
                    if (phase[index].d >= 3) continue;
                    if (same_seq(lq.seqs[s], lq.seqs[j])) sp += phase[index].s - phase[index].d;
                    else if (same_seq(lq.seqs[k], lq.seqs[j])) kp += phase[index].s - phase[index].d;
                }
                if (sp < kp) s = k;
            } else if (lq.seqs[0].len > 50 && lq.seqs[s].kscore < lq.len / 3 && lq.seqs[s].kscore < 3) {
                const int sl = remove_differ_len(lq);
                if (sl <= 3) {  // large length SD: keep the seed's own version
                    s = 0;
                    lq.seqs[s].kscore = 65534;
                }
            }
            if (lq.seqs[s].kscore > 2 || lq.seqs[s].kscore >= lq.len / 2) {
                lq.sudoseed = lq.seqs[s].seq;
                lq.sudoseed_len = lq.seqs[s].len;
                lq.has_seed = true;
                if (lq.seqs[s].kscore < lq.len / 2)
                    for (char &ch : lq.sudoseed) ch = (char)tolower(ch);
                lq.len = -2;
            } else {
                remove_differ_len(lq);
                if (lq.len > 4) {
                    std::stable_sort(lq.seqs.begin(), lq.seqs.begin() + lq.len,
                                     [](const LqSeq &a, const LqSeq &b) { return a.len < b.len; });  // compare_seq_by_len
                    k = lq.len / 2;
                    while (lq.len > k && (lq.seqs[lq.len - 1].len > 2 * lq.seqs[k].len ||
                                          lq.seqs[lq.len - 1].len >= 1.4 * lq.seqs[lq.len - 2].len))
                        lq.len--;
                    if (k == lq.len) { lq.len = 0; continue; }
                    j = 0;
                    k = lq.len / 2;
                    if (lq.seqs[j].len < lq.seqs[k].len / 2) {
                        std::reverse(lq.seqs.begin(), lq.seqs.begin() + lq.len);
                        while (lq.seqs[lq.len - 1].len < lq.seqs[k].len / 2) lq.len--;
                        if (k == lq.len) { lq.len = 0; continue; }
                    }
                }
                count_kmers(lq, bins, kLqCanMax, 0);
                count_kscore(lq, bins.data(), 0);
                unsigned klastscore, kmaxscore;
                unsigned kmaxlen = lq.seqs[0].len;
                if (kmaxlen > 100) {
                    uint16_t saved[65536 / 256];  // indexed by source-read rank (< aligned reads)
                    std::vector<uint16_t> big;
                    uint16_t *sv = saved;
                    if (n_aligned > sizeof(saved) / sizeof(saved[0])) { big.resize(n_aligned); sv = big.data(); }
                    for (j = 0; j < lq.len; j++) sv[lq.seqs[j].order] = lq.seqs[j].kscore;
                    count_kmers(lq, bins, kLqCanMax, 1);
                    count_kscore(lq, bins.data(), 1);
                    for (j = 0; j < lq.len; j++) lq.seqs[j].kscore = (uint16_t)(lq.seqs[j].kscore + sv[lq.seqs[j].order]);
                }
                sort_by_kscore_desc(lq);
                kmaxlen = lq.seqs[0].len;
                klastscore = kmaxscore = lq.seqs[0].kscore;
                for (k = j = 0; j < lq.len; j++) {
                    const unsigned ks = lq.seqs[j].kscore;
                    if (ks * 10 < kmaxscore || j >= kLqSeqMax || ks * 2 < klastscore) break;
                    klastscore = ks;
                    if (j < kKmerMaxSeq && ks > kmaxscore * 0.8 && lq.seqs[j].len > kmaxlen) {
                        kmaxlen = lq.seqs[j].len;
                        k = j;
                    }
                }
                lq.indexs = 0;
                lq.indexe = (uint8_t)(kmaxlen > (unsigned)kLqRevLen && j > 6 ? 5 : j - 1);
                if ((int)lq.indexe - (int)lq.indexs <= 1 || (lq.seqs[0].len > 20000 && lq.len < kLqCanMax / 3)) {
                    lq.len = 0;
                    continue;
                }
                j = lq.indexs;
                if (lq.seqs[0].len < 3000) k = j + 6 < lq.indexe ? 6 : lq.indexe - j + 1;
                else k = j + 2 < lq.indexe ? 2 : lq.indexe - j + 1;
                if (lq.seqs[0].len < 20000) {
                    std::vector<std::string> in;
                    for (int x = 0; x < k; x++) in.push_back(lq.seqs[j + x].seq);
                    { const uint64_t tp = now_ns(); lq.sudoseed = poa_consensus(in); poa_ns_local += now_ns() - tp; }
                } else lq.sudoseed = lq.seqs[0].seq;
                lq.has_seed = true;
                lq.sudoseed_len = (unsigned)lq.sudoseed.size();
            }
            if (max_aln_lqseq_len + (int)lq.sudoseed_len > max_aln_length) max_aln_length = max_aln_lqseq_len + (int)lq.sudoseed_len;
        }
        return max_aln_length;
    }

    // lib/nextcorrect.c:356-510 with the candidate strings (lines 373-404) already
    // gathered by the backend
    int lqseqs_from_candidates() {
        int max_aln_length = 0;
        KmerBins bins;
        for (size_t ri = 0; ri < regions.size(); ri++) {
            LqRegion &lq = regions[ri];
            RegionReq &rq = extract.regions[ri];
            int max_len_here = 0;
            const int large_seq = (int)rq.n_large;
            lq.len = 0;
            lq.seqs.clear();
            lq.has_seed = false;
            for (std::string &s : rq.cands) {
                LqSeq q;
                q.len = (uint16_t)s.size();
                q.order = (uint16_t)lq.len;
                if ((int)s.size() > max_len_here) max_len_here = (int)s.size();
                q.seq = std::move(s);
                lq.seqs.push_back(std::move(q));
                lq.len++;
            }
            if ((float)large_seq / (lq.len + large_seq) > 1.0 / 3 || lq.len <= 4 || (prm.split && lq.len < 10)) {
                lq.len = 0;
                continue;
            }
            count_kmers(lq, bins, 1, 0);
            count_kscore(lq, bins.data(), 0);
            sort_by_kscore_desc(lq);
            count_kmers(lq, bins, kKmerMaxSeq, 0);
            count_kscore(lq, bins.data(), 0);
            unsigned klastscore, kmaxscore = lq.seqs[0].kscore;
            unsigned kmaxlen = lq.seqs[0].len, kminlen;
            if (kmaxlen > 500 || (kmaxlen > 200 && kmaxscore < 200)) {
                uint16_t saved[kLqCanMax];
                if (lq.seqs[0].order) {
                    for (int j = 1; j < lq.len; j++)
                        if (!lq.seqs[j].order) {
                            std::swap(lq.seqs[0], lq.seqs[j]);
                            break;
                        }
                }
                for (int j = 0; j < lq.len; j++) saved[lq.seqs[j].order] = lq.seqs[j].kscore;
                count_kmers(lq, bins, 1, 1);
                count_kscore(lq, bins.data(), 1);
                sort_by_kscore_desc(lq);
                count_kmers(lq, bins, kKmerMaxSeq, 1);
                count_kscore(lq, bins.data(), 1);
                for (int j = 0; j < lq.len; j++)
                    lq.seqs[j].kscore = (uint16_t)(lq.seqs[j].kscore + saved[lq.seqs[j].order]);
            }
            sort_by_kscore_desc(lq);
            kminlen = kmaxlen = lq.seqs[0].len;
            klastscore = kmaxscore = lq.seqs[0].kscore;
            int j, k;
            for (k = j = 0; j < lq.len; j++) {
                const unsigned ks = lq.seqs[j].kscore;
                if (ks * 10 < kmaxscore || j >= kLqSeqMax || ks * 2 < klastscore ||
                    (j > 4 && kmaxlen > 200 && ks < kmaxscore * 0.6 && lq.seqs[j].len < kminlen * 0.8))
                    break;
                klastscore = ks;
                if (j < kKmerMaxSeq && ks > kmaxscore * 0.8) {
                    if (lq.seqs[j].len > kmaxlen) kmaxlen = lq.seqs[j].len;
                    else if (lq.seqs[j].len < kminlen) kminlen = lq.seqs[j].len;
                }
            }
            lq.indexs = 0;
            lq.indexe = (uint8_t)(kmaxlen > (unsigned)kLqRevLen && j > 6 ? 5 : j - 1);
            if ((int)lq.indexe - (int)lq.indexs <= 3) {
                lq.len = 0;
                continue;
            }
            j = lq.indexs;
            if (lq.seqs[0].len < 3000) k = j + 6 < lq.indexe ? 6 : lq.indexe - j + 1;
            else k = j + 2 < lq.indexe ? 2 : lq.indexe - j + 1;
            {
                std::vector<std::string> in;
                for (int x = 0; x < k; x++) in.push_back(lq.seqs[j + x].seq);
                { const uint64_t tp = now_ns(); lq.sudoseed = poa_consensus(in); poa_ns_local += now_ns() - tp; }
            }
            lq.has_seed = true;
            lq.sudoseed_len = (unsigned)lq.sudoseed.size();
            if (lq.sudoseed_len > 500) {
                int kmax, kmin;
                k = kmax = kmin = lq.seqs[lq.indexs].len;
                for (j = lq.indexs + 1; j <= lq.indexe && j <= lq.indexs + 4; j++) {
                    k += lq.seqs[j].len;
                    if (lq.seqs[j].len > kmax) kmax = lq.seqs[j].len;
                    else if (lq.seqs[j].len < kmin) kmin = lq.seqs[j].len;
                }
                k = kmax != kmin ? (k - kmax - kmin) / (j - lq.indexs - 2) : k / (j - lq.indexs);
                if (lq.sudoseed_len > (unsigned)(k + k / 10)) {
                    for (kmin = k, k = lq.indexs; k < j; k++)
                        if (lq.seqs[k].len != kmax && lq.seqs[k].len >= kmin) break;
                    if (j == k)
                        for (k = 0; k < lq.len && lq.seqs[k].order; k++) {}
                    if (k >= lq.len) k = 0;  // unreachable: order 0 always survives
                    lq.sudoseed = lq.seqs[k].seq;
                    lq.sudoseed_len = lq.seqs[k].len;
                }
            }
            if (max_len_here + (int)lq.sudoseed_len > max_aln_length) max_aln_length = max_len_here + (int)lq.sudoseed_len;
        }
        return max_aln_length;
    }

    // -- low-quality-region rounds (lib/nextcorrect.c:1671-1715, 1538-1669) ----------
    void start_lq_round() {
        lq_iter++;
        lq_max_aln_length += lq_max_dif_len;
        jobs.clear();
        lq_slots.clear();
        lq.pieces.clear();
        lq.lqc.clear();
        lq.ok = false;
        const int count = (int)regions.size();
        for (int i = 0; i < kLqSeqMax; i++) {
            for (int j = count - 1; j >= 0; j--) {
                LqRegion &r = regions[j];
                if (r.len <= 0) continue;
                const int sl = (int)r.sudoseed_len;
                const bool past = i + r.indexs > r.indexe;
                const int ql = past ? sl : r.seqs[i + r.indexs].len;
                LqSlot slot{i, j, -1};
                if (!(past || (i && (ql < sl * 0.5 || ql > sl * 1.3)))) {
                    AlnJob job;
                    job.q = r.seqs[i + r.indexs].seq.c_str();
                    job.q_len = ql;
                    job.t = r.sudoseed.c_str();
                    job.t_len = sl;
                    job.hq = prm.read_type == 3;
                    LqSeq &qs = r.seqs[i + r.indexs];
                    if (qs.packed.empty() && ql > 0) {
                        qs.packed.resize(((size_t)ql + 15) / 16);
                        if (!pack_2bit_lsb(qs.packed.data(), job.q, (size_t)ql)) qs.packed.clear();  // (bytes outside ACGT: packed per call, reported there)
                    }
                    job.q_words = qs.packed.empty() ? nullptr : qs.packed.data();
                    slot.job = (int)jobs.size();
                    jobs.push_back(std::move(job));
                }
                lq_slots.push_back(slot);
                lq.pieces.push_back(LqRound::Piece{slot.job, (unsigned)sl});
            }
        }
        lq.jobs = &jobs;
        lq.n_regions = (unsigned)(lq.pieces.size() / kLqSeqMax);
        lq.factor = prm.read_type == 3 ? 4 : 2;
        lq.qv_factor = prm.read_type == 3 ? 2 : 5;
        phase = PileEngine::LQ_ROUND;
        if (jobs.empty()) after_lq_round();  // nothing to align: still run the round
    }

    void after_lq_round() {
        if (lq.ok) {  // the device ran the round (K12): its walk is the string the host path builds below
            jobs.clear();
            lq_slots.clear();
            std::string lqc;
            lqc.swap(lq.lqc);
            lq.ok = false;
            rewrite_pseudo_seeds(lqc);
            return;
        }
        // generate_consensus_trimed: build the 30 linked rows, second MSA, backtrack
        const int count = (int)regions.size();
        size_t link_len = 1;
        for (int j = count - 1; j >= 0; j--)
            if (regions[j].len > 0) link_len += regions[j].sudoseed_len + 1;
        Msa msa(link_len);
        std::vector<TagList> rows(kLqSeqMax);
        size_t slot_at = 0;
        std::string t_str, q_str;
        for (int i = 0; i < kLqSeqMax; i++) {
            t_str.clear();
            q_str.clear();
            for (int j = count - 1; j >= 0; j--) {
                LqRegion &r = regions[j];
                if (r.len <= 0) continue;
                const LqSlot &slot = lq_slots[slot_at++];
                const int sl = (int)r.sudoseed_len;
                t_str.push_back('N');
                q_str.push_back('N');
                const AlnJob *job = slot.job >= 0 ? &jobs[slot.job] : nullptr;
                if (job && job->status == ALN_OK && job->ops.size() > 2) {
                    const char *q = job->q, *t = job->t;
                    int qi = 0, ti = 0;
                    for (uint8_t op : job->ops) {
                        if (op == OP_MATCH) { t_str.push_back(t[ti++]); q_str.push_back(q[qi++]); }
                        else if (op == OP_QONLY) { t_str.push_back('-'); q_str.push_back(q[qi++]); }
                        else { t_str.push_back(t[ti++]); q_str.push_back('-'); }
                    }
                    int tl = job->t_used, ql = job->q_used;
                    while (tl < sl) { t_str.push_back(r.sudoseed[tl++]); q_str.push_back('-'); }
                    int delta = 0;
                    const LqSeq &qs = r.seqs[slot.i + r.indexs];
                    while (ql < qs.len && delta++ < 250) { q_str.push_back(qs.seq[ql++]); t_str.push_back('-'); }
                } else {
                    t_str.append((size_t)sl, 'M');
                    q_str.append((size_t)sl, 'M');
                }
            }
            t_str.push_back('N');
            q_str.push_back('N');
            tags_from_strings(t_str, q_str, 0, rows[i], msa);
        }
        jobs.clear();
        lq_slots.clear();

        // get_lqseqs_from_align_tags (nextcorrect.c:1250-1338)
        msa.allocate();
        msa.count_links(rows);
        rows.clear();
        Pos cur = msa.score_lq(prm.read_type == 3 ? 4 : 2);
        const int min_qv_factor = prm.read_type == 3 ? 2 : 5;
        std::string lqc;
        while (true) {
            if (cur.b != 4) {
                const Cell &c = msa.cell(cur.t, cur.d, cur.b);
                const char ch = (char)kIntToBase[cur.b];
                lqc.push_back((int)c.best_link * min_qv_factor > (int)msa.coverage[cur.t] || ch == 'N' ? ch : (char)tolower(ch));
            }
            const Cell &c = msa.cell(cur.t, cur.d, cur.b);
            cur = Pos{c.best_t, c.best_d, c.best_b};
            if (cur.t == -1) break;
        }
        rewrite_pseudo_seeds(lqc);
    }

    // iterate_generate_consensus_trimed body (nextcorrect.c:1684-1710): the walk's characters (origin first) become the
    // regions' new pseudo-seeds; then the second round, or the splice
    void rewrite_pseudo_seeds(const std::string &lqc) {
        const int count = (int)regions.size();
        int j = count;
        int psed_len = 0;
        lq_max_dif_len = 0;
        for (size_t k = lqc.size(); k; k--) {
            const char ch = lqc[k - 1];
            if (ch != 'N') {
                if (j < 0 || j >= count) { finish_error(2); return; }  // reference: out-of-bounds write
                LqRegion &r = regions[j];
                if (r.sudoseed.size() <= r.sudoseed_len) r.sudoseed.resize(r.sudoseed_len + 1);
                if (ch < 'a') r.sudoseed[r.sudoseed_len++] = ch;
                else {
                    r.sudoseed[r.sudoseed_len++] = (char)toupper(ch);
                    r.lqcount++;
                }
            } else {
                if (j != count && j >= 0) {
                    LqRegion &r = regions[j];
                    r.sudoseed.resize(r.sudoseed_len);
                    if ((int)r.sudoseed_len > psed_len + lq_max_dif_len) lq_max_dif_len = (int)r.sudoseed_len - psed_len;
                    if (r.lqcount > r.sudoseed_len * 4 / 5) r.len = -1;
                }
                j--;
                while (j >= 0 && regions[j].len <= 0) j--;
                if (j < 0) continue;
                psed_len = (int)regions[j].sudoseed_len;
                regions[j].sudoseed_len = regions[j].lqcount = 0;
            }
        }
        if (lq_iter < 2) {
            start_lq_round();
            return;
        }
        splice();
    }

    // update_consensus_trimed (lib/nextcorrect.c:1365-1482), non-HiFi branch, then
    // trim_terminal_ssr (nextcorrect.c:2214)
    void splice() {
        std::string out;
        out.reserve((size_t)seed_len * 2 + 1);
        unsigned p = 0, i = cns.lstrip;
        int update = 1;
        int idx = (int)regions.size() - 1;
        LqReg lq[kLqRegMax + 1];  // +1 guard slot (the reference indexes lq[10] in one corner case)
        unsigned lq_m = 0, hq_m = 0;
        int lq_i = 0;
        auto usable = [&](int k) { return regions[k].len > 0 || regions[k].len == -2; };
        bool stop = false;
        while (!stop && i < cns.len - cns.rstrip) {
            p = cns.bases[i].pos;
            if (idx >= 0 && (!usable(idx) || p > regions[idx].end)) {
                idx--;
                update = 1;
            }
            if (idx >= 0 && usable(idx) && p >= regions[idx].start && p <= regions[idx].end) {
                if (update) {
                    const LqRegion &r = regions[idx];
                    for (unsigned j = 0; j < r.sudoseed_len; j++) {
                        out.push_back(r.sudoseed[j]);
                        lq_i = update_lqreg(lq, out, (unsigned)out.size() - 1, lq_i, &hq_m, &lq_m);
                        if (lq_i >= kLqRegMax) { stop = true; break; }
                    }
                    update = 0;
                }
            } else {
                out.push_back(cns.bases[i].base);
                update = 1;
                lq_i = update_lqreg(lq, out, (unsigned)out.size() - 1, lq_i, &hq_m, &lq_m);
                if (lq_i >= kLqRegMax) break;
            }
            i++;
        }
        unsigned olen = (unsigned)out.size();
        if (lq_i < kLqRegMax + 1 && lq[lq_i].end == olen - 1) lq_i++;

        if (prm.read_type == 3) {
            unsigned a = 0, z = 0, lq_total_len = 0;
            while (a < olen && out[a] >= 'a') a++;
            while (z < olen && out[olen - 1 - z] >= 'a') z++;
            if (a + z < olen) {
                if (a > 0 || z > 0) out = out.substr(a, olen - a - z);
                const int removed = (int)(a + z);
                for (int q = 0; q < lq_i && q < kLqRegMax + 1; q++) lq_total_len += lq[q].lq_total_len;
                result.len = (unsigned)out.size();
                result.seq = out;
                result.identity = 1 - (float)(lq_total_len - (unsigned)removed) / result.len;
            } else {
                result.len = 2;
                result.identity = 0;
                result.seq.clear();
            }
            if (result.len > 1000 && result.identity > 0.8) trim_terminal_ssr(result);
            phase = PileEngine::DONE;
            return;
        }
        if (lq_i) {
            lq_m = 0;
            hq_m = lq[0].start;
            p = lq[0].start;
            unsigned lq_total_len = lq[0].lq_total_len - lq[0].lqlen;
            for (i = 1; i < (unsigned)kLqRegMax && lq[i].end; i++) {
                if (lq[i].start - lq[i - 1].end > p) {
                    lq_m = lq[i - 1].end + 1;
                    hq_m = lq[i].start;
                    lq_total_len = lq[i].lq_total_len - lq[i].lqlen;
                    p = lq[i].start - lq[i - 1].end;
                }
            }
            if (i < (unsigned)kLqRegMax && olen - lq[i - 1].end > p) {
                lq_m = lq[i - 1].end + 1;
                hq_m = olen;
                lq_total_len = lq[i].lq_total_len;
            }
            result.len = hq_m - lq_m;
            result.seq = lq_m <= out.size() ? out.substr(lq_m, result.len) : std::string();
            result.identity = 1 - (float)lq_total_len / result.len;
        } else {
            if (!out.empty() && out[0] >= 'a') {
                unsigned k = 0;
                while (k < out.size() && out[k] >= 'a') k++;
                out.erase(0, k);
                lq[0].lq_total_len -= k;
            }
            result.len = (unsigned)out.size();
            result.seq = out;
            result.identity = 1 - (float)lq[0].lq_total_len / result.len;
        }
        if (result.len > 1000 && result.identity > 0.8) trim_terminal_ssr(result);
        phase = PileEngine::DONE;
    }

    ConsensusTrimed *take() {
        if (error2 || result.len <= 4) {
            ConsensusTrimed *c = make_error_seed(result.len);
            c->identity = result.identity;
            if (!error2 && result.len == 4) memcpy(c->seq, result.seq.data(), std::min<size_t>(4, result.seq.size()));
            return c;
        }
        ConsensusTrimed *c = (ConsensusTrimed *)malloc(sizeof(ConsensusTrimed));
        c->len = result.len;
        c->identity = result.identity;
        c->seq = (char *)malloc((size_t)result.len + 1);
        memcpy(c->seq, result.seq.data(), result.len);
        c->seq[result.len] = '\0';
        return c;
    }
};

PileEngine::PileEngine(const char *const *seqs, const unsigned *aln_start, const unsigned *aln_end, unsigned seq_count,
                       const CorrectParams &prm)
    : impl_(new PileImpl(seqs, nullptr, nullptr, aln_start, aln_end, seq_count, prm, nullptr)) {}
PileEngine::PileEngine(const unsigned *seq_len, const int64_t *dev_off, const unsigned *aln_start,
                       const unsigned *aln_end, unsigned seq_count, const CorrectParams &prm, const char *seed)
    : impl_(new PileImpl(nullptr, seq_len, dev_off, aln_start, aln_end, seq_count, prm, seed)) {}
PileEngine::~PileEngine() { delete impl_; }
PileEngine::Phase PileEngine::phase() const { return impl_->phase; }
MainPile *PileEngine::main_request() { return &impl_->main; }
ExtractPile *PileEngine::extract_request() { return &impl_->extract; }
void PileEngine::collect_jobs(std::vector<AlnJob *> &out) { impl_->collect(out); }
LqRound *PileEngine::lq_request() { return impl_->phase == LQ_ROUND && !impl_->jobs.empty() ? &impl_->lq : nullptr; }
void PileEngine::advance() { impl_->advance(); }
ConsensusTrimed *PileEngine::take_result() { return impl_->take(); }

namespace {
template <typename F>
void parallel_for(size_t n, int threads, F f) {
    if (threads <= 1 || n <= 1) {
        for (size_t i = 0; i < n; i++) f(i);
        return;
    }
    std::atomic<size_t> next(0);
    std::vector<std::thread> pool;
    const int nt = (int)std::min<size_t>((size_t)threads, n);
    for (int t = 0; t < nt; t++)
        pool.emplace_back([&] {
            for (;;) {
                size_t i = next.fetch_add(1);
                if (i >= n) break;
                f(i);
            }
        });
    for (auto &th : pool) th.join();
}
}  // namespace

HostProf g_prof;

bool pack_2bit_lsb(uint32_t *out, const char *s, size_t n) {
    static const struct Lut {
        uint8_t v[256];
        Lut() {
            memset(v, 0xff, sizeof(v));
            v['A'] = 0, v['C'] = 1, v['G'] = 2, v['T'] = 3;  // lib/bseq.c:11-20
        }
    } lut;
    unsigned bad = 0;
    size_t i = 0, w = 0;
    for (; i + 16 <= n; w++, i += 16) {
        uint32_t acc = 0;
        for (int b = 0; b < 16; b++) {
            const uint8_t c = lut.v[(unsigned char)s[i + b]];
            bad |= c;
            acc |= (uint32_t)(c & 3u) << (2 * b);
        }
        out[w] = acc;
    }
    if (i < n) {
        uint32_t acc = 0;
        for (int b = 0; i + b < n; b++) {
            const uint8_t c = lut.v[(unsigned char)s[i + b]];
            bad |= c;
            acc |= (uint32_t)(c & 3u) << (2 * b);
        }
        out[w] = acc;
    }
    return (bad & 0x80u) == 0;
}

int effective_cpus() {
    static const int n = [] {
        int hw = (int)std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) {
            const int a = CPU_COUNT(&set);
            if (a > 0 && a < hw) hw = a;
        }
        auto from_quota = [&](double quota, double period) {
            if (quota > 0 && period > 0) {
                const int q = (int)std::ceil(quota / period - 1e-9);
                if (q >= 1 && q < hw) hw = q;
            }
        };
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "max 100000" or "1600000 100000"
            char a[64] = {0};
            double period = 0;
            if (fscanf(f, "%63s %lf", a, &period) == 2 && strcmp(a, "max") != 0) from_quota(atof(a), period);
            fclose(f);
        } else {
            double quota = -1, period = 0;
            if (FILE *q = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (fscanf(q, "%lf", &quota) != 1) quota = -1;
                fclose(q);
            }
            if (FILE *q = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(q, "%lf", &period) != 1) period = 0;
                fclose(q);
            }
            from_quota(quota, period);
        }
        if (const char *e = getenv("NDGPU_HOST_CPUS")) hw = std::max(1, atoi(e));  // (override)
        return hw;
    }();
    return n;
}

static std::atomic<int> g_cores_total(0), g_cores_used(0);
void CoreGovernor::set_total(int total) { g_cores_total.store(total < 0 ? 0 : total); }
int CoreGovernor::acquire(int base) {
    if (base < 1) base = 1;
    const int total = g_cores_total.load();
    int used = g_cores_used.load();
    for (;;) {
        int grant = total - used;
        if (grant < base) grant = base;
        static const int cap_mul = getenv("NDGPU_GOV_CAP") ? std::max(1, atoi(getenv("NDGPU_GOV_CAP"))) : 4;
        if (grant > cap_mul * base) grant = cap_mul * base;  // thread start-up and allocator contention eat the gain beyond this
        if (g_cores_used.compare_exchange_weak(used, used + grant)) return grant;
    }
}
void CoreGovernor::release(int granted) { g_cores_used.fetch_sub(granted); }

static inline uint64_t now_ns() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
               std::chrono::steady_clock::now().time_since_epoch())
        .count();
}

void run_engines(PileEngine **eng, size_t n, Backend &be, int threads) {
    std::vector<MainPile *> mains;
    std::vector<ExtractPile *> extracts;
    std::vector<AlnJob *> jobs;
    std::vector<LqRound *> lqs;
    std::vector<size_t> live, in_lq;
    for (;;) {
        mains.clear();
        extracts.clear();
        jobs.clear();
        lqs.clear();
        live.clear();
        in_lq.clear();
        for (size_t i = 0; i < n; i++) {
            switch (eng[i]->phase()) {
                case PileEngine::MAIN: mains.push_back(eng[i]->main_request()); break;
                case PileEngine::EXTRACT: extracts.push_back(eng[i]->extract_request()); break;
                case PileEngine::LQ_ROUND:
                    in_lq.push_back(i);
                    if (LqRound *r = eng[i]->lq_request()) lqs.push_back(r);
                    break;
                default: continue;
            }
            live.push_back(i);
        }
        if (live.empty()) break;
        uint64_t t0 = now_ns();
        if (!mains.empty()) be.run_main(mains.data(), mains.size());
        uint64_t t1 = now_ns();
        if (!extracts.empty()) be.run_extract(extracts.data(), extracts.size());
        uint64_t t2 = now_ns();
        // low-quality-region rounds: whole rounds on the backend where it offers that; what it does not take (or declines)
        // goes the host way -- the alignments as a batch, the second MSA in the engine
        if (!lqs.empty()) (void)be.run_lq(lqs.data(), lqs.size());
        for (size_t i : in_lq) eng[i]->collect_jobs(jobs);
        if (!jobs.empty()) be.run_align(jobs.data(), jobs.size());
        uint64_t t3 = now_ns();
        {
            CoreLease lease(threads);
            parallel_for(live.size(), lease.n, [&](size_t k) { eng[live[k]]->advance(); });
        }
        uint64_t t4 = now_ns();
        g_prof.main_ns += t1 - t0;
        g_prof.extract_ns += t2 - t1;
        g_prof.align_ns += t3 - t2;
        g_prof.advance_ns += t4 - t3;
        g_prof.jobs += jobs.size();
        static const bool trace = getenv("NDGPU_TRACE") != nullptr;
        if (trace)  // one line per round of a sub-batch: which phases ran, on how many piles / jobs, and when (ms)
            fprintf(stderr, "[ndgpu trace] be %p piles %zu | main %zu [%.1f %.1f] extract %zu [%.1f] align %zu [%.1f] advance [%.1f]\n",
                    (void *)&be, n, mains.size(), t0 * 1e-6, t1 * 1e-6, extracts.size(), t2 * 1e-6, jobs.size(), t3 * 1e-6, t4 * 1e-6);
    }
    be.end_batch();
}

}  // namespace ndgpu
