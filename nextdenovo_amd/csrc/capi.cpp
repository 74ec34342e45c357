// C ABI of the engine (include/ndgpu_nextcorrect.h).
#include <hip/hip_runtime_api.h>
#include <cerrno>
#include <sys/uio.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ndgpu_nextcorrect.h"
#include "nd_host.h"
#include "nd_runtime.h"

using namespace ndgpu;

namespace {

CorrectParams make_params(unsigned max_mem_len, unsigned min_len_aln, unsigned max_cov_aln, unsigned min_cov,
                          unsigned lqseq_max_length, float ratio, unsigned split, unsigned fast, int read_type) {
    CorrectParams p;
    p.max_mem_len = max_mem_len;
    p.min_len_aln = min_len_aln;
    p.max_cov_aln = max_cov_aln;
    p.min_cov = min_cov;
    p.lqseq_max_length = lqseq_max_length;
    p.min_error_corrected_ratio = ratio;
    p.split = split;
    p.fast = fast;
    p.read_type = read_type;
    return p;
}

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

void run_single_alignment(char *q, int q_len, char *t, int t_len, alignment *a, int hq) {
    AlnJob job;
    job.q = q;
    job.q_len = q_len;
    job.t = t;
    job.t_len = t_len;
    job.hq = hq;
    AlnJob *jp = &job;
    {
        HipBackend be;
        be.run_align(&jp, 1);
    }
    if (job.status == ALN_NONE) return;  // reference leaves *align_rtn untouched (lib/align.c:440-560)
    a->aln_t_e = a->aln_t_s + (unsigned)job.t_used - 1;
    a->aln_t_len = (unsigned)job.t_used;
    a->aln_q_len = (unsigned)job.q_used;
    size_t n = job.ops.size();
    int qi, ti;
    if (job.status == ALN_GAP_ABORT) {
        // reference: aln_len = 2 with the last two alignment columns (lib/align.c:542-556)
        qi = job.q_used;
        ti = job.t_used;
        for (size_t c = n; c-- > 0;) {
            if (job.ops[c] != OP_TONLY) qi--;
            if (job.ops[c] != OP_QONLY) ti--;
        }
    } else {
        qi = ti = 0;
    }
    for (size_t c = 0; c < n; c++) {
        const uint8_t op = job.ops[c];
        a->t_aln_str[c] = op == OP_QONLY ? '-' : t[ti++];
        a->q_aln_str[c] = op == OP_TONLY ? '-' : q[qi++];
    }
    a->t_aln_str[n] = a->q_aln_str[n] = '\0';
    a->aln_len = (unsigned)n;
}

}  // namespace

namespace {
std::mutex g_reap_mu;
std::vector<std::thread> g_reapers;
void join_reapers() {
    std::vector<std::thread> v;
    {
        std::lock_guard<std::mutex> lock(g_reap_mu);
        v.swap(g_reapers);
    }
    for (auto &t : v) t.join();
}
struct ReaperAtExit {
    ~ReaperAtExit() { join_reapers(); }
} g_reaper_at_exit;
}  // namespace

extern "C" {

consensus_trimed *nextCorrect(char **seqs, unsigned int *aln_start, unsigned int *aln_end, unsigned int seq_count,
                              unsigned int max_mem_len, unsigned int min_len_aln, unsigned int max_cov_aln,
                              unsigned int min_cov, unsigned int lqseq_max_length, float min_error_corrected_ratio,
                              unsigned int split, unsigned int fast, int read_type) {
    PileEngine eng(seqs, aln_start, aln_end, seq_count,
                   make_params(max_mem_len, min_len_aln, max_cov_aln, min_cov, lqseq_max_length,
                               min_error_corrected_ratio, split, fast, read_type));
    PileEngine *ep = &eng;
    try {
        HipBackend be;
        run_engines(&ep, 1, be, 1);
    } catch (const DeviceOom &) {  // lib/nextcorrect.c:2254-2261: a seed whose working memory cannot be had is reported, not fatal
        DeviceAligner::context(0).release_memory();
        DeviceAligner::forget_sizes();
        return (consensus_trimed *)make_error_seed(3);
    }
    return (consensus_trimed *)eng.take_result();
}

void free_consensus_trimed(consensus_trimed *c) {
    if (!c) return;
    free(c->seq);
    free(c);
}

// lib/nextcorrect.py:236-260 over the records of one hand-over: headers and index lines are formatted into two small buffers, the
// bases go to the file from where the library holds them (writev: no copy of the 70 MB a config-2 step prints)
int ndgpu_write_records(consensus_trimed **recs, const uint32_t *ids, int n, const uint32_t *names, uint32_t min_len_seed,
                        double min_ratio, int fd_out, int fd_idx, uint64_t *pos, uint32_t *lens, float *identities) {
    std::vector<char> heads;
    std::string idx;
    struct Piece { size_t head_at, head_len; const char *seq; size_t len; };
    std::vector<Piece> pieces;
    heads.reserve((size_t)n * 40);
    uint64_t at = *pos;
    char line[96];
    for (int k = 0; k < n; k++) {
        const uint32_t i = ids[k];
        consensus_trimed *c = recs[i];
        if (!c) continue;
        const uint32_t ln = c->len;
        const float ide = c->identity;
        if (lens) lens[i] = ln;
        if (identities) identities[i] = ide;
        if (ln >= min_len_seed && ln > 4 && (double)ide >= min_ratio) {
            const int hl = snprintf(line, sizeof(line), ">%u %u %f\n", names[i], ln, (double)ide);
            pieces.push_back(Piece{heads.size(), (size_t)hl, c->seq, (size_t)ln});
            heads.insert(heads.end(), line, line + hl);
            at += (uint64_t)hl + ln + 1;
            if (fd_idx >= 0) {
                const int il = snprintf(line, sizeof(line), "%u\t%llu\t%u\n", names[i], (unsigned long long)(at - ln - 1), ln);
                idx.append(line, (size_t)il);
            }
        } else if (ln != 3 && fd_idx >= 0) {
            const int il = snprintf(line, sizeof(line), "%u\t0\t0\n", names[i]);
            idx.append(line, (size_t)il);
        }
    }
    auto write_all = [](int fd, struct iovec *iov, int cnt) {
        while (cnt > 0) {
            ssize_t w = writev(fd, iov, cnt > 1024 ? 1024 : cnt);
            if (w < 0 && errno == EINTR) continue;   // (a signal, not a failure)
            if (w < 0) return false;
            while (cnt > 0 && (size_t)w >= iov->iov_len) w -= (ssize_t)iov->iov_len, ++iov, --cnt;
            if (cnt > 0 && w > 0) iov->iov_base = (char *)iov->iov_base + w, iov->iov_len -= (size_t)w;
        }
        return true;
    };
    bool ok = true;
    static const char nl = '\n';
    std::vector<struct iovec> iov;
    iov.reserve(pieces.size() * 3);
    for (const Piece &p : pieces) {
        iov.push_back({heads.data() + p.head_at, p.head_len});
        iov.push_back({(void *)p.seq, p.len});
        iov.push_back({(void *)&nl, 1});
    }
    const off_t idx0 = fd_idx >= 0 ? lseek(fd_idx, 0, SEEK_CUR) : (off_t)-1;   // (where the index stood before this hand-over)
    if (!iov.empty()) ok = write_all(fd_out, iov.data(), (int)iov.size());
    if (ok && fd_idx >= 0 && !idx.empty()) {
        struct iovec one{(void *)idx.data(), idx.size()};
        ok = write_all(fd_idx, &one, 1);
    }
    for (int k = 0; k < n; k++) {   // (after the write: the bases were the library's until here)
        free_consensus_trimed(recs[ids[k]]);
        recs[ids[k]] = nullptr;
    }
    if (ok) *pos = at;
    else {
        // part of the hand-over may be on disk and its index lines not: cut both files back to where they stood before this call, so
        // that what a resumed run reads (lib/nextcorrect.py:205-226: the .idx decides where the .fasta is cut) belongs together
        if (ftruncate(fd_out, (off_t)*pos) == 0) (void)lseek(fd_out, (off_t)*pos, SEEK_SET);
        if (idx0 >= 0 && ftruncate(fd_idx, idx0) == 0) (void)lseek(fd_idx, idx0, SEEK_SET);
    }
    return ok ? 0 : -1;
}

int ndgpu_correct_batch(int n_piles, char ***seqs, unsigned int **aln_start, unsigned int **aln_end,
                        const unsigned int *seq_count, const unsigned int *max_mem_len,
                        const unsigned int *lqseq_max_length, unsigned int min_len_aln, unsigned int max_cov_aln,
                        unsigned int min_cov, float min_error_corrected_ratio, unsigned int split, unsigned int fast,
                        int read_type, int host_threads, consensus_trimed **out) {
    if (n_piles <= 0) return 0;
    // (what the process can have, not what the machine has: a cgroup quota of 16 CPUs on a 256-thread host; asking for more only
    // gets the whole process throttled)
    if (host_threads <= 0 || (host_threads > effective_cpus() && !getenv("NDGPU_NO_CPU_CAP"))) host_threads = effective_cpus();
    std::vector<PileEngine *> eng((size_t)n_piles, nullptr);
    parallel_for((size_t)n_piles, host_threads, [&](size_t i) {
        eng[i] = new PileEngine(seqs[i], aln_start[i], aln_end[i], seq_count[i],
                                make_params(max_mem_len[i], min_len_aln, max_cov_aln, min_cov, lqseq_max_length[i],
                                            min_error_corrected_ratio, split, fast, read_type));
    });
    // (out of device memory: the batch is halved until it fits; a single pile that does not fit is an out-of-memory seed)
    std::vector<std::pair<size_t, size_t>> todo{{0, (size_t)n_piles}};
    while (!todo.empty()) {
        const auto [a, b] = todo.back();
        todo.pop_back();
        try {
            HipBackend be(0, host_threads);
            run_engines(eng.data() + a, b - a, be, host_threads);
            for (size_t i = a; i < b; i++) out[i] = (consensus_trimed *)eng[i]->take_result();
        } catch (const DeviceOom &) {
            DeviceAligner::context(0).release_memory();
            DeviceAligner::forget_sizes();
            for (size_t i = a; i < b; i++) {  // engines restart from scratch
                delete eng[i];
                eng[i] = new PileEngine(seqs[i], aln_start[i], aln_end[i], seq_count[i],
                                        make_params(max_mem_len[i], min_len_aln, max_cov_aln, min_cov, lqseq_max_length[i],
                                                    min_error_corrected_ratio, split, fast, read_type));
            }
            if (b - a == 1) out[a] = (consensus_trimed *)make_error_seed(3);
            else {
                todo.push_back({a + (b - a) / 2, b});
                todo.push_back({a, a + (b - a) / 2});
            }
        }
    }
    for (size_t i = 0; i < (size_t)n_piles; i++) delete eng[i];
    return 0;
}

struct ndgpu_db {
    ReadDb *db;
    uint32_t *dev_pool;  // this handle's copy in HBM (forward + reverse complement), handed to every batch that names it
};

ndgpu_db *ndgpu_db_create(uint32_t n_reads, const uint32_t *words, const uint64_t *word_off, const uint32_t *len) {
    ndgpu_db *h = new ndgpu_db;
    h->db = new ReadDb(n_reads, words, word_off, len);
    h->dev_pool = DeviceAligner::upload_db(h->db->pool().data(), h->db->pool().size(), DeviceAligner::instance().device());
    if (!h->dev_pool) {
        fprintf(stderr, "[ndgpu] ndgpu_db_create: out of device memory for a read DB of %llu bases\n",
                (unsigned long long)h->db->total_bases());
        delete h->db;
        delete h;
        return nullptr;
    }
    return h;
}

// init_ovls()'s view of the read DB (lib/ovlseq.c:50-138, lib/index.c:7-36): `idx_fofn` lists one `.idx` per line, the
// `.2bit` of `/dir/.NAME.idx` is `/dir/NAME.2bit` (lib/ovlseq.c:24-37); a `.2bit` holds {0, 254}, then per read u32 id,
// u32 length, ceil(length / 16) words (lib/bseq.c:93-139).  Read ids index the DB.  NULL on any I/O or format error.
ndgpu_db *ndgpu_db_open(const char *idx_fofn) {
    FILE *f = fopen(idx_fofn, "r");
    if (!f) {
        fprintf(stderr, "[ndgpu] ndgpu_db_open: cannot open %s\n", idx_fofn);
        return nullptr;
    }
    std::vector<std::string> idx_files;
    char line[4096];
    while (fgets(line, sizeof(line), f)) {
        std::string s(line);
        while (!s.empty() && (s.back() == '\n' || s.back() == '\r' || s.back() == ' ')) s.pop_back();
        if (!s.empty() && s[0] != '#') idx_files.push_back(s);
    }
    fclose(f);
    std::vector<uint32_t> words, lens;
    std::vector<uint64_t> word_off;
    // ids index the DB: bound them by the number of lines of the .idx files (one per read, lib/index.c:7-36) -- a corrupt id
    // must not size the tables
    size_t max_reads = 0;
    for (const std::string &idx : idx_files) {
        FILE *t = fopen(idx.c_str(), "r");
        if (!t) {
            fprintf(stderr, "[ndgpu] ndgpu_db_open: cannot open %s\n", idx.c_str());
            return nullptr;
        }
        char buf[65536];
        size_t got;
        while ((got = fread(buf, 1, sizeof(buf), t)) > 0)
            for (size_t i = 0; i < got; i++) max_reads += buf[i] == '\n';
        fclose(t);
        max_reads++;  // a last line without a newline
    }
    for (const std::string &idx : idx_files) {
        const size_t slash = idx.find_last_of('/');
        const std::string dir = slash == std::string::npos ? "" : idx.substr(0, slash + 1);
        std::string name = slash == std::string::npos ? idx : idx.substr(slash + 1);
        if (name.size() < 6 || name[0] != '.' || name.substr(name.size() - 4) != ".idx") {
            fprintf(stderr, "[ndgpu] ndgpu_db_open: %s is not a .NAME.idx path\n", idx.c_str());
            return nullptr;
        }
        const std::string path = dir + name.substr(1, name.size() - 5) + ".2bit";
        FILE *b = fopen(path.c_str(), "rb");
        if (!b) {
            fprintf(stderr, "[ndgpu] ndgpu_db_open: cannot open %s\n", path.c_str());
            return nullptr;
        }
        fseek(b, 0, SEEK_END);
        const long sz = ftell(b);
        fseek(b, 2, SEEK_SET);
        const size_t nw = sz > 2 ? (size_t)(sz - 2) / 4 : 0;
        const size_t base = words.size();
        words.resize(base + nw);
        if (nw && fread(words.data() + base, 4, nw, b) != nw) {
            fclose(b);
            fprintf(stderr, "[ndgpu] ndgpu_db_open: short read on %s\n", path.c_str());
            return nullptr;
        }
        fclose(b);
        for (size_t p = base; p + 2 <= base + nw;) {
            const uint32_t id = words[p], ln = words[p + 1];
            if (p + 2 + (((size_t)ln + 15) >> 4) > base + nw || (size_t)id >= max_reads) {  // truncated / corrupt file
                fprintf(stderr, "[ndgpu] ndgpu_db_open: %s: record at word %zu (id %u, %u bases) %s\n", path.c_str(), p - base, id, ln,
                        (size_t)id >= max_reads ? "names a read the .idx files do not list" : "runs past the end of the file");
                return nullptr;
            }
            if (id >= lens.size()) {
                lens.resize((size_t)id + 1, 0);
                word_off.resize((size_t)id + 1, 0);
            }
            lens[id] = ln;
            word_off[id] = p + 2;
            p += 2 + (((size_t)ln + 15) >> 4);
        }
    }
    return ndgpu_db_create((uint32_t)lens.size(), words.data(), word_off.data(), lens.data());
}

void ndgpu_db_destroy(ndgpu_db *h) {
    if (!h) return;
    join_reapers();
    DeviceAligner::free_db(h->dev_pool);
    delete h->db;
    delete h;
}


int ndgpu_correct_piles(ndgpu_db *h, int n_piles, const uint32_t *recs, const uint64_t *pile_off,
                        unsigned int min_len_aln, unsigned int max_cov_aln, unsigned int min_cov,
                        unsigned int max_lq_length, float min_error_corrected_ratio, unsigned int split,
                        unsigned int fast, int read_type, int host_threads, consensus_trimed **out) {
    return ndgpu_correct_piles_stream(h, n_piles, recs, pile_off, min_len_aln, max_cov_aln, min_cov, max_lq_length, min_error_corrected_ratio,
                                      split, fast, read_type, host_threads, out, nullptr, nullptr);
}

int ndgpu_correct_piles_stream(ndgpu_db *h, int n_piles, const uint32_t *recs, const uint64_t *pile_off,
                               unsigned int min_len_aln, unsigned int max_cov_aln, unsigned int min_cov,
                               unsigned int max_lq_length, float min_error_corrected_ratio, unsigned int split,
                               unsigned int fast, int read_type, int host_threads, consensus_trimed **out,
                               ndgpu_piles_done_fn done, void *user) {
    if (n_piles <= 0) return 0;
    if (!h || !h->db || !h->dev_pool) return -1;
    // (what the process can have, not what the machine has: a cgroup quota of 16 CPUs on a 256-thread host; asking for more only
    // gets the whole process throttled)
    if (host_threads <= 0 || (host_threads > effective_cpus() && !getenv("NDGPU_NO_CPU_CAP"))) host_threads = effective_cpus();
    const ReadDb &db = *h->db;
    // every record must name reads of this DB and windows inside them (a sorted.ovl written against other .idx files
    // would otherwise index the host tables and the device pool out of bounds): -2, nothing is computed
    for (int i = 0; i < n_piles; i++) {
        if (pile_off[i + 1] < pile_off[i]) return -2;
        for (uint64_t r = pile_off[i]; r < pile_off[i + 1]; r++) {
            const uint32_t *c = recs + r * 8;
            if (c[0] >= db.n_reads() || c[4] >= db.n_reads() || c[5] > c[6] || c[6] >= db.length(c[4]) || c[2] > c[3] ||
                c[3] >= db.length(c[0])) {
                fprintf(stderr, "[ndgpu] ndgpu_correct_piles: record %llu of pile %d names read %u / %u or a window outside them "
                                "(DB holds %u reads)\n", (unsigned long long)(r - pile_off[i]), i, c[0], c[4], db.n_reads());
                return -2;
            }
        }
    }
    join_reapers();  // the previous call's teardown
    const uint64_t call_order = DeviceAligner::next_order();  // (an older call in flight goes first on every context)
    const auto t_call0 = std::chrono::steady_clock::now();
    std::atomic<uint64_t> build_ns{0}, take_ns{0};
    // piles per sub-batch at most (the cost target below normally cuts earlier)
    size_t sub = 384;
    if (const char *e = getenv("NDGPU_SUBBATCH")) sub = (size_t)std::max(1, atoi(e));
    // longest seeds first: the scoring DP is a sequential chain per seed, so similar lengths
    // share a launch and the long chains start early
    std::vector<uint32_t> order((size_t)n_piles);
    for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return recs[pile_off[a] * 8 + 3] > recs[pile_off[b] * 8 + 3];
    });
    int drivers = 8;
    // (a context is a host thread that launches and waits: with fewer CPUs than contexts -- a rank of 8 on a node whose container has 16
    // -- the contexts only take the CPUs from one another's host phases)
    drivers = std::min(drivers, std::max(2, host_threads));
    if (const char *e = getenv("NDGPU_CONTEXTS")) drivers = std::max(1, std::min(atoi(e), (int)DeviceAligner::kMaxContexts));
    // sub-batches: at most `sub` piles and at most `tag_budget` estimated alignment columns each, so that the
    // device buffers of a context (sized by the largest sub-batch it has seen) stay bounded whatever the seed lengths
    uint64_t tag_budget = 900000000ull;
    DeviceAligner::plan_memory(drivers, &tag_budget);  // what the device has free now decides it
    if (const char *e = getenv("NDGPU_SUBBATCH_TAGS")) tag_budget = std::max<uint64_t>(1000000ull, strtoull(e, nullptr, 10));
    // Sub-batches = consecutive ranges of the length-sorted piles of (about) equal cost, two per context, pulled from one
    // queue by whichever context is free.  Cost = estimated alignment columns (what every device phase and the host's
    // low-quality-region stage scale with).  A context alternates device-heavy phases (alignment, MSA, scoring) with
    // host-heavy ones (candidate ranking, POA, second MSA, splicing): with several sub-batches per context the phases of
    // different contexts interleave instead of all contexts being on the host -- and the device idle -- at the end of a call.
    // (Until round 2 the first sub-batches were small and held the longest seeds, because a seed's scoring chain bounded
    // the call; the segment-parallel scoring DP removed that.)
    std::vector<size_t> sub_start{0};
    {
        std::vector<uint64_t> est((size_t)n_piles);
        uint64_t total = 0;
        for (size_t k = 0; k < (size_t)n_piles; k++) {
            const uint32_t pid = order[k];
            uint64_t e = 0;
            for (uint64_t r = pile_off[pid]; r < pile_off[pid + 1]; r++) e += (uint64_t)(recs[r * 8 + 3] - recs[r * 8 + 2] + 1);
            est[k] = e + e / 6;
            total += est[k];
        }
        // (measured on config 2, 1,666 piles: 1, 2, 3, 4, 6 per context = 1147, 1095, 1181, 1269, 1376 ms per step in round 2; 1 and 2
        // within noise since.  A small call -- the share of one rank of 4 or 8 -- is a matter of latency, not of filling the device:
        // every sub-batch of a context is another pass through the same dependent phases, and with one per context the 210 piles of a
        // rank of 8 take 170 ms instead of 262, the 407 of a rank of 4 255 instead of 320: profiles/r04_rank_share_config2.txt)
        int per_ctx = n_piles >= 1024 ? 2 : 1;
        if (const char *e = getenv("NDGPU_SUBBATCHES_PER_CONTEXT")) per_ctx = std::max(1, atoi(e));
        // exactly drivers x per_ctx pieces of equal cost where the caps allow it (a cut where the running cost passes the next multiple of
        // total / pieces): with "cut before the piece would overflow" the pieces came out slightly small and a 17th, alone in a third round
        // of the contexts, ended every config-2 call ~60 ms late.  The caps (piles per sub-batch, the memory plan's columns) still cut.
        // Rounds may taper (NDGPU_TAPER=w1,w2,...: the share of the call's cost each round of the contexts takes; default: equal
        // rounds): what the LAST round leaves for the host -- its piles' candidate ranking, POA, two more rounds -- ends the call with
        // the device idle, and it is that round's share of the call's host work divided by the CPUs the process has.
        std::vector<double> weights((size_t)per_ctx, 1.0 / per_ctx);
        if (const char *e = getenv("NDGPU_TAPER")) {
            std::vector<double> w;
            for (const char *p = e; *p;) {
                char *end = nullptr;
                const double v = strtod(p, &end);
                if (end == p) break;
                if (v > 0) w.push_back(v);
                p = *end ? end + 1 : end;
            }
            double sum = 0;
            for (double v : w) sum += v;
            if (!w.empty() && sum > 0) {
                for (double &v : w) v /= sum;
                weights = w;
            }
        }
        std::vector<uint64_t> targets;   // cumulative cost at which piece k ends
        {
            double at = 0;
            for (double w : weights)
                for (int c = 0; c < drivers; c++) {
                    at += w / drivers;
                    targets.push_back((uint64_t)(at * (double)total));
                }
            if (!targets.empty()) targets.pop_back();  // (the last piece ends with the piles)
        }
        const uint64_t n_target = (uint64_t)targets.size() + 1;
        const uint64_t piece = std::min<uint64_t>(tag_budget, std::max<uint64_t>(total / n_target + 1, 2000000ull));
        const bool by_target = total / n_target + 1 <= tag_budget && total / n_target + 1 >= 2000000ull;  // (neither cap nor floor bites)
        uint64_t acc = 0, run = 0;
        size_t cnt = 0, next_cut = 0;
        for (size_t k = 0; k < (size_t)n_piles; k++) {
            bool cut = cnt && (cnt >= sub || acc + est[k] > tag_budget);
            if (!cut && cnt) {
                if (by_target) cut = next_cut < targets.size() && run + est[k] / 2 >= targets[next_cut];
                else cut = acc + est[k] > piece;
            }
            if (cut) {
                sub_start.push_back(k);
                acc = 0, cnt = 0;
                while (by_target && next_cut < targets.size() && run + est[k] / 2 >= targets[next_cut]) next_cut++;
            }
            acc += est[k], run += est[k], cnt++;
        }
        sub_start.push_back((size_t)n_piles);
    }
    const size_t n_sub = sub_start.size() - 1;
    drivers = (int)std::min<size_t>((size_t)drivers, n_sub);
    int threads_each = std::max(1, host_threads / drivers);  // (measured: more threads per context is slower)
    if (const char *e = getenv("NDGPU_THREADS_PER_CONTEXT")) threads_each = std::max(1, atoi(e));
    CoreGovernor::set_total(getenv("NDGPU_NO_BORROW") ? 0 : host_threads);
    std::atomic<uint64_t> oom_seeds{0};
    std::mutex done_mu;
    // one range of the length-sorted piles through one context; out of device memory -> the context's buffers are
    // dropped and the range is halved, down to a single pile, which is then an out-of-memory seed (len 3)
    std::function<void(int, size_t, size_t)> run_range = [&](int ctx, size_t base, size_t cnt) {
        std::vector<PileEngine *> eng(cnt, nullptr);
        const auto t_b0 = std::chrono::steady_clock::now();
        CoreLease *build_lease = new CoreLease(threads_each);
        parallel_for(cnt, build_lease->n, [&](size_t k) {
            const uint32_t pid = order[base + k];
            const uint64_t r0 = pile_off[pid], r1 = pile_off[pid + 1];
            const size_t n = (size_t)(r1 - r0);
            std::vector<unsigned> st(n), en(n), len(n);
            std::vector<int64_t> dev(n);
            unsigned max_aln = n ? recs[r0 * 8 + 3] + 1 : 0;
            for (size_t i = 0; i < n; i++) {
                const uint32_t *r = recs + (r0 + i) * 8;
                dev[i] = db.window_offset(r[4], r[5], r[6], (int)r[1]);
                len[i] = r[6] - r[5] + 1;
                st[i] = r[2];
                en[i] = r[3];
                const unsigned v = r[3] - r[2] + r[6] - r[5] + 2;
                if (v > max_aln && r[0] != r[4]) max_aln = v;
            }
            const unsigned lq = n ? std::min<unsigned>(en[0] / 2, max_lq_length) : max_lq_length;
            std::string seed;  // only the HiFi consensus compares against the seed's own bases
            if (read_type == 3 && n) seed = db.window(recs[r0 * 8 + 4], recs[r0 * 8 + 5], recs[r0 * 8 + 6], 0);
            eng[k] = new PileEngine(len.data(), dev.data(), st.data(), en.data(), (unsigned)n,
                                    make_params(max_aln, min_len_aln, max_cov_aln, min_cov, lq,
                                                min_error_corrected_ratio, split, fast, read_type),
                                    read_type == 3 ? seed.c_str() : nullptr);
        });
        delete build_lease;
        build_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_b0).count();
        bool oom = false;
        try {
            HipBackend be(ctx, threads_each, h->dev_pool, call_order, true);
            run_engines(eng.data(), cnt, be, threads_each);
        } catch (const DeviceOom &e) {
            oom = true;
            if (getenv("NDGPU_TRACE"))
                fprintf(stderr, "[ndgpu trace] out of device memory (%zu bytes wanted) in a sub-batch of %zu piles: %s\n", e.bytes, cnt,
                        cnt > 1 ? "halved" : "reported as an out-of-memory seed (len 3)");
        }
        if (oom) {
            DeviceAligner::context(ctx).release_memory();
            DeviceAligner::forget_sizes();
            for (PileEngine *e : eng) delete e;
            if (cnt == 1) {
                out[order[base]] = (consensus_trimed *)make_error_seed(3);
                oom_seeds++;
                if (done) {
                    std::lock_guard<std::mutex> lock(done_mu);
                    done(user, &order[base], 1);
                }
            } else {
                run_range(ctx, base, cnt / 2);
                run_range(ctx, base + cnt / 2, cnt - cnt / 2);
            }
            return;
        }
        const auto t_t0 = std::chrono::steady_clock::now();
        for (size_t k = 0; k < cnt; k++) out[order[base + k]] = (consensus_trimed *)eng[k]->take_result();
        if (done) {  // (the caller's hand-over of this sub-batch's records, while the other contexts work on)
            std::lock_guard<std::mutex> lock(done_mu);
            done(user, &order[base], (int)cnt);
        }
        // tearing down the per-pile host state (thousands of small vectors per pile, ~0.3 ms each; parallel frees
        // only fight over the allocator) is not on anybody's critical path: a reaper thread does it while the caller
        // goes on, and the next call (or the library's unload) waits for it
        {
            std::lock_guard<std::mutex> lock(g_reap_mu);
            g_reapers.emplace_back([v = std::move(eng)] {
                for (PileEngine *e : v) delete e;
            });
        }
        take_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_t0).count();
    };
    std::atomic<size_t> next_sub{0};
    auto drive = [&](int ctx) {
        // (the context serves this call's sub-batches one behind the other before a newer call's: the place in its line is kept
        // between them)
        DeviceAligner &dev = DeviceAligner::context(ctx);
        dev.reserve_batches(call_order);
        struct Unreserve { DeviceAligner &d; uint64_t o; ~Unreserve() { d.unreserve_batches(o); } } unreserve{dev, call_order};
        for (;;) {
            const size_t sb = next_sub.fetch_add(1);
            if (sb >= n_sub) break;
            run_range(ctx, sub_start[sb], sub_start[sb + 1] - sub_start[sb]);
        }
    };
    std::vector<std::thread> th;
    for (int c = 1; c < drivers; c++) th.emplace_back(drive, c);
    drive(0);
    for (auto &t : th) t.join();
    DeviceAligner::level_buffers(drivers);  // (nothing in flight now) no context will have to grow a buffer in the middle of the next call
    if (getenv("NDGPU_TRACE"))
        fprintf(stderr, "[ndgpu trace] correct_piles %d piles in %zu sub-batches: %.1f ms wall | engine build %.1f ms, result take %.1f ms (summed over contexts)\n",
                n_piles, n_sub, std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_call0).count() * 1e-3,
                build_ns.load() * 1e-6, take_ns.load() * 1e-6);
    if (getenv("NDGPU_PROF")) {
        fprintf(stderr, "[ndgpu prof] drivers %d x %d threads | main %.3f s  extract %.3f s  align %.3f s (%llu jobs)  "
                        "advance %.3f s  (driver-thread wall sums)\n",
                drivers, threads_each, g_prof.main_ns * 1e-9, g_prof.extract_ns * 1e-9, g_prof.align_ns * 1e-9,
                (unsigned long long)g_prof.jobs.load(), g_prof.advance_ns * 1e-9);
        fprintf(stderr, "[ndgpu prof] run_main: prep %.3f  align(K7+K8a) %.3f  tags %.3f  msa(K9+K10) %.3f  post %.3f s\n",
                g_prof.m_prep * 1e-9, g_prof.m_aln * 1e-9, g_prof.m_tags * 1e-9, g_prof.m_msa * 1e-9,
                g_prof.m_post * 1e-9);
        fprintf(stderr, "[ndgpu prof] advance, CPU seconds by phase: after main %.3f  after extract %.3f  after LQ round 1 %.3f  after round 2 + splice %.3f\n",
                g_prof.adv_ns[0] * 1e-9, g_prof.adv_ns[1] * 1e-9, g_prof.adv_ns[2] * 1e-9, g_prof.adv_ns[3] * 1e-9);
        fprintf(stderr, "[ndgpu prof] after extract: 8-mer ranking %.3f  POA %.3f  LQ round 1 layout %.3f s (CPU seconds)\n", g_prof.rank_ns * 1e-9,
                g_prof.poa_ns * 1e-9, g_prof.lqstart_ns * 1e-9);
        g_prof.rank_ns = g_prof.poa_ns = g_prof.lqstart_ns = 0;
        fprintf(stderr, "[ndgpu prof] LQ-stage alignment batches (%llu jobs): host packing %.3f s, device round trip %.3f s, host decoding %.3f s "
                        "(wall sums over contexts)\n", (unsigned long long)g_prof.c_jobs.load(), g_prof.c_pack * 1e-9, g_prof.c_dev * 1e-9,
                g_prof.c_decode * 1e-9);
        g_prof.c_pack = g_prof.c_dev = g_prof.c_decode = g_prof.c_jobs = 0;
        for (auto &a : g_prof.adv_ns) a = 0;
        g_prof.main_ns = g_prof.extract_ns = g_prof.align_ns = g_prof.advance_ns = g_prof.jobs = 0;
        g_prof.m_prep = g_prof.m_aln = g_prof.m_tags = g_prof.m_msa = g_prof.m_post = 0;
    }
    return 0;
}

void align(char *query_seq, int q_len, char *target_seq, int t_len, alignment *align_rtn, int *, uint8_t **) {
    run_single_alignment(query_seq, q_len, target_seq, t_len, align_rtn, 0);
}

void align_hq(char *query_seq, int q_len, char *target_seq, int t_len, alignment *align_rtn, int *, uint8_t **) {
    run_single_alignment(query_seq, q_len, target_seq, t_len, align_rtn, 1);
}

// replaces lib/align.c:580-679 (`align_nd`: global alignment, match 2 / mismatch -4 / gap open -4 / gap extend -2, two
// traceback bits per cell packed four to a byte).  Exported by the reference's nextcorrect.so but called from nowhere
// (its one call site, lib/ctg_cns.c:1357, is commented out), so it stays a host routine.  The definition names its
// parameters (s2, s2_l, s1, s1_l): the FIRST sequence indexes the columns; rows are written to t_aln_str.
void align_nd(const char *s2, const uint32_t s2_l, const char *s1, const uint32_t s1_l, alignment *aln) {
    static const uint8_t MMH[4] = {64, 16, 4, 1}, INS[4] = {128, 32, 8, 2}, DEL[4] = {192, 48, 12, 3};
    const int mas = 2, mis = -4, gos = -4, ges = -2;
    const size_t row = ((size_t)s2_l >> 2) + 1;
    std::vector<uint8_t> dm(row * ((size_t)s1_l + 1), 0);
    auto d = [&](uint32_t i) { return dm.data() + row * (size_t)i; };
    std::vector<int32_t> sc((size_t)s2_l + 1, 0);
    int32_t cs = 0;
    d(0)[0] |= MMH[0];
    for (uint32_t j = 1; j <= s2_l; j++) {
        sc[j] = sc[j - 1] + ((d(0)[(j - 1) >> 2] & DEL[(j - 1) & 3]) == DEL[(j - 1) & 3] ? ges : gos);
        d(0)[j >> 2] |= DEL[j & 3];
    }
    for (uint32_t i = 1; i <= s1_l; i++) {
        uint8_t *di = d(i);
        const uint8_t *dp = d(i - 1);
        for (uint32_t j = 0; j <= s2_l; j++) {
            if (j == 0) {
                cs = sc[0] + ((dp[0] & DEL[0]) == INS[0] ? ges : gos);
                di[0] = INS[0];
            } else {
                const uint32_t k = j & 3;
                const int32_t ms = sc[j - 1] + (s1[i - 1] == s2[j - 1] ? mas : mis);
                const int32_t is = sc[j] + ((dp[j >> 2] & DEL[k]) == INS[k] ? ges : gos);
                const int32_t ds = cs + ((di[(j - 1) >> 2] & DEL[(j - 1) & 3]) == DEL[(j - 1) & 3] ? ges : gos);
                sc[j - 1] = cs;
                uint8_t mv;
                if (ms > is) {
                    if (ms > ds) cs = ms, mv = MMH[k];
                    else cs = ds, mv = DEL[k];
                } else {
                    if (is > ds) cs = is, mv = INS[k];
                    else cs = ds, mv = DEL[k];
                }
                di[j >> 2] |= mv;
                if (j == s2_l) sc[j] = cs;
            }
        }
    }
    uint32_t a = s1_l, b = s2_l, n = 0;
    while (a != 0 || b != 0) {
        const uint32_t k = b & 3;
        const uint8_t mv = d(a)[b >> 2] & DEL[k];
        if (mv == MMH[k]) {
            aln->t_aln_str[n] = s1[--a];
            aln->q_aln_str[n] = s2[--b];
        } else if (mv == DEL[k]) {
            aln->q_aln_str[n] = s2[--b];
            aln->t_aln_str[n] = '-';
        } else {
            aln->t_aln_str[n] = s1[--a];
            aln->q_aln_str[n] = '-';
        }
        n++;
    }
    aln->aln_len = n;
    aln->t_aln_str[n] = aln->q_aln_str[n] = '\0';
    reverse_str(aln->t_aln_str, (int)n);
    reverse_str(aln->q_aln_str, (int)n);
}

// The device owns the DP state; these exist so that code written against
// lib/align.h links unchanged.  V gets the reference's size, D a 1-slot table.
void malloc_vd(int **V, uint8_t ***D, uint64_t max_mem_d) {
    *V = (int *)malloc((size_t)(max_mem_d ? max_mem_d : 1) * 2 * sizeof(int));
    *D = (uint8_t **)calloc(1, sizeof(uint8_t *));
}
void clean_V(int *V, int max_mem_d) {
    if (V && max_mem_d > 0) memset(V, 0, (size_t)max_mem_d * 2 * sizeof(int));
}
void destory_vd(int *V, uint8_t **D) {
    free(V);
    free(D);
}

void reverse_str(char *str, int len) {
    for (int a = 0, b = len - 1; a < b; a++, b--) std::swap(str[a], str[b]);
}

void revcomp_bseq(char *str, int len) {
    // complement table semantics of lib/align.c:3-20 for the IUPAC letters it maps
    static const char *from = "ACGTUMRWSYKVHDBNacgtumrwsykvhdbn";
    static const char *to = "TGCAAKYWSRMBDHVNtgcaakywsrmbdhvn";
    static unsigned char lut[256];
    static bool init = false;
    if (!init) {
        for (int i = 0; i < 256; i++) lut[i] = (unsigned char)i;
        for (int i = 0; from[i]; i++) lut[(unsigned char)from[i]] = (unsigned char)to[i];
        init = true;
    }
    int a = 0, b = len - 1;
    while (a < b) {
        const unsigned char x = lut[(unsigned char)str[a]], y = lut[(unsigned char)str[b]];
        str[a++] = (char)y;
        str[b--] = (char)x;
    }
    if (a == b) str[a] = (char)lut[(unsigned char)str[a]];
}

void str_tolower(char *p) {
    for (; *p; ++p) *p |= 0x20;
}
void str_toupper(char *p) {
    for (; *p; ++p) *p &= (char)0xdf;
}

char *poa_to_consensus(const void *seqs, const int seq_count) {
    // struct seq_ { uint16_t order, kscore, len; char seq[10000]; }  (lib/nextcorrect.h:63-68)
    const size_t stride = 3 * sizeof(uint16_t) + 10000;
    std::vector<std::string> in;
    for (int i = 0; i < seq_count; i++) {
        const unsigned char *rec = (const unsigned char *)seqs + (size_t)i * stride;
        uint16_t len;
        memcpy(&len, rec + 4, sizeof(len));
        in.emplace_back((const char *)rec + 6, (size_t)len);
    }
    std::string r = poa_consensus(in);
    char *out = (char *)malloc(r.size() + 1);
    memcpy(out, r.c_str(), r.size() + 1);
    return out;
}

void ndgpu_get_stats(ndgpu_stats *o) {
    RuntimeStats s = DeviceAligner::total_stats();
    o->tasks = s.tasks;
    o->wide_tasks = s.wide_tasks;
    o->cells = s.cells;
    o->d_steps = s.d_steps;
    o->trace_bits = s.trace_bits;
    o->columns = s.columns;
    o->pool_bases = s.pool_bases;
    o->seq_bases = s.seq_bases;
    o->max_band = s.max_band;
    o->forward_launches = s.forward_launches;
    o->forward_ms = s.forward_ms;
    o->traceback_ms = s.traceback_ms;
    o->tags_ms = s.tags_ms;
    o->links_ms = s.links_ms;
    o->score_ms = s.score_ms;
    o->extract_ms = s.extract_ms;
    o->piles = s.piles;
    o->tags = s.tags;
    o->cells_msa = s.cells_msa;
    o->path_items = s.path_items;
    o->links = s.links;
    o->score_launches = s.score_launches;
    o->backtrack_ms = s.backtrack_ms;
    o->score_segments = s.score_segments;
    o->score_repairs = s.score_repairs;
    o->score_slow_piles = s.score_slow_piles;
    o->trace_words = s.trace_words;
    o->lq_rounds = s.lq_rounds;
    o->lq_declined = s.lq_declined;
    o->lq_ms = s.lq_ms;
    o->allocs = s.allocs, o->alloc_ms = s.alloc_ms, o->level_allocs = s.level_allocs, o->level_ms = s.level_ms;
    o->traceback_launches = s.traceback_launches, o->lq_launches = s.lq_launches, o->lq_columns = s.lq_columns;
    o->lq_aln_columns = s.lq_aln_columns, o->lq_bases = s.lq_bases, o->lq_out = s.lq_out;
    o->lq_jobs = s.lq_jobs, o->lq_repairs = s.lq_repairs;
    o->tb_tasks = s.tb_tasks, o->tb_walkers = s.tb_walkers, o->tb_fallbacks = s.tb_fallbacks;
}

void ndgpu_reset_stats(void) { DeviceAligner::reset_all_stats(); }

uint64_t ndgpu_release_memory(void) {
    size_t f0 = 0, f1 = 0, t = 0;
    (void)hipMemGetInfo(&f0, &t);
    join_reapers();
    for (int c = 0; c < DeviceAligner::kMaxContexts; c++)
        if (DeviceAligner *d = DeviceAligner::peek(c)) (void)d->release_memory_if_idle();  // a context with a batch open keeps its buffers
    (void)hipMemGetInfo(&f1, &t);
    return f1 > f0 ? (uint64_t)(f1 - f0) : 0;
}

void ndgpu_reserve_device_memory(uint64_t bytes) { DeviceAligner::reserve_device_memory(bytes); }

int ndgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

}  // extern "C"
