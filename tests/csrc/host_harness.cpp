// tests/csrc/host_harness.cpp -- TEST-ONLY glue.
// Runs the product's host-side consensus engine (nextdenovo_amd/csrc/consensus.cpp,
// poa.cpp) over a CPU Backend built from the oracle (oracle/ond_oracle.c,
// oracle/msa_oracle.c), so that `pytest -m "not gpu"` can check the host logic against
// the reference without a GPU.  The shipped library never links the oracle; its only
// Backend is HipBackend.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../nextdenovo_amd/csrc/nd_host.h"
#include "../../oracle/nd_oracle.h"

using namespace ndgpu;

namespace {

uint8_t base_code(char c) {  // lib/nextcorrect.c:52-62
    switch (c) {
        case 'A': case 'a': return 0;
        case 'T': case 't': return 1;
        case 'G': case 'g': return 2;
        case 'C': case 'c': return 3;
        case 'N': return 5;
        case 'M': return 6;
        default: return 4;
    }
}

class OracleBackend : public Backend {
  public:
    std::vector<std::vector<std::vector<nd_oracle_tag>>> kept;  // [slot][read] tags

    void run_align(AlnJob **jobs, size_t n) override {
        for (size_t i = 0; i < n; i++) {
            AlnJob &j = *jobs[i];
            nd_oracle_aln r;
            std::vector<uint8_t> ops((size_t)j.q_len + j.t_len + 1);
            nd_oracle_align(j.q, j.q_len, j.t, j.t_len, j.hq, &r, nullptr, nullptr, ops.data());
            j.status = r.status;
            j.q_used = r.q_used;
            j.t_used = r.t_used;
            if (r.status == 1) {
                ops.resize((size_t)r.aln_len);
                j.ops.swap(ops);
            } else j.ops.clear();
        }
    }

    void run_main(MainPile **piles, size_t n) override {
        kept.clear();
        kept.resize(n);
        for (size_t p = 0; p < n; p++) {
            MainPile &M = *piles[p];
            M.slot = (int)p;
            M.path.clear();
            const int L = (int)M.aln_end[0] + 1;
            std::vector<std::vector<nd_oracle_tag>> &reads = kept[p];
            int total = 0;
            for (unsigned i = 0; i < M.n && (unsigned)(total / L) <= M.max_cov_aln; i++) {
                std::vector<nd_oracle_tag> tags;
                if (i == 0) {
                    if (M.seq_len[0] < M.min_len_aln) continue;
                    total += (int)(M.aln_end[0] - M.aln_start[0] + 1);
                    for (unsigned t = 0; t < M.seq_len[0]; t++)
                        tags.push_back(nd_oracle_tag{(int32_t)(M.aln_start[0] + t), 0, base_code(M.seqs[0][t])});
                } else {
                    const char *q = M.seqs[i];
                    const int ql = (int)M.seq_len[i];
                    const char *t = M.seqs[0] + M.aln_start[i];
                    const int tl = (int)(M.aln_end[i] - M.aln_start[i] + 1);
                    nd_oracle_aln r;
                    std::vector<uint8_t> ops((size_t)ql + tl + 1);
                    nd_oracle_align(q, ql, t, tl, M.hq, &r, nullptr, nullptr, ops.data());
                    if (r.status != 1) continue;
                    unsigned ts = M.aln_start[i], te = M.aln_end[i];
                    int shift = 0;
                    const int len = nd_oracle_shift(ops.data(), r.aln_len, 8, &ts, &te, &shift);
                    if ((unsigned)len < M.min_len_aln || len == 0) continue;
                    total += (int)(te - ts + 1);
                    int qi = 0;
                    for (int c = 0; c < shift; c++) qi += ops[c] != 2;
                    int32_t tp = (int32_t)ts - 1;
                    uint16_t delta = 0;
                    for (int c = 0; c < len; c++) {
                        const uint8_t op = ops[shift + c];
                        if (op != 1) { tp++; delta = 0; }
                        tags.push_back(nd_oracle_tag{tp, delta++, op == 2 ? (uint8_t)4 : base_code(q[qi++])});
                    }
                }
                reads.push_back(std::move(tags));
            }
            M.n_aligned = (unsigned)reads.size();
            std::vector<const nd_oracle_tag *> ptr;
            std::vector<uint32_t> len;
            size_t cap = 16;
            for (auto &r : reads) { ptr.push_back(r.data()); len.push_back((uint32_t)r.size()); cap += r.size(); }
            std::vector<nd_oracle_path> path(cap);
            const long np = reads.empty() ? 0 : nd_oracle_msa_path(ptr.data(), len.data(), (int)reads.size(), L, M.factor,
                                                                   path.data(), (long)cap);
            for (long k = 0; k < np; k++)
                M.path.push_back(PathStep{path[k].t_pos, path[k].delta, path[k].base, path[k].link_count, path[k].coverage});
        }
    }

    void run_extract(ExtractPile **piles, size_t n) override {
        static const char kI2B[] = "ATGC-NM";
        for (size_t p = 0; p < n; p++) {
            const auto &reads = kept[piles[p]->slot];
            for (RegionReq &rq : piles[p]->regions) {
                rq.cands.clear();
                rq.cand_rank.clear();
                rq.n_large = 0;
                const unsigned lim0 = rq.max_len0 ? rq.max_len0 : rq.max_len;
                unsigned rank = 0;
                const int start = (int)rq.start, end = (int)rq.end;
                for (const auto &tg : reads) {  // lib/nextcorrect.c:373-404 (and :757-784 for HiFi)
                    const unsigned lim = rank == 0 ? lim0 : rq.max_len;
                    const unsigned my_rank = rank++;
                    if (!(tg.front().t_pos <= start && tg.back().t_pos >= end)) continue;
                    std::string s;
                    bool too_long = false;
                    for (size_t k = (size_t)(start - tg.front().t_pos); k < tg.size() && tg[k].t_pos <= end; k++)
                        if (tg[k].t_pos >= start && tg[k].base != 4) {
                            s.push_back(kI2B[tg[k].base]);
                            if (s.size() > lim - 1) { rq.n_large++; too_long = true; break; }
                        }
                    if (!s.empty() && !too_long) {
                        rq.cands.push_back(std::move(s));
                        rq.cand_rank.push_back((uint16_t)my_rank);
                    }
                    if (rq.cands.size() >= 40) break;
                }
            }
        }
    }

    void end_batch() override { kept.clear(); }
};

}  // namespace

extern "C" ConsensusTrimed *ndtest_correct(char **seqs, unsigned *aln_start, unsigned *aln_end, unsigned seq_count,
                                           unsigned max_mem_len, unsigned min_len_aln, unsigned max_cov_aln,
                                           unsigned min_cov, unsigned lqseq_max_length, float ratio, unsigned split,
                                           unsigned fast, int read_type) {
    CorrectParams p;
    p.max_mem_len = max_mem_len; p.min_len_aln = min_len_aln; p.max_cov_aln = max_cov_aln; p.min_cov = min_cov;
    p.lqseq_max_length = lqseq_max_length; p.min_error_corrected_ratio = ratio; p.split = split; p.fast = fast;
    p.read_type = read_type;
    PileEngine eng(seqs, aln_start, aln_end, seq_count, p);
    PileEngine *ep = &eng;
    OracleBackend be;
    run_engines(&ep, 1, be, 1);
    return eng.take_result();
}

extern "C" void ndtest_free(ConsensusTrimed *c) {
    free(c->seq);
    free(c);
}

extern "C" int ndtest_poa(const char **seqs, int n, char *out, int cap) {
    std::vector<std::string> v;
    for (int i = 0; i < n; i++) v.emplace_back(seqs[i]);
    std::string r = poa_consensus(v);
    if ((int)r.size() + 1 > cap) return -1;
    memcpy(out, r.c_str(), r.size() + 1);
    return (int)r.size();
}

// CPU seconds the engine spent in PileEngine::advance since the last call, by phase (after main / after extract / after LQ
// round 1 / after round 2 + splice): where the host side of the low-quality-region stage goes (tools/host_profile.py).
extern "C" void ndtest_advance_profile(double out[4]) {
    for (int i = 0; i < 4; i++) {
        out[i] = g_prof.adv_ns[i].load() * 1e-9;
        g_prof.adv_ns[i] = 0;
    }
}
// ... and inside "after extract": 8-mer ranking, POA, laying out LQ round 1
extern "C" void ndtest_extract_profile(double out[3]) {
    out[0] = g_prof.rank_ns.load() * 1e-9, out[1] = g_prof.poa_ns.load() * 1e-9, out[2] = g_prof.lqstart_ns.load() * 1e-9;
    g_prof.rank_ns = 0, g_prof.poa_ns = 0, g_prof.lqstart_ns = 0;
}
