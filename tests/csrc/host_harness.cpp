// tests/csrc/host_harness.cpp -- TEST-ONLY glue.
// Runs the product's host-side consensus engine (nextdenovo_amd/csrc/consensus.cpp,
// poa.cpp) with the CPU oracle (oracle/ond_oracle.c) plugged in as the alignment
// backend, so that `pytest -m "not gpu"` can check the host logic against the
// reference without a GPU.  The shipped library never links the oracle.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../nextdenovo_amd/csrc/nd_host.h"
#include "../../oracle/nd_oracle.h"

using namespace ndgpu;

static void oracle_backend(AlnJob **jobs, size_t n, void *) {
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

extern "C" ConsensusTrimed *ndtest_correct(char **seqs, unsigned *aln_start, unsigned *aln_end, unsigned seq_count,
                                           unsigned max_mem_len, unsigned min_len_aln, unsigned max_cov_aln,
                                           unsigned min_cov, unsigned lqseq_max_length, float ratio, unsigned split,
                                           unsigned fast, int read_type) {
    CorrectParams p;
    p.max_mem_len = max_mem_len; p.min_len_aln = min_len_aln; p.max_cov_aln = max_cov_aln; p.min_cov = min_cov;
    p.lqseq_max_length = lqseq_max_length; p.min_error_corrected_ratio = ratio; p.split = split; p.fast = fast;
    p.read_type = read_type;
    PileEngine eng(seqs, aln_start, aln_end, seq_count, p);
    std::vector<AlnJob *> jobs;
    while (!eng.done()) {
        jobs.clear();
        eng.collect_jobs(jobs);
        oracle_backend(jobs.data(), jobs.size(), nullptr);
        eng.advance();
    }
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
