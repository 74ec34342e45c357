// Host-side declarations of the MI355X read-correction engine (internal; the
// public C ABI is include/ndgpu_nextcorrect.h).
#pragma once

#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

namespace ndgpu {

// ABI-compatible with the reference's `consensus_trimed` (lib/nextcorrect.h:70-74).
struct ConsensusTrimed {
    unsigned int len;
    float identity;
    char *seq;
};

// Column kinds of a pairwise alignment, as produced by the HIP traceback kernel.
enum : uint8_t { OP_MATCH = 0, OP_QONLY = 1, OP_TONLY = 2 };

enum : int { ALN_NONE = 0, ALN_OK = 1, ALN_GAP_ABORT = 2 };

// One query-vs-target alignment request.  Sequences are ASCII on the host side;
// the device runtime packs them to 2 bit.  `hq` selects the align_hq
// thresholds (lib/align.c:563-570) instead of align (:572-578).
struct AlnJob {
    const char *q = nullptr;
    int q_len = 0;
    const char *t = nullptr;
    int t_len = 0;
    int hq = 0;
    // optional: `t` points into a larger buffer shared by many jobs (the seed); the
    // runtime then packs/uploads that buffer once
    const char *t_owner = nullptr;
    int t_owner_len = 0;
    // optional: base offsets of q[0] / t[0] inside the device-resident read DB
    // (ReadDb); when >= 0 nothing is packed or uploaded for that sequence
    int64_t q_dev = -1;
    int64_t t_dev = -1;
    // optional: q already packed (2 bit per base, 16 bases per word, first base in the low bits: pack_2bit_lsb below)
    const uint32_t *q_words = nullptr;
    // results
    int status = ALN_NONE;
    int q_used = 0;            // aln_q_len
    int t_used = 0;            // aln_t_len
    std::vector<uint8_t> ops;  // forward column kinds (empty unless ALN_OK)
};

// One visited cell of the main MSA's best_pp walk, origin first (device: PathItem).
struct PathStep {
    int32_t t_pos;
    uint16_t delta;
    uint8_t base;   // A0 T1 G2 C3 -4 N5 (lib/nextcorrect.c:52-62)
    uint16_t link;  // best_link_count of the cell
    uint16_t cov;   // coverage of the column
};

// Main-phase request of one pile: align every read to its seed window, build the MSA
// link graph, score it and walk best_pp (lib/nextcorrect.c:2271-2293, 2130-2202).
// Executed entirely on the device by the HIP backend.
struct MainPile {
    // inputs
    unsigned n = 0;                      // records, [0] = the seed
    const char *const *seqs = nullptr;   // ASCII sequences, or nullptr when dev_off is used
    const unsigned *seq_len = nullptr;
    const unsigned *aln_start = nullptr;
    const unsigned *aln_end = nullptr;
    const int64_t *dev_off = nullptr;    // ReadDb pool offsets of seqs[i][0], or nullptr
    unsigned min_len_aln = 500, max_cov_aln = 130;
    int factor = 3;
    int hq = 0;
    // outputs
    std::vector<PathStep> path;
    unsigned n_aligned = 0;              // reads that entered the MSA (incl. the seed)
    int slot = -1;                       // backend-private handle, valid until end_batch()
};

// Candidate strings of one low-quality region (lib/nextcorrect.c:373-404).
struct RegionReq {
    unsigned start = 0, end = 0;         // inclusive seed columns
    unsigned max_len = 0;                // lqseq_max_length
    unsigned max_len0 = 0;               // limit for the seed's own candidate (0: same as max_len)
    std::vector<std::string> cands;      // <= 40, in pile order
    std::vector<uint16_t> cand_rank;     // source read of each candidate (index among aligned reads)
    unsigned n_large = 0;                // reads longer than max_len - 1 seen before the 40th candidate
};

struct ExtractPile {
    int slot = -1;
    std::vector<RegionReq> regions;
};

// One low-quality-region round of a pile (generate_consensus_trimed + get_lqseqs_from_align_tags, lib/nextcorrect.c:1538-1669,
// 1250-1338) as a device request: align the jobs, link the regions' pseudo-seeds with 'N' columns, second MSA over the
// <= 30 rows, scoring DP, walk.  A backend that cannot serve it (or declines a pile: ok = false) leaves the round to the host
// path (run_align on the same jobs + the engine's own MSA).
struct LqRound {
    struct Piece {
        int job;                          // index into *jobs, -1: nothing to align in this (row, region) slot
        unsigned sl;                      // pseudo-seed length of the region
    };
    std::vector<AlnJob> *jobs = nullptr;  // q / q_len / q_words, t / t_len (the region's pseudo-seed), hq
    std::vector<Piece> pieces;            // row-major: 30 rows x n_regions, regions in the reference's (descending) order
    unsigned n_regions = 0;
    int factor = 2, qv_factor = 5;        // lib/nextcorrect.c:1262, 1303
    // outputs
    bool ok = false;
    std::string lqc;                      // the walk's characters, origin first
};

// ASCII [ACGT]* -> 2 bit per base, 16 bases per word, first base in the low bits (the device's sequence pools); false on any
// other byte.  out must hold (n + 15) / 16 words.
bool pack_2bit_lsb(uint32_t *out, const char *s, size_t n);

// Executes the device-side work of a batch of piles.  The product has exactly one
// implementation (HipBackend, device_runtime.hip); tests plug the CPU oracle here to
// exercise the host logic without a GPU.
class Backend {
  public:
    virtual ~Backend() {}
    virtual void run_main(MainPile **piles, size_t n) = 0;
    virtual void run_extract(ExtractPile **piles, size_t n) = 0;
    virtual void run_align(AlnJob **jobs, size_t n) = 0;
    // low-quality-region rounds on the backend; false: not offered (every round goes the host way)
    virtual bool run_lq(LqRound **rounds, size_t n) { (void)rounds; (void)n; return false; }
    virtual void end_batch() = 0;  // releases whatever run_main kept for run_extract
};

struct CorrectParams {
    unsigned max_mem_len = 0;
    unsigned min_len_aln = 500;
    unsigned max_cov_aln = 130;
    unsigned min_cov = 4;
    unsigned lqseq_max_length = 10000;
    float min_error_corrected_ratio = 0.8f;
    unsigned split = 0;
    unsigned fast = 0;
    int read_type = 1;  // 1 ont, 2 clr, 3 hifi
};

std::string poa_consensus(const std::vector<std::string> &seqs);

class PileImpl;

// Per-seed consensus state machine.  Device work is exposed as requests so that many
// piles share each launch:
//     MAIN (MainPile) -> EXTRACT (ExtractPile) -> LQ round 1 (AlnJob) -> LQ round 2 -> DONE
class PileEngine {
  public:
    enum Phase { MAIN = 0, EXTRACT = 1, LQ_ROUND = 2, DONE = 3 };
    // ASCII form (the nextCorrect ABI): seqs[i] NUL-terminated.
    PileEngine(const char *const *seqs, const unsigned *aln_start, const unsigned *aln_end, unsigned seq_count,
               const CorrectParams &prm);
    // Resident-DB form: only lengths and ReadDb pool offsets, no host copy of the reads
    // (`seed` = ASCII copy of the seed itself, needed by the HiFi consensus only; may be null).
    PileEngine(const unsigned *seq_len, const int64_t *dev_off, const unsigned *aln_start, const unsigned *aln_end,
               unsigned seq_count, const CorrectParams &prm, const char *seed = nullptr);
    ~PileEngine();
    PileEngine(const PileEngine &) = delete;
    PileEngine &operator=(const PileEngine &) = delete;

    Phase phase() const;
    bool done() const { return phase() == DONE; }
    MainPile *main_request();        // valid in MAIN
    ExtractPile *extract_request();  // valid in EXTRACT
    void collect_jobs(std::vector<AlnJob *> &out);  // LQ_ROUND (host path)
    LqRound *lq_request();           // LQ_ROUND: the round as one device request (nullptr: nothing to hand over)
    void advance();                  // consume the finished request(s), move on
    ConsensusTrimed *take_result();  // malloc'd, caller frees with free_consensus_trimed

  private:
    PileImpl *impl_;
};

// Wall-clock accounting of the host driver (summed over driver threads), NDGPU_PROF=1 prints it.
struct HostProf {
    std::atomic<uint64_t> main_ns{0}, extract_ns{0}, align_ns{0}, advance_ns{0}, build_ns{0}, jobs{0};
    std::atomic<uint64_t> m_prep{0}, m_aln{0}, m_tags{0}, m_msa{0}, m_post{0};  // inside run_main
    std::atomic<uint64_t> adv_ns[4] = {{0}, {0}, {0}, {0}};  // CPU time inside PileEngine::advance by phase (summed over threads)
    std::atomic<uint64_t> rank_ns{0}, poa_ns{0}, lqstart_ns{0};  // inside "after extract": 8-mer ranking, POA, laying out LQ round 1
    std::atomic<uint64_t> c_pack{0}, c_dev{0}, c_decode{0}, c_jobs{0};  // LQ-stage alignment batches: host packing / device round trip / host decoding
};
extern HostProf g_prof;

// The CPUs this process can actually have: the smaller of the hardware threads, the scheduling affinity and the cgroup CPU quota
// (cpu.max / cpu.cfs_quota_us).  A container on a 256-thread host with a quota of 16 CPUs that starts 256 busy threads spends its
// quota in the first quarter of every 100 ms period and is then throttled as a whole -- the threads that launch kernels included.
int effective_cpus();

// Host cores are shared by the driver threads of all device contexts: each host section is guaranteed its base share and
// borrows the cores that no other section is using at that moment (the tail of a call, when few contexts are still
// busy, would otherwise run on an eighth of the machine).  total = 0 switches borrowing off.
struct CoreGovernor {
    static void set_total(int total);
    static int acquire(int base);    // threads granted (>= base)
    static void release(int granted);
};
struct CoreLease {
    int n;
    explicit CoreLease(int base) : n(CoreGovernor::acquire(base)) {}
    ~CoreLease() { CoreGovernor::release(n); }
    CoreLease(const CoreLease &) = delete;
    CoreLease &operator=(const CoreLease &) = delete;
};

// Drives a set of engines to completion over one backend (host phases on `threads`).
void run_engines(PileEngine **eng, size_t n, Backend &be, int threads);

ConsensusTrimed *make_error_seed(unsigned len);

// Read database kept resident in HBM for the lifetime of a correction run: every
// read forward AND reverse-complemented, 2 bit per base, LSB-first words (the device
// packing of nd_device.h), so that any strand-corrected overlap substring
// (reference: getseq/subbit_, lib/ovlseq.c:39-48, lib/bseq.c:241-255) is a plain
// window of the pool -- no per-pile extraction or upload.
class ReadDb {
  public:
    // `words`: reference .2bit payload layout (lib/bseq.c:114-139): 16 bases per
    // uint32, first base in the two most significant bits; read i starts at
    // words[word_off[i]] and has len[i] bases.
    ReadDb(uint32_t n_reads, const uint32_t *words, const uint64_t *word_off, const uint32_t *len);
    uint32_t n_reads() const { return (uint32_t)len_.size(); }
    uint32_t length(uint32_t r) const { return len_[r]; }
    // pool offset (bases) of base `start` of the window [start, end] of read r, after
    // optional reverse-complement of the window
    int64_t window_offset(uint32_t r, uint32_t start, uint32_t end, int rev) const {
        return rev ? (int64_t)(rc_off_[r] + (len_[r] - 1 - end)) : (int64_t)(fwd_off_[r] + start);
    }
    // ASCII copy of the same window (host side)
    std::string window(uint32_t r, uint32_t start, uint32_t end, int rev) const;
    const std::vector<uint32_t> &pool() const { return pool_; }
    uint64_t total_bases() const { return total_; }

  private:
    std::vector<uint32_t> pool_;
    std::vector<uint64_t> fwd_off_, rc_off_;
    std::vector<uint32_t> len_;
    uint64_t total_ = 0;
};

}  // namespace ndgpu
