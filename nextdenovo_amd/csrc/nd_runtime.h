// Device runtime interface (one instance per process, lazily created on the first
// alignment so that a fork()ed worker pool -- lib/nextcorrect.py:232 -- initialises
// HIP in the child, never in the parent).
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "nd_host.h"

namespace ndgpu {

struct RuntimeStats {
    uint64_t tasks = 0;            // alignments executed
    uint64_t wide_tasks = 0;       // of which needed the HBM-ring kernel
    uint64_t cells = 0;            // (d,k) cells evaluated
    uint64_t d_steps = 0;          // edit steps executed
    uint64_t trace_bits = 0;       // 1-bit move records of alignments that finished
    uint64_t columns = 0;          // alignment columns produced
    uint64_t pool_bases = 0;       // bases packed + uploaded per batch (0 for DB-resident sequences)
    uint64_t seq_bases = 0;        // sum of q_len + t_len over all alignments (operand bases)
    uint32_t max_band = 0;
    uint32_t forward_launches = 0;
    double forward_ms = 0;         // HIP-event time of the forward kernel launches
    double traceback_ms = 0;       // K8a
    double tags_ms = 0;            // K8s + accept + K8b + column scan
    double links_ms = 0;           // K9
    double score_ms = 0;           // K10 (+ backtrack)
    double extract_ms = 0;         // K11
    uint64_t piles = 0;            // piles through the device main phase
    uint64_t tags = 0;             // alignment tags generated on the device
    uint64_t cells_msa = 0;        // MSA cells
    uint64_t path_items = 0;
    uint64_t links = 0;            // distinct MSA links
    uint64_t score_launches = 0;
    double backtrack_ms = 0;
    uint64_t score_segments = 0;   // K10 segments scored
    uint64_t score_repairs = 0;    // of which the stitch kernel scored again (a boundary check failed)
    uint64_t score_slow_piles = 0; // piles that went through the int64 HBM-resident scoring kernel
    uint64_t trace_words = 0;      // 64-bit words of trace records the register-path forward kernel wrote
    uint64_t lq_rounds = 0;        // low-quality-region rounds (pile x round) handed to K12
    uint64_t lq_declined = 0;      // of which the kernel declined (host path took them)
    double lq_ms = 0;              // K12
    uint64_t allocs = 0;           // device / pinned buffers (re)allocated while batches were running, since the last reset
    double alloc_ms = 0;           // wall time of those calls (an allocation in the middle of a step stalls every context)
    uint64_t level_allocs = 0;     // the same between batches (level_buffers: nothing in flight)
    double level_ms = 0;
    // per-kernel launch counts and the algorithmic units of K8a / K12, for the roofline entries of bench.py
    uint64_t traceback_launches = 0;  // K8a launches (main phase + low-quality-region rounds)
    uint64_t lq_launches = 0;         // K12 launches
    uint64_t lq_columns = 0;          // columns of the linked pseudo-seeds K12 walked
    uint64_t lq_aln_columns = 0;      // alignment columns (2-bit kinds) K12 read
    uint64_t lq_bases = 0;            // candidate bases (2-bit) K12 read
    uint64_t lq_out = 0;              // consensus characters K12 wrote
    uint64_t lq_jobs = 0;             // K12 jobs (runs of regions scored from a speculative start)
    uint64_t lq_repairs = 0;          // of which the stitch kernel scored again (failed boundary check)
    uint64_t tb_tasks = 0;            // alignments whose traceback ran in segments
    uint64_t tb_walkers = 0;          // walkers (segments) of those
    uint64_t tb_fallbacks = 0;        // of the alignments: refused by the stitch, walked again by the one-lane kernel
};

// Thrown when a device (or pinned host) allocation fails for lack of memory.  The C ABI catches it, releases the
// context's buffers, retries with half the piles and -- for a single pile that still does not fit -- reports the
// reference's out-of-memory seed (len == 3, lib/nextcorrect.c:2254-2261).  Every other HIP error aborts with a message.
struct DeviceOom {
    size_t bytes;
};

class DeviceAligner {
  public:
    static constexpr int kMaxContexts = 16;
    static DeviceAligner &instance();          // context 0
    static DeviceAligner &context(int i);
    static DeviceAligner *peek(int i);  // nullptr if context i was never used
    static RuntimeStats total_stats();
    // device memory plan of one batch call over `drivers` contexts: sets their trace budgets, returns the column budget of a sub-batch
    static void plan_memory(int drivers, uint64_t *tag_budget);
    static void level_buffers(int drivers);             // after a batch call: every context up to the largest sizes any context met
    static void reserve_device_memory(uint64_t bytes);  // left free by every later plan (another stage's working set)
    static void reset_all_stats();
    void align_batch(AlnJob **jobs, size_t n);
    // device main phase / candidate extraction of a batch of piles (see Backend in nd_host.h);
    // begin_batch()/end_batch() bracket one batch and serialise batches of one process
    // order: who goes first when several callers wait for this context (lower first; equal: any) -- the calls of a process take a
    // number each (next_order), so that a context serves the older of two batch calls in flight before the newer one
    void begin_batch(uint64_t order = 0, bool reserved = false);
    // a thread that will bring this context several batches of one call keeps its place in the line between them
    void reserve_batches(uint64_t order);
    void unreserve_batches(uint64_t order);
    static uint64_t next_order();
    void run_main(MainPile **piles, size_t n);
    void run_extract(ExtractPile **piles, size_t n);
    void run_lq(LqRound **rounds, size_t n);
    void end_batch();
    // resident read DB: every ndgpu_db handle owns its device copy (upload_db / free_db); a batch names the one its
    // AlnJob::q_dev / t_dev and MainPile::dev_off index into (use_db; nullptr: sequences come with the batch)
    static uint32_t *upload_db(const uint32_t *pool_words, size_t n_words, int device);  // nullptr: out of device memory
    static void free_db(uint32_t *dev_pool);
    void use_db(const uint32_t *dev_pool);
    int device() const;
    void set_host_threads(int n);   // threads used for packing / decoding inside a batch
    // after a DeviceOom: wait for the stream, drop every grow-only buffer of this context (they are re-created on demand)
    void release_memory();
    // after a DeviceOom: drop the process-wide "largest size any context held of a buffer" marks, so that the retry of a smaller
    // range asks for what IT needs and not for what just failed
    static void forget_sizes();
    // the public entry's form: only when no batch is open on this context (a batch keeps its main-phase buffers alive from
    // run_main to end_batch, between the calls that hold the context's lock); false = the context is busy, nothing released
    bool release_memory_if_idle();
    void *stream() const;
    RuntimeStats stats() const;
    void reset_stats();

  private:
    DeviceAligner();
    ~DeviceAligner();
    void run_chunk(AlnJob **jobs, size_t n);
    void run_wide(AlnJob **jobs, size_t n, const std::vector<int32_t> &ids);
    struct State;
    State *s_;
};

// The product's only Backend: every request runs in HIP kernels on the device.
class HipBackend : public Backend {
  public:
    explicit HipBackend(int ctx = 0, int host_threads = 1, const uint32_t *db_pool = nullptr, uint64_t order = ~0ull, bool reserved = false)
        : dev_(DeviceAligner::context(ctx)) {
        dev_.begin_batch(order == ~0ull ? DeviceAligner::next_order() : order, reserved);
        dev_.set_host_threads(host_threads);
        dev_.use_db(db_pool);
    }
    ~HipBackend() override { finish(); }
    void run_main(MainPile **piles, size_t n) override { dev_.run_main(piles, n); }
    void run_extract(ExtractPile **piles, size_t n) override { dev_.run_extract(piles, n); }
    void run_align(AlnJob **jobs, size_t n) override { dev_.align_batch(jobs, n); }
    bool run_lq(LqRound **rounds, size_t n) override {
        static const bool host_lq = getenv("NDGPU_LQ_HOST") != nullptr;  // test hook: the host path of the rounds
        if (host_lq) return false;
        dev_.run_lq(rounds, n);
        return true;
    }
    void end_batch() override { finish(); }

  private:
    void finish() {
        if (open_) dev_.end_batch();
        open_ = false;
    }
    DeviceAligner &dev_;
    bool open_ = true;
};

}  // namespace ndgpu
