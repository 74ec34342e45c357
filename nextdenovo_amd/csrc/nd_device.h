// Device-side task/record layouts shared by the HIP kernels and the host runtime.
#pragma once

#include <cstdint>

namespace ndgpu {

// Sequences live in one pool of 2-bit codes, 16 bases per uint32, base i of a word
// at bits [2i, 2i+1] (LSB first, so that the first mismatch of a snake is a
// count-trailing-zeros of the XOR).  Offsets are in BASES, so a target window
// may start anywhere inside a seed without re-packing.
struct AlnTask {
    uint64_t q_off;      // first query base in the pool
    uint64_t t_off;      // first target base in the pool
    int32_t q_len;
    int32_t t_len;
    int32_t max_d;       // edit budget  (lib/align.c:567,575)
    int32_t band;        // band cap     (lib/align.c:568,576)
    uint64_t trace_off;  // uint64-word offset of trace row 0 (row d at trace_off + d*row_words)
    uint64_t mink_off;   // index of row 0 in the per-row min_k array
    uint64_t ops_off;    // first ops word of this task (uint32 units)
    uint32_t ops_cap;    // capacity in columns (= q_len + t_len)
    uint32_t row_words;  // uint64 words per trace row (2 in the LDS fast path)
    uint64_t v_off;      // wide path only: first int of this task's global V scratch
    uint32_t v_mask;     // wide path only: V ring size - 1 (power of two - 1)
    uint32_t pad_;
};

enum : int32_t {
    ST_NONE = 0,       // band cap or edit budget exhausted: no alignment
    ST_FINISHED = 1,   // forward sweep reached (q_len, t_len)
    ST_ALIGNED = 2,    // traceback done, ops valid
    ST_GAP_ABORT = 3,  // > 250 consecutive gap columns in traceback (lib/align.c:542-545)
    ST_NEED_WIDE = 4,  // live band exceeded the LDS fast path; rerun in the wide kernel
};

struct AlnOut {
    int32_t status;
    int32_t d_final;
    int32_t k_final;
    int32_t x_final;   // aln_q_len
    int32_t y_final;   // aln_t_len
    int32_t n_cols;    // alignment columns; ops occupy columns [ops_cap - n_cols, ops_cap)
    int32_t d_steps;   // counters for the roofline accounting
    int32_t max_band;
    int64_t cells;
};

constexpr uint64_t kOffDb = 1ull << 63;   // offset flag: sequence lives in the resident read DB
constexpr uint64_t kOffMask = kOffDb - 1;

constexpr int kFastVSize = 256;        // LDS ring of furthest-reaching x per diagonal
constexpr int kFastRowWords = 2;       // 128 same-parity diagonals per row
constexpr int kFastMaxBand = 253;      // band + 3 <= kFastVSize

// launchers (ond_kernels.hip)
void launch_ond_forward(const AlnTask *tasks, AlnOut *outs, const uint32_t *pool, const uint32_t *db_pool,
                        uint64_t *trace, int32_t *trace_mink, int n_tasks, void *stream);
void launch_ond_forward_wide(const AlnTask *tasks, AlnOut *outs, const uint32_t *pool, const uint32_t *db_pool,
                             uint64_t *trace, int32_t *trace_mink, int32_t *vscratch, const int32_t *task_ids, int n_ids, void *stream);
// task_ids == nullptr: tasks [0, n); otherwise the listed tasks only
void launch_ond_traceback(const AlnTask *tasks, AlnOut *outs, const uint32_t *pool, const uint32_t *db_pool,
                          const uint64_t *trace,
                          const int32_t *trace_mink, uint32_t *ops, const int32_t *task_ids, int n, void *stream);

}  // namespace ndgpu
