// Device-side task/record layouts shared by the HIP kernels and the host runtime.
#pragma once

#include <cstdint>

#include "nd_lockstep.h"

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
    uint64_t trace_off;  // uint64-word offset of the task's trace (register path: a stream of <= 2 * max_d words;
                         // wide path: row d at trace_off + d*row_words)
    uint64_t mink_off;   // wide path: index of row 0 in the per-row min_k array; register path with checkpoints (segmented
                         // traceback): the task's first checkpoint slot
    uint64_t ops_off;    // first ops word of this task (uint32 units)
    uint32_t ops_cap;    // capacity in columns (= q_len + t_len)
    uint32_t row_words;  // wide path only: uint64 words per trace row
    uint64_t v_off;      // wide path only: first int of this task's global V scratch
    uint32_t v_mask;     // wide path only: V ring size - 1 (power of two - 1)
    uint32_t seg_off;    // segmented traceback: the task's first walker slot (TbSeg / TbSegOut)
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
    uint32_t trace_end;  // register path: words of trace stream written (the finishing step's record ends here)
    int32_t fin_idx;     // register path: bits [7:0] cell index of k_final in the finishing step; bits [15:8] (checkpointing
                         // forward kernel) cell index, in the last checkpoint row, of the cell the finishing cell descends from
};

// ---- segmented traceback (K8a in pieces; ond_kernels.hip) ----
// The forward kernel leaves a checkpoint every 2^cshift edit steps: per cell of that row its furthest-reaching x and the cell of
// the checkpoint row before that its move bits lead back to (packed: x in [23:0], cell in [31:24]; 128 cells a slot), per row the
// trace position behind its record and its min_k.  A chase along those cells names, for every checkpoint row under the finishing
// step, the cell the traceback passes through; one WALKER per checkpoint then walks its stretch of rows on a lane of its own.
struct TbSeg {            // a walker: starts at the top of row d_top, owns (emits) rows d_own .. d_end
    int32_t task;         // index into the launch's task table; -1: unused slot
    int32_t d_top;        // first row walked (a checkpoint row, or the finishing step for a task's first walker)
    int32_t d_own;        // first row owned: d_top for the first walker, d_top - warm otherwise (the rows above it are walked
                          // without output: the walk's x falls in with the true walk's within a few rows, see tb_walk_kernel)
    int32_t d_end;        // last row owned
    int32_t x;            // the state at the top of row d_top: query position, cell index, the row's min_k, trace position
    int32_t idx;
    int32_t min_k;
    uint32_t pos;
};
struct TbSegOut {
    int32_t x_own, k_own; // the walk's state at the top of row d_own ...
    int32_t x_end, k_end; // ... and at the top of row d_end - 1 (what the next walker's x_own / k_own must equal)
    int32_t lead, trail;  // gap columns before the first match run of the owned rows / behind the last (lib/align.c:542-545 counts
                          // across walkers: the stitch carries them)
    uint32_t rows;        // owned rows walked
    uint32_t flags;       // kTbReset | kTbAbort | kTbTerminal | kTbBad
};
enum : uint32_t {
    kTbReset = 1,         // a match run inside the owned rows (the gap counter restarted)
    kTbAbort = 2,         // more than 250 gap columns in a row inside the owned rows
    kTbTerminal = 4,      // reached the alignment's start
    kTbBad = 8,           // the walk ended where it must not (in the warm-up rows): the task is walked again in one piece
};
constexpr int kCkptCells = 128;
constexpr int32_t kTbRefused = 1 << 16;   // AlnOut::fin_idx: the stitch refused the task (host statistics) ...
constexpr int32_t kTbSeen = 1 << 17;      // ... of the tasks it saw

// ---- main-phase consensus on the device (msa_kernels.hip) ---------------------------------
// Packed alignment tag (reference: align_tag, lib/nextcorrect.h:28-32): t_pos+1 in bits
// [31:11] (so the head sentinel t_pos=-1,delta=0,base=0 packs to 0), delta in [10:3]
// (an insertion run is at most 251 columns, lib/align.c:542), base code in [2:0]
// (A0 T1 G2 C3 -4, lib/nextcorrect.c:52-62).
constexpr uint32_t kTagHead = 0u;
__host__ __device__ inline uint32_t tag_pack(int32_t t_pos, uint32_t delta, uint32_t base) {
    return ((uint32_t)(t_pos + 1) << 11) | (delta << 3) | base;
}
__host__ __device__ inline int32_t tag_tpos(uint32_t g) { return (int32_t)(g >> 11) - 1; }
__host__ __device__ inline uint32_t tag_delta(uint32_t g) { return (g >> 3) & 0xffu; }
__host__ __device__ inline uint32_t tag_base(uint32_t g) { return g & 7u; }

struct ReadDev {            // one per pile record (index 0 of a pile = the seed itself)
    int32_t task;           // index into the AlnTask/AlnOut tables, -1 for the seed
    uint32_t aln_start;     // inclusive seed window as handed to nextCorrect
    uint32_t aln_end;
    uint64_t tag_off;       // first tag slot of this read (uint32 units)
    uint64_t colidx_off;    // first column-index slot (uint32 units), capacity = window length
    // filled on the device
    uint32_t shift;         // first kept alignment column (get_align_shift)
    uint32_t aln_len;       // kept columns (0: dropped)
    uint32_t t_s, t_e;      // seed coordinates after trimming
    uint32_t q_start;       // query offset of column `shift`
    uint32_t accepted;      // passed min_len_aln and the coverage cut
    uint32_t pad_;
};

struct PileDev {
    uint32_t seed_len;
    uint32_t n_reads;       // records incl. the seed
    uint32_t first_read;    // index of the seed's ReadDev
    uint32_t min_len_aln;
    uint32_t max_cov_aln;
    int32_t factor;         // 3 (ont/clr) or 4 (hifi)   lib/nextcorrect.c:2147
    uint64_t seed_off;      // pool offset of the seed (bit 63: resident DB)
    uint64_t col_off;       // first slot of the per-column arrays (seed_len + 1 slots per pile)
    uint64_t acc_off;       // first slot of the accepted-read list (n_reads slots)
    // filled on the device
    uint32_t n_acc;         // accepted reads (incl. seed)
    uint32_t n_cells;       // 6 * sum(max_size)
    uint32_t n_tags;        // sum of kept columns = upper bound of link entries
    uint32_t path_len;
    // filled by the host between the two halves of the phase
    uint64_t cell_off;      // first cell of this pile
    uint64_t ent_off;       // first link entry of this pile
    uint64_t path_off;      // first path slot (capacity n_cells / 6)
    int32_t origin_t;       // backtrack origin written by the scoring kernel
    uint32_t origin_db;     // delta << 3 | base
    uint32_t err;           // nonzero: device-side capacity error
    uint32_t n_links;       // distinct (pp,ppp) links of the pile (written by the scoring kernel)
    // scoring segments (K10): filled by the host with the cell / link offsets
    uint32_t seg_off;       // index of the pile's first segment record
    uint32_t n_seg;
    uint32_t n_repair;      // segments the stitch kernel scored again (written by it)
    uint32_t tier;          // 1: scored with the large LDS tables (the column scan's verdict, fixed for the launch)
};

// ---- low-quality-region rounds (K12, lq_kernels.hip) ----
struct LqPieceDev {          // one (row, region) slot of a pile's round, row-major (30 rows x n_regions)
    int32_t task;            // index into the round's AlnTask / AlnOut tables, -1: nothing aligned ('M' row)
    uint32_t sl;             // length of the region's pseudo-seed
};
struct LqPileDev {
    uint32_t first_piece, n_regions;
    uint32_t link_len;       // columns of the linked pseudo-seed: 1 + sum(sl + 1)
    int32_t factor;          // 2 (HiFi 4)     lib/nextcorrect.c:1262
    int32_t qv_factor;       // 5 (HiFi 2)     lib/nextcorrect.c:1303
    uint32_t row_cap;        // capacity in cell rows ((column, delta) pairs) of the pile's cell records
    uint32_t out_cap;        // capacity of the pile's output characters
    uint32_t first_job, n_jobs;  // the pile's jobs (K12a), in column order
    uint32_t n_repair;       // (written by the stitch) jobs it scored again from their predecessor's true scores
    uint64_t cell_off;       // first cell record (6 per cell row); cell_off / 6 = the pile's first scratch character
    uint64_t out_off;        // first output character
    // written by the kernels
    uint32_t out_len;
    uint32_t err;            // nonzero: declined (capacity, or an alignment that does not end at both sequence ends)
    uint32_t n_rows;         // cell rows of the pile
    uint32_t pad_;
};
struct LqJobDev {            // K12a's unit of work: regions [g_a, g_b) of a pile, each with the 'N' column in front of it
    uint32_t pile;
    uint32_t g_a, g_b;
    uint32_t t0, t1;         // columns [t0, t1) of the linked pseudo-seed (t0 = the 'N' in front of region g_a; the last job ends behind the closing 'N')
    uint32_t row_cap, lnk_cap;   // capacity of the job's streams: cell-row headers, link words
    uint32_t row0;           // (stitch) the job's first cell row within the pile
    uint64_t hdr_off, lnk_off;
    // written by K12a
    uint32_t n_rows, n_links, err;
    // written by K12b (scores): links of the boundary column it started from / of its last column, the smallest offset-carrying
    // candidate and the largest absolute value it compared (raw), a capacity error
    uint32_t score_err;
    uint32_t n_spec, n_fin;
    int32_t mn, mx;
    // written by K12d (walk)
    uint32_t out_len, walk_err, walk_end, pad_;
};
constexpr int kLqLinkCap = 384;      // links per column the scoring tables (and a job's boundary planes) hold
// K12: links per job (lq_links), scores + best links per job from a speculative start (lq_score), boundary checks / repairs per pile
// (lq_stitch), walk per job (lq_walk), the pile's characters (lq_gather).  bnd: 4 x kLqLinkCap words per job; tmp_chars: one per cell row.
void launch_lq_msa(LqPileDev *piles, LqJobDev *jobs, const LqPieceDev *pieces, const AlnTask *tasks, const AlnOut *outs, const uint32_t *ops,
                   const uint32_t *pool, uint64_t *hdr, uint32_t *lnk, uint32_t *cell_rec, int32_t *bnd, char *tmp_chars, char *out_chars,
                   int n_piles, int n_jobs, uint32_t warm, uint32_t force_repair, void *stream);

// ---- scoring DP (K10), segment-parallel: see the head comment of the K10 section in msa_kernels.hip ----
struct SegItem {            // work item of the segment kernel
    uint32_t pile;
    uint32_t seg;
};
struct SegSum {             // one per (pile, segment)
    // written by the kernel that scored the segment
    int32_t vmin, amax;           // smallest offset-carrying / largest absolute link score from the column before the segment on
    int32_t bmax_rel, bmax_abs;   // largest cell best of the segment by kind (INT32_MIN: none)
    uint32_t links;               // distinct links of the segment's columns
    uint32_t n_fin;               // links of its last column (entries of `fin`)
    uint32_t flags;               // 1: a column does not fit the LDS tables, 2: a raw score left the int32 working range
    // written by the stitch kernel
    int32_t dlt;                  // constant between the predecessor's `fin` and this segment's `spec`
    uint32_t ok;                  // the boundary check passed
    uint32_t enc;                 // 1: raw scores above kRelThr carry `off`
    long long off;                // true score = raw + off
};
struct K10Args {
    PileDev *piles;
    const uint32_t *coverage, *max_size, *cell_base, *ent_base, *cell_start, *cell_len, *ent_pp, *ent_ppp, *ent_cnt;
    uint32_t *cell_best_pp, *cell_best_link;
    int32_t *cell_best;           // best score of every cell (raw)
    SegSum *sums;
    int32_t *spec, *fin;          // kSegEnts raw scores per segment: what it held for the column before its first / computed for its last
    uint32_t seg_len, warm;       // columns per segment, warm-up columns of a speculative segment (< seg_len)
    int32_t guard;                // raw scores beyond it send the pile to the int64 kernel
    uint32_t force_repair;        // test hook: every segment with index % force_repair == 1 is treated as failed
};
constexpr int kSegEnts = 512;     // = the large tier's link slots per column

struct PathItem {           // one visited cell of the best_pp walk (origin first)
    uint32_t tag;           // packed (t_pos, delta, base)
    uint16_t link;          // best_link_count of the cell
    uint16_t cov;           // coverage of the column
};

struct ColBlock {           // work item of the link-counting kernel
    uint32_t pile;
    uint32_t col0;
};

struct RegionDev {          // low-quality region whose candidate strings are wanted
    uint32_t pile;
    uint32_t start, end;    // inclusive seed columns
    uint32_t max_len;       // lqseq_max_length
    uint32_t max_len0;      // limit for the seed's own candidate (HiFi: DAG_MAX_LENGTH, else = max_len)
    // outputs
    uint32_t n_ok;          // candidates written (<= 40)
    uint32_t n_large;       // reads that exceeded max_len before the 40th candidate
    uint32_t cand_off[40];  // byte offsets into the string pool
    uint16_t cand_len[40];
    uint16_t cand_rank[40]; // position of the source read among the pile's aligned reads
};

constexpr int kColBlock = 32;          // columns per link-counting work item
constexpr int kLinkCap = 192;          // distinct (pp,ppp) links per cell held in LDS (<= 1.5 x max_cov_aln reads reach a cell)
constexpr int kLinkCapSmall = 64;      // capacity of the first attempt of the link-counting kernel

void launch_shift_scan(const AlnTask *tasks, const AlnOut *outs, const uint32_t *ops, ReadDev *reads, int n_reads,
                       void *stream);
void launch_pile_accept(PileDev *piles, ReadDev *reads, uint32_t *acc_list, uint32_t *cov_diff, int n_piles,
                        void *stream);
void launch_make_tags(const PileDev *piles, const ReadDev *reads, const AlnTask *tasks, const uint32_t *ops,
                      const uint32_t *pool, const uint32_t *db_pool, const uint32_t *read_pile, uint32_t *tags,
                      uint32_t *colidx, uint32_t *ins_count, uint32_t *ins_max, int n_reads, void *stream);
// in: cov_diff (difference array), ins_max; out (in place): coverage, max_size; plus offsets
void launch_col_scan(PileDev *piles, uint32_t *cov_diff, const uint32_t *ins_count, uint32_t *ins_max,
                     uint32_t *cell_base, uint32_t *ent_base, int n_piles, void *stream);
void launch_count_links(const PileDev *piles, const ReadDev *reads, const uint32_t *acc_list, const ColBlock *blocks,
                        const uint32_t *tags, const uint32_t *colidx, const uint32_t *max_size,
                        const uint32_t *cell_base, const uint32_t *ent_base, uint32_t *cell_start, uint32_t *cell_len,
                        uint32_t *ent_pp, uint32_t *ent_ppp, uint32_t *ent_cnt, uint32_t *err, int n_blocks, bool full_capacity,
                        void *stream);
// segment kernels of both table tiers (the large one on stream_large at the same time) -> stitch -> int64 kernel for
// the piles they left (err == 2) -> best_pp walk
// (ent_score == nullptr: the int64 kernel is not launched and leaves err == 2 piles for a second call with rescue = true,
// which launches only that kernel and the walk) then the best_pp walk, cut at the same segments: items_all = every (pile, segment) of the launch, bt_exit / bt_steps =
// kBtSlots entries per segment, bt_entry / bt_off = one per segment
void launch_score_backtrack(const K10Args &a, const SegItem *items_small, int n_small, const SegItem *items_large, int n_large,
                            const SegItem *items_all, int n_all, long long *ent_score, bool rescue, PathItem *path, uint32_t *bt_exit,
                            uint32_t *bt_steps, uint32_t *bt_entry, uint32_t *bt_off, int n_piles, void *stream,
                            void *ev_after_fast, void *stream_large, void *ev_fork, void *ev_join);
constexpr int kBtSlots = 192;
void launch_extract(const PileDev *piles, const ReadDev *reads, const uint32_t *acc_list, const uint32_t *tags,
                    const uint32_t *colidx, RegionDev *regions, char *strpool, unsigned long long *strpool_cursor,
                    unsigned long long strpool_cap, int n_regions, void *stream);

constexpr uint64_t kOffDb = 1ull << 63;   // offset flag: sequence lives in the resident read DB
constexpr uint64_t kOffMask = kOffDb - 1;

// Words of zeros behind the last sequence of every 2-bit pool (per-batch pools, the resident read DB): the forward kernel fetches 64
// bases -- five words -- at a time, starting anywhere up to the last base (csrc/ond_kernels.hip: fetch64_rel), and a walker of the
// traceback fills its window with the 16 words from a sequence's first on, however short the sequence (fetch64_win)
constexpr uint32_t kPoolPadWords = 24;
constexpr uint32_t kTracePadWords = 16;  // words behind the last task's trace: a walker's window holds 16 record words from the task's first on
constexpr int kFastRowWords = 2;       // register path: at most two 64-bit trace words per edit step (one up to 56 cells)
constexpr int kFastMaxBand = 238;      // register path: at most 120 same-parity diagonals per edit step (two per lane, 7-bit offsets)

// launchers (ond_kernels.hip)
void launch_ond_forward(const AlnTask *tasks, AlnOut *outs, const uint32_t *pool, const uint32_t *db_pool,
                        uint64_t *trace, int n_tasks, void *stream, const int32_t *order = nullptr);  // order: device, n_tasks ids, longest first
// The forward kernel with checkpoints + the traceback in segments (chase, walkers, stitch, and the one-lane walk for what the stitch
// refuses).  ck_cells: kCkptCells words per checkpoint slot, ck_hdr: one uint2 per slot (AlnTask::mink_off = a task's first slot,
// ((max_d - 1) >> cshift) slots each); segs / seg_outs: AlnTask::seg_off = a task's first walker slot, slots + 1 each; n_slots = all of them.
struct TbArgs {
    uint32_t *ck_cells;
    void *ck_hdr;
    TbSeg *segs;
    TbSegOut *seg_outs;
    int n_slots;
    int cshift, warm;
};
void launch_ond_forward_ckpt(const AlnTask *tasks, AlnOut *outs, const uint32_t *pool, const uint32_t *db_pool, uint64_t *trace,
                             uint32_t *ops, const TbArgs &tb, int n_tasks, void *stream, const int32_t *order);
void launch_ond_traceback_seg(const AlnTask *tasks, AlnOut *outs, const uint32_t *pool, const uint32_t *db_pool, const uint64_t *trace,
                              uint32_t *ops, const TbArgs &tb, int n_tasks, void *stream);
void launch_ond_forward_wide(const AlnTask *tasks, AlnOut *outs, const uint32_t *pool, const uint32_t *db_pool,
                             uint64_t *trace, int32_t *trace_mink, int32_t *vscratch, const int32_t *task_ids, int n_ids, void *stream,
                             const AlnTask *wtasks = nullptr, uint32_t max_ring = 0);
constexpr size_t kWideLdsBytes = 64u << 10;   // the wide path keeps V rings up to this size in LDS (a workgroup's default limit)  // wtasks: the listed tasks' records (wide-path fields), in list order
// task_ids == nullptr: tasks [0, n), traces in the register path's stream format, walked in the order `order` lists them
// (device, n ids; nullptr = table order); otherwise the listed (wide-band) tasks, traces in the wide kernel's row format
void launch_ond_traceback(const AlnTask *tasks, AlnOut *outs, const uint32_t *pool, const uint32_t *db_pool,
                          const uint64_t *trace,
                          const int32_t *trace_mink, uint32_t *ops, const int32_t *task_ids, int n, void *stream,
                          const int32_t *order = nullptr, const AlnTask *wtasks = nullptr);

}  // namespace ndgpu
