// Low-quality-region rounds on the device (gfx950, wave64).
//
// The reference re-assembles every low-quality region of a seed from the pile: <= 30 candidate sequences per region are
// aligned to the region's pseudo-seed, the regions are joined with 'N' columns into one linked pseudo-seed, and a second,
// small MSA over these <= 30 rows is scored and walked (generate_consensus_trimed, lib/nextcorrect.c:1538-1669;
// get_lqseqs_from_align_tags, :1250-1338: six symbols, plain maximum, factor 2 / HiFi 4, origin = the last cell); twice
// per seed (iterate_generate_consensus_trimed, :1671-1715).  The alignments are on the device already (K7 / K8a); K12 keeps the
// rest here.  Until round 4 K12 was ONE wavefront per pile doing everything column by column (~300 instructions per cell row on a
// chain as long as the pile's regions together): the longest pile of a launch set the launch's length, and under eight contexts'
// load every instruction of that chain waited its turn.  What a cell row's links ARE does not depend on any score -- only which of
// them wins does -- and a region's links depend on the region before it through two tags per row, which the tail of that row's
// alignment gives away.  So:
//
// K12a lq_links  one wavefront per JOB = a run of regions of one pile (each with the 'N' column in front of it; the last job also
//             takes the closing 'N').  Lanes = rows (<= 30, row order = the reference's first-seen order of links).  A lane derives
//             its row's next tag on the fly from the 2-bit column kinds K8a left in HBM and the 2-bit candidate bases (no tag
//             arrays, no strings); per cell row (column, delta) the six cells' (pp, ppp) links are collected in first-seen order
//             with K9's ballot leader loop and written out: a header (links per cell, coverage) and one word per link -- count,
//             the cell of pp, the cell of ppp -- relative to the link's own column, so that the stream means the same wherever
//             the job sits.  Thousands of jobs per launch: throughput work.
// K12b lq_score  one wavefront per pile walks the jobs' streams in order: a link's score = the best score among the links of pp's
//             cell that continue ppp, + 10 x count - factor x coverage (two column tables in LDS), every cell's best link chosen
//             with the reference's sequential tie-break, one word per cell written (cell row and symbol of its best predecessor,
//             whether its own character is confident); then the wavefront walks the best predecessors from the last cell
//             through an LDS window and emits the consensus characters.  ~100 instructions per cell row, its inputs a sequential
//             stream fetched a row ahead.
// A pile that does not fit the LDS tables (an insertion run of >= 48 columns, > 384 links in a column) or whose alignments do
// not end at both sequence ends is declined (err != 0): the host path (consensus.cpp) takes it.
#include <hip/hip_runtime.h>

#include <climits>

#include "nd_device.h"

namespace ndgpu {

namespace {

constexpr int kLqRows = 30;          // LQSEQ_MAX_CAN_COUNT rows of the second MSA (lib/nextcorrect.h)
constexpr int kLqDeltaCap = 48;      // cell rows per column the LDS tables hold
constexpr int kLqCellCap = 32;       // links per cell (<= 30 rows)
constexpr int kLqWalkRows = 256;     // cell rows the walk stages in LDS at a time
// A cell's record, all the walk needs: the cell row and symbol of its best predecessor and whether its own character is
// confident -- [31:4] cell row (kLqNoRow: none, the walk ends), [3] best_link * qv_factor > coverage, [2:0] symbol.
constexpr uint32_t kLqNoRow = 0xfffffffu;

// A link of the stream K12a writes and K12b reads, one word: [5:0] count; [14:6] cell of pp (delta * 6 + symbol), [15] pp lies in
// the column before the link's own, [16] pp is the head (no predecessor); [25:17] cell of ppp, [27:26] how many columns before the
// link's own ppp lies (0..2), [28] ppp is the head.
constexpr uint32_t kLnkPpHead = 1u << 16, kLnkPppHead = 1u << 28;
// A cell row's header: [35:0] links per cell, six bits each; [41:36] coverage of the column; [42] first cell row of a column.
constexpr uint64_t kHdrD0 = 1ull << 42;

__device__ __forceinline__ uint32_t lq_op_at(const uint32_t *__restrict__ W, uint32_t col) {
    return (W[col >> 4] >> ((col & 15u) * 2u)) & 3u;
}
__device__ __forceinline__ uint32_t lq_code_at(const uint32_t *__restrict__ pool, uint64_t off) {
    return (pool[off >> 4] >> ((uint32_t)(off & 15u) * 2u)) & 3u;
}
// read code (A0 C1 G2 T3) -> consensus code (A0 T1 G2 C3, lib/nextcorrect.c:52-62)
__device__ __forceinline__ uint32_t lq_cns_code(uint32_t c) { return (0x1230u >> (c * 4u)) & 7u; }

// ---------------------------------------------------------------------------------------------------------------------------
// K12a: the links of a job's cell rows
__global__ __launch_bounds__(64) void lq_links_kernel(LqJobDev *__restrict__ jobs, const LqPileDev *__restrict__ piles,
                                                       const LqPieceDev *__restrict__ pieces, const AlnTask *__restrict__ tasks,
                                                       const AlnOut *__restrict__ outs, const uint32_t *__restrict__ ops,
                                                       const uint32_t *__restrict__ pool, uint64_t *__restrict__ hdr_out,
                                                       uint32_t *__restrict__ lnk_out) {
    __shared__ uint32_t l_pp[6][kLqCellCap], l_ppp[6][kLqCellCap], l_cnt[6][kLqCellCap];

    LqJobDev &JD = jobs[blockIdx.x];
    const LqJobDev J = JD;
    const LqPileDev P = piles[J.pile];
    const int lane = (int)threadIdx.x;
    if (P.link_len == 0) {  // a round the host did not lay out (it takes it itself)
        if (lane == 0) JD.err = 9u, JD.n_rows = 0u, JD.n_links = 0u;
        return;
    }
    const bool row_ok = lane < kLqRows;
    const LqPieceDev *__restrict__ my = pieces + P.first_piece + (uint32_t)(row_ok ? lane : 0) * P.n_regions;
    uint64_t *__restrict__ H = hdr_out + J.hdr_off;
    uint32_t *__restrict__ L = lnk_out + J.lnk_off;

    uint32_t err = 0;
    // the lane's row: current piece
    bool aligned = false;
    const uint32_t *W = nullptr;
    uint32_t col = 0, col_end = 0;
    uint64_t qo = 0;
    uint32_t p1 = kTagHead, p2 = kTagHead;  // the row's two previous tags
    // the word of 16 column kinds / 16 bases the row is reading, and the word after it
    uint32_t ops_wi = 0x7fffffffu, ops_w = 0, ops_nx = 0;
    uint64_t q_wi = ~0ull >> 1;
    uint32_t q_w = 0, q_nx = 0;
    auto op_peek = [&]() -> uint32_t {
        const uint32_t wi = col >> 4;
        if (wi != ops_wi) {
            ops_w = wi == ops_wi + 1u ? ops_nx : W[wi];
            ops_wi = wi;
            ops_nx = W[wi + 1u];
        }
        return (ops_w >> ((col & 15u) * 2u)) & 3u;
    };
    auto q_take = [&]() -> uint32_t {
        const uint64_t wi = qo >> 4;
        if (wi != q_wi) {
            q_w = wi == q_wi + 1ull ? q_nx : pool[wi];
            q_wi = wi;
            q_nx = pool[wi + 1ull];
        }
        const uint32_t c = (q_w >> ((uint32_t)(qo & 15u) * 2u)) & 3u;
        qo++;
        return lq_cns_code(c);
    };
    auto load_piece = [&](uint32_t g) {
        aligned = false;
        if (!row_ok) return;
        const LqPieceDev pc = my[g];
        if (pc.task < 0) return;
        const AlnOut O = outs[pc.task];
        if (O.status != ST_ALIGNED || O.n_cols <= 2) return;  // no alignment / the > 250-gap marker: an 'M' row (nextcorrect.c:1601)
        const AlnTask T = tasks[pc.task];
        if (O.x_final != T.q_len || O.y_final != T.t_len) {  // (the reference pads the unaligned tails; a finished sweep has none)
            err = 2;
            return;
        }
        aligned = true;
        W = ops + T.ops_off;
        col = T.ops_cap - (uint32_t)O.n_cols;
        col_end = T.ops_cap;
        qo = T.q_off & kOffMask;
        ops_wi = col >> 4;
        ops_w = W[ops_wi];
        ops_nx = W[ops_wi + 1u];
        q_wi = qo >> 4;
        q_w = pool[q_wi];
        q_nx = pool[q_wi + 1ull];
    };

    // ---- where the rows stand when the job begins: the last two tags of every row in the region before the job's first 'N'
    //      (lib/nextcorrect.c:1601-1640: a row has a tag in every column).  An aligned row's come from the tail of its column
    //      kinds: kind 0 = both bases, 1 = a candidate base hanging on the last column (delta = its place in the run), 2 = a gap.
    if (J.g_a > 0 && row_ok) {
        const uint32_t gp = J.g_a - 1u;
        const uint32_t t_end = J.t0 - 1u;       // last column of region gp (the host starts no job behind an empty region)
        const LqPieceDev pc = my[gp];
        bool al = false;
        AlnOut O;
        AlnTask T;
        if (pc.task >= 0) {
            O = outs[pc.task];
            if (O.status == ST_ALIGNED && O.n_cols > 2) {
                T = tasks[pc.task];
                if (O.x_final != T.q_len || O.y_final != T.t_len) err = 2;
                else al = true;
            }
        }
        if (al) {
            // the job before checks "no columns of the finished piece left over" only where an 'N' column follows INSIDE it; for the
            // region in front of this job that column is this job's first, so the count is taken here: the alignment must hold
            // exactly one kind that is not a hanging base (kind 1) per column of the region
            uint32_t cols_t = 0;
            const uint32_t *Wc = ops + T.ops_off;
            for (uint32_t c = T.ops_cap - (uint32_t)O.n_cols; c < T.ops_cap; c++) cols_t += lq_op_at(Wc, c) != 1u;
            if (cols_t != pc.sl) err = 7;
        }
        if (!al) {
            p1 = tag_pack((int32_t)t_end, 0u, 6u);
            p2 = pc.sl >= 2u ? tag_pack((int32_t)t_end - 1, 0u, 6u) : tag_pack((int32_t)t_end - 1, 0u, 5u);  // (the 'N' in front of a one-column region)
        } else {
            const uint32_t *Wp = ops + T.ops_off;
            const uint32_t c1 = T.ops_cap, c0 = T.ops_cap - (uint32_t)O.n_cols;  // kinds [c0, c1)
            const uint64_t q0 = T.q_off & kOffMask;
            uint32_t qn = (uint32_t)T.q_len;     // candidate bases not yet stepped over, from the end
            // the run of hanging bases the alignment ends with
            uint32_t r = 0;
            while (c1 - r > c0 && lq_op_at(Wp, c1 - 1u - r) == 1u && r <= (uint32_t)kLqDeltaCap) r++;
            if (r > (uint32_t)kLqDeltaCap) err = 4;
            else if (r >= 2u) {
                p1 = tag_pack((int32_t)t_end, r, lq_cns_code(lq_code_at(pool, q0 + qn - 1u)));
                p2 = tag_pack((int32_t)t_end, r - 1u, lq_cns_code(lq_code_at(pool, q0 + qn - 2u)));
            } else {
                uint32_t i = c1 - 1u;            // kind index of the last tag
                if (r == 1u) {
                    p1 = tag_pack((int32_t)t_end, 1u, lq_cns_code(lq_code_at(pool, q0 + qn - 1u)));
                    qn--, i--;
                    const uint32_t k = lq_op_at(Wp, i);  // the column's own tag (kind 0 or 2: the column exists)
                    p2 = tag_pack((int32_t)t_end, 0u, k == 2u ? 4u : lq_cns_code(lq_code_at(pool, q0 + qn - 1u)));
                } else {
                    const uint32_t k = lq_op_at(Wp, i);
                    p1 = tag_pack((int32_t)t_end, 0u, k == 2u ? 4u : lq_cns_code(lq_code_at(pool, q0 + qn - 1u)));
                    if (k == 0u) qn--;
                    i--;                         // (n_cols > 2: there is a kind before)
                    const uint32_t k2 = lq_op_at(Wp, i);
                    if (k2 != 1u) p2 = tag_pack((int32_t)t_end - 1, 0u, k2 == 2u ? 4u : lq_cns_code(lq_code_at(pool, q0 + qn - 1u)));
                    else {                       // a hanging base of the column before (the 'N' column if the region has one column)
                        uint32_t r2 = 0;
                        while (i + 1u - r2 > c0 && lq_op_at(Wp, i - r2) == 1u && r2 <= (uint32_t)kLqDeltaCap) r2++;
                        if (r2 > (uint32_t)kLqDeltaCap) err = 4;
                        p2 = tag_pack((int32_t)t_end - 1, r2, lq_cns_code(lq_code_at(pool, q0 + qn - 1u)));
                    }
                }
            }
        }
    }

    uint32_t row = 0, n_lnk = 0;  // cell rows / links written so far
    uint32_t t = J.t0;            // column
    uint32_t g = J.g_a, c_in = 0; // region coming up and column inside the current one
    bool sep = true;
    uint32_t sl_cur = 0;
    const uint32_t t_stop = J.t1;

    while (t < t_stop && !__ballot(err != 0)) {
        // -------- one column
        uint32_t coverage = 0;
        uint32_t d = 0;
        for (;; d++) {
            // ---- the lane's tag in cell row (t, d), if any
            bool has = false;
            uint32_t base = 0;
            if (d == 0) {
                has = row_ok;
                if (sep) base = 5;
                else if (aligned) {
                    const uint32_t op = col < col_end ? op_peek() : 1u;
                    if (op == 1u) err = 3;   // the row has no column for this target base
                    else {
                        base = op == 0u ? q_take() : 4u;
                        col++;
                    }
                } else base = 6;
            } else if (aligned && col < col_end && op_peek() == 1u) {
                has = true;
                base = q_take();
                col++;
            }
            if (d > 0 && !__ballot(has)) break;
            if (d >= (uint32_t)kLqDeltaCap || row >= J.row_cap) {
                err = 4;
                break;
            }
            const uint32_t cur = tag_pack((int32_t)t, d, base);
            uint32_t pp = kTagHead, ppp = kTagHead;
            if (has) {
                pp = p1;
                ppp = p2;
                p2 = p1;
                p1 = cur;
            }
            if (d == 0) coverage = (uint32_t)__popcll(__ballot(has && base != 6u));  // nextcorrect.c:1525
            const bool counted = has && base != 6u && (pp & 7u) != 6u;   // update_msa skips 'M' tags (nextcorrect.c:222)

            // ---- links of the six cells, first-seen order (K9's leader loop, all six cells in one pass: a link is (base, pp,
            //      ppp), the earliest row that carries an unseen one leads, its cell's list grows by one)
            unsigned long long cnt6 = 0;  // links per cell so far, 8 bits each (the same in every lane)
            {
                unsigned long long rem = __ballot(counted);
                while (rem) {
                    const int ld = __ffsll((long long)rem) - 1;
                    const uint32_t kb = (uint32_t)__builtin_amdgcn_readlane((int)base, ld);
                    const uint32_t kp = (uint32_t)__builtin_amdgcn_readlane((int)pp, ld);
                    const uint32_t kpp = (uint32_t)__builtin_amdgcn_readlane((int)ppp, ld);
                    const bool in_rem = (rem >> lane) & 1ull;
                    const unsigned long long same = __ballot(in_rem && base == kb && pp == kp && ppp == kpp);
                    const uint32_t n0 = (uint32_t)(cnt6 >> (8u * kb)) & 0xffu;
                    if (lane == ld) {
                        l_pp[kb][n0] = kp;
                        l_ppp[kb][n0] = kpp;
                        l_cnt[kb][n0] = (uint32_t)__popcll(same);
                    }
                    cnt6 += 1ull << (8u * kb);
                    rem &= ~same;
                }
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t n_cell[6], st_cell[6];
            uint32_t n_row = 0;
#pragma unroll
            for (uint32_t bb = 0; bb < 6; bb++) {
                n_cell[bb] = (uint32_t)(cnt6 >> (8u * bb)) & 0xffu;
                st_cell[bb] = n_row;
                n_row += n_cell[bb];
            }
            if (n_lnk + n_row > J.lnk_cap) {
                err = 5;
                break;
            }
            // ---- the row's stream: header by lane 0, lane j the row's j-th link (cells in order)
            if (lane == 0)
                H[row] = (uint64_t)n_cell[0] | (uint64_t)n_cell[1] << 6 | (uint64_t)n_cell[2] << 12 | (uint64_t)n_cell[3] << 18 |
                         (uint64_t)n_cell[4] << 24 | (uint64_t)n_cell[5] << 30 | (uint64_t)coverage << 36 | (d == 0 ? kHdrD0 : 0ull);
            if ((uint32_t)lane < n_row) {
                const uint32_t j = (uint32_t)lane;
                const uint32_t bb = (j >= st_cell[1]) + (j >= st_cell[2]) + (j >= st_cell[3]) + (j >= st_cell[4]) + (j >= st_cell[5]);
                // (an empty cell shares its start with the next one: the comparisons step over it)
                const uint32_t k = j - (bb == 0 ? st_cell[0] : bb == 1 ? st_cell[1] : bb == 2 ? st_cell[2] : bb == 3 ? st_cell[3]
                                                 : bb == 4 ? st_cell[4] : st_cell[5]);
                const uint32_t mpp = l_pp[bb][k], mppp = l_ppp[bb][k];
                uint32_t w = l_cnt[bb][k] & 63u;
                if (mpp == kTagHead) w |= kLnkPpHead | kLnkPppHead;
                else {
                    const uint32_t pt = (uint32_t)tag_tpos(mpp);
                    if (pt != t && pt + 1u != t) err = 6;  // (every row has a tag in every column: cannot happen)
                    const uint32_t ci = tag_delta(mpp) * 6u + tag_base(mpp);
                    if (ci >= (uint32_t)kLqDeltaCap * 6u) err = 4;
                    w |= ci << 6 | (pt == t ? 0u : 1u << 15);
                    if (mppp == kTagHead) w |= kLnkPppHead;
                    else {
                        const uint32_t qt = (uint32_t)tag_tpos(mppp);
                        const uint32_t back = t - qt;
                        const uint32_t cj = tag_delta(mppp) * 6u + tag_base(mppp);
                        if (back > 2u) err = 6;
                        if (cj >= (uint32_t)kLqDeltaCap * 6u) err = 4;
                        w |= cj << 17 | (back & 3u) << 26;
                    }
                }
                L[n_lnk + j] = w;
            }
            __builtin_amdgcn_wave_barrier();
            n_lnk += n_row;
            row++;
            // ---- an 'N' column ends a region: the rows move on to their pieces of the next one, whose leading insertions
            //      hang on this column
            if (d == 0 && sep) {
                if (aligned && col != col_end) err = 7;  // columns of the finished piece left over
                if (g < P.n_regions) {
                    load_piece(g);
                    sl_cur = pieces[P.first_piece + g].sl;
                } else aligned = false;
            }
        }
        // ---- next column
        t++;
        if (sep) {
            sep = false;
            c_in = 0;
            if (sl_cur == 0) {  // an empty pseudo-seed: its closing 'N' follows at once
                sep = true;
                g++;
            }
        } else if (++c_in == sl_cur) {
            sep = true;
            g++;
        }
    }
    const unsigned long long eb = __ballot(err != 0);
    const uint32_t e_first = (uint32_t)__shfl((int)err, eb ? __ffsll((long long)eb) - 1 : 0, 64);
    if (lane == 0) {
        JD.n_rows = row;
        JD.n_links = n_lnk;
        JD.err = e_first;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// K12b: scores and best links -- one wavefront per JOB (the same jobs K12a cut: runs of regions of ~192 columns)
//
// A link's score needs the scores of the column before, so a pile's columns are one chain -- until round 4 one wavefront walked it from
// the pile's first column to its last (10 ms per launch at ~1 % of the device, and the last kernel of every step).  The chain is cut
// the way K10 cuts the main scoring DP (msa_kernels.hip): job j > 0 does not know the scores its first column reads, so it starts
// `warm` columns early -- on the tail of job j - 1's stream, which K12a wrote position-independent -- with every unknown predecessor
// score = one constant C.  Within a few dozen columns the best paths of all links of a column have a common ancestor, and from there on
// the scores it holds differ from the true ones by ONE constant; every decision of this DP compares scores with scores.  Nothing of that
// is assumed: the job stores the scores it held for the last column before its own (`spec`) and the ones it computed for its last
// column (`fin`), and lq_stitch_kernel checks job by job that `spec` is the predecessor's `fin` up to one constant (and equal outright
// for the scores that carry no constant: a head link scores its gain, a link nobody continues scores 0, lib/nextcorrect.c:1273-1289),
// and that no comparison of an offset-carrying score with an absolute one (the floor at 0, an absolute link, the -10 a cell's best
// starts from) could have gone the other way: the smallest true offset-carrying candidate of the job exceeds its largest absolute one.
// A job that fails is scored again in the stitch kernel from its predecessor's true scores (NDGPU_K12_FORCE=repair forces that for
// every second job; NDGPU_K12_WARM sets the warm-up).
constexpr uint32_t kLnkAbs = 1u << 31;   // (table copy of a link word) its score carries no constant
constexpr int32_t kLqSpecC = 1 << 29;    // the constant a speculative start gives every unknown predecessor score
constexpr int32_t kLqNone = INT_MIN;

struct LqLds {                            // the scoring tables of one wavefront
    uint32_t lk[2][kLqLinkCap];           // the links of the current column and the one before ...
    int32_t sc[2][kLqLinkCap];            // ... and their scores
    uint16_t cst[2][kLqDeltaCap * 6], cn[2][kLqDeltaCap * 6];
};
struct LqScoreSt {                        // wave-uniform state of a scoring pass (+ two per-lane trackers)
    uint32_t row = 0;                     // global cell row of the next stream row
    uint32_t row_col0 = 0, row_prev0 = 0; // first cell row of this column / of the one before
    int cur_tab = 1;
    uint32_t used_cur = 0, used_prev = 0; // cell rows of the current / the other table's last use
    uint32_t n_tab = 0, d = 0;
    bool any = false;                     // a column has been seen
    bool spec_pending = false, spec_col = false;  // the pass starts without a column before it / this is that first column
    uint32_t err = 0;
    int32_t mn = INT_MAX, mx = 0;         // per lane: smallest offset-carrying candidate, largest absolute value (the floor included)
};

__device__ __forceinline__ void lq_tables_clear(LqLds &T, int lane) {
    for (int i = lane; i < kLqDeltaCap * 6; i += 64) T.cn[0][i] = T.cn[1][i] = 0;
    __builtin_amdgcn_wave_barrier();
}

// Rows [r0, r1) of a job's stream (headers H, links L, `lb` = the first link of row r0).  WRITE: the cell records go out (a job's own
// rows); otherwise the rows only build the tables (warm-up).
template <bool WRITE>
__device__ __forceinline__ void lq_score_rows(LqLds &T, LqScoreSt &S, const uint64_t *__restrict__ H, const uint32_t *__restrict__ L,
                                              uint32_t r0, uint32_t r1, uint32_t lb, uint32_t *__restrict__ rec, int32_t factor,
                                              int32_t qv_factor, int lane) {
    if (r0 >= r1) return;
    uint64_t h_next = H[r0];                      // the row's header and links are fetched a row ahead
    uint32_t w_next = L[lb + (uint32_t)lane];     // (the stream carries 64 words of padding)
    for (uint32_t r = r0; r < r1; r++) {
        const uint64_t h = h_next;
        const uint32_t w = w_next;
        uint32_t n_cell[6], st_cell[6];
        uint32_t n_row = 0;
#pragma unroll
        for (uint32_t bb = 0; bb < 6; bb++) {
            n_cell[bb] = (uint32_t)(h >> (6u * bb)) & 63u;
            st_cell[bb] = n_row;
            n_row += n_cell[bb];
        }
        if (r + 1u < r1) {
            h_next = H[r + 1u];
            w_next = L[lb + n_row + (uint32_t)lane];
        }
        const uint32_t coverage = (uint32_t)(h >> 36) & 63u;
        if (h & kHdrD0) {  // a new column: the tables swap roles
            const uint32_t u = S.d + 1u < (uint32_t)kLqDeltaCap ? S.d + 1u : (uint32_t)kLqDeltaCap;
            if (S.any) {
                S.used_cur = S.used_prev;
                S.used_prev = u;
            }
            S.any = true;
            S.cur_tab ^= 1;
            S.row_prev0 = S.row_col0;
            S.row_col0 = S.row;
            for (uint32_t i = (uint32_t)lane; i < S.used_cur * 6u; i += 64) T.cn[S.cur_tab][i] = 0;
            __builtin_amdgcn_wave_barrier();
            S.n_tab = 0;
            S.d = 0;
            S.spec_col = S.spec_pending;
            S.spec_pending = false;
        } else S.d++;
        if (S.d >= (uint32_t)kLqDeltaCap) {
            S.err = 4;
            break;
        }
        if (S.n_tab + n_row > (uint32_t)kLqLinkCap) {
            S.err = 5;
            break;
        }
        const int cur = S.cur_tab;
        const uint32_t d = S.d, n_tab = S.n_tab;
        const int32_t penalty = factor * (int32_t)coverage;
        if (lane < 6) {
            T.cst[cur][d * 6u + (uint32_t)lane] = (uint16_t)(n_tab + (lane == 0 ? st_cell[0] : lane == 1 ? st_cell[1] : lane == 2 ? st_cell[2]
                                                                      : lane == 3 ? st_cell[3] : lane == 4 ? st_cell[4] : st_cell[5]));
            T.cn[cur][d * 6u + (uint32_t)lane] = (uint16_t)(lane == 0 ? n_cell[0] : lane == 1 ? n_cell[1] : lane == 2 ? n_cell[2]
                                                           : lane == 3 ? n_cell[3] : lane == 4 ? n_cell[4] : n_cell[5]);
        }
        // ---- score every link of the row (nextcorrect.c:1273-1289): lane jj takes the row's jj-th link
        if ((uint32_t)lane < n_row) {
            const int32_t gain = 10 * (int32_t)(w & 63u) - penalty;
            int32_t sc;
            bool ab;
            const uint32_t before = (w >> 15) & 1u;
            if (w & kLnkPpHead) sc = gain, ab = true;
            else if (S.spec_col && before) {  // the column before the pass's first: unknown, every score there = C
                sc = kLqSpecC + gain;
                ab = false;
                S.mn = sc < S.mn ? sc : S.mn;
            } else {
                int32_t b_off = kLqNone, b_abs = kLqNone;
                const int tb = before ? cur ^ 1 : cur;
                const uint32_t ci = (w >> 6) & 511u;
                const uint32_t s0 = T.cst[tb][ci], sn = T.cn[tb][ci];
                // a link of pp's cell continues ppp when ITS pp is ppp: the same cell, as many columns back
                const uint32_t want_head = w & kLnkPppHead;
                const uint32_t want = (w >> 17) & 511u, want_before = ((w >> 26) & 3u) - before;
                for (uint32_t q = s0; q < s0 + sn; q++) {
                    const uint32_t o = T.lk[tb][q];
                    const bool m = want_head ? (o & kLnkPpHead) != 0u
                                             : !(o & kLnkPpHead) && ((o >> 6) & 511u) == want && ((o >> 15) & 1u) == want_before;
                    if (m) {
                        const int32_t s2 = T.sc[tb][q] + gain;
                        if (o & kLnkAbs) b_abs = s2 > b_abs ? s2 : b_abs;
                        else {
                            b_off = s2 > b_off ? s2 : b_off;
                            S.mn = s2 < S.mn ? s2 : S.mn;
                        }
                    }
                }
                if (b_off != kLqNone) {        // (an offset-carrying candidate wins over the floor and over every absolute one -- checked by the stitch)
                    sc = b_off;
                    ab = false;
                    if (b_abs > S.mx) S.mx = b_abs;
                } else {
                    sc = b_abs > 0 ? b_abs : 0;
                    ab = true;
                }
            }
            if (ab && sc > S.mx) S.mx = sc;
            T.lk[cur][n_tab + (uint32_t)lane] = (w & ~kLnkAbs) | (ab ? kLnkAbs : 0u);
            T.sc[cur][n_tab + (uint32_t)lane] = sc;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- every cell's best link, sequential tie-break (nextcorrect.c:1290-1296): lane bb owns cell bb
        if (WRITE && lane < 6) {
            const uint32_t s0 = T.cst[cur][d * 6u + (uint32_t)lane], n = T.cn[cur][d * 6u + (uint32_t)lane];
            int32_t best = -10;
            uint32_t best_w = kLnkPpHead, best_link = 0;
            for (uint32_t k = 0; k < n; k++) {
                const int32_t sc = T.sc[cur][s0 + k];
                const uint32_t o = T.lk[cur][s0 + k];
                const uint32_t pb = (o & kLnkPpHead) ? 0u : ((o >> 6) & 511u) % 6u;   // symbol of pp (the head's tag is 0)
                if (sc > best || (sc == best && pb != 4u)) {
                    best = sc;
                    best_w = o;
                    best_link = o & 63u;
                }
            }
            const uint32_t bci = (best_w >> 6) & 511u;
            const bool head = (best_w & kLnkPpHead) != 0u;
            const uint32_t prow = head ? kLqNoRow : (((best_w >> 15) & 1u) ? S.row_prev0 : S.row_col0) + bci / 6u;
            const uint32_t conf = (int32_t)best_link * qv_factor > (int32_t)coverage ? 8u : 0u;   // nextcorrect.c:1306
            rec[(uint64_t)S.row * 6u + (uint32_t)lane] = prow << 4 | conf | (head ? 0u : bci % 6u);
        }
        __builtin_amdgcn_wave_barrier();
        S.n_tab += n_row;
        lb += n_row;
        S.row++;
    }
}

// Where the last `warm` columns of a job's stream begin: (row, first link of that row); (0, 0) when the job holds no more columns.
__device__ __forceinline__ void lq_warm_start(const uint64_t *__restrict__ H, uint32_t n_rows, uint32_t n_links, uint32_t warm, int lane,
                                              uint32_t &row_out, uint32_t &lnk_out) {
    uint32_t cols = 0, links_behind = 0;  // columns / links of the rows looked at so far (from the end)
    uint32_t hi = n_rows;                 // rows [hi, n_rows) are looked at
    row_out = 0, lnk_out = 0;
    while (hi > 0) {
        const uint32_t lo = hi > 64u ? hi - 64u : 0u;
        const uint32_t r = hi - 1u - (uint32_t)lane;      // lane 0 = the last row of the window
        const bool in = (uint32_t)lane < hi - lo;
        const uint64_t h = in ? H[r] : 0ull;
        uint32_t nl = 0;
#pragma unroll
        for (uint32_t bb = 0; bb < 6; bb++) nl += (uint32_t)(h >> (6u * bb)) & 63u;
        const unsigned long long d0 = __ballot(in && (h & kHdrD0));
        const uint32_t have = (uint32_t)__popcll(d0);
        if (cols + have >= warm) {   // the column that completes the warm-up starts in this window: its D0 row is the (warm - cols)-th set bit
            unsigned long long m = d0;
            for (uint32_t k = 1; k < warm - cols; k++) m &= m - 1ull;
            const int cut = __ffsll((long long)m) - 1;   // lane of that row
            // links of the rows from the cut on: lanes 0..cut of this window + everything behind
            uint32_t v = (lane <= cut) ? nl : 0u;
            for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
            row_out = hi - 1u - (uint32_t)cut;
            lnk_out = n_links - (links_behind + v);
            return;
        }
        cols += have;
        uint32_t v = nl;
        for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
        links_behind += v;
        hi = lo;
    }
}

// global cell row of a job's first row = the rows of the pile's jobs before it
__device__ __forceinline__ uint32_t lq_rows_before(const LqJobDev *__restrict__ jobs, uint32_t first_job, uint32_t jb, int lane) {
    uint32_t v = 0;
    for (uint32_t k = first_job + (uint32_t)lane; k < jb; k += 64) v += jobs[k].n_rows;
    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
    return v;
}

__device__ __forceinline__ int32_t lq_wave_min(int32_t v) {
    for (int off = 32; off > 0; off >>= 1) {
        const int32_t o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ int32_t lq_wave_max(int32_t v) {
    for (int off = 32; off > 0; off >>= 1) {
        const int32_t o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// a job's boundary planes: [spec scores | spec flags | fin scores | fin flags], kLqLinkCap words each
__device__ __forceinline__ int32_t *lq_plane(int32_t *bnd, uint32_t jb, int which) { return bnd + ((uint64_t)jb * 4u + (uint32_t)which) * kLqLinkCap; }

__device__ __forceinline__ void lq_dump_table(const LqLds &T, int tab, uint32_t n, int32_t *__restrict__ sc, int32_t *__restrict__ fl, int lane) {
    for (uint32_t i = (uint32_t)lane; i < n; i += 64) {
        sc[i] = T.sc[tab][i];
        fl[i] = (T.lk[tab][i] & kLnkAbs) ? 1 : 0;
    }
}

__global__ __launch_bounds__(64) void lq_score_kernel(const LqPileDev *__restrict__ piles, LqJobDev *__restrict__ jobs,
                                                       const uint64_t *__restrict__ hdr_in, const uint32_t *__restrict__ lnk_in,
                                                       uint32_t *__restrict__ cell_rec, int32_t *__restrict__ bnd, uint32_t warm) {
    __shared__ LqLds T;
    const uint32_t jb = blockIdx.x;
    LqJobDev &JD = jobs[jb];
    const LqJobDev J = JD;
    const LqPileDev P = piles[J.pile];
    const int lane = (int)threadIdx.x;
    if (P.link_len == 0 || P.n_jobs == 0) return;  // (K12a marked the jobs of such a pile: err 9)
    const bool first = jb == P.first_job;
    if (J.err || (!first && jobs[jb - 1u].err)) {   // the pile is declined (the stitch reports it)
        if (lane == 0) JD.n_spec = JD.n_fin = 0u, JD.score_err = 0u;
        return;
    }
    uint32_t *__restrict__ rec = cell_rec + P.cell_off;
    const uint32_t row0 = lq_rows_before(jobs, P.first_job, jb, lane);
    LqScoreSt S;
    lq_tables_clear(T, lane);
    uint32_t n_spec = 0;
    if (!first) {
        // ---- warm-up on the tail of the job before
        const LqJobDev Jp = jobs[jb - 1u];
        uint32_t wr = 0, wl = 0;
        lq_warm_start(hdr_in + Jp.hdr_off, Jp.n_rows, Jp.n_links, warm, lane, wr, wl);
        S.row = row0 - (Jp.n_rows - wr);
        S.spec_pending = !(wr == 0u && jb - 1u == P.first_job);  // (a warm-up that reaches the pile's first column is exact)
        lq_score_rows<false>(T, S, hdr_in + Jp.hdr_off, lnk_in + Jp.lnk_off, wr, Jp.n_rows, wl, rec, P.factor, P.qv_factor, lane);
        n_spec = S.n_tab;
        lq_dump_table(T, S.cur_tab, n_spec, lq_plane(bnd, jb, 0), lq_plane(bnd, jb, 1), lane);
        S.mn = INT_MAX, S.mx = 0;   // (what the warm-up compared decides nothing that is kept)
    }
    if (!S.err) lq_score_rows<true>(T, S, hdr_in + J.hdr_off, lnk_in + J.lnk_off, 0u, J.n_rows, 0u, rec, P.factor, P.qv_factor, lane);
    lq_dump_table(T, S.cur_tab, S.n_tab, lq_plane(bnd, jb, 2), lq_plane(bnd, jb, 3), lane);
    const int32_t mn = lq_wave_min(S.mn), mx = lq_wave_max(S.mx);
    if (lane == 0) {
        JD.n_spec = n_spec;
        JD.n_fin = S.n_tab;
        JD.mn = mn;
        JD.mx = mx;
        JD.score_err = S.err;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// K12c: the boundary checks, job by job (one wavefront per pile), and the repair of a job that fails one
__global__ __launch_bounds__(64) void lq_stitch_kernel(LqPileDev *__restrict__ piles, LqJobDev *__restrict__ jobs,
                                                        const uint64_t *__restrict__ hdr_in, const uint32_t *__restrict__ lnk_in,
                                                        uint32_t *__restrict__ cell_rec, int32_t *__restrict__ bnd, uint32_t force_repair) {
    __shared__ LqLds T;
    LqPileDev &PD = piles[blockIdx.x];
    const LqPileDev P = PD;
    const int lane = (int)threadIdx.x;
    if (P.link_len == 0 || P.n_jobs == 0) {
        if (lane == 0) PD.err = 9u, PD.out_len = 0u, PD.n_rows = 0u;
        return;
    }
    uint32_t err = 0;
    for (uint32_t j = 0; j < P.n_jobs; j++) {  // (a job that gave up: the pile goes the host way)
        const uint32_t e = jobs[P.first_job + j].err;
        if (e && !err) err = e;
    }
    uint32_t rows = 0, repairs = 0;
    long long off_prev = 0;   // true score = raw + off for the offset-carrying scores of the job before
    for (uint32_t j = 0; j < P.n_jobs && !err; j++) {
        const uint32_t jb = P.first_job + j;
        const LqJobDev J = jobs[jb];
        if (J.score_err) {
            err = J.score_err;
            break;
        }
        long long off = 0;
        if (j > 0) {
            const LqJobDev Jp = jobs[jb - 1u];
            const int32_t *fs = lq_plane(bnd, jb - 1u, 2), *ff = lq_plane(bnd, jb - 1u, 3), *ss = lq_plane(bnd, jb, 0), *sf = lq_plane(bnd, jb, 1);
            bool ok = Jp.n_fin == J.n_spec;
            bool have = false;
            long long dl = 0;
            if (ok) {
                // true score of every link of the boundary column by the job before / as the job held it
                bool bad = false;
                for (uint32_t i0 = 0; i0 < J.n_spec; i0 += 64) {
                    const uint32_t i = i0 + (uint32_t)lane;
                    const bool in = i < J.n_spec;
                    const long long tp = in ? (long long)fs[i] + (ff[i] ? 0ll : off_prev) : 0ll;
                    const bool s_abs = in && sf[i] != 0;
                    if (s_abs && tp != (long long)ss[i]) bad = true;
                    const unsigned long long rel = __ballot(in && !s_abs);
                    if (rel) {
                        const int ld = __ffsll((long long)rel) - 1;
                        const long long mine = tp - (long long)ss[in ? i : 0];
                        const long long d0 = (long long)(((unsigned long long)(uint32_t)__shfl((int)(mine >> 32), ld, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)mine, ld, 64));
                        if (!have) dl = d0, have = true;
                        if (in && !s_abs && mine != dl) bad = true;
                    }
                }
                ok = !__ballot(bad);
            }
            off = have ? dl : 0ll;
            // no comparison of an offset-carrying candidate with an absolute value went the other way in truth
            if (ok && J.mn != INT_MAX && !((long long)J.mn + off > (long long)J.mx)) ok = false;
            if (force_repair && (j % force_repair) == 1u) ok = false;
            if (!ok) {
                // ---- repair: the boundary column's links once more (their layout), the true scores of the job before put in their
                //      place, then the job's own rows -- exact, every score absolute
                repairs++;
                LqScoreSt S;
                lq_tables_clear(T, lane);
                uint32_t wr = 0, wl = 0;
                lq_warm_start(hdr_in + Jp.hdr_off, Jp.n_rows, Jp.n_links, 1u, lane, wr, wl);
                S.row = rows - (Jp.n_rows - wr);
                S.spec_pending = true;
                lq_score_rows<false>(T, S, hdr_in + Jp.hdr_off, lnk_in + Jp.lnk_off, wr, Jp.n_rows, wl, cell_rec + P.cell_off, P.factor, P.qv_factor, lane);
                if (S.err || S.n_tab != Jp.n_fin) err = S.err ? S.err : 10u;
                else {
                    for (uint32_t i = (uint32_t)lane; i < S.n_tab; i += 64) {
                        T.sc[S.cur_tab][i] = (int32_t)((long long)fs[i] + (ff[i] ? 0ll : off_prev));
                        T.lk[S.cur_tab][i] |= kLnkAbs;
                    }
                    __builtin_amdgcn_wave_barrier();
                    S.mn = INT_MAX, S.mx = 0;
                    lq_score_rows<true>(T, S, hdr_in + J.hdr_off, lnk_in + J.lnk_off, 0u, J.n_rows, 0u, cell_rec + P.cell_off, P.factor, P.qv_factor, lane);
                    if (S.err) err = S.err;
                    lq_dump_table(T, S.cur_tab, S.n_tab, lq_plane(bnd, jb, 2), lq_plane(bnd, jb, 3), lane);
                    __threadfence();
                    if (lane == 0) jobs[jb].n_fin = S.n_tab;
                    __builtin_amdgcn_wave_barrier();
                }
                off = 0;
            }
        }
        if (lane == 0) jobs[jb].row0 = rows;
        off_prev = off;
        rows += J.n_rows;
    }
    if (lane == 0) {
        PD.err = err;
        PD.n_rows = rows;
        PD.n_repair = repairs;
        if (err) PD.out_len = 0u;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// K12d: the walk (nextcorrect.c:1302-1318), one wavefront per job.  Every row's tags pass through the 'N' cell of every 'N' column (a
// row's tag after (t, d) is (t, d + 1) or (t + 1, 0), and (t, 0) of an 'N' column is the 'N' for every row), so the best-predecessor
// walk from the pile's last cell visits the 'N' cell in front of every job: a job's stretch of it starts at the best predecessor of the
// next job's first cell (the pile's last cell for the last job) and ends with its own first cell.  The characters go to the job's
// stretch of a scratch array in walk order; lq_gather_kernel strings the stretches together, last job first.
__global__ __launch_bounds__(64) void lq_walk_kernel(const LqPileDev *__restrict__ piles, LqJobDev *__restrict__ jobs,
                                                      const uint32_t *__restrict__ cell_rec, char *__restrict__ tmp_chars) {
    __shared__ uint32_t win[kLqWalkRows * 6];    // the walk's window of cell records
    const uint32_t jb = blockIdx.x;
    LqJobDev &JD = jobs[jb];
    const LqJobDev J = JD;
    const LqPileDev P = piles[J.pile];
    const int lane = (int)threadIdx.x;
    if (P.link_len == 0 || P.n_jobs == 0 || P.err) {
        if (lane == 0) JD.out_len = 0u, JD.walk_err = 0u, JD.walk_end = 0u;
        return;
    }
    const uint32_t *__restrict__ rec = cell_rec + P.cell_off;
    char *__restrict__ out = tmp_chars + P.cell_off / 6u + J.row0;
    const uint32_t first = J.row0, end = J.row0 + J.n_rows;
    uint32_t err = 0, out_len = 0;
    uint32_t wrow, wb;
    bool ended = false;   // the walk found a cell without a predecessor here: nothing before it belongs to the consensus
    if (jb + 1u == P.first_job + P.n_jobs) wrow = end - 1u, wb = 5u;   // the origin: the pile's last cell
    else {
        const uint32_t v0 = rec[(uint64_t)end * 6u + 5u];               // the 'N' cell in front of the next job
        wrow = v0 >> 4, wb = v0 & 7u;
        if (wrow == kLqNoRow) ended = true, wrow = first;               // (no row reaches that 'N' with a counted link: the walk ends there)
        else if (wrow >= end || wrow < first) err = 11, wrow = first;
    }
    uint32_t w_lo = end;
    // Every lane walks (the state is uniform), the records come through an LDS window of kLqWalkRows cell rows filled with coalesced
    // loads -- a predecessor is always an earlier row -- and lane 0 writes the characters.
    while (!err && !ended) {
        if (wrow < w_lo) {
            const uint32_t hi = wrow + 1u, lo = hi > first + (uint32_t)kLqWalkRows ? hi - (uint32_t)kLqWalkRows : first;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t i = (uint32_t)lane; i < (hi - lo) * 6u; i += 64) win[i] = rec[(uint64_t)lo * 6u + i];
            __builtin_amdgcn_wave_barrier();
            w_lo = lo;
        }
        const uint32_t v = win[(wrow - w_lo) * 6u + wb];
        if (wb != 4u) {
            if (out_len >= J.n_rows) {
                err = 8;
                break;
            }
            const char ch = wb == 0u ? 'A' : wb == 1u ? 'T' : wb == 2u ? 'G' : wb == 3u ? 'C' : 'N';
            if (lane == 0) out[out_len] = ((v & 8u) || wb == 5u) ? ch : (char)(ch + 32);
            out_len++;
        }
        if ((v >> 4) == kLqNoRow) {                 // no predecessor (the pile's first cell; an 'N' no row reaches with a counted link):
            ended = true;                           // the walk ends (nextcorrect.c:1302-1318)
            break;
        }
        if (wrow == first && wb == 5u) break;      // the job's own first cell: the stretch ends (its predecessor is the job before's)
        wrow = v >> 4;
        wb = v & 7u;
        if (wrow < first) err = 11;                 // left the job without meeting its first cell: cannot happen
    }
    if (lane == 0) JD.out_len = out_len, JD.walk_err = err, JD.walk_end = ended ? 1u : 0u;
}

// K12e: the pile's characters = its jobs' stretches, last job first (walk order)
__global__ __launch_bounds__(64) void lq_gather_kernel(LqPileDev *__restrict__ piles, const LqJobDev *__restrict__ jobs,
                                                        const char *__restrict__ tmp_chars, char *__restrict__ out_chars) {
    LqPileDev &PD = piles[blockIdx.x];
    const LqPileDev P = PD;
    const int lane = (int)threadIdx.x;
    if (P.link_len == 0 || P.n_jobs == 0 || P.err) return;
    uint32_t err = 0;
    uint64_t total = 0;
    uint32_t j_stop = 0;   // the job the walk ends in
    for (uint32_t j = P.n_jobs; j-- > 0;) {
        const LqJobDev &J = jobs[P.first_job + j];
        if (J.walk_err && !err) err = J.walk_err;
        total += J.out_len;
        if (J.walk_end) {
            j_stop = j;
            break;
        }
    }
    if (!err && total > P.out_cap) err = 8;
    if (err) {
        if (lane == 0) PD.err = err, PD.out_len = 0u;
        return;
    }
    char *__restrict__ out = out_chars + P.out_off;
    uint32_t at = 0;
    for (uint32_t j = P.n_jobs; j-- > j_stop;) {
        const LqJobDev &J = jobs[P.first_job + j];
        const char *__restrict__ src = tmp_chars + P.cell_off / 6u + J.row0;
        for (uint32_t i = (uint32_t)lane; i < J.out_len; i += 64) out[at + i] = src[i];
        at += J.out_len;
    }
    if (lane == 0) PD.out_len = at;
}

}  // namespace

void launch_lq_msa(LqPileDev *piles, LqJobDev *jobs, const LqPieceDev *pieces, const AlnTask *tasks, const AlnOut *outs, const uint32_t *ops,
                   const uint32_t *pool, uint64_t *hdr, uint32_t *lnk, uint32_t *cell_rec, int32_t *bnd, char *tmp_chars, char *out_chars,
                   int n_piles, int n_jobs, uint32_t warm, uint32_t force_repair, void *stream) {
    if (n_piles <= 0) return;
    hipStream_t st = (hipStream_t)stream;
    if (n_jobs > 0) {
        hipLaunchKernelGGL(lq_links_kernel, dim3((unsigned)n_jobs), dim3(64), 0, st, jobs, piles, pieces, tasks, outs, ops, pool, hdr, lnk);
        hipLaunchKernelGGL(lq_score_kernel, dim3((unsigned)n_jobs), dim3(64), 0, st, piles, jobs, hdr, lnk, cell_rec, bnd, warm ? warm : 1u);
    }
    hipLaunchKernelGGL(lq_stitch_kernel, dim3((unsigned)n_piles), dim3(64), 0, st, piles, jobs, hdr, lnk, cell_rec, bnd, force_repair);
    if (n_jobs > 0) hipLaunchKernelGGL(lq_walk_kernel, dim3((unsigned)n_jobs), dim3(64), 0, st, piles, jobs, cell_rec, tmp_chars);
    hipLaunchKernelGGL(lq_gather_kernel, dim3((unsigned)n_piles), dim3(64), 0, st, piles, jobs, tmp_chars, out_chars);
}

}  // namespace ndgpu
