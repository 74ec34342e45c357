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
constexpr int kLqLinkCap = 384;      // links per column
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
// K12b: scores, best links, walk
__global__ __launch_bounds__(64) void lq_score_kernel(LqPileDev *__restrict__ piles, const LqJobDev *__restrict__ jobs,
                                                       const uint64_t *__restrict__ hdr_in, const uint32_t *__restrict__ lnk_in,
                                                       uint32_t *__restrict__ cell_rec, char *__restrict__ out_chars) {
    __shared__ uint32_t tab_lk[2][kLqLinkCap];   // the links of the current column and the one before ...
    __shared__ int32_t tab_sc[2][kLqLinkCap];    // ... and their scores
    __shared__ uint16_t cell_st[2][kLqDeltaCap * 6], cell_n[2][kLqDeltaCap * 6];
    __shared__ uint32_t win[kLqWalkRows * 6];    // the walk's window of cell records

    LqPileDev &PD = piles[blockIdx.x];
    const LqPileDev P = PD;
    const int lane = (int)threadIdx.x;
    if (P.link_len == 0 || P.n_jobs == 0) {
        if (lane == 0) PD.err = 9u, PD.out_len = 0u;
        return;
    }
    // one wavefront per pile and every cell row waits for the one before: the launch lasts as long as its longest pile, so these
    // wavefronts ask the SIMD's arbiter for priority over the other contexts' throughput kernels
    __builtin_amdgcn_s_setprio(3);
    uint32_t *__restrict__ rec = cell_rec + P.cell_off;
    uint32_t err = 0;
    for (uint32_t j = 0; j < P.n_jobs; j++) {  // (a job that gave up: the pile goes the host way)
        const uint32_t e = jobs[P.first_job + j].err;
        if (e && !err) err = e;
    }
    for (int i = lane; i < kLqDeltaCap * 6; i += 64) cell_n[0][i] = cell_n[1][i] = 0;
    __builtin_amdgcn_wave_barrier();

    uint32_t row = 0;                      // cell rows written so far
    uint32_t row_col0 = 0, row_prev0 = 0;  // first cell row of this column / of the one before
    int cur_tab = 1;
    uint32_t used_cur = 0, used_prev = 0;  // cell rows of the current / the other table's last use
    uint32_t n_tab = 0, d = 0;
    const int32_t factor = P.factor;

    for (uint32_t j = 0; j < P.n_jobs && !err; j++) {
        const LqJobDev J = jobs[P.first_job + j];
        const uint64_t *__restrict__ H = hdr_in + J.hdr_off;
        const uint32_t *__restrict__ L = lnk_in + J.lnk_off;
        uint32_t lb = 0;                                    // first link of the row
        uint64_t h_next = J.n_rows ? H[0] : 0ull;           // the row's header and links are fetched a row ahead:
        uint32_t w_next = J.n_rows ? L[lane] : 0u;          // (the stream carries 64 words of padding)
        for (uint32_t r = 0; r < J.n_rows; r++) {
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
            if (r + 1u < J.n_rows) {
                h_next = H[r + 1u];
                w_next = L[lb + n_row + (uint32_t)lane];
            }
            const uint32_t coverage = (uint32_t)(h >> 36) & 63u;
            if (h & kHdrD0) {  // a new column: the tables swap roles
                const uint32_t u = d + 1u < (uint32_t)kLqDeltaCap ? d + 1u : (uint32_t)kLqDeltaCap;
                if (row) {
                    used_cur = used_prev;
                    used_prev = u;
                }
                cur_tab ^= 1;
                row_prev0 = row_col0;
                row_col0 = row;
                for (uint32_t i = (uint32_t)lane; i < used_cur * 6u; i += 64) cell_n[cur_tab][i] = 0;
                __builtin_amdgcn_wave_barrier();
                n_tab = 0;
                d = 0;
            } else d++;
            if (d >= (uint32_t)kLqDeltaCap || row >= P.row_cap) {
                err = 4;
                break;
            }
            if (n_tab + n_row > (uint32_t)kLqLinkCap) {
                err = 5;
                break;
            }
            const int32_t penalty = factor * (int32_t)coverage;
            if (lane < 6) {
                cell_st[cur_tab][d * 6u + (uint32_t)lane] = (uint16_t)(n_tab + (lane == 0 ? st_cell[0] : lane == 1 ? st_cell[1] : lane == 2 ? st_cell[2]
                                                                                : lane == 3 ? st_cell[3] : lane == 4 ? st_cell[4] : st_cell[5]));
                cell_n[cur_tab][d * 6u + (uint32_t)lane] = (uint16_t)(lane == 0 ? n_cell[0] : lane == 1 ? n_cell[1] : lane == 2 ? n_cell[2]
                                                                     : lane == 3 ? n_cell[3] : lane == 4 ? n_cell[4] : n_cell[5]);
            }
            // ---- score every link of the row (nextcorrect.c:1273-1289): lane jj takes the row's jj-th link
            if ((uint32_t)lane < n_row) {
                const int32_t gain = 10 * (int32_t)(w & 63u) - penalty;
                int32_t sc;
                if (w & kLnkPpHead) sc = gain;
                else {
                    sc = 0;
                    const uint32_t before = (w >> 15) & 1u;
                    const int tb = before ? cur_tab ^ 1 : cur_tab;
                    const uint32_t ci = (w >> 6) & 511u;
                    const uint32_t s0 = cell_st[tb][ci], sn = cell_n[tb][ci];
                    // a link of pp's cell continues ppp when ITS pp is ppp: the same cell, as many columns back
                    const uint32_t want_head = w & kLnkPppHead;
                    const uint32_t want = (w >> 17) & 511u, want_before = ((w >> 26) & 3u) - before;
                    for (uint32_t q = s0; q < s0 + sn; q++) {
                        const uint32_t o = tab_lk[tb][q];
                        const bool m = want_head ? (o & kLnkPpHead) != 0u
                                                 : !(o & kLnkPpHead) && ((o >> 6) & 511u) == want && ((o >> 15) & 1u) == want_before;
                        if (m) {
                            const int32_t s2 = tab_sc[tb][q] + gain;
                            sc = s2 > sc ? s2 : sc;
                        }
                    }
                }
                tab_lk[cur_tab][n_tab + (uint32_t)lane] = w;
                tab_sc[cur_tab][n_tab + (uint32_t)lane] = sc;
            }
            __builtin_amdgcn_wave_barrier();
            // ---- every cell's best link, sequential tie-break (nextcorrect.c:1290-1296): lane bb owns cell bb
            if (lane < 6) {
                const uint32_t s0 = cell_st[cur_tab][d * 6u + (uint32_t)lane], n = cell_n[cur_tab][d * 6u + (uint32_t)lane];
                int32_t best = -10;
                uint32_t best_w = kLnkPpHead, best_link = 0;
                for (uint32_t k = 0; k < n; k++) {
                    const int32_t sc = tab_sc[cur_tab][s0 + k];
                    const uint32_t o = tab_lk[cur_tab][s0 + k];
                    const uint32_t pb = (o & kLnkPpHead) ? 0u : ((o >> 6) & 511u) % 6u;   // symbol of pp (the head's tag is 0)
                    if (sc > best || (sc == best && pb != 4u)) {
                        best = sc;
                        best_w = o;
                        best_link = o & 63u;
                    }
                }
                const uint32_t bci = (best_w >> 6) & 511u;
                const bool head = (best_w & kLnkPpHead) != 0u;
                const uint32_t prow = head ? kLqNoRow : (((best_w >> 15) & 1u) ? row_prev0 : row_col0) + bci / 6u;
                const uint32_t conf = (int32_t)best_link * P.qv_factor > (int32_t)coverage ? 8u : 0u;   // nextcorrect.c:1306
                rec[(uint64_t)row * 6u + (uint32_t)lane] = prow << 4 | conf | (head ? 0u : bci % 6u);
            }
            __builtin_amdgcn_wave_barrier();
            n_tab += n_row;
            lb += n_row;
            row++;
        }
    }
    const bool failed = __ballot(err != 0) != 0ull;
    uint32_t out_len = 0;
    if (!failed && row > 0 && row < kLqNoRow) {
        // ---- the walk (nextcorrect.c:1302-1318): from the last cell along the best predecessors, one character per non-gap cell.
        //      Every lane walks (the state is uniform), the records come through an LDS window of kLqWalkRows cell rows filled with
        //      coalesced loads -- a predecessor is always an earlier row -- and lane 0 writes the characters.
        __threadfence();
        char *__restrict__ out = out_chars + P.out_off;
        uint32_t wrow = row - 1u, wb = 5u, w_lo = row;
        for (;;) {
            if (wrow < w_lo) {
                const uint32_t hi = wrow + 1u, lo = hi > (uint32_t)kLqWalkRows ? hi - (uint32_t)kLqWalkRows : 0u;
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i = (uint32_t)lane; i < (hi - lo) * 6u; i += 64) win[i] = rec[(uint64_t)lo * 6u + i];
                __builtin_amdgcn_wave_barrier();
                w_lo = lo;
            }
            const uint32_t v = win[(wrow - w_lo) * 6u + wb];
            if (wb != 4u) {
                if (out_len >= P.out_cap) {
                    err = 8;
                    break;
                }
                const char ch = wb == 0u ? 'A' : wb == 1u ? 'T' : wb == 2u ? 'G' : wb == 3u ? 'C' : 'N';
                if (lane == 0) out[out_len] = ((v & 8u) || wb == 5u) ? ch : (char)(ch + 32);
                out_len++;
            }
            if ((v >> 4) == kLqNoRow) break;
            wrow = v >> 4;
            wb = v & 7u;
        }
    }
    const unsigned long long eb = __ballot(err != 0);
    const uint32_t e_first = (uint32_t)__shfl((int)err, eb ? __ffsll((long long)eb) - 1 : 0, 64);
    if (lane == 0) {
        PD.out_len = out_len;
        PD.err = e_first;
    }
}

}  // namespace

void launch_lq_msa(LqPileDev *piles, LqJobDev *jobs, const LqPieceDev *pieces, const AlnTask *tasks, const AlnOut *outs, const uint32_t *ops,
                   const uint32_t *pool, uint64_t *hdr, uint32_t *lnk, uint32_t *cell_rec, char *out_chars, int n_piles, int n_jobs,
                   void *stream) {
    if (n_piles <= 0) return;
    if (n_jobs > 0)
        hipLaunchKernelGGL(lq_links_kernel, dim3((unsigned)n_jobs), dim3(64), 0, (hipStream_t)stream, jobs, piles, pieces, tasks, outs, ops,
                           pool, hdr, lnk);
    hipLaunchKernelGGL(lq_score_kernel, dim3((unsigned)n_piles), dim3(64), 0, (hipStream_t)stream, piles, jobs, hdr, lnk, cell_rec, out_chars);
}

}  // namespace ndgpu
