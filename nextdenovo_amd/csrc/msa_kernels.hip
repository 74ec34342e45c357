// Main-phase consensus kernels for gfx950 (wave64): everything between the O(ND)
// traceback and the O(L) low-quality-region logic runs on the device, so neither the
// alignment columns nor the ~10^6 tags of a pile ever cross PCIe.
//
// K8s shift_scan     get_align_shift(aln, 8)           lib/nextcorrect.c:102-154
//     pile_accept    min_len_aln + coverage cut         lib/nextcorrect.c:2271,2289-2292
// K8b make_tags      get_align_tags                     lib/nextcorrect.c:1485-1536
//     col_scan       per-column coverage / max_size / cell and link offsets
//                    (allocate_msa_mem, lib/nextcorrect.c:175-198)
// K9  count_links    update_msa                         lib/nextcorrect.c:212-250
// K10 score_backtrack scoring DP + global pick + best_pp walk
//                                                        lib/nextcorrect.c:2149-2202, 1907-1982
// K11 extract        candidate strings of low-quality regions
//                                                        lib/nextcorrect.c:373-404
//
// Bit-exactness notes: a (column, delta) slot holds at most one tag per read, so the
// reference's first-seen order of (pp,ppp) links inside a cell is the order of the
// reads that carry them; K9 keeps lanes in pile order and elects link leaders with
// ctz(ballot), which reproduces that order.  K10 keeps the reference's sequential
// tie-break state per cell (one lane per base symbol).
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "nd_device.h"

namespace ndgpu {

namespace {

__device__ __forceinline__ uint32_t op_at(const uint32_t *__restrict__ W, uint32_t col) {
    return (W[col >> 4] >> ((col & 15u) * 2u)) & 3u;
}
__device__ __forceinline__ uint32_t code_at(const uint32_t *__restrict__ pool, uint64_t off) {
    return (pool[off >> 4] >> ((uint32_t)(off & 15u) * 2u)) & 3u;
}
// read-DB code (A0 C1 G2 T3, lib/bseq.c:11-20) -> consensus code (A0 T1 G2 C3, lib/nextcorrect.c:52-62)
__device__ __forceinline__ uint32_t cns_code(uint32_t c) { return (0x1230u >> (c * 4u)) & 7u; }

__device__ __forceinline__ unsigned long long lanes_le(int lane) {  // bits 0..lane
    return lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
}

// ---- K8s -------------------------------------------------------------------------
__global__ __launch_bounds__(64) void shift_scan_kernel(const AlnTask *__restrict__ tasks,
                                                         const AlnOut *__restrict__ outs,
                                                         const uint32_t *__restrict__ ops, ReadDev *__restrict__ reads,
                                                         int n_reads) {
    const int i = (int)(blockIdx.x * 64 + threadIdx.x);
    if (i >= n_reads) return;
    ReadDev &R = reads[i];
    R.aln_len = 0;
    R.accepted = 0;
    if (R.task < 0) return;  // the seed: handled by pile_accept
    const AlnOut o = outs[R.task];
    if (o.status != ST_ALIGNED) return;
    const AlnTask T = tasks[R.task];
    const uint32_t *W = ops + T.ops_off;
    const uint32_t n = (uint32_t)o.n_cols, c0 = T.ops_cap - n;
    int run = 0;
    uint32_t tc = 0, qc = 0, j;
    bool found = false;
    for (j = 0; j < n; j++) {
        const uint32_t op = op_at(W, c0 + j);
        run = op == 0 ? run + 1 : 0;
        tc += op != 1u;
        qc += op != 2u;
        if (run == 8) {
            found = true;
            break;
        }
    }
    if (!found) return;
    const uint32_t shift = j - 7;
    run = 0;
    uint32_t tb = 0;
    int jj;
    for (jj = (int)n - 1; jj >= 0; jj--) {
        const uint32_t op = op_at(W, c0 + (uint32_t)jj);
        run = op == 0 ? run + 1 : 0;
        tb += op != 1u;
        if (run == 8) break;
    }
    R.shift = c0 + shift;  // absolute column inside the task's ops region
    R.aln_len = (uint32_t)(jj + 7) - shift + 1;
    R.t_s = R.aln_start + tc - 8;
    R.t_e = R.aln_end - tb + 8;
    R.q_start = qc - 8;
}

__global__ __launch_bounds__(64) void pile_accept_kernel(PileDev *__restrict__ piles, ReadDev *__restrict__ reads,
                                                          uint32_t *__restrict__ acc_list,
                                                          uint32_t *__restrict__ cov_diff, int n_piles) {
    const int i = (int)(blockIdx.x * 64 + threadIdx.x);
    if (i >= n_piles) return;
    PileDev &P = piles[i];
    int total = 0;
    uint32_t nacc = 0, ntags = 0;
    const int L = (int)P.seed_len;
    for (uint32_t r = 0; r < P.n_reads; r++) {
        if ((uint32_t)(total / L) > P.max_cov_aln) break;  // lib/nextcorrect.c:2271
        ReadDev &R = reads[P.first_read + r];
        if (r == 0) {  // the seed aligned to itself (lib/nextcorrect.c:2279-2282)
            R.shift = 0;
            R.aln_len = P.seed_len;
            R.t_s = R.aln_start;
            R.t_e = R.aln_end;
            R.q_start = 0;
        }
        if (R.aln_len >= P.min_len_aln) {
            total += (int)(R.t_e - R.t_s + 1);
            R.accepted = 1;
            acc_list[P.acc_off + nacc++] = P.first_read + r;
            ntags += R.aln_len;
            // coverage as a difference array: every accepted read has exactly one delta-0 tag
            // on each column of [t_s, t_e]
            atomicAdd(&cov_diff[P.col_off + R.t_s], 1u);
            atomicAdd(&cov_diff[P.col_off + R.t_e + 1], 0xffffffffu);
        }
    }
    P.n_acc = nacc;
    P.n_tags = ntags;
}

// ---- K8b -------------------------------------------------------------------------
__global__ __launch_bounds__(64) void make_tags_kernel(const PileDev *__restrict__ piles,
                                                        const ReadDev *__restrict__ reads,
                                                        const AlnTask *__restrict__ tasks,
                                                        const uint32_t *__restrict__ ops,
                                                        const uint32_t *__restrict__ pool,
                                                        const uint32_t *__restrict__ db_pool,
                                                        const uint32_t *__restrict__ read_pile,
                                                        uint32_t *__restrict__ tags, uint32_t *__restrict__ colidx,
                                                        uint32_t *__restrict__ ins_count,
                                                        uint32_t *__restrict__ ins_max) {
    const int r = (int)blockIdx.x;
    const ReadDev R = reads[r];
    if (!R.accepted) return;
    const PileDev P = piles[read_pile[r]];
    const int lane = (int)threadIdx.x;
    uint32_t *tg = tags + R.tag_off;
    uint32_t *ci = colidx + R.colidx_off;
    if (R.task < 0) {
        const uint32_t *sp = (P.seed_off >> 63) ? db_pool : pool;
        const uint64_t so = P.seed_off & kOffMask;
        for (uint32_t t = (uint32_t)lane; t < R.aln_len; t += 64) {
            tg[t] = tag_pack((int32_t)(R.t_s + t), 0, cns_code(code_at(sp, so + t)));
            ci[t] = t;
        }
        return;
    }
    const AlnTask T = tasks[R.task];
    const uint32_t *W = ops + T.ops_off;
    const uint32_t *qp = (T.q_off >> 63) ? db_pool : pool;
    const uint64_t qo = T.q_off & kOffMask;
    uint32_t *icnt = ins_count + P.col_off;
    uint32_t *imax = ins_max + P.col_off;
    int32_t carry_t = (int32_t)R.t_s - 1;
    uint32_t carry_delta = 0, carry_q = R.q_start;
    // The column words and the query's bases come 64 words at a time -- one word per lane, 1,024 columns or bases, handed to the
    // lane that needs them by a shuffle.  Loaded per 64 columns they put two loads into every round of the loop, and a load's
    // s_waitcnt also waits for every store and atomic issued before it (one counter, in order): each round then waited for the
    // atomics of the round before it to come back from memory.
    const uint32_t w_last = (R.shift + R.aln_len - 1u) >> 4;
    const uint64_t q_last = (qo + (uint64_t)(uint32_t)T.q_len - 1u) >> 4;
    uint32_t w_base = 0, opw = 0, qw = 0;
    uint64_t q_base = 0;
    bool have_w = false, have_q = false;
    for (uint32_t c0 = 0; c0 < R.aln_len; c0 += 64) {
        const uint32_t c = c0 + (uint32_t)lane;
        const bool valid = c < R.aln_len;
        if (!have_w || ((R.shift + c0 + 63u) >> 4) > w_base + 63u) {
            w_base = (R.shift + c0) >> 4;
            opw = w_base + (uint32_t)lane <= w_last ? W[w_base + (uint32_t)lane] : 0u;
            ND_LOADED(opw);
            have_w = true;
        }
        const uint32_t col = R.shift + c;
        const uint32_t word = (uint32_t)__shfl((int)opw, (int)((col >> 4) - w_base), 64);
        const uint32_t op = valid ? (word >> ((col & 15u) * 2u)) & 3u : 0u;
        const bool is_t = valid && op != 1u, is_q = valid && op != 2u;
        const unsigned long long mt = __ballot(is_t), mq = __ballot(is_q);
        const unsigned long long le = lanes_le(lane);
        const int32_t t_pos = carry_t + (int32_t)__popcll(mt & le);
        const unsigned long long mm = mt & le;
        const uint32_t delta = mm ? (uint32_t)(lane - (63 - __clzll((long long)mm))) : carry_delta + (uint32_t)lane + 1u;
        const uint32_t qidx = carry_q + (uint32_t)__popcll(mq & (le >> 1));
        if (!have_q || ((qo + carry_q + 63u) >> 4) > q_base + 63u) {
            q_base = (qo + carry_q) >> 4;
            qw = q_base + (uint64_t)lane <= q_last ? qp[q_base + (uint64_t)lane] : 0u;
            ND_LOADED(qw);
            have_q = true;
        }
        const uint64_t q_at = qo + qidx;
        const uint32_t qword = (uint32_t)__shfl((int)qw, (int)((q_at >> 4) - q_base), 64);
        if (valid) {
            const uint32_t base = is_q ? cns_code((qword >> ((uint32_t)(q_at & 15u) * 2u)) & 3u) : 4u;
            tg[c] = tag_pack(t_pos, delta, base);
            if (delta == 0) ci[(uint32_t)t_pos - R.t_s] = c;
            else if (lane == 63 || c + 1u == R.aln_len || ((mt >> (lane + 1)) & 1ull)) {
                // An insertion run is counted once, by its last column (of these 64): device-scope atomics are served behind the
                // L2s, one memory transaction each.  (Looking at the maximum first to skip its atomic would be a load again.)
                atomicAdd(&icnt[t_pos], mm ? delta : (uint32_t)lane + 1u);
                atomicMax(&imax[t_pos], delta + 1u);
            }
        }
        const uint32_t nv = R.aln_len - c0 < 64u ? R.aln_len - c0 : 64u;
        carry_delta = (uint32_t)__shfl((int)delta, (int)nv - 1, 64);
        carry_t += (int32_t)__popcll(mt);
        carry_q += (uint32_t)__popcll(mq);
    }
}

constexpr int kColCellsSmall = 96, kColEntsSmall = 256;   // small scoring tables (K10): max_size <= 16, <= 256 link slots per column

// ---- column scan: coverage, max_size, cell/link offsets ------------------------------
__global__ __launch_bounds__(64) void col_scan_kernel(PileDev *__restrict__ piles, uint32_t *__restrict__ cov_diff,
                                                       const uint32_t *__restrict__ ins_count,
                                                       uint32_t *__restrict__ ins_max, uint32_t *__restrict__ cell_base,
                                                       uint32_t *__restrict__ ent_base) {
    PileDev &P = piles[blockIdx.x];
    const int lane = (int)threadIdx.x;
    const uint32_t L = P.seed_len;
    uint32_t *cov = cov_diff + P.col_off;       // in: difference array, out: coverage
    const uint32_t *icnt = ins_count + P.col_off;
    uint32_t *ms = ins_max + P.col_off;         // in: max(delta+1) of insertion tags, out: max_size
    uint32_t *cb = cell_base + P.col_off;
    uint32_t *eb = ent_base + P.col_off;
    uint32_t run_cov = 0, run_cells = 0, run_ents = 0;
    uint32_t big = 0;  // a column of this lane needs the large scoring tables
    // The three inputs of the next 64 columns are asked for before this round's stores and waited for behind them (ND_LOADED at the
    // end of the round: s_waitcnt vmcnt(4), the four stores stay in flight).  A load's wait covers everything issued before it, so
    // with the loads at the top every round waited for its predecessor's stores to be acknowledged.  The rounds of 64 whole columns
    // store without a condition (the compiler can count them); the seed's last columns are a round of their own.
    // (Columns past the seed's end read its last column instead -- a load under a condition meets its default at a join, and the
    // join waits; what such a lane computes is never stored, and no lane above it in a prefix sum is.)
    const uint32_t t_last = L ? L - 1u : 0u;
    uint32_t n_cov, n_ms, n_ic;
    {
        const uint32_t tn = (uint32_t)lane < t_last ? (uint32_t)lane : t_last;
        n_cov = cov[tn], n_ms = ms[tn], n_ic = icnt[tn];
        ND_LOADED(n_cov);
        ND_LOADED(n_ms);
        ND_LOADED(n_ic);
    }
    auto round = [&](uint32_t t0, auto whole) {
        constexpr bool kWhole = decltype(whole)::value;
        const uint32_t t = t0 + (uint32_t)lane;
        const bool v = kWhole || t < L;
        uint32_t c = n_cov;
        const uint32_t ms_in = n_ms, ic_in = n_ic;
        if (kWhole) {
            const uint32_t tn = t + 64u < t_last ? t + 64u : t_last;
            n_cov = cov[tn], n_ms = ms[tn], n_ic = icnt[tn];
        }
        // inclusive wave prefix sums
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)c, o, 64);
            if (lane >= o) c += u;
        }
        c += run_cov;
        uint32_t m = 0, e = 0;
        if (v) {
            m = c ? (ms_in > 1u ? ms_in : 1u) : 0u;
            e = c + ic_in;
        }
        if (m * 6u > (uint32_t)kColCellsSmall || e > (uint32_t)kColEntsSmall) big = 1;
        uint32_t pc = m * 6u, pe = e;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u1 = (uint32_t)__shfl_up((int)pc, o, 64);
            const uint32_t u2 = (uint32_t)__shfl_up((int)pe, o, 64);
            if (lane >= o) {
                pc += u1;
                pe += u2;
            }
        }
        if (v) {
            cov[t] = c;
            ms[t] = m;
            cb[t] = run_cells + pc - m * 6u;
            eb[t] = run_ents + pe - e;
        }
        run_cov = (uint32_t)__shfl((int)c, 63, 64);
        run_cells += (uint32_t)__shfl((int)pc, 63, 64);
        run_ents += (uint32_t)__shfl((int)pe, 63, 64);
        if (kWhole) {
            ND_LOADED(n_cov);
            ND_LOADED(n_ms);
            ND_LOADED(n_ic);
        }
    };
    uint32_t t0 = 0;
    for (; t0 + 64u <= L; t0 += 64) round(t0, std::true_type{});
    if (t0 < L) round(t0, std::false_type{});
    const bool any_big = __ballot(big != 0) != 0ull;
    if (lane == 0) {
        cb[L] = run_cells;
        eb[L] = run_ents;
        P.n_cells = run_cells;
        P.err = any_big ? 3u : 0u;  // 3: scored with the large LDS tables (score_fast_kernel tiers)
    }
}

// ---- K9 --------------------------------------------------------------------------
// CAP = links per cell the LDS lists hold.  The launch tries the small capacity first (4.6 KB of LDS per wavefront instead
// of 13.8 KB: the kernel is latency-bound and lives on resident wavefronts); a cell that overflows raises err[0] and the
// host repeats the sub-batch with kLinkCap.  Seven wavefronts per SIMD for the small capacity (72 VGPRs, 36 B of scratch per lane):
// 148 -> 127 ms per config-2 step with the kernel alone on the device; at 6 (80 VGPRs) 133 ms, at 8 (64 VGPRs, 76 B of scratch) 134 ms.
// Round 6: the second chunk of reads lost its register window (it was what spilled), and at EIGHT wavefronts per SIMD the kernel now takes
// 63 registers with 12 bytes of scratch per lane and column: 124.2 (seven, no scratch) -> 118.8 ms.
template <int CAP>
__global__ __launch_bounds__(64, CAP <= 64 ? 8 : 3) void count_links_kernel(const PileDev *__restrict__ piles,
                                                          const ReadDev *__restrict__ reads,
                                                          const uint32_t *__restrict__ acc_list,
                                                          const ColBlock *__restrict__ blocks,
                                                          const uint32_t *__restrict__ tags,
                                                          const uint32_t *__restrict__ colidx,
                                                          const uint32_t *__restrict__ max_size,
                                                          const uint32_t *__restrict__ cell_base,
                                                          const uint32_t *__restrict__ ent_base,
                                                          uint32_t *__restrict__ cell_start,
                                                          uint32_t *__restrict__ cell_len, uint32_t *__restrict__ ent_pp,
                                                          uint32_t *__restrict__ ent_ppp, uint32_t *__restrict__ ent_cnt,
                                                          uint32_t *__restrict__ err) {
    __shared__ uint32_t l_pp[6][CAP], l_ppp[6][CAP], l_cnt[6][CAP];
    const ColBlock B = blocks[blockIdx.x];
    const PileDev P = piles[B.pile];
    const int lane = (int)threadIdx.x;
    const uint32_t *ms = max_size + P.col_off;
    const uint32_t *cb = cell_base + P.col_off;
    const uint32_t *eb = ent_base + P.col_off;
    const uint32_t *acc = acc_list + P.acc_off;
    const uint32_t t_end = B.col0 + kColBlock < P.seed_len ? B.col0 + kColBlock : P.seed_len;

    // Per-lane read descriptors of the first kRegChunks x 64 accepted reads stay in registers for
    // the whole column block; deeper piles reload the rest from HBM.
    constexpr int kRegChunks = 2;
    constexpr int kWinChunks = 1;   // of which keep a 32-byte window of their tag stream in registers
    uint32_t g_ts[kRegChunks], g_te[kRegChunks], g_len[kRegChunks];
    const uint32_t *g_ci[kRegChunks];
    uint32_t g_tg[kRegChunks];  // first tag slot of the read (an index into `tags`: the table of a sub-batch holds < 2^32 slots)
#pragma unroll
    for (int ch = 0; ch < kRegChunks; ch++) {
        const uint32_t rank = (uint32_t)ch * 64u + (uint32_t)lane;
        g_ts[ch] = 1;
        g_te[ch] = 0;  // empty interval: never covers a column
        g_len[ch] = 0;
        g_ci[ch] = nullptr;
        g_tg[ch] = 0;
        if (rank < P.n_acc) {
            const ReadDev *R = &reads[acc[rank]];
            g_ts[ch] = R->t_s;
            g_te[ch] = R->t_e;
            g_len[ch] = R->aln_len;
            g_ci[ch] = colidx + R->colidx_off;
            g_tg[ch] = (uint32_t)R->tag_off;
        }
    }
    const uint32_t n_chunks = (P.n_acc + 63u) / 64u;
    // A register-resident read is scanned strictly in tag order (column by column, delta by delta), so its stream is read
    // once, 32 bytes at a time, and the two previous tags of every tag are simply carried along.  Which (column, delta) a tag
    // belongs to is written in the tag itself, so the column index of the read is consulted once per column block (where the
    // read's stream enters the block) and not once per column: a lane's loads stay inside one tag stream, and a 128-byte line
    // of it is asked for four times instead of eight (the lines of 64 lanes x the resident wavefronts do not fit the caches,
    // and every further request was another trip to HBM: 18 GB of fetches per launch for ~2.6 GB of tables).
    uint4 w_lo[kRegChunks], w_hi[kRegChunks];
    uint32_t w_i[kRegChunks], w_p1[kRegChunks], w_p2[kRegChunks], w_pos[kRegChunks];
#pragma unroll
    for (int ch = 0; ch < kRegChunks; ch++) {
        w_lo[ch] = w_hi[ch] = make_uint4(0, 0, 0, 0);
        w_i[ch] = 0xffffffffu;  // no window yet
        w_p1[ch] = w_p2[ch] = kTagHead;
        w_pos[ch] = 0xffffffffu;  // next tag of the read: none inside this block
        const uint32_t t_first = B.col0 > g_ts[ch] ? B.col0 : g_ts[ch];
        if (t_first < t_end && t_first <= g_te[ch]) w_pos[ch] = g_ci[ch][t_first - g_ts[ch]];
    }

    for (uint32_t t = B.col0; t < t_end; t++) {
        const uint32_t width = ms[t];
        uint64_t e = P.ent_off + eb[t];
        for (uint32_t d = 0; d < width; d++) {
            uint32_t n_cell[6] = {0, 0, 0, 0, 0, 0};  // links collected so far in the six cells of (t, d): the same in every lane
            for (uint32_t chn = 0; chn < n_chunks; chn++) {
                bool has = false;
                uint32_t cur = 0, pp = kTagHead, ppp = kTagHead;
                uint32_t i0 = 1, nx = 0;
                const uint32_t *tg = nullptr;
                if (chn < (uint32_t)kRegChunks) {
                    const uint32_t key = ((t + 1u) << 8) | d;  // tag >> 3 of a tag of cell row (t, d)
#pragma unroll
                    for (int ch = 0; ch < kRegChunks; ch++)
                        if ((uint32_t)ch == chn) {
                            const uint32_t i = w_pos[ch];
                            if (i < g_len[ch]) {
                                const uint32_t *tp = tags + g_tg[ch];
                                if (w_i[ch] == 0xffffffffu) {  // first tag of this read inside the column block
                                    if (i > 0) w_p1[ch] = tp[i - 1];
                                    if (i > 1) w_p2[ch] = tp[i - 2];
                                }
                                uint32_t c;
                                if (ch < kWinChunks) {
                                    if (w_i[ch] == 0xffffffffu || i - w_i[ch] >= 8u) {
                                        w_lo[ch] = *reinterpret_cast<const uint4 *>(tp + i);
                                        w_hi[ch] = *reinterpret_cast<const uint4 *>(tp + i + 4);
                                        w_i[ch] = i;
                                    }
                                    const uint32_t k = i - w_i[ch];
                                    const uint4 w = k < 4u ? w_lo[ch] : w_hi[ch];
                                    const uint32_t k4 = k & 3u;
                                    c = k4 == 0 ? w.x : k4 == 1 ? w.y : k4 == 2 ? w.z : w.w;
                                } else {
                                    // the second 64 reads of a deep pile: the next tag straight from the stream (it sits in the line the tag
                                    // before it came from).  A 32-byte window for this chunk too was 8 registers more than the kernel's 72
                                    // hold: the compiler spilled it around every column -- 32 bytes of scratch per lane and column written
                                    // and read back, 8.9 GB of the kernel's 10.5 GB of writes per launch by the counters.
                                    w_i[ch] = i;
                                    c = tp[i];
                                }
                                if ((c >> 3) == key) {  // the read's next tag sits in this cell row: consume it
                                    cur = c;
                                    pp = w_p1[ch];
                                    ppp = w_p2[ch];
                                    w_p2[ch] = w_p1[ch];
                                    w_p1[ch] = c;
                                    w_pos[ch] = i + 1;
                                    has = true;
                                }
                            }
                        }
                } else {
                    const uint32_t rank = chn * 64u + (uint32_t)lane;
                    if (rank < P.n_acc) {
                        const ReadDev *R = &reads[acc[rank]];
                        const uint32_t ts = R->t_s, te = R->t_e;
                        if (t >= ts && t <= te) {
                            const uint32_t *ci = colidx + R->colidx_off;
                            i0 = ci[t - ts];
                            nx = t == te ? R->aln_len : ci[t + 1 - ts];
                            tg = tags + R->tag_off;
                        }
                    }
                }
                if (chn >= (uint32_t)kRegChunks) {
                    const uint32_t i = i0 + d;
                    if (i < nx) {
                        has = true;
                        cur = tg[i];
                        if (i > 0) pp = tg[i - 1];
                        if (i > 1) ppp = tg[i - 2];
                    }
                }
                const uint32_t b = cur & 7u;
#pragma unroll
                for (uint32_t bb = 0; bb < 6; bb++) {
                    const bool mine = has && b == bb;
                    if (!__ballot(mine)) continue;
                    uint32_t n0 = n_cell[bb];
                    int found = -1;
                    if (mine) {
                        for (uint32_t j = 0; j < n0; j++)
                            if (l_pp[bb][j] == pp && l_ppp[bb][j] == ppp) {
                                found = (int)j;
                                break;
                            }
                        if (found >= 0) atomicAdd(&l_cnt[bb][found], 1u);
                    }
                    unsigned long long rem = __ballot(mine && found < 0);
                    while (rem) {
                        const int ld = __ffsll((long long)rem) - 1;  // earliest read that carries a new link
                        const uint32_t kp = (uint32_t)__shfl((int)pp, ld, 64);
                        const uint32_t kpp = (uint32_t)__shfl((int)ppp, ld, 64);
                        const bool in_rem = (rem >> lane) & 1ull;
                        const unsigned long long same = __ballot(in_rem && pp == kp && ppp == kpp);
                        if (lane == ld) {
                            if (n0 < (uint32_t)CAP) {
                                l_pp[bb][n0] = pp;
                                l_ppp[bb][n0] = ppp;
                                l_cnt[bb][n0] = (uint32_t)__popcll(same);
                            } else {
                                atomicExch(err, 1u);
                            }
                        }
                        n0 = n0 < (uint32_t)CAP ? n0 + 1 : n0;
                        rem &= ~same;
                    }
                    n_cell[bb] = n0;
                    __builtin_amdgcn_wave_barrier();  // one wavefront: LDS operations complete in program order
                }
            }
            // flush the six cells of (t, d), links contiguous per cell in first-seen order
            const uint64_t cell0 = P.cell_off + cb[t] + (uint64_t)d * 6u;
#pragma unroll
            for (uint32_t bb = 0; bb < 6; bb++) {
                const uint32_t n = n_cell[bb];
                if (lane == 0) {
                    cell_start[cell0 + bb] = (uint32_t)(e - P.ent_off);
                    cell_len[cell0 + bb] = n;
                }
                for (uint32_t j = (uint32_t)lane; j < n; j += 64) {
                    ent_pp[e + j] = l_pp[bb][j];
                    ent_ppp[e + j] = l_ppp[bb][j];
                    ent_cnt[e + j] = l_cnt[bb][j];
                }
                e += n;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- K10 -------------------------------------------------------------------------
__device__ __forceinline__ long long ld_score(const long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_score(long long *p, long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ long long readlane_i64(long long v, int src) {
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), src);
    return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ long long shfl_i64(long long v, int src) {
    const int lo = __shfl((int)(v & 0xffffffffll), src, 64);
    const int hi = __shfl((int)(v >> 32), src, 64);
    return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
}

// Scoring DP, segment-parallel.  The DP is a dependent chain over (column, delta) steps, 10^4-10^6 long per seed, and
// one step costs microseconds whatever the hardware does (a handful of dependent LDS round trips), so a seed is cut into
// segments of `seg_len` columns that are scored AT THE SAME TIME by different workgroups and stitched afterwards:
//   * segment 0 starts at column 0 and is exact as it stands;
//   * segment i > 0 does not know the scores its first column reads (those of the links of column c0 - 1).  It starts
//     `warm` columns early with every predecessor score set to one constant (kBig) and scores forward.  In this DP the best
//     paths of all links of a column join within a few dozen columns (reads share three consecutive tags with the majority
//     almost everywhere), so by column c0 - 1 the scores it holds differ from the true ones by ONE constant -- and every
//     decision of the DP compares scores with scores, so the decisions from c0 on are the true ones;
//   * nothing of this is assumed: the stitch kernel CHECKS it.  Segment i stores the scores it held for column c0 - 1
//     (`spec`), segment i - 1 the ones it computed for the same column (`fin`); the segment is accepted iff the two differ
//     by one constant over all offset-carrying links (and are equal for the others), which also yields its offset
//     off_i = off_{i-1} + that constant.  The DP also has absolute values -- a read's first link scores 10 * count -
//     factor * coverage from nothing, a link score is floored at 0, a cell's best starts at -10 -- and a comparison of an
//     offset-carrying score with an absolute one is decided by the guess kBig in the segment and by off_i in truth; both
//     give "the offset-carrying one is larger" iff the smallest true offset-carrying score of the segment exceeds its
//     largest absolute one, which is tracked (vmin, amax) and checked.  A segment that fails any check is scored again by
//     the stitch kernel from the true scores of its predecessor (`repair`), and the check of its successor is redone;
//   * the global pick (lib/nextcorrect.c:2194-2199) needs true values of every cell's best score: the cell bests are
//     stored, and the stitch kernel finds the last segment whose maximum reaches the running maximum - 3000 and scans it.
// Scores are int32 (raw); true score = raw + off (int64) for raw > kRelThr in an offset-carrying segment.  A pile with a
// column that does not fit the LDS tables, a raw score beyond the guard or a failed absolute-value check goes to the
// int64 HBM-resident kernel below.
constexpr int kColCells = 192, kColEnts = 512;            // max_size <= 32, <= 512 link slots per column
constexpr int32_t kNoScore = INT32_MIN;
constexpr int32_t kBig = 1 << 29;      // the constant a speculative segment starts from
constexpr int32_t kRelThr = 1 << 28;   // raw score above: carries the segment's offset; at or below: absolute
constexpr int32_t kAbsLim = 1 << 27;   // absolute values must stay below (the band up to kRelThr is nobody's)

template <int CELLS, int ENTS>
struct ColTab {
    uint32_t cstart[CELLS];
    uint32_t clen[CELLS];
    uint2 ps[ENTS];  // per link: x = pp tag, y = score (int32 bits) -- one 8-byte LDS access for both
};

// per-link operands of the column being scored, one 16-byte LDS access: before a link is scored
// {ppp, resolve word, gain, count}; after it {score at the best predecessor (scmax), improving predecessor
// score (impr), best predecessor score (nsmax), count}
struct LinkAux {
    uint32_t a, b_;
    int32_t c;
    uint32_t cnt;
};

template <int CELLS, int ENTS>
struct K10Smem {
    ColTab<CELLS, ENTS> tab[3];                                   // columns p-1, p, p+1 (slot = column mod 3)
    __attribute__((aligned(16))) LinkAux aux[3][ENTS];            // same slots
    uint32_t bpp[2][CELLS], blink[2][CELLS];                      // slot = column & 1
    int32_t best[2][CELLS];
    uint32_t meta[2][5][64];                                      // width, cell0, e0, ecap, coverage of 2 x 64 columns
    uint32_t colw[3], colc0[3], stop[3];                          // per prepared column: width, first cell, "does not fit"
    uint32_t r_links, r_flags, r_nfin;                            // results of a range
    int32_t r_vmin, r_amax, r_brel, r_babs;
};

__device__ __forceinline__ int32_t wave_min_i32(int32_t v) {
    for (int o = 32; o; o >>= 1) {
        const int32_t u = __shfl_xor(v, o, 64);
        v = u < v ? u : v;
    }
    return v;
}
__device__ __forceinline__ int32_t wave_max_i32(int32_t v) {
    for (int o = 32; o; o >>= 1) {
        const int32_t u = __shfl_xor(v, o, 64);
        v = u > v ? u : v;
    }
    return v;
}

// Scores the columns [mode 0: c0, else w0) .. c1 of one pile and keeps the results of [c0, c1).  Three wavefronts, one
// barrier per column:
//   loader (wave 1), one column ahead: the column's cell / link tables HBM -> registers (prefetch) -> LDS, every link
//     resolved (predecessor cell range, gain, mask of the predecessor links it continues), finished results stored;
//   scorer (wave 0), the dependent chain: one link per lane takes the best of its matching predecessors' scores;
//   folder (wave 2), one column behind: the five symbol cells of every step fold their links with the reference's
//     sequential tie-break rules (lib/nextcorrect.c:2164-2192).
// mode 0: the range starts at the seed's first column (w0 = c0 = 0), scores are true values;
// mode 1: speculative start: column w0 - 1 is loaded with every score = kBig, columns [w0, c0) are the warm-up;
// mode 2: exact restart (w0 = c0): column c0 - 1 is loaded with `init_scores` (the predecessor segment's `fin`).
// enc: scores above kRelThr carry an offset (always in mode 1; in mode 2 what the predecessor's scores are).
// The summary goes to A.sums[sidx], the scores of column c1 - 1 to A.fin, in mode 1 those of column c0 - 1 to A.spec.
template <int CELLS, int ENTS>
__device__ void score_range(K10Smem<CELLS, ENTS> &S, const K10Args &A, const PileDev &P, const uint32_t sidx, const uint32_t c0,
                            const uint32_t c1, const uint32_t w0, const int mode, const bool enc,
                            const int32_t *__restrict__ init_scores) {
    using Tab = ColTab<CELLS, ENTS>;
    const int wave = (int)(threadIdx.x >> 6);
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t L = c1;
    const uint32_t first = mode == 0 ? c0 : w0 - 1u;  // first column whose tables are loaded
    const uint32_t *cov = A.coverage + P.col_off;
    const uint32_t *ms = A.max_size + P.col_off;
    const uint32_t *cb = A.cell_base + P.col_off;
    const uint32_t *eb = A.ent_base + P.col_off;
    const uint32_t *cs = A.cell_start + P.cell_off;
    const uint32_t *cl = A.cell_len + P.cell_off;
    const uint32_t *epp = A.ent_pp + P.ent_off;
    const uint32_t *eppp = A.ent_ppp + P.ent_off;
    const uint32_t *ecnt = A.ent_cnt + P.ent_off;
    uint32_t *bpp_out = A.cell_best_pp + P.cell_off;
    uint32_t *blk_out = A.cell_best_link + P.cell_off;
    int32_t *best_out = A.cell_best + P.cell_off;
    const int32_t factor = P.factor;
    const int32_t guard = A.guard;

    // ---- loader state
    uint32_t pf_cs = 0, pf_cl = 0, pf_pp[4] = {0, 0, 0, 0}, pf_ppp[4] = {0, 0, 0, 0}, pf_cnt[4] = {0, 0, 0, 0};
    uint32_t n_links = 0;
    auto load_meta = [&](uint32_t p0) {  // 64 columns of metadata, one coalesced load per array
        const uint32_t p = p0 + (uint32_t)lane;
        uint32_t w_ = 0, c_ = 0, e_ = 0, k_ = 0, v_ = 0;
        if (p < L) {
            w_ = ms[p];
            c_ = cb[p];
            e_ = eb[p];
            k_ = eb[p + 1] - e_;  // link capacity of the column (>= links actually present)
            v_ = cov[p];
        }
        const uint32_t par = (p0 >> 6) & 1u;
        S.meta[par][0][lane] = w_, S.meta[par][1][lane] = c_, S.meta[par][2][lane] = e_, S.meta[par][3][lane] = k_;
        S.meta[par][4][lane] = v_;
    };
    auto prefetch = [&](uint32_t cell0, uint32_t ncell, uint32_t e0, uint32_t ecap) {
        if ((uint32_t)lane < ncell) {
            pf_cs = cs[cell0 + lane];
            pf_cl = cl[cell0 + lane];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t e = (uint32_t)lane + 64u * (uint32_t)j;
            if (e < ecap) {
                pf_pp[j] = epp[e0 + e];
                pf_ppp[j] = eppp[e0 + e];
                pf_cnt[j] = ecnt[e0 + e];
            }
        }
    };
    // prepare column q: tables -> LDS, links resolved (not for the start column of modes 1 / 2: its links are never
    // scored and the column before it is not loaded).  Runs on the loader wave only.
    auto prepare = [&](uint32_t q, bool resolve) {
        const int ml = (int)(q & 63u);
        const uint32_t mp = (q >> 6) & 1u, slot = q % 3u;
        const uint32_t width = S.meta[mp][0][ml], cell0 = S.meta[mp][1][ml], e0 = S.meta[mp][2][ml], ecap = S.meta[mp][3][ml];
        const int32_t pen = factor * (int32_t)S.meta[mp][4][ml];
        const uint32_t ncell = width * 6u;
        const bool fits = ncell <= (uint32_t)CELLS && ecap <= (uint32_t)ENTS;
        if (lane == 0) S.colw[slot] = width, S.colc0[slot] = cell0, S.stop[slot] = fits ? 0u : 1u;
        Tab &cur = S.tab[slot];
        const Tab &prv = S.tab[(q + 2u) % 3u];
        LinkAux *aux = S.aux[slot];
        if (fits) {
            if ((uint32_t)lane < ncell) {
                cur.cstart[lane] = pf_cs - e0;
                cur.clen[lane] = pf_cl;
            }
            for (uint32_t c = (uint32_t)lane + 64u; c < ncell; c += 64) {
                cur.cstart[c] = cs[cell0 + c] - e0;
                cur.clen[c] = cl[cell0 + c];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t e = (uint32_t)lane + 64u * (uint32_t)j;
                if (e < ecap) {
                    cur.ps[e].x = pf_pp[j];
                    aux[e].a = pf_ppp[j];
                    aux[e].cnt = pf_cnt[j];
                }
            }
            for (uint32_t e = (uint32_t)lane + 256u; e < ecap; e += 64) {
                cur.ps[e].x = epp[e0 + e];
                aux[e].a = eppp[e0 + e];
                aux[e].cnt = ecnt[e0 + e];
            }
        }
        // the loads of column q+1 start now (and the next metadata block when q closes one)
        if (q + 1 < L) {
            if (ml == 63) load_meta(q + 1);
            __builtin_amdgcn_wave_barrier();
            const int nl = (int)((q + 1) & 63u);
            const uint32_t np_ = ((q + 1) >> 6) & 1u;
            prefetch(S.meta[np_][1][nl], S.meta[np_][0][nl] * 6u, S.meta[np_][2][nl], S.meta[np_][3][nl]);
        }
        if (!fits || ncell == 0 || !resolve) return;
        __builtin_amdgcn_wave_barrier();
        // resolve every link of the column once
        const uint32_t nent = cur.cstart[ncell - 1u] + cur.clen[ncell - 1u];
        if (q >= c0) n_links += nent;
        for (uint32_t e = (uint32_t)lane; e < nent; e += 64) {
            const uint32_t mpp = cur.ps[e].x;
            uint32_t res = 0;
            if (mpp != kTagHead) {
                const bool same = (uint32_t)tag_tpos(mpp) == q;
                const Tab &T = same ? cur : prv;
                const uint32_t pc = tag_delta(mpp) * 6u + tag_base(mpp);
                res = ((uint32_t)same << 31) | (T.cstart[pc] << 12) | T.clen[pc];
            }
            // which links of the predecessor cell continue this link (their pp equals this link's ppp): tags only, no
            // scores, so the loader can do it; the scorer then touches matching predecessors only.  Cells with more
            // than 32 links keep ppp and are scanned by the scorer.
            const uint32_t pn = res & 0xfffu;
            if (mpp != kTagHead && pn <= 32u) {
                const Tab &T = (res >> 31) ? cur : prv;
                const uint32_t ps = (res >> 12) & 0x7ffffu, mppp = aux[e].a;
                uint32_t match = 0;
                for (uint32_t k = 0; k < pn; k++)
                    if (T.ps[ps + k].x == mppp) match |= 1u << k;
                aux[e].a = match;
            }
            aux[e].b_ = res;
            aux[e].c = 10 * (int32_t)aux[e].cnt - pen;
        }
    };
    auto store_results = [&](uint32_t q) {  // best_pp / best_link / best score of column q (folded one iteration ago)
        const uint32_t pw = S.colw[q % 3u] * 6u, pc0 = S.colc0[q % 3u], sl = q & 1u;
        for (uint32_t c = (uint32_t)lane; c < pw; c += 64)
            if (c % 6u < 5u) {
                bpp_out[pc0 + c] = S.bpp[sl][c];
                blk_out[pc0 + c] = S.blink[sl][c];
                best_out[pc0 + c] = S.best[sl][c];
            }
    };

    // ---- scorer state: overflow / band flag, smallest offset-carrying and largest absolute link score seen from column
    // c0 - 1 on (what the absolute-value check of the stitch needs)
    bool sc_overflow = false;
    int32_t t_vmin = INT32_MAX, t_amax = 0;
    // ---- folder state: largest cell best of the owned columns, by kind
    int32_t f_brel = INT32_MIN, f_babs = INT32_MIN;

    __syncthreads();  // the previous user of S is done
    if (threadIdx.x == 0) S.r_links = 0, S.r_flags = 0, S.r_nfin = 0, S.r_vmin = INT32_MAX, S.r_amax = 0, S.r_brel = INT32_MIN, S.r_babs = INT32_MIN;
    if (wave == 1) {
        load_meta(first & ~63u);
        __builtin_amdgcn_wave_barrier();
        const int fl = (int)(first & 63u);
        const uint32_t fp = (first >> 6) & 1u;
        prefetch(S.meta[fp][1][fl], S.meta[fp][0][fl] * 6u, S.meta[fp][2][fl], S.meta[fp][3][fl]);
        prepare(first, mode == 0);
    }
    __syncthreads();

    bool stopped = false;
    // iteration p: loader prepares column p+1 and stores column p-2, scorer scores column p, folder folds column p-1
    for (uint32_t p = first; p <= L; p++) {
        if (p < L && S.stop[p % 3u]) {  // column p does not fit the LDS tables: the pile goes to the HBM-resident kernel
            stopped = true;
            break;
        }
        if (wave == 1) {
            // (column p+1 first: its commit drains the memory counter, nothing younger than the prefetched loads may be
            // in flight; the stores come after)
            const bool st_ok = p >= c0 + 2u;
            uint32_t sw = 0, sc0 = 0;
            if (st_ok) sw = S.colw[(p - 2u) % 3u] * 6u, sc0 = S.colc0[(p - 2u) % 3u];  // slot p+1 = slot p-2: read before prepare
            ND_LOCKSTEP();
            if (p + 1 < L) prepare(p + 1, true);
            if (st_ok) {
                const uint32_t sl = (p - 2u) & 1u;
                for (uint32_t c = (uint32_t)lane; c < sw; c += 64)
                    if (c % 6u < 5u) {
                        bpp_out[sc0 + c] = S.bpp[sl][c];
                        blk_out[sc0 + c] = S.blink[sl][c];
                        best_out[sc0 + c] = S.best[sl][c];
                    }
            }
        } else if (wave == 0) {
            const uint32_t slot = p % 3u;
            const uint32_t width = p < L ? S.colw[slot] : 0u;
            Tab &cur = S.tab[slot];
            const uint32_t nent = width ? cur.cstart[width * 6u - 1u] + cur.clen[width * 6u - 1u] : 0u;
            const bool track = enc && p + 1u >= c0;
            if (mode != 0 && p == first) {
                // start column: scores are given, not computed
                for (uint32_t e = (uint32_t)lane; e < nent; e += 64) {
                    const int32_t v = mode == 1 ? kBig : init_scores[e];
                    cur.ps[e].y = (uint32_t)v;
                    if (track) {
                        if (v > kRelThr) t_vmin = v < t_vmin ? v : t_vmin;
                        else t_amax = v > t_amax ? v : t_amax;
                        if (v > kAbsLim && v <= kRelThr) sc_overflow = true;
                    }
                    if (v > guard) sc_overflow = true;
                }
            } else if (width) {
                const Tab &prv = S.tab[(p + 2u) % 3u];
                LinkAux *aux = S.aux[slot];
                uint32_t step_est = 0, step_n = 0;
                if ((uint32_t)lane < width) {
                    step_est = cur.cstart[(uint32_t)lane * 6u];
                    step_n = cur.cstart[(uint32_t)lane * 6u + 4u] + cur.clen[(uint32_t)lane * 6u + 4u] - step_est;
                }
                for (uint32_t d = 0; d < width; d++) {
                    const uint32_t est = (uint32_t)__builtin_amdgcn_readlane((int)step_est, (int)d);
                    const uint32_t n_step = (uint32_t)__builtin_amdgcn_readlane((int)step_n, (int)d);
                    // one link per lane (64 at a time): final score + the three numbers the cell's sequential state
                    // needs from it, left in the link's operand record for the folder
                    for (uint32_t g0 = 0; g0 < n_step; g0 += 64) {
                        const uint32_t g_n = n_step - g0 < 64u ? n_step - g0 : 64u;
                        if ((uint32_t)lane < g_n) {
                            int32_t r_sc = 0, r_impr = kNoScore, r_nsmax = kNoScore, r_scmax = 0;
                            const uint32_t idx = est + g0 + (uint32_t)lane;
                            const LinkAux ax = aux[idx];
                            const uint32_t mpp = cur.ps[idx].x, mppp = ax.a, res = ax.b_;
                            const int32_t gain = ax.c;
                            if (mpp == kTagHead) {
                                r_sc = gain;
                            } else {
                                const Tab &T = (res >> 31) ? cur : prv;
                                const uint32_t ps = (res >> 12) & 0x7ffffu, pn = res & 0xfffu;
                                if (pn <= 32u) {
                                    // mppp holds the loader's match mask: visit the matching predecessors in order,
                                    // two scores per LDS round trip
                                    uint32_t m = mppp;
                                    while (m) {
                                        const uint32_t k1 = (uint32_t)__builtin_ctz(m);
                                        m &= m - 1u;
                                        const uint32_t k2 = m ? (uint32_t)__builtin_ctz(m) : k1;
                                        const bool two = m != 0u;
                                        m &= m - 1u;
                                        const int32_t ns1 = (int32_t)T.ps[ps + k1].y, ns2 = (int32_t)T.ps[ps + k2].y;
                                        if (ns1 + gain > r_sc) r_sc = ns1 + gain, r_impr = ns1;
                                        if (ns1 > r_nsmax) r_nsmax = ns1, r_scmax = r_sc;
                                        if (two) {
                                            if (ns2 + gain > r_sc) r_sc = ns2 + gain, r_impr = ns2;
                                            if (ns2 > r_nsmax) r_nsmax = ns2, r_scmax = r_sc;
                                        }
                                    }
                                } else
                                for (uint32_t k0 = 0; k0 < pn; k0 += 4) {  // 4 predecessor links per LDS round trip
                                    uint32_t key[4];
                                    int32_t nsv[4];
#pragma unroll
                                    for (int u = 0; u < 4; u++) {
                                        const uint32_t k = k0 + (uint32_t)u < pn ? k0 + (uint32_t)u : pn - 1u;
                                        const uint2 t2 = T.ps[ps + k];
                                        key[u] = t2.x;
                                        nsv[u] = (int32_t)t2.y;
                                    }
#pragma unroll
                                    for (int u = 0; u < 4; u++) {
                                        if (k0 + (uint32_t)u < pn && key[u] == mppp) {
                                            const int32_t ns = nsv[u];
                                            if (ns + gain > r_sc) {
                                                r_sc = ns + gain;
                                                r_impr = ns;
                                            }
                                            if (ns > r_nsmax) {
                                                r_nsmax = ns;
                                                r_scmax = r_sc;
                                            }
                                        }
                                    }
                                }
                            }
                            cur.ps[idx].y = (uint32_t)r_sc;
                            LinkAux wr;
                            wr.a = (uint32_t)r_scmax, wr.b_ = (uint32_t)r_impr, wr.c = r_nsmax, wr.cnt = ax.cnt;
                            aux[idx] = wr;
                            if (r_sc > guard) sc_overflow = true;
                            if (track) {
                                if (r_sc > kRelThr) t_vmin = r_sc < t_vmin ? r_sc : t_vmin;
                                else t_amax = r_sc > t_amax ? r_sc : t_amax;
                                if (r_sc > kAbsLim && r_sc <= kRelThr) sc_overflow = true;
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();  // scores of (p,d) in LDS before (p,d+1) reads them (one wave, in order)
                }
            }
            // the scores of the two columns the stitch compares
            if (p + 1u == L) {
                __builtin_amdgcn_wave_barrier();
                int32_t *dst = A.fin + (size_t)sidx * (size_t)kColEnts;
                for (uint32_t e = (uint32_t)lane; e < nent; e += 64) dst[e] = (int32_t)cur.ps[e].y;
                if (lane == 0) S.r_nfin = nent;
            } else if (mode == 1 && p + 1u == c0) {
                __builtin_amdgcn_wave_barrier();
                int32_t *dst = A.spec + (size_t)sidx * (size_t)kColEnts;
                for (uint32_t e = (uint32_t)lane; e < nent; e += 64) dst[e] = (int32_t)cur.ps[e].y;
            }
        } else if (p > first && p - 1u >= c0) {
            const uint32_t q = p - 1u, slot = q % 3u, sl = q & 1u;
            const uint32_t width = S.colw[slot];
            if (width) {
                const Tab &cur = S.tab[slot];
                const LinkAux *aux = S.aux[slot];
                // every (delta, symbol) cell of the column folds its links on its own lane (12 deltas x 5 symbols per pass)
                for (uint32_t d0 = 0; d0 < width; d0 += 12u) {
                    const uint32_t dl = (uint32_t)lane / 5u, d = d0 + dl, cb_ = (uint32_t)lane - dl * 5u;
                    const uint32_t n_d = width - d0 < 12u ? width - d0 : 12u;
                    if (dl < n_d) {
                        int32_t best = -10;                  // state of cell (d, cb_)
                        uint32_t bpp = kTagHead, blink = 0;  // best_pp / best_link of the cell
                        const uint32_t cst = cur.cstart[d * 6u + cb_], cn = cur.clen[d * 6u + cb_];
                        int32_t via = kNoScore, via_next = kNoScore;
                        for (uint32_t m0 = 0; m0 < cn; m0 += 4) {  // 4 links per LDS round trip
                            int32_t a_sc[4], a_impr[4], a_ns[4], a_scm[4];
                            uint32_t a_pp[4], a_cnt[4];
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                const uint32_t idx = cst + (m0 + (uint32_t)u < cn ? m0 + (uint32_t)u : cn - 1u);
                                const uint2 t2 = cur.ps[idx];
                                const LinkAux ax = aux[idx];
                                a_pp[u] = t2.x, a_sc[u] = (int32_t)t2.y;
                                a_scm[u] = (int32_t)ax.a, a_impr[u] = (int32_t)ax.b_, a_ns[u] = ax.c, a_cnt[u] = ax.cnt;
                            }
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                if (m0 + (uint32_t)u < cn) {
                                    const uint32_t pb = tag_base(a_pp[u]);
                                    if (a_impr[u] != kNoScore) via_next = a_impr[u];
                                    if (a_ns[u] > via && (pb == 4u || pb == cb_)) {
                                        via = a_ns[u];
                                        best = a_scm[u];
                                        bpp = a_pp[u], blink = a_cnt[u];
                                    }
                                    if (a_sc[u] > best || (a_sc[u] == best && pb != 4u)) {
                                        via = via_next;
                                        best = a_sc[u];
                                        bpp = a_pp[u], blink = a_cnt[u];
                                    }
                                }
                            }
                        }
                        S.bpp[sl][d * 6u + cb_] = bpp;
                        S.blink[sl][d * 6u + cb_] = blink;
                        S.best[sl][d * 6u + cb_] = best;
                        if (enc && best > kRelThr) f_brel = best > f_brel ? best : f_brel;
                        else f_babs = best > f_babs ? best : f_babs;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (wave == 1) {
        if (!stopped && L > c0) store_results(L - 1u);  // column L-2 was stored in the last iteration (p = L)
        if (lane == 0) S.r_links = n_links;
    } else if (wave == 0) {
        const int32_t vmin = wave_min_i32(t_vmin), amax = wave_max_i32(t_amax);
        const bool ovf = __ballot(sc_overflow) != 0ull;
        if (lane == 0) {
            S.r_vmin = vmin, S.r_amax = amax;
            if (ovf) atomicOr(&S.r_flags, 2u);
        }
    } else {
        const int32_t brel = wave_max_i32(f_brel), babs = wave_max_i32(f_babs);
        if (lane == 0) {
            S.r_brel = brel, S.r_babs = babs;
            if (stopped) atomicOr(&S.r_flags, 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        SegSum &R = A.sums[sidx];
        R.vmin = S.r_vmin, R.amax = S.r_amax, R.bmax_rel = S.r_brel, R.bmax_abs = S.r_babs;
        R.links = S.r_links, R.n_fin = S.r_nfin, R.flags = S.r_flags;
    }
}

// Phase A: one workgroup per (pile, segment) of this table tier.  (Five wavefronts per SIMD asked for: the small tier then takes 96 registers
// instead of 101 -- no scratch -- and six workgroups fit a compute unit where five did; the kernel is parked half its cycles.)
template <int CELLS, int ENTS>
__global__ __launch_bounds__(192, 5) void score_seg_kernel(const K10Args A, const SegItem *__restrict__ items) {
    __shared__ K10Smem<CELLS, ENTS> S;
    const SegItem it = items[blockIdx.x];
    const PileDev &P = A.piles[it.pile];
    const uint32_t c0 = it.seg * A.seg_len;
    const uint32_t c1 = it.seg + 1u == P.n_seg ? P.seed_len : c0 + A.seg_len;
    if (it.seg == 0) score_range<CELLS, ENTS>(S, A, P, P.seg_off, 0u, c1, 0u, 0, false, nullptr);
    else score_range<CELLS, ENTS>(S, A, P, P.seg_off + it.seg, c0, c1, c0 - A.warm, 1, true, nullptr);
}

// The boundary check of segment i (see the head comment): `fin` of its predecessor against its own `spec`, n links.
// All threads of the wave call it; returns ok, *dlt = the constant (0 when no link carries an offset).
__device__ __forceinline__ bool check_boundary(const int32_t *__restrict__ fin_prev, const int32_t *__restrict__ spec,
                                               uint32_t n, bool pred_enc, int lane, int32_t *dlt) {
    bool bad = false, have = false;
    int32_t d0 = 0;
    for (uint32_t e0 = 0; e0 < n; e0 += 64) {
        const uint32_t e = e0 + (uint32_t)lane;
        bool rel = false;
        int32_t d = 0;
        if (e < n) {
            const int32_t u = fin_prev[e], v = spec[e];
            if (v > kRelThr) {
                rel = true;
                d = (int32_t)((uint32_t)u - (uint32_t)v);
                if (pred_enc && u <= kRelThr) bad = true;
            } else if (u != v) {
                bad = true;
            }
        }
        const unsigned long long rm = __ballot(rel);
        if (rm) {
            const int32_t dl = __shfl(d, __ffsll((long long)rm) - 1, 64);
            if (!have) d0 = dl, have = true;
            if (rel && d != d0) bad = true;
        }
    }
    *dlt = have ? d0 : 0;
    return __ballot(bad) == 0ull;
}

// Phase B + C: one workgroup per pile checks its segments, repairs the ones that fail, and makes the global pick.
template <int CELLS, int ENTS, bool REDO>
__global__ __launch_bounds__(192) void score_stitch_kernel(const K10Args A) {
    __shared__ K10Smem<CELLS, ENTS> S;
    __shared__ long long sh_off, sh_g, sh_pm[192];
    __shared__ uint32_t sh_enc, sh_i, sh_fail, sh_flags, sh_links, sh_repairs, sh_seg;
    __shared__ int sh_last;
    PileDev &P = A.piles[blockIdx.x];
    if (REDO != (P.tier != 0u) || P.err == 2) return;  // the column scan chose the tier of every pile; 2: HBM-resident kernel
    const int wave = (int)(threadIdx.x >> 6);
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t n_seg = P.n_seg, sg = A.seg_len;
    SegSum *sums = A.sums + P.seg_off;
    const int32_t *fin = A.fin + (size_t)P.seg_off * kColEnts;
    const int32_t *spec = A.spec + (size_t)P.seg_off * kColEnts;
    auto seg_end = [&](uint32_t i) { return i + 1u == n_seg ? P.seed_len : (i + 1u) * sg; };

    if (threadIdx.x == 0) sh_off = 0, sh_enc = 0, sh_i = 1, sh_fail = 0, sh_flags = 0, sh_links = 0, sh_repairs = 0;
    __syncthreads();
    // pass 0 / 1 (parallel over segments): flags, link totals, boundary checks
    {
        uint32_t fl = 0, lk = 0;
        for (uint32_t i = threadIdx.x; i < n_seg; i += 192) {
            fl |= sums[i].flags;
            lk += sums[i].links;
        }
        if (fl) atomicOr(&sh_flags, fl);
        atomicAdd(&sh_links, lk);
        for (uint32_t i = 1u + (uint32_t)wave; i < n_seg; i += 3u) {
            int32_t dlt;
            const bool ok = check_boundary(fin + (size_t)(i - 1u) * kColEnts, spec + (size_t)i * kColEnts, sums[i - 1u].n_fin,
                                           i > 1u, lane, &dlt);
            if (lane == 0) sums[i].ok = ok ? 1u : 0u, sums[i].dlt = dlt;
        }
        if (threadIdx.x == 0) sums[0].off = 0, sums[0].enc = 0;
    }
    __threadfence_block();
    __syncthreads();
    if (sh_flags) {
        if (threadIdx.x == 0) P.err = 2;
        return;
    }
    // pass 2: offsets in segment order, 64 segments per round; the first segment that fails a check is repaired
    for (;;) {
        const uint32_t i0 = sh_i;
        __syncthreads();  // every wavefront has read sh_i before wavefront 0 moves it on
        if (i0 >= n_seg) break;
        if (wave == 0) {
            const uint32_t seg = i0 + (uint32_t)lane;
            const bool have = seg < n_seg;
            uint32_t ok = 0;
            int32_t dlt = 0, vmin = INT32_MAX, amax = 0;
            if (have) ok = sums[seg].ok, dlt = sums[seg].dlt, vmin = sums[seg].vmin, amax = sums[seg].amax;
            long long pre = have ? (long long)dlt : 0ll;
            for (int o = 1; o < 64; o <<= 1) {
                const long long u = shfl_i64(pre, lane >= o ? lane - o : lane);
                if (lane >= o) pre += u;
            }
            const long long offs = sh_off + pre;
            bool valid = have && ok && (vmin == INT32_MAX || (long long)vmin + offs > (long long)amax);
            if (A.force_repair && have && (seg % A.force_repair) == 1u) valid = false;
            const unsigned long long badm = __ballot(have && !valid), havem = __ballot(have);
            const int nvalid = badm ? __ffsll((long long)badm) - 1 : __popcll(havem);
            if (lane < nvalid) sums[seg].off = offs, sums[seg].enc = 1u;
            const long long last_off = shfl_i64(offs, nvalid > 0 ? nvalid - 1 : 0);
            if (lane == 0) {
                if (nvalid > 0) sh_off = last_off, sh_enc = 1u;
                sh_i = i0 + (uint32_t)nvalid;
                sh_fail = badm ? 1u : 0u;
            }
        }
        __syncthreads();
        if (sh_fail) {
            const uint32_t f = sh_i;
            const bool enc = sh_enc != 0u;
            score_range<CELLS, ENTS>(S, A, P, P.seg_off + f, f * sg, seg_end(f), f * sg, 2, enc, fin + (size_t)(f - 1u) * kColEnts);
            __threadfence_block();
            __syncthreads();
            const SegSum R = sums[f];
            const bool dead = R.flags != 0u || (enc && R.vmin != INT32_MAX && !((long long)R.vmin + sh_off > (long long)R.amax));
            if (dead) {  // a column that does not fit / a raw score out of range / absolute values in reach: int64 kernel
                if (threadIdx.x == 0) P.err = 2;
                return;
            }
            if (wave == 0 && f + 1u < n_seg) {
                int32_t dlt;
                const bool ok = check_boundary(fin + (size_t)f * kColEnts, spec + (size_t)(f + 1u) * kColEnts, R.n_fin, enc, lane, &dlt);
                if (lane == 0) sums[f + 1u].ok = ok ? 1u : 0u, sums[f + 1u].dlt = dlt;
            }
            if (threadIdx.x == 0) {
                sums[f].off = sh_off, sums[f].enc = sh_enc;
                sh_i = f + 1u, sh_fail = 0, sh_repairs++;
            }
            __threadfence_block();
        }
        __syncthreads();
    }
    // Phase C: the global pick.  True maximum of every segment, running maximum before it (g), the last segment whose
    // maximum reaches g - 3000 holds the answer (its maximum cell passes the test, and no later cell does).
    if (wave == 0) {
        long long carry = -10;
        int last = -1;
        long long g_last = -10;
        for (uint32_t i0 = 0; i0 < n_seg; i0 += 64) {
            const uint32_t seg = i0 + (uint32_t)lane;
            long long tm = LLONG_MIN;
            if (seg < n_seg) {
                const SegSum R = sums[seg];
                if (R.bmax_abs != INT32_MIN) tm = R.bmax_abs;
                if (R.bmax_rel != INT32_MIN) {
                    const long long t2 = (long long)R.bmax_rel + R.off;
                    tm = t2 > tm ? t2 : tm;
                }
            }
            long long pm = tm;  // inclusive prefix maximum
            for (int o = 1; o < 64; o <<= 1) {
                const long long u = shfl_i64(pm, lane >= o ? lane - o : lane);
                if (lane >= o && u > pm) pm = u;
            }
            long long g = shfl_i64(pm, lane ? lane - 1 : 0);
            if (lane == 0 || g < carry) g = lane == 0 ? carry : (g < carry ? carry : g);
            const bool q = tm != LLONG_MIN && tm >= g - 3000;
            const unsigned long long qm = __ballot(q);
            if (qm) {
                const int hl = 63 - __clzll((long long)qm);
                last = (int)i0 + hl;
                g_last = shfl_i64(g, hl);
            }
            const long long top = shfl_i64(pm, 63);
            carry = top > carry ? top : carry;
        }
        if (lane == 0) sh_last = last, sh_g = g_last;
    }
    __syncthreads();
    const int last = sh_last;
    int32_t o_t = -1;
    uint32_t o_db = 0;
    if (last >= 0) {
        const uint32_t *cb = A.cell_base + P.col_off;
        const int32_t *best = A.cell_best + P.cell_off;
        const SegSum R = sums[last];
        const uint32_t ca = cb[(uint32_t)last * sg], ce = cb[seg_end((uint32_t)last)];
        const uint32_t n = ce - ca, per = (n + 191u) / 192u;
        const uint32_t a = ca + threadIdx.x * per < ce ? ca + threadIdx.x * per : ce;
        const uint32_t b = a + per < ce ? a + per : ce;
        auto val = [&](uint32_t c) -> long long {
            const int32_t raw = best[c];
            return (R.enc && raw > kRelThr) ? (long long)raw + R.off : (long long)raw;
        };
        long long mx = LLONG_MIN;
        for (uint32_t c = a; c < b; c++)
            if (c % 6u < 5u) {
                const long long v = val(c);
                mx = v > mx ? v : mx;
            }
        sh_pm[threadIdx.x] = mx;
        if (threadIdx.x == 0) sh_seg = 0;
        __syncthreads();
        long long run = sh_g;
        for (uint32_t t = 0; t < threadIdx.x; t++) run = sh_pm[t] > run ? sh_pm[t] : run;
        uint32_t hit = 0;  // cell index + 1 of the last cell of this thread's chunk that passes
        for (uint32_t c = a; c < b; c++)
            if (c % 6u < 5u) {
                const long long v = val(c);
                if (v >= run - 3000) {
                    hit = c + 1u;
                    if (v > run) run = v;
                }
            }
        if (hit) atomicMax(&sh_seg, hit);
        __syncthreads();
        if (threadIdx.x == 0 && sh_seg) {
            const uint32_t cell = sh_seg - 1u;
            uint32_t lo = (uint32_t)last * sg, hi = seg_end((uint32_t)last);  // greatest column t in [lo, hi) with cb[t] <= cell
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if (cb[mid] <= cell) lo = mid;
                else hi = mid;
            }
            o_t = (int32_t)lo;
            const uint32_t r = cell - cb[lo];
            o_db = ((r / 6u) << 3) | (r % 6u);
        }
    }
    if (threadIdx.x == 0) {
        P.err = 0;
        P.origin_t = o_t;
        P.origin_db = o_db;
        P.n_links = sh_links;
        P.n_repair = sh_repairs;
    }
}

// Slow path (rare): same DP with every table in HBM, for piles whose columns exceed the
// LDS tables of the fast kernel.  Runs only on piles flagged err == 2.
__global__ __launch_bounds__(64) void score_slow_kernel(
    PileDev *__restrict__ piles, const uint32_t *__restrict__ coverage, const uint32_t *__restrict__ max_size,
    const uint32_t *__restrict__ cell_base, const uint32_t *__restrict__ cell_start,
    const uint32_t *__restrict__ cell_len, const uint32_t *__restrict__ ent_pp, const uint32_t *__restrict__ ent_ppp,
    const uint32_t *__restrict__ ent_cnt, long long *__restrict__ ent_score, uint32_t *__restrict__ cell_best_pp,
    uint32_t *__restrict__ cell_best_link) {
    PileDev &P = piles[blockIdx.x];
    if (P.err != 2) return;
    const int lane = (int)threadIdx.x;
    const uint32_t b = (uint32_t)lane;
    const bool act = lane < 5;  // symbols A T G C - (lib/nextcorrect.c:2151)
    const uint32_t L = P.seed_len;
    const uint32_t *cov = coverage + P.col_off;
    const uint32_t *ms = max_size + P.col_off;
    const uint32_t *cb = cell_base + P.col_off;
    const uint32_t *cs = cell_start + P.cell_off;
    const uint32_t *cl = cell_len + P.cell_off;
    const uint32_t *epp = ent_pp + P.ent_off;
    const uint32_t *eppp = ent_ppp + P.ent_off;
    const uint32_t *ecnt = ent_cnt + P.ent_off;
    long long *esc = ent_score + P.ent_off;
    uint32_t *bpp_out = cell_best_pp + P.cell_off;
    uint32_t *blk_out = cell_best_link + P.cell_off;
    const long long factor = P.factor;

    long long gbest = -10;
    int32_t o_t = -1;
    uint32_t o_db = 0;
    for (uint32_t p = 0; p < L; p++) {
        const uint32_t width = ms[p];
        const long long pen = factor * (long long)cov[p];
        for (uint32_t d = 0; d < width; d++) {
            long long best = -10;
            uint32_t bpp = kTagHead, blink = 0;
            if (act) {
                const uint32_t cell = cb[p] + d * 6u + b;
                const uint32_t st = cs[cell], n = cl[cell];
                long long via = LLONG_MIN, via_next = LLONG_MIN;
                for (uint32_t m = 0; m < n; m++) {
                    const uint32_t mpp = epp[st + m], mppp = eppp[st + m];
                    const long long gain = 10ll * (long long)ecnt[st + m] - pen;
                    long long sc = 0;
                    if (mpp == kTagHead) {
                        sc = gain;
                    } else {
                        const uint32_t pc = cb[tag_tpos(mpp)] + tag_delta(mpp) * 6u + tag_base(mpp);
                        const uint32_t ps = cs[pc], pn = cl[pc];
                        const uint32_t pb = tag_base(mpp);
                        for (uint32_t k = 0; k < pn; k++) {
                            if (epp[ps + k] != mppp) continue;
                            const long long ns = ld_score(&esc[ps + k]);
                            const long long s = ns + gain;
                            if (s > sc) {
                                sc = s;
                                via_next = ns;
                            }
                            if (ns > via && (pb == 4u || pb == b)) {
                                via = ns;
                                best = sc;
                                bpp = mpp;
                                blink = ecnt[st + m];
                            }
                        }
                    }
                    st_score(&esc[st + m], sc);
                    if (sc > best || (sc == best && tag_base(mpp) != 4u)) {
                        via = via_next;
                        best = sc;
                        bpp = mpp;
                        blink = ecnt[st + m];
                    }
                }
                bpp_out[cell] = bpp;
                blk_out[cell] = blink;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // scores of this (p,d) are in L2 before any later read
            for (int bb = 0; bb < 5; bb++) {
                const long long v = shfl_i64(best, bb);
                if (v >= gbest - 3000) {
                    o_t = (int32_t)p;
                    o_db = (d << 3) | (uint32_t)bb;
                    if (v > gbest) gbest = v;
                }
            }
        }
    }
    if (lane == 0) {
        P.origin_t = o_t;
        P.origin_db = o_db;
        P.err = 0;
        P.n_repair = 0xffffffffu;  // marker for the host's counters: scored by this kernel
    }
}

// best_pp walk from the origin (lib/nextcorrect.c:1907-1982 visits exactly these cells).  The walk is a pointer chase
// (each step needs the cell the previous one named) of one step per consensus position, so -- like the scoring DP -- it is
// cut at the scoring segments and done in three passes:
//   bt_spec   one workgroup per (pile, segment): where does the walk leave the segment, and after how many steps, for EVERY
//             cell of the segment's last column it could enter through (one lane per cell; the chases are independent, so
//             their memory latencies overlap); the segment that holds the origin walks from the origin instead;
//   bt_stitch one lane per pile chains the segments from the origin's downwards: entry cell, step count, output offset;
//   bt_emit   one lane per (pile, segment) repeats its segment's walk from the now known entry and writes the path items.
// A walk's state is the packed tag of the current cell (t_pos + 1, delta, base), exactly what best_pp stores.
constexpr uint32_t kBtCand = 192;        // entry cells tabulated per segment (6 x 32 deltas); wider last columns: see bt_stitch
constexpr uint32_t kBtUnspec = 0xffffffffu;
static_assert(kBtCand == (uint32_t)kBtSlots && kSegEnts == kColEnts, "host-side buffer sizes");

struct BtWalk {
    const uint32_t *cb, *bpp;
    // one step: cell of tag g -> its best_pp
    __device__ __forceinline__ uint32_t next(uint32_t g) const {
        const uint32_t t = (uint32_t)tag_tpos(g);
        return bpp[cb[t] + (g & 0x7ffu) - ((g & 0x7ffu) >> 3) * 2u];  // delta * 6 + base = (delta << 3 | base) - 2 * delta
    }
};

__global__ __launch_bounds__(192) void bt_spec_kernel(const PileDev *__restrict__ piles, const SegItem *__restrict__ items,
                                                       const uint32_t *__restrict__ max_size, const uint32_t *__restrict__ cell_base,
                                                       const uint32_t *__restrict__ cell_best_pp, uint32_t seg_len,
                                                       uint32_t *__restrict__ bt_exit, uint32_t *__restrict__ bt_steps) {
    const SegItem it = items[blockIdx.x];
    const PileDev &P = piles[it.pile];
    if (P.err == 2u || P.origin_t < 0) return;  // 2: not scored yet (waits for the int64 kernel's pass): nothing to walk
    const uint32_t c0 = it.seg * seg_len, c1 = it.seg + 1u == P.n_seg ? P.seed_len : c0 + seg_len;
    const uint32_t ot = (uint32_t)P.origin_t;
    if (ot < c0) return;  // the walk starts below this segment
    const size_t sidx = (size_t)(P.seg_off + it.seg);
    BtWalk W{cell_base + P.col_off, cell_best_pp + P.cell_off};
    uint32_t *ex = bt_exit + sidx * kBtCand, *stp = bt_steps + sidx * kBtCand;
    if (ot < c1) {  // the origin's segment: one walk, from the origin, result in slot 0
        if (threadIdx.x == 0) {
            uint32_t g = tag_pack((int32_t)ot, P.origin_db >> 3, P.origin_db & 7u), n = 0;
            do {
                g = W.next(g);
                n++;
            } while (g != kTagHead && (uint32_t)tag_tpos(g) >= c0);
            ex[0] = g, stp[0] = n;
        }
        return;
    }
    const uint32_t ncell = (max_size + P.col_off)[c1 - 1u] * 6u;
    if (ncell > kBtCand) {  // too many entry cells to tabulate: bt_stitch walks this segment itself
        if (threadIdx.x == 0) stp[0] = kBtUnspec;
        return;
    }
    const uint32_t r = threadIdx.x;
    if (r < ncell && r % 6u < 5u) {
        uint32_t g = tag_pack((int32_t)(c1 - 1u), r / 6u, r % 6u), n = 0;
        do {
            g = W.next(g);
            n++;
        } while (g != kTagHead && (uint32_t)tag_tpos(g) >= c0);
        ex[r] = g, stp[r] = n;
    }
}

__global__ __launch_bounds__(64) void bt_stitch_kernel(PileDev *__restrict__ piles, const uint32_t *__restrict__ cell_base,
                                                        const uint32_t *__restrict__ cell_best_pp, uint32_t seg_len,
                                                        const uint32_t *__restrict__ bt_exit, const uint32_t *__restrict__ bt_steps,
                                                        uint32_t *__restrict__ bt_entry, uint32_t *__restrict__ bt_off, int n_piles) {
    const int i = (int)(blockIdx.x * 64 + threadIdx.x);
    if (i >= n_piles) return;
    PileDev &P = piles[i];
    uint32_t *entry = bt_entry + P.seg_off, *off = bt_off + P.seg_off;
    for (uint32_t s_ = 0; s_ < P.n_seg; s_++) entry[s_] = kTagHead;  // segments the walk does not visit emit nothing
    if (P.err == 2u || P.origin_t < 0) {
        P.path_len = 0;
        return;
    }
    BtWalk W{cell_base + P.col_off, cell_best_pp + P.cell_off};
    uint32_t seg = (uint32_t)P.origin_t / seg_len;
    if (seg >= P.n_seg) seg = P.n_seg - 1u;  // the last segment takes the remainder
    uint32_t g = tag_pack(P.origin_t, P.origin_db >> 3, P.origin_db & 7u), len = 0, slot = 0;
    for (;;) {
        const size_t sidx = (size_t)(P.seg_off + seg);
        entry[seg] = g;
        off[seg] = len;
        uint32_t n = 0, gx;
        if (slot >= kBtCand || bt_steps[sidx * kBtCand] == kBtUnspec) {  // not tabulated: walk it here
            const uint32_t c0 = seg * seg_len;
            gx = g, n = 0;
            do {
                gx = W.next(gx);
                n++;
            } while (gx != kTagHead && (uint32_t)tag_tpos(gx) >= c0);
        } else {
            gx = bt_exit[sidx * kBtCand + slot];
            n = bt_steps[sidx * kBtCand + slot];
        }
        len += n;
        if (gx == kTagHead || seg == 0) break;
        g = gx;
        seg--;
        slot = tag_delta(g) * 6u + tag_base(g);
    }
    P.path_len = len;
}

__global__ __launch_bounds__(64) void bt_emit_kernel(const PileDev *__restrict__ piles, const SegItem *__restrict__ items, int n_items,
                                                      const uint32_t *__restrict__ coverage, const uint32_t *__restrict__ cell_base,
                                                      const uint32_t *__restrict__ cell_best_pp, const uint32_t *__restrict__ cell_best_link,
                                                      uint32_t seg_len, const uint32_t *__restrict__ bt_entry,
                                                      const uint32_t *__restrict__ bt_off, PathItem *__restrict__ path) {
    const int i = (int)(blockIdx.x * 64 + threadIdx.x);
    if (i >= n_items) return;
    const SegItem it = items[i];
    const PileDev &P = piles[it.pile];
    uint32_t g = bt_entry[P.seg_off + it.seg];
    if (g == kTagHead) return;
    const uint32_t c0 = it.seg * seg_len;
    const uint32_t *cb = cell_base + P.col_off, *cov = coverage + P.col_off;
    const uint32_t *bpp = cell_best_pp + P.cell_off, *blk = cell_best_link + P.cell_off;
    PathItem *out = path + P.path_off + bt_off[P.seg_off + it.seg];
    do {
        const uint32_t t = (uint32_t)tag_tpos(g);
        const uint32_t cell = cb[t] + tag_delta(g) * 6u + tag_base(g);
        PathItem pi;
        pi.tag = g;
        pi.link = (uint16_t)blk[cell];
        pi.cov = (uint16_t)cov[t];
        *out++ = pi;
        g = bpp[cell];
    } while (g != kTagHead && (uint32_t)tag_tpos(g) >= c0);
}

// ---- K11 -------------------------------------------------------------------------
__global__ __launch_bounds__(64) void extract_kernel(const PileDev *__restrict__ piles,
                                                      const ReadDev *__restrict__ reads,
                                                      const uint32_t *__restrict__ acc_list,
                                                      const uint32_t *__restrict__ tags,
                                                      const uint32_t *__restrict__ colidx, RegionDev *__restrict__ regions,
                                                      char *__restrict__ strpool, unsigned long long *__restrict__ cursor,
                                                      unsigned long long cap) {
    RegionDev &G = regions[blockIdx.x];
    const PileDev P = piles[G.pile];
    const int lane = (int)threadIdx.x;
    const uint32_t *acc = acc_list + P.acc_off;
    const uint32_t start = G.start, end = G.end;
    uint32_t ok_total = 0, large = 0;
    for (uint32_t r0 = 0; r0 < P.n_acc && ok_total < 40u; r0 += 64) {
        const uint32_t rank = r0 + (uint32_t)lane;
        int status = 0;  // 1: candidate, 2: longer than max_len - 1
        uint32_t len = 0, i0 = 0, i1 = 0;
        const uint32_t max_len = rank == 0 ? G.max_len0 : G.max_len;
        const uint32_t *tg = nullptr;
        if (rank < P.n_acc) {
            const ReadDev *R = &reads[acc[rank]];
            if (R->t_s <= start && R->t_e >= end) {  // lib/nextcorrect.c:377
                const uint32_t *ci = colidx + R->colidx_off;
                tg = tags + R->tag_off;
                i0 = ci[start - R->t_s];
                i1 = end == R->t_e ? R->aln_len : ci[end + 1 - R->t_s];
                for (uint32_t i = i0; i < i1; i++)
                    if ((tg[i] & 7u) != 4u) {
                        if (++len > max_len - 1u) {
                            status = 2;
                            break;
                        }
                    }
                if (status != 2 && len > 0) status = 1;
            }
        }
        unsigned long long okm = __ballot(status == 1), lgm = __ballot(status == 2);
        const uint32_t need = 40u - ok_total;  // stop right after the 40th candidate (lib/nextcorrect.c:402)
        const uint32_t my_rank = (uint32_t)__popcll(okm & lanes_le(lane));
        if ((uint32_t)__popcll(okm) >= need) {
            const unsigned long long cutm = __ballot(status == 1 && my_rank == need);
            const int cut = __ffsll((long long)cutm) - 1;
            okm &= lanes_le(cut);
            lgm &= lanes_le(cut);
        }
        if (status == 1 && ((okm >> lane) & 1ull)) {
            const uint32_t slot = ok_total + my_rank - 1u;
            const unsigned long long off = atomicAdd(cursor, (unsigned long long)len);
            G.cand_off[slot] = (uint32_t)off;
            G.cand_len[slot] = (uint16_t)len;
            G.cand_rank[slot] = (uint16_t)rank;
            if (off + len <= cap) {
                uint32_t k = 0;
                for (uint32_t i = i0; i < i1; i++) {
                    const uint32_t bs = tg[i] & 7u;
                    if (bs != 4u) strpool[off + k++] = "ATGC-NM"[bs];
                }
            }
        }
        ok_total += (uint32_t)__popcll(okm);
        large += (uint32_t)__popcll(lgm);
    }
    if (lane == 0) {
        G.n_ok = ok_total;
        G.n_large = large;
    }
}

}  // namespace

void launch_shift_scan(const AlnTask *tasks, const AlnOut *outs, const uint32_t *ops, ReadDev *reads, int n_reads,
                       void *stream) {
    if (n_reads <= 0) return;
    hipLaunchKernelGGL(shift_scan_kernel, dim3((unsigned)((n_reads + 63) / 64)), dim3(64), 0, (hipStream_t)stream, tasks,
                       outs, ops, reads, n_reads);
}

void launch_pile_accept(PileDev *piles, ReadDev *reads, uint32_t *acc_list, uint32_t *cov_diff, int n_piles,
                        void *stream) {
    if (n_piles <= 0) return;
    hipLaunchKernelGGL(pile_accept_kernel, dim3((unsigned)((n_piles + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                       piles, reads, acc_list, cov_diff, n_piles);
}

void launch_make_tags(const PileDev *piles, const ReadDev *reads, const AlnTask *tasks, const uint32_t *ops,
                      const uint32_t *pool, const uint32_t *db_pool, const uint32_t *read_pile, uint32_t *tags,
                      uint32_t *colidx, uint32_t *ins_count, uint32_t *ins_max, int n_reads, void *stream) {
    if (n_reads <= 0) return;
    hipLaunchKernelGGL(make_tags_kernel, dim3((unsigned)n_reads), dim3(64), 0, (hipStream_t)stream, piles, reads, tasks,
                       ops, pool, db_pool, read_pile, tags, colidx, ins_count, ins_max);
}

void launch_col_scan(PileDev *piles, uint32_t *cov_diff, const uint32_t *ins_count, uint32_t *ins_max,
                     uint32_t *cell_base, uint32_t *ent_base, int n_piles, void *stream) {
    if (n_piles <= 0) return;
    hipLaunchKernelGGL(col_scan_kernel, dim3((unsigned)n_piles), dim3(64), 0, (hipStream_t)stream, piles, cov_diff,
                       ins_count, ins_max, cell_base, ent_base);
}

void launch_count_links(const PileDev *piles, const ReadDev *reads, const uint32_t *acc_list, const ColBlock *blocks,
                        const uint32_t *tags, const uint32_t *colidx, const uint32_t *max_size,
                        const uint32_t *cell_base, const uint32_t *ent_base, uint32_t *cell_start, uint32_t *cell_len,
                        uint32_t *ent_pp, uint32_t *ent_ppp, uint32_t *ent_cnt, uint32_t *err, int n_blocks,
                        bool full_capacity, void *stream) {
    if (n_blocks <= 0) return;
    if (full_capacity)
        hipLaunchKernelGGL(count_links_kernel<kLinkCap>, dim3((unsigned)n_blocks), dim3(64), 0, (hipStream_t)stream, piles, reads,
                           acc_list, blocks, tags, colidx, max_size, cell_base, ent_base, cell_start, cell_len, ent_pp,
                           ent_ppp, ent_cnt, err);
    else
        hipLaunchKernelGGL(count_links_kernel<kLinkCapSmall>, dim3((unsigned)n_blocks), dim3(64), 0, (hipStream_t)stream, piles,
                           reads, acc_list, blocks, tags, colidx, max_size, cell_base, ent_base, cell_start, cell_len, ent_pp,
                           ent_ppp, ent_cnt, err);
}

void launch_score_backtrack(const K10Args &a, const SegItem *items_small, int n_small, const SegItem *items_large, int n_large,
                            const SegItem *items_all, int n_all, long long *ent_score, bool rescue, PathItem *path, uint32_t *bt_exit,
                            uint32_t *bt_steps, uint32_t *bt_entry, uint32_t *bt_off, int n_piles, void *stream,
                            void *ev_after_fast, void *stream_large, void *ev_fork, void *ev_join) {
    if (n_piles <= 0) return;
    hipStream_t st = (hipStream_t)stream;
    static const bool dbg = getenv("NDGPU_DEBUG_LAUNCH") != nullptr;
    auto mark = [&](const char *what) {
        if (!dbg) return;
        (void)hipStreamSynchronize(st);
        if (stream_large) (void)hipStreamSynchronize((hipStream_t)stream_large);
        fprintf(stderr, "[ndgpu dbg %p] k10: %s (small %d large %d all %d piles %d)\n", (void *)st, what, n_small, n_large, n_all, n_piles);
        fflush(stderr);
    };
    mark("begin");
    const bool forked = !rescue && n_large > 0 && stream_large && stream_large != stream;
    if (!rescue && n_large > 0) {  // the piles that need the large tables are scored at the same time on a second stream
        hipStream_t s2 = forked ? (hipStream_t)stream_large : st;
        if (forked) {
            (void)hipEventRecord((hipEvent_t)ev_fork, st);
            (void)hipStreamWaitEvent(s2, (hipEvent_t)ev_fork, 0);
        }
        hipLaunchKernelGGL((score_seg_kernel<kColCells, kColEnts>), dim3((unsigned)n_large), dim3(192), 0, s2, a, items_large);
        hipLaunchKernelGGL((score_stitch_kernel<kColCells, kColEnts, true>), dim3((unsigned)n_piles), dim3(192), 0, s2, a);
        if (forked) (void)hipEventRecord((hipEvent_t)ev_join, s2);
    }
    mark("large tier launched");
    if (!rescue && n_small > 0)
        hipLaunchKernelGGL((score_seg_kernel<kColCellsSmall, kColEntsSmall>), dim3((unsigned)n_small), dim3(192), 0, st, a, items_small);
    mark("small seg done");
    if (!rescue)
        hipLaunchKernelGGL((score_stitch_kernel<kColCellsSmall, kColEntsSmall, false>), dim3((unsigned)n_piles), dim3(192), 0, st, a);
    mark("small stitch done");
    if (forked) (void)hipStreamWaitEvent(st, (hipEvent_t)ev_join, 0);
    if (ev_after_fast) (void)hipEventRecord((hipEvent_t)ev_after_fast, st);
    if (ent_score)  // (nullptr: the int64 kernel's score array is not allocated; piles left at err == 2 get the rescue pass)
        hipLaunchKernelGGL(score_slow_kernel, dim3((unsigned)n_piles), dim3(64), 0, st, a.piles, a.coverage, a.max_size,
                       a.cell_base, a.cell_start, a.cell_len, a.ent_pp, a.ent_ppp, a.ent_cnt, ent_score, a.cell_best_pp,
                       a.cell_best_link);
    mark("slow done");
    // best_pp walk: every item of both lists (the segments of all piles the scoring kernels handled), plus the piles of
    // the int64 kernel, whose segments are in `items_all` too
    if (n_all > 0) {
        hipLaunchKernelGGL(bt_spec_kernel, dim3((unsigned)n_all), dim3(192), 0, st, a.piles, items_all, a.max_size, a.cell_base,
                           a.cell_best_pp, a.seg_len, bt_exit, bt_steps);
        mark("bt_spec done");
        hipLaunchKernelGGL(bt_stitch_kernel, dim3((unsigned)((n_piles + 63) / 64)), dim3(64), 0, st, a.piles, a.cell_base,
                           a.cell_best_pp, a.seg_len, bt_exit, bt_steps, bt_entry, bt_off, n_piles);
        mark("bt_stitch done");
        hipLaunchKernelGGL(bt_emit_kernel, dim3((unsigned)((n_all + 63) / 64)), dim3(64), 0, st, a.piles, items_all, n_all, a.coverage,
                           a.cell_base, a.cell_best_pp, a.cell_best_link, a.seg_len, bt_entry, bt_off, path);
    }
}

void launch_extract(const PileDev *piles, const ReadDev *reads, const uint32_t *acc_list, const uint32_t *tags,
                    const uint32_t *colidx, RegionDev *regions, char *strpool, unsigned long long *strpool_cursor,
                    unsigned long long strpool_cap, int n_regions, void *stream) {
    if (n_regions <= 0) return;
    hipLaunchKernelGGL(extract_kernel, dim3((unsigned)n_regions), dim3(64), 0, (hipStream_t)stream, piles, reads,
                       acc_list, tags, colidx, regions, strpool, strpool_cursor, strpool_cap);
}

}  // namespace ndgpu
