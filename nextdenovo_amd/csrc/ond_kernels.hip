// Hand-written CDNA4 (gfx950) kernels for the banded greedy O(ND) aligner that
// NextDenovo's consensus is built on (reference: core() via align()/align_hq(),
// lib/align.c:428-578).  Integer DP, no MFMA.
//
// K7  ond_forward   : one 64-lane wavefront per alignment, the furthest-reaching values
//                     of an edit step in registers (lane l = diagonal min_k + 2l, a second
//                     cell per lane on the steps wider than 64 cells); a step is one wave
//                     permute + one DPP shift (the reference's V[k-1], V[k+1]), a snake
//                     (XOR + ctz over 64 bases a round: five adjacent 16-base words per operand
//                     from the HBM-resident pool), two ballots (move bits -> trace record, finish test), one wave
//                     max (best anti-diagonal) and two ballots for the band re-centring.
//                     Traceback memory is 1 bit per evaluated cell + 8 bits of offset per
//                     step in a record stream, instead of the reference's one byte per cell
//                     of an O(max_d^2) triangle.
// K7w ond_forward_wide: V[] in a global scratch ring and band-cap-wide trace rows, for the
//                     rare alignments whose live band exceeds the register path (exactness
//                     of the band-cap / edit-budget failure semantics).
// K8a ond_traceback : one lane per alignment walks d -> 0 reading the move bits,
//                     re-deriving match runs with clz over 64 bases a round, and emits
//                     2-bit column kinds back to front.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "nd_device.h"

namespace ndgpu {

namespace {

// maximum over the 64 lanes (all of them active) with data-parallel-primitive moves instead of six LDS permutes: the
// forward kernel is bound by instruction issue, and this reduction sits on every edit step
__device__ __forceinline__ int wave_max_i32(int v) {
    auto step = [](int x, auto ctrl, auto rows) {
        const int y = __builtin_amdgcn_update_dpp(INT_MIN, x, decltype(ctrl)::value, decltype(rows)::value, 0xf, false);
        return y > x ? y : x;
    };
    v = step(v, std::integral_constant<int, 0xb1>{}, std::integral_constant<int, 0xf>{});   // quad_perm [1,0,3,2]
    v = step(v, std::integral_constant<int, 0x4e>{}, std::integral_constant<int, 0xf>{});   // quad_perm [2,3,0,1]
    v = step(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{});  // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{});  // row_mirror: every row of 16 reduced
    v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});  // row_bcast:15 into rows 1 and 3
    v = step(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});  // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// K7: one wavefront per alignment, the furthest-reaching values in REGISTERS.  Lane l owns cell l of an edit step (diagonal
// min_k + 2l) and, on the steps whose live band is wider than 64 cells, cell 64 + l as well.  Every value a step reads was
// written by the step before it: cell c' of step d + 1 is diagonal k' = min_k' + 2c' with min_k' = new_min - 1 =
// min_k + 2j - 1 (j = index of the first diagonal the re-centring keeps), so V[k' + 1] is cell j + c' of step d and V[k' - 1]
// cell j + c' - 1; the two edge diagonals never look outside (k' == min_k' takes V[k' + 1], k' == max_k' takes V[k' - 1] + 1:
// lib/align.c:443).  So the reference's V[] array is two registers per lane, moved by one wave permute with a uniform shift
// (ds_bpermute) and one wave_shr:1 DPP move: no LDS ring, no workgroup barrier, no second look at V[] for the re-centring.
//
// Trace: a stream of 64-bit words per alignment, one record per edit step, written front to back and read back to front by
// the traceback (which visits the steps d, d - 1, d - 2, ... without exception).  The LAST word of a record (the first one the
// traceback meets) is its header: bits [55:0] = move bits of cells 0..55, bit 56 = "this record has a second word", bits
// [63:57] = j (above).  Records of steps with more than 56 cells carry the move bits of cells 56..119 in the word before the
// header.  min_k is never stored: the traceback follows the CELL INDEX of its diagonal, idx(d - 1) = idx(d) + j(d - 1) - left.
// 8 bytes per step for bands up to 110 diagonals, 16 beyond: the algorithmic 1 bit per evaluated cell + the step's offset.
constexpr int kStreamBits = 56;                           // move bits in a header word
constexpr uint64_t kStreamMask = (1ull << kStreamBits) - 1;

__device__ __forceinline__ int wave_shr1(int first, int v) {  // lane l gets v of lane l - 1, lane 0 gets `first`
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false);
}

// 64 bases starting at base `pos` of a sequence (pos counts from the sequence's first word), as four words of 16: five words are
// read (the pools are padded for it: kPoolPadWords) and shifted into place
struct Bases64 { uint32_t w[4]; };
static_assert(kPoolPadWords >= 5, "fetch64 reads five words from the word that holds a sequence's last base");
__device__ __forceinline__ Bases64 fetch64(const uint32_t *__restrict__ p, uint32_t sub);
__device__ __forceinline__ Bases64 fetch64_rel(const uint32_t *__restrict__ seq, uint32_t pos) { return fetch64(seq + (pos >> 4), pos & 15u); }
__device__ __forceinline__ Bases64 fetch64_abs(const uint32_t *__restrict__ pool, uint64_t off) {  // off: base offset into the pool (64 bits: a
    return fetch64(pool + (off >> 4), (uint32_t)(off & 15u));                                       // resident read DB exceeds 2^32 bases)
}
__device__ __forceinline__ Bases64 fetch64(const uint32_t *__restrict__ p, uint32_t sub) {
    struct W4 { uint32_t x, y, z, w; } v;   // (five adjacent words, 4-byte aligned: the compiler merges them into wide loads)
    v.x = p[0], v.y = p[1], v.z = p[2], v.w = p[3];
    const uint32_t v4 = p[4];
    // ({hi, lo} >> s)[31:0] with s = 0, 2, ..., 30 is ONE v_alignbit_b32 (full rate); written as a 64-bit shift it compiled to
    // v_lshrrev_b64 -- eight of them per edit step in K7, eight in K8a, on kernels that are bound by instruction issue
    const uint32_t s = sub * 2u;
    Bases64 r;
    r.w[0] = __builtin_amdgcn_alignbit(v.y, v.x, s);
    r.w[1] = __builtin_amdgcn_alignbit(v.z, v.y, s);
    r.w[2] = __builtin_amdgcn_alignbit(v.w, v.z, s);
    r.w[3] = __builtin_amdgcn_alignbit(v4, v.w, s);
    return r;
}

// The snake of one cell (lib/align.c:452-455): how far the diagonal runs on equal bases from (x, x - k).  An edit step lasts as long as
// its slowest lane, and a lane that compares 16 bases per round trip to the cache needs another round for every 16 equal bases -- with
// 30-odd live diagonals nearly every step has a lane that needs three; 64 bases per round make the second round rare (a run of 64 equal
// bases between two reads with 10 % differences).  Same loads in bytes, a third of the dependent rounds.
__device__ __forceinline__ int snake64(const uint32_t *__restrict__ qp, const uint32_t *__restrict__ tp, uint32_t q_sh, uint32_t t_sh,
                                       int q_len, int t_len, int x, int k) {
    int y = x - k;
    for (;;) {
        int rem = q_len - x;
        const int rt = t_len - y;
        rem = rt < rem ? rt : rem;
        if (rem <= 0) break;
        const Bases64 a = fetch64_rel(qp, q_sh + (uint32_t)x);
        const Bases64 b = fetch64_rel(tp, t_sh + (uint32_t)y);
        const uint32_t d0 = a.w[0] ^ b.w[0], d1 = a.w[1] ^ b.w[1], d2 = a.w[2] ^ b.w[2], d3 = a.w[3] ^ b.w[3];
        int m = d0 ? (__builtin_ctz(d0) >> 1) : d1 ? 16 + (__builtin_ctz(d1) >> 1) : d2 ? 32 + (__builtin_ctz(d2) >> 1) : d3 ? 48 + (__builtin_ctz(d3) >> 1) : 64;
        m = m < rem ? m : rem;
        x += m;
        y += m;
        if (m < 64) break;
    }
    return x;
}

// CKPT: the kernel also leaves what the segmented traceback starts from (nd_device.h: TbSeg).  Every cell carries `org`, the cell of
// the last checkpoint row its move bits lead back to: it rides the moves V takes (one more permute, one more DPP shift, one select per
// step); every 2^cshift steps the cells store (x, org) and start again from their own index.  The kernel also zeroes the task's column
// words: walkers that share a word OR their columns into it.
template <bool CKPT>
__global__ __launch_bounds__(64) void ond_forward_kernel(const AlnTask *__restrict__ tasks, AlnOut *__restrict__ outs,
                                                          const uint32_t *__restrict__ pool,
                                                          const uint32_t *__restrict__ db_pool,
                                                          uint64_t *__restrict__ trace, const int32_t *__restrict__ order,
                                                          uint32_t *__restrict__ ck_cells, uint2 *__restrict__ ck_hdr,
                                                          uint32_t *__restrict__ ops, int cshift) {
    // `order` lists the launch's tasks longest first: workgroups are dispatched in index order, so the long dependent chains
    // start first and the short ones fill in behind them
    const int tid = order ? order[blockIdx.x] : (int)blockIdx.x;
    const AlnTask T = tasks[tid];
    const int lane = (int)threadIdx.x;
    // An alignment is a chain of dependent edit steps, and the launch (and whatever waits for it) lasts as long as its longest
    // chain; a wavefront that shares its SIMD with seven others gets an eighth of the issue slots.  The long chains therefore
    // ask the SIMD's arbiter for priority over the short ones -- of this launch and of the other contexts' kernels alike.
    {
        const int total = T.q_len + T.t_len;
        if (total > 160000) __builtin_amdgcn_s_setprio(3);
        else if (total > 80000) __builtin_amdgcn_s_setprio(2);
        else if (total > 40000) __builtin_amdgcn_s_setprio(1);
    }

    int min_k = 0, max_k = 0, best_m = -1;
    int status = ST_NONE, fin_k = 0, fin_x = 0, fin_d = -1, fin_idx = 0;
    int d_steps = 0, max_band = 0;
    long long cells = 0;
    const int q_len = T.q_len, t_len = T.t_len;
    // bit 63 of an offset selects the resident read DB instead of the per-batch pool; inside the kernel a base is addressed
    // by a 32-bit word index relative to its sequence's first word (a sequence is < 2^31 bases)
    const uint64_t q_off = T.q_off & kOffMask, t_off = T.t_off & kOffMask;
    const uint32_t *__restrict__ qp = ((T.q_off >> 63) ? db_pool : pool) + (q_off >> 4);
    const uint32_t *__restrict__ tp = ((T.t_off >> 63) ? db_pool : pool) + (t_off >> 4);
    const uint32_t q_sh = (uint32_t)(q_off & 15u), t_sh = (uint32_t)(t_off & 15u);
    uint64_t *__restrict__ S = trace + T.trace_off;
    // The record words wait in a register of the lanes -- lane l holds the l-th word not yet written -- and go out 63 or 64 at a
    // time, one 512-byte store.  Written step by step (8 bytes from lane 0) the store sat in front of the next step's loads, and a
    // load's s_waitcnt waits for every memory operation issued before it: each step waited for its predecessor's store as well.
    uint64_t rec = 0;
    uint32_t pos_base = 0, fill = 0;   // words written, words waiting (the same in every lane); the stream's length is their sum
    auto flush = [&]() {
        if ((uint32_t)lane < fill) S[pos_base + (uint32_t)lane] = rec;
        pos_base += fill;
        fill = 0;
    };
    auto put = [&](bool wide, uint64_t hi, uint64_t lo) {
        if (wide) {
            if ((uint32_t)lane == fill) rec = hi;
            fill++;
        }
        if ((uint32_t)lane == fill) rec = lo;
        fill++;
        if (fill >= 63u) flush();   // (room for a two-word record is kept)
    };
    if (CKPT) {
        uint32_t *__restrict__ W = ops + T.ops_off;
        const uint32_t nw = (T.ops_cap + 15u) / 16u + 1u;
        for (uint32_t i = (uint32_t)lane; i < nw; i += 64u) W[i] = 0u;
    }
    const int cmask = CKPT ? (1 << cshift) - 1 : 0;
    int porg0 = 0, porg1 = 0;  // the step before: org of cell `lane` / of cell 64 + `lane`
    int fin_org = 0;

    int px0 = 0, px1 = 0;  // the step before: x of cell `lane` / of cell 64 + `lane` (the reference memsets V: all zero)
    int pj = 0;            // its j
    bool pwide = false;    // it had more than 64 cells

    for (int d = 0; d < T.max_d && max_k - min_k <= T.band; d++) {
        const int band = max_k - min_k;
        if (band > kFastMaxBand) {
            status = ST_NEED_WIDE;
            break;
        }
        const int ncell = (band >> 1) + 1;  // (the band never shrinks below 0: the row's best diagonal always survives)
        const bool two = ncell > 64;
        d_steps++;
        cells += ncell;
        max_band = band > max_band ? band : max_band;

        // ---- cells 0..63
        const int src = pj + lane;
        int vp = __shfl(px0, src & 63, 64);
        int op = CKPT ? __shfl(porg0, src & 63, 64) : 0;
        if (pwide) {
            const int hi = __shfl(px1, src & 63, 64);
            vp = src >= 64 ? hi : vp;
            if (CKPT) {
                const int ohi = __shfl(porg1, src & 63, 64);
                op = src >= 64 ? ohi : op;
            }
        }
        const int vm = wave_shr1(0, vp);
        const int om = CKPT ? wave_shr1(0, op) : 0;
        const int k0 = min_k + 2 * lane;
        const bool act0 = k0 <= max_k;
        int x0 = 0, org0 = 0;
        bool left0 = false;
        if (act0) {
            const bool down = (k0 == min_k) || (k0 != max_k && vm < vp);  // lib/align.c:443
            left0 = !down;
            if (CKPT) org0 = down ? op : om;
            x0 = snake64(qp, tp, q_sh, t_sh, q_len, t_len, down ? vp : vm + 1, k0);
        }
        const unsigned long long lb0 = __ballot(act0 && left0);
        const unsigned long long fb0 = __ballot(act0 && x0 >= q_len && x0 - k0 >= t_len);
        int row_best = act0 ? 2 * x0 - k0 : -1;
        // ---- cells 64..127 (only the steps whose band is wider than 126 diagonals)
        int x1 = 0, k1 = 0, org1 = 0;
        bool act1 = false;
        unsigned long long lb1 = 0, fb1 = 0;
        if (two && !fb0) {
            const int vp1 = __shfl(px1, src & 63, 64);       // cell 64 + pj + lane of the step before
            const int vm1 = wave_shr1(__builtin_amdgcn_readlane(vp, 63), vp1);
            const int op1 = CKPT ? __shfl(porg1, src & 63, 64) : 0;
            const int om1 = CKPT ? wave_shr1(__builtin_amdgcn_readlane(op, 63), op1) : 0;
            k1 = k0 + 128;
            act1 = k1 <= max_k;
            bool left1 = false;
            if (act1) {
                const bool down = k1 != max_k && vm1 < vp1;
                left1 = !down;
                if (CKPT) org1 = down ? op1 : om1;
                x1 = snake64(qp, tp, q_sh, t_sh, q_len, t_len, down ? vp1 : vm1 + 1, k1);
                const int m1 = 2 * x1 - k1;
                row_best = m1 > row_best ? m1 : row_best;
            }
            lb1 = __ballot(act1 && left1);
            fb1 = __ballot(act1 && x1 >= q_len && x1 - k1 >= t_len);
        }

        const bool wide_rec = ncell > kStreamBits;
        const uint64_t bits_lo = lb0 & kStreamMask, bits_hi = (lb0 >> kStreamBits) | (lb1 << (64 - kStreamBits));
        if (fb0 | fb1) {
            // several diagonals may finish in one step: the smallest k wins (lib/align.c:467-470)
            const int fl = fb0 ? __ffsll((long long)fb0) - 1 : __ffsll((long long)fb1) - 1;
            fin_idx = fb0 ? fl : 64 + fl;
            fin_k = min_k + 2 * fin_idx;
            fin_x = fb0 ? __shfl(x0, fl, 64) : __shfl(x1, fl, 64);
            if (CKPT) fin_org = fb0 ? __shfl(org0, fl, 64) : __shfl(org1, fl, 64);
            fin_d = d;
            status = ST_FINISHED;
            put(wide_rec, bits_hi, ((uint64_t)wide_rec << kStreamBits) | bits_lo);
            break;
        }

        const int rb = wave_max_i32(row_best);
        best_m = rb > best_m ? rb : best_m;

        // band re-centring (lib/align.c:473-489): first and last diagonal within 150 of the best anti-diagonal
        const int thr = best_m - 150;
        const bool ok0 = act0 && 2 * x0 - k0 >= thr, ok1 = act1 && 2 * x1 - k1 >= thr;
        const unsigned long long qlo0 = __ballot(ok0 && k0 < max_k), qhi0 = __ballot(ok0 && k0 > min_k);
        unsigned long long qlo1 = 0, qhi1 = 0;
        if (two) {
            qlo1 = __ballot(ok1 && k1 < max_k);
            qhi1 = __ballot(ok1);
        }
        int j = ncell - 1, jmax = 0;  // new_min = max_k, new_max = min_k when nothing qualifies
        if (qlo0) j = __ffsll((long long)qlo0) - 1;
        else if (qlo1) j = 64 + __ffsll((long long)qlo1) - 1;
        if (qhi1) jmax = 64 + 63 - __clzll((long long)qhi1);
        else if (qhi0) jmax = 63 - __clzll((long long)qhi0);

        put(wide_rec, bits_hi, ((uint64_t)j << (kStreamBits + 1)) | ((uint64_t)wide_rec << kStreamBits) | bits_lo);

        if (CKPT && d && (d & cmask) == 0) {  // a checkpoint row: what a walker that starts at the top of this row needs
            const uint64_t slot = T.mink_off + (uint64_t)((d >> cshift) - 1);
            if (act0) ck_cells[slot * kCkptCells + (uint32_t)lane] = (uint32_t)x0 | ((uint32_t)org0 << 24);
            if (act1) ck_cells[slot * kCkptCells + 64u + (uint32_t)lane] = (uint32_t)x1 | ((uint32_t)org1 << 24);
            if (lane == 0) ck_hdr[slot] = make_uint2(pos_base + fill, (uint32_t)min_k);
            org0 = lane;
            org1 = 64 + lane;
        }

        max_k = min_k + 2 * jmax + 1;
        min_k = min_k + 2 * j - 1;
        px0 = x0;
        px1 = x1;
        porg0 = org0;
        porg1 = org1;
        pj = j;
        pwide = two;
    }

    const uint32_t pos = pos_base + fill;
    flush();
    if (lane == 0) {
        AlnOut o;
        o.status = status;
        o.d_final = fin_d;
        o.k_final = fin_k;
        o.x_final = fin_x;
        o.y_final = fin_x - fin_k;
        o.n_cols = 0;
        o.d_steps = d_steps;
        o.max_band = max_band;
        o.cells = cells;
        o.trace_end = pos;
        o.fin_idx = fin_idx | (fin_org << 8);
        outs[tid] = o;
    }
}

// K7w: the same sweep with V[] in a global scratch ring and trace rows as wide as the band cap, for the rare alignments whose
// live band exceeds the register path (exactness of the band-cap / edit-budget failure semantics).  Trace: row d = row_words
// 64-bit words of move bits (bit c = diagonal min_k + 2c) + min_k in a side array.
// LDSV: the V ring in the LDS of the compute unit (the launch's largest ring is at most 64 KB).  A launch of this path is a few dozen
// wavefronts, each with a compute unit to itself, and lasts as long as its longest alignment: with the ring in global memory every
// edit step read V[k - 1], V[k + 1] from and wrote V[k] to the caches and waited for the stores before the next step read them.
constexpr int kWidePasses = 4;   // passes of a wide step whose snakes share a round of loads
template <bool LDSV>
__global__ __launch_bounds__(64) void ond_forward_wide_kernel(const AlnTask *__restrict__ tasks, AlnOut *__restrict__ outs,
                                                          const uint32_t *__restrict__ pool,
                                                          const uint32_t *__restrict__ db_pool,
                                                          uint64_t *__restrict__ trace,
                                                          int32_t *__restrict__ trace_mink,
                                                          int32_t *__restrict__ vscratch,
                                                          const int32_t *__restrict__ ids, const AlnTask *__restrict__ wtasks) {
    // (wtasks: the listed tasks' records with the wide path's fields -- rows, min_k, V ring -- in list order; the table itself keeps the
    // register path's: one upload instead of one per task)
    const int tid = ids[blockIdx.x];
    const AlnTask T = wtasks ? wtasks[blockIdx.x] : tasks[tid];
    const int lane = (int)threadIdx.x;
    extern __shared__ int32_t wide_v[];
    int32_t *V = LDSV ? wide_v : vscratch + T.v_off;
    const uint32_t vmask = T.v_mask;

    for (uint32_t i = (uint32_t)lane; i <= vmask; i += 64) V[i] = 0;  // the reference memsets V per alignment
    __syncthreads();

    int min_k = 0, max_k = 0, best_m = -1;
    int status = ST_NONE, fin_k = 0, fin_x = 0, fin_d = -1;
    int d_steps = 0, max_band = 0;
    long long cells = 0;
    const int q_len = T.q_len, t_len = T.t_len;
    // bit 63 of an offset selects the resident read DB instead of the per-batch pool; inside the kernel a base is addressed
    // by a 32-bit word index relative to its sequence's first word (a sequence is < 2^31 bases)
    const uint64_t q_off = T.q_off & kOffMask, t_off = T.t_off & kOffMask;
    const uint32_t *__restrict__ qp = ((T.q_off >> 63) ? db_pool : pool) + (q_off >> 4);
    const uint32_t *__restrict__ tp = ((T.t_off >> 63) ? db_pool : pool) + (t_off >> 4);
    const uint32_t q_sh = (uint32_t)(q_off & 15u), t_sh = (uint32_t)(t_off & 15u);
    const uint64_t row0 = T.trace_off, mk0 = T.mink_off;
    const uint32_t row_words = T.row_words;

    for (int d = 0; d < T.max_d && max_k - min_k <= T.band; d++) {
        const int band = max_k - min_k;
        const int ncell = band >= 0 ? (band >> 1) + 1 : 0;
        const int npass = (ncell + 63) >> 6;
        d_steps++;
        cells += ncell;
        max_band = band > max_band ? band : max_band;
        if (lane == 0) trace_mink[mk0 + d] = min_k;

        int row_best = -1;
        bool done = false;
        int x_keep = 0;
        // The passes of a step are independent (every cell reads the step before: the other parity of V[]), so kWidePasses of them
        // go side by side: their starts, then the first 64 bases of all their snakes in ONE round of loads -- every lane loads, a
        // lane without a snake loads its sequence's first words: a load under a condition meets its default at a join and the join
        // waits -- and then, pass by pass in order, what has an order: move bits, V[], the finish (smallest k first).
        for (int ps0 = 0; ps0 < npass && !done; ps0 += kWidePasses) {
            int xs[kWidePasses], rems[kWidePasses];
            bool lefts[kWidePasses], acts[kWidePasses];
            Bases64 qa[kWidePasses], tb[kWidePasses];
#pragma unroll
            for (int u = 0; u < kWidePasses; u++) {
                const int k = min_k + 2 * ((ps0 + u) * 64 + lane);
                const bool act = ps0 + u < npass && k <= max_k;
                int x = 0;
                bool left = false;
                if (act) {
                    const int vm = V[(uint32_t)(k - 1) & vmask];
                    const int vp = V[(uint32_t)(k + 1) & vmask];
                    // lib/align.c:443
                    const bool down = (k == min_k) || (k != max_k && vm < vp);
                    x = down ? vp : vm + 1;
                    left = !down;
                }
                int rem = q_len - x;
                const int rt = t_len - (x - k);
                rem = rt < rem ? rt : rem;
                if (!act) rem = 0;
                xs[u] = x, lefts[u] = left, acts[u] = act, rems[u] = rem;
                qa[u] = fetch64_rel(qp, q_sh + (uint32_t)(rem > 0 ? x : 0));
                tb[u] = fetch64_rel(tp, t_sh + (uint32_t)(rem > 0 ? x - k : 0));
            }
#pragma unroll
            for (int u = 0; u < kWidePasses; u++) {
                if (rems[u] > 0) {  // (lib/align.c:452-455)
                    const uint32_t d0 = qa[u].w[0] ^ tb[u].w[0], d1 = qa[u].w[1] ^ tb[u].w[1], d2 = qa[u].w[2] ^ tb[u].w[2], d3 = qa[u].w[3] ^ tb[u].w[3];
                    int m = d0 ? (__builtin_ctz(d0) >> 1) : d1 ? 16 + (__builtin_ctz(d1) >> 1) : d2 ? 32 + (__builtin_ctz(d2) >> 1) : d3 ? 48 + (__builtin_ctz(d3) >> 1) : 64;
                    m = m < rems[u] ? m : rems[u];
                    xs[u] += m;
                    if (m == 64) xs[u] = snake64(qp, tp, q_sh, t_sh, q_len, t_len, xs[u], min_k + 2 * ((ps0 + u) * 64 + lane));
                }
            }
#pragma unroll
            for (int u = 0; u < kWidePasses; u++) {
                const int ps = ps0 + u;
                if (ps >= npass || done) continue;
                const int k = min_k + 2 * (ps * 64 + lane);
                const bool act = acts[u];
                const int x = xs[u];
                const unsigned long long lb = __ballot(act && lefts[u]);
                if (lane == 0) trace[row0 + (uint64_t)d * row_words + (uint32_t)ps] = lb;
                const int y = x - k;
                const unsigned long long fb = __ballot(act && x >= q_len && y >= t_len);
                if (act) {
                    V[(uint32_t)k & vmask] = x;
                    const int m = x + y;
                    row_best = m > row_best ? m : row_best;
                }
                x_keep = x;
                if (fb) {
                    // several diagonals may finish in one step: the smallest k wins (lib/align.c:467-470)
                    const int fl = __ffsll((long long)fb) - 1;
                    fin_k = min_k + 2 * (ps * 64 + fl);
                    fin_x = __shfl(x, fl, 64);
                    fin_d = d;
                    status = ST_FINISHED;
                    done = true;
                }
            }
        }
        if (done) break;

        const int rb = wave_max_i32(row_best);
        best_m = rb > best_m ? rb : best_m;
        __syncthreads();  // V[] of this step visible to every lane

        // band re-centring (lib/align.c:473-489)
        int new_min = max_k, new_max = min_k;
        const int thr = best_m - 150;
        for (int ps = 0; ps < npass; ps++) {
            const int k = min_k + 2 * (ps * 64 + lane);
            const int xv = npass == 1 ? x_keep : V[(uint32_t)k & vmask];  // single pass: x is still in a register
            const bool q = k < max_k && (2 * xv - k >= thr);
            const unsigned long long qb = __ballot(q);
            if (qb) {
                new_min = min_k + 2 * (ps * 64 + (__ffsll((long long)qb) - 1));
                break;
            }
        }
        for (int ps = npass - 1; ps >= 0; ps--) {
            const int k = min_k + 2 * (ps * 64 + lane);
            const int xv = npass == 1 ? x_keep : V[(uint32_t)k & vmask];
            const bool q = k <= max_k && k > min_k && (2 * xv - k >= thr);
            const unsigned long long qb = __ballot(q);
            if (qb) {
                new_max = min_k + 2 * (ps * 64 + (63 - __clzll((long long)qb)));
                break;
            }
        }
        max_k = new_max + 1;
        min_k = new_min - 1;
    }

    if (lane == 0) {
        AlnOut o;
        o.status = status;
        o.d_final = fin_d;
        o.k_final = fin_k;
        o.x_final = fin_x;
        o.y_final = fin_x - fin_k;
        o.n_cols = 0;
        o.d_steps = d_steps;
        o.max_band = max_band;
        o.cells = cells;
        o.trace_end = 0;
        o.fin_idx = 0;
        outs[tid] = o;
    }
}

// K8a: one lane per alignment walks d -> 0.  STREAM: the register path's record stream (see K7), read back to front -- the
// header of the current step's record is kept in a register, its cell index follows the walk; otherwise the wide kernel's rows.
template <bool STREAM>
__global__ __launch_bounds__(64) void ond_traceback_kernel(const AlnTask *__restrict__ tasks, AlnOut *__restrict__ outs,
                                                            const uint32_t *__restrict__ pool,
                                                            const uint32_t *__restrict__ db_pool,
                                                            const uint64_t *__restrict__ trace,
                                                            const int32_t *__restrict__ trace_mink,
                                                            uint32_t *__restrict__ ops,
                                                            const int32_t *__restrict__ ids, int n_tasks,
                                                            const AlnTask *__restrict__ wtasks = nullptr) {
    const int slot = (int)(blockIdx.x * 64 + threadIdx.x);
    if (slot >= n_tasks) return;
    const int gid = ids ? ids[slot] : slot;
    {  // a wavefront of long walks asks for priority over short ones (see K7); its lanes are neighbours of one pile
        const int total = (int)__builtin_amdgcn_readfirstlane(tasks[gid].ops_cap);
        if (total > 160000) __builtin_amdgcn_s_setprio(3);
        else if (total > 80000) __builtin_amdgcn_s_setprio(2);
        else if (total > 40000) __builtin_amdgcn_s_setprio(1);
    }
    if (outs[gid].status != ST_FINISHED) return;
    const AlnTask T = (!STREAM && wtasks) ? wtasks[slot] : tasks[gid];
    const uint32_t *__restrict__ qp = (T.q_off >> 63) ? db_pool : pool;
    const uint32_t *__restrict__ tp = (T.t_off >> 63) ? db_pool : pool;
    const uint64_t q_off = T.q_off & kOffMask, t_off = T.t_off & kOffMask;
    int x = outs[gid].x_final - 1;  // 0-based last query base (lib/align.c:492)
    int k = outs[gid].k_final;
    int d = outs[gid].d_final;
    int gap = 0;
    uint32_t col = T.ops_cap;  // columns [col, ops_cap) are written
    uint32_t acc = 0;
    uint32_t *W = ops + T.ops_off;
    bool aborted = false;
    const uint64_t *__restrict__ S = trace + T.trace_off;
    uint32_t pos = STREAM ? (uint32_t)outs[gid].trace_end : 0u;  // one past the record of step d
    int idx = outs[gid].fin_idx & 0xff;                           // cell index of diagonal k in step d
    uint64_t hdr = (STREAM && pos) ? S[pos - 1] : 0ull;
    // the two words before the header: the second word of this record if it has one, and -- whichever length it has -- the header of the
    // record before it.  Loaded a step ahead of their use, so that the walk never waits for the record it steps into.
    uint64_t c1 = (STREAM && pos >= 2) ? S[pos - 2] : 0ull, c2 = (STREAM && pos >= 3) ? S[pos - 3] : 0ull;

    for (;;) {
        // match run, back to front (lib/align.c:502-507), 64 bases per compare: every lane walks an alignment of its own and the
        // wavefront repeats this loop until its slowest lane is through -- with 16 bases a round, some lane of 64 nearly always
        // needs a third round (a run of 32 equal bases); with 64 a second round is rare
        for (;;) {
            const int yy = x - k;
            const int avail = (x < yy ? x : yy) + 1;
            if (avail <= 0) break;
            const int n = avail < 64 ? avail : 64;
            const Bases64 a = fetch64_abs(qp, q_off + (uint64_t)(uint32_t)(x - n + 1));
            const Bases64 b = fetch64_abs(tp, t_off + (uint64_t)(uint32_t)(yy - n + 1));
            int m = n;  // bases [0, n) of the fetch are the run's candidates, the last one first
#pragma unroll
            for (int i = 3; i >= 0; --i) {
                const int nb = n - 16 * i;  // candidates in word i
                if (nb <= 0) continue;
                uint32_t diff = a.w[i] ^ b.w[i];
                if (nb < 16) diff &= (1u << (2 * nb)) - 1u;
                if (diff) {
                    m = n - 1 - (16 * i + ((31 - __builtin_clz(diff)) >> 1));
                    break;
                }
            }
            if (m) {
                int left_to_emit = m;  // match columns are code 0: only the cursor moves
                while (left_to_emit > 0) {
                    const uint32_t room = ((col - 1u) & 15u) + 1u;
                    const uint32_t take = (uint32_t)left_to_emit < room ? (uint32_t)left_to_emit : room;
                    col -= take;
                    left_to_emit -= (int)take;
                    if ((col & 15u) == 0) {
                        W[col >> 4] = acc;
                        acc = 0;
                    }
                }
                x -= m;
                gap = 0;
            }
            if (m < n) break;
        }
        if (x < 0 && x - k < 0) break;
        bool left;
        if (x < k) left = true;  // lib/align.c:512: forced query-consuming move
        else if (x >= 0) {
            if (STREAM) {
                const bool second = idx >= kStreamBits && pos >= 2;
                const uint64_t w = second ? c1 : hdr;
                left = (w >> ((second ? idx - kStreamBits : idx) & 63)) & 1ull;
            } else {
                const int ix = (k - trace_mink[T.mink_off + (uint64_t)(uint32_t)d]) >> 1;
                left = (trace[T.trace_off + (uint64_t)(uint32_t)d * T.row_words + (uint32_t)(ix >> 6)] >> (ix & 63)) & 1ull;
            }
        } else left = false;
        uint32_t code;
        int nk, nx;
        if (left) { nk = k - 1; nx = x - 1; code = 1u; if (x < 0) gap = 260; }
        else { nk = k + 1; nx = x; code = 2u; if (x - k < 0) gap = 260; }
        if (gap < 260) {
            col--;
            acc |= code << ((col & 15u) * 2u);
            if ((col & 15u) == 0) {
                W[col >> 4] = acc;
                acc = 0;
            }
        }
        if (gap++ > 250) {  // lib/align.c:542-545
            aborted = true;
            break;
        }
        d--;
        k = nk;
        x = nx;
        if (STREAM) {  // step d - 1: its record ends where this one began; idx(d - 1) = idx(d) + j(d - 1) - left
            const uint32_t len = 1u + (uint32_t)((hdr >> kStreamBits) & 1ull);
            pos = pos > len ? pos - len : 0u;
            hdr = pos ? (len == 1u ? c1 : c2) : 0ull;   // = S[pos - 1]
            idx += (int)(hdr >> (kStreamBits + 1)) - (left ? 1 : 0);
            c1 = pos >= 2 ? S[pos - 2] : 0ull;
            c2 = pos >= 3 ? S[pos - 3] : 0ull;
        }
    }
    if ((col & 15u) != 0) W[col >> 4] = acc;
    outs[gid].n_cols = aborted ? 2 : (int32_t)(T.ops_cap - col);
    outs[gid].status = aborted ? ST_GAP_ABORT : ST_ALIGNED;
}


// ---- K8a in segments -----------------------------------------------------------------------------------------------------------
// The one-lane walk above is a chain of d dependent steps (5 x 10^4 for a 200 kb pair) on a device with half a million lanes.  Cut
// it: the forward kernel left a checkpoint every C = 2^cshift steps (see K7, CKPT), the chase below names the cell the walk passes
// through at every checkpoint row, and one lane per checkpoint -- a WALKER -- walks C rows.
//
// What a walker cannot know is its x: the reference's traceback extends a match run as far back as the bases agree
// (lib/align.c:502-507), which may be further than the forward snake of that cell began, so at the top of a row the walk stands at
// V[d][k] - 1 - o with an overshoot o >= 0 that depends on the rows above.  But o is forgotten quickly: wherever o(d) does not exceed
// the length of cell (d, k)'s snake, o(d - 1) is a property of that cell alone.  So a walker starts `warm` rows ABOVE the rows it
// owns with o = 0, walks them without output, and owns the rows from there on.  Nothing of that is assumed: every walker reports the
// state (x, k) it reached at the top of its first owned row and the one it left behind its last, the stitch compares neighbours, and
// a task with a boundary that does not agree (or with any other irregularity) is walked again by the one-lane kernel, which
// overwrites every column word of the task.  The forced moves of lib/align.c:512 (x < k) leave the move bits' path; they are caught
// by the same comparison (k is compared too).
//
// Where a column goes needs no count of the columns above it: from the finishing cell (x_f, y_f, d_f) to the top of row d at (x, y)
// the walk has emitted ((x_f - 1 - x) + (y_f - 1 - y) + (d_f - d)) / 2 columns -- a match column consumes a base of both sequences,
// a gap column one base and one row.  Walkers that share a 16-column word OR their part into it (the forward kernel zeroed the words).
// The > 250-gap-columns abort (lib/align.c:542-545) counts across walkers: each reports the gap columns before its first match run
// and behind its last, the stitch carries the count.

__global__ __launch_bounds__(64) void tb_chase_kernel(const AlnTask *__restrict__ tasks, const AlnOut *__restrict__ outs,
                                                       const uint32_t *__restrict__ ck_cells, const uint2 *__restrict__ ck_hdr,
                                                       TbSeg *__restrict__ segs, int n_tasks, int cshift, int warm) {
    const int gid = (int)(blockIdx.x * 64 + threadIdx.x);
    if (gid >= n_tasks) return;
    const AlnTask T = tasks[gid];
    const AlnOut O = outs[gid];
    TbSeg *__restrict__ G = segs + T.seg_off;
    const int cap = (T.max_d > 0 ? (T.max_d - 1) >> cshift : 0) + 1;  // walker slots of the task
    int live = 0;
    if (O.status == ST_FINISHED) {
        const int n = O.d_final > 0 ? (O.d_final - 1) >> cshift : 0;  // checkpoint rows under the finishing step: C, 2C, ..., nC
        const int C = 1 << cshift;
        live = n + 1;
        TbSeg g;
        g.task = gid;
        g.d_top = g.d_own = O.d_final;
        g.d_end = n ? n * C - warm + 1 : 0;
        g.x = O.x_final - 1;  // 0-based last query base (lib/align.c:492)
        g.idx = O.fin_idx & 0xff;
        g.min_k = O.k_final - 2 * g.idx;
        g.pos = O.trace_end;
        G[0] = g;
        uint32_t cell = ((uint32_t)O.fin_idx >> 8) & 0xffu;
        for (int i = n; i >= 1; --i) {
            const uint64_t slot = T.mink_off + (uint64_t)(i - 1);
            const uint32_t c = ck_cells[slot * kCkptCells + cell];
            const uint2 h = ck_hdr[slot];
            g.d_top = i * C;
            g.d_own = i * C - warm;
            g.d_end = i > 1 ? (i - 1) * C - warm + 1 : 0;
            g.x = (int)(c & 0xffffffu) - 1;
            g.idx = (int)cell;
            g.min_k = (int)h.y;
            g.pos = h.x;
            G[n - i + 1] = g;
            cell = c >> 24;
        }
    }
    for (int w = live; w < cap; w++) G[w].task = -1;
}

// the record stream read back to front (see the one-lane kernel): header of the current row, the two words before it
struct TbCursor {
    uint32_t pos;
    uint64_t hdr, c1, c2;
};

// A walker's windows in LDS.  Every lane of a walker wavefront reads a trace and two sequences of its own, back to front, a few
// bytes per row: as loads from memory that is six instructions per row with 64 different cache lines each -- the address path
// serves one line per cycle, and the lines a compute unit's 32 resident wavefronts hold open (6000) outgrow its caches, so most
// of them come from HBM again and again (profiles/pmc_traffic.json: 30 x the algorithmic bytes).  Here a lane copies the next
// kWinSeqWords words of each sequence and kWinTraceWords trace records behind its cursor into a column of LDS ([word][lane]: the
// lanes of a read fall on different banks) with full-line loads, once per 20-odd rows, and reads its rows from there.
// (Sizes from a sweep on the device, profiles/r06_tb_walk_windows.txt: 10 KB of LDS per wavefront; larger windows are refilled less often
// but leave room for fewer wavefronts, and the wavefronts are what hides a refill.)
constexpr int kWinSeqWords = 12;    // 192 bases of a sequence: the five words of a fetch and seven more behind them
constexpr int kWinTraceWords = 8;   // 8 record words
static_assert(kWinSeqWords % 4 == 0 && kWinSeqWords >= 8 && kWinTraceWords % 2 == 0, "filled four / two words at a time");
static_assert(kPoolPadWords >= (uint32_t)kWinSeqWords && kTracePadWords >= (uint32_t)kWinTraceWords, "a window is filled from word 0 on");
struct TbWin {
    uint32_t *q, *t;        // the lane's column of each window
    uint64_t *s;
    uint32_t qb, tb, sb;    // the first word each holds (of the sequence / of the task's trace); ~0: nothing yet
};

// 64 bases from base `pos` of the sequence that starts at `seq` (pos counts from seq's first word), through the lane's window
__device__ __forceinline__ Bases64 fetch64_win(const uint32_t *__restrict__ seq, uint32_t pos, uint32_t *__restrict__ win, uint32_t &base) {
    const uint32_t ws = pos >> 4;
    // (when one lane of the wavefront is out of its window every lane at this fetch moves its window: the lanes use theirs up at
    // different rates, and a row in which any of them waits for memory costs the wavefront that wait -- together they wait once
    // in a dozen rows instead of in nearly every one)
    if (__ballot(ws < base || ws + 4u >= base + (uint32_t)kWinSeqWords)) {   // the five words ws .. ws + 4 end the new window
        base = ws + 5u > (uint32_t)kWinSeqWords ? ws + 5u - (uint32_t)kWinSeqWords : 0u;
        const uint32_t *__restrict__ src = seq + base;
#pragma unroll
        for (int i = 0; i < kWinSeqWords; i += 4) {
            struct W4 { uint32_t x, y, z, w; } v;
            v.x = src[i], v.y = src[i + 1], v.z = src[i + 2], v.w = src[i + 3];
            win[(i + 0) * 64] = v.x, win[(i + 1) * 64] = v.y, win[(i + 2) * 64] = v.z, win[(i + 3) * 64] = v.w;
        }
    }
    const uint32_t *__restrict__ p = win + (ws - base) * 64u;
    const uint32_t v0 = p[0], v1 = p[64], v2 = p[128], v3 = p[192], v4 = p[256];
    const uint32_t s = (pos & 15u) * 2u;
    Bases64 r;
    r.w[0] = __builtin_amdgcn_alignbit(v1, v0, s);
    r.w[1] = __builtin_amdgcn_alignbit(v2, v1, s);
    r.w[2] = __builtin_amdgcn_alignbit(v3, v2, s);
    r.w[3] = __builtin_amdgcn_alignbit(v4, v3, s);
    return r;
}

// record word p of the task's trace through the lane's window
__device__ __forceinline__ uint64_t trace_win(const uint64_t *__restrict__ S, uint32_t p, uint64_t *__restrict__ win, uint32_t &base) {
    if (__ballot(p < base || p >= base + (uint32_t)kWinTraceWords)) {   // word p ends the new window
        base = p + 1u > (uint32_t)kWinTraceWords ? p + 1u - (uint32_t)kWinTraceWords : 0u;
        const uint64_t *__restrict__ src = S + base;
#pragma unroll
        for (int i = 0; i < kWinTraceWords; i += 2) {
            const uint64_t a = src[i], b = src[i + 1];
            win[(i + 0) * 64] = a, win[(i + 1) * 64] = b;
        }
    }
    return win[(p - base) * 64u];
}

// the match run behind (x, x - k), back to front (lib/align.c:502-507), 64 bases per compare; returns its length, x moves
// (WIN: q_off / t_off are the sequences' first bases within the words qp / tp point at, and the bases come through `win`)
template <bool WIN>
__device__ __forceinline__ int tb_match_run(const uint32_t *__restrict__ qp, const uint32_t *__restrict__ tp, uint64_t q_off, uint64_t t_off,
                                            int &x, int k, TbWin &win) {
    int total = 0;
    for (;;) {
        const int yy = x - k;
        const int avail = (x < yy ? x : yy) + 1;
        if (avail <= 0) break;
        const int n = avail < 64 ? avail : 64;
        Bases64 a, b;
        if (WIN) {
            a = fetch64_win(qp, (uint32_t)q_off + (uint32_t)(x - n + 1), win.q, win.qb);
            b = fetch64_win(tp, (uint32_t)t_off + (uint32_t)(yy - n + 1), win.t, win.tb);
        } else {
            a = fetch64_abs(qp, q_off + (uint64_t)(uint32_t)(x - n + 1));
            b = fetch64_abs(tp, t_off + (uint64_t)(uint32_t)(yy - n + 1));
        }
        int m = n;
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            const int nb = n - 16 * i;
            if (nb <= 0) continue;
            uint32_t diff = a.w[i] ^ b.w[i];
            if (nb < 16) diff &= (1u << (2 * nb)) - 1u;
            if (diff) {
                m = n - 1 - (16 * i + ((31 - __builtin_clz(diff)) >> 1));
                break;
            }
        }
        x -= m;
        total += m;
        if (m < n) break;
    }
    return total;
}

template <bool WIN>
__global__ __launch_bounds__(64) void tb_walk_kernel(const TbSeg *__restrict__ segs, TbSegOut *__restrict__ seg_outs,
                                                      const AlnTask *__restrict__ tasks, const AlnOut *__restrict__ outs,
                                                      const uint32_t *__restrict__ pool, const uint32_t *__restrict__ db_pool,
                                                      const uint64_t *__restrict__ trace, uint32_t *__restrict__ ops, int n_slots) {
    const int slot = (int)(blockIdx.x * 64 + threadIdx.x);
    if (slot >= n_slots) return;
    const TbSeg G = segs[slot];
    if (G.task < 0) return;
    const AlnTask T = tasks[G.task];
    const uint32_t *__restrict__ qp = (T.q_off >> 63) ? db_pool : pool;
    const uint32_t *__restrict__ tp = (T.t_off >> 63) ? db_pool : pool;
    uint64_t q_off = T.q_off & kOffMask, t_off = T.t_off & kOffMask;
    const uint64_t *__restrict__ S = trace + T.trace_off;
    uint32_t *__restrict__ W = ops + T.ops_off;

    __shared__ uint32_t l_q[WIN ? kWinSeqWords * 64 : 1], l_t[WIN ? kWinSeqWords * 64 : 1];
    __shared__ uint64_t l_s[WIN ? kWinTraceWords * 64 : 1];
    TbWin win;
    win.q = l_q + (WIN ? threadIdx.x : 0), win.t = l_t + (WIN ? threadIdx.x : 0), win.s = l_s + (WIN ? threadIdx.x : 0);
    win.qb = win.tb = win.sb = 0xffffffffu;
    if (WIN) {   // the windows count words from the sequence's first
        qp += q_off >> 4, tp += t_off >> 4;
        q_off &= 15u, t_off &= 15u;
    }
    auto rec = [&](uint32_t p) -> uint64_t { return WIN ? trace_win(S, p, win.s, win.sb) : S[p]; };

    int x = G.x, k = G.min_k + 2 * G.idx, d = G.d_top, idx = G.idx;
    uint32_t pos = G.pos;
    uint64_t hdr = pos ? rec(pos - 1) : 0ull;
    uint64_t c1 = pos >= 2 ? rec(pos - 2) : 0ull;   // the word before the header, loaded a row ahead: the header of the row below, or -- for
                                                    // the few rows of more than 56 cells -- this record's second word
    TbSegOut R;
    R.x_own = R.k_own = R.x_end = R.k_end = 0;
    R.lead = R.trail = 0;
    R.rows = 0;
    R.flags = 0;

    // one row: the move (lib/align.c:509-541) and the step into the record of the row below
    auto move = [&](bool &left) {
        if (x < k) left = true;  // lib/align.c:512: forced query-consuming move
        else if (x >= 0) {
            const bool second = idx >= kStreamBits && pos >= 2;
            const uint64_t w = second ? c1 : hdr;
            left = (w >> ((second ? idx - kStreamBits : idx) & 63)) & 1ull;
        } else left = false;
    };
    auto step_down = [&](bool left) {
        d--;
        if (left) { k--; x--; } else k++;
        const uint32_t len = 1u + (uint32_t)((hdr >> kStreamBits) & 1ull);
        pos = pos > len ? pos - len : 0u;
        hdr = pos ? (len == 1u ? c1 : rec(pos - 1)) : 0ull;   // (behind a two-word record the header is not in hand: one load the row waits for)
        idx += (int)(hdr >> (kStreamBits + 1)) - (left ? 1 : 0);
        c1 = pos >= 2 ? rec(pos - 2) : 0ull;
    };

    // the rows above the owned ones: the walk only (no columns, no gap count)
    while (d > G.d_own) {
        (void)tb_match_run<WIN>(qp, tp, q_off, t_off, x, k, win);
        if (x < 0 && x - k < 0) {
            R.flags = kTbBad;
            break;
        }
        bool left;
        move(left);
        step_down(left);
    }
    if (R.flags) {
        seg_outs[slot] = R;
        return;
    }

    R.x_own = x;
    R.k_own = k;
    uint32_t col;
    {
        const AlnOut O = outs[G.task];
        const int done = ((O.x_final - 1 - x) + (O.y_final - 1 - (x - k)) + (O.d_final - d)) >> 1;  // columns the walk has emitted above
        col = T.ops_cap - (uint32_t)(done < 0 ? 0 : done);
    }
    // Column words: a match column is code 0 and the words were zeroed by the forward kernel, so a match run only moves the cursor;
    // gap columns collect in `acc` until the cursor leaves their word.  The walker's first word may hold columns of the walker above
    // and its last one columns of the walker below: those two are OR-ed into place, every word between them is this walker's alone.
    const uint32_t head_w = (col & 15u) ? (col >> 4) : 0xffffffffu;
    uint32_t cur_w = col >> 4, acc = 0;
    int gap = 0;
    bool reset = false;
    for (;;) {
        const int m = tb_match_run<WIN>(qp, tp, q_off, t_off, x, k, win);
        if (m) {
            col -= (uint32_t)m;
            gap = 0;
            reset = true;
        }
        if (x < 0 && x - k < 0) {
            R.flags |= d == 0 ? kTbTerminal : kTbBad;  // (the start of the alignment above row 0: the column count is not the closed form's)
            break;
        }
        bool left;
        move(left);
        uint32_t code;
        if (left) { code = 1u; if (x < 0) gap = 260; }
        else { code = 2u; if (x - k < 0) gap = 260; }
        if (gap < 260) {
            col--;
            const uint32_t wi = col >> 4;
            if (wi != cur_w) {
                if (acc) {
                    if (cur_w == head_w) atomicOr(&W[cur_w], acc);
                    else W[cur_w] = acc;
                }
                acc = 0;
                cur_w = wi;
            }
            acc |= code << ((col & 15u) * 2u);
        }
        if (!reset) R.lead++;
        R.rows++;
        if (gap++ > 250) {  // lib/align.c:542-545 (with the gap columns of this walker alone; the stitch adds what came before)
            R.flags |= kTbAbort;
            break;
        }
        step_down(left);
        if (d < G.d_end) break;
    }
    if (acc) atomicOr(&W[cur_w], acc);
    R.x_end = x;
    R.k_end = k;
    R.trail = gap;
    if (reset) R.flags |= kTbReset;
    seg_outs[slot] = R;
}

// one lane per task: the walkers' boundaries, the gap count across them, the verdict.  A task it refuses keeps ST_FINISHED and is
// walked by the one-lane kernel, launched behind this one over the same tasks (it skips every task that has its verdict).
__global__ __launch_bounds__(64) void tb_stitch_kernel(const AlnTask *__restrict__ tasks, AlnOut *__restrict__ outs,
                                                        const TbSegOut *__restrict__ seg_outs, int n_tasks, int cshift) {
    const int gid = (int)(blockIdx.x * 64 + threadIdx.x);
    if (gid >= n_tasks) return;
    const AlnOut O = outs[gid];
    if (O.status != ST_FINISHED) return;
    const TbSegOut *__restrict__ R = seg_outs + tasks[gid].seg_off;
    const int n = O.d_final > 0 ? (O.d_final - 1) >> cshift : 0;
    bool bad = false, aborted = false, ended = false;
    int gap = 0, px = 0, pk = 0;
    for (int w = 0; w <= n && !bad && !aborted && !ended; w++) {
        const TbSegOut r = R[w];
        if (r.flags & kTbBad) bad = true;
        else if (w && (r.x_own != px || r.k_own != pk)) bad = true;
        else {
            // gap columns in a row: a walker's first run continues the run the walker above ended with
            if (r.lead && gap + r.lead - 1 > 250) aborted = true;
            else if (r.flags & kTbAbort) aborted = true;
            else {
                gap = (r.flags & kTbReset) ? r.trail : gap + (int)r.rows;
                px = r.x_end;
                pk = r.k_end;
                if (r.flags & kTbTerminal) {
                    ended = true;
                    if (w != n) bad = true;   // the alignment's start above the last walker's rows: not this kernel's case
                } else if (w == n) bad = true;  // the last walker left row 0 without reaching the start: neither
            }
        }
    }
    outs[gid].fin_idx = O.fin_idx | kTbSeen | (bad ? kTbRefused : 0);  // (for the host's counts)
    if (bad) return;
    outs[gid].n_cols = aborted ? 2 : (O.x_final + O.y_final + O.d_final) >> 1;
    outs[gid].status = aborted ? ST_GAP_ABORT : ST_ALIGNED;
}


}  // namespace

void launch_ond_forward(const AlnTask *tasks, AlnOut *outs, const uint32_t *pool, const uint32_t *db_pool, uint64_t *trace,
                        int n_tasks, void *stream, const int32_t *order) {
    if (n_tasks <= 0) return;
    hipLaunchKernelGGL(ond_forward_kernel<false>, dim3((unsigned)n_tasks), dim3(64), 0, (hipStream_t)stream, tasks, outs, pool, db_pool, trace,
                       order, (uint32_t *)nullptr, (uint2 *)nullptr, (uint32_t *)nullptr, 0);
}

void launch_ond_forward_ckpt(const AlnTask *tasks, AlnOut *outs, const uint32_t *pool, const uint32_t *db_pool, uint64_t *trace,
                             uint32_t *ops, const TbArgs &tb, int n_tasks, void *stream, const int32_t *order) {
    if (n_tasks <= 0) return;
    // (NDGPU_K7_LDS_KB: dynamic LDS a forward wavefront asks for and never touches -- an occupancy knob for A/B runs: fewer resident K7
    // wavefronts leave wavefront slots and issue cycles to the kernels of the other contexts)
    static const size_t k7_lds = getenv("NDGPU_K7_LDS_KB") ? (size_t)atoi(getenv("NDGPU_K7_LDS_KB")) << 10 : 0;
    hipLaunchKernelGGL(ond_forward_kernel<true>, dim3((unsigned)n_tasks), dim3(64), k7_lds, (hipStream_t)stream, tasks, outs, pool, db_pool, trace,
                       order, tb.ck_cells, (uint2 *)tb.ck_hdr, ops, tb.cshift);
}

void launch_ond_traceback_seg(const AlnTask *tasks, AlnOut *outs, const uint32_t *pool, const uint32_t *db_pool, const uint64_t *trace,
                              uint32_t *ops, const TbArgs &tb, int n_tasks, void *stream) {
    if (n_tasks <= 0) return;
    const dim3 per_task((unsigned)((n_tasks + 63) / 64)), per_slot((unsigned)((tb.n_slots + 63) / 64));
    hipLaunchKernelGGL(tb_chase_kernel, per_task, dim3(64), 0, (hipStream_t)stream, tasks, outs, tb.ck_cells, (const uint2 *)tb.ck_hdr, tb.segs,
                       n_tasks, tb.cshift, tb.warm);
    // (NDGPU_TB_LDS_KB: dynamic LDS a walker wavefront asks for and never touches -- an occupancy knob: every lane walks a trace and two
    // sequences of its own, so the lines a compute unit's resident walkers hold open outgrow the caches with their number)
    static const size_t walk_lds = getenv("NDGPU_TB_LDS_KB") ? (size_t)atoi(getenv("NDGPU_TB_LDS_KB")) << 10 : 0;
    // (NDGPU_TB_WIN=0: the walkers read memory directly, for A/B runs)
    static const bool walk_win = !getenv("NDGPU_TB_WIN") || atoi(getenv("NDGPU_TB_WIN")) != 0;
    if (walk_win)
        hipLaunchKernelGGL(tb_walk_kernel<true>, per_slot, dim3(64), walk_lds, (hipStream_t)stream, tb.segs, tb.seg_outs, tasks, outs, pool, db_pool,
                           trace, ops, tb.n_slots);
    else
        hipLaunchKernelGGL(tb_walk_kernel<false>, per_slot, dim3(64), walk_lds, (hipStream_t)stream, tb.segs, tb.seg_outs, tasks, outs, pool, db_pool,
                           trace, ops, tb.n_slots);
    hipLaunchKernelGGL(tb_stitch_kernel, per_task, dim3(64), 0, (hipStream_t)stream, tasks, outs, tb.seg_outs, n_tasks, tb.cshift);
    // what the stitch refused (still ST_FINISHED) in one piece
    hipLaunchKernelGGL(ond_traceback_kernel<true>, per_task, dim3(64), 0, (hipStream_t)stream, tasks, outs, pool, db_pool, trace,
                       (const int32_t *)nullptr, ops, (const int32_t *)nullptr, n_tasks, (const AlnTask *)nullptr);
}

void launch_ond_forward_wide(const AlnTask *tasks, AlnOut *outs, const uint32_t *pool, const uint32_t *db_pool,
                             uint64_t *trace,
                             int32_t *trace_mink, int32_t *vscratch, const int32_t *task_ids, int n_ids, void *stream,
                             const AlnTask *wtasks, uint32_t max_ring) {
    if (n_ids <= 0) return;
    // (max_ring: the largest V ring among the listed tasks, in ints; 0: unknown -- the rings stay in `vscratch`.  NDGPU_WIDE_LDS=0: A/B)
    static const bool lds_ok = !getenv("NDGPU_WIDE_LDS") || atoi(getenv("NDGPU_WIDE_LDS")) != 0;
    if (lds_ok && max_ring && (size_t)max_ring * sizeof(int32_t) <= kWideLdsBytes)
        hipLaunchKernelGGL(ond_forward_wide_kernel<true>, dim3((unsigned)n_ids), dim3(64), (size_t)max_ring * sizeof(int32_t), (hipStream_t)stream,
                           tasks, outs, pool, db_pool, trace, trace_mink, vscratch, task_ids, wtasks);
    else
        hipLaunchKernelGGL(ond_forward_wide_kernel<false>, dim3((unsigned)n_ids), dim3(64), 0, (hipStream_t)stream, tasks, outs,
                           pool, db_pool, trace, trace_mink, vscratch, task_ids, wtasks);
}

// task_ids == nullptr: every task of the table, traces in the register path's stream format -- in the order `order` lists them
// (longest first: the 64 walks of a wavefront then are of one length class and end together, and the long ones start first), or
// in table order; otherwise the listed (wide-band) tasks, traces in the wide kernel's row format
void launch_ond_traceback(const AlnTask *tasks, AlnOut *outs, const uint32_t *pool, const uint32_t *db_pool,
                          const uint64_t *trace,
                          const int32_t *trace_mink, uint32_t *ops, const int32_t *task_ids, int n_tasks, void *stream,
                          const int32_t *order, const AlnTask *wtasks) {
    if (n_tasks <= 0) return;
    if (task_ids)
        hipLaunchKernelGGL(ond_traceback_kernel<false>, dim3((unsigned)((n_tasks + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                           tasks, outs, pool, db_pool, trace, trace_mink, ops, task_ids, n_tasks, wtasks);
    else
        hipLaunchKernelGGL(ond_traceback_kernel<true>, dim3((unsigned)((n_tasks + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                           tasks, outs, pool, db_pool, trace, (const int32_t *)nullptr, ops, order, n_tasks, (const AlnTask *)nullptr);
}

}  // namespace ndgpu
