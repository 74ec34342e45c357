// Hand-written CDNA4 (gfx950) kernels for the banded greedy O(ND) aligner that
// NextDenovo's consensus is built on (reference: core() via align()/align_hq(),
// lib/align.c:428-578).  Integer DP, no MFMA.
//
// K7  ond_forward   : one 64-lane wavefront per alignment.  Lane l of pass p owns
//                     diagonal k = min_k + 2*(64p + l); all diagonals of one edit
//                     step d are independent, so a step is: 2 LDS reads of the
//                     furthest-reaching ring V[], a snake (XOR + ctz over 16-base
//                     2-bit words fetched from the HBM-resident pool), 1 LDS write,
//                     two wave ballots (move bits -> trace row, finish test), one
//                     wave max (best anti-diagonal) and two ballots for the band
//                     re-centring.  Traceback memory is 1 bit per evaluated cell
//                     plus min_k per step, instead of the reference's one byte per
//                     cell of an O(max_d^2) triangle.
// K7w ond_forward<W>: same code with V[] in a global scratch ring and a wider
//                     trace row, for the rare alignments whose live band exceeds
//                     the 253-diagonal LDS fast path (exactness of the band-cap /
//                     edit-budget failure semantics).
// K8a ond_traceback : one lane per alignment walks d -> 0 reading the move bits,
//                     re-deriving match runs with clz over 16-base words, and emits
//                     2-bit column kinds back to front.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "nd_device.h"

namespace ndgpu {

namespace {

typedef uint64_t __attribute__((aligned(4))) u64_a4;  // dwordx2 loads need 4-byte alignment only

__device__ __forceinline__ uint32_t fetch16(const uint32_t *__restrict__ pool, uint64_t off) {
    const uint32_t s = (uint32_t)(off & 15u) * 2u;
    const uint64_t v = *(const u64_a4 *)(pool + (off >> 4));
    return (uint32_t)(v >> s);
}

// maximum over the 64 lanes (all of them active) with data-parallel-primitive moves instead of six LDS permutes: the
// forward kernel is bound by instruction issue, and this reduction sits on every edit step
__device__ __forceinline__ uint32_t fetch16_rel(const uint32_t *__restrict__ seq, uint32_t pos) {  // pos: base index from seq's first word
    const uint64_t v = *(const u64_a4 *)(seq + (pos >> 4));
    return (uint32_t)(v >> ((pos & 15u) * 2u));
}

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

template <bool WIDE>
__global__ __launch_bounds__(64) void ond_forward_kernel(const AlnTask *__restrict__ tasks, AlnOut *__restrict__ outs,
                                                          const uint32_t *__restrict__ pool,
                                                          const uint32_t *__restrict__ db_pool,
                                                          uint64_t *__restrict__ trace,
                                                          int32_t *__restrict__ trace_mink,
                                                          int32_t *__restrict__ vscratch,
                                                          const int32_t *__restrict__ ids) {
    __shared__ int32_t v_lds[WIDE ? 1 : kFastVSize];
    const int tid = WIDE ? ids[blockIdx.x] : (int)blockIdx.x;
    const AlnTask T = tasks[tid];
    const int lane = (int)threadIdx.x;
    int32_t *V = WIDE ? (vscratch + T.v_off) : v_lds;
    const uint32_t vmask = WIDE ? T.v_mask : (uint32_t)(kFastVSize - 1);

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
    const uint32_t row_words = WIDE ? T.row_words : (uint32_t)kFastRowWords;

    for (int d = 0; d < T.max_d && max_k - min_k <= T.band; d++) {
        const int band = max_k - min_k;
        if (!WIDE && band > kFastMaxBand) {
            status = ST_NEED_WIDE;
            break;
        }
        const int ncell = band >= 0 ? (band >> 1) + 1 : 0;
        const int npass = (ncell + 63) >> 6;
        d_steps++;
        cells += ncell;
        max_band = band > max_band ? band : max_band;
        if (lane == 0) trace_mink[mk0 + d] = min_k;

        int row_best = -1;
        bool done = false;
        int x_keep = 0;
        for (int ps = 0; ps < npass; ps++) {
            const int k = min_k + 2 * (ps * 64 + lane);
            const bool act = k <= max_k;
            int x = 0;
            bool left = false;
            if (act) {
                const int vm = V[(uint32_t)(k - 1) & vmask];
                const int vp = V[(uint32_t)(k + 1) & vmask];
                // lib/align.c:443
                const bool down = (k == min_k) || (k != max_k && vm < vp);
                x = down ? vp : vm + 1;
                left = !down;
                int y = x - k;
                // snake: 16 bases per XOR, first mismatch = ctz/2 (lib/align.c:452-455)
                for (;;) {
                    int rem = q_len - x;
                    const int rt = t_len - y;
                    rem = rt < rem ? rt : rem;
                    if (rem <= 0) break;
                    const uint32_t a = fetch16_rel(qp, q_sh + (uint32_t)x);
                    const uint32_t b = fetch16_rel(tp, t_sh + (uint32_t)y);
                    const uint32_t diff = a ^ b;
                    int m = diff ? (__builtin_ctz(diff) >> 1) : 16;
                    m = m < rem ? m : rem;
                    x += m;
                    y += m;
                    if (m < 16) break;
                }
            }
            const unsigned long long lb = __ballot(act && left);
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
                break;
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
        outs[tid] = o;
    }
}

__global__ __launch_bounds__(64) void ond_traceback_kernel(const AlnTask *__restrict__ tasks, AlnOut *__restrict__ outs,
                                                            const uint32_t *__restrict__ pool,
                                                            const uint32_t *__restrict__ db_pool,
                                                            const uint64_t *__restrict__ trace,
                                                            const int32_t *__restrict__ trace_mink,
                                                            uint32_t *__restrict__ ops,
                                                            const int32_t *__restrict__ ids, int n_tasks) {
    const int slot = (int)(blockIdx.x * 64 + threadIdx.x);
    if (slot >= n_tasks) return;
    const int gid = ids ? ids[slot] : slot;
    if (outs[gid].status != ST_FINISHED) return;
    const AlnTask T = tasks[gid];
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

    for (;;) {
        // match run, back to front (lib/align.c:502-507), 16 bases per compare
        for (;;) {
            const int yy = x - k;
            const int avail = (x < yy ? x : yy) + 1;
            if (avail <= 0) break;
            const int n = avail < 16 ? avail : 16;
            const uint32_t a = fetch16(qp, q_off + (uint64_t)(uint32_t)(x - n + 1));
            const uint32_t b = fetch16(tp, t_off + (uint64_t)(uint32_t)(yy - n + 1));
            uint32_t diff = a ^ b;
            if (n < 16) diff &= (1u << (2 * n)) - 1u;
            const int m = diff ? n - 1 - ((31 - __builtin_clz(diff)) >> 1) : n;
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
            const int idx = (k - trace_mink[T.mink_off + (uint64_t)(uint32_t)d]) >> 1;
            left = (trace[T.trace_off + (uint64_t)(uint32_t)d * T.row_words + (uint32_t)(idx >> 6)] >> (idx & 63)) & 1ull;
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
    }
    if ((col & 15u) != 0) W[col >> 4] = acc;
    outs[gid].n_cols = aborted ? 2 : (int32_t)(T.ops_cap - col);
    outs[gid].status = aborted ? ST_GAP_ABORT : ST_ALIGNED;
}


// K7, two alignments per wavefront (NDGPU_K7=pair; the default is the kernel above until the two have been compared inside the
// full pipeline on the device).  The live band of a raw-read alignment is mostly narrower than 64 diagonals, so half of a
// wavefront's lanes idle in the kernel above and every alignment pays for a whole wavefront's issue slots.  Here lanes 0-31 own
// task 2b and lanes 32-63 task 2b + 1 of workgroup b: a pass covers 32 diagonals, each half has its own furthest-reaching ring
// in LDS, reads its own 32 bits of the wave ballots and reduces its own maximum; the loops run while either half is busy.  The
// trace layout is the one the traceback kernels read (bit c of a row = diagonal min_k + 2c), so nothing downstream changes.
__device__ __forceinline__ int half_max_i32(int v) {  // maximum over the 32 lanes of the caller's half, in every lane of it
    auto step = [](int x, auto ctrl) {
        const int y = __builtin_amdgcn_update_dpp(INT_MIN, x, decltype(ctrl)::value, 0xf, 0xf, false);
        return y > x ? y : x;
    };
    v = step(v, std::integral_constant<int, 0xb1>{});   // quad_perm [1,0,3,2]
    v = step(v, std::integral_constant<int, 0x4e>{});   // quad_perm [2,3,0,1]
    v = step(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{});  // row_mirror: every lane holds its row's maximum
    const int o = __shfl_xor(v, 16, 64);                // the other row of the half
    return o > v ? o : v;
}

__global__ __launch_bounds__(64) void ond_forward_pair_kernel(const AlnTask *__restrict__ tasks, AlnOut *__restrict__ outs,
                                                               const uint32_t *__restrict__ pool,
                                                               const uint32_t *__restrict__ db_pool,
                                                               uint64_t *__restrict__ trace, int32_t *__restrict__ trace_mink,
                                                               const int32_t *__restrict__ order, int n_tasks) {
    __shared__ int32_t v_lds[2][kFastVSize];
    const int lane = (int)threadIdx.x, h = lane >> 5, l = lane & 31;
    const int slot = 2 * (int)blockIdx.x + h;
    const bool have = slot < n_tasks;
    // `order` lists the tasks by length, so that the two halves of a wavefront finish together; nullptr = as they come
    const int tid = order ? order[have ? slot : 2 * (int)blockIdx.x] : (have ? slot : 2 * (int)blockIdx.x);
    const AlnTask T = tasks[tid];
    int32_t *V = v_lds[h];
    const uint32_t vmask = (uint32_t)(kFastVSize - 1);
    for (uint32_t i = (uint32_t)l; i <= vmask; i += 32) V[i] = 0;
    __syncthreads();

    int min_k = 0, max_k = 0, best_m = -1;
    int status = ST_NONE, fin_k = 0, fin_x = 0, fin_d = -1;
    int d_steps = 0, max_band = 0, d = 0;
    long long cells = 0;
    bool alive = have;
    const int q_len = T.q_len, t_len = T.t_len;
    const uint64_t q_off = T.q_off & kOffMask, t_off = T.t_off & kOffMask;
    const uint32_t *__restrict__ qp = ((T.q_off >> 63) ? db_pool : pool) + (q_off >> 4);
    const uint32_t *__restrict__ tp = ((T.t_off >> 63) ? db_pool : pool) + (t_off >> 4);
    const uint32_t q_sh = (uint32_t)(q_off & 15u), t_sh = (uint32_t)(t_off & 15u);
    const uint64_t row0 = T.trace_off, mk0 = T.mink_off;
    const int hs = 32 * h;  // first lane of this half

    for (;;) {
        // one edit step of either half that still has one to do (the conditions of the loop head above, lib/align.c:437)
        bool go = alive && d < T.max_d && max_k - min_k <= T.band;
        const int band = max_k - min_k;
        if (go && band > kFastMaxBand) {
            status = ST_NEED_WIDE;
            go = false;
        }
        if (!go) alive = false;
        if (__ballot(go) == 0ull) break;
        const int ncell = band >= 0 ? (band >> 1) + 1 : 0;
        const int npass = go ? (ncell + 31) >> 5 : 0;
        const int np0 = __shfl(npass, 0, 64), np1 = __shfl(npass, 32, 64);
        const int np_max = np0 > np1 ? np0 : np1;
        if (go) {
            d_steps++;
            cells += ncell;
            max_band = band > max_band ? band : max_band;
            if (l == 0) trace_mink[mk0 + d] = min_k;
        }
        int row_best = -1, x_keep = 0;
        bool done = false;
        uint64_t row_acc = 0;
        for (int ps = 0; ps < np_max; ps++) {
            const bool pact = go && !done && ps < npass;
            const int k = min_k + 2 * (ps * 32 + l);
            const bool act = pact && k <= max_k;
            int x = 0;
            bool left = false;
            if (act) {
                const int vm = V[(uint32_t)(k - 1) & vmask];
                const int vp = V[(uint32_t)(k + 1) & vmask];
                const bool down = (k == min_k) || (k != max_k && vm < vp);  // lib/align.c:443
                x = down ? vp : vm + 1;
                left = !down;
                int y = x - k;
                for (;;) {  // snake: 16 bases per XOR (lib/align.c:452-455)
                    int rem = q_len - x;
                    const int rt = t_len - y;
                    rem = rt < rem ? rt : rem;
                    if (rem <= 0) break;
                    const uint32_t a = fetch16_rel(qp, q_sh + (uint32_t)x);
                    const uint32_t b = fetch16_rel(tp, t_sh + (uint32_t)y);
                    const uint32_t diff = a ^ b;
                    int m = diff ? (__builtin_ctz(diff) >> 1) : 16;
                    m = m < rem ? m : rem;
                    x += m;
                    y += m;
                    if (m < 16) break;
                }
            }
            const uint32_t lb = (uint32_t)(__ballot(act && left) >> hs);
            const int y = x - k;
            const uint32_t fb = (uint32_t)(__ballot(act && x >= q_len && y >= t_len) >> hs);
            if (pact) {
                row_acc |= (uint64_t)lb << (32 * (ps & 1));
                if (((ps & 1) || ps == npass - 1) && l == 0) trace[row0 + (uint64_t)d * kFastRowWords + (uint32_t)(ps >> 1)] = row_acc;
                if (ps & 1) row_acc = 0;
            }
            if (act) {
                V[(uint32_t)k & vmask] = x;
                const int m = x + y;
                row_best = m > row_best ? m : row_best;
            }
            if (pact) x_keep = x;  // (a pass run only for the other half must not disturb this half's register copy)
            const int fl = fb ? __ffs((int)fb) - 1 : 0;
            const int fx = __shfl(x, hs + fl, 64);
            if (pact && fb) {  // several diagonals may finish in one step: the smallest k wins (lib/align.c:467-470)
                // (the row's word may still be pending: the passes that would have completed it are not run)
                if (!(ps & 1) && ps != npass - 1 && l == 0) trace[row0 + (uint64_t)d * kFastRowWords + (uint32_t)(ps >> 1)] = row_acc;
                fin_k = min_k + 2 * (ps * 32 + fl);
                fin_x = fx;
                fin_d = d;
                status = ST_FINISHED;
                done = true;
            }
        }
        if (go && done) alive = false, go = false;
        const int rb = half_max_i32(row_best);
        if (go) best_m = rb > best_m ? rb : best_m;
        __syncthreads();  // V[] of this step visible to every lane

        // band re-centring (lib/align.c:473-489)
        int new_min = max_k, new_max = min_k;
        const int thr = best_m - 150;
        bool found = false;
        for (int ps = 0; ps < np_max; ps++) {
            const int k = min_k + 2 * (ps * 32 + l);
            const bool in = go && !found && ps < npass;
            const int xv = npass == 1 ? x_keep : (in ? V[(uint32_t)k & vmask] : 0);
            const bool q = in && k < max_k && (2 * xv - k >= thr);
            const uint32_t qb = (uint32_t)(__ballot(q) >> hs);
            if (in && qb) {
                new_min = min_k + 2 * (ps * 32 + (__ffs((int)qb) - 1));
                found = true;
            }
        }
        found = false;
        for (int ps = np_max - 1; ps >= 0; ps--) {
            const int k = min_k + 2 * (ps * 32 + l);
            const bool in = go && !found && ps < npass;
            const int xv = npass == 1 ? x_keep : (in ? V[(uint32_t)k & vmask] : 0);
            const bool q = in && k <= max_k && k > min_k && (2 * xv - k >= thr);
            const uint32_t qb = (uint32_t)(__ballot(q) >> hs);
            if (in && qb) {
                new_max = min_k + 2 * (ps * 32 + (31 - __clz((int)qb)));
                found = true;
            }
        }
        if (go) {
            max_k = new_max + 1;
            min_k = new_min - 1;
            d++;
        }
        __syncthreads();  // the next step's first reads of V[] come after this step's last ones
    }

    if (have && l == 0) {
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
        outs[tid] = o;
    }
}

// K8a, wavefront-per-alignment form (NDGPU_K8A=wave; the default is the lane-per-alignment kernel above until the two have been
// compared inside the full pipeline on the device).  The walk is one dependent chain per alignment; in the kernel above each link
// of it is a round trip to HBM / L2 and a wavefront lasts as long as the longest of its 64 chains.  Here a wavefront owns ONE
// alignment: it stages the next 128 trace rows (move bits + min_k) and the next 1024 bases of both sequences in LDS with
// coalesced loads, and walks them there -- every lane holds the same walk state, so the loop is uniform and a link costs an LDS
// broadcast read instead of a memory round trip; lane 0 writes the packed columns.  Same output, bit for bit.
constexpr int kTbRows = 128;
constexpr int kTbSeqWords = 64;  // 16 bases each; + 2 words of slack for the unaligned 64-bit fetch

__device__ __forceinline__ uint32_t fetch16_win(const uint32_t *win, uint64_t w0, uint64_t off) {  // off: absolute base offset
    const uint32_t i = (uint32_t)((off >> 4) - w0), sh = (uint32_t)(off & 15u) * 2u;
    const uint64_t v = (uint64_t)win[i] | ((uint64_t)win[i + 1] << 32);
    return (uint32_t)(v >> sh);
}

__global__ __launch_bounds__(64) void ond_traceback_wave_kernel(const AlnTask *__restrict__ tasks, AlnOut *__restrict__ outs,
                                                                 const uint32_t *__restrict__ pool,
                                                                 const uint32_t *__restrict__ db_pool,
                                                                 const uint64_t *__restrict__ trace,
                                                                 const int32_t *__restrict__ trace_mink,
                                                                 uint32_t *__restrict__ ops, int n_tasks) {
    __shared__ uint64_t s_tr[kTbRows * kFastRowWords];
    __shared__ int32_t s_mk[kTbRows];
    __shared__ uint32_t s_q[kTbSeqWords + 2], s_t[kTbSeqWords + 2];
    const int gid = (int)blockIdx.x;
    if (gid >= n_tasks) return;
    if (outs[gid].status != ST_FINISHED) return;
    const int lane = (int)threadIdx.x;
    const AlnTask T = tasks[gid];
    const uint32_t *__restrict__ qp = (T.q_off >> 63) ? db_pool : pool;
    const uint32_t *__restrict__ tp = (T.t_off >> 63) ? db_pool : pool;
    const uint64_t q_off = T.q_off & kOffMask, t_off = T.t_off & kOffMask;
    int x = outs[gid].x_final - 1, k = outs[gid].k_final, d = outs[gid].d_final;
    int gap = 0;
    uint32_t col = T.ops_cap, acc = 0;
    uint32_t *W = ops + T.ops_off;
    bool aborted = false;
    int d_lo = 0, d_hi = -1;            // trace rows staged: [d_lo, d_hi]
    uint64_t qw0 = 0, tw0 = 0;          // first staged word of either sequence
    bool q_ok = false, t_ok = false;

    for (;;) {
        for (;;) {  // match run, back to front (lib/align.c:502-507), 16 bases per compare
            const int yy = x - k;
            const int avail = (x < yy ? x : yy) + 1;
            if (avail <= 0) break;
            const int n = avail < 16 ? avail : 16;
            const uint64_t qa = q_off + (uint64_t)(uint32_t)(x - n + 1), ta = t_off + (uint64_t)(uint32_t)(yy - n + 1);
            if (!q_ok || (qa >> 4) < qw0) {  // the window ends two words above the current position and reaches 1024 bases down
                __syncthreads();
                const uint64_t we = ((q_off + (uint64_t)(uint32_t)x) >> 4) + 2;
                qw0 = we > (uint64_t)(kTbSeqWords + 2) ? we - (uint64_t)(kTbSeqWords + 2) : 0;
                for (int i = lane; i < kTbSeqWords + 2; i += 64) s_q[i] = qp[qw0 + (uint64_t)i];
                q_ok = true;
                __syncthreads();
            }
            if (!t_ok || (ta >> 4) < tw0) {
                __syncthreads();
                const uint64_t we = ((t_off + (uint64_t)(uint32_t)yy) >> 4) + 2;
                tw0 = we > (uint64_t)(kTbSeqWords + 2) ? we - (uint64_t)(kTbSeqWords + 2) : 0;
                for (int i = lane; i < kTbSeqWords + 2; i += 64) s_t[i] = tp[tw0 + (uint64_t)i];
                t_ok = true;
                __syncthreads();
            }
            const uint32_t a = fetch16_win(s_q, qw0, qa), b = fetch16_win(s_t, tw0, ta);
            uint32_t diff = a ^ b;
            if (n < 16) diff &= (1u << (2 * n)) - 1u;
            const int m = diff ? n - 1 - ((31 - __builtin_clz(diff)) >> 1) : n;
            if (m) {
                int left_to_emit = m;  // match columns are code 0: only the cursor moves
                while (left_to_emit > 0) {
                    const uint32_t room = ((col - 1u) & 15u) + 1u;
                    const uint32_t take = (uint32_t)left_to_emit < room ? (uint32_t)left_to_emit : room;
                    col -= take;
                    left_to_emit -= (int)take;
                    if ((col & 15u) == 0) {
                        if (lane == 0) W[col >> 4] = acc;
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
            if (d < d_lo || d > d_hi) {
                __syncthreads();
                d_hi = d, d_lo = d - kTbRows + 1 > 0 ? d - kTbRows + 1 : 0;
                const int nr = d_hi - d_lo + 1;
                for (int i = lane; i < nr * kFastRowWords; i += 64)
                    s_tr[i] = trace[T.trace_off + (uint64_t)(uint32_t)d_lo * kFastRowWords + (uint64_t)i];
                for (int i = lane; i < nr; i += 64) s_mk[i] = trace_mink[T.mink_off + (uint64_t)(uint32_t)(d_lo + i)];
                __syncthreads();
            }
            const int idx = (k - s_mk[d - d_lo]) >> 1;
            left = (s_tr[(d - d_lo) * kFastRowWords + (idx >> 6)] >> (idx & 63)) & 1ull;
        } else left = false;
        uint32_t code;
        int nk, nx;
        if (left) { nk = k - 1; nx = x - 1; code = 1u; if (x < 0) gap = 260; }
        else { nk = k + 1; nx = x; code = 2u; if (x - k < 0) gap = 260; }
        if (gap < 260) {
            col--;
            acc |= code << ((col & 15u) * 2u);
            if ((col & 15u) == 0) {
                if (lane == 0) W[col >> 4] = acc;
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
    }
    if (lane == 0) {
        if ((col & 15u) != 0) W[col >> 4] = acc;
        outs[gid].n_cols = aborted ? 2 : (int32_t)(T.ops_cap - col);
        outs[gid].status = aborted ? ST_GAP_ABORT : ST_ALIGNED;
    }
}

}  // namespace

bool ond_forward_pairs() {
    static const bool pair_form = getenv("NDGPU_K7") && !strcmp(getenv("NDGPU_K7"), "pair");
    return pair_form;
}

void launch_ond_forward(const AlnTask *tasks, AlnOut *outs, const uint32_t *pool, const uint32_t *db_pool,
                        uint64_t *trace, int32_t *trace_mink,
                        int n_tasks, void *stream, const int32_t *order) {
    if (n_tasks <= 0) return;
    if (ond_forward_pairs()) {
        hipLaunchKernelGGL(ond_forward_pair_kernel, dim3((unsigned)((n_tasks + 1) / 2)), dim3(64), 0, (hipStream_t)stream, tasks, outs,
                           pool, db_pool, trace, trace_mink, order, n_tasks);
        return;
    }
    hipLaunchKernelGGL(ond_forward_kernel<false>, dim3((unsigned)n_tasks), dim3(64), 0, (hipStream_t)stream, tasks, outs,
                       pool, db_pool, trace, trace_mink, (int32_t *)nullptr, (const int32_t *)nullptr);
}

void launch_ond_forward_wide(const AlnTask *tasks, AlnOut *outs, const uint32_t *pool, const uint32_t *db_pool,
                             uint64_t *trace,
                             int32_t *trace_mink, int32_t *vscratch, const int32_t *task_ids, int n_ids, void *stream) {
    if (n_ids <= 0) return;
    hipLaunchKernelGGL(ond_forward_kernel<true>, dim3((unsigned)n_ids), dim3(64), 0, (hipStream_t)stream, tasks, outs,
                       pool, db_pool, trace, trace_mink, vscratch, task_ids);
}

void launch_ond_traceback(const AlnTask *tasks, AlnOut *outs, const uint32_t *pool, const uint32_t *db_pool,
                          const uint64_t *trace,
                          const int32_t *trace_mink, uint32_t *ops, const int32_t *task_ids, int n_tasks, void *stream) {
    if (n_tasks <= 0) return;
    static const bool wave_form = getenv("NDGPU_K8A") && !strcmp(getenv("NDGPU_K8A"), "wave");
    if (wave_form && !task_ids) {  // (the rare wide-band tasks, addressed through task_ids, have wider trace rows: lane kernel)
        hipLaunchKernelGGL(ond_traceback_wave_kernel, dim3((unsigned)n_tasks), dim3(64), 0, (hipStream_t)stream, tasks, outs, pool,
                           db_pool, trace, trace_mink, ops, n_tasks);
        return;
    }
    hipLaunchKernelGGL(ond_traceback_kernel, dim3((unsigned)((n_tasks + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                       tasks, outs, pool, db_pool, trace, trace_mink, ops, task_ids, n_tasks);
}

}  // namespace ndgpu
