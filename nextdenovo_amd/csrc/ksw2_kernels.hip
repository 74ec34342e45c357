// ksw2_kernels.hip -- minimap2's two-piece affine-gap extension kernel on the MI355X:
//     ksw_extd2_sse   minimap2/ksw2_extd2_sse.c:26-399   (caller: mm_align_pair, minimap2/align.c:331; the -c / -a path only)
// with ksw_apply_zdrop / ksw_backtrack / ksw_push_cigar of minimap2/ksw2.h:101-184.  Exported with the reference's signature and
// as a batched entry (ndgpu_ksw_extd2_batch): the calls come in batches of independent problems (the gaps between the anchors
// of every chain, the two end extensions), one wavefront per problem.
//
// The DP is Suzuki & Kasahara's difference recurrence over anti-diagonals r = i + j: the 8-bit differences u, v (of H), x, y
// (first gap piece), x2, y2 (second piece) are indexed by the target position and live in LDS (targets up to 4096 bases; longer
// ones use an HBM slice); the lanes take the cells of a diagonal 64 at a time, highest positions first, so that a cell's left
// neighbour still holds the previous diagonal when it is read.  A diagonal costs a handful of LDS round trips and two wave
// reductions (the exact maximum), whatever its length up to 64 cells.  Bit-exactness with the SSE code needs more than the
// recurrence (oracle/ksw2_oracle.c lists it): the reference computes whole 16-byte blocks around the true cell range and later
// diagonals read those extra cells; per-cell scores are refreshed in runs of 16 from the range's start; 8-bit arithmetic wraps;
// the maximum is searched four positions at a time, which decides between equal maxima; `qe` keeps its pre-swap value for the
// first cell.  The backtrack matrix (one byte per computed cell) is in HBM; the walk over it is one lane's work.
// There is no CPU path: without a HIP device the calls fail loudly.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <stdexcept>
#include <vector>

#include "../../include/ndgpu_overlap.h"
#include "ovl_pool.h"

namespace {

constexpr int kNegInf = -0x40000000;  // KSW_NEG_INF
constexpr int kLdsTarget = 4096;      // longest (16-rounded) target whose difference arrays are kept in LDS (11 bytes a position)
constexpr int kLdsSmall = 1024;       // the short problems (gap filling between anchors) run in a tier of their own: 11 KB of LDS a
                                      // wavefront instead of 44 KB, four times the wavefronts per compute unit
enum { F_SCORE_ONLY = 0x01, F_RIGHT = 0x02, F_GENERIC_SC = 0x04, F_APPROX_MAX = 0x08, F_APPROX_DROP = 0x10, F_EXTZ_ONLY = 0x40,
       F_REV_CIGAR = 0x80 };

struct KswJobDev {
    uint64_t q_off, t_off;       // codes in the byte pool
    uint64_t diff_off;           // 7 x tl16 bytes of difference arrays (targets beyond kLdsTarget only)
    uint64_t h_off;              // tl16 int32 (same)
    uint64_t p_off;              // (qlen + tlen - 1) x n_col bytes
    uint64_t off_off;            // 2 x (qlen + tlen - 1) int32
    uint64_t cigar_off;          // qlen + tlen + 2 uint32
    int32_t qlen, tlen, w, zdrop, end_bonus, flag;
    int8_t m, q, e, q2, e2, mat[25 + 2];
};

struct Ez {
    int32_t max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, n_cigar, reach_end;
};

__device__ __forceinline__ bool zdrop_test(Ez &ez, int32_t H, int r, int t, int zdrop, int e) {  // ksw_apply_zdrop, rotated
    if (H > ez.max) {
        ez.max = H, ez.max_t = t, ez.max_q = r - t;
    } else if (t >= ez.max_t && r - t >= ez.max_q) {
        const int tl = t - ez.max_t, ql = (r - t) - ez.max_q, l = tl > ql ? tl - ql : ql - tl;
        if (zdrop >= 0 && ez.max - H > zdrop + l * e) {
            ez.zdropped = 1;
            return true;
        }
    }
    return false;
}

__device__ __forceinline__ int push_op(uint32_t *cigar, int n, uint32_t op, int len) {
    if (n == 0 || op != (cigar[n - 1] & 0xfu)) {
        cigar[n] = (uint32_t)len << 4 | op;
        return n + 1;
    }
    cigar[n - 1] += (uint32_t)len << 4;
    return n;
}

// ksw_backtrack, rotated matrix, no introns
__device__ int backtrack(bool is_rev, const uint8_t *p, const int32_t *off, const int32_t *off_end, int n_col, int i0, int j0,
                         uint32_t *cigar) {
    int n = 0, i = i0, j = j0, state = 0;
    while (i >= 0 && j >= 0) {
        const int r = i + j;
        int force = -1;
        if (i < off[r]) force = 2;
        if (i > off_end[r]) force = 1;
        const uint32_t tmp = force < 0 ? p[(size_t)r * (size_t)n_col + (size_t)(i - off[r])] : 0u;
        if (state == 0) state = (int)(tmp & 7u);
        else if (!(tmp >> (state + 2) & 1u)) state = 0;
        if (state == 0) state = (int)(tmp & 7u);
        if (force >= 0) state = force;
        if (state == 0) n = push_op(cigar, n, 0, 1), --i, --j;
        else if (state == 1 || state == 3) n = push_op(cigar, n, 2, 1), --i;
        else n = push_op(cigar, n, 1, 1), --j;
    }
    if (i >= 0) n = push_op(cigar, n, 2, i + 1);
    if (j >= 0) n = push_op(cigar, n, 1, j + 1);
    if (!is_rev)
        for (int k = 0; k < n >> 1; ++k) {
            const uint32_t t = cigar[k];
            cigar[k] = cigar[n - 1 - k], cigar[n - 1 - k] = t;
        }
    return n;
}

__device__ __forceinline__ long long wave_max_i64(long long v) {
    for (int o = 32; o; o >>= 1) {
        const int lo = __shfl_xor((int)(v & 0xffffffffll), o, 64), hi = __shfl_xor((int)(v >> 32), o, 64);
        const long long u = (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
        v = u > v ? u : v;
    }
    return v;
}

__global__ void __launch_bounds__(64) ksw_extd2_kernel(const KswJobDev *__restrict__ jobs, const uint8_t *__restrict__ pool,
                                                        int8_t *__restrict__ diff_pool, int32_t *__restrict__ h_pool,
                                                        uint8_t *__restrict__ p_pool, int32_t *__restrict__ off_pool,
                                                        uint32_t *__restrict__ cigar_pool, Ez *__restrict__ results,
                                                        const int32_t *__restrict__ ids, int lds_cap) {
    extern __shared__ int32_t lds_h[];  // lds_cap int32 of H, then 7 x lds_cap bytes of differences (nothing for the HBM tier)
    int8_t *lds_diff = (int8_t *)(lds_h + lds_cap);
    const int job = ids[blockIdx.x];
    const KswJobDev &J = jobs[job];
    const int lane = (int)threadIdx.x;
    const int qlen = J.qlen, tlen = J.tlen, flag = J.flag, m = J.m;
    const uint8_t *query = pool + J.q_off, *target = pool + J.t_off;
    const bool with_cigar = !(flag & F_SCORE_ONLY), approx_max = (flag & F_APPROX_MAX) != 0, right = with_cigar && (flag & F_RIGHT);
    Ez ez;
    ez.max_q = ez.max_t = ez.mqe_t = ez.mte_q = -1;
    ez.max = 0, ez.score = ez.mqe = ez.mte = kNegInf;
    ez.n_cigar = 0, ez.zdropped = 0, ez.reach_end = 0;
    int q = J.q, e = J.e, q2 = J.q2, e2 = J.e2;
    const int qe_first = q + e;  // the reference takes q + e before the two pieces may be swapped, for the first cell only
    bool run = !(m <= 1 || qlen <= 0 || tlen <= 0);
    if (run && q2 + e2 < q + e) {
        int t_ = q;
        q = q2, q2 = t_, t_ = e, e = e2, e2 = t_;
    }
    const int qe = q + e, qe2 = q2 + e2;
    const int8_t sc_mch = J.mat[0], sc_mis = J.mat[1];
    const int8_t sc_N = run ? (J.mat[m * m - 1] == 0 ? (int8_t)-e2 : J.mat[m * m - 1]) : (int8_t)0;
    int w = J.w;
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    const int tl16 = (tlen + 15) / 16 * 16;
    int n_col = qlen < tlen ? qlen : tlen;
    n_col = (((n_col < w + 1 ? n_col : w + 1) + 15) / 16 + 1) * 16;
    if (run) {
        int min_sc = J.mat[1];
        for (int t = 1; t < m * m; ++t) min_sc = min_sc < J.mat[t] ? min_sc : J.mat[t];
        if (-min_sc > 2 * (q + e)) run = false;  // "otherwise, we won't see any mismatches"
    }
    int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
    if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
    const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;

    const bool in_lds = tl16 <= lds_cap;
    int8_t *u = in_lds ? lds_diff : diff_pool + J.diff_off;
    int8_t *v = u + tl16, *x = v + tl16, *y = x + tl16, *x2 = y + tl16, *y2 = x2 + tl16, *s = y2 + tl16;
    int32_t *H = in_lds ? lds_h : h_pool + J.h_off;
    uint8_t *p = p_pool + J.p_off;
    int32_t *off = off_pool + J.off_off, *off_end = off + (qlen + tlen - 1);

    if (run) {
        for (int t = lane; t < tl16; t += 64) {
            u[t] = v[t] = x[t] = y[t] = (int8_t)(-q - e);
            x2[t] = y2[t] = (int8_t)(-q2 - e2);
            s[t] = 0;
            if (!approx_max) H[t] = kNegInf;
        }
    }
    __syncthreads();

    int last_st = -1, last_en = -1;
    int32_t H0 = 0, last_H0_t = 0;
    const int n_diag = run ? qlen + tlen - 1 : 0;
    for (int r = 0; r < n_diag; ++r) {
        int st = 0, en = tlen - 1;
        if (st < r - qlen + 1) st = r - qlen + 1;
        if (en > r) en = r;
        if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
        if (en > (r + w) >> 1) en = (r + w) >> 1;
        if (st > en) {
            ez.zdropped = 1;
            break;
        }
        const int st0 = st, en0 = en;
        st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
        int8_t x1, x21, v1;
        if (st > 0) {
            if (st - 1 >= last_st && st - 1 <= last_en) x1 = x[st - 1], x21 = x2[st - 1], v1 = v[st - 1];
            else x1 = (int8_t)(-q - e), x21 = (int8_t)(-q2 - e2), v1 = (int8_t)(-q - e);
        } else {
            x1 = (int8_t)(-q - e), x21 = (int8_t)(-q2 - e2);
            v1 = (int8_t)(r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2);
        }
        if (en >= r && lane == 0) {
            y[r] = (int8_t)(-q - e), y2[r] = (int8_t)(-q2 - e2);
            u[r] = (int8_t)(r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2);
        }
        // scores: the reference refreshes runs of 16 from st0; positions behind the query's / target's end pair with zero padding
        if (!(flag & F_GENERIC_SC)) {
            int s_end = st0 + (en0 - st0) / 16 * 16 + 16;  // exclusive
            if (s_end > tl16) s_end = tl16;
            for (int t = st0 + lane; t < s_end; t += 64) {
                const uint8_t a = t < tlen ? target[t] : (uint8_t)0, b = t <= r ? query[r - t] : (uint8_t)0;
                s[t] = (a == (uint8_t)(m - 1) || b == (uint8_t)(m - 1)) ? sc_N : a == b ? sc_mch : sc_mis;
            }
        } else {
            for (int t = st0 + lane; t <= en0; t += 64) s[t] = J.mat[target[t] * m + query[r - t]];
        }
        if (with_cigar && lane == 0) off[r] = st, off_end[r] = en;
        __syncthreads();
        // cells of the 16-byte blocks around [st0, en0], 64 at a time from the top: the left neighbour (t - 1) of a pass's
        // lowest cell belongs to the next pass, so every cell reads what diagonal r - 1 left
        for (int hi = en; hi >= st; hi -= 64) {
            const int t = hi - lane;
            const bool act = t >= st;
            int8_t z = 0, xt1 = 0, vt1 = 0, x2t1 = 0, ut = 0, yt = 0, y2t = 0;
            if (act) {
                z = s[t];
                xt1 = t > st ? x[t - 1] : x1, vt1 = t > st ? v[t - 1] : v1, x2t1 = t > st ? x2[t - 1] : x21;
                ut = u[t], yt = y[t], y2t = y2[t];
            }
            __builtin_amdgcn_wave_barrier();  // every lane has read its neighbour's cell before that neighbour overwrites it (the
                                              // wavefront runs in lock step; this only keeps the compiler from moving a store up)
            if (act) {
                int8_t a = (int8_t)(xt1 + vt1), b = (int8_t)(yt + ut), a2 = (int8_t)(x2t1 + vt1), b2 = (int8_t)(y2t + ut);
                uint8_t d;
                if (!right) {
                    d = a > z ? 1 : 0;
                    z = z > a ? z : a;
                    d = b > z ? 2 : d;
                    z = z > b ? z : b;
                    d = a2 > z ? 3 : d;
                    z = z > a2 ? z : a2;
                    d = b2 > z ? 4 : d;
                    z = z > b2 ? z : b2;
                } else {
                    d = z > a ? 0 : 1;
                    z = z > a ? z : a;
                    d = z > b ? d : 2;
                    z = z > b ? z : b;
                    d = z > a2 ? d : 3;
                    z = z > a2 ? z : a2;
                    d = z > b2 ? d : 4;
                    z = z > b2 ? z : b2;
                }
                z = z < sc_mch ? z : sc_mch;
                u[t] = (int8_t)(z - vt1), v[t] = (int8_t)(z - ut);
                int8_t tmp = (int8_t)(z - q);
                a = (int8_t)(a - tmp), b = (int8_t)(b - tmp);
                tmp = (int8_t)(z - q2);
                a2 = (int8_t)(a2 - tmp), b2 = (int8_t)(b2 - tmp);
                const bool ka = right ? !(0 > a) : a > 0, kb = right ? !(0 > b) : b > 0;
                const bool ka2 = right ? !(0 > a2) : a2 > 0, kb2 = right ? !(0 > b2) : b2 > 0;
                x[t] = (int8_t)((a > 0 ? a : 0) - qe), y[t] = (int8_t)((b > 0 ? b : 0) - qe);
                x2[t] = (int8_t)((a2 > 0 ? a2 : 0) - qe2), y2[t] = (int8_t)((b2 > 0 ? b2 : 0) - qe2);
                if (with_cigar) {
                    d |= (ka ? 0x08 : 0) | (kb ? 0x10 : 0) | (ka2 ? 0x20 : 0) | (kb2 ? 0x40 : 0);
                    p[(size_t)r * (size_t)n_col + (size_t)(t - st)] = d;
                }
            }
        }
        __syncthreads();
        bool stop = false;
        if (!approx_max) {
            int32_t max_H, max_t;
            if (r > 0) {
                // H[en0] first (from the neighbour's value of diagonal r - 1), then H[t] += v[t] below it; the winner among equal
                // maxima is the one the reference's four-at-a-time search finds: en0, then position classes (t - st0) mod 4 in
                // order, each from the left, then the remainder from the left
                const int32_t h_top = en0 > 0 ? H[en0 - 1] + u[en0] : H[en0] + v[en0];
                __builtin_amdgcn_wave_barrier();
                const int en1 = st0 + (en0 - st0) / 4 * 4, n_grp = (en1 - st0) / 4;
                // key = (score, -rank) as one signed 64-bit number (the score's bits moved up without shifting a negative value)
                auto pack = [](int32_t hv, int rk) { return (long long)(((unsigned long long)(uint32_t)hv << 32) | (uint32_t)(0x7fffffff - rk)); };
                long long best = pack(h_top, 0);
                for (int t = st0 + lane; t < en0; t += 64) {
                    const int32_t h = H[t] + (int32_t)v[t];
                    H[t] = h;
                    const int rank = t < en1 ? 1 + ((t - st0) & 3) * n_grp + ((t - st0) >> 2) : 1 + 4 * n_grp + (t - en1);
                    const long long key = pack(h, rank);
                    best = key > best ? key : best;
                }
                if (lane == 0) H[en0] = h_top;
                best = wave_max_i64(best);
                max_H = (int32_t)(best >> 32);
                const int rank = 0x7fffffff - (int)(best & 0xffffffffll);
                if (rank == 0) max_t = en0;
                else if (rank <= 4 * n_grp) max_t = st0 + ((rank - 1) % n_grp) * 4 + (rank - 1) / n_grp;
                else max_t = en1 + (rank - 1 - 4 * n_grp);
            } else {
                if (lane == 0) H[0] = (int32_t)v[0] - qe_first;
                max_H = (int32_t)v[0] - qe_first, max_t = 0;
            }
            __syncthreads();
            if (en0 == tlen - 1 && H[en0] > ez.mte) ez.mte = H[en0], ez.mte_q = r - en;
            if (r - st0 == qlen - 1 && H[st0] > ez.mqe) ez.mqe = H[st0], ez.mqe_t = st0;
            if (zdrop_test(ez, max_H, r, max_t, J.zdrop, e2)) stop = true;
            else if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = H[tlen - 1];
        } else {
            if (r > 0) {
                if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
                    const int32_t d0 = v[last_H0_t], d1 = u[last_H0_t + 1];
                    if (d0 > d1) H0 += d0;
                    else H0 += d1, ++last_H0_t;
                } else if (last_H0_t >= st0 && last_H0_t <= en0) {
                    H0 += v[last_H0_t];
                } else {
                    ++last_H0_t, H0 += u[last_H0_t];
                }
            } else H0 = (int32_t)v[0] - qe_first, last_H0_t = 0;
            if ((flag & F_APPROX_DROP) && zdrop_test(ez, H0, r, last_H0_t, J.zdrop, e2)) stop = true;
            else if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = H0;
        }
        if (stop) break;
        last_st = st, last_en = en;
        __syncthreads();  // the next diagonal's boundary reads and presets come after everybody is done with this one
    }
    __syncthreads();
    if (lane == 0) {
        if (run && with_cigar) {
            const bool rev = (flag & F_REV_CIGAR) != 0;
            uint32_t *cigar = cigar_pool + J.cigar_off;
            if (!ez.zdropped && !(flag & F_EXTZ_ONLY)) ez.n_cigar = backtrack(rev, p, off, off_end, n_col, tlen - 1, qlen - 1, cigar);
            else if (!ez.zdropped && (flag & F_EXTZ_ONLY) && ez.mqe + J.end_bonus > ez.max) {
                ez.reach_end = 1;
                ez.n_cigar = backtrack(rev, p, off, off_end, n_col, ez.mqe_t, qlen - 1, cigar);
            } else if (ez.max_t >= 0 && ez.max_q >= 0) ez.n_cigar = backtrack(rev, p, off, off_end, n_col, ez.max_t, ez.max_q, cigar);
        }
        results[job] = ez;
    }
}

bool hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    fprintf(stderr, "[ndgpu_overlap] %s: %s\n", what, hipGetErrorString(e));
    return false;
}

template <class T> struct Dev {  // a block of the overlap library's pool (csrc/ovl_pool.h): no driver call per batch in steady state
    T *p = nullptr;
    bool alloc(size_t n) {
        try {
            p = (T *)ndovl::pool_alloc(sizeof(T) * (n ? n : 1));
        } catch (const std::exception &) {
            p = nullptr;
            return false;
        }
        return true;
    }
    ~Dev() {
        if (p) ndovl::pool_free(p);
    }
};

}  // namespace

static int run_batch(const ndgpu_ksw_job *jobs, int n, ndgpu_ksw_result *res);

// device bytes one problem needs beyond its sequences: the backtrack matrix dominates ((qlen + tlen) x (band + 16) bytes)
static uint64_t scratch_of(const ndgpu_ksw_job &J) {
    const uint64_t tl16 = ((uint64_t)(J.tlen > 0 ? J.tlen : 0) + 15) / 16 * 16, diags = (uint64_t)(J.qlen > 0 ? J.qlen : 0) + (uint64_t)(J.tlen > 0 ? J.tlen : 0);
    const int w = J.w < 0 ? (J.tlen > J.qlen ? J.tlen : J.qlen) : J.w;
    uint64_t n_col = (uint64_t)(J.qlen < J.tlen ? J.qlen : J.tlen);
    n_col = (((n_col < (uint64_t)w + 1 ? n_col : (uint64_t)w + 1) + 15) / 16 + 1) * 16;
    return ((J.flag & F_SCORE_ONLY) ? 0 : diags * (n_col + 8)) + 11 * tl16 + 5 * diags;
}

extern "C" int ndgpu_ksw_extd2_batch(const ndgpu_ksw_job *jobs, int n, ndgpu_ksw_result *res) {
    // sub-batches bounded by device scratch: a caller may hand over every gap of a read set, and the backtrack matrix of a 1 kb x 1 kb
    // problem is 2 MB.  A launch should hold tens of thousands of problems -- one wavefront each, 8,192 resident at a time, and a
    // launch lasts at least as long as its longest problem: with the 8 GB of round 3 a 20 Mb read set went through 190 launches
    // of ~4,000 wavefronts, each with its own allocations (79 s; profiles/r04_overlap_c_timing.json) -- so the budget is a third of
    // what the device has free (what the pool holds idle counts as free), NDGPU_KSW_SCRATCH_GB overrides.
    uint64_t budget = 8ull << 30;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::max<uint64_t>(budget, ((uint64_t)free_b + ndovl::pool_cached_bytes()) / 3);
    }
    if (const char *e = getenv("NDGPU_KSW_SCRATCH_GB")) budget = (uint64_t)(atof(e) * (double)(1ull << 30));
    int a = 0;
    while (a < n) {
        int b = a;
        uint64_t used = 0;
        while (b < n && (b == a || used + scratch_of(jobs[b]) <= budget)) used += scratch_of(jobs[b++]);
        const int rc = run_batch(jobs + a, b - a, res + a);
        if (rc != 0) return rc;
        a = b;
    }
    return 0;
}

static int run_batch(const ndgpu_ksw_job *jobs, int n, ndgpu_ksw_result *res) {
    if (n <= 0) return 0;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        fprintf(stderr, "[ndgpu_overlap] FATAL: no HIP device visible: ksw_extd2 has no CPU fallback\n");
        return -1;
    }
    int dev = 0;
    if (const char *e = getenv("NDGPU_DEVICE")) dev = atoi(e) % n_dev;
    if (!hip_ok(hipSetDevice(dev), "hipSetDevice")) return -1;
    std::vector<KswJobDev> h((size_t)n);
    std::vector<uint8_t> pool;
    uint64_t diff_total = 0, h_total = 0, p_total = 0, off_total = 0, cigar_total = 0;
    std::vector<int32_t> tier[3];  // by target length: small LDS tables, large LDS tables, difference arrays in HBM
    for (int i = 0; i < n; i++) {
        const ndgpu_ksw_job &J = jobs[i];
        if (J.qlen < 0 || J.tlen < 0 || J.m < 0 || J.m > 5 || (J.qlen > 0 && !J.query) || (J.tlen > 0 && !J.target) || (J.m > 0 && !J.mat)) return -2;
        KswJobDev &D = h[(size_t)i];
        memset(&D, 0, sizeof(D));
        D.qlen = J.qlen, D.tlen = J.tlen, D.w = J.w, D.zdrop = J.zdrop, D.end_bonus = J.end_bonus, D.flag = J.flag;
        D.m = J.m, D.q = J.gapo, D.e = J.gape, D.q2 = J.gapo2, D.e2 = J.gape2;
        for (int k = 0; k < J.m * J.m; k++) D.mat[k] = J.mat[k];
        D.q_off = pool.size();
        pool.insert(pool.end(), J.query, J.query + J.qlen);
        D.t_off = pool.size();
        pool.insert(pool.end(), J.target, J.target + J.tlen);
        for (int k = 0; k < J.qlen && J.m > 0; k++)
            if (J.query[k] >= (uint8_t)J.m) return -2;
        for (int k = 0; k < J.tlen && J.m > 0; k++)
            if (J.target[k] >= (uint8_t)J.m) return -2;
        const uint64_t tl16 = ((uint64_t)J.tlen + 15) / 16 * 16, diags = (uint64_t)(J.qlen + J.tlen > 0 ? J.qlen + J.tlen - 1 : 0);
        int w = J.w < 0 ? (J.tlen > J.qlen ? J.tlen : J.qlen) : J.w;
        uint64_t n_col = (uint64_t)(J.qlen < J.tlen ? J.qlen : J.tlen);
        n_col = (((n_col < (uint64_t)w + 1 ? n_col : (uint64_t)w + 1) + 15) / 16 + 1) * 16;
        D.diff_off = diff_total, D.h_off = h_total;
        if (tl16 > (uint64_t)kLdsTarget) diff_total += 7 * tl16, h_total += tl16;
        tier[tl16 <= (uint64_t)kLdsSmall ? 0 : tl16 <= (uint64_t)kLdsTarget ? 1 : 2].push_back(i);
        D.p_off = p_total, D.off_off = off_total, D.cigar_off = cigar_total;
        if (!(J.flag & F_SCORE_ONLY)) p_total += diags * n_col + 16, off_total += 2 * diags;
        cigar_total += (uint64_t)J.qlen + (uint64_t)J.tlen + 2;
    }
    pool.push_back(0);
    Dev<KswJobDev> d_jobs;
    Dev<uint8_t> d_pool, d_p;
    Dev<int8_t> d_diff;
    Dev<int32_t> d_h, d_off;
    Dev<uint32_t> d_cigar;
    Dev<Ez> d_res;
    Dev<int32_t> d_ids;
    std::vector<int32_t> ids;
    for (auto &t : tier) {  // the longest problems of a tier first: a launch lasts as long as its last wavefront
        std::stable_sort(t.begin(), t.end(), [&](int32_t a_, int32_t b_) { return jobs[a_].qlen + jobs[a_].tlen > jobs[b_].qlen + jobs[b_].tlen; });
        ids.insert(ids.end(), t.begin(), t.end());
    }
    hipStream_t st = nullptr;
    bool ok = hip_ok(hipStreamCreate(&st), "hipStreamCreate") && d_jobs.alloc((size_t)n) && d_pool.alloc(pool.size()) && d_p.alloc(p_total) &&
              d_diff.alloc(diff_total) && d_h.alloc(h_total) && d_off.alloc(off_total) && d_cigar.alloc(cigar_total) && d_res.alloc((size_t)n) && d_ids.alloc((size_t)n);
    std::vector<Ez> h_res((size_t)n);
    std::vector<uint32_t> h_cigar((size_t)cigar_total + 1);
    if (ok) {
        ok = hip_ok(hipMemcpyAsync(d_pool.p, pool.data(), pool.size(), hipMemcpyHostToDevice, st), "upload") &&
             hip_ok(hipMemcpyAsync(d_jobs.p, h.data(), sizeof(KswJobDev) * (size_t)n, hipMemcpyHostToDevice, st), "upload") &&
             hip_ok(hipMemcpyAsync(d_ids.p, ids.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st), "upload");
        if (ok) {
            const int caps[3] = {kLdsSmall, kLdsTarget, 0};
            size_t first = 0;
            for (int k = 0; k < 3; k++) {
                if (!tier[k].empty())
                    hipLaunchKernelGGL(ksw_extd2_kernel, dim3((unsigned)tier[k].size()), dim3(64), (size_t)caps[k] * 11, st, d_jobs.p, d_pool.p,
                                       d_diff.p, d_h.p, d_p.p, d_off.p, d_cigar.p, d_res.p, d_ids.p + first, caps[k]);
                first += tier[k].size();
            }
            ok = hip_ok(hipGetLastError(), "launch") &&
                 hip_ok(hipMemcpyAsync(h_res.data(), d_res.p, sizeof(Ez) * (size_t)n, hipMemcpyDeviceToHost, st), "download") &&
                 hip_ok(hipMemcpyAsync(h_cigar.data(), d_cigar.p, sizeof(uint32_t) * (size_t)cigar_total, hipMemcpyDeviceToHost, st), "download") &&
                 hip_ok(hipStreamSynchronize(st), "sync");
        }
    }
    if (st) {
        // (a failure after the launches: the kernels may still be running on `st`, and the blocks below go back to a pool that hands
        // them to the next caller at once -- hipStreamDestroy does not wait)
        if (!ok) (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
    }
    if (!ok) return -1;
    for (int i = 0; i < n; i++) {
        const Ez &z = h_res[(size_t)i];
        ndgpu_ksw_result &R = res[i];
        R.max = z.max, R.zdropped = z.zdropped, R.max_q = z.max_q, R.max_t = z.max_t, R.mqe = z.mqe, R.mqe_t = z.mqe_t, R.mte = z.mte;
        R.mte_q = z.mte_q, R.score = z.score, R.n_cigar = z.n_cigar, R.reach_end = z.reach_end;
        R.cigar = nullptr;
        if (z.n_cigar > 0) {
            R.cigar = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)z.n_cigar);
            if (!R.cigar) return -1;
            memcpy(R.cigar, h_cigar.data() + h[(size_t)i].cigar_off, sizeof(uint32_t) * (size_t)z.n_cigar);
        }
    }
    return 0;
}

// minimap2/ksw2.h:60-61.  ez->cigar is the caller's grow-by-doubling buffer (ksw_push_cigar, ksw2.h:96-110): with km == NULL it
// belongs to malloc, otherwise to the reference's arena allocator (kalloc.c), whose krealloc is looked up in the host program
// -- minimap2's mm_align_pair passes an arena -- and never mixed with free().
extern "C" void ksw_extd2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                              int8_t gapo, int8_t gape, int8_t gapo2, int8_t gape2, int w, int zdrop, int end_bonus, int flag,
                              ndgpu_ksw_extz *ez) {
    (void)km;
    ndgpu_ksw_job j;
    j.qlen = qlen, j.query = query, j.tlen = tlen, j.target = target, j.m = m, j.mat = mat, j.gapo = gapo, j.gape = gape, j.gapo2 = gapo2;
    j.gape2 = gape2, j.w = w, j.zdrop = zdrop, j.end_bonus = end_bonus, j.flag = flag;
    ndgpu_ksw_result r;
    memset(&r, 0, sizeof(r));
    if (ndgpu_ksw_extd2_batch(&j, 1, &r) != 0) abort();  // fail loudly: no device, no result
    ez->max = (uint32_t)r.max & 0x7fffffffu, ez->zdropped = (uint32_t)r.zdropped & 1u;
    ez->max_q = r.max_q, ez->max_t = r.max_t, ez->mqe = r.mqe, ez->mqe_t = r.mqe_t, ez->mte = r.mte, ez->mte_q = r.mte_q, ez->score = r.score;
    ez->n_cigar = r.n_cigar, ez->reach_end = r.reach_end;
    if (r.n_cigar > 0) {
        if (!ez->cigar || ez->m_cigar < r.n_cigar) {
            int cap = ez->m_cigar > 0 ? ez->m_cigar : 4;
            while (cap < r.n_cigar) cap <<= 1;  // the capacities the reference's doubling would have reached
            if (!km) ez->cigar = (uint32_t *)realloc(ez->cigar, sizeof(uint32_t) * (size_t)cap);
            else {
                typedef void *(*krealloc_t)(void *, void *, size_t);
                static const krealloc_t kr = (krealloc_t)dlsym(RTLD_DEFAULT, "krealloc");
                if (!kr) {
                    fprintf(stderr, "[ndgpu_overlap] ksw_extd2_sse: called with an arena (km != NULL) but the host program exports no krealloc\n");
                    abort();
                }
                ez->cigar = (uint32_t *)kr(km, ez->cigar, sizeof(uint32_t) * (size_t)cap);
            }
            if (!ez->cigar) abort();
            ez->m_cigar = cap;
        }
        memcpy(ez->cigar, r.cigar, sizeof(uint32_t) * (size_t)r.n_cigar);
        free(r.cigar);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// ksw_ll_i16 (minimap2/ksw2_ll_sse.c:85-156 with the query profile of ksw_ll_qinit, :32-83, size 2): the local-alignment SCORE
// behind the inversion test of a z-dropped gap (mm_test_zdrop, minimap2/align.c:71-87) and behind mm_align1_inv (:790-845).
// It is Farrar's striped Smith-Waterman on eight 16-bit lanes, and its values are those of the striped schedule, not of the
// textbook recurrence: a vertical gap (E) is opened from H as it stands BEFORE the lazy-F pass, so a cell whose value arrived
// through an F that crossed a stripe boundary opens no E; the query is padded to a multiple of 8 with zero-score columns that
// take part in the maximum; of equal maxima the last target row and the last cell in stripe order win.  The kernel therefore runs
// the schedule itself: one problem per wavefront, lanes 0..7 = the eight stripes (the other lanes idle along so that the
// wavefront's control flow stays uniform), the H / E / Hmax rows in an HBM scratch slice [segment][stripe].
namespace {

struct LlJobDev {
    uint64_t q_off, t_off, h_off;  // codes in the byte pool; 4 x slen x 8 int16 of scratch
    int32_t qlen, tlen, gapo, gape;
    int8_t mat[25 + 3];
};
struct LlRes { int32_t score, qe, te; };

__device__ __forceinline__ int sat16(int v) { return v > 32767 ? 32767 : v < -32768 ? -32768 : v; }
__device__ __forceinline__ int subs_u16(int a, int b) { return a > b ? a - b : 0; }  // _mm_subs_epu16 on values that are never negative

__global__ __launch_bounds__(64) void ksw_ll_kernel(const LlJobDev *__restrict__ jobs, const uint8_t *__restrict__ pool,
                                                    int16_t *__restrict__ scratch, LlRes *__restrict__ res) {
    const LlJobDev J = jobs[blockIdx.x];
    const int lane = (int)threadIdx.x;
    const bool live = lane < 8;
    const int slen = (J.qlen + 7) / 8;
    const uint8_t *query = pool + J.q_off, *target = pool + J.t_off;
    int16_t *H0 = scratch + J.h_off, *H1 = H0 + (size_t)slen * 8, *E = H1 + (size_t)slen * 8, *Hmax = E + (size_t)slen * 8;
    const int gapoe = J.gapo + J.gape, gape = J.gape;
    if (live)
        for (int j = 0; j < slen; ++j) H0[j * 8 + lane] = 0, E[j * 8 + lane] = 0, Hmax[j * 8 + lane] = 0;
    __syncthreads();
    int gmax = 0, te = -1;
    for (int i = 0; i < J.tlen; ++i) {
        const int8_t *ma = J.mat + (int)target[i] * 5;
        int f = 0, mx = 0;
        int h = live && slen > 0 ? (int)H0[(slen - 1) * 8 + lane] : 0;
        h = __shfl_up(h, 1, 64);  // _mm_slli_si128(h, 2): stripe k starts from the end of stripe k - 1
        if (lane == 0) h = 0;
        for (int j = 0; j < slen; ++j) {
            const int pos = j + lane * slen;
            const int sc = live && pos < J.qlen ? (int)ma[query[pos]] : 0;
            h = sat16(h + sc);
            int e = live ? (int)E[j * 8 + lane] : 0;
            h = h > e ? h : e;
            h = h > f ? h : f;
            mx = mx > h ? mx : h;
            if (live) H1[j * 8 + lane] = (int16_t)h;
            h = subs_u16(h, gapoe);
            e = subs_u16(e, gape);
            e = e > h ? e : h;
            if (live) E[j * 8 + lane] = (int16_t)e;
            f = subs_u16(f, gape);
            f = f > h ? f : h;
            h = live ? (int)H0[j * 8 + lane] : 0;
        }
        bool done = false;
        for (int k = 0; k < 8 && !done; ++k) {  // the lazy-F pass
            f = __shfl_up(f, 1, 64);
            if (lane == 0) f = 0;
            for (int j = 0; j < slen; ++j) {
                int hh = live ? (int)H1[j * 8 + lane] : 0;
                hh = hh > f ? hh : f;
                if (live) H1[j * 8 + lane] = (int16_t)hh;
                hh = subs_u16(hh, gapoe);
                f = subs_u16(f, gape);
                if (!(__ballot(live && f > hh) & 0xffull)) {
                    done = true;
                    break;
                }
            }
        }
        for (int o = 4; o; o >>= 1) {
            const int v = __shfl_xor(mx, o, 64);
            mx = mx > v ? mx : v;
        }
        const int imax = __shfl(mx, 0, 64);
        if (imax >= gmax) {
            gmax = imax, te = i;
            if (live)
                for (int j = 0; j < slen; ++j) Hmax[j * 8 + lane] = H1[j * 8 + lane];
        }
        int16_t *t = H1;
        H1 = H0, H0 = t;
    }
    __syncthreads();
    if (lane == 0) {
        int qe = -1;
        for (int i = 0; i < slen * 8; ++i)
            if ((int)(uint16_t)Hmax[i] == gmax) qe = i / 8 + i % 8 * slen;
        LlRes r;
        r.score = gmax, r.qe = qe, r.te = te;
        res[blockIdx.x] = r;
    }
}

}  // namespace

// a batch of ksw_ll_i16 problems (see above; include/ndgpu_overlap.h); used by csrc/ovl_cigar.cpp
extern "C" int ndgpu_ksw_ll_batch(const ndgpu_ll_job *jobs, int n, ndgpu_ll_result *out) {
    if (n <= 0) return 0;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        fprintf(stderr, "[ndgpu_overlap] FATAL: no HIP device visible: ksw_ll has no CPU fallback\n");
        return -1;
    }
    int dev = 0;
    if (const char *e = getenv("NDGPU_DEVICE")) dev = atoi(e) % n_dev;
    if (!hip_ok(hipSetDevice(dev), "hipSetDevice")) return -1;
    std::vector<LlJobDev> h((size_t)n);
    std::vector<uint8_t> pool;
    uint64_t h_total = 0;
    for (int i = 0; i < n; i++) {
        const ndgpu_ll_job &J = jobs[i];
        if (J.qlen < 0 || J.tlen < 0 || (J.qlen > 0 && !J.query) || (J.tlen > 0 && !J.target) || !J.mat) return -2;
        LlJobDev &D = h[(size_t)i];
        memset(&D, 0, sizeof(D));
        D.qlen = J.qlen, D.tlen = J.tlen, D.gapo = J.gapo, D.gape = J.gape;
        for (int k = 0; k < 25; k++) D.mat[k] = J.mat[k];
        for (int k = 0; k < J.qlen; k++)
            if (J.query[k] > 4) return -2;
        for (int k = 0; k < J.tlen; k++)
            if (J.target[k] > 4) return -2;
        D.q_off = pool.size();
        pool.insert(pool.end(), J.query, J.query + J.qlen);
        D.t_off = pool.size();
        pool.insert(pool.end(), J.target, J.target + J.tlen);
        D.h_off = h_total;
        h_total += 4ull * 8ull * (uint64_t)((J.qlen + 7) / 8);
    }
    pool.push_back(0);
    Dev<LlJobDev> d_jobs;
    Dev<uint8_t> d_pool;
    Dev<int16_t> d_h;
    Dev<LlRes> d_res;
    hipStream_t st = nullptr;
    bool ok = hip_ok(hipStreamCreate(&st), "hipStreamCreate") && d_jobs.alloc((size_t)n) && d_pool.alloc(pool.size()) && d_h.alloc(h_total + 1) &&
              d_res.alloc((size_t)n);
    std::vector<LlRes> r((size_t)n);
    if (ok) {
        ok = hip_ok(hipMemcpyAsync(d_pool.p, pool.data(), pool.size(), hipMemcpyHostToDevice, st), "upload") &&
             hip_ok(hipMemcpyAsync(d_jobs.p, h.data(), sizeof(LlJobDev) * (size_t)n, hipMemcpyHostToDevice, st), "upload");
        if (ok) {
            hipLaunchKernelGGL(ksw_ll_kernel, dim3((unsigned)n), dim3(64), 0, st, d_jobs.p, d_pool.p, d_h.p, d_res.p);
            ok = hip_ok(hipGetLastError(), "launch") &&
                 hip_ok(hipMemcpyAsync(r.data(), d_res.p, sizeof(LlRes) * (size_t)n, hipMemcpyDeviceToHost, st), "download") &&
                 hip_ok(hipStreamSynchronize(st), "sync");
        }
    }
    if (st) {
        // (a failure after the launches: the kernels may still be running on `st`, and the blocks below go back to a pool that hands
        // them to the next caller at once -- hipStreamDestroy does not wait)
        if (!ok) (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
    }
    if (!ok) return -1;
    for (int i = 0; i < n; i++) out[i].score = r[(size_t)i].score, out[i].qe = r[(size_t)i].qe, out[i].te = r[(size_t)i].te;
    return 0;
}
