// ext_kernels.hip -- the prefix / extension members of NextDenovo's greedy O(ND) family on the device:
//   ide         lib/align.c:80-141     edit steps until either sequence is exhausted -> (matches, block length)
//   alnpos      lib/align.c:146-253    the same forward pass + traceback -> columns, matches, start / end coordinates
//   extend_fwd  lib/align.c:256-340    forward pass with the running score (x + y) * d_factor - d; its peak is the extension
//   extend_rev  lib/align.c:343-426    the same from the 3' ends
// exported with the reference's signatures (lib/align.h:51-58; callers: minimap2/map.c:385-482, 941-956, lib/ctg_cns.c) and as a
// batched entry.  These calls come in batches of short, independent problems (the unaligned ends of every overlap of a read:
// max_d <= ide_ml = 6000, band 500), so one lane owns one problem: furthest-reaching x per diagonal in its slice of an HBM
// scratch array, one move bit per (d, k) cell for `alnpos`.  There is no CPU path: without a HIP device the calls fail loudly.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ndgpu_nextcorrect.h"

namespace {

enum { K_IDE = 0, K_ALNPOS = 1, K_FWD = 2, K_REV = 3 };

struct ExtJobDev {
    uint64_t q_off, t_off, fr_off, tr_off;
    int32_t ql, tl, max_d, band, kind;
    float d_factor;
};

// (x + y) * d_factor - d with the product rounded to float before the subtraction, as the host code computes it: a fused
// multiply-add keeps the exact product and flips near-ties of the peak test (hipcc contracts a * b - c by default, and HIP's
// __fmul_rn is a plain multiplication that gets contracted all the same), so the product is pinned in a register.
__device__ __forceinline__ float score_of(int xy, float f, int d) {
#pragma clang fp contract(off)
    float p = (float)xy * f;
    asm volatile("" : "+v"(p));
    return p - (float)d;
}

__global__ void __launch_bounds__(64) ext_kernel(const char *__restrict__ pool, const ExtJobDev *__restrict__ jobs, int n,
                                                  int32_t *__restrict__ fr_pool, uint32_t *__restrict__ tr_pool,
                                                  ndgpu_ext_result *__restrict__ outs) {
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= n) return;
    const ExtJobDev J = jobs[j];
    const char *q = pool + J.q_off, *t = pool + J.t_off;
    int32_t *fr = fr_pool + J.fr_off;  // zeroed by the host (clean_V)
    uint32_t *tr = tr_pool + J.tr_off;
    const int off = J.max_d + 2, ql = J.ql, tl = J.tl;
    const bool rev = J.kind == K_REV, ext = J.kind == K_FWD || J.kind == K_REV;
    int lo = 0, hi = 0, reach = -1, d = 0, k = 0, done = 0, x = 0, y = 0, fin_k = 0, o1 = 0, o2 = 0;
    float peak = 0;
    for (d = 0; d < J.max_d && hi - lo <= J.band && !done; ++d) {
        for (k = lo; k <= hi; k += 2) {
            int left;
            if (k == lo || (k != hi && fr[k - 1 + off] < fr[k + 1 + off])) x = fr[k + 1 + off], left = 0;
            else x = fr[k - 1 + off] + 1, left = 1;
            if (J.kind == K_ALNPOS) {
                const uint64_t bit = (uint64_t)d * (uint64_t)(d + 1) / 2 + (uint64_t)((k + d) >> 1);
                const uint32_t m = 1u << (bit & 31);
                uint32_t w = tr[bit >> 5];
                tr[bit >> 5] = left ? (w | m) : (w & ~m);
            }
            y = x - k;
            if (rev) {
                while (x < ql && y < tl && q[ql - x - 1] == t[tl - y - 1]) ++x, ++y;
            } else {
                while (x < ql && y < tl && q[x] == t[y]) ++x, ++y;
            }
            fr[k + off] = x;
            if (x + y > reach) {
                reach = x + y;
                if (ext) {
                    const float score = score_of(x + y, J.d_factor, d);
                    if (score > peak) peak = score, o1 = x, o2 = y;
                    else if (score < peak - 30) { done = 2; break; }
                }
            }
            if (x >= ql || y >= tl) {
                if (J.kind == K_IDE) o1 = x - (k + d) / 2, o2 = y + (k + d) / 2;
                else if (ext) {
                    const float score = score_of(x + y, J.d_factor, d);
                    if (score > 0) o1 = x, o2 = y;
                }
                done = 1, fin_k = k;
                break;
            }
        }
        if (done) break;
        int nlo = hi, nhi = lo;  // band re-centring (lib/align.c:473-489)
        for (int k2 = lo; k2 < nlo; k2 += 2)
            if (fr[k2 + off] * 2 - k2 >= reach - 150) nlo = k2;
        for (int k2 = hi; k2 > nhi; k2 -= 2)
            if (fr[k2 + off] * 2 - k2 >= reach - 150) nhi = k2;
        hi = nhi + 1, lo = nlo - 1;
    }
    ndgpu_ext_result r;
    r.done = done == 1 ? 1 : 0;
    r.a = o1, r.b = o2;
    for (int i = 0; i < 6; i++) r.pos[i] = 0;
    if (J.kind == K_ALNPOS && done == 1) {
        int cols = 0, gaps = 0;
        const uint32_t q_e = (uint32_t)x, t_e = (uint32_t)y;
        k = fin_k;
        --x;
        for (;;) {
            while (x >= 0 && x >= k && q[x] == t[x - k]) --x, ++cols;
            if (x < 0 || x - k < 0) break;
            const uint64_t bit = (uint64_t)d * (uint64_t)(d + 1) / 2 + (uint64_t)((k + d) >> 1);
            if (x < k || (tr[bit >> 5] >> (bit & 31) & 1u)) --k, --x;
            else ++k;
            ++cols, ++gaps, --d;
        }
        r.pos[0] = (uint32_t)cols, r.pos[1] = (uint32_t)(cols - gaps), r.pos[2] = (uint32_t)(x + 1 - k), r.pos[3] = t_e;
        r.pos[4] = (uint32_t)(x + 1), r.pos[5] = q_e;
    }
    outs[j] = r;
}

bool hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    fprintf(stderr, "[ndgpu] %s: %s\n", what, hipGetErrorString(e));
    return false;
}

}  // namespace

extern "C" int ndgpu_ext_batch(const ndgpu_ext_job *jobs, int n, ndgpu_ext_result *res) {
    if (n <= 0) return 0;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        fprintf(stderr, "[ndgpu] FATAL: no HIP device visible: ide / alnpos / extend_fwd / extend_rev have no CPU fallback\n");
        return -1;
    }
    int dev = 0;
    if (const char *e = getenv("NDGPU_DEVICE")) dev = atoi(e) % n_dev;
    if (!hip_ok(hipSetDevice(dev), "hipSetDevice")) return -1;
    std::vector<ExtJobDev> h(n);
    std::vector<char> pool;
    uint64_t fr_total = 0, tr_total = 1;
    for (int i = 0; i < n; i++) {
        const ndgpu_ext_job &J = jobs[i];
        if (J.q_len < 0 || J.t_len < 0 || J.max_d < 0 || J.kind < 0 || J.kind > 3) return -2;
        ExtJobDev &D = h[i];
        D.ql = J.q_len, D.tl = J.t_len, D.max_d = J.max_d, D.band = J.band_size, D.kind = J.kind, D.d_factor = J.d_factor;
        D.q_off = pool.size();
        pool.insert(pool.end(), J.q, J.q + J.q_len);
        D.t_off = pool.size();
        pool.insert(pool.end(), J.t, J.t + J.t_len);
        D.fr_off = fr_total;
        fr_total += 2 * ((uint64_t)J.max_d + 2) + 2;
        D.tr_off = tr_total;
        if (J.kind == K_ALNPOS) tr_total += ((uint64_t)J.max_d * ((uint64_t)J.max_d + 1) / 2 + 31) / 32 + 1;
    }
    pool.push_back(0);
    char *d_pool = nullptr;
    ExtJobDev *d_jobs = nullptr;
    int32_t *d_fr = nullptr;
    uint32_t *d_tr = nullptr;
    ndgpu_ext_result *d_out = nullptr;
    hipStream_t st = nullptr;
    bool ok = hip_ok(hipStreamCreate(&st), "hipStreamCreate") && hip_ok(hipMalloc((void **)&d_pool, pool.size()), "hipMalloc") &&
              hip_ok(hipMalloc((void **)&d_jobs, sizeof(ExtJobDev) * (size_t)n), "hipMalloc") &&
              hip_ok(hipMalloc((void **)&d_fr, sizeof(int32_t) * fr_total), "hipMalloc") &&
              hip_ok(hipMalloc((void **)&d_tr, sizeof(uint32_t) * tr_total), "hipMalloc") &&
              hip_ok(hipMalloc((void **)&d_out, sizeof(ndgpu_ext_result) * (size_t)n), "hipMalloc");
    if (ok) {
        ok = hip_ok(hipMemcpyAsync(d_pool, pool.data(), pool.size(), hipMemcpyHostToDevice, st), "upload") &&
             hip_ok(hipMemcpyAsync(d_jobs, h.data(), sizeof(ExtJobDev) * (size_t)n, hipMemcpyHostToDevice, st), "upload") &&
             hip_ok(hipMemsetAsync(d_fr, 0, sizeof(int32_t) * fr_total, st), "memset");
        if (ok) {
            hipLaunchKernelGGL(ext_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_pool, d_jobs, n, d_fr, d_tr, d_out);
            ok = hip_ok(hipGetLastError(), "launch") &&
                 hip_ok(hipMemcpyAsync(res, d_out, sizeof(ndgpu_ext_result) * (size_t)n, hipMemcpyDeviceToHost, st), "download") &&
                 hip_ok(hipStreamSynchronize(st), "sync");
        }
    }
    if (d_pool) (void)hipFree(d_pool);
    if (d_jobs) (void)hipFree(d_jobs);
    if (d_fr) (void)hipFree(d_fr);
    if (d_tr) (void)hipFree(d_tr);
    if (d_out) (void)hipFree(d_out);
    if (st) (void)hipStreamDestroy(st);
    return ok ? 0 : -1;
}

namespace {
ndgpu_ext_result one(const char *q, int ql, const char *t, int tl, int max_d, int band, float f, int kind) {
    ndgpu_ext_job j;
    j.q = q, j.q_len = ql, j.t = t, j.t_len = tl, j.max_d = max_d, j.band_size = band, j.d_factor = f, j.kind = kind;
    ndgpu_ext_result r;
    memset(&r, 0, sizeof(r));
    if (ndgpu_ext_batch(&j, 1, &r) != 0) abort();  // fail loudly: no device, no result
    return r;
}
}  // namespace

// lib/align.h:51-58: V / D are the reference's scratch arrays; the device owns the DP state, so they are not touched.
extern "C" void ide(const char *query_seq, int q_len, const char *target_seq, int t_len, int *, uint8_t **, int max_d, int band_size,
                    int *mlen, int *blen) {
    const ndgpu_ext_result r = one(query_seq, q_len, target_seq, t_len, max_d, band_size, 0.f, K_IDE);
    if (r.done) *mlen = r.a, *blen = r.b;  // left untouched otherwise, as the reference does
}

extern "C" void alnpos(const char *query_seq, int q_len, const char *target_seq, int t_len, int *, uint8_t **, int max_d, int band_size,
                       alignpos *aln) {
    const ndgpu_ext_result r = one(query_seq, q_len, target_seq, t_len, max_d, band_size, 0.f, K_ALNPOS);
    if (!r.done) return;
    aln->aln_len = r.pos[0], aln->aln_mlen = r.pos[1], aln->aln_t_s = r.pos[2], aln->aln_t_e = r.pos[3];
    aln->aln_q_s = r.pos[4], aln->aln_q_e = r.pos[5];
}

extern "C" void extend_fwd(const char *query_seq, int q_len, const char *target_seq, int t_len, int *, uint8_t **, int max_d,
                           int band_size, float d_factor, int *bstx, int *bsty) {
    const ndgpu_ext_result r = one(query_seq, q_len, target_seq, t_len, max_d, band_size, d_factor, K_FWD);
    *bstx = r.a, *bsty = r.b;
}

extern "C" void extend_rev(const char *query_seq, int q_len, const char *target_seq, int t_len, int *, uint8_t **, int max_d,
                           int band_size, float d_factor, int *bstx, int *bsty) {
    const ndgpu_ext_result r = one(query_seq, q_len, target_seq, t_len, max_d, band_size, d_factor, K_REV);
    *bstx = r.a, *bsty = r.b;
}
