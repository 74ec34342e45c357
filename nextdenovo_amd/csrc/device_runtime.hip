// Host runtime around the O(ND) kernels: packs ASCII sequences to the 2-bit pool,
// lays out per-task trace / ops regions in HBM, launches forward + traceback on one
// HIP stream, and hands column-kind streams back to the consensus engine.
//
// Buffers are grow-only and sized for MI355X's 288 GB HBM: a batch keeps every
// task's full trace (max_d rows x 16 B) resident, so there is no second pass and
// no host round trip between the forward sweep and the traceback.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "nd_device.h"
#include "nd_host.h"
#include "nd_runtime.h"

namespace ndgpu {

namespace {

#define HIP_CHECK(expr)                                                                                       \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            fprintf(stderr, "[ndgpu] HIP error %s at %s:%d: %s\n", hipGetErrorName(e_), __FILE__, __LINE__,   \
                    hipGetErrorString(e_));                                                                   \
            abort();                                                                                          \
        }                                                                                                     \
    } while (0)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    void reserve(size_t n) {
        if (n <= cap) return;
        if (p) HIP_CHECK(hipFree(p));
        size_t want = n + n / 4 + 1024;
        HIP_CHECK(hipMalloc((void **)&p, want * sizeof(T)));
        cap = want;
    }
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};

template <typename T>
struct PinBuf {
    T *p = nullptr;
    size_t cap = 0;
    void reserve(size_t n) {
        if (n <= cap) return;
        if (p) HIP_CHECK(hipHostFree(p));
        size_t want = n + n / 4 + 1024;
        HIP_CHECK(hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault));
        cap = want;
    }
    ~PinBuf() {
        if (p) (void)hipHostFree(p);
    }
};

struct CodeLut {
    uint8_t v[256];
    CodeLut() {
        memset(v, 0xff, sizeof(v));
        v['A'] = 0;
        v['C'] = 1;
        v['G'] = 2;
        v['T'] = 3;
    }
};
const CodeLut kCode;

// ASCII [ACGT]* -> 2-bit, LSB-first, appended at a word boundary.  Returns false on
// any other byte (the reference compares raw bytes; we only accept what lib/bseq.c
// can emit from a .2bit DB).
bool pack_append(std::vector<uint32_t> &pool, const char *s, size_t n) {
    const size_t w0 = pool.size();
    pool.resize(w0 + (n + 15) / 16);
    uint32_t *out = pool.data() + w0;
    size_t i = 0;
    unsigned bad = 0;
    for (size_t w = 0; i + 16 <= n; w++, i += 16) {
        uint32_t acc = 0;
        for (int b = 0; b < 16; b++) {
            const uint8_t c = kCode.v[(unsigned char)s[i + b]];
            bad |= c;
            acc |= (uint32_t)(c & 3u) << (2 * b);
        }
        out[w] = acc;
    }
    if (i < n) {
        uint32_t acc = 0;
        for (int b = 0; i + b < n; b++) {
            const uint8_t c = kCode.v[(unsigned char)s[i + b]];
            bad |= c;
            acc |= (uint32_t)(c & 3u) << (2 * b);
        }
        out[(n + 15) / 16 - 1] = acc;
    }
    return (bad & 0x80u) == 0;
}

}  // namespace

struct DeviceAligner::State {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    DevBuf<uint32_t> d_pool, d_ops, d_db;
    DevBuf<AlnTask> d_tasks;
    DevBuf<AlnOut> d_outs;
    DevBuf<uint64_t> d_trace;
    DevBuf<int32_t> d_mink, d_v, d_ids;
    PinBuf<uint32_t> h_ops;
    PinBuf<AlnOut> h_outs;
    std::vector<uint32_t> pool;
    std::vector<AlnTask> tasks;
    RuntimeStats stats;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    size_t trace_budget_bytes = (size_t)48 << 30;
};

DeviceAligner::DeviceAligner() : s_(new State) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        fprintf(stderr,
                "[ndgpu] FATAL: no HIP device visible (hipGetDeviceCount: %s). This library has no CPU fallback.\n",
                hipGetErrorString(e));
        abort();
    }
    const char *env = getenv("NDGPU_DEVICE");
    s_->device = env ? atoi(env) : 0;
    if (s_->device >= n) s_->device = s_->device % n;
    HIP_CHECK(hipSetDevice(s_->device));
    HIP_CHECK(hipStreamCreateWithFlags(&s_->stream, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreate(&s_->ev0));
    HIP_CHECK(hipEventCreate(&s_->ev1));
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > ((size_t)8 << 30))
        s_->trace_budget_bytes = free_b / 3;
    else s_->trace_budget_bytes = (size_t)2 << 30;
}

DeviceAligner::~DeviceAligner() { delete s_; }

DeviceAligner &DeviceAligner::instance() {
    static DeviceAligner *g = new DeviceAligner();  // intentionally leaked: no HIP calls at exit
    return *g;
}

void DeviceAligner::set_db(const uint32_t *pool_words, size_t n_words) {
    std::lock_guard<std::mutex> lock(s_->mu);
    HIP_CHECK(hipSetDevice(s_->device));
    s_->d_db.reserve(n_words + 2);
    HIP_CHECK(hipMemcpy(s_->d_db.p, pool_words, n_words * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(s_->d_db.p + n_words, 0, 2 * sizeof(uint32_t)));
}

void *DeviceAligner::stream() const { return s_->stream; }
RuntimeStats DeviceAligner::stats() const { return s_->stats; }
void DeviceAligner::reset_stats() { s_->stats = RuntimeStats(); }

static void limits_for(int total, int hq, int *max_d, int *band) {
    // lib/align.c:567-568,575-576 -- double arithmetic on the host, exactly as the reference
    if (hq) {
        *max_d = (int)((total > 1000 ? 0.1 : 0.5) * total);
        *band = (int)((total > 1000 ? 0.03 : 0.3) * total);
    } else {
        *max_d = (int)(0.4 * total);
        *band = (int)((total > 5000 ? 0.1 : 1) * total);
    }
}

void DeviceAligner::align_batch(AlnJob **jobs, size_t n) {
    if (n == 0) return;
    std::lock_guard<std::mutex> lock(s_->mu);
    HIP_CHECK(hipSetDevice(s_->device));
    size_t done = 0;
    while (done < n) {
        // take as many jobs as fit the trace budget
        size_t take = 0, bytes = 0;
        while (done + take < n) {
            const AlnJob &j = *jobs[done + take];
            int md, bd;
            limits_for(j.q_len + j.t_len, j.hq, &md, &bd);
            const size_t b = (size_t)md * (kFastRowWords * 8 + 4);
            if (take && bytes + b > s_->trace_budget_bytes) break;
            bytes += b;
            take++;
        }
        run_chunk(jobs + done, take);
        done += take;
    }
}

void DeviceAligner::run_chunk(AlnJob **jobs, size_t n) {
    State &S = *s_;
    std::vector<uint32_t> &pool = S.pool;
    std::vector<AlnTask> &tasks = S.tasks;
    pool.clear();
    tasks.assign(n, AlnTask());
    std::unordered_map<const char *, uint64_t> owners;  // shared target buffers packed once
    std::vector<uint8_t> bad(n, 0);
    uint64_t trace_words = 0, mink_rows = 0, ops_words = 0;
    for (size_t i = 0; i < n; i++) {
        AlnJob &j = *jobs[i];
        AlnTask &t = tasks[i];
        j.status = ALN_NONE;
        j.ops.clear();
        j.q_used = j.t_used = 0;
        t.q_len = j.q_len;
        t.t_len = j.t_len;
        S.stats.seq_bases += (uint64_t)j.q_len + (uint64_t)j.t_len;
        if (j.q_dev >= 0) t.q_off = (uint64_t)j.q_dev | kOffDb;
        else {
            t.q_off = (uint64_t)pool.size() * 16;
            if (!pack_append(pool, j.q, (size_t)j.q_len)) bad[i] = 1;
            S.stats.pool_bases += (uint64_t)j.q_len;
        }
        if (j.t_dev >= 0) t.t_off = (uint64_t)j.t_dev | kOffDb;
        else if (j.t_owner) {
            auto it = owners.find(j.t_owner);
            uint64_t base;
            if (it == owners.end()) {
                base = (uint64_t)pool.size() * 16;
                if (!pack_append(pool, j.t_owner, (size_t)j.t_owner_len)) bad[i] = 1;
                S.stats.pool_bases += (uint64_t)j.t_owner_len;
                owners.emplace(j.t_owner, bad[i] ? UINT64_MAX : base);
            } else base = it->second;
            if (base == UINT64_MAX) bad[i] = 1;
            t.t_off = base + (uint64_t)(j.t - j.t_owner);
        } else {
            t.t_off = (uint64_t)pool.size() * 16;
            if (!pack_append(pool, j.t, (size_t)j.t_len)) bad[i] = 1;
            S.stats.pool_bases += (uint64_t)j.t_len;
        }
        int md, bd;
        limits_for(j.q_len + j.t_len, j.hq, &md, &bd);
        if (bad[i]) md = 0;
        t.max_d = md;
        t.band = bd;
        t.row_words = kFastRowWords;
        t.trace_off = trace_words;
        t.mink_off = mink_rows;
        t.ops_off = ops_words;
        t.ops_cap = (uint32_t)(j.q_len + j.t_len);
        trace_words += (uint64_t)md * kFastRowWords;
        mink_rows += (uint64_t)md;
        ops_words += (uint64_t)(t.ops_cap + 15) / 16 + 1;
    }
    pool.push_back(0);
    pool.push_back(0);  // fetch16 reads one word past the last base

    S.d_pool.reserve(pool.size());
    S.d_tasks.reserve(n);
    S.d_outs.reserve(n);
    S.d_trace.reserve(trace_words + 2);
    S.d_mink.reserve(mink_rows + 2);
    S.d_ops.reserve(ops_words + 2);
    S.h_ops.reserve(ops_words + 2);
    S.h_outs.reserve(n);

    hipStream_t st = S.stream;
    HIP_CHECK(hipMemcpyAsync(S.d_pool.p, pool.data(), pool.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(S.d_tasks.p, tasks.data(), n * sizeof(AlnTask), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipEventRecord(S.ev0, st));
    launch_ond_forward(S.d_tasks.p, S.d_outs.p, S.d_pool.p, S.d_db.p, S.d_trace.p, S.d_mink.p, (int)n, st);
    HIP_CHECK(hipEventRecord(S.ev1, st));
    launch_ond_traceback(S.d_tasks.p, S.d_outs.p, S.d_pool.p, S.d_db.p, S.d_trace.p, S.d_mink.p, S.d_ops.p, nullptr, (int)n, st);
    HIP_CHECK(hipMemcpyAsync(S.h_outs.p, S.d_outs.p, n * sizeof(AlnOut), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(S.h_ops.p, S.d_ops.p, ops_words * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipGetLastError());
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, S.ev0, S.ev1));
    S.stats.forward_ms += ms;
    S.stats.forward_launches++;
    S.stats.tasks += n;

    // rare: live band wider than the LDS fast path -> rerun those with V in HBM
    std::vector<int32_t> wide;
    for (size_t i = 0; i < n; i++)
        if (S.h_outs.p[i].status == ST_NEED_WIDE) wide.push_back((int32_t)i);
    if (!wide.empty()) run_wide(jobs, n, wide);

    for (size_t i = 0; i < n; i++) {
        AlnJob &j = *jobs[i];
        const AlnOut &o = S.h_outs.p[i];
        const AlnTask &t = tasks[i];
        S.stats.cells += (uint64_t)o.cells;
        S.stats.d_steps += (uint64_t)o.d_steps;
        if ((uint32_t)o.max_band > S.stats.max_band) S.stats.max_band = (uint32_t)o.max_band;
        if (bad[i]) {
            static bool warned = false;
            if (!warned) {
                fprintf(stderr, "[ndgpu] sequence with bytes outside [ACGT]: alignment skipped\n");
                warned = true;
            }
            continue;
        }
        if (o.status == ST_ALIGNED) {
            j.status = ALN_OK;
            j.q_used = o.x_final;
            j.t_used = o.y_final;
            const uint32_t nc = (uint32_t)o.n_cols, c0 = t.ops_cap - nc;
            const uint32_t *W = S.h_ops.p + t.ops_off;
            j.ops.resize(nc);
            for (uint32_t c = 0; c < nc; c++) {
                const uint32_t cc = c0 + c;
                j.ops[c] = (uint8_t)((W[cc >> 4] >> ((cc & 15u) * 2u)) & 3u);
            }
            S.stats.trace_bits += (uint64_t)o.cells;
            S.stats.columns += nc;
        } else if (o.status == ST_GAP_ABORT) {
            j.status = ALN_GAP_ABORT;
            j.q_used = o.x_final;
            j.t_used = o.y_final;
            // the reference reports aln_len = 2: keep the last two alignment columns
            const uint32_t *W = S.h_ops.p + t.ops_off;
            j.ops.resize(2);
            for (uint32_t c = 0; c < 2; c++) {
                const uint32_t cc = t.ops_cap - 2 + c;
                j.ops[c] = (uint8_t)((W[cc >> 4] >> ((cc & 15u) * 2u)) & 3u);
            }
        } else {
            j.status = ALN_NONE;
        }
    }
}

void DeviceAligner::run_wide(AlnJob **jobs, size_t n, const std::vector<int32_t> &ids) {
    State &S = *s_;
    (void)jobs;
    // process in groups bounded by the trace budget; wide rows are band-cap sized
    size_t at = 0;
    DevBuf<uint64_t> trace;
    DevBuf<int32_t> mink;
    while (at < ids.size()) {
        size_t take = 0;
        uint64_t tw = 0, mr = 0, vw = 0;
        std::vector<AlnTask> patch;
        while (at + take < ids.size()) {
            AlnTask t = S.tasks[ids[at + take]];
            const uint32_t rw = (uint32_t)((t.band / 2 + 2 + 63) / 64);
            uint32_t ring = 256;
            while (ring < (uint32_t)t.band + 4) ring <<= 1;
            const uint64_t need = (uint64_t)t.max_d * rw * 8;
            if (take && (tw * 8 + need) > S.trace_budget_bytes) break;
            t.row_words = rw;
            t.trace_off = tw;
            t.mink_off = mr;
            t.v_off = vw;
            t.v_mask = ring - 1;
            tw += (uint64_t)t.max_d * rw;
            mr += (uint64_t)t.max_d;
            vw += ring;
            patch.push_back(t);
            take++;
        }
        trace.reserve(tw + 2);
        mink.reserve(mr + 2);
        S.d_v.reserve(vw + 2);
        S.d_ids.reserve(take);
        hipStream_t st = S.stream;
        for (size_t i = 0; i < take; i++) {
            S.tasks[ids[at + i]] = patch[i];
            HIP_CHECK(hipMemcpyAsync(S.d_tasks.p + ids[at + i], &S.tasks[ids[at + i]], sizeof(AlnTask),
                                     hipMemcpyHostToDevice, st));
        }
        HIP_CHECK(hipMemcpyAsync(S.d_ids.p, ids.data() + at, take * sizeof(int32_t), hipMemcpyHostToDevice, st));
        launch_ond_forward_wide(S.d_tasks.p, S.d_outs.p, S.d_pool.p, S.d_db.p, trace.p, mink.p, S.d_v.p, S.d_ids.p, (int)take, st);
        launch_ond_traceback(S.d_tasks.p, S.d_outs.p, S.d_pool.p, S.d_db.p, trace.p, mink.p, S.d_ops.p, S.d_ids.p, (int)take, st);
        HIP_CHECK(hipStreamSynchronize(st));
        for (size_t i = 0; i < take; i++) {
            const int32_t id = ids[at + i];
            const AlnTask &t = S.tasks[id];
            HIP_CHECK(hipMemcpy(&S.h_outs.p[id], S.d_outs.p + id, sizeof(AlnOut), hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(S.h_ops.p + t.ops_off, S.d_ops.p + t.ops_off,
                                ((uint64_t)(t.ops_cap + 15) / 16 + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
        }
        S.stats.wide_tasks += take;
        at += take;
    }
    (void)n;
}

void hip_align_backend(AlnJob **jobs, size_t n, void *) { DeviceAligner::instance().align_batch(jobs, n); }

}  // namespace ndgpu
