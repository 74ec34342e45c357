// Host runtime around the O(ND) kernels: packs ASCII sequences to the 2-bit pool,
// lays out per-task trace / ops regions in HBM, launches forward + traceback on one
// HIP stream, and hands column-kind streams back to the consensus engine.
//
// Buffers are grow-only and sized for MI355X's 288 GB HBM: a batch keeps every
// task's full trace (max_d rows x 16 B) resident, so there is no second pass and
// no host round trip between the forward sweep and the traceback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <set>
#include <thread>
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

static const bool g_debug_alloc = getenv("NDGPU_DEBUG_ALLOC") != nullptr;  // log every device buffer range (fault triage)
static const bool g_debug_launch = getenv("NDGPU_DEBUG_LAUNCH") != nullptr;  // triage: name + synchronise every launch group
#define NDGPU_DBG(st, ...)                                                                  \
    do {                                                                                    \
        if (g_debug_launch) {                                                               \
            (void)hipStreamSynchronize(st);                                                 \
            fprintf(stderr, "[ndgpu dbg %p] ", (void *)(st));                               \
            fprintf(stderr, __VA_ARGS__);                                                   \
            fprintf(stderr, "\n");                                                          \
            fflush(stderr);                                                                 \
        }                                                                                   \
    } while (0)
static std::mutex g_dbg_mu;  // NDGPU_DEBUG_LAUNCH=2: one device phase at a time over all contexts
static const bool g_debug_exclusive = getenv("NDGPU_DEBUG_LAUNCH") && atoi(getenv("NDGPU_DEBUG_LAUNCH")) >= 2;
static const bool g_debug_nofree = getenv("NDGPU_DEBUG_NOFREE") != nullptr;  // triage: outgrown buffers are leaked, not freed

// Test hook: NDGPU_OOM_ABOVE=bytes makes every device allocation larger than that fail as if the memory were exhausted.
static inline bool oom_injected(size_t bytes) {
    static const size_t lim = getenv("NDGPU_OOM_ABOVE") ? (size_t)strtoull(getenv("NDGPU_OOM_ABOVE"), nullptr, 10) : 0;
    return lim && bytes > lim;
}
// The same for the pinned host arenas: NDGPU_PINNED_OOM_ABOVE=bytes.
static inline bool oom_injected_pinned(size_t bytes) {
    static const size_t lim = getenv("NDGPU_PINNED_OOM_ABOVE") ? (size_t)strtoull(getenv("NDGPU_PINNED_OOM_ABOVE"), nullptr, 10) : 0;
    return lim && bytes > lim;
}

static std::atomic<unsigned long long> g_reserved_bytes{0};  // kept free for the caller's other stage (ndgpu_reserve_device_memory)
static std::atomic<long long> g_dev_bytes{0};
// (re)allocations of device / pinned buffers since the last ndgpu_reset_stats and the wall time the calls took: an allocation in
// the middle of a step stalls every context (section 6 of DESIGN.md), so a steady-state step should show none
static std::atomic<unsigned long long> g_alloc_calls{0}, g_alloc_ns{0}, g_level_calls{0}, g_level_ns{0};
static std::atomic<int> g_leveling{0};  // level_buffers is at work (no kernel in flight): counted apart  // device memory held by the grow-only buffers of all contexts
// The runtime itself allocates on the device (kernel arguments, staging, scratch): a device filled to the last byte makes
// launches and copies fail where nothing can be done about it.  Allocations that would leave less than this are refused
// like an exhausted device (-> DeviceOom -> the sub-batch is halved).
constexpr size_t kDeviceHeadroom = (size_t)3 << 30;
static inline hipError_t guarded_malloc(void **p, size_t bytes) {
    if (oom_injected(bytes)) return hipErrorOutOfMemory;
    size_t free_b = 0, total_b = 0;
    if (bytes > ((size_t)64 << 20) && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < bytes + kDeviceHeadroom)
        return hipErrorOutOfMemory;
    return hipMalloc(p, bytes);
}

// The contexts pull their sub-batches from one queue, so a context may meet a larger sub-batch than it has seen while another
// context has met it before.  A buffer that has to grow therefore grows to the most ANY context has asked of the buffer of that
// name: after the first step or two every context holds the sizes of the largest sub-batch and nothing is (re)allocated in steady
// state -- a hipFree / hipHostMalloc in the middle of a step stalls every queue of the device (seconds, measured).
// A mark is recorded only AFTER an allocation of that size succeeded, and every mark is dropped when a context runs out of device
// memory (clear_high_water, from release_memory): a size that could not be had must not be asked for again by the halves of the
// sub-batch that failed (lib/nextcorrect.c:2254-2261: only a single pile that does not fit is an out-of-memory seed).
static std::mutex g_hw_mu;
static std::unordered_map<std::string, size_t> g_hw;
static size_t high_water(const char *name) {
    if (!name || !*name) return 0;
    std::lock_guard<std::mutex> lock(g_hw_mu);
    auto it = g_hw.find(name);
    return it == g_hw.end() ? 0 : it->second;
}
static void record_high_water(const char *name, size_t bytes) {
    if (!name || !*name) return;
    std::lock_guard<std::mutex> lock(g_hw_mu);
    size_t &v = g_hw[name];
    if (bytes > v) v = bytes;
}
static void clear_high_water() {
    std::lock_guard<std::mutex> lock(g_hw_mu);
    g_hw.clear();
}

struct AllocTimer {  // counts one (re)allocation and its wall time
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~AllocTimer() {
        const unsigned long long ns = (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        if (g_leveling.load()) g_level_calls++, g_level_ns += ns;
        else g_alloc_calls++, g_alloc_ns += ns;
    }
};

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    const char *name = "";
    void reserve(size_t n) {
        if (n <= cap) return;
        const size_t asked = n;
        n = std::max(n, high_water(name) / sizeof(T));  // (what any context has HELD of this buffer; never a size that failed)
        const AllocTimer alloc_timer;
        if (p) {
            if (g_debug_alloc) fprintf(stderr, "[ndgpu alloc] free %s %p\n", name, (void *)p);
            if (!g_debug_nofree) HIP_CHECK(hipFree(p));
            g_dev_bytes -= (long long)(cap * sizeof(T));
        }
        p = nullptr;
        cap = 0;
        // most wanted first: the mark with growth slack, the mark, what this call needs with slack, what this call needs
        size_t want = 0;
        hipError_t rc = hipErrorOutOfMemory;
        const size_t tries[4] = {n + n / 4 + 1024, n + 1024, asked + asked / 4 + 1024, asked + 1024};
        for (int t = 0; t < 4 && rc == hipErrorOutOfMemory; t++) {
            if (t && tries[t] >= want) continue;
            want = tries[t];
            rc = guarded_malloc((void **)&p, want * sizeof(T));
        }
        if (rc == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            p = nullptr;
            throw DeviceOom{want * sizeof(T)};
        }
        HIP_CHECK(rc);
        cap = want;
        record_high_water(name, std::min(want - 1024, n) * sizeof(T));
        g_dev_bytes += (long long)(cap * sizeof(T));
        if (g_debug_alloc)
            fprintf(stderr, "[ndgpu alloc] %s %p .. %p (%zu bytes, asked %zu)\n", name, (void *)p, (void *)((char *)p + want * sizeof(T)),
                    want * sizeof(T), asked * sizeof(T));
    }
    void release() {
        if (p) {
            (void)hipFree(p);
            g_dev_bytes -= (long long)(cap * sizeof(T));
        }
        p = nullptr;
        cap = 0;
    }
    void level() {  // (while the device is idle) up to what any context has asked of the buffer of this name
        const size_t want = high_water(name) / sizeof(T);
        if (want > cap) reserve(want);
    }
    ~DevBuf() {
        if (p) {
            (void)hipFree(p);
            g_dev_bytes -= (long long)(cap * sizeof(T));
        }
    }
};

template <typename T>
struct PinBuf {
    T *p = nullptr;
    size_t cap = 0;
    const char *name = "";
    void reserve(size_t n) {
        if (n <= cap) return;
        const size_t asked = n;
        n = std::max(n, high_water(name) / sizeof(T));
        const AllocTimer alloc_timer;
        if (p && !g_debug_nofree) HIP_CHECK(hipHostFree(p));
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 1024;
        hipError_t rc = oom_injected_pinned(want * sizeof(T)) ? hipErrorOutOfMemory : hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
        if (rc == hipErrorOutOfMemory && asked + 1024 < want) {  // what this call needs, no more
            (void)hipGetLastError();
            want = asked + 1024;
            rc = oom_injected_pinned(want * sizeof(T)) ? hipErrorOutOfMemory : hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
        }
        if (rc == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            p = nullptr;
            throw DeviceOom{want * sizeof(T)};
        }
        HIP_CHECK(rc);
        cap = want;
        record_high_water(name, std::min(want - 1024, n) * sizeof(T));
        if (g_debug_alloc) fprintf(stderr, "[ndgpu alloc] pinned %p .. %p\n", (void *)p, (void *)((char *)p + want * sizeof(T)));
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    void level() {
        const size_t want = high_water(name) / sizeof(T);
        if (want > cap) reserve(want);
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
    const uint32_t *db_pool = nullptr;  // resident read DB of the batch in progress (owned by its ndgpu_db handle)
    hipStream_t stream = nullptr, lat_stream = nullptr;  // lat_stream: reserved compute units (see the constructor)
    hipStream_t stream2 = nullptr;                        // scoring launch of the piles that need the large LDS tables
    hipEvent_t ev_lat0 = nullptr, ev_lat1 = nullptr, ev_fork = nullptr, ev_join = nullptr;
    int reserved_cus = 0;
    std::mutex mu;
    DevBuf<uint32_t> d_pool, d_ops;
    DevBuf<AlnTask> d_tasks;
    DevBuf<AlnOut> d_outs;
    DevBuf<uint64_t> d_trace;
    DevBuf<uint64_t> d_wtrace;  // wide-band alignments (K7w): trace rows and their min_k -- members, not locals of run_wide: a local
    DevBuf<int32_t> d_wmink;    // buffer was a hipMalloc + hipFree per call, and hipFree waits for every stream of the device
    DevBuf<AlnTask> d_wtasks;   // the wide tasks' records of one group, in list order (one upload, not one per task)
    DevBuf<int32_t> d_v, d_ids;
    // segmented traceback (ond_kernels.hip): checkpoint cells / headers the forward kernel leaves, the walkers and what they report
    DevBuf<uint32_t> d_ck_cells;
    DevBuf<uint2> d_ck_hdr;
    DevBuf<TbSeg> d_tbseg;
    DevBuf<TbSegOut> d_tbout;
    std::vector<int32_t> order;
    std::vector<uint32_t> order_cls;
    // main-phase state (alive from run_main to end_batch)
    // Held from begin_batch to end_batch.  Not a plain mutex: with two batch calls in flight (the caller's pipeline: the tail of one
    // call under the main phases of the next) the context must serve the OLDER call's sub-batches first, whichever thread asked first.
    struct OrderedLock {
        std::mutex m;
        std::condition_variable cv;
        bool held = false;
        std::multiset<uint64_t> waiting;
        // reserved: the caller holds a reservation of this order (reserve()): its place in the line stays taken between its batches, so
        // a newer call's thread that is already waiting does not slip in while the older call's thread fetches its next sub-batch
        void lock(uint64_t order = 0, bool reserved = false) {
            std::unique_lock<std::mutex> l(m);
            auto it = reserved ? waiting.end() : waiting.insert(order);
            cv.wait(l, [&] { return !held && *waiting.begin() == order; });
            if (!reserved) waiting.erase(it);
            held = true;
        }
        void reserve(uint64_t order) {
            std::lock_guard<std::mutex> l(m);
            waiting.insert(order);
        }
        void unreserve(uint64_t order) {
            {
                std::lock_guard<std::mutex> l(m);
                auto it = waiting.find(order);
                if (it != waiting.end()) waiting.erase(it);
            }
            cv.notify_all();
        }
        void unlock() {
            {
                std::lock_guard<std::mutex> l(m);
                held = false;
            }
            cv.notify_all();
        }
        bool try_lock() {
            std::lock_guard<std::mutex> l(m);
            if (held || !waiting.empty()) return false;
            held = true;
            return true;
        }
    } batch_mu;
    DevBuf<ReadDev> d_reads;
    DevBuf<PileDev> d_piles;
    DevBuf<uint32_t> d_read_pile, d_acc, d_tags, d_colidx, d_cov /* coverage | insertion counts | longest insertion: one block, one fill */, d_cellbase, d_entbase;
    DevBuf<uint32_t> d_cell_start, d_cell_len, d_cell_bpp, d_cell_blink, d_ent_pp, d_ent_ppp, d_ent_cnt, d_err;
    DevBuf<long long> d_ent_score;
    DevBuf<int32_t> d_cell_best, d_spec, d_fin;  // K10 segments: cell bests, boundary scores (kSegEnts per segment)
    DevBuf<SegSum> d_sums;
    DevBuf<SegItem> d_items;
    DevBuf<uint32_t> d_bt_exit, d_bt_steps, d_bt_entry, d_bt_off;  // best_pp walk by segments
    DevBuf<PathItem> d_path;
    DevBuf<ColBlock> d_blocks;
    DevBuf<RegionDev> d_regions;
    DevBuf<char> d_strpool;
    DevBuf<unsigned long long> d_cursor;
    // low-quality-region rounds (K12)
    DevBuf<LqPileDev> d_lq_piles;
    DevBuf<LqPieceDev> d_lq_pieces;
    DevBuf<uint32_t> d_lq_rec;
    DevBuf<LqJobDev> d_lq_jobs;   // K12a's jobs and their streams: a header per cell row, a word per link
    DevBuf<uint64_t> d_lq_hdr;
    DevBuf<uint32_t> d_lq_lnk;
    DevBuf<char> d_lq_out, d_lq_tmp;   // the piles' characters / the jobs' stretches of the walk (one slot per cell row)
    DevBuf<int32_t> d_lq_bnd;          // K12b's boundary planes: 4 x kLqLinkCap words per job
    std::vector<ReadDev> reads;
    std::vector<PileDev> piles;
    hipEvent_t evs[8] = {nullptr};
    std::vector<hipEvent_t> lq_evs;  // run_lq: K7 / K8a brackets per chunk
    PinBuf<uint32_t> h_ops;
    PinBuf<AlnOut> h_outs;
    std::vector<uint32_t> pool;
    std::vector<AlnTask> tasks;
    RuntimeStats stats;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    size_t trace_budget_bytes = (size_t)48 << 30;
    int host_threads = 1;
    bool k9_full_capacity = getenv("NDGPU_K9_FULL") != nullptr;  // skip the small-capacity first attempt of K9
    uint64_t k9_retries = 0;

    // Host <-> device transfers go through two pinned arenas of the context.  An asynchronous copy from / to pageable
    // memory makes the runtime pin the caller's pages for the duration of the copy; with eight contexts doing that at
    // the same time from neighbouring heap blocks, one context's unpin took a page another context's copy was still
    // using (GPU memory access faults on host heap addresses).  Everything a kernel or a copy engine touches is now
    // either device memory or these arenas.
    PinBuf<char> up, down;
    size_t up_used = 0, down_used = 0;
    struct Pending {
        void *dst;
        size_t off, bytes;
    };
    std::vector<Pending> pending;
    void drain() {  // the stream is idle: hand the downloaded bytes to their owners, recycle both arenas
        for (const Pending &q : pending) memcpy(q.dst, down.p + q.off, q.bytes);
        pending.clear();
        up_used = down_used = 0;
    }
    void sync_drain(hipStream_t st) {
        HIP_CHECK(hipStreamSynchronize(st));
        drain();
    }
    void reserve_down(size_t bytes, hipStream_t st) {  // room for `bytes` of downloads without moving the arena (views stay valid)
        if (down_used + bytes + 4096 <= down.cap) return;
        sync_drain(st);
        down.reserve(bytes + 4096);
    }
    void h2d(void *dst, const void *src, size_t bytes, hipStream_t st) {
        if (!bytes) return;
        const size_t need = (bytes + 255) & ~(size_t)255;
        if (up_used + need > up.cap) {
            sync_drain(st);
            up.reserve(std::max(need, up.cap * 2));
        }
        memcpy(up.p + up_used, src, bytes);
        HIP_CHECK(hipMemcpyAsync(dst, up.p + up_used, bytes, hipMemcpyHostToDevice, st));
        up_used += need;
    }
    // device -> host: the bytes land in the arena; dst_host (if given) receives them at the next drain; the returned
    // pointer is valid from the next stream synchronisation until the next drain / reserve
    void *d2h(void *dst_host, const void *src_dev, size_t bytes, hipStream_t st) {
        const size_t need = (bytes + 255) & ~(size_t)255;
        if (down_used + need > down.cap) {
            sync_drain(st);
            down.reserve(std::max(need, down.cap * 2));
        }
        char *at = down.p + down_used;
        if (bytes) HIP_CHECK(hipMemcpyAsync(at, src_dev, bytes, hipMemcpyDeviceToHost, st));
        if (dst_host && bytes) pending.push_back(Pending{dst_host, down_used, bytes});
        down_used += need;
        return at;
    }
};

static int g_ctx_creating = -1;  // index of the context under construction (guarded by g_ctx_mu)

DeviceAligner::DeviceAligner() : s_(new State) {
    // every context drives its own stream; with the runtime's default of 4 hardware queues streams share a queue and
    // a 0.5 s scoring launch of one context stalls the small kernels of another (must be set before HIP initialises)
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
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
    // A host thread that waits for its stream sleeps instead of spinning: eight driver threads spinning in hipStreamSynchronize are
    // eight cores the host phases of the other contexts do not get -- half of a 16-CPU container (NDGPU_SPIN_SYNC=1: the runtime's default)
    if (!getenv("NDGPU_SPIN_SYNC")) (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
    // Optional CU partition (NDGPU_RESERVED_CUS=n, default off): the first n compute units are kept for the scoring
    // launches of the small sub-batches that hold the longest seeds (lat_stream), every other kernel of every context
    // runs on the rest.  Measured on config 2: the long chains gain nothing (their 3.5 us per column is the chain
    // itself, not interference: 691 ms alone vs 821 ms with 8 contexts resident), so it stays off.
    {
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, s_->device));
        const int n_cu = prop.multiProcessorCount;
        int reserve = 0;
        if (const char *e = getenv("NDGPU_RESERVED_CUS")) reserve = atoi(e);
        if (reserve < 0 || reserve * 2 > n_cu) reserve = 0;
        s_->reserved_cus = reserve;
        if (reserve) {
            const uint32_t words = (uint32_t)((n_cu + 31) / 32);
            std::vector<uint32_t> rest(words, 0), res(words, 0);
            for (int c = 0; c < n_cu; c++) (c < reserve ? res : rest)[c >> 5] |= 1u << (c & 31);
            HIP_CHECK(hipExtStreamCreateWithCUMask(&s_->stream, words, rest.data()));
            HIP_CHECK(hipExtStreamCreateWithCUMask(&s_->lat_stream, words, res.data()));
            HIP_CHECK(hipEventCreateWithFlags(&s_->ev_lat0, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&s_->ev_lat1, hipEventDisableTiming));
        } else {
            // contexts 0..2 receive the sub-batches with the longest seeds (capi.cpp deals sub-batch j to context j
            // in the first round): their kernels go first when the device is oversubscribed
            int least = 0, greatest = 0;
            HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            const int prio = (g_ctx_creating >= 0 && g_ctx_creating < 3 && !getenv("NDGPU_NO_STREAM_PRIO")) ? greatest : least;
            HIP_CHECK(hipStreamCreateWithPriority(&s_->stream, hipStreamNonBlocking, prio));
            HIP_CHECK(hipStreamCreateWithPriority(&s_->stream2, hipStreamNonBlocking, prio));
            HIP_CHECK(hipEventCreateWithFlags(&s_->ev_fork, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&s_->ev_join, hipEventDisableTiming));
        }
    }
#define NDGPU_NAME(x) s_->x.name = #x;
    NDGPU_NAME(h_ops) NDGPU_NAME(h_outs) NDGPU_NAME(up) NDGPU_NAME(down)
    NDGPU_NAME(d_lq_piles) NDGPU_NAME(d_lq_pieces) NDGPU_NAME(d_lq_rec) NDGPU_NAME(d_lq_jobs) NDGPU_NAME(d_lq_hdr) NDGPU_NAME(d_lq_lnk)
    NDGPU_NAME(d_lq_out) NDGPU_NAME(d_lq_tmp) NDGPU_NAME(d_lq_bnd)
    NDGPU_NAME(d_pool) NDGPU_NAME(d_ops) NDGPU_NAME(d_tasks) NDGPU_NAME(d_outs) NDGPU_NAME(d_trace) NDGPU_NAME(d_v) NDGPU_NAME(d_wtrace) NDGPU_NAME(d_wmink) NDGPU_NAME(d_wtasks)
    NDGPU_NAME(d_ck_cells) NDGPU_NAME(d_ck_hdr) NDGPU_NAME(d_tbseg) NDGPU_NAME(d_tbout)
    NDGPU_NAME(d_ids) NDGPU_NAME(d_reads) NDGPU_NAME(d_piles) NDGPU_NAME(d_read_pile) NDGPU_NAME(d_acc) NDGPU_NAME(d_tags)
    NDGPU_NAME(d_colidx) NDGPU_NAME(d_cov) NDGPU_NAME(d_cellbase) NDGPU_NAME(d_entbase)
    NDGPU_NAME(d_cell_start) NDGPU_NAME(d_cell_len) NDGPU_NAME(d_cell_bpp) NDGPU_NAME(d_cell_blink) NDGPU_NAME(d_ent_pp)
    NDGPU_NAME(d_ent_ppp) NDGPU_NAME(d_ent_cnt) NDGPU_NAME(d_err) NDGPU_NAME(d_ent_score) NDGPU_NAME(d_cell_best) NDGPU_NAME(d_spec)
    NDGPU_NAME(d_fin) NDGPU_NAME(d_sums) NDGPU_NAME(d_items) NDGPU_NAME(d_bt_exit) NDGPU_NAME(d_bt_steps) NDGPU_NAME(d_bt_entry)
    NDGPU_NAME(d_bt_off) NDGPU_NAME(d_path) NDGPU_NAME(d_blocks) NDGPU_NAME(d_regions) NDGPU_NAME(d_strpool) NDGPU_NAME(d_cursor)
#undef NDGPU_NAME
    const unsigned ev_flags = getenv("NDGPU_SPIN_SYNC") ? hipEventDefault : hipEventBlockingSync;  // (hipEventSynchronize sleeps, see above)
    HIP_CHECK(hipEventCreateWithFlags(&s_->ev0, ev_flags));
    HIP_CHECK(hipEventCreateWithFlags(&s_->ev1, ev_flags));
    for (auto &e : s_->evs) HIP_CHECK(hipEventCreateWithFlags(&e, ev_flags));
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > ((size_t)8 << 30))
        s_->trace_budget_bytes = std::min<size_t>(free_b / 24, (size_t)8 << 30);
    else s_->trace_budget_bytes = (size_t)2 << 30;
}

DeviceAligner::~DeviceAligner() { delete s_; }

static DeviceAligner *g_ctx[DeviceAligner::kMaxContexts] = {nullptr};
static std::mutex g_ctx_mu;

DeviceAligner &DeviceAligner::context(int i) {
    // intentionally leaked: no HIP calls at exit.  Contexts own a stream + buffer set each, so
    // batches driven from different host threads overlap on the device.
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    i = i < 0 ? 0 : i % kMaxContexts;
    if (!g_ctx[i]) {
        g_ctx_creating = i;
        g_ctx[i] = new DeviceAligner();
        g_ctx_creating = -1;
    }
    return *g_ctx[i];
}

DeviceAligner *DeviceAligner::peek(int i) {
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    return g_ctx[i];
}

DeviceAligner &DeviceAligner::instance() { return context(0); }

RuntimeStats DeviceAligner::total_stats() {
    RuntimeStats t;
    for (int i = 0; i < kMaxContexts; i++) {
        DeviceAligner *cp = peek(i);
        if (!cp) continue;
        DeviceAligner &c = *cp;
        const RuntimeStats &s = c.s_->stats;
        t.tasks += s.tasks; t.wide_tasks += s.wide_tasks; t.cells += s.cells; t.d_steps += s.d_steps;
        t.trace_bits += s.trace_bits; t.trace_words += s.trace_words; t.lq_rounds += s.lq_rounds; t.lq_declined += s.lq_declined; t.lq_ms += s.lq_ms; t.columns += s.columns; t.pool_bases += s.pool_bases; t.seq_bases += s.seq_bases;
        t.max_band = s.max_band > t.max_band ? s.max_band : t.max_band;
        t.forward_launches += s.forward_launches; t.forward_ms += s.forward_ms; t.traceback_ms += s.traceback_ms;
        t.tags_ms += s.tags_ms; t.links_ms += s.links_ms; t.score_ms += s.score_ms; t.extract_ms += s.extract_ms;
        t.piles += s.piles; t.tags += s.tags; t.cells_msa += s.cells_msa; t.path_items += s.path_items;
        t.links += s.links; t.score_launches += s.score_launches; t.backtrack_ms += s.backtrack_ms;
        t.score_segments += s.score_segments; t.score_repairs += s.score_repairs; t.score_slow_piles += s.score_slow_piles;
        t.traceback_launches += s.traceback_launches; t.lq_launches += s.lq_launches; t.lq_columns += s.lq_columns;
        t.lq_aln_columns += s.lq_aln_columns; t.lq_bases += s.lq_bases; t.lq_out += s.lq_out; t.lq_jobs += s.lq_jobs; t.lq_repairs += s.lq_repairs;
        t.tb_tasks += s.tb_tasks; t.tb_walkers += s.tb_walkers; t.tb_fallbacks += s.tb_fallbacks;
    }
    t.allocs = g_alloc_calls.load(), t.alloc_ms = (double)g_alloc_ns.load() * 1e-6;
    t.level_allocs = g_level_calls.load(), t.level_ms = (double)g_level_ns.load() * 1e-6;
    return t;
}

void DeviceAligner::reset_all_stats() {
    g_alloc_calls = 0, g_alloc_ns = 0, g_level_calls = 0, g_level_ns = 0;
    for (int i = 0; i < kMaxContexts; i++)
        if (DeviceAligner *c = peek(i)) c->reset_stats();
}

uint32_t *DeviceAligner::upload_db(const uint32_t *pool_words, size_t n_words, int device) {
    HIP_CHECK(hipSetDevice(device));
    uint32_t *d = nullptr;
    const hipError_t e = hipMalloc((void **)&d, (n_words + kPoolPadWords) * sizeof(uint32_t));
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        return nullptr;
    }
    HIP_CHECK(e);
    if (g_debug_alloc) fprintf(stderr, "[ndgpu alloc] db_pool %p .. %p\n", (void *)d, (void *)(d + n_words + kPoolPadWords));
    HIP_CHECK(hipMemcpy(d, pool_words, n_words * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(d + n_words, 0, kPoolPadWords * sizeof(uint32_t)));
    return d;
}

void DeviceAligner::free_db(uint32_t *dev_pool) {
    if (!dev_pool) return;
    (void)hipDeviceSynchronize();
    (void)hipFree(dev_pool);
}

void DeviceAligner::use_db(const uint32_t *dev_pool) { s_->db_pool = dev_pool; }
int DeviceAligner::device() const { return s_->device; }

void DeviceAligner::release_memory() {
    State &S = *s_;
    std::lock_guard<std::mutex> lock(S.mu);
    (void)hipSetDevice(S.device);
    (void)hipStreamSynchronize(S.stream);
    if (S.stream2) (void)hipStreamSynchronize(S.stream2);
    (void)hipGetLastError();
    S.pending.clear();
    S.up_used = S.down_used = 0;
#define NDGPU_REL(x) S.x.release();
    NDGPU_REL(d_lq_piles) NDGPU_REL(d_lq_pieces) NDGPU_REL(d_lq_rec) NDGPU_REL(d_lq_jobs) NDGPU_REL(d_lq_hdr) NDGPU_REL(d_lq_lnk)
    NDGPU_REL(d_lq_out) NDGPU_REL(d_lq_tmp) NDGPU_REL(d_lq_bnd)
    NDGPU_REL(d_pool) NDGPU_REL(d_ops) NDGPU_REL(d_tasks) NDGPU_REL(d_outs) NDGPU_REL(d_trace) NDGPU_REL(d_v) NDGPU_REL(d_wtrace) NDGPU_REL(d_wmink) NDGPU_REL(d_wtasks)
    NDGPU_REL(d_ck_cells) NDGPU_REL(d_ck_hdr) NDGPU_REL(d_tbseg) NDGPU_REL(d_tbout)
    NDGPU_REL(d_ids) NDGPU_REL(d_reads) NDGPU_REL(d_piles) NDGPU_REL(d_read_pile) NDGPU_REL(d_acc) NDGPU_REL(d_tags)
    NDGPU_REL(d_colidx) NDGPU_REL(d_cov) NDGPU_REL(d_cellbase) NDGPU_REL(d_entbase)
    NDGPU_REL(d_cell_start) NDGPU_REL(d_cell_len) NDGPU_REL(d_cell_bpp) NDGPU_REL(d_cell_blink) NDGPU_REL(d_ent_pp)
    NDGPU_REL(d_ent_ppp) NDGPU_REL(d_ent_cnt) NDGPU_REL(d_err) NDGPU_REL(d_ent_score) NDGPU_REL(d_cell_best) NDGPU_REL(d_spec)
    NDGPU_REL(d_fin) NDGPU_REL(d_sums) NDGPU_REL(d_items) NDGPU_REL(d_bt_exit) NDGPU_REL(d_bt_steps) NDGPU_REL(d_bt_entry)
    NDGPU_REL(d_bt_off) NDGPU_REL(d_path) NDGPU_REL(d_blocks) NDGPU_REL(d_regions) NDGPU_REL(d_strpool) NDGPU_REL(d_cursor)
    NDGPU_REL(h_ops) NDGPU_REL(h_outs) NDGPU_REL(up) NDGPU_REL(down)
#undef NDGPU_REL
}

// After a batch call, with no kernel in flight: every context brings its buffers up to the sizes the largest sub-batch any context
// met has asked for, so that the next call -- whichever context then meets that sub-batch -- allocates nothing in the middle of
// a step.  (A context that is out of device memory keeps what it has.)
void DeviceAligner::level_buffers(int drivers) {
    struct Mark { Mark() { g_leveling++; } ~Mark() { g_leveling--; } } mark;
    for (int c = 0; c < drivers && c < kMaxContexts; c++) {
        DeviceAligner *d = peek(c);
        if (!d) continue;
        State &S = *d->s_;
        // a context with a batch open (another caller's thread) keeps its main-phase buffers alive between run_main and end_batch:
        // growing them here would free live device state.  It is skipped, like release_memory_if_idle skips it.
        std::unique_lock<State::OrderedLock> batch(S.batch_mu, std::try_to_lock);
        if (!batch.owns_lock()) continue;
        std::lock_guard<std::mutex> lock(S.mu);
        (void)hipSetDevice(S.device);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
        // (what the memory plan keeps free stays free: headroom + the caller's reservation for its other stage)
        long long budget = (long long)free_b - (long long)((size_t)10 << 30) - (long long)g_reserved_bytes.load();
        auto lvl = [&](auto &buf, bool device) {
            const size_t esz = sizeof(*buf.p);
            const size_t want = high_water(buf.name) / esz;
            if (want <= buf.cap) return;
            const long long growth = (long long)((want + want / 4 + 1024 - buf.cap) * esz);
            if (device) {
                if (growth > budget) return;
                budget -= growth;
            }
            buf.level();
        };
        try {
#define NDGPU_LVL(x) lvl(S.x, #x[0] == 'd');
            NDGPU_LVL(d_lq_piles) NDGPU_LVL(d_lq_pieces) NDGPU_LVL(d_lq_rec) NDGPU_LVL(d_lq_jobs) NDGPU_LVL(d_lq_hdr) NDGPU_LVL(d_lq_lnk) NDGPU_LVL(d_lq_out) NDGPU_LVL(d_lq_tmp) NDGPU_LVL(d_lq_bnd)
            NDGPU_LVL(d_pool) NDGPU_LVL(d_ops) NDGPU_LVL(d_tasks) NDGPU_LVL(d_outs) NDGPU_LVL(d_trace) NDGPU_LVL(d_v) NDGPU_LVL(d_wtrace) NDGPU_LVL(d_wmink) NDGPU_LVL(d_wtasks)
            NDGPU_LVL(d_ck_cells) NDGPU_LVL(d_ck_hdr) NDGPU_LVL(d_tbseg) NDGPU_LVL(d_tbout)
            NDGPU_LVL(d_ids) NDGPU_LVL(d_reads) NDGPU_LVL(d_piles) NDGPU_LVL(d_read_pile) NDGPU_LVL(d_acc) NDGPU_LVL(d_tags)
            NDGPU_LVL(d_colidx) NDGPU_LVL(d_cov) NDGPU_LVL(d_cellbase) NDGPU_LVL(d_entbase)
            NDGPU_LVL(d_cell_start) NDGPU_LVL(d_cell_len) NDGPU_LVL(d_cell_bpp) NDGPU_LVL(d_cell_blink) NDGPU_LVL(d_ent_pp)
            NDGPU_LVL(d_ent_ppp) NDGPU_LVL(d_ent_cnt) NDGPU_LVL(d_err) NDGPU_LVL(d_cell_best) NDGPU_LVL(d_spec)
            NDGPU_LVL(d_fin) NDGPU_LVL(d_sums) NDGPU_LVL(d_items) NDGPU_LVL(d_bt_exit) NDGPU_LVL(d_bt_steps) NDGPU_LVL(d_bt_entry)
            NDGPU_LVL(d_bt_off) NDGPU_LVL(d_path) NDGPU_LVL(d_blocks) NDGPU_LVL(d_regions) NDGPU_LVL(d_strpool) NDGPU_LVL(d_cursor)
            NDGPU_LVL(h_ops) NDGPU_LVL(h_outs) NDGPU_LVL(up) NDGPU_LVL(down)
#undef NDGPU_LVL
        } catch (const DeviceOom &) {
        }
    }
}

void DeviceAligner::forget_sizes() { clear_high_water(); }

bool DeviceAligner::release_memory_if_idle() {
    if (!s_->batch_mu.try_lock()) return false;
    release_memory();
    s_->batch_mu.unlock();
    return true;
}

// How much of the device the consensus contexts may use, decided at the start of every batch call from what is free NOW
// (the overlap library's cache, the read DB and everything else stay where they are) plus what the contexts already
// hold: per context a trace budget (the forward / traceback chunk size) and a budget of alignment columns per sub-batch
// (~40 bytes of tags, column indexes, link tables and cell tables per column, growth slack included).
void DeviceAligner::reserve_device_memory(uint64_t bytes) { g_reserved_bytes = bytes; }

void DeviceAligner::plan_memory(int drivers, uint64_t *tag_budget) {
    size_t free_b = 0, total_b = 0;
    const int dev = context(0).device();
    (void)hipSetDevice(dev);
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
    const long long held = g_dev_bytes.load();
    long long avail = (long long)free_b + held - (long long)((size_t)10 << 30)   // headroom: runtime, pinned staging
                      - (long long)g_reserved_bytes.load();                      // what the caller's other stage will need again
    if (const char *e = getenv("NDGPU_DEVICE_BUDGET_GB")) avail = std::min<long long>(avail, (long long)(atof(e) * (double)(1ull << 30)));
    if (avail < ((long long)4 << 30)) avail = (long long)4 << 30;
    const long long per_ctx = avail / std::max(1, drivers);
    const size_t trace = (size_t)std::min<long long>((long long)8 << 30, std::max<long long>(per_ctx / 5, (long long)256 << 20));
    const long long cols = (per_ctx - (long long)trace - ((long long)1 << 30)) / 40;
    *tag_budget = (uint64_t)std::min<long long>(900000000ll, std::max<long long>(20000000ll, cols));
    for (int i = 0; i < drivers && i < kMaxContexts; i++) context(i).s_->trace_budget_bytes = trace;
    if (getenv("NDGPU_TRACE"))
        fprintf(stderr, "[ndgpu trace] memory plan: %.1f GB free + %.1f GB held by the contexts -> %d contexts x (%.1f GB trace + %llu M columns)\n",
                free_b / 1073741824.0, held / 1073741824.0, drivers, trace / 1073741824.0, (unsigned long long)(*tag_budget / 1000000));
}

void *DeviceAligner::stream() const { return s_->stream; }
RuntimeStats DeviceAligner::stats() const { return s_->stats; }
void DeviceAligner::reset_stats() { s_->stats = RuntimeStats(); }

static inline uint64_t wall_ns() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
               std::chrono::steady_clock::now().time_since_epoch())
        .count();
}

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

// ---- the traceback in segments (ond_kernels.hip: tb_chase / tb_walk / tb_stitch) ----
// NDGPU_K8_SEG = rows per walker (a power of two; 0: the one-lane kernel everywhere), NDGPU_K8_WARM = rows a walker walks before the rows
// it owns, NDGPU_K8_MINLEN = a launch takes the segmented path when its longest pair has at least that many bases (q + t).  Test hooks
// and A/B knobs; the defaults are what the bench was measured with.
struct TbConfig {
    int cshift = 8, warm = 32;
    bool on = true;
    uint32_t minlen = 8192;
    TbConfig() {
        if (const char *e = getenv("NDGPU_K8_SEG")) {
            const int c = atoi(e);
            on = c >= 4;
            cshift = 2;
            while ((2 << cshift) <= c) cshift++;
        }
        if (const char *e = getenv("NDGPU_K8_WARM")) warm = atoi(e);
        if (warm < 1) warm = 1;
        if (warm > (1 << cshift)) warm = 1 << cshift;
        if (const char *e = getenv("NDGPU_K8_MINLEN")) minlen = (uint32_t)atoll(e);
    }
};
static const TbConfig &tb_config() {
    static const TbConfig c;
    return c;
}
static inline uint32_t tb_ck_slots(int max_d) { return max_d > 0 ? (uint32_t)(max_d - 1) >> tb_config().cshift : 0u; }
// device bytes a task adds to its launch when the launch is walked in segments
static inline uint64_t tb_bytes(int max_d) {
    if (!tb_config().on) return 0;
    const uint64_t k = tb_ck_slots(max_d);
    return k * (kCkptCells * sizeof(uint32_t) + sizeof(uint2)) + (k + 1) * (sizeof(TbSeg) + sizeof(TbSegOut));
}
// tasks [a, b) of one launch: their checkpoint / walker slots (AlnTask::mink_off, seg_off); false: the launch keeps the one-lane kernel
static bool tb_assign(AlnTask *tasks, size_t a, size_t b, uint64_t *ck_slots, uint64_t *seg_slots) {
    const TbConfig &c = tb_config();
    *ck_slots = *seg_slots = 0;
    if (!c.on) return false;
    uint32_t longest = 0;
    for (size_t i = a; i < b; i++) {
        if ((uint32_t)tasks[i].q_len >= (1u << 24)) return false;  // (a checkpoint cell holds x in 24 bits)
        longest = std::max(longest, (uint32_t)tasks[i].q_len + (uint32_t)tasks[i].t_len);
    }
    if (longest < c.minlen) return false;
    uint64_t ck = 0, sg = 0;
    for (size_t i = a; i < b; i++) {
        const uint32_t k = tb_ck_slots(tasks[i].max_d);
        tasks[i].mink_off = ck;
        tasks[i].seg_off = (uint32_t)sg;
        ck += k;
        sg += k + 1;
    }
    if (sg >= (1ull << 31)) return false;
    *ck_slots = ck;
    *seg_slots = sg;
    return true;
}

void DeviceAligner::align_batch(AlnJob **jobs, size_t n) {
    if (n == 0) return;
    std::unique_lock<std::mutex> dbg_lock;
    if (g_debug_exclusive) dbg_lock = std::unique_lock<std::mutex>(g_dbg_mu);
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
            const size_t b = (size_t)md * (kFastRowWords * 8) + (size_t)tb_bytes(md);
            if (take && bytes + b > s_->trace_budget_bytes) break;
            bytes += b;
            take++;
        }
        run_chunk(jobs + done, take);
        done += take;
    }
}

namespace {
template <typename F>
void par_ranges(size_t n, int base_threads, F f) {  // f(begin, end) over contiguous ranges
    if (base_threads <= 1 || n < 4096) {
        f((size_t)0, n);
        return;
    }
    CoreLease lease(base_threads);
    const int threads = lease.n;
    const size_t nt = std::min<size_t>((size_t)threads, (n + 2047) / 2048);
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; t++) {
        const size_t a = n * t / nt, b = n * (t + 1) / nt;
        th.emplace_back([=] { f(a, b); });
    }
    for (auto &x : th) x.join();
}

// ASCII -> 2-bit into a preallocated word range (same coding as pack_append)
bool pack_into(uint32_t *out, const char *s, size_t n) {
    unsigned bad = 0;
    size_t i = 0, w = 0;
    for (; i + 16 <= n; w++, i += 16) {
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
        out[w] = acc;
    }
    return (bad & 0x80u) == 0;
}
}  // namespace

void DeviceAligner::set_host_threads(int n) { s_->host_threads = n < 1 ? 1 : n; }

void DeviceAligner::run_chunk(AlnJob **jobs, size_t n) {
    State &S = *s_;
    const uint64_t tc0 = wall_ns();
    std::vector<uint32_t> &pool = S.pool;
    std::vector<AlnTask> &tasks = S.tasks;
    tasks.assign(n, AlnTask());
    std::vector<uint8_t> bad(n, 0);
    // pass 1 (serial, O(1) per job): offsets of every per-task region
    std::vector<uint64_t> qw(n + 1), tw(n + 1);
    uint64_t trace_words = 0, ops_words = 0, pool_words = 0;
    for (size_t i = 0; i < n; i++) {
        const AlnJob &j = *jobs[i];
        AlnTask &t = tasks[i];
        t.q_len = j.q_len;
        t.t_len = j.t_len;
        qw[i] = pool_words;
        if (j.q_dev < 0) pool_words += ((uint64_t)j.q_len + 15) / 16;
        tw[i] = pool_words;
        if (j.t_dev < 0) pool_words += ((uint64_t)j.t_len + 15) / 16;
        int md, bd;
        limits_for(j.q_len + j.t_len, j.hq, &md, &bd);
        t.max_d = md;
        t.band = bd;
        t.row_words = kFastRowWords;
        t.trace_off = trace_words;
        t.ops_off = ops_words;
        t.ops_cap = (uint32_t)(j.q_len + j.t_len);
        trace_words += (uint64_t)md * kFastRowWords;
        ops_words += (uint64_t)(t.ops_cap + 15) / 16 + 1;
        S.stats.seq_bases += (uint64_t)j.q_len + (uint64_t)j.t_len;
        S.stats.pool_bases += (j.q_dev < 0 ? (uint64_t)j.q_len : 0) + (j.t_dev < 0 ? (uint64_t)j.t_len : 0);
    }
    pool.assign(pool_words, 0);
    // pass 2 (parallel): pack the sequences
    par_ranges(n, S.host_threads, [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            AlnJob &j = *jobs[i];
            AlnTask &t = tasks[i];
            j.status = ALN_NONE;
            j.ops.clear();
            j.q_used = j.t_used = 0;
            if (j.q_dev >= 0) t.q_off = (uint64_t)j.q_dev | kOffDb;
            else {
                t.q_off = qw[i] * 16;
                if (!pack_into(pool.data() + qw[i], j.q, (size_t)j.q_len)) bad[i] = 1;
            }
            if (j.t_dev >= 0) t.t_off = (uint64_t)j.t_dev | kOffDb;
            else {
                t.t_off = tw[i] * 16;
                if (!pack_into(pool.data() + tw[i], j.t, (size_t)j.t_len)) bad[i] = 1;
            }
            if (bad[i]) t.max_d = 0;
        }
    });
    pool.insert(pool.end(), kPoolPadWords, 0u);  // the kernels fetch up to five words from a sequence's last base on

    S.d_pool.reserve(pool.size());
    S.d_tasks.reserve(n);
    S.d_outs.reserve(n);
    S.d_trace.reserve(trace_words + kTracePadWords);
    S.d_ops.reserve(ops_words + 2);
    S.h_ops.reserve(ops_words + 2);
    S.h_outs.reserve(n);
    uint64_t ck_slots = 0, seg_slots = 0;
    const bool seg = tb_assign(tasks.data(), 0, n, &ck_slots, &seg_slots);
    TbArgs tb{};
    if (seg) {
        S.d_ck_cells.reserve(ck_slots * kCkptCells + 1);
        S.d_ck_hdr.reserve(ck_slots + 1);
        S.d_tbseg.reserve(seg_slots);
        S.d_tbout.reserve(seg_slots);
        tb = TbArgs{S.d_ck_cells.p, S.d_ck_hdr.p, S.d_tbseg.p, S.d_tbout.p, (int)seg_slots, tb_config().cshift, tb_config().warm};
    }

    hipStream_t st = S.stream;
    const uint64_t tc1 = wall_ns();
    S.h2d(S.d_pool.p, pool.data(), pool.size() * sizeof(uint32_t), st);
    S.h2d(S.d_tasks.p, tasks.data(), n * sizeof(AlnTask), st);
    HIP_CHECK(hipEventRecord(S.ev0, st));
    NDGPU_DBG(st, "chunk: forward %zu tasks", n);
    if (seg) launch_ond_forward_ckpt(S.d_tasks.p, S.d_outs.p, S.d_pool.p, S.db_pool, S.d_trace.p, S.d_ops.p, tb, (int)n, st, nullptr);
    else launch_ond_forward(S.d_tasks.p, S.d_outs.p, S.d_pool.p, S.db_pool, S.d_trace.p, (int)n, st);
    HIP_CHECK(hipEventRecord(S.ev1, st));
    NDGPU_DBG(st, "chunk: traceback");
    if (seg) launch_ond_traceback_seg(S.d_tasks.p, S.d_outs.p, S.d_pool.p, S.db_pool, S.d_trace.p, S.d_ops.p, tb, (int)n, st);
    else launch_ond_traceback(S.d_tasks.p, S.d_outs.p, S.d_pool.p, S.db_pool, S.d_trace.p, nullptr, S.d_ops.p, nullptr, (int)n, st);
    NDGPU_DBG(st, "chunk: done");
    HIP_CHECK(hipMemcpyAsync(S.h_outs.p, S.d_outs.p, n * sizeof(AlnOut), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(S.h_ops.p, S.d_ops.p, ops_words * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    S.sync_drain(st);
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
    const uint64_t tc2 = wall_ns();

    for (size_t i = 0; i < n; i++) {
        const AlnOut &o = S.h_outs.p[i];
        S.stats.cells += (uint64_t)o.cells;
        S.stats.d_steps += (uint64_t)o.d_steps;
        S.stats.trace_words += (uint64_t)o.trace_end;
        if (o.fin_idx & kTbSeen) {
            S.stats.tb_tasks++;
            S.stats.tb_walkers += (uint64_t)(o.d_final > 0 ? (o.d_final - 1) >> tb_config().cshift : 0) + 1;
            if (o.fin_idx & kTbRefused) S.stats.tb_fallbacks++;
        }
        if ((uint32_t)o.max_band > S.stats.max_band) S.stats.max_band = (uint32_t)o.max_band;
        if (o.status == ST_ALIGNED) {
            S.stats.trace_bits += (uint64_t)o.cells;
            S.stats.columns += (uint64_t)o.n_cols;
        }
        if (bad[i]) {
            static bool warned = false;
            if (!warned) {
                fprintf(stderr, "[ndgpu] sequence with bytes outside [ACGT]: alignment skipped\n");
                warned = true;
            }
        }
    }
    par_ranges(n, S.host_threads, [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            AlnJob &j = *jobs[i];
            const AlnOut &o = S.h_outs.p[i];
            const AlnTask &t = tasks[i];
            if (bad[i]) continue;
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
    });
    g_prof.c_pack += tc1 - tc0, g_prof.c_dev += tc2 - tc1, g_prof.c_decode += wall_ns() - tc2, g_prof.c_jobs += n;
}

void DeviceAligner::run_wide(AlnJob **jobs, size_t n, const std::vector<int32_t> &ids) {
    State &S = *s_;
    const bool ops_to_host = jobs != nullptr;  // the device main phase keeps ops in HBM
    // process in groups bounded by the trace budget; wide rows are band-cap sized
    size_t at = 0;
    DevBuf<uint64_t> &trace = S.d_wtrace;
    DevBuf<int32_t> &mink = S.d_wmink;
    while (at < ids.size()) {
        size_t take = 0;
        uint64_t tw = 0, mr = 0, vw = 0;
        uint32_t max_ring = 0;
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
            max_ring = std::max(max_ring, ring);
            patch.push_back(t);
            take++;
        }
        trace.reserve(tw + 2);
        mink.reserve(mr + 2);
        S.d_v.reserve(vw + 2);
        S.d_ids.reserve(take);
        S.d_wtasks.reserve(take);
        hipStream_t st = S.stream;
        // the group's records go up as ONE table in list order (the kernels of this path read it; the main table keeps the register
        // path's fields, which nobody reads again) and the whole span of results comes back in ONE copy: a task at a time was 2 x 8,307
        // blit kernels per step of the ultra-long read set
        int32_t lo = ids[at], hi = ids[at];
        for (size_t i = 0; i < take; i++) {
            S.tasks[ids[at + i]] = patch[i];
            lo = std::min(lo, ids[at + i]);
            hi = std::max(hi, ids[at + i]);
        }
        S.h2d(S.d_wtasks.p, patch.data(), take * sizeof(AlnTask), st);
        S.h2d(S.d_ids.p, ids.data() + at, take * sizeof(int32_t), st);
        launch_ond_forward_wide(S.d_tasks.p, S.d_outs.p, S.d_pool.p, S.db_pool, trace.p, mink.p, S.d_v.p, S.d_ids.p, (int)take, st, S.d_wtasks.p, max_ring);
        launch_ond_traceback(S.d_tasks.p, S.d_outs.p, S.d_pool.p, S.db_pool, trace.p, mink.p, S.d_ops.p, S.d_ids.p, (int)take, st, nullptr,
                             S.d_wtasks.p);
        HIP_CHECK(hipMemcpyAsync(S.h_outs.p + lo, S.d_outs.p + lo, (size_t)(hi - lo + 1) * sizeof(AlnOut), hipMemcpyDeviceToHost, st));
        S.sync_drain(st);
        for (size_t i = 0; i < take; i++) {
            const int32_t id = ids[at + i];
            const AlnTask &t = S.tasks[id];
            if (ops_to_host)
                HIP_CHECK(hipMemcpy(S.h_ops.p + t.ops_off, S.d_ops.p + t.ops_off,
                                    ((uint64_t)(t.ops_cap + 15) / 16 + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
        }
        S.stats.wide_tasks += take;
        at += take;
    }
    (void)n;
}

// Low-quality-region rounds of a batch of piles on the device: K7 / K8a over every (row, region) alignment, then K12 (lq_links + lq_score:
// linked pseudo-seed, second MSA, DP, walk) -- the column streams stay in HBM, what comes back is each pile's walk string.
// A round the kernel declines (r->ok stays false) is left to the caller's host path.
constexpr uint32_t kLenClasses = 16384;  // 64-base length classes of the longest-first launch order (run_main); the last one holds >= 1 Mb
constexpr uint64_t kLqJobColumns = 192;     // columns of a K12a job (a pile of 3,000 columns is ~15 wavefronts' worth of link building)
constexpr uint64_t kLqMaxColumns = 400000;  // linked pseudo-seed columns K12 takes per pile (see run_lq)
constexpr uint32_t kLqWarmColumns = 64;    // columns a K12b job starts before its own first one (speculative start, checked by the stitch)

void DeviceAligner::run_lq(LqRound **rounds, size_t n) {
    if (n == 0) return;
    State &S = *s_;
    std::unique_lock<std::mutex> dbg_lock;
    if (g_debug_exclusive) dbg_lock = std::unique_lock<std::mutex>(g_dbg_mu);
    std::lock_guard<std::mutex> lock(S.mu);
    HIP_CHECK(hipSetDevice(S.device));
    hipStream_t st = S.stream;
    const uint64_t tc0 = wall_ns();

    // ---- layout: per round its pieces, tasks (one per piece with a job), sequence words (candidates once each, pseudo-seeds once
    //      per region) and output regions
    std::vector<LqPileDev> piles(n);
    std::vector<LqPieceDev> pieces;
    std::vector<AlnTask> &tasks = S.tasks;
    tasks.clear();
    struct Src {
        const uint32_t *words;  // packed already, or
        const char *ascii;
        uint32_t len;
        uint64_t word_off;
    };
    std::vector<Src> srcs;
    std::vector<uint8_t> usable(n, 1);
    uint64_t pool_words = 0, ops_words = 0, cell_rows = 0, out_bytes = 0, hdr_words = 0, lnk_words = 0;
    std::vector<LqJobDev> jobs;
    size_t n_piece_total = 0;
    for (size_t r = 0; r < n; r++) n_piece_total += rounds[r]->pieces.size();
    pieces.reserve(n_piece_total);
    for (size_t r = 0; r < n; r++) {
        LqRound &R = *rounds[r];
        R.ok = false;
        R.lqc.clear();
        LqPileDev &P = piles[r];
        memset(&P, 0, sizeof(P));
        const uint32_t nr = R.n_regions;
        if (nr == 0 || R.pieces.size() != (size_t)nr * 30u) {
            usable[r] = 0;
            continue;
        }
        uint64_t link_len = 1, ins_cap = 0;
        for (uint32_t g = 0; g < nr; g++) link_len += (uint64_t)R.pieces[g].sl + 1;
        // Until round 4 K12b was one wavefront per pile (~1 us per cell row; config 3 had K12 launches of 250 ms) and a pile whose
        // low-quality regions added up to more than 12,000 columns was left to the host path.  Scored job by job (lq_kernels.hip) the
        // chain is as long as a job, not as the pile: the bound is what the packed records can address (below), well above this.
        static const uint64_t max_cols = getenv("NDGPU_K12_MAX_COLUMNS") ? strtoull(getenv("NDGPU_K12_MAX_COLUMNS"), nullptr, 10) : kLqMaxColumns;  // (test hook)
        if (link_len > max_cols) {
            usable[r] = 0;
            continue;
        }
        P.first_piece = (uint32_t)pieces.size();
        P.n_regions = nr;
        P.factor = R.factor;
        P.qv_factor = R.qv_factor;
        std::vector<uint64_t> t_off(nr, ~0ull);  // word offset of every region's pseudo-seed, packed on first use
        for (size_t k = 0; k < R.pieces.size(); k++) {
            const LqRound::Piece &pc = R.pieces[k];
            LqPieceDev d;
            d.task = -1;
            d.sl = pc.sl;
            if (pc.job >= 0) {
                const AlnJob &j = (*R.jobs)[(size_t)pc.job];
                const uint32_t g = (uint32_t)(k % nr);
                AlnTask t;
                memset(&t, 0, sizeof(t));
                t.q_len = j.q_len;
                t.t_len = j.t_len;
                srcs.push_back(Src{j.q_words, j.q, (uint32_t)j.q_len, pool_words});
                t.q_off = pool_words * 16;
                pool_words += ((uint64_t)j.q_len + 15) / 16;
                if (t_off[g] == ~0ull) {
                    t_off[g] = pool_words;
                    srcs.push_back(Src{nullptr, j.t, (uint32_t)j.t_len, pool_words});
                    pool_words += ((uint64_t)j.t_len + 15) / 16;
                }
                t.t_off = t_off[g] * 16;
                int md, bd;
                limits_for(j.q_len + j.t_len, j.hq, &md, &bd);
                t.max_d = md;
                t.band = bd;
                t.row_words = kFastRowWords;
                t.ops_off = ops_words;
                t.ops_cap = (uint32_t)(j.q_len + j.t_len);
                ops_words += (uint64_t)(t.ops_cap + 15) / 16 + 1;
                ins_cap += (uint64_t)j.q_len;
                d.task = (int32_t)tasks.size();
                tasks.push_back(t);
                S.stats.seq_bases += (uint64_t)j.q_len + (uint64_t)j.t_len;
                S.stats.pool_bases += (uint64_t)j.q_len;
            }
            pieces.push_back(d);
        }
        if (link_len + ins_cap >= (1ull << 27) || link_len >= (1ull << 20)) {  // beyond the packed tag's column field / the record's row field
            usable[r] = 0;
            continue;
        }
        P.link_len = (uint32_t)link_len;
        P.out_cap = (uint32_t)(2 * link_len + 64);
        P.cell_off = cell_rows * 6;
        P.out_off = out_bytes;
        out_bytes += P.out_cap;
        // K12a's jobs: runs of regions of about kLqJobColumns columns (each region with the 'N' column in front of it); a job
        // starts only behind a region that has columns (its rows' first tags come from the tail of that region's alignments).
        // Capacities: cell rows = columns + the longest insertion run after every column -- bounded by the candidates' bases, in
        // practice a fraction of the columns: three times the columns are laid out, a job that needs more declines the pile (host
        // path); links <= tags = the alignments' columns (<= q_len + t_len each) + a tag per row of an unaligned region's columns
        // + 30 per 'N'.
        static const uint64_t job_cols = getenv("NDGPU_K12_JOB_COLUMNS") ? strtoull(getenv("NDGPU_K12_JOB_COLUMNS"), nullptr, 10) : kLqJobColumns;  // (test hook: 1 = every region a job)
        P.first_job = (uint32_t)jobs.size();
        {
            uint32_t g = 0, t = 0;
            while (g < nr) {
                LqJobDev jb;
                memset(&jb, 0, sizeof(jb));
                jb.pile = (uint32_t)r, jb.g_a = g, jb.t0 = t;
                uint64_t cols = 0, ins = 0, tags = 0;
                do {
                    const uint32_t sl = R.pieces[g].sl;
                    cols += (uint64_t)sl + 1;
                    tags += 30;
                    for (uint32_t row = 0; row < 30u; row++) {
                        const LqRound::Piece &pc = R.pieces[(size_t)row * nr + g];
                        if (pc.job >= 0) {
                            const AlnJob &j = (*R.jobs)[(size_t)pc.job];
                            ins += (uint64_t)j.q_len;
                            tags += (uint64_t)j.q_len + (uint64_t)j.t_len;
                        } else tags += sl;
                    }
                    t += sl + 1;
                    g++;
                } while (g < nr && (cols < job_cols || R.pieces[g - 1].sl == 0));
                jb.g_b = g;
                if (g == nr) cols += 1, tags += 30, t += 1;  // the closing 'N'
                jb.t1 = t;
                jb.row_cap = (uint32_t)std::min<uint64_t>(cols + ins, 3 * cols + 256);
                jb.lnk_cap = (uint32_t)std::min<uint64_t>(tags, (uint64_t)jb.row_cap * 30u);
                jb.hdr_off = hdr_words, jb.lnk_off = lnk_words;
                hdr_words += jb.row_cap;
                lnk_words += jb.lnk_cap;
                P.row_cap += jb.row_cap;
                jobs.push_back(jb);
            }
        }
        P.n_jobs = (uint32_t)jobs.size() - P.first_job;
        cell_rows += P.row_cap;
    }
    const size_t nt = tasks.size();
    if (nt == 0) {  // nothing K12 takes in this call: every pile goes the host way
        S.stats.lq_rounds += n, S.stats.lq_declined += n;
        return;
    }

    // ---- sequence words (parallel): memcpy of what is packed already, packing of the rest
    std::vector<uint32_t> &pool = S.pool;
    pool.assign(pool_words + kPoolPadWords, 0);  // (the kernels fetch up to five words from a sequence's last base on)
    std::atomic<int> bad_any{0};
    par_ranges(srcs.size(), S.host_threads, [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            const Src &x = srcs[i];
            if (x.words) memcpy(pool.data() + x.word_off, x.words, (((size_t)x.len + 15) / 16) * sizeof(uint32_t));
            else if (!pack_into(pool.data() + x.word_off, x.ascii, x.len)) bad_any = 1;
        }
    });
    if (bad_any.load()) return;  // bytes outside [ACGT]: the host path reports them

    S.d_pool.reserve(pool.size());
    S.d_tasks.reserve(nt);
    S.d_outs.reserve(nt);
    S.d_ops.reserve(ops_words + 2);
    S.d_lq_piles.reserve(n);
    S.d_lq_pieces.reserve(pieces.size());
    S.d_lq_rec.reserve(cell_rows * 6 + 6);
    S.d_lq_jobs.reserve(jobs.size() + 1);
    S.d_lq_hdr.reserve(hdr_words + 1);
    S.d_lq_lnk.reserve(lnk_words + 64);   // (K12b fetches a row's 64 link slots ahead)
    S.d_lq_out.reserve(out_bytes + 1);
    S.d_lq_tmp.reserve(cell_rows + 1);
    S.d_lq_bnd.reserve((jobs.size() + 1) * 4 * (size_t)kLqLinkCap);

    // forward / traceback chunks bounded by the trace budget (the column streams of every chunk stay resident)
    std::vector<size_t> chunk_end;
    uint64_t max_tw = 0;
    {
        uint64_t tw = 0, extra = 0;
        for (size_t i = 0; i < nt; i++) {
            const uint64_t need = (uint64_t)tasks[i].max_d * kFastRowWords, tbb = tb_bytes(tasks[i].max_d);
            if (i && (tw + need) * 8 + extra + tbb > S.trace_budget_bytes) {
                chunk_end.push_back(i);
                max_tw = std::max(max_tw, tw);
                tw = extra = 0;
            }
            tasks[i].trace_off = tw;
            tw += need;
            extra += tbb;
        }
        chunk_end.push_back(nt);
        max_tw = std::max(max_tw, tw);
    }
    S.d_trace.reserve(max_tw + kTracePadWords);
    // the traceback in segments where a launch holds long pairs (regions of several kb): slots per launch
    struct ChunkTb { bool seg; uint64_t slots; };
    std::vector<ChunkTb> chunk_tb;
    {
        uint64_t max_ck = 0, max_sg = 0;
        size_t a = 0;
        for (size_t b : chunk_end) {
            uint64_t ck = 0, sg = 0;
            const bool seg = b > a && tb_assign(tasks.data(), a, b, &ck, &sg);
            max_ck = std::max(max_ck, ck);
            max_sg = std::max(max_sg, sg);
            chunk_tb.push_back(ChunkTb{seg, sg});
            a = b;
        }
        if (max_sg) {
            S.d_ck_cells.reserve(max_ck * kCkptCells + 1);
            S.d_ck_hdr.reserve(max_ck + 1);
            S.d_tbseg.reserve(max_sg);
            S.d_tbout.reserve(max_sg);
        }
    }

    const uint64_t tc1 = wall_ns();
    S.h2d(S.d_pool.p, pool.data(), pool.size() * sizeof(uint32_t), st);
    S.h2d(S.d_tasks.p, tasks.data(), nt * sizeof(AlnTask), st);
    S.h2d(S.d_lq_piles.p, piles.data(), n * sizeof(LqPileDev), st);
    S.h2d(S.d_lq_pieces.p, pieces.data(), pieces.size() * sizeof(LqPieceDev), st);
    if (!jobs.empty()) S.h2d(S.d_lq_jobs.p, jobs.data(), jobs.size() * sizeof(LqJobDev), st);
    // (HIP-event brackets per kernel: K7, K8a per chunk -- read after the round's one synchronisation)
    while (S.lq_evs.size() < 2 * chunk_end.size() + 1) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        S.lq_evs.push_back(e);
    }
    HIP_CHECK(hipEventRecord(S.lq_evs[0], st));
    {
        size_t a = 0, c = 0;
        for (size_t b : chunk_end) {
            NDGPU_DBG(st, "lq: forward / traceback %zu..%zu of %zu tasks", a, b, nt);
            const ChunkTb ctb = chunk_tb[c];
            const TbArgs tb{S.d_ck_cells.p, S.d_ck_hdr.p, S.d_tbseg.p, S.d_tbout.p, (int)ctb.slots, tb_config().cshift, tb_config().warm};
            if (ctb.seg)
                launch_ond_forward_ckpt(S.d_tasks.p + a, S.d_outs.p + a, S.d_pool.p, S.db_pool, S.d_trace.p, S.d_ops.p, tb, (int)(b - a), st, nullptr);
            else launch_ond_forward(S.d_tasks.p + a, S.d_outs.p + a, S.d_pool.p, S.db_pool, S.d_trace.p, (int)(b - a), st);
            HIP_CHECK(hipEventRecord(S.lq_evs[2 * c + 1], st));
            if (ctb.seg)
                launch_ond_traceback_seg(S.d_tasks.p + a, S.d_outs.p + a, S.d_pool.p, S.db_pool, S.d_trace.p, S.d_ops.p, tb, (int)(b - a), st);
            else
                launch_ond_traceback(S.d_tasks.p + a, S.d_outs.p + a, S.d_pool.p, S.db_pool, S.d_trace.p, nullptr, S.d_ops.p, nullptr,
                                     (int)(b - a), st);
            HIP_CHECK(hipEventRecord(S.lq_evs[2 * c + 2], st));
            a = b;
            c++;
        }
    }
    HIP_CHECK(hipEventRecord(S.evs[1], st));
    NDGPU_DBG(st, "lq: msa of %zu piles", n);
    // test hooks: NDGPU_K12_WARM = warm-up columns of a job's speculative start; NDGPU_K12_FORCE=repair: every second job is scored again
    // by the stitch kernel as if its boundary check had failed
    static const uint32_t k12_warm = getenv("NDGPU_K12_WARM") ? (uint32_t)std::max(1, atoi(getenv("NDGPU_K12_WARM"))) : kLqWarmColumns;
    static const uint32_t k12_force = (getenv("NDGPU_K12_FORCE") && !strcmp(getenv("NDGPU_K12_FORCE"), "repair")) ? 2u : 0u;
    launch_lq_msa(S.d_lq_piles.p, S.d_lq_jobs.p, S.d_lq_pieces.p, S.d_tasks.p, S.d_outs.p, S.d_ops.p, S.d_pool.p, S.d_lq_hdr.p, S.d_lq_lnk.p,
                  S.d_lq_rec.p, S.d_lq_bnd.p, S.d_lq_tmp.p, S.d_lq_out.p, (int)n, (int)jobs.size(), k12_warm, k12_force, st);
    HIP_CHECK(hipEventRecord(S.evs[2], st));
    S.h_outs.reserve(nt + 1);
    HIP_CHECK(hipMemcpyAsync(S.h_outs.p, S.d_outs.p, nt * sizeof(AlnOut), hipMemcpyDeviceToHost, st));
    std::vector<char> out(out_bytes + 1);
    S.d2h(piles.data(), S.d_lq_piles.p, n * sizeof(LqPileDev), st);
    if (out_bytes) S.d2h(out.data(), S.d_lq_out.p, out_bytes, st);
    S.sync_drain(st);
    HIP_CHECK(hipGetLastError());
    float ms = 0;
    for (size_t c = 0; c < chunk_end.size(); c++) {
        HIP_CHECK(hipEventElapsedTime(&ms, S.lq_evs[2 * c], S.lq_evs[2 * c + 1]));
        S.stats.forward_ms += ms;
        HIP_CHECK(hipEventElapsedTime(&ms, S.lq_evs[2 * c + 1], S.lq_evs[2 * c + 2]));
        S.stats.traceback_ms += ms;
    }
    S.stats.forward_launches += chunk_end.size();
    S.stats.traceback_launches += chunk_end.size();
    HIP_CHECK(hipEventElapsedTime(&ms, S.evs[1], S.evs[2]));
    S.stats.lq_ms += ms;
    S.stats.lq_launches++;
    S.stats.tasks += nt;
    const uint64_t tc2 = wall_ns();

    for (size_t i = 0; i < nt; i++) {
        const AlnOut &o = S.h_outs.p[i];
        S.stats.cells += (uint64_t)o.cells;
        S.stats.d_steps += (uint64_t)o.d_steps;
        S.stats.trace_words += (uint64_t)o.trace_end;
        if (o.fin_idx & kTbSeen) {
            S.stats.tb_tasks++;
            S.stats.tb_walkers += (uint64_t)(o.d_final > 0 ? (o.d_final - 1) >> tb_config().cshift : 0) + 1;
            if (o.fin_idx & kTbRefused) S.stats.tb_fallbacks++;
        }
        if ((uint32_t)o.max_band > S.stats.max_band) S.stats.max_band = (uint32_t)o.max_band;
        if (o.status == ST_ALIGNED) {
            S.stats.trace_bits += (uint64_t)o.cells;
            S.stats.columns += (uint64_t)o.n_cols;
        }
    }
    for (size_t r = 0; r < n; r++) {
        LqRound &R = *rounds[r];
        S.stats.lq_rounds++;
        const LqPileDev &P = piles[r];
        // (an alignment whose live band left the register path: K12 saw it as unaligned, so its pile goes the host way, where
        // run_chunk reruns it in the wide kernel)
        bool need_wide = false;
        if (usable[r])
            for (uint32_t k = 0; k < 30u * P.n_regions && !need_wide; k++) {
                const int32_t t = pieces[P.first_piece + k].task;
                need_wide = t >= 0 && S.h_outs.p[t].status == ST_NEED_WIDE;
            }
        if (!usable[r] || need_wide || P.err != 0) {
            S.stats.lq_declined++;
            static const bool trace = getenv("NDGPU_TRACE") != nullptr;
            if (trace) fprintf(stderr, "[ndgpu trace] K12 declined a pile (code %u): host path\n", usable[r] ? (need_wide ? 1u : P.err) : 9u);
            continue;
        }
        R.lqc.assign(out.data() + P.out_off, P.out_len);
        R.ok = true;
        S.stats.lq_repairs += P.n_repair;
        S.stats.lq_jobs += P.n_jobs;
        S.stats.lq_columns += P.link_len;
        S.stats.lq_out += P.out_len;
        for (uint32_t k = 0; k < 30u * P.n_regions; k++) {
            const int32_t t = pieces[P.first_piece + k].task;
            if (t < 0) continue;
            S.stats.lq_bases += (uint64_t)tasks[(size_t)t].q_len;
            if (S.h_outs.p[t].status == ST_ALIGNED) S.stats.lq_aln_columns += (uint64_t)S.h_outs.p[t].n_cols;
        }
    }
    g_prof.c_pack += tc1 - tc0, g_prof.c_dev += tc2 - tc1, g_prof.c_decode += wall_ns() - tc2, g_prof.c_jobs += nt;
}

void DeviceAligner::begin_batch(uint64_t order, bool reserved) { s_->batch_mu.lock(order, reserved); }
void DeviceAligner::reserve_batches(uint64_t order) { s_->batch_mu.reserve(order); }
void DeviceAligner::unreserve_batches(uint64_t order) { s_->batch_mu.unreserve(order); }
uint64_t DeviceAligner::next_order() {
    static std::atomic<uint64_t> n{1};
    return n.fetch_add(1);
}
void DeviceAligner::end_batch() { s_->batch_mu.unlock(); }

// Main phase of a batch of piles, entirely on the device:
//   K7 forward -> K8a traceback -> K8s shift scan -> accept -> K8b tags -> column scan
//   -> [one host sync: exact cell / link totals] -> K9 link counting -> K10 scoring + walk.

void DeviceAligner::run_main(MainPile **mp, size_t np) {
    State &S = *s_;
    uint64_t tp0 = wall_ns();
    std::unique_lock<std::mutex> dbg_lock;
    if (g_debug_exclusive) dbg_lock = std::unique_lock<std::mutex>(g_dbg_mu);
    std::lock_guard<std::mutex> lock(S.mu);
    HIP_CHECK(hipSetDevice(S.device));
    hipStream_t st = S.stream;
    std::vector<uint32_t> &pool = S.pool;
    std::vector<AlnTask> &tasks = S.tasks;
    std::vector<ReadDev> &reads = S.reads;
    std::vector<PileDev> &piles = S.piles;
    pool.clear();
    tasks.clear();
    reads.clear();
    piles.assign(np, PileDev());
    std::vector<uint32_t> read_pile;
    std::vector<uint8_t> bad_pile(np, 0);
    uint64_t ops_words = 0, tag_slots = 0, colidx_slots = 0, col_slots = 0, acc_slots = 0;
    for (size_t p = 0; p < np; p++) {
        MainPile &M = *mp[p];
        PileDev &P = piles[p];
        M.path.clear();
        M.slot = (int)p;
        memset(&P, 0, sizeof(P));
        P.seed_len = M.aln_end[0] + 1;
        P.n_reads = M.n;
        P.first_read = (uint32_t)reads.size();
        P.min_len_aln = M.min_len_aln;
        P.max_cov_aln = M.max_cov_aln;
        P.factor = M.factor;
        P.col_off = col_slots;
        col_slots += (uint64_t)P.seed_len + 1;
        P.acc_off = acc_slots;
        acc_slots += M.n;
        if (M.dev_off) P.seed_off = (uint64_t)M.dev_off[0] | kOffDb;
        else {
            P.seed_off = (uint64_t)pool.size() * 16;
            if (!pack_append(pool, M.seqs[0], M.seq_len[0])) bad_pile[p] = 1;
            S.stats.pool_bases += M.seq_len[0];
        }
        for (unsigned i = 0; i < M.n; i++) {
            ReadDev R;
            memset(&R, 0, sizeof(R));
            R.aln_start = M.aln_start[i];
            R.aln_end = M.aln_end[i];
            uint64_t tag_cap, ci_cap;
            if (i == 0) {
                R.task = -1;
                tag_cap = ci_cap = P.seed_len;
            } else {
                AlnTask t;
                memset(&t, 0, sizeof(t));
                t.q_len = (int32_t)M.seq_len[i];
                t.t_len = (int32_t)(M.aln_end[i] - M.aln_start[i] + 1);
                if (M.dev_off) t.q_off = (uint64_t)M.dev_off[i] | kOffDb;
                else {
                    t.q_off = (uint64_t)pool.size() * 16;
                    if (!pack_append(pool, M.seqs[i], M.seq_len[i])) bad_pile[p] = 1;
                    S.stats.pool_bases += M.seq_len[i];
                }
                t.t_off = P.seed_off + M.aln_start[i];
                int md, bd;
                limits_for(t.q_len + t.t_len, M.hq, &md, &bd);
                t.max_d = md;
                t.band = bd;
                t.row_words = kFastRowWords;
                t.ops_off = ops_words;
                t.ops_cap = (uint32_t)(t.q_len + t.t_len);
                ops_words += (uint64_t)(t.ops_cap + 15) / 16 + 1;
                S.stats.seq_bases += (uint64_t)t.q_len + (uint64_t)t.t_len;
                R.task = (int32_t)tasks.size();
                tasks.push_back(t);
                tag_cap = t.ops_cap;
                ci_cap = (uint64_t)t.t_len;
            }
            R.tag_off = tag_slots;
            tag_slots += tag_cap;
            R.colidx_off = colidx_slots;
            colidx_slots += ci_cap + 1;
            reads.push_back(R);
            read_pile.push_back((uint32_t)p);
        }
        if (bad_pile[p]) {  // bytes outside [ACGT]: nothing of this pile is aligned
            fprintf(stderr, "[ndgpu] pile with bytes outside [ACGT]: reported as uncorrectable\n");
            for (uint32_t r = P.first_read + 1; r < reads.size(); r++) tasks[reads[r].task].max_d = 0;
            P.min_len_aln = 0xffffffffu;
        }
    }
    pool.insert(pool.end(), kPoolPadWords, 0u);
    const size_t nt = tasks.size(), nr = reads.size();

    // forward/traceback chunks bounded by the trace budget
    std::vector<size_t> chunk_end;
    {
        uint64_t tw = 0, extra = 0;
        for (size_t i = 0; i < nt; i++) {
            const uint64_t need = (uint64_t)tasks[i].max_d * (kFastRowWords * 8), tbb = tb_bytes(tasks[i].max_d);
            if (i && (tw * 8 + extra + need + tbb) > S.trace_budget_bytes) {
                chunk_end.push_back(i);
                tw = extra = 0;
            }
            tasks[i].trace_off = tw;
            tw += (uint64_t)tasks[i].max_d * kFastRowWords;
            extra += tbb;
        }
        chunk_end.push_back(nt);
    }
    uint64_t max_tw = 0;
    // the traceback in segments: checkpoint and walker slots per launch
    struct ChunkTb { bool seg; uint64_t slots; };
    std::vector<ChunkTb> chunk_tb;
    uint64_t max_ck = 0, max_sg = 0;
    {
        size_t a = 0;
        for (size_t b : chunk_end) {
            uint64_t ck = 0, sg = 0;
            bool seg = false;
            if (b > a) {
                const AlnTask &l = tasks[b - 1];
                max_tw = std::max<uint64_t>(max_tw, l.trace_off + (uint64_t)l.max_d * kFastRowWords);
                seg = tb_assign(tasks.data(), a, b, &ck, &sg);
                max_ck = std::max(max_ck, ck);
                max_sg = std::max(max_sg, sg);
            }
            chunk_tb.push_back(ChunkTb{seg, sg});
            a = b;
        }
    }
    if (max_sg) {
        S.d_ck_cells.reserve(max_ck * kCkptCells + 1);
        S.d_ck_hdr.reserve(max_ck + 1);
        S.d_tbseg.reserve(max_sg);
        S.d_tbout.reserve(max_sg);
    }

    S.d_pool.reserve(pool.size());
    S.d_tasks.reserve(nt + 1);
    S.d_outs.reserve(nt + 1);
    S.h_outs.reserve(nt + 1);
    S.d_trace.reserve(max_tw + kTracePadWords);
    S.d_ops.reserve(ops_words + 2);
    S.d_reads.reserve(nr);
    S.d_piles.reserve(np);
    S.d_read_pile.reserve(nr);
    S.d_acc.reserve(acc_slots + 1);
    S.d_tags.reserve(tag_slots + 9);  // (K9 reads 32-byte windows: up to 7 tags past a read's last one)
    S.d_colidx.reserve(colidx_slots + 1);
    S.d_cov.reserve(3 * (col_slots + 1));  // per column: coverage, insertion count, longest insertion -- three arrays in one block
    S.d_cellbase.reserve(col_slots + 1);
    S.d_entbase.reserve(col_slots + 1);
    S.d_err.reserve(4);

    uint64_t tp1 = wall_ns();
    g_prof.m_prep += tp1 - tp0;
    S.h2d(S.d_pool.p, pool.data(), pool.size() * sizeof(uint32_t), st);
    if (nt) S.h2d(S.d_tasks.p, tasks.data(), nt * sizeof(AlnTask), st);
    S.h2d(S.d_reads.p, reads.data(), nr * sizeof(ReadDev), st);
    S.h2d(S.d_piles.p, piles.data(), np * sizeof(PileDev), st);
    S.h2d(S.d_read_pile.p, read_pile.data(), nr * sizeof(uint32_t), st);
    uint32_t *const d_cov = S.d_cov.p, *const d_inscnt = d_cov + (col_slots + 1), *const d_insmax = d_inscnt + (col_slots + 1);
    HIP_CHECK(hipMemsetAsync(d_cov, 0, 3 * (col_slots + 1) * sizeof(uint32_t), st));  // (one fill for the three)
    HIP_CHECK(hipMemsetAsync(S.d_err.p, 0, 4 * sizeof(uint32_t), st));

    {
        size_t a = 0, ci = 0;
        for (size_t b : chunk_end) {
            const ChunkTb ctb = chunk_tb[ci++];
            if (b > a) {
                NDGPU_DBG(st, "main: forward %zu..%zu of %zu tasks, %zu piles", a, b, nt, np);
                const int32_t *order = nullptr;
                static const bool lpt = !getenv("NDGPU_K7_NO_ORDER");
                if (lpt && b - a > 64) {
                    // longest alignments first (their chains bound the launch): a counting sort over 64-base length classes
                    // -- a sub-batch holds up to a million tasks and this runs on the context's critical path
                    std::vector<int32_t> &ord = S.order;
                    std::vector<uint32_t> &cls = S.order_cls;
                    const size_t m = b - a;
                    ord.resize(m);
                    cls.assign(kLenClasses + 1, 0);
                    auto cls_of = [&](size_t i) {
                        const uint32_t c = ((uint32_t)tasks[a + i].q_len + (uint32_t)tasks[a + i].t_len) >> 6;
                        return (kLenClasses - 1) - std::min<uint32_t>(c, kLenClasses - 1);  // class 0 = the longest
                    };
                    for (size_t i = 0; i < m; i++) cls[cls_of(i) + 1]++;
                    for (uint32_t c = 0; c < kLenClasses; c++) cls[c + 1] += cls[c];
                    for (size_t i = 0; i < m; i++) ord[cls[cls_of(i)]++] = (int32_t)i;
                    S.d_ids.reserve(m);
                    S.h2d(S.d_ids.p, ord.data(), m * sizeof(int32_t), st);
                    order = S.d_ids.p;
                }
                const TbArgs tb{S.d_ck_cells.p, S.d_ck_hdr.p, S.d_tbseg.p, S.d_tbout.p, (int)ctb.slots, tb_config().cshift, tb_config().warm};
                HIP_CHECK(hipEventRecord(S.evs[0], st));
                if (ctb.seg)
                    launch_ond_forward_ckpt(S.d_tasks.p + a, S.d_outs.p + a, S.d_pool.p, S.db_pool, S.d_trace.p, S.d_ops.p, tb, (int)(b - a), st, order);
                else launch_ond_forward(S.d_tasks.p + a, S.d_outs.p + a, S.d_pool.p, S.db_pool, S.d_trace.p, (int)(b - a), st, order);
                HIP_CHECK(hipEventRecord(S.evs[1], st));
                NDGPU_DBG(st, "main: traceback");
                // (K8a stays in table order: measured in round 5, the 64 walks of a wavefront ordered longest first like K7's --
                // equal lengths, long walks first -- cost 605 ms of traceback per step against 496: the lanes of a wavefront in pile
                // order walk neighbouring windows of one seed and share its cache lines; NDGPU_K8_ORDER=1 switches the order on)
                static const bool k8_order = getenv("NDGPU_K8_ORDER") != nullptr;
                if (ctb.seg)
                    launch_ond_traceback_seg(S.d_tasks.p + a, S.d_outs.p + a, S.d_pool.p, S.db_pool, S.d_trace.p, S.d_ops.p, tb, (int)(b - a), st);
                else
                    launch_ond_traceback(S.d_tasks.p + a, S.d_outs.p + a, S.d_pool.p, S.db_pool, S.d_trace.p, nullptr,
                                         S.d_ops.p, nullptr, (int)(b - a), st, k8_order ? order : nullptr);
                NDGPU_DBG(st, "main: traceback done");
                HIP_CHECK(hipEventRecord(S.evs[2], st));
                HIP_CHECK(hipEventSynchronize(S.evs[2]));
                float ms = 0;
                HIP_CHECK(hipEventElapsedTime(&ms, S.evs[0], S.evs[1]));
                S.stats.forward_ms += ms;
                S.stats.forward_launches++;
                HIP_CHECK(hipEventElapsedTime(&ms, S.evs[1], S.evs[2]));
                S.stats.traceback_ms += ms;
                S.stats.traceback_launches++;
            }
            a = b;
        }
    }
    if (nt) {
        HIP_CHECK(hipMemcpyAsync(S.h_outs.p, S.d_outs.p, nt * sizeof(AlnOut), hipMemcpyDeviceToHost, st));
        S.sync_drain(st);
        std::vector<int32_t> wide;
        for (size_t i = 0; i < nt; i++) {
            const AlnOut &o = S.h_outs.p[i];
            S.stats.cells += (uint64_t)o.cells;
            S.stats.d_steps += (uint64_t)o.d_steps;
            S.stats.trace_words += (uint64_t)o.trace_end;
        if (o.fin_idx & kTbSeen) {
            S.stats.tb_tasks++;
            S.stats.tb_walkers += (uint64_t)(o.d_final > 0 ? (o.d_final - 1) >> tb_config().cshift : 0) + 1;
            if (o.fin_idx & kTbRefused) S.stats.tb_fallbacks++;
        }
            if ((uint32_t)o.max_band > S.stats.max_band) S.stats.max_band = (uint32_t)o.max_band;
            if (o.status == ST_NEED_WIDE) wide.push_back((int32_t)i);
            if (o.status == ST_ALIGNED) {   // (K8a's output: 2-bit column kinds -- the term bench.py's roofline prices it with)
                S.stats.trace_bits += (uint64_t)o.cells;
                S.stats.columns += (uint64_t)o.n_cols;
            }
        }
        if (!wide.empty()) run_wide(nullptr, nt, wide);
        S.stats.tasks += nt;
    }

    uint64_t tp2 = wall_ns();
    g_prof.m_aln += tp2 - tp1;
    HIP_CHECK(hipEventRecord(S.evs[0], st));
    NDGPU_DBG(st, "main: shift_scan");
    launch_shift_scan(S.d_tasks.p, S.d_outs.p, S.d_ops.p, S.d_reads.p, (int)nr, st);
    NDGPU_DBG(st, "main: pile_accept");
    launch_pile_accept(S.d_piles.p, S.d_reads.p, S.d_acc.p, d_cov, (int)np, st);
    NDGPU_DBG(st, "main: make_tags");
    launch_make_tags(S.d_piles.p, S.d_reads.p, S.d_tasks.p, S.d_ops.p, S.d_pool.p, S.db_pool, S.d_read_pile.p,
                     S.d_tags.p, S.d_colidx.p, d_inscnt, d_insmax, (int)nr, st);
    NDGPU_DBG(st, "main: col_scan");
    launch_col_scan(S.d_piles.p, d_cov, d_inscnt, d_insmax, S.d_cellbase.p, S.d_entbase.p, (int)np, st);
    NDGPU_DBG(st, "main: col_scan done");
    HIP_CHECK(hipEventRecord(S.evs[1], st));
    S.d2h(piles.data(), S.d_piles.p, np * sizeof(PileDev), st);
    S.sync_drain(st);

    uint64_t tp3 = wall_ns();
    g_prof.m_tags += tp3 - tp2;
    uint64_t cells = 0, ents = 0, paths = 0;
    std::vector<ColBlock> blocks;
    // scoring segments (K10): `seg_len` columns each, the last one takes the remainder; the two table tiers get a work
    // list each.  Test hooks: NDGPU_K10_FORCE = large (every pile through the large tables) | slow (every pile through the
    // int64 HBM-resident kernel) | seq (one segment per pile: no speculation) | repair (every second segment is scored
    // again by the stitch kernel as if its check had failed); NDGPU_K10_SEG / NDGPU_K10_WARM / NDGPU_K10_GUARD set the
    // segment length, the warm-up length and the raw-score guard.
    struct K10Cfg {
        uint32_t seg_len = 1024, warm = 128, force_repair = 0;
        int32_t guard = 1 << 30;
        bool large = false, slow = false;
        K10Cfg() {
            if (const char *e = getenv("NDGPU_K10_SEG")) seg_len = (uint32_t)std::max(16, atoi(e));
            if (const char *e = getenv("NDGPU_K10_WARM")) warm = (uint32_t)std::max(1, atoi(e));
            if (const char *e = getenv("NDGPU_K10_GUARD")) guard = atoi(e);
            if (const char *e = getenv("NDGPU_K10_FORCE")) {
                large = !strcmp(e, "large"), slow = !strcmp(e, "slow");
                if (!strcmp(e, "seq")) seg_len = 0x7fffffffu;
                if (!strcmp(e, "repair")) force_repair = 2;
            }
            if (warm >= seg_len) warm = seg_len - 1;
        }
    };
    static const K10Cfg k10;
    std::vector<SegItem> items_small, items_large, items_slow;  // (slow: piles of the int64 kernel; only the walk uses them)
    uint32_t n_segs = 0;
    for (size_t p = 0; p < np; p++) {
        PileDev &P = piles[p];
        if (k10.large) P.err = 3;
        if (k10.slow) P.err = 2;
        P.n_seg = std::max<uint32_t>(1u, (uint32_t)(((uint64_t)P.seed_len + k10.seg_len / 2) / k10.seg_len));
        P.seg_off = n_segs;
        n_segs += P.n_seg;
        P.n_repair = 0;
        P.tier = P.err == 3 ? 1u : 0u;
        std::vector<SegItem> &dst = P.err == 2 ? items_slow : P.err == 3 ? items_large : items_small;
        for (uint32_t g = 0; g < P.n_seg; g++) dst.push_back(SegItem{(uint32_t)p, g});
    }
    for (size_t p = 0; p < np; p++) {
        PileDev &P = piles[p];
        P.cell_off = cells;
        P.ent_off = ents;
        P.path_off = paths;
        cells += P.n_cells;
        ents += P.n_tags;
        paths += P.n_cells / 6 + 1;
        for (uint32_t c = 0; c < P.seed_len; c += kColBlock) blocks.push_back(ColBlock{(uint32_t)p, c});
        S.stats.tags += P.n_tags;
        S.stats.cells_msa += P.n_cells;
    }
    S.stats.piles += np;
    S.d_cell_start.reserve(cells + 1);
    S.d_cell_len.reserve(cells + 1);
    S.d_cell_bpp.reserve(cells + 1);
    S.d_cell_blink.reserve(cells + 1);
    S.d_ent_pp.reserve(ents + 1);
    S.d_ent_ppp.reserve(ents + 1);
    S.d_ent_cnt.reserve(ents + 1);
    // (d_ent_score, 8 bytes per link, belongs to the int64 kernel: allocated only when a pile needs it -- see the rescue pass)
    S.d_path.reserve(paths + 1);
    S.d_blocks.reserve(blocks.size() + 1);
    S.d_cell_best.reserve(cells + 1);
    S.d_sums.reserve(n_segs + 1);
    S.d_spec.reserve((size_t)n_segs * kSegEnts + 1);
    S.d_fin.reserve((size_t)n_segs * kSegEnts + 1);
    const size_t n_items_all = items_small.size() + items_large.size() + items_slow.size();
    S.d_items.reserve(n_items_all + 1);
    S.d_bt_exit.reserve((size_t)n_segs * kBtSlots + 1);
    S.d_bt_steps.reserve((size_t)n_segs * kBtSlots + 1);
    S.d_bt_entry.reserve(n_segs + 1);
    S.d_bt_off.reserve(n_segs + 1);
    S.h2d(S.d_blocks.p, blocks.data(), blocks.size() * sizeof(ColBlock), st);
    S.h2d(S.d_items.p, items_small.data(), items_small.size() * sizeof(SegItem), st);
    S.h2d(S.d_items.p + items_small.size(), items_large.data(), items_large.size() * sizeof(SegItem), st);
    S.h2d(S.d_items.p + items_small.size() + items_large.size(), items_slow.data(), items_slow.size() * sizeof(SegItem), st);
    const PathItem *hpath = nullptr;  // view into the download arena
    std::vector<PileDev> piles_out(np);
    uint32_t herr[4] = {0, 0, 0, 0};
    // attempt 0 counts links with the small LDS lists; a cell with more distinct links than they hold raises err[0] and the
    // sub-batch is counted and scored again with the full capacity (everything the kernels write is rewritten)
    K10Args ka;
    ka.piles = S.d_piles.p;
    ka.coverage = d_cov, ka.max_size = d_insmax, ka.cell_base = S.d_cellbase.p, ka.ent_base = S.d_entbase.p;
    ka.cell_start = S.d_cell_start.p, ka.cell_len = S.d_cell_len.p;
    ka.ent_pp = S.d_ent_pp.p, ka.ent_ppp = S.d_ent_ppp.p, ka.ent_cnt = S.d_ent_cnt.p;
    ka.cell_best_pp = S.d_cell_bpp.p, ka.cell_best_link = S.d_cell_blink.p, ka.cell_best = S.d_cell_best.p;
    ka.sums = S.d_sums.p, ka.spec = S.d_spec.p, ka.fin = S.d_fin.p;
    ka.seg_len = k10.seg_len, ka.warm = k10.warm, ka.guard = k10.guard, ka.force_repair = k10.force_repair;
    for (int attempt = 0; attempt < 2; attempt++) {
        S.reserve_down(np * sizeof(PileDev) + paths * sizeof(PathItem) + 1024, st);
        S.h2d(S.d_piles.p, piles.data(), np * sizeof(PileDev), st);
        if (attempt) HIP_CHECK(hipMemsetAsync(S.d_err.p, 0, 4 * sizeof(uint32_t), st));
        HIP_CHECK(hipEventRecord(S.evs[2], st));
        NDGPU_DBG(st, "main: count_links %zu blocks, cells %llu ents %llu segs %u", blocks.size(), (unsigned long long)cells,
                  (unsigned long long)ents, n_segs);
        launch_count_links(S.d_piles.p, S.d_reads.p, S.d_acc.p, S.d_blocks.p, S.d_tags.p, S.d_colidx.p, d_insmax,
                           S.d_cellbase.p, S.d_entbase.p, S.d_cell_start.p, S.d_cell_len.p, S.d_ent_pp.p, S.d_ent_ppp.p,
                           S.d_ent_cnt.p, S.d_err.p, (int)blocks.size(), attempt != 0 || S.k9_full_capacity, st);
        HIP_CHECK(hipEventRecord(S.evs[3], st));
        NDGPU_DBG(st, "main: score + walk");
        // a sub-batch small enough for the reserved compute units (4 two-wave blocks each) scores there
        const bool on_reserved = S.lat_stream && np <= (size_t)S.reserved_cus * 2;
        hipStream_t sst = on_reserved ? S.lat_stream : st;
        if (on_reserved) {
            HIP_CHECK(hipEventRecord(S.ev_lat0, st));
            HIP_CHECK(hipStreamWaitEvent(sst, S.ev_lat0, 0));
            HIP_CHECK(hipEventRecord(S.evs[3], sst));
        }
        launch_score_backtrack(ka, S.d_items.p, (int)items_small.size(), S.d_items.p + items_small.size(), (int)items_large.size(),
                               S.d_items.p, (int)n_items_all, S.d_ent_score.cap >= ents + 1 ? S.d_ent_score.p : nullptr, false,
                               S.d_path.p, S.d_bt_exit.p, S.d_bt_steps.p,
                               S.d_bt_entry.p, S.d_bt_off.p, (int)np, sst, S.evs[7], on_reserved ? nullptr : S.stream2, S.ev_fork,
                               S.ev_join);
        HIP_CHECK(hipEventRecord(S.evs[4], sst));
        NDGPU_DBG(st, "main: score + walk done");
        if (on_reserved) {
            HIP_CHECK(hipEventRecord(S.ev_lat1, sst));
            HIP_CHECK(hipStreamWaitEvent(st, S.ev_lat1, 0));
        }
        const void *v_piles = S.d2h(nullptr, S.d_piles.p, np * sizeof(PileDev), st);  // (re-taken by the rescue pass)
        hpath = (const PathItem *)S.d2h(nullptr, S.d_path.p, paths * sizeof(PathItem), st);
        const void *v_err = S.d2h(nullptr, S.d_err.p, sizeof(herr), st);
        HIP_CHECK(hipStreamSynchronize(st));
        HIP_CHECK(hipGetLastError());
        memcpy(piles_out.data(), v_piles, np * sizeof(PileDev));
        memcpy(herr, v_err, sizeof(herr));
        // rescue pass: piles the segment kernels handed to the int64 HBM-resident kernel (err == 2: a column wider than the
        // LDS tables, raw scores out of the int32 working range) when its 8-byte-per-link score array was not there yet
        bool rescue = false;
        for (size_t p = 0; p < np; p++) rescue = rescue || piles_out[p].err == 2;
        if (rescue) {
            S.d_ent_score.reserve(ents + 1);
            launch_score_backtrack(ka, nullptr, 0, nullptr, 0, S.d_items.p, (int)n_items_all, S.d_ent_score.p, true, S.d_path.p,
                                   S.d_bt_exit.p, S.d_bt_steps.p, S.d_bt_entry.p, S.d_bt_off.p, (int)np, st, nullptr, nullptr, nullptr,
                                   nullptr);
            S.reserve_down(np * sizeof(PileDev) + paths * sizeof(PathItem) + 1024, st);
            v_piles = S.d2h(nullptr, S.d_piles.p, np * sizeof(PileDev), st);
            hpath = (const PathItem *)S.d2h(nullptr, S.d_path.p, paths * sizeof(PathItem), st);
            HIP_CHECK(hipStreamSynchronize(st));
            HIP_CHECK(hipGetLastError());
            memcpy(piles_out.data(), v_piles, np * sizeof(PileDev));
        }
        static const bool force_retry = getenv("NDGPU_K9_FORCE_RETRY") != nullptr;  // test hook: take the overflow path
        if ((!herr[0] && !force_retry) || attempt || S.k9_full_capacity) break;
        S.k9_retries++;
        if (getenv("NDGPU_TRACE")) fprintf(stderr, "[ndgpu trace] K9: a cell holds more than %d distinct links, sub-batch repeated with %d\n", kLinkCapSmall, kLinkCap);
    }
    piles.swap(piles_out);
    uint64_t tp4 = wall_ns();
    g_prof.m_msa += tp4 - tp3;
    if (herr[0]) {
        fprintf(stderr, "[ndgpu] FATAL: more than %d distinct links in one MSA cell (device capacity)\n", kLinkCap);
        abort();
    }
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, S.evs[0], S.evs[1]));
    S.stats.tags_ms += ms;
    HIP_CHECK(hipEventElapsedTime(&ms, S.evs[2], S.evs[3]));
    S.stats.links_ms += ms;
    HIP_CHECK(hipEventElapsedTime(&ms, S.evs[3], S.evs[7]));
    S.stats.score_ms += ms;
    S.stats.score_launches++;
    HIP_CHECK(hipEventElapsedTime(&ms, S.evs[7], S.evs[4]));
    S.stats.backtrack_ms += ms;
    {
        static const bool trace = getenv("NDGPU_TRACE") != nullptr;
        if (trace) {
            float t_links = 0, t_score = 0, t_back = 0;
            (void)hipEventElapsedTime(&t_links, S.evs[2], S.evs[3]);
            (void)hipEventElapsedTime(&t_score, S.evs[3], S.evs[7]);
            (void)hipEventElapsedTime(&t_back, S.evs[7], S.evs[4]);
            uint32_t longest = 0;
            for (size_t p = 0; p < np; p++) longest = std::max(longest, piles[p].seed_len);
            uint32_t rep = 0, slow = 0;
            for (size_t p = 0; p < np; p++) {
                if (piles[p].n_repair == 0xffffffffu) slow++;
                else rep += piles[p].n_repair;
            }
            fprintf(stderr, "[ndgpu trace] run_main %zu piles longest %u | host prep %.1f align %.1f tags %.1f msa %.1f ms | K9 %.1f K10 %.1f backtrack %.1f ms | %u segments, %u repaired, %u piles through the int64 kernel\n",
                    np, longest, (tp1 - tp0) * 1e-6, (tp2 - tp1) * 1e-6, (tp3 - tp2) * 1e-6, (tp4 - tp3) * 1e-6, t_links, t_score, t_back, n_segs, rep, slow);
        }
    }
    for (size_t p = 0; p < np; p++) {
        const PileDev &P = piles[p];
        mp[p]->n_aligned = P.n_acc;
        if (bad_pile[p]) continue;
        S.stats.path_items += P.path_len;
        S.stats.links += P.n_links;
        S.stats.score_segments += P.n_seg;
        if (P.n_repair == 0xffffffffu) S.stats.score_slow_piles++;  // marker left by the int64 kernel
        else S.stats.score_repairs += P.n_repair;
    }
    {  // unpack the walk of every pile (one step per consensus position): piles dealt to the context's host threads
        std::atomic<size_t> next(0);
        auto work = [&] {
            for (;;) {
                const size_t p = next.fetch_add(1);
                if (p >= np) break;
                if (bad_pile[p]) continue;
                const PileDev &P = piles[p];
                MainPile &M = *mp[p];
                M.path.resize(P.path_len);
                const PathItem *src = hpath + P.path_off;
                for (uint32_t k = 0; k < P.path_len; k++) {
                    PathStep &d = M.path[k];
                    d.t_pos = tag_tpos(src[k].tag);
                    d.delta = (uint16_t)tag_delta(src[k].tag);
                    d.base = (uint8_t)tag_base(src[k].tag);
                    d.link = src[k].link;
                    d.cov = src[k].cov;
                }
            }
        };
        if (S.host_threads <= 1 || np < 4) {
            work();
        } else {
            CoreLease lease(S.host_threads);
            const size_t nth = std::min<size_t>((size_t)lease.n, np);
            std::vector<std::thread> th;
            for (size_t t = 1; t < nth; t++) th.emplace_back(work);
            work();
            for (auto &x : th) x.join();
        }
    }
    g_prof.m_post += wall_ns() - tp4;
}

void DeviceAligner::run_extract(ExtractPile **ep, size_t n) {
    State &S = *s_;
    std::unique_lock<std::mutex> dbg_lock;
    if (g_debug_exclusive) dbg_lock = std::unique_lock<std::mutex>(g_dbg_mu);
    std::lock_guard<std::mutex> lock(S.mu);
    HIP_CHECK(hipSetDevice(S.device));
    hipStream_t st = S.stream;
    std::vector<RegionDev> regs;
    for (size_t i = 0; i < n; i++)
        for (RegionReq &r : ep[i]->regions) {
            RegionDev g;
            memset(&g, 0, sizeof(g));
            g.pile = (uint32_t)ep[i]->slot;
            g.start = r.start;
            g.end = r.end;
            g.max_len = r.max_len;
            g.max_len0 = r.max_len0 ? r.max_len0 : r.max_len;
            regs.push_back(g);
        }
    if (regs.empty()) return;
    S.d_regions.reserve(regs.size());
    S.d_cursor.reserve(2);
    // first guess of the string pool: 64 MB, or less when the regions cannot produce that much (<= 40 candidates of <= max_len
    // characters each); the kernel reports what it needed and a pool that was too small is retaken at the exact size
    size_t bound = (size_t)1 << 20;
    for (const RegionDev &g : regs) bound += (size_t)40 * ((size_t)g.max_len + 64);
    size_t cap = std::max<size_t>(S.d_strpool.cap, std::min<size_t>((size_t)64 << 20, bound));
    std::vector<char> hstr;
    for (;;) {
        S.d_strpool.reserve(cap);
        cap = S.d_strpool.cap;
        S.h2d(S.d_regions.p, regs.data(), regs.size() * sizeof(RegionDev), st);
        HIP_CHECK(hipMemsetAsync(S.d_cursor.p, 0, sizeof(unsigned long long), st));
        HIP_CHECK(hipEventRecord(S.evs[5], st));
        NDGPU_DBG(st, "extract: %zu regions", regs.size());
        launch_extract(S.d_piles.p, S.d_reads.p, S.d_acc.p, S.d_tags.p, S.d_colidx.p, S.d_regions.p, S.d_strpool.p,
                       S.d_cursor.p, (unsigned long long)cap, (int)regs.size(), st);
        HIP_CHECK(hipEventRecord(S.evs[6], st));
        unsigned long long used = 0;
        S.d2h(&used, S.d_cursor.p, sizeof(used), st);
        S.sync_drain(st);
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, S.evs[5], S.evs[6]));
        S.stats.extract_ms += ms;
        if (used <= cap) {
            hstr.resize((size_t)used + 1);
            S.d2h(regs.data(), S.d_regions.p, regs.size() * sizeof(RegionDev), st);
            if (used) S.d2h(hstr.data(), S.d_strpool.p, (size_t)used, st);
            S.sync_drain(st);
            break;
        }
        cap = (size_t)used + ((size_t)16 << 20);  // pool too small: rerun with the exact size
    }
    std::vector<size_t> first(n + 1, 0);
    for (size_t i = 0; i < n; i++) first[i + 1] = first[i] + ep[i]->regions.size();
    auto fill = [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            size_t k = first[i];
            for (RegionReq &r : ep[i]->regions) {
                const RegionDev &g = regs[k++];
                r.n_large = g.n_large;
                r.cands.resize(g.n_ok);
                r.cand_rank.resize(g.n_ok);
                for (uint32_t c = 0; c < g.n_ok; c++) {
                    r.cands[c].assign(hstr.data() + g.cand_off[c], g.cand_len[c]);
                    r.cand_rank[c] = g.cand_rank[c];
                }
            }
        }
    };
    if (n < 32 || S.host_threads <= 1) {
        fill(0, n);
    } else {
        CoreLease lease(S.host_threads);
        const size_t nt = std::min<size_t>((size_t)lease.n, n / 8);
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt; t++) th.emplace_back(fill, n * t / nt, n * (t + 1) / nt);
        for (auto &x : th) x.join();
    }
}

}  // namespace ndgpu
