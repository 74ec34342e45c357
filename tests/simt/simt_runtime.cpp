// TEST INFRASTRUCTURE ONLY (see include/hip/hip_runtime.h): the fibre scheduler behind the lane-accurate interpreter and the
// host-heap stand-ins for the HIP runtime calls the consensus library makes.
#include <hip/hip_runtime.h>

#include <signal.h>
#include <sys/mman.h>
#include <ucontext.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

namespace simt {

namespace {

constexpr size_t kStack = 256u << 10;

enum State { RUNNABLE, WAIT_WAVE, WAIT_BLOCK, DONE };

// Context switch between the scheduler and a lane.  swapcontext() costs a signal-mask system call per switch, which is most of
// the interpreter's run time, so on x86-64 the switch is the six callee-saved registers and the stack pointer.
#if defined(__x86_64__)
extern "C" void simt_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl simt_switch
.type simt_switch,@function
simt_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size simt_switch,.-simt_switch
)");
struct Context {
    void *sp;
};
#else
struct Context {
    ucontext_t uc;
};
#endif

struct Fibre {
    Lane lane;
    Context ctx;
    State state;
    char *stack;
};

struct Wave {
    uint64_t slot[64];
    uint64_t snap[64];
    uint64_t snap_mask;
    uint64_t live, arrived;
};

}  // namespace

struct Block {
    BlockIds ids;
    std::vector<Fibre> fibres;
    std::vector<Wave> waves;
    Context sched;
    const std::function<void()> *body;
    unsigned live;
};

thread_local Lane *g_lane = nullptr;
static thread_local Block *g_block = nullptr;
static std::atomic<int> g_schedule{getenv("SIMT_SCHEDULE") ? atoi(getenv("SIMT_SCHEDULE")) : 0};
static std::atomic<int> g_lanes_descending{getenv("SIMT_LANES_DESCENDING") ? atoi(getenv("SIMT_LANES_DESCENDING")) : 0};

const BlockIds &block_ids() { return g_block->ids; }

static inline void switch_context(Context &from, Context &to) {
#if defined(__x86_64__)
    simt_switch(&from.sp, to.sp);
#else
    swapcontext(&from.uc, &to.uc);
#endif
}

static void yield_to_scheduler() {
    Fibre *f = (Fibre *)((char *)g_lane - offsetof(Fibre, lane));
    switch_context(f->ctx, g_block->sched);
}

Snap wave_sync(uint64_t v) {
    Lane *L = g_lane;
    Wave &W = g_block->waves[L->wave];
    W.slot[L->lane] = v;
    W.arrived |= 1ull << L->lane;
    ((Fibre *)((char *)L - offsetof(Fibre, lane)))->state = WAIT_WAVE;
    yield_to_scheduler();
    return Snap{W.snap, W.snap_mask};
}

void block_sync() {
    Lane *L = g_lane;
    ((Fibre *)((char *)L - offsetof(Fibre, lane)))->state = WAIT_BLOCK;
    yield_to_scheduler();
}

static void trampoline() {
    Block *B = g_block;
    (*B->body)();
    Fibre *f = (Fibre *)((char *)g_lane - offsetof(Fibre, lane));
    f->state = DONE;
    B->waves[f->lane.wave].live &= ~(1ull << f->lane.lane);
    B->live--;
    for (;;) switch_context(f->ctx, B->sched);  // a finished lane is never resumed
}

namespace {

struct StackPool {
    std::vector<char *> stacks;
    ~StackPool() {
        for (char *s : stacks) munmap(s, kStack);
    }
    char *get(size_t i) {
        while (stacks.size() <= i) {
            void *p = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (p == MAP_FAILED) {
                perror("simt: mmap of a lane stack");
                abort();
            }
            stacks.push_back((char *)p);
        }
        return stacks[i];
    }
};

void run_block(Block &B, StackPool &pool, unsigned n_threads) {
    const unsigned n_waves = (n_threads + 63) / 64;
    B.fibres.resize(n_threads);
    B.waves.assign(n_waves, Wave{});
    B.live = n_threads;
    g_block = &B;
    for (unsigned t = 0; t < n_threads; t++) {
        Fibre &f = B.fibres[t];
        f.lane.tid = t;
        f.lane.wave = t / 64;
        f.lane.lane = t % 64;
        f.lane.blk = &B;
        f.lane.tidx = dim3(t % B.ids.bdim.x, (t / B.ids.bdim.x) % B.ids.bdim.y, t / (B.ids.bdim.x * B.ids.bdim.y));
        f.state = RUNNABLE;
        f.stack = pool.get(t);
#if defined(__x86_64__)
        {   // what simt_switch pops: six registers, then `ret` into trampoline with the stack as after a call
            void **top = (void **)(f.stack + kStack);
            top[-1] = nullptr;
            top[-2] = (void *)trampoline;
            for (int r = 3; r <= 8; r++) top[-r] = nullptr;
            f.ctx.sp = (void *)(top - 8);
        }
#else
        getcontext(&f.ctx.uc);
        f.ctx.uc.uc_stack.ss_sp = f.stack;
        f.ctx.uc.uc_stack.ss_size = kStack;
        f.ctx.uc.uc_link = &B.sched.uc;
        makecontext(&f.ctx.uc, trampoline, 0);
#endif
        B.waves[f.lane.wave].live |= 1ull << f.lane.lane;
    }
    // Which wavefront runs next is not defined by the programming model.  Schedule 0 gives every wavefront one slice in turn;
    // 1 / 2 let the lowest / highest numbered wavefront that can run keep running (it runs ahead of the others up to its next
    // workgroup barrier), >= 3 picks at random with that seed.  A kernel that lacks a barrier between wavefronts passes under
    // some of these and fails under others.
    const int policy = g_schedule.load();
    uint64_t rng = 0x9e3779b97f4a7c15ull * (uint64_t)(policy + 1) + B.ids.bidx.x;
    auto run_wave = [&](unsigned w) {
        bool any = false;
        for (;;) {
            bool ran = false;
            const unsigned t1 = std::min(n_threads, w * 64 + 64);
            for (unsigned k = w * 64; k < t1; k++) {
                // (the lanes of a wavefront between two rendezvous points run one after the other: lowest first, or highest first
                // -- code that relies on lock step without saying so passes under at most one of the two orders)
                const unsigned t = g_lanes_descending.load() ? t1 - 1 - (k - w * 64) : k;
                Fibre &f = B.fibres[t];
                if (f.state != RUNNABLE) continue;
                g_lane = &f.lane;
                switch_context(B.sched, f.ctx);
                ran = true;
            }
            Wave &W = B.waves[w];
            if (W.arrived && W.arrived == W.live) {
                memcpy(W.snap, W.slot, sizeof(W.snap));
                W.snap_mask = W.arrived;
                for (uint64_t m = W.arrived; m; m &= m - 1) B.fibres[w * 64 + (unsigned)__builtin_ctzll(m)].state = RUNNABLE;
                W.arrived = 0;
                ran = true;
            }
            if (!ran) break;
            any = true;
            if (policy == 0) break;
        }
        return any;
    };
    while (B.live) {
        bool progressed = false;
        if (policy >= 3) {
            rng ^= rng << 13, rng ^= rng >> 7, rng ^= rng << 17;
            const unsigned w0 = (unsigned)(rng % n_waves);
            for (unsigned i = 0; i < n_waves && !progressed; i++) progressed = run_wave((w0 + i) % n_waves);
        } else {
            for (unsigned i = 0; i < n_waves; i++) {
                if (run_wave(policy == 2 ? n_waves - 1 - i : i)) {
                    progressed = true;
                    if (policy != 0) break;
                }
            }
        }
        if (B.live) {
            unsigned at_barrier = 0;
            for (unsigned t = 0; t < n_threads; t++) at_barrier += B.fibres[t].state == WAIT_BLOCK;
            if (at_barrier == B.live) {
                for (unsigned t = 0; t < n_threads; t++)
                    if (B.fibres[t].state == WAIT_BLOCK) B.fibres[t].state = RUNNABLE;
                progressed = true;
            }
        }
        if (B.live && !progressed) {
            fprintf(stderr, "simt: workgroup (%u,%u,%u) is stuck: a wave-wide operation or __syncthreads() was reached by only part "
                            "of its lanes\n", B.ids.bidx.x, B.ids.bidx.y, B.ids.bidx.z);
            for (unsigned t = 0; t < n_threads; t++)
                if (B.fibres[t].state != DONE) fprintf(stderr, "  thread %u: %s\n", t, B.fibres[t].state == WAIT_WAVE ? "wave op" : "barrier");
            abort();
        }
    }
    g_block = nullptr;
    g_lane = nullptr;
}

unsigned worker_count() {
    static const unsigned n = [] {
        const char *e = getenv("SIMT_THREADS");
        unsigned v = e ? (unsigned)atoi(e) : std::thread::hardware_concurrency();
        return v ? v : 1u;
    }();
    return n;
}

}  // namespace

static thread_local std::vector<uint64_t> g_dyn_lds;
void *dynamic_lds() { return g_dyn_lds.data(); }

static thread_local const char *g_kernel_name = nullptr;

// a wild access inside a kernel: say which kernel and which thread before dying (there is no debugger on the box)
static void on_fault(int sig, siginfo_t *info, void *) {
    char msg[256];
    int n;
    if (g_kernel_name && g_block && g_lane)
        n = snprintf(msg, sizeof(msg), "simt: signal %d (address %p) in kernel %s, workgroup (%u,%u,%u), thread %u\n", sig, info->si_addr,
                     g_kernel_name, g_block->ids.bidx.x, g_block->ids.bidx.y, g_block->ids.bidx.z, g_lane->tid);
    else n = snprintf(msg, sizeof(msg), "simt: signal %d outside a kernel (host code of the library)\n", sig);
    if (write(2, msg, (size_t)n) < 0) {}
    signal(sig, SIG_DFL);
    raise(sig);
}

void launch(const char *kernel_name, dim3 grid, dim3 block, size_t dynamic_lds_bytes, const std::function<void()> &body) {
    static const bool handlers = [] {
        struct sigaction sa;
        memset(&sa, 0, sizeof(sa));
        sa.sa_sigaction = on_fault;
        sa.sa_flags = SA_SIGINFO;
        sigaction(SIGSEGV, &sa, nullptr);
        sigaction(SIGBUS, &sa, nullptr);
        sigaction(SIGFPE, &sa, nullptr);
        return true;
    }();
    (void)handlers;
    const unsigned long long n_blocks = (unsigned long long)grid.x * grid.y * grid.z;
    const unsigned n_threads = block.x * block.y * block.z;
    if (!n_blocks || !n_threads) return;
    std::atomic<unsigned long long> next{0};
    auto work = [&]() {
        StackPool pool;
        Block B;
        g_kernel_name = kernel_name;
        g_dyn_lds.assign(dynamic_lds_bytes / 8 + 1, 0);
        B.body = &body;
        B.ids.bdim = block;
        B.ids.gdim = grid;
        for (;;) {
            const unsigned long long b = next.fetch_add(1);
            if (b >= n_blocks) break;
            B.ids.bidx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((unsigned long long)grid.x * grid.y)));
            run_block(B, pool, n_threads);
        }
    };
    const unsigned n_workers = (unsigned)std::min<unsigned long long>(worker_count(), n_blocks);
    if (n_workers <= 1) {
        // a launch from inside a running fibre is not a thing; from a plain host thread run in place
        work();
        return;
    }
    std::vector<std::thread> th;
    for (unsigned i = 0; i < n_workers; i++) th.emplace_back(work);
    for (auto &t : th) t.join();
}

}  // namespace simt

// ------------------------------------------------------------------ runtime API
namespace {

std::atomic<size_t> g_allocated{0};

size_t pretend_total() {
    const char *e = getenv("SIMT_DEVICE_GB");
    return (size_t)(e ? atof(e) : 24.0) * (1ull << 30);
}

std::mutex g_alloc_mu;
std::unordered_map<void *, size_t> g_alloc_size;

// exactly n bytes, 256-byte aligned like hipMalloc: under AddressSanitizer (SIMT_SANITIZE=address) an access one byte past a device
// buffer is reported
void *alloc_tracked(size_t n) {
    if (n == 0) n = 1;
    if (g_allocated.load() + n > pretend_total()) return nullptr;
    void *p = nullptr;
    if (posix_memalign(&p, 256, n) != 0 || !p) return nullptr;
    // device memory comes uninitialised: poison it, so that a kernel that relies on zeroes it never wrote fails here too
    // (large buffers: the first and last megabyte only -- the tests' working sets are far smaller than the planned capacities)
    static const bool poison = !getenv("SIMT_NO_POISON");
    if (poison) {
        const size_t edge = 1u << 20;
        if (n <= 2 * edge) memset(p, 0xcd, n);
        else memset(p, 0xcd, edge), memset((char *)p + n - edge, 0xcd, edge);
    }
    {
        std::lock_guard<std::mutex> lock(g_alloc_mu);
        g_alloc_size[p] = n;
    }
    g_allocated += n;
    return p;
}

void free_tracked(void *q) {
    if (!q) return;
    {
        std::lock_guard<std::mutex> lock(g_alloc_mu);
        auto it = g_alloc_size.find(q);
        if (it != g_alloc_size.end()) {
            g_allocated -= it->second;
            g_alloc_size.erase(it);
        }
    }
    free(q);
}

}  // namespace

struct simtStream { int unused; };
struct simtEvent { std::chrono::steady_clock::time_point t; };

const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : "error"; }
const char *hipGetErrorName(hipError_t e) { return e == hipSuccess ? "hipSuccess" : e == hipErrorOutOfMemory ? "hipErrorOutOfMemory" : "hipError"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    memset(p, 0, sizeof(*p));
    strcpy(p->name, "simt interpreter");
    strcpy(p->gcnArchName, "gfx950");
    p->totalGlobalMem = pretend_total();
    p->multiProcessorCount = 256;
    return hipSuccess;
}
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) {
    const size_t t = pretend_total(), a = g_allocated.load();
    *total_b = t;
    *free_b = a < t ? t - a : 0;
    return hipSuccess;
}
hipError_t hipMalloc(void **p, size_t n) { *p = alloc_tracked(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { free_tracked(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) { *s = new simtStream{0}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { return hipStreamCreate(s); }
hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, unsigned, const unsigned *) { return hipStreamCreate(s); }
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new simtEvent{std::chrono::steady_clock::now()}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipSetDeviceFlags(unsigned) { return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}

// ------------------------------------------------------------------ self-test of the interpreter (tests/test_simt_kernels.py)
namespace {

unsigned mix(unsigned x) {
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}

void selftest_kernel(int *bad) {
    __shared__ int lds[192];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto fail = [&](int code) { atomicOr(bad, 1 << code); };
    auto val = [&](int l) { return (int)(mix((unsigned)(wave * 64 + l) + 17u * blockIdx.x) >> 4); };
    const int v = val(lane);

    if ((int)__ballot(lane % 3 == 0) != (int)0x49249249 || __ballot(lane % 3 == 0) != 0x9249249249249249ull) fail(0);
    if (__shfl(v, 63 - lane, 64) != val(63 - lane)) fail(1);
    if (__shfl(v, lane + 5, 16) != val((lane & ~15) + ((lane + 5) & 15))) fail(1);
    if (__shfl_up(v, 3u, 64) != (lane >= 3 ? val(lane - 3) : v)) fail(2);
    if (__shfl_down(v, 7u, 64) != (lane + 7 < 64 ? val(lane + 7) : v)) fail(2);
    if (__shfl_xor(v, 32, 64) != val(lane ^ 32)) fail(3);
    if (__builtin_amdgcn_readlane(v, 41) != val(41)) fail(4);

    const int old = -1;
    if (__builtin_amdgcn_update_dpp(old, v, 0xb1, 0xf, 0xf, false) != val(lane ^ 1)) fail(5);
    if (__builtin_amdgcn_update_dpp(old, v, 0x4e, 0xf, 0xf, false) != val(lane ^ 2)) fail(5);
    if (__builtin_amdgcn_update_dpp(old, v, 0x141, 0xf, 0xf, false) != val((lane & ~7) | (7 - (lane & 7)))) fail(6);
    if (__builtin_amdgcn_update_dpp(old, v, 0x140, 0xf, 0xf, false) != val((lane & ~15) | (15 - (lane & 15)))) fail(6);
    if (__builtin_amdgcn_update_dpp(old, v, 0x142, 0xa, 0xf, false) != (((lane >> 4) & 1) ? val((lane & ~15) - 1) : old)) fail(7);
    if (__builtin_amdgcn_update_dpp(old, v, 0x143, 0xc, 0xf, false) != ((lane >> 5) ? val(31) : old)) fail(7);
    if (__builtin_amdgcn_update_dpp(old, v, 0x111, 0xf, 0xf, false) != ((lane & 15) ? val(lane - 1) : old)) fail(8);
    if (__builtin_amdgcn_update_dpp(old, v, 0x111, 0xf, 0xf, true) != ((lane & 15) ? val(lane - 1) : 0)) fail(8);
    if (__builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false) != (lane ? val(lane - 1) : old)) fail(8);

    // workgroup barrier: every thread sees what every other thread wrote before it
    lds[tid] = v + wave;
    __syncthreads();
    const int o = (tid + 65) % 192;
    if (lds[o] != (int)(mix((unsigned)o + 17u * blockIdx.x) >> 4) + (o >> 6)) fail(9);
    __syncthreads();

    // lanes that have left the kernel do not take part in wave-wide operations, and do not hold up the rest
    if (lane >= 40 && wave == 1) return;
    const unsigned long long m = __ballot(true);
    if (m != (wave == 1 ? (1ull << 40) - 1 : ~0ull)) fail(10);
    __syncthreads();
    lds[tid] = tid;

    __builtin_amdgcn_wave_barrier();
    if (lds[tid ^ 1] != (tid ^ 1) && !(wave == 1 && (lane ^ 1) >= 40)) fail(11);
}

}  // namespace

extern "C" void simt_set_schedule(int policy) { simt::g_schedule.store(policy); }
extern "C" void simt_set_lane_order(int descending) { simt::g_lanes_descending.store(descending); }

extern "C" int simt_selftest() {
    int *bad = nullptr;
    if (hipMalloc((void **)&bad, sizeof(int)) != hipSuccess) return -1;
    *bad = 0;
    hipLaunchKernelGGL(selftest_kernel, dim3(37), dim3(192), 0, (hipStream_t) nullptr, bad);
    const int r = *bad;
    hipFree(bad);
    return r;
}
